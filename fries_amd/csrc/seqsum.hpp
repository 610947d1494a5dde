// Bit-exact parallel evaluation of a left-to-right floating-point running sum
//     S_i = fl(S_{i-1} + a_i),  a_i >= 0,
// which is what the reference computes in `lbound += ...` (FRIES/compress_utils.cpp:314, :739)
// and in its in-order norms (:35, :82, :100, :262, :272).  A tree sum differs from it by ~sqrt(N)
// ulps, enough to move a systematic sample across an element boundary once N ~ 1e5.
//
// Why it parallelises: while S stays inside one binade [2^e, 2^(e+1)) it is an integer multiple
// M of ulp = 2^(e-52), and adding a_i adds the integer RN(a_i / ulp) -- except for exact ties,
// where round-half-even makes the increment depend on the parity of M.  So inside a binade the
// running sum is an integer prefix sum of per-element "parity maps" (d0, d1) = increment if M is
// even / odd, and maps compose associatively:  (g then f)_p = g_p + f_{(p + g_p) & 1}.
// The binade of every partial sum is known in advance from an ordinary (inexact) prefix sum
// with a rigorous error margin; only the few stretches that may straddle a power of two
// ("dirty") are added element by element.
//
// Granularity: tile = 1024 elements (one workgroup), sub-tile = 64 elements.  Dirty tiles are
// resolved at sub-tile level, dirty sub-tiles sequentially.  The positive sequence crosses
// about log2(total / first term) binades, so only a few dozen sub-tiles are ever sequential.
#pragma once
#include "fries_dev.hpp"

#define FR_SUBS_PER_TILE 16
#define FR_SEQ_TILE 1024

struct SeqRec {
    double approx;      // tile / sub-tile sum in tree order
    double carry;       // classify: approximate, chain: exact running sum entering the unit
    long long d0, d1;   // composed parity map of the unit (valid when clean)
    int e;              // binade exponent of every partial sum inside the unit (valid when clean)
    uint32_t dirty;
};
struct SeqWork {
    SeqRec *tiles;      // [FR_MAX_PART]
    SeqRec *subs;       // [FR_MAX_PART * 16]
    double *total;      // [1] exact sum of everything (+ start)
    double *tsum;       // [FR_MAX_PART] the tiles' tree sums again, contiguous (k_seq_maps adds up the ones before its tile)
#ifdef FR_SEQ_TIMING
    int dbg = 0;        // FRIES_SEQ_DBG: k_seq_chain prints where its time goes (build with -DFR_SEQ_TIMING)
#else
    static constexpr int dbg = 0;
#endif
};

// where a chain starts: 0 + norms[0] + ... + norms[n-1], added left to right -- the lbound a rank inherits from the
// ranks before it (seed_sys, compress_utils.cpp:113-116).  n == 0: the chain starts at 0.
struct SeqStart {
    const double *norms; int n;
    __device__ double value() const { double s = 0; for (int p = 0; p < n; p++) s += norms[p]; return s; }
};
static inline SeqStart fr_seq_from_zero() { SeqStart s; s.norms = nullptr; s.n = 0; return s; }

struct PMap { long long d0, d1; };
__device__ __forceinline__ PMap fr_pm_id() { PMap m; m.d0 = 0; m.d1 = 0; return m; }
// g first, then f
__device__ __forceinline__ PMap fr_pm_compose(const PMap &g, const PMap &f) {
    PMap h;
    h.d0 = g.d0 + ((g.d0 & 1) ? f.d1 : f.d0);
    h.d1 = g.d1 + (((1 + g.d1) & 1) ? f.d1 : f.d0);
    return h;
}
// increment (in ulps of binade e) that adding `a` causes, as a parity map
__device__ __forceinline__ PMap fr_pm_elem(double a, double scale /* 2^(52-e) */) {
    PMap m;
    double x = a * scale;                // exact power-of-two scaling (or underflow to < 0.5)
    double q = floor(x), f = x - q;      // exact
    long long qi = (long long)q;
    if (f < 0.5) { m.d0 = qi; m.d1 = qi; }
    else if (f > 0.5) { m.d0 = qi + 1; m.d1 = qi + 1; }
    else { m.d0 = qi + (qi & 1); m.d1 = qi + ((qi & 1) ^ 1); }     // tie: round half to even
    return m;
}
__device__ __forceinline__ int fr_exp_of(double x) { return (int)((__double_as_longlong(x) >> 52) & 0x7ff) - 1023; }

// rigorous relative error bound for both the sequential and the tree sum of n non-negative terms
__device__ __forceinline__ double fr_seq_eps(unsigned n) { return 4.0 * ((double)n + 2048.0) * 1.1102230246251565e-16; }

// classify a unit whose exact running sum enters within [lo_c, hi_c] = carry*(1 -+ eps) and
// leaves within sum*(1 -+ eps): clean iff all of it lies strictly inside one binade
__device__ __forceinline__ bool fr_seq_clean(double carry_apx, double leave_apx, double eps, int *e) {
    double lo = carry_apx * (1.0 - eps), hi = leave_apx * (1.0 + eps);
    if (hi == 0) { *e = 0; return true; }        // nothing but zeros so far: every map is the identity
    if (!(lo > 0) || !(hi < 1.0e300)) return false;
    if (lo < 2.3e-308) return false;             // keep away from subnormals
    int el = fr_exp_of(lo), eh = fr_exp_of(hi);
    *e = el;
    return el == eh && hi < ldexp(1.0, el + 1);
}

__device__ __forceinline__ double fr_seq_apply_map(double carry, int e, long long d0, long long d1) {
    // carry in binade e (every clean unit that is not all zeros): its integer multiple of the ulp is its mantissa with the implicit
    // bit, and the result, still inside the binade, is that integer's low 52 bits under the same exponent -- a dozen integer
    // instructions instead of two 64-bit float <-> integer conversions (~50) on the one wave that walks the chain
    const long long cb = __double_as_longlong(carry);
    if (((cb >> 52) & 0x7ffll) == (long long)(e + 1023)) {
        const long long Mi = (cb & 0xFFFFFFFFFFFFFll) | (1ll << 52);
        const long long R = Mi + ((Mi & 1) ? d1 : d0);
        if (R >= (1ll << 52) && R < (1ll << 53)) return __longlong_as_double(((long long)(e + 1023) << 52) | (R & 0xFFFFFFFFFFFFFll));
    }
    double ulp = ldexp(1.0, e - 52);
    long long M = (long long)(carry * ldexp(1.0, 52 - e));    // exact integer
    long long d = (M & 1) ? d1 : d0;
    return (double)(M + d) * ulp;
}

// ---- S1: tree sums per tile and per 64-element sub-tile
template <class Acc>
__global__ void __launch_bounds__(FR_BLOCK) k_seq_sums(SeqWork Q, Acc acc) {
    __shared__ double shd[4];
    const unsigned n = acc.count();
    const unsigned ntile = (n + FR_SEQ_TILE - 1) / FR_SEQ_TILE;
    if (blockIdx.x >= ntile) return;
    size_t base = (size_t)blockIdx.x * FR_SEQ_TILE + (size_t)threadIdx.x * 4;
    double s = 0;
    for (int it = 0; it < 4; it++) { size_t i = base + it; if (i < n) s += acc.get(i); }
    double g = s;                                   // 16-lane groups = sub-tiles
    for (int off = 8; off > 0; off >>= 1) g += __shfl_xor(g, off);
    if ((threadIdx.x & 15) == 0) Q.subs[(size_t)blockIdx.x * FR_SUBS_PER_TILE + (threadIdx.x >> 4)].approx = g;
    double t = fr_block_sum(s, shd);
    if (threadIdx.x == 0) { Q.tiles[blockIdx.x].approx = t; Q.tsum[blockIdx.x] = t; }
}

// ---- S3: parity maps of clean tiles; sub-tile classification + maps inside dirty tiles
// The tile's approximate carry and its clean / dirty classification are formed here as well (they were a one-workgroup launch of their own,
// k_seq_classify): the workgroup adds up the tree sums of the tiles before its own -- any fixed order of n non-negative terms stays inside
// fr_seq_eps(n), and which tiles get classified "may straddle" only decides who is walked sub-tile by sub-tile, never the sum.
template <class Acc>
__global__ void __launch_bounds__(FR_BLOCK) k_seq_maps(SeqWork Q, Acc acc, SeqStart st) {
    __shared__ PMap shm[FR_BLOCK];
    __shared__ double shc[4];
    const unsigned n = acc.count();
    const unsigned ntile = (n + FR_SEQ_TILE - 1) / FR_SEQ_TILE;
    if (blockIdx.x >= ntile) return;
    SeqRec tr;
    {
        double x = 0;
        for (unsigned i = threadIdx.x; i < blockIdx.x; i += FR_BLOCK) x += Q.tsum[i];
        const double c_in = st.value() + fr_block_sum(x, shc);
        int e_t = 0;
        const bool clean_t = fr_seq_clean(c_in, c_in + Q.tsum[blockIdx.x], fr_seq_eps(n), &e_t);
        tr.carry = c_in; tr.e = e_t; tr.dirty = clean_t ? 0u : 1u;
        if (threadIdx.x == 0) { Q.tiles[blockIdx.x].carry = c_in; Q.tiles[blockIdx.x].e = e_t; Q.tiles[blockIdx.x].dirty = tr.dirty; }
    }
    size_t base = (size_t)blockIdx.x * FR_SEQ_TILE + (size_t)threadIdx.x * 4;
    double a[4];
    for (int it = 0; it < 4; it++) { size_t i = base + it; a[it] = i < n ? acc.get(i) : 0.0; }
    if (!tr.dirty) {
        double scale = ldexp(1.0, 52 - tr.e);
        PMap m = fr_pm_elem(a[0], scale);
        for (int it = 1; it < 4; it++) m = fr_pm_compose(m, fr_pm_elem(a[it], scale));
        shm[threadIdx.x] = m;
        __syncthreads();
        for (int stride = 1; stride < FR_BLOCK; stride <<= 1) {        // ordered tree: left operand first
            PMap r;
            bool act = (threadIdx.x % (2 * stride)) == 0 && threadIdx.x + stride < FR_BLOCK;
            if (act) r = fr_pm_compose(shm[threadIdx.x], shm[threadIdx.x + stride]);
            __syncthreads();
            if (act) shm[threadIdx.x] = r;
            __syncthreads();
        }
        if (threadIdx.x == 0) { Q.tiles[blockIdx.x].d0 = shm[0].d0; Q.tiles[blockIdx.x].d1 = shm[0].d1; }
        return;
    }
    // dirty tile: approximate carry of each sub-tile, classification, maps of the clean ones
    const int sub = threadIdx.x >> 4, sl = threadIdx.x & 15;
    SeqRec *sr = &Q.subs[(size_t)blockIdx.x * FR_SUBS_PER_TILE + sub];
    double c_in = tr.carry;
    for (int j = 0; j < sub; j++) c_in += Q.subs[(size_t)blockIdx.x * FR_SUBS_PER_TILE + j].approx;
    double c_out = c_in + sr->approx;
    int e = 0;
    bool clean = fr_seq_clean(c_in, c_out, fr_seq_eps(n), &e);
    PMap m = fr_pm_id();
    if (clean) {
        double scale = ldexp(1.0, 52 - e);
        m = fr_pm_elem(a[0], scale);
        for (int it = 1; it < 4; it++) m = fr_pm_compose(m, fr_pm_elem(a[it], scale));
    }
    shm[threadIdx.x] = m;
    __syncthreads();
    if (sl == 0) {
        PMap r = shm[threadIdx.x];
        for (int k = 1; k < 16; k++) r = fr_pm_compose(r, shm[threadIdx.x + k]);
        sr->carry = c_in; sr->e = e; sr->dirty = clean ? 0u : 1u; sr->d0 = r.d0; sr->d1 = r.d1;
    }
}

// ---- S4: the chain.  One wave.  Tile records are fetched 64 at a time (the next batch is in flight while the
// current one is consumed); inside a batch, runs of clean same-binade tiles are resolved by an ordered scan of parity
// maps over lanes, a tile that may straddle a power of two is walked through its 16 sub-tiles (fetched together), and a
// sub-tile that may straddle one is added element by element from registers (v_readlane broadcasts).
__device__ __forceinline__ double fr_bcast_f64(double v, int src_lane /* wave-uniform */) {
    long long b = __double_as_longlong(v);
    uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, src_lane);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), src_lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ long long fr_bcast_i64(long long b, int src_lane) {
    uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, src_lane);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), src_lane);
    return (long long)(((unsigned long long)hi << 32) | lo);
}

// inclusive prefix sum of one int64 per lane over the wave: Kogge-Stone inside rows of 16 lanes (DPP row shifts, zeros
// shifted in), then the row totals are broadcast down (row_bcast15 / row_bcast31) -- no LDS crossbar traffic
template <int CTRL, int ROW_MASK> __device__ __forceinline__ long long fr_dpp_i64z(long long v) {
    uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, ROW_MASK, 0xf, true);
    uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)((unsigned long long)v >> 32), CTRL, ROW_MASK, 0xf, true);
    return (long long)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ long long fr_wave_incl_i64(long long v) {
    v += fr_dpp_i64z<0x111, 0xf>(v);       // row_shr:1
    v += fr_dpp_i64z<0x112, 0xf>(v);       // row_shr:2
    v += fr_dpp_i64z<0x114, 0xf>(v);       // row_shr:4
    v += fr_dpp_i64z<0x118, 0xf>(v);       // row_shr:8
    v += fr_dpp_i64z<0x142, 0xa>(v);       // row_bcast15 -> rows 1 and 3
    v += fr_dpp_i64z<0x143, 0xc>(v);       // row_bcast31 -> rows 2 and 3
    return v;
}

// What the chain needs about the tiles that may straddle a power of two, staged in LDS by the whole workgroup before wave 0
// starts walking: their sub-tile records and the 64 elements of every sub-tile that has to be added one by one.  Fetched on
// demand by the one walking wave, each of these costs a dependent global-load latency (~2 us) -- with ~15 such tiles per sum
// that was most of the kernel's 70 us.
#define FR_SEQ_DT_CAP 64            // dirty tiles staged (more are handled from global memory, correct but slow)
#define FR_SEQ_DS_CAP 64            // dirty sub-tiles whose elements are staged
struct SeqSubL { long long d0, d1; int e; uint32_t dirty; int slot; int pad; };
struct SeqStage {
    uint32_t n_dt, n_ds;
    uint32_t dt_list[FR_SEQ_DT_CAP];
    SeqSubL subs[FR_SEQ_DT_CAP * FR_SUBS_PER_TILE];
    double elems[FR_SEQ_DS_CAP * 64];
    uint32_t scan[8];
};

// all FR_BLOCK threads: ordered list of the first FR_SEQ_DT_CAP dirty tiles, their sub-tile records, the elements of their dirty sub-tiles
template <class Acc>
__device__ __forceinline__ void fr_seq_stage(const SeqWork &Q, const Acc &acc, SeqStage *sh) {
    const unsigned n = acc.count();
    const unsigned ntile = (n + FR_SEQ_TILE - 1) / FR_SEQ_TILE;
    const unsigned per = (ntile + FR_BLOCK - 1) / FR_BLOCK;
    const unsigned lo = threadIdx.x * per, hi = lo + per < ntile ? lo + per : ntile;
    uint32_t cnt = 0;
    for (unsigned t = lo; t < hi; t++) cnt += Q.tiles[t].dirty ? 1u : 0u;
    uint32_t tot;
    uint32_t o = fr_block_scan_u32(cnt, sh->scan, &tot) - cnt;
    for (unsigned t = lo; t < hi && o < FR_SEQ_DT_CAP; t++) if (Q.tiles[t].dirty) sh->dt_list[o++] = t;
    if (threadIdx.x == 0) { sh->n_dt = tot < FR_SEQ_DT_CAP ? tot : FR_SEQ_DT_CAP; sh->n_ds = 0; }
    __syncthreads();
    const unsigned n_dt = sh->n_dt;
    for (unsigned i = threadIdx.x; i < n_dt * FR_SUBS_PER_TILE; i += FR_BLOCK) {
        const unsigned t = sh->dt_list[i / FR_SUBS_PER_TILE], j = i % FR_SUBS_PER_TILE;
        const SeqRec r = Q.subs[(size_t)t * FR_SUBS_PER_TILE + j];
        SeqSubL l; l.d0 = r.d0; l.d1 = r.d1; l.e = r.e; l.dirty = r.dirty; l.slot = -1; l.pad = 0;
        if (r.dirty && (size_t)t * FR_SEQ_TILE + (size_t)j * 64 < n) {
            const uint32_t slot = atomicAdd(&sh->n_ds, 1u);
            if (slot < FR_SEQ_DS_CAP) l.slot = (int)slot;
        }
        sh->subs[i] = l;
    }
    __syncthreads();
    // Runs of clean sub-tiles in one binade, composed ahead of the walk: every clean sub-tile's (d0, d1) becomes the map of its run up to
    // and including itself (one thread per staged tile, 16 compositions), so that the walking wave reads a run's map instead of scanning for it
    if (threadIdx.x < n_dt) {
        SeqSubL *q = &sh->subs[threadIdx.x * FR_SUBS_PER_TILE];
        PMap run = fr_pm_id(); int run_e = 0; bool in_run = false;
        for (int j = 0; j < FR_SUBS_PER_TILE; j++) {
            if (q[j].dirty) { in_run = false; continue; }
            PMap m; m.d0 = q[j].d0; m.d1 = q[j].d1;
            if (in_run && q[j].e == run_e) run = fr_pm_compose(run, m);
            else { run = m; run_e = q[j].e; in_run = true; }
            q[j].d0 = run.d0; q[j].d1 = run.d1;
        }
    }
    for (unsigned i = threadIdx.x; i < n_dt * FR_SUBS_PER_TILE * 64; i += FR_BLOCK) {
        const unsigned si = i >> 6, k = i & 63;
        const int slot = sh->subs[si].slot;         // uniform over the wave: 64 consecutive i share one sub-tile
        if (slot < 0) continue;
        const unsigned t = sh->dt_list[si / FR_SUBS_PER_TILE], j = si % FR_SUBS_PER_TILE;
        const size_t e = (size_t)t * FR_SEQ_TILE + (size_t)j * 64 + k;
        sh->elems[slot * 64 + k] = e < n ? acc.get(e) : 0.0;
    }
    __syncthreads();
}

#define FR_SEQ_LCHUNK 2048          // tile records staged per round
struct SeqTileL { long long d0, d1; int e; uint32_t dirty; };

// wave 0: walks tiles [c0, c1) whose records sit in tl[0, c1 - c0); *di_io = dirty tiles met so far == index into the staged list
template <class Acc>
__device__ __forceinline__ double fr_seq_chain_wave(SeqWork Q, Acc acc, double start, const SeqStage *sh, const SeqTileL *tl, unsigned c0, unsigned c1, unsigned *di_io, unsigned long long *t_dirty = nullptr) {
    const unsigned n = acc.count();
    const int lane = fr_lane();
    double carry = start;
    unsigned di = *di_io;
    const unsigned n_dt = sh->n_dt;
    for (unsigned t0 = c0; t0 < c1; t0 += 64) {
        SeqTileL r;
        r.dirty = 1; r.e = 0; r.d0 = r.d1 = 0;
        if (t0 + lane < c1) r = tl[t0 - c0 + lane];
        const int n_here = (c1 - t0) < 64u ? (int)(c1 - t0) : 64;
        int pos = 0;
        while (pos < n_here) {
            const int e0 = __builtin_amdgcn_readlane(r.e, pos);
            const unsigned dflag = (unsigned)__builtin_amdgcn_readlane((int)r.dirty, pos);
            const unsigned t = t0 + pos;
            if (dflag) {
                const unsigned long long td0 = Q.dbg ? wall_clock64() : 0ull;
                // tile t may straddle a power of two: walk its sub-tiles -- runs of clean sub-tiles in one binade are composed by
                // a scan over lanes (as for tiles), only a sub-tile that may straddle one is added element by element
                const bool staged = di < n_dt;          // then sh->dt_list[di] == t
                SeqSubL sl;
                sl.dirty = 1; sl.e = 0; sl.d0 = sl.d1 = 0; sl.slot = -1; sl.pad = 0;
                if (lane < FR_SUBS_PER_TILE) {
                    if (staged) sl = sh->subs[di * FR_SUBS_PER_TILE + lane];
                    else { const SeqRec q = Q.subs[(size_t)t * FR_SUBS_PER_TILE + lane]; sl.d0 = q.d0; sl.d1 = q.d1; sl.e = q.e; sl.dirty = q.dirty; }
                }
                if (lane == 0) Q.tiles[t].carry = carry;
                const size_t t_lo = (size_t)t * FR_SEQ_TILE;
                const int n_sub_here = (int)(((n - t_lo) + 63) / 64 < (size_t)FR_SUBS_PER_TILE ? ((n - t_lo) + 63) / 64 : (size_t)FR_SUBS_PER_TILE);
                int jp = 0;
                while (jp < n_sub_here) {
                    const unsigned sd = (unsigned)__builtin_amdgcn_readlane((int)sl.dirty, jp);
                    if (sd) {
                        const size_t e_lo = t_lo + (size_t)jp * 64;
                        const size_t e_hi = e_lo + 64 < n ? e_lo + 64 : n;
                        const int slot = __builtin_amdgcn_readlane(sl.slot, jp);
                        if (lane == 0) Q.subs[(size_t)t * FR_SUBS_PER_TILE + jp].carry = carry;
                        double mine_a;
                        if (slot >= 0) mine_a = sh->elems[slot * 64 + lane];
                        else mine_a = (e_lo + lane < e_hi) ? acc.get(e_lo + lane) : 0.0;
#pragma unroll
                        for (int k = 0; k < 64; k++) carry = carry + fr_bcast_f64(mine_a, k);
                        jp++;
                        continue;
                    }
                    const int se = __builtin_amdgcn_readlane(sl.e, jp);
                    const unsigned long long okm = __ballot(lane >= jp && lane < n_sub_here && !sl.dirty && sl.e == se);
                    const unsigned long long shm = okm >> jp;
                    const int run = (~shm == 0ull) ? 64 : (__ffsll((long long)~shm) - 1);
                    PMap m; m.d0 = sl.d0; m.d1 = sl.d1;
                    if (lane < jp || lane >= jp + run) m = fr_pm_id();
                    if (!staged)                                                        // (staged tiles carry their runs' inclusive maps already: fr_seq_stage)
                    for (int off = 1; off < FR_SUBS_PER_TILE; off <<= 1) {              // inclusive ordered scan of maps over the 16 lanes
                        PMap o; o.d0 = __shfl_up(m.d0, off); o.d1 = __shfl_up(m.d1, off);
                        if (lane >= off) m = fr_pm_compose(o, m);
                    }
                    PMap ex; ex.d0 = __shfl_up(m.d0, 1); ex.d1 = __shfl_up(m.d1, 1);
                    if (lane == 0 || lane == jp) ex = fr_pm_id();
                    if (lane >= jp && lane < jp + run) Q.subs[(size_t)t * FR_SUBS_PER_TILE + lane].carry = fr_seq_apply_map(carry, se, ex.d0, ex.d1);
                    const int last = jp + run - 1;
                    carry = fr_seq_apply_map(carry, se, fr_bcast_i64(m.d0, last), fr_bcast_i64(m.d1, last));
                    jp += run;
                }
                if (lane >= n_sub_here && lane < FR_SUBS_PER_TILE) Q.subs[(size_t)t * FR_SUBS_PER_TILE + lane].carry = carry;
                di++;
                pos++;
                if (Q.dbg && t_dirty) *t_dirty += wall_clock64() - td0;
                continue;
            }
            // run of clean tiles in binade e0 starting at lane `pos`
            const unsigned long long ok = __ballot(lane >= pos && lane < n_here && !r.dirty && r.e == e0);
            const unsigned long long sh_ = ok >> pos;
            int run = (~sh_ == 0ull) ? 64 : (__ffsll((long long)~sh_) - 1);        // lanes pos .. pos+run-1 (run >= 1)
            // (d0, d1) of a clean tile = the map of its run up to and including it (fr_seq_runs, all four waves, before the walk)
            PMap m; m.d0 = r.d0; m.d1 = r.d1;
            PMap ex;                                                             // exclusive map of my tile
            ex.d0 = __shfl_up(m.d0, 1); ex.d1 = __shfl_up(m.d1, 1);
            if (lane == 0 || lane == pos) ex = fr_pm_id();
            if (lane >= pos && lane < pos + run) Q.tiles[t0 + lane].carry = fr_seq_apply_map(carry, e0, ex.d0, ex.d1);
            const int last = pos + run - 1;
            carry = fr_seq_apply_map(carry, e0, fr_bcast_i64(m.d0, last), fr_bcast_i64(m.d1, last));
            pos += run;
        }
    }
    *di_io = di;
    return carry;
}

// Inclusive maps of the runs of clean same-binade tiles inside every batch of 64 staged tile records, in place (a run = what the walking
// wave composes in one step: it ends at a tile that may straddle a power of two, at a change of binade and at the batch's end).  The four
// waves take the batches in turn; segmented scan over lanes -- integer sums when no map of the batch holds an exact tie (the usual case).
__device__ __forceinline__ void fr_seq_runs(SeqTileL *tl, unsigned n_here_total) {
    const int lane = fr_lane(), w = threadIdx.x >> 6;
    for (unsigned t0 = w * 64u; t0 < n_here_total; t0 += FR_BLOCK) {
        const bool in = t0 + lane < n_here_total;
        SeqTileL r; r.dirty = 1; r.e = 0; r.d0 = r.d1 = 0;
        if (in) r = tl[t0 + lane];
        const int e_prev = __shfl_up(r.e, 1); const unsigned d_prev = (unsigned)__shfl_up((int)r.dirty, 1);
        const bool head = lane == 0 || r.dirty || d_prev || r.e != e_prev;
        const unsigned long long H = __ballot(head);
        const bool clean = in && !r.dirty;
        PMap m; m.d0 = clean ? r.d0 : 0; m.d1 = clean ? r.d1 : 0;
        if (!__any(m.d0 != m.d1)) {
            const long long inc = fr_wave_incl_i64(m.d0);
            // start of my run: the highest head at or below my lane
            const unsigned long long below = H & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
            const int hpos = 63 - __builtin_clzll(below);                // lane 0 is always a head
            const long long before = __shfl(inc, hpos > 0 ? hpos - 1 : 0);
            m.d0 = m.d1 = inc - (hpos > 0 ? before : 0ll);
        }
        else {
            int flag = head ? 1 : 0;
            for (int off = 1; off < 64; off <<= 1) {
                PMap o; o.d0 = __shfl_up(m.d0, off); o.d1 = __shfl_up(m.d1, off);
                const int of = __shfl_up(flag, off);
                if (lane >= off && !flag) { m = fr_pm_compose(o, m); flag |= of; }
            }
        }
        if (clean) { tl[t0 + lane].d0 = m.d0; tl[t0 + lane].d1 = m.d1; }
    }
}

// one workgroup: everybody stages (dirty tiles once, tile records in rounds of FR_SEQ_LCHUNK), wave 0 walks
template <class Acc>
__global__ void __launch_bounds__(FR_BLOCK) k_seq_chain(SeqWork Q, Acc acc, SeqStart st) {
    __shared__ SeqStage sh;
    __shared__ SeqTileL tl[FR_SEQ_LCHUNK];
    const unsigned long long tq0 = Q.dbg ? wall_clock64() : 0ull;
    fr_seq_stage(Q, acc, &sh);
    const unsigned long long tq1 = Q.dbg ? wall_clock64() : 0ull;
    unsigned long long t_load = 0, t_walk = 0, t_dirty = 0;
    const unsigned n = acc.count();
    const unsigned ntile = (n + FR_SEQ_TILE - 1) / FR_SEQ_TILE;
    double carry = st.value();
    unsigned di = 0;
    for (unsigned c0 = 0; c0 < ntile; c0 += FR_SEQ_LCHUNK) {
        const unsigned c1 = c0 + FR_SEQ_LCHUNK < ntile ? c0 + FR_SEQ_LCHUNK : ntile;
        if (c0) __syncthreads();        // wave 0 is done with the previous round's records
        const unsigned long long ta = Q.dbg ? wall_clock64() : 0ull;
        for (unsigned t = c0 + threadIdx.x; t < c1; t += FR_BLOCK) {
            const SeqRec q = Q.tiles[t];
            SeqTileL l; l.d0 = q.d0; l.d1 = q.d1; l.e = q.e; l.dirty = q.dirty;
            tl[t - c0] = l;
        }
        __syncthreads();
        fr_seq_runs(tl, c1 - c0);
        __syncthreads();
        const unsigned long long tb = Q.dbg ? wall_clock64() : 0ull;
        if (threadIdx.x < 64) carry = fr_seq_chain_wave(Q, acc, carry, &sh, tl, c0, c1, &di, &t_dirty);
        if (Q.dbg) { t_load += tb - ta; t_walk += wall_clock64() - tb; }
    }
#ifdef FR_SEQ_TIMING
    if (Q.dbg && threadIdx.x == 0) printf("[seq_chain] n %u tiles %u dirty tiles %u dirty subs %u: stage %llu load %llu walk %llu of which dirty tiles %llu (x10 ns)\n", n, ntile, sh.n_dt, sh.n_ds, tq1 - tq0, t_load, t_walk, t_dirty);
#endif
    if (threadIdx.x == 0) *Q.total = carry;
}

// ---- S5: exact running sums for the 4 consecutive elements of this thread.
// All 256 threads of the workgroup owning tile `tile` must call it.  sh: workgroup scratch.
struct SeqShared { PMap m[FR_BLOCK]; double a[FR_SEQ_TILE]; };

template <class Acc>
__device__ __forceinline__ void fr_seq_prefix4(const SeqWork &Q, const Acc &acc, unsigned tile, SeqShared *sh, double S[4], double *S_before) {
    const unsigned n = acc.count();
    const SeqRec tr = Q.tiles[tile];
    size_t base = (size_t)tile * FR_SEQ_TILE + (size_t)threadIdx.x * 4;
    double a[4];
    for (int it = 0; it < 4; it++) { size_t i = base + it; a[it] = i < n ? acc.get(i) : 0.0; }
    if (!tr.dirty) {
        double scale = ldexp(1.0, 52 - tr.e), ulp = ldexp(1.0, tr.e - 52);
        PMap loc[4];
        loc[0] = fr_pm_elem(a[0], scale);
        for (int it = 1; it < 4; it++) loc[it] = fr_pm_compose(loc[it - 1], fr_pm_elem(a[it], scale));
        // exclusive scan of the per-thread maps across the workgroup (ordered)
        const int lane = fr_lane(), w = threadIdx.x >> 6;
        PMap m = loc[3];
        for (int off = 1; off < 64; off <<= 1) {
            PMap o; o.d0 = __shfl_up(m.d0, off); o.d1 = __shfl_up(m.d1, off);
            if (lane >= off) m = fr_pm_compose(o, m);
        }
        if (lane == 63) sh->m[w] = m;
        __syncthreads();
        PMap wbase = fr_pm_id();
        for (int k = 0; k < w; k++) wbase = fr_pm_compose(wbase, sh->m[k]);
        PMap ex; ex.d0 = __shfl_up(m.d0, 1); ex.d1 = __shfl_up(m.d1, 1);
        if (lane == 0) ex = fr_pm_id();
        ex = fr_pm_compose(wbase, ex);
        __syncthreads();
        // (every partial sum of a clean tile lies in binade tr.e: integer <-> double by the mantissa bits, see fr_seq_apply_map)
        const long long cb = __double_as_longlong(tr.carry);
        const bool inb = ((cb >> 52) & 0x7ffll) == (long long)(tr.e + 1023);
        long long M = inb ? ((cb & 0xFFFFFFFFFFFFFll) | (1ll << 52)) : (long long)(tr.carry * scale);
        int p = (int)(M & 1);
        const long long ebits = (long long)(tr.e + 1023) << 52;
        auto to_double = [&](long long R) { return (inb && R >= (1ll << 52) && R < (1ll << 53)) ? __longlong_as_double(ebits | (R & 0xFFFFFFFFFFFFFll)) : (double)R * ulp; };
        long long dex = p ? ex.d1 : ex.d0;
        *S_before = to_double(M + dex);
        for (int it = 0; it < 4; it++) {
            PMap c = fr_pm_compose(ex, loc[it]);
            long long d = p ? c.d1 : c.d0;
            S[it] = to_double(M + d);
        }
        return;
    }
    // dirty tile: per sub-tile
    const int sub = threadIdx.x >> 4, sl = threadIdx.x & 15;
    const SeqRec sr = Q.subs[(size_t)tile * FR_SUBS_PER_TILE + sub];
    for (int it = 0; it < 4; it++) sh->a[threadIdx.x * 4 + it] = a[it];
    __syncthreads();
    size_t sub_lo = (size_t)tile * FR_SEQ_TILE + (size_t)sub * 64;
    if (sub_lo >= n) { for (int it = 0; it < 4; it++) S[it] = sr.carry; *S_before = sr.carry; __syncthreads(); return; }
    if (!sr.dirty) {
        double scale = ldexp(1.0, 52 - sr.e), ulp = ldexp(1.0, sr.e - 52);
        // sequential composition over the preceding elements of the sub-tile (<= 60 of them)
        PMap ex = fr_pm_id();
        const double *sa = &sh->a[sub * 64];
        for (int k = 0; k < sl * 4; k++) ex = fr_pm_compose(ex, fr_pm_elem(sa[k], scale));
        long long M = (long long)(sr.carry * scale);
        int p = (int)(M & 1);
        *S_before = (double)(M + (p ? ex.d1 : ex.d0)) * ulp;
        PMap c = ex;
        for (int it = 0; it < 4; it++) {
            c = fr_pm_compose(c, fr_pm_elem(a[it], scale));
            S[it] = (double)(M + (p ? c.d1 : c.d0)) * ulp;
        }
    }
    else {
        double s = sr.carry;
        const double *sa = &sh->a[sub * 64];
        for (int k = 0; k < sl * 4; k++) s = s + sa[k];
        *S_before = s;
        for (int it = 0; it < 4; it++) { s = s + a[it]; S[it] = s; }
    }
    __syncthreads();
}
