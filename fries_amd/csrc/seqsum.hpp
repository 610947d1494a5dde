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
    const uint32_t *skip = nullptr;     // launched ahead of the host's look at a flag: *skip != 0 = the input is not final, leave at once (run_stage)
    uint32_t *tk_word = nullptr; uint32_t tk = 0;      // k_seq_sums only: raise this ticket when the kernel starts (everything enqueued before it has finished; fr_ticket_reserve)
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
    if (Q.tk_word && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(Q.tk_word, Q.tk, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (Q.skip && *Q.skip) return;
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
    if (Q.skip && *Q.skip) return;
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

// ---- S4: the chain, one workgroup.  Only what is inherently sequential is walked by one wave: the SEGMENTS --
//   * a run of clean tiles in one binade (inside a batch of 64 tile records): one parity-map application,
//   * inside a tile that may straddle a power of two: a run of clean sub-tiles in one binade (one application), or a sub-tile that may
//     straddle one: its 64 elements added one by one --
// about 100 of them per million elements.  Everything else is done by all four waves around the walk: the records are staged in LDS,
// the runs' maps composed, the segments listed in order (integer scan), and afterwards every tile and sub-tile forms its own entry
// carry from its segment's entry carry and the map of the part of the run before it.  (Until round 3 the walking wave also found the
// runs by ballots, composed the sub-tile runs by shuffles and stored every tile's carry itself: 46 us per sum at a million elements,
// all of it dependent scalar-like work on one wave.)
// Tiles are taken in chunks of FR_SEQ_LCHUNK records; a chunk ends early after FR_SEQ_DT_CAP tiles that may straddle.
#define FR_SEQ_DT_CAP 64            // dirty tiles per chunk
#define FR_SEQ_DS_CAP 64            // dirty sub-tiles whose elements are staged per chunk (the others are read from global memory by the walk)
#define FR_SEQ_LCHUNK 1024          // tile records per chunk
#define FR_SEQ_SEG_CAP (FR_SEQ_LCHUNK + FR_SEQ_DT_CAP * FR_SUBS_PER_TILE)
#define FR_SEG_TILE_RUN 0u          // low 30 bits: chunk-relative index of the run's LAST tile (its record holds the run's map)
#define FR_SEG_SUB_RUN 1u           // index into subs[] of the run's last sub-tile
#define FR_SEG_SUB_SEQ 2u           // index into subs[] of the sub-tile to add element by element
struct SeqSubL { long long d0, d1; int e; uint32_t dirty; int slot; int pad; };
struct SeqTileL { long long d0, d1; int e; uint32_t dirty; };
struct SeqChainSh {
    SeqTileL tl[FR_SEQ_LCHUNK];
    SeqSubL subs[FR_SEQ_DT_CAP * FR_SUBS_PER_TILE];
    double elems[FR_SEQ_DS_CAP * 64];
    double seg_in[FR_SEQ_SEG_CAP + 1];          // exact running sum entering each segment; [nseg] = leaving the chunk
    uint32_t segs[FR_SEQ_SEG_CAP];
    uint32_t segoff[FR_SEQ_LCHUNK];             // segments started before this tile
    uint32_t dt_list[FR_SEQ_DT_CAP];            // chunk-relative tile indices, ascending
    uint32_t dt_nseg[FR_SEQ_DT_CAP];            // segments each of them starts
    uint32_t slot_sub[FR_SEQ_DS_CAP];           // which sub-tile's elements a slot holds (index into subs[])
    uint32_t scan[8];
    uint32_t n_ds, cut, nseg;
};

// Inclusive maps of the runs of clean same-binade tiles inside every batch of 64 staged tile records, in place (a run ends at a tile
// that may straddle a power of two, at a change of binade and at the batch's end).  The four waves take the batches in turn; segmented
// scan over lanes -- integer sums when no map of the batch holds an exact tie (the usual case).
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
// does clean tile t (chunk-relative) start / end a run?  (the same rule fr_seq_runs applies)
__device__ __forceinline__ bool fr_seq_tile_head(const SeqTileL *tl, unsigned t) { return (t & 63u) == 0u || tl[t - 1].dirty || tl[t - 1].e != tl[t].e; }
__device__ __forceinline__ bool fr_seq_tile_last(const SeqTileL *tl, unsigned t, unsigned L) { return t + 1 == L || ((t + 1) & 63u) == 0u || tl[t + 1].dirty || tl[t + 1].e != tl[t].e; }

// fr_seq_apply_map on wave-uniform operands: everything stays in scalar registers (the walk is one dependent chain; on the vector
// unit every step of it waits out the VALU's issue latency)
__device__ __forceinline__ unsigned long long fr_seq_apply_map_u(unsigned long long cb, int e, long long d0, long long d1) {
    if ((long long)((cb >> 52) & 0x7ffull) == (long long)(e + 1023)) {
        const long long Mi = (long long)((cb & 0xFFFFFFFFFFFFFull) | (1ull << 52));
        const long long R = Mi + ((Mi & 1) ? d1 : d0);
        if (R >= (1ll << 52) && R < (1ll << 53)) return ((unsigned long long)(e + 1023) << 52) | ((unsigned long long)R & 0xFFFFFFFFFFFFFull);
    }
    return (unsigned long long)__double_as_longlong(fr_seq_apply_map(__longlong_as_double((long long)cb), e, d0, d1));
}
__device__ __forceinline__ unsigned long long fr_uniform_u64(unsigned long long v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
// sub-tiles of a tile that hold elements
__device__ __forceinline__ int fr_seq_n_sub(unsigned n, size_t t_lo) {
    const size_t left = ((size_t)n - t_lo + 63) / 64;
    return (int)(left < (size_t)FR_SUBS_PER_TILE ? left : (size_t)FR_SUBS_PER_TILE);
}

template <class Acc>
__global__ void __launch_bounds__(FR_BLOCK) k_seq_chain(SeqWork Q, Acc acc, SeqStart st) {
    __shared__ SeqChainSh sh;
    if (Q.skip && *Q.skip) return;
    const unsigned n = acc.count();
    const unsigned ntile = (n + FR_SEQ_TILE - 1) / FR_SEQ_TILE;
    const int lane = fr_lane();
    const int grp = threadIdx.x >> 4, j16 = threadIdx.x & 15;     // 16-lane groups: one tile that may straddle each, lane = sub-tile
    constexpr unsigned PER = FR_SEQ_LCHUNK / FR_BLOCK;          // tile records per thread
    unsigned long long carry = (unsigned long long)__double_as_longlong(st.value());          // the exact running sum, as bits (wave 0's copy is walked on)
#ifdef FR_SEQ_TIMING
    unsigned long long tq[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tl0 = wall_clock64();
#define FR_SEQ_T(i) do { if (Q.dbg) { const unsigned long long now_ = wall_clock64(); tq[i] += now_ - tl0; tl0 = now_; } } while (0)
#else
#define FR_SEQ_T(i) do { } while (0)
#endif
    for (unsigned c0 = 0; c0 < ntile; ) {
        unsigned L = ntile - c0 < FR_SEQ_LCHUNK ? ntile - c0 : FR_SEQ_LCHUNK;
        if (c0) __syncthreads();
        // A: tile records of the chunk
        for (unsigned t = threadIdx.x; t < L; t += FR_BLOCK) {
            const SeqRec q = Q.tiles[c0 + t];
            SeqTileL l; l.d0 = q.d0; l.d1 = q.d1; l.e = q.e; l.dirty = q.dirty;
            sh.tl[t] = l;
        }
        if (threadIdx.x == 0) { sh.n_ds = 0; sh.cut = L; }
        __syncthreads();
        FR_SEQ_T(0);
        // B: the tiles that may straddle a power of two, in order (a listed tile's `dirty` becomes 1 + its place in the list); the chunk
        // ends in front of the (FR_SEQ_DT_CAP + 1)-th
        const unsigned lo = threadIdx.x * PER;
        uint32_t cnt = 0;
#pragma unroll
        for (unsigned k = 0; k < PER; k++) if (lo + k < L && sh.tl[lo + k].dirty) cnt++;
        uint32_t tot_dt;
        uint32_t o = fr_block_scan_u32(cnt, sh.scan, &tot_dt) - cnt;
#pragma unroll
        for (unsigned k = 0; k < PER; k++) if (lo + k < L && sh.tl[lo + k].dirty) { if (o < FR_SEQ_DT_CAP) { sh.dt_list[o] = lo + k; sh.tl[lo + k].dirty = 1u + o; } else if (o == FR_SEQ_DT_CAP) sh.cut = lo + k; o++; }
        __syncthreads();
        L = sh.cut;
        const unsigned n_dt = tot_dt < FR_SEQ_DT_CAP ? tot_dt : FR_SEQ_DT_CAP;
        FR_SEQ_T(1);
        // C: their sub-tile records; a slot for the elements of every sub-tile that has to be added one by one
        for (unsigned i = threadIdx.x; i < n_dt * FR_SUBS_PER_TILE; i += FR_BLOCK) {
            const unsigned t = c0 + sh.dt_list[i / FR_SUBS_PER_TILE], j = i % FR_SUBS_PER_TILE;
            const SeqRec r = Q.subs[(size_t)t * FR_SUBS_PER_TILE + j];
            SeqSubL l; l.d0 = r.d0; l.d1 = r.d1; l.e = r.e; l.dirty = r.dirty; l.slot = -1; l.pad = 0;
            if (r.dirty && (size_t)t * FR_SEQ_TILE + (size_t)j * 64 < n) {
                const uint32_t slot = atomicAdd(&sh.n_ds, 1u);
                if (slot < FR_SEQ_DS_CAP) { l.slot = (int)slot; sh.slot_sub[slot] = i; }
            }
            sh.subs[i] = l;
        }
        __syncthreads();
        FR_SEQ_T(2);
        // D: the staged elements; the runs' maps of the clean tiles (in place, all waves)
        {
            const unsigned n_st = sh.n_ds < FR_SEQ_DS_CAP ? sh.n_ds : FR_SEQ_DS_CAP;
            for (unsigned i = threadIdx.x; i < n_st * 64; i += FR_BLOCK) {
                const unsigned slot = i >> 6, k = i & 63, si = sh.slot_sub[slot];
                const size_t e = (size_t)(c0 + sh.dt_list[si / FR_SUBS_PER_TILE]) * FR_SEQ_TILE + (size_t)(si % FR_SUBS_PER_TILE) * 64 + k;
                sh.elems[slot * 64 + k] = e < n ? acc.get(e) : 0.0;
            }
        }
        fr_seq_runs(sh.tl, L);
        FR_SEQ_T(3);
        // E: inside the tiles that may straddle (16 lanes each): the runs' maps of the clean sub-tiles, which segment of its tile every
        // sub-tile belongs to (pad), how many segments the tile starts (pad of its sub-tile 0 ... kept in dt_nseg)
        for (unsigned di = (unsigned)grp; di < ((n_dt + 15u) & ~15u); di += FR_BLOCK / 16) {
            const bool have = di < n_dt;
            const int n_sub_here = have ? fr_seq_n_sub(n, (size_t)(c0 + sh.dt_list[di]) * FR_SEQ_TILE) : 0;
            SeqSubL q; q.d0 = q.d1 = 0; q.e = 0; q.dirty = 1; q.slot = -1; q.pad = 0;
            const bool valid = j16 < n_sub_here;
            if (valid) q = sh.subs[di * FR_SUBS_PER_TILE + j16];
            const int e_prev = __shfl_up(q.e, 1, 16); const unsigned d_prev = (unsigned)__shfl_up((int)q.dirty, 1, 16);
            const bool head = valid && (j16 == 0 || q.dirty || d_prev || q.e != e_prev);
            PMap m; m.d0 = (valid && !q.dirty) ? q.d0 : 0; m.d1 = (valid && !q.dirty) ? q.d1 : 0;
            int flag = head ? 1 : 0;
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                PMap o2; o2.d0 = __shfl_up(m.d0, off, 16); o2.d1 = __shfl_up(m.d1, off, 16);
                const int of = __shfl_up(flag, off, 16);
                if (j16 >= off && !flag) { m = fr_pm_compose(o2, m); flag |= of; }
            }
            const unsigned hb = (unsigned)((__ballot(head) >> ((lane >> 4) << 4)) & 0xFFFFull);      // heads of my 16 lanes
            if (valid) {
                SeqSubL *w = &sh.subs[di * FR_SUBS_PER_TILE + j16];
                if (!q.dirty) { w->d0 = m.d0; w->d1 = m.d1; }
                w->pad = __popc(hb & ((2u << j16) - 1u)) - 1;
            }
            if (have && j16 == 0) sh.dt_nseg[di] = (uint32_t)__popc(hb);
        }
        __syncthreads();
        FR_SEQ_T(4);
        // F: segments started per tile -> offsets, and the segment list.  A clean tile starts one where its run starts; a tile that may
        // straddle starts one per run of clean sub-tiles and per sub-tile added one by one.
        uint32_t sc = 0;
#pragma unroll
        for (unsigned k = 0; k < PER; k++) {
            const unsigned t = lo + k;
            if (t >= L) continue;
            const uint32_t dd = sh.tl[t].dirty;
            sc += dd ? sh.dt_nseg[dd - 1u] : (fr_seq_tile_head(sh.tl, t) ? 1u : 0u);
        }
        uint32_t nseg;
        uint32_t so = fr_block_scan_u32(sc, sh.scan, &nseg) - sc;
#pragma unroll
        for (unsigned k = 0; k < PER; k++) {
            const unsigned t = lo + k;
            if (t >= L) continue;
            sh.segoff[t] = so;
            const uint32_t dd = sh.tl[t].dirty;
            if (!dd) {
                const bool head = fr_seq_tile_head(sh.tl, t);
                if (fr_seq_tile_last(sh.tl, t, L)) sh.segs[head ? so : so - 1] = (FR_SEG_TILE_RUN << 30) | t;
                so += head ? 1u : 0u;
            }
            else so += sh.dt_nseg[dd - 1u];
        }
        __syncthreads();
        for (unsigned di = (unsigned)grp; di < n_dt; di += FR_BLOCK / 16) {
            const unsigned t = sh.dt_list[di];
            const int n_sub_here = fr_seq_n_sub(n, (size_t)(c0 + t) * FR_SEQ_TILE);
            if (j16 < n_sub_here) {
                const SeqSubL *q = &sh.subs[di * FR_SUBS_PER_TILE];
                const bool last = q[j16].dirty || j16 + 1 == n_sub_here || q[j16 + 1].pad != q[j16].pad;
                if (last) sh.segs[sh.segoff[t] + (uint32_t)q[j16].pad] = ((q[j16].dirty ? FR_SEG_SUB_SEQ : FR_SEG_SUB_RUN) << 30) | (di * FR_SUBS_PER_TILE + (unsigned)j16);
            }
        }
        __syncthreads();
        FR_SEQ_T(5);
        // G: the walk (wave 0): 64 segments per batch, lane k fetches segment k's record, the wave then applies them one after the other --
        // on the scalar unit, except for the 64 floating-point additions of a sub-tile that is added one by one
        if (threadIdx.x < 64) {
            carry = fr_uniform_u64(carry);
            for (unsigned s0 = 0; s0 < nseg; s0 += 64) {
                const unsigned n_here = nseg - s0 < 64u ? nseg - s0 : 64u;
                uint32_t kind = 3u, idx = 0; long long d0 = 0, d1 = 0; int e = 0, slot = -1;
                if ((unsigned)lane < n_here) {
                    const uint32_t sg = sh.segs[s0 + lane];
                    kind = sg >> 30; idx = sg & 0x3FFFFFFFu;
                    if (kind == FR_SEG_TILE_RUN) { const SeqTileL r = sh.tl[idx]; d0 = r.d0; d1 = r.d1; e = r.e; }
                    else { const SeqSubL r = sh.subs[idx]; d0 = r.d0; d1 = r.d1; e = r.e; slot = r.slot; }
                }
                unsigned long long my_in = 0;
                for (unsigned k = 0; k < n_here; k++) {
                    if ((unsigned)lane == k) my_in = carry;
                    const uint32_t kk = (uint32_t)__builtin_amdgcn_readlane((int)kind, (int)k);
                    if (kk == FR_SEG_SUB_SEQ) {
                        const int sl = __builtin_amdgcn_readlane(slot, (int)k);
                        double mine_a;
                        if (sl >= 0) mine_a = sh.elems[sl * 64 + lane];
                        else {
                            const uint32_t ix = (uint32_t)__builtin_amdgcn_readlane((int)idx, (int)k);
                            const size_t e_lo = (size_t)(c0 + sh.dt_list[ix / FR_SUBS_PER_TILE]) * FR_SEQ_TILE + (size_t)(ix % FR_SUBS_PER_TILE) * 64;
                            mine_a = e_lo + lane < n ? acc.get(e_lo + lane) : 0.0;
                        }
                        double cv = __longlong_as_double((long long)carry);
#pragma unroll
                        for (int q = 0; q < 64; q++) cv = cv + fr_bcast_f64(mine_a, q);
                        carry = fr_uniform_u64((unsigned long long)__double_as_longlong(cv));
                    }
                    else {
                        const long long a0 = fr_bcast_i64(d0, (int)k), a1 = fr_bcast_i64(d1, (int)k);
                        if (a0 | a1) carry = fr_uniform_u64(fr_seq_apply_map_u(carry, __builtin_amdgcn_readlane(e, (int)k), a0, a1));
                    }
                }
                if ((unsigned)lane < n_here) sh.seg_in[s0 + lane] = __longlong_as_double((long long)my_in);
            }
            if (lane == 0) sh.seg_in[nseg] = __longlong_as_double((long long)carry);
        }
        __syncthreads();
        FR_SEQ_T(6);
        // H: every tile's and sub-tile's entry carry from its segment's (all threads; the tiles that may straddle by 16 lanes each)
#pragma unroll
        for (unsigned k = 0; k < PER; k++) {
            const unsigned t = lo + k;
            if (t >= L) continue;
            const uint32_t so_t = sh.segoff[t];
            if (!sh.tl[t].dirty) {
                const bool head = fr_seq_tile_head(sh.tl, t);
                const double cin = sh.seg_in[head ? so_t : so_t - 1];
                Q.tiles[c0 + t].carry = head ? cin : fr_seq_apply_map(cin, sh.tl[t].e, sh.tl[t - 1].d0, sh.tl[t - 1].d1);
            }
            else Q.tiles[c0 + t].carry = sh.seg_in[so_t];
        }
        for (unsigned di = (unsigned)grp; di < n_dt; di += FR_BLOCK / 16) {
            const unsigned t = sh.dt_list[di];
            const int n_sub_here = fr_seq_n_sub(n, (size_t)(c0 + t) * FR_SEQ_TILE);
            const SeqSubL *q = &sh.subs[di * FR_SUBS_PER_TILE];
            const uint32_t so_t = sh.segoff[t];
            double cin;
            if (j16 < n_sub_here) {
                const SeqSubL me = q[j16];
                cin = sh.seg_in[so_t + (uint32_t)me.pad];
                if (!me.dirty && j16 > 0 && q[j16 - 1].pad == me.pad) cin = fr_seq_apply_map(cin, me.e, q[j16 - 1].d0, q[j16 - 1].d1);
            }
            else cin = sh.seg_in[so_t + sh.dt_nseg[di]];        // the sum leaving the tile
            Q.subs[(size_t)(c0 + t) * FR_SUBS_PER_TILE + j16].carry = cin;
        }
        FR_SEQ_T(7);
#ifdef FR_SEQ_TIMING
        if (Q.dbg && threadIdx.x == 0) printf("[seq_chain] n %u tiles %u chunk %u+%u dirty tiles %u dirty subs %u segments %u | A %llu B %llu C %llu D %llu E %llu F %llu G %llu H %llu (x10 ns, cumulative over chunks)\n", n, ntile, c0, L, n_dt, sh.n_ds, nseg, tq[0], tq[1], tq[2], tq[3], tq[4], tq[5], tq[6], tq[7]);
#endif
        c0 += L;
    }
    if (threadIdx.x == 0) *Q.total = __longlong_as_double((long long)carry);
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
