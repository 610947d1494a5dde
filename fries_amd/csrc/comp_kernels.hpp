// Device formulation of comp_sub = find_keep_sub + sys_sub
// (FRIES/compress_utils.cpp:130-276, 702-794, 797-820) for the five HB-PP stages.
//
// find_keep_sub: the reference sweeps the elements sequentially in blocks of 8; each block
// tests its elements against the running remaining norm with the sample budget as it stood at
// the block's start (compress_utils.cpp:159-178), and sweeps repeat until one keeps nothing.
// Because that budget is stale inside a block, the kept set is NOT the plain fixed point
// {x : x * (n - #kept) >= norm}: it depends on the sweep.  The sweep is a chain through two
// scalars (norm removed so far, samples used so far).  k_fks_iter replays every sweep for
// every 8-block in parallel from *guessed* chain values, publishes each block's deltas, and is
// re-run with the prefix sums of those deltas as the new guesses until nothing changes; a
// consistent assignment is unique (block 0 is always right, then block 1, ...), so the fixed
// point is the reference's result, including the budget each wt_remain was divided by.
//
// sys_sub: an in-order prefix sum of wt_remain gives every element its lbound; the comb
// positions are exact (teeth.hpp); an element starts from tooth T(lbound of its predecessor)
// and replays the reference's inner loop.  Output slots come from an integer prefix sum of the
// per-element emission counts, so the output order is the reference's.
#pragma once
#include "fries_dev.hpp"
#include "teeth.hpp"
#include "hbpp_rows.hpp"
#include "seqsum.hpp"

#define FR_ITEMS 4
#define FR_TILE (FR_BLOCK * FR_ITEMS)
#define FR_MAX_ROUNDS 600
#define FR_FKS_PMAX 64          // sweeps tracked per replay (the reference needs 2-6 in steady state, more when the budget exceeds the elements)
#define FR_FKS_TILE (FR_BLOCK * 8)  // elements per workgroup in k_fks_iter
#define FR_MAX_FIX 8192         // comb repairs a stage may ask for through the short in-order walk (a handful in practice; frisys_hh's
                                // thousands go through k_sys_prop); beyond it FR_ERR_BACKLOG

// ascending sort of list[0, n), n <= FR_MAX_FIX, by the whole workgroup (bitonic over the next power of two; the tail is padded)
__device__ __forceinline__ void fr_sort_fix_list(uint32_t *list, uint32_t n) {
    uint32_t m = 1;
    while (m < n) m <<= 1;
    for (uint32_t i = n + threadIdx.x; i < m; i += blockDim.x) list[i] = 0xFFFFFFFFu;
    __syncthreads();
    for (uint32_t k = 2; k <= m; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    const uint32_t a = list[i], b = list[l];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { list[i] = b; list[l] = a; }
                }
            }
            __syncthreads();
        }
}

struct CompState {
    double G;            // remaining norm entering this round
    double loc_norm;     // final: what find_keep_sub returns
    double unit;         // final: loc_norm / n_rem
    uint32_t n_rem;      // samples still to distribute
    uint32_t n_in;       // number of elements of this stage
    uint32_t done;
    uint32_t pbuf;       // partial buffer holding per-block sums of wt_remain
    uint32_t n_out;      // emissions written by sys_write
    uint32_t n_fix;      // elements whose tooth count had to be repaired
    uint32_t changed;    // k_fks_iter: some block's deltas differ from the previous replay
    uint32_t n_pass;     // sweeps the reference would have run
};

struct StageElems {      // one ping-pong half
    double *val;
    uint32_t *pos;       // parent position in the solution vector
    uint32_t *code;      // 4 orbital-code bytes (hbpp_rows.hpp)
    uint32_t *ndiv;      // >0: uniform subdivision into ndiv equal parts
    uint32_t *nsub;      // row length (jagged stages)
    double *rinv;        // cached 1 / norm of the element's row (non-uniform elements)
    uint32_t *raux;      // cached RowInfo::aux
    det_t *det;          // the parent's determinant, copied at prep time: every replay reads it coalesced instead of gathering V.dets[pos]
};

struct CompWork {
    uint32_t cap;                 // element capacity of every array below
    StageElems el[2];             // current / previous stage
    double *wt_remain;
    uint32_t *keep;               // bit s = sub-element s preserved exactly
    double *S;                    // inclusive lbound after each element
    uint32_t *kin, *cnt;
    uint32_t *kend;               // tooth index after each element (propagation repair, see k_sys_prop)
    uint32_t *act[2], *act_n;     // active lists of the propagation repair, and their lengths [2]
    uint32_t *kstart;             // tag of the repair round in which the element heads a chain (k_sys_walk)
    uint32_t prop;                // 1: repair by parallel propagation (many repairs expected), 0: short sequential fix-up
    uint32_t *e_wi, *e_sub;       // emissions: source element, sub index
    double *e_val;
    double *psum[2];              // per-block partial sums (FR_MAX_PART each)
    uint32_t *pcnt[2];
    CompState *state;             // [FR_MAX_ROUNDS + 2]; last slot = final
    Teeth *teeth;
    uint32_t *fix_list;           // elements with a tooth backlog (rare)
    SeqWork seq;                  // exact in-order sum of wt_remain
    double row1[2];               // stage-1 sub-weights (fr_row1)
    // Emissions staged by k_sys_count so that k_sys_write does not replay sys_sub a second time: FR_STG_SLOTS records per lane (its four
    // elements' emissions in order), what does not fit in a per-tile spill list; a tile the comb repair touched, or whose spill list
    // overflowed, is flagged and written the old way (rows evaluated again).  nullptr: no staging (propagation repair: frisys_hh).
    uint4 *stg, *spill; uint32_t *spill_cnt; uint8_t *tile_dirty;
#ifdef FR_SYS_TIMING
    unsigned long long *tdbg;     // FRIES_SYS_DBG: [workgroup][8] time stamps of k_sys_count / k_sys_write (build with -DFR_SYS_TIMING)
#endif
};
#ifdef FR_SYS_TIMING
#define FR_SYS_T(i) do { if (W.tdbg && threadIdx.x == 0 && blockIdx.x < 8192) W.tdbg[(size_t)blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define FR_SYS_T(i) do { } while (0)
#endif

// wt_remain of the current stage as the addend sequence of sys_sub's lbound (compress_utils.cpp:739)
struct AccWt {
    const double *wt; const CompState *st0;
    __device__ unsigned count() const { return st0->n_in; }
    __device__ double get(size_t i) const { return wt[i]; }
};


// In-block prefix of per-thread sums with one fixed association, shared by the round kernels
// (which publish the block total) and k_sys_count (which continues the same sums), so that
// "lbound after the last element of block b" is bit-identical in both.
// Returns the exclusive prefix of this thread; *block_total = prefix after thread 255.
__device__ __forceinline__ double fr_block_excl_f64(double tsum, double *sh /* >= 12 */, double *block_total) {
    int lane = fr_lane(), w = threadIdx.x >> 6;
    double winc = tsum;
    for (int off = 1; off < 64; off <<= 1) { double t = __shfl_up(winc, off); if (lane >= off) winc += t; }
    if (lane == 63) sh[w] = winc;
    __syncthreads();
    double base = 0;
    for (int k = 0; k < w; k++) base += sh[k];
    double incl = w ? base + winc : winc;
    if (lane == 63) sh[4 + w] = incl;
    double texcl = __shfl_up(incl, 1);
    __syncthreads();
    if (lane == 0) texcl = w ? sh[4 + w - 1] : 0.0;
    if (threadIdx.x == FR_BLOCK - 1) sh[8] = texcl + tsum;
    __syncthreads();
    *block_total = sh[8];
    __syncthreads();
    return texcl;
}

// exclusive scans across the 64 lanes of a wave (storage order), no LDS, no barrier
__device__ __forceinline__ uint32_t fr_wave_excl_u32(uint32_t x, uint32_t *total) {
    int lane = fr_lane();
    uint32_t v = x;
    for (int off = 1; off < 64; off <<= 1) { uint32_t t = __shfl_up(v, off); if (lane >= off) v += t; }
    *total = __shfl(v, 63);
    return v - x;
}
__device__ __forceinline__ double fr_wave_excl_f64(double x, double *total) {
    int lane = fr_lane();
    double v = x;
    for (int off = 1; off < 64; off <<= 1) { double t = __shfl_up(v, off); if (lane >= off) v += t; }
    *total = __shfl(v, 63);
    double e = __shfl_up(v, 1);
    return lane ? e : 0.0;
}

// ------------------------------------------------------------------ stage dispatch
// Per-element row: setup -> RowInfo, visit -> (sub, normalised weight) ascending.
template <int STAGE, bool NEW_HB>
__device__ __forceinline__ RowInfo fr_row_setup(const HbTables &T, det_t det, uint32_t code, double p_doub) {
    if (STAGE == 1) { RowInfo r; r.inv_norm = 1; r.tot = 1; r.nsub = 2; r.aux = 0; return r; }
    if (STAGE == 2) return fr_row2_setup<NEW_HB>(T, det);
    if (STAGE == 3) return NEW_HB ? fr_row3h_setup(T, det, fr_c(code, 1)) : fr_row3_setup(T, det, fr_c(code, 1));
    if (STAGE == 4) {
        unsigned o1_idx = fr_c(code, 1), o2_idx = fr_c(code, 2);
        unsigned o1 = fr_nth_bit(det, o1_idx), o2 = fr_nth_bit(det, o2_idx);
        bool excl = NEW_HB && ((o1_idx / (T.n_elec / 2)) == (o2 / T.n_orb));
        RowInfo r = fr_row4_setup(T, det, o1, excl);
        r.aux = o1 | (excl ? 0x100u : 0u);
        return r;
    }
    // STAGE == 5
    unsigned o1 = fr_nth_bit(det, fr_c(code, 1)), o2 = fr_nth_bit(det, fr_c(code, 2)), u1 = fr_c(code, 3);
    RowInfo r = fr_row5_setup<NEW_HB>(T, det, o1, o2, u1);
    r.aux = o1 | (o2 << 8);
    return r;
}

template <int STAGE, bool NEW_HB, class F>
__device__ __forceinline__ void fr_row_visit(const HbTables &T, det_t det, uint32_t code, const RowInfo &ri, double p_doub, F f) {
    if (STAGE == 1) { f(0u, ri.inv_norm); f(1u, ri.tot); }        // the two first-level weights travel in the RowInfo (fr_row1)
    else if (STAGE == 2) fr_row2_visit<NEW_HB>(T, det, [&](unsigned s, double w) { f(s, w * ri.inv_norm); });
    else if (STAGE == 3) {
        if (NEW_HB) fr_row3h_visit(T, det, fr_c(code, 1), ri.aux, [&](unsigned s, double w) { f(s, w * ri.inv_norm); });
        else fr_row3_visit(T, det, fr_c(code, 1), ri.aux, [&](unsigned s, double w) { f(s, w * ri.inv_norm); });
    }
    else if (STAGE == 4) fr_row4_visit(T, det, ri.aux & 0xffu, (ri.aux & 0x100u) != 0, [&](unsigned s, double w) { f(s, w * ri.inv_norm); });
    else fr_row5_visit<NEW_HB>(T, det, ri.aux & 0xffu, (ri.aux >> 8) & 0xffu, fr_c(code, 3), [&](unsigned s, double w) { f(s, w * ri.inv_norm); });
}

// Stage 1 has one row for every element: {p_doub, 1 - p_doub} for molecules (heat_bathPP.cpp:714-727), {t, g} for
// Hubbard-Holstein (frisys_hh.cpp:191-194)
__device__ __forceinline__ RowInfo fr_row1(const double w[2]) { RowInfo r; r.inv_norm = w[0]; r.tot = w[1]; r.nsub = 2; r.aux = 0; return r; }

// RowInfo of a stored element: the prep kernel cached what the visit needs
__device__ __forceinline__ RowInfo fr_row_cached(const StageElems &E, size_t e) {
    RowInfo r; r.inv_norm = E.rinv[e]; r.aux = E.raux[e]; r.nsub = E.nsub[e]; r.tot = 0;
    return r;
}

// number of sub-weights comp_sub sees for this element (sub_sizes[] or the column count)
template <int STAGE, bool NEW_HB>
__device__ __forceinline__ unsigned fr_row_len(const HbTables &T, uint32_t nsub_stored) {
    if (STAGE == 1) return 2;
    if (STAGE == 2) return T.n_elec - (NEW_HB ? 1 : 0);
    if (STAGE == 3) return NEW_HB ? nsub_stored : T.n_elec;
    if (STAGE == 4) return T.n_orb - T.n_elec / 2;
    return nsub_stored;
}

// ------------------------------------------------------------------ replay of sys_sub for one element
// Everything one element contributes to sys_sub.  The kernels fetch these for their four elements in one batch (independent
// loads, one wait) before any element is evaluated: evaluated one after the other, each element costs three dependent
// global-load latencies (value -> parent position -> determinant) and the kernels were bound by exactly that.
struct ElemIn {
    double v, wr, rinv;
    uint32_t nd, kp, code, pos, nsub, raux;
    det_t det;
};
template <int STAGE, int N>
__device__ __forceinline__ void fr_load_elems(const CompWork &W, const VecDev &V, int cur, size_t base, unsigned n_in, ElemIn (&x)[N]) {
    const StageElems &E = W.el[cur];
    const size_t last = n_in ? n_in - 1 : 0;
#pragma unroll
    for (int it = 0; it < N; it++) {
        const size_t e = base + it;
        const size_t ec = e < n_in ? e : last;          // clamped: the loads stay unconditional
        x[it].v = E.val[ec]; x[it].nd = E.ndiv[ec]; x[it].wr = W.wt_remain[ec]; x[it].kp = W.keep[ec];
        if (STAGE != 1) { x[it].code = E.code[ec]; x[it].pos = E.pos[ec]; x[it].nsub = E.nsub[ec]; x[it].raux = E.raux[ec]; x[it].rinv = E.rinv[ec]; }
        else { x[it].code = 0; x[it].pos = 0; x[it].nsub = 2; x[it].raux = 0; x[it].rinv = 1.0; }
        if (e >= n_in) x[it].v = 0;
    }
#pragma unroll
    for (int it = 0; it < N; it++) { const size_t e = base + it; x[it].det = (STAGE != 1) ? E.det[e < n_in ? e : last] : 0ull; }
}

#define FR_STG_SLOTS 8          // staged records per lane (the mean is 4: one per element)
#define FR_STG_SPILL 16384      // spill records per tile of 1024 elements (the head tiles of stages 2-4, where the heavy determinants' children keep most of their sub-weights, spill 7-9 000)
// where a lane of k_sys_count stages its emissions: record = {value (2 words), sub | item << 16, lane << 16 | ordinal within the lane}
struct StageOut { uint4 *slots; uint4 *spill; uint32_t *spill_n /* LDS */; uint32_t n_lane; uint32_t item; };
// Returns the number of emissions; the cursor advances over consumed teeth.  EMIT 1: writes (wi, sub, value) triples starting at slot `out`;
// EMIT 2: stages them through *so (k_sys_count); EMIT 0: counts only.
template <int STAGE, bool NEW_HB, int EMIT>
__device__ __forceinline__ uint32_t fr_sys_element(const CompWork &W, const HbTables &T, const Teeth *th, const ElemIn &x,
                                                   size_t e, double lbound, ToothCur &cur, double unit, double p_doub, size_t out, StageOut *so = nullptr) {
    const double v = x.v;
    if (v == 0) return 0;
    const uint32_t nd = x.nd, kp = x.kp;
    uint32_t n = 0;
    const double wr = x.wr;
    auto emit = [&](uint32_t sub, double val) {
        if (EMIT == 1) { size_t o = out + n; if (o < W.cap) { W.e_wi[o] = (uint32_t)e; W.e_sub[o] = sub; W.e_val[o] = val; } }
        if (EMIT == 2) {
            const unsigned long long vb = (unsigned long long)__double_as_longlong(val);
            const uint32_t ord = so->n_lane++;
            if (ord < FR_STG_SLOTS) { *so->slots = make_uint4((uint32_t)vb, (uint32_t)(vb >> 32), sub | so->item, 0u); so->slots += FR_BLOCK; }      // slot q of the workgroup's 256 lanes side by side: coalesced in k_sys_write
            else { const uint32_t q = atomicAdd(so->spill_n, 1u); if (q < FR_STG_SPILL) so->spill[q] = make_uint4((uint32_t)vb, (uint32_t)(vb >> 32), sub | so->item, (threadIdx.x << 16) | (ord & 0xffffu)); }
        }
        n++;
    };
    if (nd > 0) {
        if (kp & 1u) {
            double part = v / nd;
            for (uint32_t s = 0; s < nd; s++) emit(s, part);
        }
        else {
            while (cur.rn < lbound) {
                double q = (lbound - cur.rn) * nd / v;
                uint32_t s = q >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)q;
                if (s < nd) emit(s, unit);
                fr_cur_next(th, cur);
            }
        }
    }
    else {
        if (wr < v || cur.rn < lbound) {
            double sub_lbound = lbound - wr;
            RowInfo ri;
            if (STAGE == 1) ri = fr_row1(W.row1);
            else { ri.inv_norm = x.rinv; ri.aux = x.raux; ri.nsub = x.nsub; ri.tot = 0; }
            unsigned n_sub = fr_row_len<STAGE, NEW_HB>(T, ri.nsub);
            fr_row_visit<STAGE, NEW_HB>(T, x.det, x.code, ri, p_doub, [&](unsigned s, double w) {
                if (s >= n_sub) return;
                if (((kp >> s) & 1u) && w != 0) emit(s, v * w);
                else {
                    sub_lbound += v * w;
                    if (cur.rn < sub_lbound && w != 0) { emit(s, unit); fr_cur_next(th, cur); }
                }
            });
        }
    }
    return n;
}
// the same from a comb index (the repair kernels, which re-evaluate single elements)
template <int STAGE, bool NEW_HB, int EMIT>
__device__ __forceinline__ uint32_t fr_sys_element(const CompWork &W, const HbTables &T, const Teeth *th, const ElemIn &x,
                                                   size_t e, double lbound, uint32_t *k, double unit, double p_doub, size_t out) {
    ToothCur cur;
    fr_cur_seek_k(th, cur, *k);
    const uint32_t n = fr_sys_element<STAGE, NEW_HB, EMIT>(W, T, th, x, e, lbound, cur, unit, p_doub, out);
    *k = cur.k;
    return n;
}

// exact lbound prefix + per-element emission counts
template <int STAGE, bool NEW_HB>
__global__ void __launch_bounds__(FR_BLOCK) k_sys_count(CompWork W, VecDev V, const HbTables *Tg, int cur, double p_doub) {
    __shared__ HbTables T;
    __shared__ SeqShared seqsh;
    __shared__ uint32_t shu[4];
    __shared__ uint32_t sh_spill;
    if (W.seq.skip && *W.seq.skip) return;
    CompState *fin = &W.state[FR_MAX_ROUNDS + 1];
    const unsigned n_in = fin->n_in;
    const unsigned nblk = (n_in + FR_TILE - 1) / FR_TILE;
    if (blockIdx.x >= nblk) return;
    if (threadIdx.x == 0) sh_spill = 0;         // (the barriers of the staging below lie between this and the first emission)
    FR_SYS_T(0);
    if (STAGE != 1) fr_stage_tables(&T, Tg);
    __shared__ Teeth Tsh;
    fr_stage_teeth(&Tsh, W.teeth);
    const Teeth *th = &Tsh;
    const double unit = fin->unit;
    AccWt acc{W.wt_remain, &W.state[0]};
    double Sx[4], Sprev;
    FR_SYS_T(1);
    fr_seq_prefix4(W.seq, acc, blockIdx.x, &seqsh, Sx, &Sprev);
    FR_SYS_T(2);
    size_t base = (size_t)blockIdx.x * FR_TILE + (size_t)threadIdx.x * FR_ITEMS;
    ElemIn x[FR_ITEMS];
    fr_load_elems<STAGE, FR_ITEMS>(W, V, cur, base, n_in, x);
#ifdef FR_SYS_TIMING
    if (W.tdbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif
    FR_SYS_T(3);
    uint32_t cnt_t = 0;
    // The comb pointer an element starts from is T(lbound of its predecessor), the number of teeth below it (the assumption the repair
    // kernels check).  A lane looks that up once, for its first element; where an element leaves the comb exactly there for its successor --
    // the tooth it stands at is not below its lbound and the last one it took is -- the successor continues from the cursor.
    ToothCur tc;
    bool have = false;
    const bool staging = W.stg != nullptr;
    StageOut so;
    so.slots = staging ? W.stg + (size_t)blockIdx.x * FR_BLOCK * FR_STG_SLOTS + threadIdx.x : nullptr;
    so.spill = staging ? W.spill + (size_t)blockIdx.x * FR_STG_SPILL : nullptr;
    so.spill_n = &sh_spill; so.n_lane = 0; so.item = 0;
#pragma unroll
    for (int it = 0; it < FR_ITEMS; it++) {
        size_t e = base + it;
        if (e >= n_in) break;
        double Se = Sx[it];
        if (!have) { if (e == 0) fr_cur_seek_k(th, tc, 0u); else fr_cur_seek_below(th, tc, Sprev); }
        tc.last = -INFINITY;
        const uint32_t kin = tc.k;
        so.item = (uint32_t)it << 16;
        uint32_t c = staging ? fr_sys_element<STAGE, NEW_HB, 2>(W, T, th, x[it], e, Se, tc, unit, p_doub, 0, &so)
                             : fr_sys_element<STAGE, NEW_HB, 0>(W, T, th, x[it], e, Se, tc, unit, p_doub, 0);
        const uint32_t k = tc.k;
        W.S[e] = Se; W.kin[e] = kin; W.cnt[e] = c;
        cnt_t += c;
        if (W.prop) W.kend[e] = k;
        // k == T(Se)?  The teeth do not decrease: k teeth lie below Se iff tooth k does not and tooth k - 1 does (tooth kin - 1 < Sprev <= Se)
        have = tc.rn >= Se && (k == kin ? (e == 0 || Sprev <= Se) : tc.last < Se);
        if (!have) {      // the comb is not where the next lane assumes: repaired by k_sys_fixup / k_sys_prop
            have = k == fr_teeth_below(th, Se);         // (the short test is sufficient, not necessary)
            if (!have) {
                uint32_t slot = atomicAdd(&fin->n_fix, 1u);
                if (W.prop) { if (e + 1 < n_in) W.act[0][atomicAdd(&W.act_n[0], 1u)] = (uint32_t)(e + 1); }
                else if (slot < FR_MAX_FIX) W.fix_list[slot] = (uint32_t)e;
            }
        }
        Sprev = Se;
    }
    FR_SYS_T(4);
    uint32_t bc = fr_block_sum_u32(cnt_t, shu);
    if (threadIdx.x == 0) {
        W.pcnt[1][blockIdx.x] = bc;
        if (staging) { W.spill_cnt[blockIdx.x] = sh_spill; W.tile_dirty[blockIdx.x] = sh_spill > FR_STG_SPILL ? 1 : 0; }       // (the raw count: a flagged tile's list is not read)
    }
    FR_SYS_T(5);
}

// Repairs the (rare) elements after which the reference's comb lags behind lbound: walk
// forward sequentially from each flagged element until the tooth index re-synchronises.
template <int STAGE, bool NEW_HB>
__global__ void __launch_bounds__(FR_BLOCK) k_sys_fixup(CompWork W, VecDev V, const HbTables *Tg, int cur, double p_doub, uint32_t *err) {
    if (W.seq.skip && *W.seq.skip) return;
    CompState *fin = &W.state[FR_MAX_ROUNDS + 1];
    uint32_t nf = fin->n_fix;
    if (nf == 0) return;
    if (nf > FR_MAX_FIX) { if (threadIdx.x == 0) atomicOr(err, FR_ERR_BACKLOG); nf = FR_MAX_FIX; }
    const HbTables &T = *Tg;
    const Teeth *th = W.teeth;
    const unsigned n_in = fin->n_in;
    fr_sort_fix_list(W.fix_list, nf);       // ascending element index
    if (threadIdx.x != 0) return;
    size_t done_upto = 0;
    for (uint32_t i = 0; i < nf; i++) {
        size_t e = W.fix_list[i];
        if (e < done_upto) continue;
        uint32_t k = W.kin[e];
        ElemIn x1[1];
        fr_load_elems<STAGE, 1>(W, V, cur, e, n_in, x1);
        fr_sys_element<STAGE, NEW_HB, 0>(W, T, th, x1[0], e, W.S[e], &k, fin->unit, p_doub, 0);
        for (size_t e2 = e + 1; e2 < n_in; e2++) {
            uint32_t kin = k;
            fr_load_elems<STAGE, 1>(W, V, cur, e2, n_in, x1);
            uint32_t c = fr_sys_element<STAGE, NEW_HB, 0>(W, T, th, x1[0], e2, W.S[e2], &k, fin->unit, p_doub, 0);
            uint32_t old = W.cnt[e2];
            W.kin[e2] = kin; W.cnt[e2] = c;
            W.pcnt[1][e2 / FR_TILE] += c - old;
            if (W.tile_dirty) W.tile_dirty[e2 / FR_TILE] = 1;        // what k_sys_count staged for this tile is stale: written from the rows again
            done_upto = e2 + 1;
            if (k == fr_teeth_below(th, W.S[e2])) break;
        }
    }
}

// Parallel form of the repair, for inputs that need thousands of them (frisys_hh: its stage-1 rows {t, g} do not sum to one, so an
// element that was never examined keeps wt_remain = value while its sub-weights span 1.7 x value, and the row walk may take a tooth
// that lies beyond the element's own range, compress_utils.cpp:766-790).  Every lane of k_sys_count assumed the comb pointer
// T(lbound of its predecessor); an element whose predecessor left the comb elsewhere is re-evaluated from the pointer the
// predecessor really left, and if its own exit pointer changes its successor is re-evaluated in the next round.  The dependency
// only runs forward, so the rounds end with every element started from its predecessor's exit pointer: the sequential result.
template <int STAGE, bool NEW_HB>
__global__ void __launch_bounds__(FR_BLOCK) k_sys_prop(CompWork W, VecDev V, const HbTables *Tg, int cur, double p_doub, int in) {
    __shared__ HbTables T;
    __shared__ Teeth Tsh;
    const uint32_t n_act = W.act_n[in];
    if (blockIdx.x * blockDim.x >= n_act) return;        // rounds are enqueued in batches: most of a late batch finds nothing to do
    if (STAGE != 1) fr_stage_tables(&T, Tg);
    fr_stage_teeth(&Tsh, W.teeth);
    CompState *fin = &W.state[FR_MAX_ROUNDS + 1];
    const unsigned n_in = fin->n_in;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_act; i += gridDim.x * blockDim.x) {
        const uint32_t e = W.act[in][i];
        const uint32_t kin_new = W.kend[e - 1];
        if (kin_new == W.kin[e]) continue;
        ElemIn x1[1];
        fr_load_elems<STAGE, 1>(W, V, cur, e, n_in, x1);
        uint32_t k = kin_new;
        const uint32_t c = fr_sys_element<STAGE, NEW_HB, 0>(W, T, &Tsh, x1[0], e, W.S[e], &k, fin->unit, p_doub, 0);
        const uint32_t old = W.cnt[e];
        W.kin[e] = kin_new; W.cnt[e] = c;
        if (c != old) atomicAdd(&W.pcnt[1][e / FR_TILE], c - old);
        if (k != W.kend[e]) {
            W.kend[e] = k;
            if (e + 1 < n_in) W.act[in ^ 1][atomicAdd(&W.act_n[in ^ 1], 1u)] = e + 1;
        }
    }
}

// The same repair with every chain followed by ONE lane: a lane takes an element of the list and keeps walking forward while the
// comb pointer it hands on keeps changing, instead of handing each step to the next round (a chain of 300 elements then costs one
// launch, not 300).  Segments stay disjoint: every list element carries this round's tag in W.kstart, and a walker that reaches a
// tagged element stops there and queues it for the next round (its owner may have started from the pointer that has just moved).
// The number of rounds is the number of times chains run into one another, not their length.
static __global__ void __launch_bounds__(FR_BLOCK) k_sys_mark(CompWork W, int in, uint32_t tag) {
    const uint32_t n_act = W.act_n[in];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_act; i += gridDim.x * blockDim.x) W.kstart[W.act[in][i]] = tag;
}
template <int STAGE, bool NEW_HB>
__global__ void __launch_bounds__(FR_BLOCK) k_sys_walk(CompWork W, VecDev V, const HbTables *Tg, int cur, double p_doub, int in, uint32_t tag) {
    __shared__ HbTables T;
    __shared__ Teeth Tsh;
    const uint32_t n_act = W.act_n[in];
    if (blockIdx.x * blockDim.x >= n_act) return;
    if (STAGE != 1) fr_stage_tables(&T, Tg);
    fr_stage_teeth(&Tsh, W.teeth);
    CompState *fin = &W.state[FR_MAX_ROUNDS + 1];
    const unsigned n_in = fin->n_in;
    const double unit = fin->unit;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_act; i += gridDim.x * blockDim.x) {
        uint32_t e = W.act[in][i];
        while (true) {
            const uint32_t kin_new = W.kend[e - 1];
            if (kin_new == W.kin[e]) break;
            ElemIn x1[1];
            fr_load_elems<STAGE, 1>(W, V, cur, e, n_in, x1);
            uint32_t k = kin_new;
            const uint32_t c = fr_sys_element<STAGE, NEW_HB, 0>(W, T, &Tsh, x1[0], e, W.S[e], &k, unit, p_doub, 0);
            const uint32_t old = W.cnt[e];
            W.kin[e] = kin_new; W.cnt[e] = c;
            if (c != old) atomicAdd(&W.pcnt[1][e / FR_TILE], c - old);
            if (k == W.kend[e]) break;              // the comb leaves this element where it did before: the chain ends
            W.kend[e] = k;
            e++;
            if (e >= n_in) break;
            if (W.kstart[e] == tag) { W.act[in ^ 1][atomicAdd(&W.act_n[in ^ 1], 1u)] = e; break; }     // somebody else's start: next round
        }
    }
}

template <int STAGE, bool NEW_HB>
__global__ void __launch_bounds__(FR_BLOCK) k_sys_write(CompWork W, VecDev V, const HbTables *Tg, int cur, double p_doub, uint32_t *err, uint32_t *host_out) {
    __shared__ HbTables T;
    __shared__ uint32_t shu[4];
    CompState *fin = &W.state[FR_MAX_ROUNDS + 1];
    const unsigned n_in = fin->n_in;
    const unsigned nblk = (n_in + FR_TILE - 1) / FR_TILE;
    if (blockIdx.x >= nblk) return;
    FR_SYS_T(0);
    const bool copy_path = W.stg && !W.tile_dirty[blockIdx.x];        // (uniform over the workgroup) the staged emissions stand: no table, no comb needed
    if (!copy_path && STAGE != 1) fr_stage_tables(&T, Tg);
    __shared__ Teeth Tsh;
    if (!copy_path) fr_stage_teeth(&Tsh, W.teeth);
    const Teeth *th = &Tsh;
    FR_SYS_T(1);
    // copy path: the lane's staged records and the tile's spill count are requested before anything else (whether a slot is in use is only
    // known after the counts and two block scans: fetched then, every step would be a memory round trip of its own)
    uint4 rec[FR_STG_SLOTS];
    uint32_t n_sp = 0;
    if (copy_path) {
        const uint4 *slots = W.stg + (size_t)blockIdx.x * FR_BLOCK * FR_STG_SLOTS + threadIdx.x;
#pragma unroll
        for (int q = 0; q < FR_STG_SLOTS; q++) rec[q] = slots[(size_t)q * FR_BLOCK];
        n_sp = W.spill_cnt[blockIdx.x];
    }
    const uint32_t *pc = W.pcnt[1];
    uint32_t off;
    {
        uint32_t x = 0;
        for (unsigned i = threadIdx.x; i < blockIdx.x; i += blockDim.x) x += pc[i];
        off = fr_block_sum_u32(x, shu);
    }
    size_t base = (size_t)blockIdx.x * FR_TILE + (size_t)threadIdx.x * FR_ITEMS;
    uint32_t c[FR_ITEMS], tsum = 0;
#pragma unroll
    for (int it = 0; it < FR_ITEMS; it++) { size_t e = base + it; c[it] = e < n_in ? W.cnt[e] : 0; tsum += c[it]; }
    uint32_t tot;
    uint32_t incl = fr_block_scan_u32(tsum, shu, &tot);
    size_t o = (size_t)off + (incl - tsum);
    FR_SYS_T(2);
    if (copy_path) {
        FR_SYS_T(3);
        // the emissions as k_sys_count staged them: a copy, no row is evaluated again
        __shared__ uint32_t sh_o[FR_BLOCK];
        sh_o[threadIdx.x] = (uint32_t)o;
        const uint32_t n_st = tsum < FR_STG_SLOTS ? tsum : FR_STG_SLOTS;
#pragma unroll
        for (int q = 0; q < FR_STG_SLOTS; q++) if ((uint32_t)q < n_st) {
            const size_t p = o + (size_t)q;
            if (p < W.cap) { W.e_wi[p] = (uint32_t)(base + (rec[q].z >> 16)); W.e_sub[p] = rec[q].z & 0xffffu; W.e_val[p] = __longlong_as_double((long long)(((unsigned long long)rec[q].y << 32) | rec[q].x)); }
        }
#pragma unroll
        for (int it = 0; it < FR_ITEMS; it++) { const size_t e = base + it; if (e < n_in) W.keep[e] = 0; }
        __syncthreads();
        const uint4 *sp = W.spill + (size_t)blockIdx.x * FR_STG_SPILL;
        const size_t tile_base = (size_t)blockIdx.x * FR_TILE;
        // (eight records per lane in flight: a head tile lists ~9 000, and one load per trip would make the loop a chain of memory latencies)
        const uint32_t n_list = n_sp < FR_STG_SPILL ? n_sp : FR_STG_SPILL;
        for (uint32_t i0 = threadIdx.x; i0 < n_list; i0 += 8 * FR_BLOCK) {
            uint4 r[8];
#pragma unroll
            for (int k = 0; k < 8; k++) { const uint32_t i = i0 + (uint32_t)k * FR_BLOCK; r[k] = sp[i < n_list ? i : n_list - 1]; }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (i0 + (uint32_t)k * FR_BLOCK >= n_list) break;
                const uint32_t lane_t = r[k].w >> 16;
                const size_t p = (size_t)sh_o[lane_t] + (r[k].w & 0xffffu);
                if (p < W.cap) { W.e_wi[p] = (uint32_t)(tile_base + (size_t)lane_t * FR_ITEMS + (r[k].z >> 16)); W.e_sub[p] = r[k].z & 0xffffu; W.e_val[p] = __longlong_as_double((long long)(((unsigned long long)r[k].y << 32) | r[k].x)); }
            }
        }
        o += tsum;
    }
    else {
    ElemIn x[FR_ITEMS];
    fr_load_elems<STAGE, FR_ITEMS>(W, V, cur, base, n_in, x);
    uint32_t kin4[FR_ITEMS]; double S4[FR_ITEMS];
#pragma unroll
    for (int it = 0; it < FR_ITEMS; it++) { size_t e = base + it; size_t ec = e < n_in ? e : (n_in ? n_in - 1 : 0); kin4[it] = W.kin[ec]; S4[it] = W.S[ec]; }
#ifdef FR_SYS_TIMING
    if (W.tdbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif
    FR_SYS_T(3);
    ToothCur tc;
    bool have = false;
#pragma unroll
    for (int it = 0; it < FR_ITEMS; it++) {
        size_t e = base + it;
        if (e >= n_in) break;
        if (c[it]) {
            if (!have || tc.k != kin4[it]) fr_cur_seek_k(th, tc, kin4[it]);
            have = true;
            fr_sys_element<STAGE, NEW_HB, 1>(W, T, th, x[it], e, S4[it], tc, fin->unit, p_doub, o);
        }
        W.keep[e] = 0;
        o += c[it];
    }
    }
    FR_SYS_T(4);
    if (blockIdx.x == nblk - 1 && threadIdx.x == FR_BLOCK - 1) {
        fin->n_out = (uint32_t)o;
        if (host_out) *host_out = (uint32_t)o;          // host-coherent pinned memory: read by the host after its next wait on the stream
        if (o > W.cap) atomicOr(err, FR_ERR_SPAWN_CAP);
    }
}
