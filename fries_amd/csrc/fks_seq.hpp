// find_keep_sub in the reference's own order, for stages the parallel replay cannot decide reliably.
//
// The replay (fks2.hpp) forms every threshold as "norm entering the sweep minus a prefix sum of the removed weight".  When a
// sweep removes almost the whole norm -- the start-up regime, where the sample budget exceeds the number of sub-elements and
// everything is preserved -- the running norm of the reference (compress_utils.cpp:190-192, 246-247: one subtraction per kept
// element, in storage order) ends up 10-20 orders of magnitude below where it started, i.e. it IS its own accumulated
// rounding error, and that noise decides the last few comparisons and the norm the next sweep starts from.  A prefix sum
// cannot reproduce it.  So when the settled replay reports a collapse (remaining norm < 1e-3 of the norm entering the
// stage) the stage is redone here: one wave walks the elements in order and performs the reference's subtractions one by
// one.  Sweep scalars travel over the ranks exactly as in the reference (sum_mpi before and after every sweep).
// Steady-state stages (remaining norm ~ 0.3 of the start) never come here; this path is slow by design (~30 ms per sweep
// and million elements) and only has to be right.
#pragma once
#include "fks2.hpp"

struct FksSeq {
    double L;               // loc_one_norm
    double G;               // glob_one_norm: sum over the ranks at the sweep's start, then this rank's running value
    uint32_t K;             // loc_sampled of the sweep
    uint32_t n_samp;        // budget entering the sweep
    int32_t last_pass;
    int32_t go;             // 1: run another sweep
    int32_t resum;          // 1: loc_one_norm must be re-summed from wt_remain before the next sweep
    int32_t n_sweeps;
    uint32_t glob_sampled;
};

// wt_remain <- value, nothing preserved (find_keep_sub's first loop, compress_utils.cpp:136-139)
static __global__ void __launch_bounds__(FR_BLOCK) k_fks_seq_reset(CompWork W, int cur) {
    const unsigned n_in = W.state[0].n_in;
    const StageElems E = W.el[cur];
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_in; e += (size_t)gridDim.x * blockDim.x) { W.wt_remain[e] = E.val[e]; W.keep[e] = 0; }
}
struct AccVal {
    const double *val; const CompState *st0;
    __device__ unsigned count() const { return st0->n_in; }
    __device__ double get(size_t i) const { return val[i]; }
};
static __global__ void k_fks_seq_begin(CompWork W, FksSeq *Q, const double *loc_total, double *send) {
    Q->L = *loc_total; Q->G = 0; Q->K = 0; Q->n_samp = W.state[0].n_rem; Q->last_pass = 0; Q->go = 1; Q->resum = 0; Q->n_sweeps = 0; Q->glob_sampled = 1;
    *send = Q->L;
}
// glob_one_norm = sum_mpi(loc_one_norm) (compress_utils.cpp:154), rank order
static __global__ void k_fks_seq_norm(FksSeq *Q, const double *all, int n_ranks) {
    double g = 0;
    for (int r = 0; r < n_ranks; r++) g += all[r];
    Q->G = g; Q->K = 0;
    if (g < 0) Q->go = 0;           // :155-157
}
static __global__ void k_fks_seq_put_k(FksSeq *Q, uint32_t *send) { *send = Q->K; }
// after the sweep: glob_sampled = sum_mpi(loc_sampled) and the last_pass bookkeeping (:251-265)
static __global__ void k_fks_seq_post(FksSeq *Q, const uint32_t *all, int n_ranks, double *send) {
    if (!Q->go) return;             // the loop was left at its top (negative norm): nothing after that line runs
    uint32_t gs = 0;
    for (int r = 0; r < n_ranks; r++) gs += all[r];
    Q->n_samp -= gs;
    Q->n_sweeps++;
    Q->resum = 0;
    if (Q->last_pass && gs) Q->last_pass = 0;
    if (gs == 0 && !Q->last_pass) { Q->last_pass = 1; gs = 1; Q->resum = 1; }
    Q->glob_sampled = gs;
    Q->go = gs > 0 ? 1 : 0;
    *send = Q->L;
}
static __global__ void k_fks_seq_take_resum(FksSeq *Q, const double *loc_total, double *send) { Q->L = *loc_total; *send = Q->L; }
// hands the result to the rest of the stage (k_comp_finalize2 reads G_last / n_last / n_pass)
static __global__ void k_fks_seq_end(FksSeq *Q, Fks2Work F) {
    FksScal *S = F.scal;
    S->G_last = Q->G; S->n_last = Q->n_samp; S->n_pass = Q->n_sweeps;
    F.saved->valid = 0;             // no warm start from a stage that went through here
}

// start_tile > 0: the walk takes over at element start_tile * 256 with the state the parallel form (below) established there
template <int STAGE, bool NEW_HB>
__global__ void __launch_bounds__(64) k_fks_seq_sweep(CompWork W, const HbTables *Tg, int cur, double p_doub, FksSeq *Q,
                                                      uint32_t start_tile, const double *gb, const double *lb, const uint32_t *kb) {
    __shared__ HbTables T;
    if (STAGE != 1) fr_stage_tables(&T, Tg); else __syncthreads();
    if (!Q->go) return;
    const unsigned n_in = W.state[0].n_in;
    const StageElems E = W.el[cur];
    const int lane = fr_lane(), f = lane & 7;
    double G = Q->G, L = Q->L;
    uint32_t K = 0;
    if (start_tile) { G = gb[(size_t)start_tile * 32]; L = lb[(size_t)start_tile * 32]; K = kb[(size_t)start_tile * 32]; }
    const size_t e_start = (size_t)start_tile * 256;
    const uint32_t n_samp = Q->n_samp;
    // batch of 64 elements = 8 of the reference's blocks; the next batch's loads are in flight while this one is walked
    double v_n = 0, wr_n = 0; uint32_t nd_n = 1, kp_n = 0;
    auto issue = [&](size_t base) {
        const size_t e = base + lane;
        const bool lv = e < n_in;
        v_n = lv ? E.val[e] : 0.0; nd_n = lv ? E.ndiv[e] : 1u; wr_n = lv ? W.wt_remain[e] : 0.0; kp_n = lv ? W.keep[e] : 0u;
    };
    issue(e_start);
    for (size_t base = e_start; base < n_in; base += 64) {
        const size_t e = base + lane;
        const bool live = e < n_in;
        const double v = v_n; const uint32_t nd = nd_n;
        double wr = wr_n; uint32_t kp = kp_n;
        if (base + 64 < n_in) issue(base + 64);
        bool touched = false;
        int b = 0;
        while (b < 8) {
            // flags of every remaining block under the current (norm, budget): exact for the first block that has one
            const double wf = (double)(n_samp - K);
            double cw = v * wf;
            if (nd > 0) cw /= nd;
            const bool flag = live && wr > 0 && cw >= G && (lane >> 3) >= b;
            const unsigned long long mask = __ballot(flag);
            if (!mask) break;
            const int fb = (__ffsll((long long)mask) - 1) >> 3;
            const unsigned bm = (unsigned)((mask >> (8 * fb)) & 0xffull);
            const bool mine = flag && (lane >> 3) == fb;
            // speculative evaluation of the block's flagged rows with the norm at the block's start ...
            uint32_t new_kp = kp, add = 0;
            double new_wr = wr, change = 0, mu = 0, mk = INFINITY, used_G = G;
            RowInfo ri = fr_row1(W.row1);
            det_t det = 0; uint32_t code = 0;
            unsigned n_sub = 2;
            auto eval_row = [&](double gl) {
                const unsigned full = (n_sub / 8) * 8;
                uint32_t kk = kp, a = 0;
                double rem = 0, m = 0, k1 = INFINITY;
                fr_row_visit<STAGE, NEW_HB>(T, det, code, ri, p_doub, [&](unsigned s, double w) {
                    if (s >= n_sub || ((kk >> s) & 1u)) return;
                    const double sub_magn = cw * w;
                    const double thr = s < full ? 1e-12 : 1e-10;          // compress_utils.cpp:213 / :233
                    if (sub_magn >= gl && fabs(sub_magn) > thr) { kk |= 1u << s; a++; k1 = sub_magn < k1 ? sub_magn : k1; }
                    else { rem += sub_magn; m = sub_magn > m ? sub_magn : m; }
                });
                rem /= wf;                                                 // :243
                new_kp = kk; add = a; new_wr = rem; change = wr - rem; mu = m; mk = k1; used_G = gl;
            };
            if (mine) {
                if (nd > 0) { new_kp = kp | 1u; new_wr = 0; add = nd; change = v; }
                else {
                    if (STAGE != 1) { code = E.code[e]; det = E.det[e]; ri = fr_row_cached(E, e); }
                    n_sub = fr_row_len<STAGE, NEW_HB>(T, ri.nsub);
                    eval_row(G);
                }
            }
            // ... then the reference's order: one flagged element after the other, each against the running norm
            for (unsigned m = bm; m; m &= m - 1) {
                const int src = fb * 8 + (__ffs((int)m) - 1);
                if (lane == src && nd == 0 && used_G != G && (mu >= G || mk < G)) eval_row(G);      // a sub-weight between the two norms: redo
                const uint32_t dK = (uint32_t)__builtin_amdgcn_readlane((int)add, src);
                const double dL = fr_bcast_f64(change, src);
                const uint32_t unif = (uint32_t)__builtin_amdgcn_readlane((int)nd, src);
                K += dK; L = L - dL; G = G - dL;
                if (lane == src) { kp = new_kp; wr = new_wr; touched = true; }
                if (unif > 0 && G < 0) break;                              // :193-195: leaves the block's remaining flagged elements alone
            }
            b = fb + 1;
        }
        if (touched) { W.keep[e] = kp; W.wt_remain[e] = wr; }
    }
    if (lane == 0) { Q->L = L; Q->G = G; Q->K = K; }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same sweep, parallel (round 3).  What is sequential in a sweep is one chain of subtractions: G <- G - change(e) for every
// element the sweep touches, in storage order; everything else -- flags, rows, preserved bits, the remaining weights -- is a function
// of the state (G, K) a block of 8 elements is entered with (K = loc_sampled, an integer prefix sum).  So:
//   1. k_fsq_spec   every block of 8 decides from an assumed entry state (gb[b], kb[b]) -- the walk's own arithmetic, the in-block
//                   order settled over DPP as in the replay -- and writes its elements' new state and change aside;
//   2. k_fsq_scan / k_fsq_expand   kb[] = exact prefix sums of the blocks' counts (and, while the entry norms are still guesses,
//                   gb[] = start norm minus a tree-summed prefix of the changes: good to a rounding error, which is all a guess needs);
//   3. k_fsq_chain  ONE wave performs the reference's subtractions on the changes (a stream of doubles read through the scalar
//                   cache: two dependent v_add_f64 per element and nothing else) and leaves the exact entry norm of every block;
//   4. k_fsq_spec again with the exact entry states.  The first block whose output differs from the one the chain was run on is
//                   where the assumed sequence left the true one: everything before it is final (its inputs came from final
//                   outputs), and steps 2-4 repeat from that tile.  No difference: the sweep is done, k_fsq_commit stores it.
// Steps 1-2 are iterated first (cheap) until the decisions stop moving, so the chain usually runs once and the comparison in 4 finds
// differences only where the running norm has become its own rounding noise (the last blocks of a collapsing sweep).  If the exact
// rounds do not close within a few repetitions the one-wave walk above takes over from the last confirmed tile, so the result is the
// reference's in every case and only the time varies.
// (FksSq / FksSqCtl / FR_SQ_TILE: fks2.hpp, next to the replay's work arrays, because the context holds one)
static __global__ void __launch_bounds__(FR_BLOCK) k_fsq_init(CompWork W, const FksSeq *Q, FksSq SQ) {
    if (!Q->go) return;
    const unsigned n_in = W.state[0].n_in;
    const size_t e = (size_t)blockIdx.x * FR_SQ_TILE + threadIdx.x;
    const bool live = e < n_in;
    SQ.dl[e] = 0.0; SQ.nwr[e] = live ? W.wt_remain[e] : 0.0; SQ.nkp[e] = live ? W.keep[e] : 0u;
    if ((threadIdx.x & 7) == 0) { const size_t b = e >> 3; SQ.gb[b] = Q->G; SQ.lb[b] = Q->L; SQ.kb[b] = 0u; SQ.dk[b] = 0u; SQ.dgb[b] = 0.0; }
    if (threadIdx.x == 0) { SQ.tk[blockIdx.x] = 0u; SQ.tg[blockIdx.x] = 0.0; SQ.tany[blockIdx.x] = 0; }
    if (e == 0) { SQ.ctl->first_changed = FR_SQ_INF; SQ.ctl->K_tot = 0u; SQ.ctl->G_end = Q->G; SQ.ctl->L_end = Q->L; }
}

template <int STAGE, bool NEW_HB>
__global__ void __launch_bounds__(FR_BLOCK) k_fsq_spec(CompWork W, const HbTables *Tg, int cur, double p_doub, const FksSeq *Q, FksSq SQ, uint32_t from_tile) {
    __shared__ HbTables T;
    __shared__ uint32_t sh_k[4]; __shared__ double sh_g[4]; __shared__ uint32_t sh_any[4];
    if (STAGE != 1) fr_stage_tables(&T, Tg); else __syncthreads();
    if (!Q->go) return;
    const unsigned n_in = W.state[0].n_in;
    const StageElems E = W.el[cur];
    const unsigned tile = from_tile + blockIdx.x;
    const size_t e = (size_t)tile * FR_SQ_TILE + threadIdx.x;
    const size_t b = e >> 3;
    const int lane = fr_lane(), f = lane & 7, wv = threadIdx.x >> 6;
    const bool live = e < n_in;
    double v = 0, wr = 0; uint32_t nd = 1, kp = 0;
    if (live) { v = E.val[e]; nd = E.ndiv[e]; wr = W.wt_remain[e]; kp = W.keep[e]; }
    const double Gb = SQ.gb[b];
    const double wf = (double)(Q->n_samp - SQ.kb[b]);
    double cw = v * wf;
    if (nd > 0) cw /= nd;
    bool flagged = live && wr > 0 && cw >= Gb;                  // compress_utils.cpp:172-180: the block's flags against the norm it is entered with
    uint32_t new_kp = kp, add = 0;
    double new_wr = wr, change = 0, mu = 0, mk = INFINITY, used_G = Gb;
    if (__any(flagged)) {
        RowInfo ri = fr_row1(W.row1);
        det_t det = 0; uint32_t code = 0;
        unsigned n_sub = 2;
        bool need_eval = false;
        if (flagged) {
            if (nd > 0) { new_kp = kp | 1u; new_wr = 0; add = nd; change = v; }
            else {
                if (STAGE != 1) { code = E.code[e]; det = E.det[e]; ri = fr_row_cached(E, e); }
                n_sub = fr_row_len<STAGE, NEW_HB>(T, ri.nsub);
                need_eval = true;
            }
        }
        double gl = Gb;
        for (int round = 0; round < 9; round++) {               // lane f's running norm is final after f rounds
            if (need_eval) {
                const unsigned full = (n_sub / 8) * 8;
                uint32_t kk = kp, a = 0;
                double rem = 0, m = 0, k1 = INFINITY;
                fr_row_visit<STAGE, NEW_HB>(T, det, code, ri, p_doub, [&](unsigned s, double w) {
                    if (s >= n_sub || ((kk >> s) & 1u)) return;
                    const double sub_magn = cw * w;
                    const double thr = s < full ? 1e-12 : 1e-10;          // compress_utils.cpp:213 / :233
                    if (sub_magn >= gl && fabs(sub_magn) > thr) { kk |= 1u << s; a++; k1 = sub_magn < k1 ? sub_magn : k1; }
                    else { rem += sub_magn; m = sub_magn > m ? sub_magn : m; }
                });
                rem /= wf;                                                 // :243
                new_kp = kk; add = a; new_wr = rem; change = wr - rem; mu = m; mk = k1; used_G = gl;
            }
            gl = fr_grp8_running(Gb, flagged ? change : 0.0, f);
            need_eval = flagged && nd == 0 && used_G != gl && (mu >= gl || mk < gl);
            if (!__any(need_eval)) break;
        }
        // :193-195: a uniform element that drives the norm below zero ends the block; the flagged elements behind it stay as they were
        const bool brk = flagged && nd > 0 && gl - change < 0;
        const unsigned bm = (unsigned)((__ballot(brk) >> (lane & 56)) & 0xffull);
        if (bm && f > __ffs((int)bm) - 1) flagged = false;
    }
    if (!flagged) { new_kp = kp; new_wr = wr; add = 0; change = 0.0; }
    const bool changed = live && (__double_as_longlong(change) != __double_as_longlong(SQ.dl[e]) || new_kp != SQ.nkp[e] ||
                                  __double_as_longlong(new_wr) != __double_as_longlong(SQ.nwr[e]));
    const unsigned long long cm = __ballot(changed);
    if (cm && lane == 0) atomicMin(&SQ.ctl->first_changed, (uint32_t)((e + (size_t)(__ffsll((long long)cm) - 1)) >> 3));
    SQ.dl[e] = change; SQ.nkp[e] = new_kp; SQ.nwr[e] = new_wr;
    const uint32_t gk = fr_grp8_sum_u32(add);
    const double gg = fr_grp8_sum(change);
    if (f == 0) { SQ.dk[b] = gk; SQ.dgb[b] = gg; }
    // tile totals
    uint32_t wk = (f == 0) ? gk : 0u; double wg = (f == 0) ? gg : 0.0;
    for (int o = 32; o >= 8; o >>= 1) { wk += __shfl_down(wk, o); wg += __shfl_down(wg, o); }
    const unsigned long long fm = __ballot(flagged);
    if (lane == 0) { sh_k[wv] = wk; sh_g[wv] = wg; sh_any[wv] = fm ? 1u : 0u; }
    __syncthreads();
    if (threadIdx.x == 0) {
        SQ.tk[tile] = sh_k[0] + sh_k[1] + sh_k[2] + sh_k[3];
        SQ.tg[tile] = (sh_g[0] + sh_g[1]) + (sh_g[2] + sh_g[3]);
        SQ.tany[tile] = (uint8_t)(sh_any[0] | sh_any[1] | sh_any[2] | sh_any[3]);
    }
}

// exclusive prefixes of the tiles' counts (exact) and changes (tree sums) from from_tile on; one workgroup
static __global__ void __launch_bounds__(1024) k_fsq_scan(FksSq SQ, uint32_t from_tile, uint32_t n_tiles) {
    __shared__ uint32_t wk[16]; __shared__ double wg[16];
    __shared__ uint32_t carry_k; __shared__ double carry_g;
    const int lane = fr_lane(), wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) { carry_k = SQ.kb[(size_t)from_tile * 32]; carry_g = 0.0; }
    const double g0 = SQ.gb[(size_t)from_tile * 32];
    __syncthreads();
    for (uint32_t base = from_tile; base < n_tiles; base += 1024) {
        const uint32_t t = base + threadIdx.x;
        const uint32_t k = t < n_tiles ? SQ.tk[t] : 0u; const double g = t < n_tiles ? SQ.tg[t] : 0.0;
        uint32_t ik = k; double ig = g;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t a = __shfl_up(ik, o); const double c = __shfl_up(ig, o); if (lane >= o) { ik += a; ig += c; } }
        if (lane == 63) { wk[wv] = ik; wg[wv] = ig; }
        __syncthreads();
        uint32_t pk = carry_k; double pg = carry_g;
        for (int q = 0; q < wv; q++) { pk += wk[q]; pg += wg[q]; }
        if (t < n_tiles) { SQ.tkx[t] = pk + ik - k; SQ.tgx[t] = g0 - (pg + ig - g); }
        __syncthreads();
        if (threadIdx.x == 1023) { carry_k = pk + ik; carry_g = pg + ig; }
        __syncthreads();
    }
    if (threadIdx.x == 0) SQ.ctl->K_tot = carry_k;
}
// entry count (and, approx != 0, entry norm) of every block of the tiles from from_tile on
static __global__ void __launch_bounds__(FR_BLOCK) k_fsq_expand(FksSq SQ, uint32_t from_tile, uint32_t n_tiles, int approx) {
    const uint32_t t = from_tile + blockIdx.x * (FR_BLOCK / 32) + (threadIdx.x >> 5);
    const int j = threadIdx.x & 31;
    const bool on = t < n_tiles;
    const size_t b = (size_t)(on ? t : from_tile) * 32 + j;
    const uint32_t k = SQ.dk[b]; const double g = SQ.dgb[b];
    uint32_t ik = k; double ig = g;
    for (int o = 1; o < 32; o <<= 1) { const uint32_t a = __shfl_up(ik, o, 32); const double c = __shfl_up(ig, o, 32); if (j >= o) { ik += a; ig += c; } }
    if (!on) return;
    SQ.kb[b] = SQ.tkx[t] + ik - k;
    if (approx) SQ.gb[b] = SQ.tgx[t] - (ig - g);
}

// The reference's running norms (compress_utils.cpp:190-192, 246-247: glob_one_norm -= change, loc_one_norm -= change per touched element,
// in storage order), given the changes: one wave.  Lanes 0-31 carry the global norm, lanes 32-63 the local one, so one dependent
// v_add_f64 per element advances both; the changes of a tile reach every lane as LDS broadcast reads (16 bytes = two elements each), the
// next tile's are in flight meanwhile.  Lane j (and 32 + j) keeps the value block j of the tile is entered with.
static __global__ void __launch_bounds__(64) k_fsq_chain(const FksSeq *Q, const double *__restrict__ dl, const uint8_t *__restrict__ tany,
                                                         double *__restrict__ gb, double *__restrict__ lb, FksSqCtl *__restrict__ ctl,
                                                         uint32_t from_tile, uint32_t n_tiles, int sparse_max) {
    __shared__ double2 sh[FR_SQ_TILE / 2];
    if (!Q->go) return;
    const int lane = fr_lane(), j = lane & 31;
    const bool is_loc = lane >= 32;
    double X;
    if (from_tile == 0) X = is_loc ? Q->L : Q->G; else X = is_loc ? lb[(size_t)from_tile * 32] : gb[(size_t)from_tile * 32];
    double *const out = is_loc ? lb : gb;
    double2 n0 = make_double2(0, 0), n1 = n0; uint32_t n_any = 0;
    auto issue = [&](uint32_t t) {
        if (t < n_tiles) { const double2 *p = (const double2 *)(dl + (size_t)t * FR_SQ_TILE); n0 = p[lane]; n1 = p[64 + lane]; n_any = tany[t]; }
        else n_any = 0;
    };
    issue(from_tile);
    for (uint32_t t = from_tile; t < n_tiles; t++) {
        const double2 c0 = n0, c1 = n1;
        const uint32_t any = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_any);
        issue(t + 1);
        double C = X;
        unsigned long long m0 = 0, m1 = 0;
        if (any) { m0 = __ballot(c0.x != 0.0 || c0.y != 0.0); m1 = __ballot(c1.x != 0.0 || c1.y != 0.0); }
        if (any && __popcll(m0) + __popcll(m1) <= sparse_max) {
            // few touched elements (the later sweeps of a stage): only the pairs that hold one, in order; lane i of c0 holds elements
            // 2i, 2i + 1 of the tile (block i / 4), lane i of c1 elements 128 + 2i, 129 + 2i (block 16 + i / 4)
            for (int h = 0; h < 2; h++) {
                unsigned long long m = h ? m1 : m0;
                const double2 src = h ? c1 : c0;
                while (m) {
                    const int i = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    X -= fr_bcast_f64(src.x, i); X -= fr_bcast_f64(src.y, i);
                    if (j > (i >> 2) + 16 * h) C = X;          // the blocks behind this pair are entered with the new value
                }
            }
        }
        else if (any) {
            sh[lane] = c0; sh[64 + lane] = c1;              // one wave: its LDS accesses execute in program order
            // the reads of two blocks ahead are issued before a block's eight subtractions
            double2 r[3][4];
#pragma unroll
            for (int q = 0; q < 2; q++)
#pragma unroll
                for (int k = 0; k < 4; k++) r[q][k] = sh[q * 4 + k];
#pragma unroll
            for (int q = 0; q < 32; q++) {
                if (q + 2 < 32) {
#pragma unroll
                    for (int k = 0; k < 4; k++) r[(q + 2) % 3][k] = sh[(q + 2) * 4 + k];
                }
                if (j == q) C = X;
                const double2 *a = r[q % 3];
                X -= a[0].x; X -= a[0].y; X -= a[1].x; X -= a[1].y; X -= a[2].x; X -= a[2].y; X -= a[3].x; X -= a[3].y;
            }
        }
        out[(size_t)t * 32 + j] = C;
    }
    if (j == 0) { if (is_loc) ctl->L_end = X; else ctl->G_end = X; }
}

// stores the sweep: elements of the tiles before upto_tile; fin: the whole sweep was settled here (else the walk finishes it and leaves the scalars)
static __global__ void __launch_bounds__(FR_BLOCK) k_fsq_commit(CompWork W, FksSeq *Q, FksSq SQ, uint32_t upto_tile, int fin) {
    if (!Q->go) return;
    const unsigned n_in = W.state[0].n_in;
    const size_t e = (size_t)blockIdx.x * FR_SQ_TILE + threadIdx.x;
    if (blockIdx.x < upto_tile && e < n_in && SQ.tany[blockIdx.x]) { W.keep[e] = SQ.nkp[e]; W.wt_remain[e] = SQ.nwr[e]; }
    if (fin && e == 0) { Q->G = SQ.ctl->G_end; Q->L = SQ.ctl->L_end; Q->K = SQ.ctl->K_tot; }
}
