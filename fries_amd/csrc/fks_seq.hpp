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

template <int STAGE, bool NEW_HB>
__global__ void __launch_bounds__(64) k_fks_seq_sweep(CompWork W, const HbTables *Tg, int cur, double p_doub, FksSeq *Q) {
    __shared__ HbTables T;
    if (STAGE != 1) fr_stage_tables(&T, Tg); else __syncthreads();
    if (!Q->go) return;
    const unsigned n_in = W.state[0].n_in;
    const StageElems E = W.el[cur];
    const int lane = fr_lane(), f = lane & 7;
    double G = Q->G, L = Q->L;
    uint32_t K = 0;
    const uint32_t n_samp = Q->n_samp;
    // batch of 64 elements = 8 of the reference's blocks; the next batch's loads are in flight while this one is walked
    double v_n = 0, wr_n = 0; uint32_t nd_n = 1, kp_n = 0;
    auto issue = [&](size_t base) {
        const size_t e = base + lane;
        const bool lv = e < n_in;
        v_n = lv ? E.val[e] : 0.0; nd_n = lv ? E.ndiv[e] : 1u; wr_n = lv ? W.wt_remain[e] : 0.0; kp_n = lv ? W.keep[e] : 0u;
    };
    issue(0);
    for (size_t base = 0; base < n_in; base += 64) {
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
