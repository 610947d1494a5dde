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
// Steady-state stages (remaining norm ~ 0.3 of the start) never come here.  The one-wave walk of this first half (~80 ms per sweep and million
// elements) is the fallback of the parallel form in the second half of this file.
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
//   3. k_fsq_maps<0> / k_fsq_chain / k_fsq_maps<1>   the reference's subtractions on the changes, exactly: tile by tile as integer arithmetic while
//                   the norm stays inside a binade, element by element (one wave, one dependent v_add_f64 per element) where it does not -- see
//                   "the chain" below; leaves the exact entry norm of every block;
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
    if (threadIdx.x == 0) { SQ.tk[blockIdx.x] = 0u; SQ.tg[blockIdx.x] = 0.0; SQ.tany[blockIdx.x] = 0; SQ.tgx[blockIdx.x] = Q->G; SQ.mflag[blockIdx.x] = 0; SQ.fast[blockIdx.x] = 0; }
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

// ---- the chain.  The reference's running norms (compress_utils.cpp:190-192, 246-247: glob_one_norm -= change, loc_one_norm -= change per touched
// element, in storage order) are one dependent subtraction per element: 8.5 cycles each on one lane, 3.5 ms per million, however it is fed.  But while the
// norm X stays inside one binade [2^e, 2^(e+1)) every value it takes is an integer multiple of u = 2^(e-52), and fl(X - d) = X - D u with
// D = d / u rounded to the nearest integer (the remainder decides; an exact half is a tie, settled by the parity of X / u -- rare, and left to the walk).
// So for a tile that is entered and left in the same binade the chain is X <- X - (sum of D) u: the D of its 256 elements and their sum are formed by all
// lanes of a workgroup BEFORE the chain (k_fsq_maps<0>, with the binade read off the tree-summed norm the guesses already use), the one-wave chain
// (k_fsq_chain) checks with the EXACT norm that the assumption holds -- same exponent, and the sum of |change| plus rounding slop keeps every intermediate
// value inside the binade -- and takes the tile in one multiply-subtract; k_fsq_maps<1> then expands the entries of the tile's 32 blocks from the exact
// tile entry.  A tile that fails the check -- the norm crosses a power of two inside it, a tie, a norm gone to zero or below (the last tiles of a
// collapsing sweep) -- is walked element by element as before.  Nothing rests on the approximation: it only proposes the binade.
__device__ __forceinline__ double fr_pow2(int n) { return __longlong_as_double((long long)(n + 1023) << 52); }      // -1022 <= n <= 1023
__device__ __forceinline__ int fr_expo(double x) { return (int)((__double_as_longlong(x) >> 52) & 0x7ff) - 1023; }  // x > 0, normal
// d in units of u (inv_u = 1 / u, a power of two): false if the remainder is an exact half or the quotient is out of range
__device__ __forceinline__ bool fr_fsq_units(double d, double inv_u, double *D) {
    const double t = d * inv_u;
    if (!(fabs(t) < 9007199254740992.0)) return false;
    const double q = floor(t), r = t - q;
    if (r == 0.5) return false;
    *D = q + (r > 0.5 ? 1.0 : 0.0);
    return true;
}
__device__ __forceinline__ bool fr_fsq_expo_ok(double x, int *e) {
    if (!(x > 0.0) || !(x < INFINITY)) return false;
    const int k = fr_expo(x);
    if (k < -900 || k > 900) return false;
    *e = k;
    return true;
}
// every value X - (partial sums of the changes) of a tile entered with X stays in [2^e, 2^(e+1)), rounding included
__device__ __forceinline__ bool fr_fsq_clean(double X, int e, double sabs) {
    if (!(X > 0.0) || !(X < INFINITY) || fr_expo(X) != e) return false;
    const double p = fr_pow2(e), u = fr_pow2(e - 52);
    const double B = sabs * (1.0 + 1e-9) + 600.0 * u;          // 256 roundings of at most u / 2 each, the tree sum's own error, slack
    return X - B >= p && X + B < 2.0 * p;
}

// MODE 0: the tiles' integer decrements under the binade their approximate entry norm lies in.  MODE 1 (after the chain): entries of the blocks of the
// tiles the chain took in one step, and of the tiles the sweep does not touch
template <int MODE>
__global__ void __launch_bounds__(FR_BLOCK) k_fsq_maps(const FksSeq *Q, FksSq SQ, uint32_t from_tile) {
    __shared__ double sh_a[4], sh_b[4], sh_c[4]; __shared__ uint32_t sh_bad[4];
    __shared__ double sh_pg[32], sh_pl[32];
    if (!Q->go) return;
    const unsigned tile = from_tile + blockIdx.x;
    const size_t e = (size_t)tile * FR_SQ_TILE + threadIdx.x;
    const int lane = fr_lane(), f = lane & 7, wv = threadIdx.x >> 6;
    const bool any = SQ.tany[tile] != 0;
    if (MODE == 1) {
        const bool fast = SQ.fast[tile] != 0;
        if (any && !fast) return;                       // walked by the chain, which left its entries
        if (!any) { if (threadIdx.x < 32) { SQ.gb[(size_t)tile * 32 + threadIdx.x] = SQ.gt[tile]; SQ.lb[(size_t)tile * 32 + threadIdx.x] = SQ.lt[tile]; } return; }
    }
    else if (!any) { if (threadIdx.x == 0) SQ.mflag[tile] = 0; return; }
    const double d = SQ.dl[e];
    int eg = 0, el = 0;
    bool ok;
    if (MODE == 0) {
        const double ga = SQ.tgx[tile], la = ga - (Q->G - Q->L);
        ok = fr_fsq_expo_ok(ga, &eg) && fr_fsq_expo_ok(la, &el);
    }
    else { eg = SQ.eG[tile]; el = SQ.eL[tile]; ok = true; }
    double DG = 0.0, DL = 0.0;
    bool mine = ok && fr_fsq_units(d, fr_pow2(52 - eg), &DG) && fr_fsq_units(d, fr_pow2(52 - el), &DL);
    if (MODE == 0) {
        double a = DG, bsum = DL, c = fabs(d);
        for (int o = 32; o >= 1; o >>= 1) { a += __shfl_down(a, o); bsum += __shfl_down(bsum, o); c += __shfl_down(c, o); }       // integers below 2^53 (or the tile is refused): exact in any order
        const unsigned long long bad = __ballot(!mine);
        if (lane == 0) { sh_a[wv] = a; sh_b[wv] = bsum; sh_c[wv] = c; sh_bad[wv] = bad ? 1u : 0u; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const double mg = (sh_a[0] + sh_a[1]) + (sh_a[2] + sh_a[3]), ml = (sh_b[0] + sh_b[1]) + (sh_b[2] + sh_b[3]);
            const bool good = !(sh_bad[0] | sh_bad[1] | sh_bad[2] | sh_bad[3]) && mg < 9007199254740992.0 && ml < 9007199254740992.0 && mg > -4503599627370496.0 && ml > -4503599627370496.0;
            SQ.mG[tile] = mg; SQ.mL[tile] = ml; SQ.sabs[tile] = (sh_c[0] + sh_c[1]) + (sh_c[2] + sh_c[3]); SQ.eG[tile] = eg; SQ.eL[tile] = el; SQ.mflag[tile] = good ? 1 : 0;
        }
    }
    else {
        // (mine holds for every lane: the same arithmetic on the same numbers as in MODE 0, which accepted the tile)
        const double bg = fr_grp8_sum(DG), bl = fr_grp8_sum(DL);
        if (f == 0) { sh_pg[threadIdx.x >> 3] = bg; sh_pl[threadIdx.x >> 3] = bl; }
        __syncthreads();
        if (threadIdx.x < 32) {
            double pg = 0.0, pl = 0.0;
            for (unsigned q = 0; q < threadIdx.x; q++) { pg += sh_pg[q]; pl += sh_pl[q]; }
            SQ.gb[(size_t)tile * 32 + threadIdx.x] = SQ.gt[tile] - pg * fr_pow2(eg - 52);
            SQ.lb[(size_t)tile * 32 + threadIdx.x] = SQ.lt[tile] - pl * fr_pow2(el - 52);
        }
    }
}

// one wave: the tiles in order.  use_maps = 0: every touched tile element by element (the fallback's and the check's form)
static __global__ void __launch_bounds__(64) k_fsq_chain(const FksSeq *Q, FksSq SQ, double *gb, double *lb, uint32_t from_tile, uint32_t n_tiles, int sparse_max, int use_maps) {
    __shared__ double2 sh[FR_SQ_TILE / 2];
    if (!Q->go) return;
    const int lane = fr_lane(), j = lane & 31;
    const bool is_loc = lane >= 32;
    double XG, XL;                                       // the same value in every lane
    if (from_tile == 0) { XG = Q->G; XL = Q->L; } else { XG = SQ.gt[from_tile]; XL = SQ.lt[from_tile]; }
    double *const out = is_loc ? lb : gb;
    unsigned long long n_fast = 0, n_dense = 0;
    for (uint32_t base = from_tile; base < n_tiles; base += 64) {
        const uint32_t tl = base + lane;
        const bool on = tl < n_tiles;
        const uint32_t a_any = on ? SQ.tany[tl] : 0u, a_flag = (on && use_maps) ? SQ.mflag[tl] : 0u;
        const double a_mG = (on && use_maps) ? SQ.mG[tl] : 0.0, a_mL = (on && use_maps) ? SQ.mL[tl] : 0.0, a_sabs = (on && use_maps) ? SQ.sabs[tl] : 0.0;
        const int a_eG = (on && use_maps) ? SQ.eG[tl] : 0, a_eL = (on && use_maps) ? SQ.eL[tl] : 0;
        double capG = 0.0, capL = 0.0; uint32_t a_fast = 0u;
        const int cnt = n_tiles - base < 64u ? (int)(n_tiles - base) : 64;
        // All 64 tiles at once when the touched ones share one binade per chain (most batches: the norm halves at 1/2, 3/4, 7/8 ... of the way):
        // the entry of tile i is X - (decrements of the tiles before it) u, integers again, so a wave scan gives every lane its tile's entry and
        // every lane makes its own check.  One tile that fails sends the batch through the tile-by-tile loop below.
        {
            const unsigned long long am = __ballot(a_any != 0u);
            if (am) {
                const int first = __ffsll((long long)am) - 1;
                const int eg0 = __builtin_amdgcn_readlane(a_eG, first), el0 = __builtin_amdgcn_readlane(a_eL, first);
                if (!__ballot(a_any != 0u && (a_flag == 0u || a_eG != eg0 || a_eL != el0))) {
                    const double ug = fr_pow2(eg0 - 52), ul = fr_pow2(el0 - 52);
                    double pg = a_any ? a_mG : 0.0, pl = a_any ? a_mL : 0.0;
                    for (int o = 1; o < 64; o <<= 1) { const double x = __shfl_up(pg, o), y = __shfl_up(pl, o); if (lane >= o) { pg += x; pl += y; } }
                    const double myG = XG - (pg - (a_any ? a_mG : 0.0)) * ug, myL = XL - (pl - (a_any ? a_mL : 0.0)) * ul;
                    const bool bad = a_any != 0u && !(fr_fsq_clean(myG, eg0, a_sabs) && fr_fsq_clean(myL, el0, a_sabs));
                    // (a prefix that is not exact -- sums beyond 2^53 -- gives a lane a wrong entry, but then a tile before it fails its own check)
                    if (!__ballot(bad)) {
                        if (on) { SQ.gt[tl] = myG; SQ.lt[tl] = myL; SQ.fast[tl] = a_any ? 1 : 0; }
                        XG -= fr_bcast_f64(pg, 63) * ug; XL -= fr_bcast_f64(pl, 63) * ul;
                        n_fast += (unsigned long long)__popcll(am);
                        continue;
                    }
                }
            }
        }
        for (int i = 0; i < cnt; i++) {
            if (lane == i) { capG = XG; capL = XL; }
            if (!__builtin_amdgcn_readlane((int)a_any, i)) continue;
            if (__builtin_amdgcn_readlane((int)a_flag, i)) {
                const int eg = __builtin_amdgcn_readlane(a_eG, i), el = __builtin_amdgcn_readlane(a_eL, i);
                const double sab = fr_bcast_f64(a_sabs, i);
                if (fr_fsq_clean(XG, eg, sab) && fr_fsq_clean(XL, el, sab)) {
                    XG -= fr_bcast_f64(a_mG, i) * fr_pow2(eg - 52);
                    XL -= fr_bcast_f64(a_mL, i) * fr_pow2(el - 52);
                    if (lane == i) a_fast = 1u;
                    n_fast++;
                    continue;
                }
            }
            // element by element: lanes 0-31 carry the global norm, lanes 32-63 the local one, so one dependent v_add_f64 per element advances
            // both; the changes reach every lane as LDS broadcast reads (16 bytes = two elements each).  Lane j (and 32 + j) keeps the value block j
            // of the tile is entered with.
            n_dense++;
            const uint32_t t = base + (uint32_t)i;
            const double2 *p = (const double2 *)(SQ.dl + (size_t)t * FR_SQ_TILE);
            const double2 c0 = p[lane], c1 = p[64 + lane];
            double X = is_loc ? XL : XG;
            double C = X;
            const unsigned long long m0 = __ballot(c0.x != 0.0 || c0.y != 0.0), m1 = __ballot(c1.x != 0.0 || c1.y != 0.0);
            if (__popcll(m0) + __popcll(m1) <= sparse_max) {
                // few touched elements: only the pairs that hold one, in order; lane i of c0 holds elements 2i, 2i + 1 of the tile (block i / 4),
                // lane i of c1 elements 128 + 2i, 129 + 2i (block 16 + i / 4)
                for (int h = 0; h < 2; h++) {
                    unsigned long long m = h ? m1 : m0;
                    const double2 src = h ? c1 : c0;
                    while (m) {
                        const int k = __ffsll((long long)m) - 1;
                        m &= m - 1;
                        X -= fr_bcast_f64(src.x, k); X -= fr_bcast_f64(src.y, k);
                        if (j > (k >> 2) + 16 * h) C = X;          // the blocks behind this pair are entered with the new value
                    }
                }
            }
            else {
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
            XG = fr_bcast_f64(X, 0); XL = fr_bcast_f64(X, 32);
        }
        if (on) { SQ.gt[tl] = capG; SQ.lt[tl] = capL; SQ.fast[tl] = (uint8_t)a_fast; }
    }
    if (lane == 0) { SQ.ctl->G_end = XG; SQ.ctl->L_end = XL; atomicAdd(&SQ.ctl->n_fast, n_fast); atomicAdd(&SQ.ctl->n_dense, n_dense); }
}
// FRIES_FSQ_CHECK: the entries the integer form produced (gb, lb) against those of the element-by-element chain (gb2, lb2), bit for bit
static __global__ void __launch_bounds__(FR_BLOCK) k_fsq_compare(FksSq SQ, uint32_t from_tile, uint32_t n_tiles) {
    const size_t b = (size_t)from_tile * 32 + (size_t)blockIdx.x * FR_BLOCK + threadIdx.x;
    if (b >= (size_t)n_tiles * 32) return;
    if (__double_as_longlong(SQ.gb[b]) != __double_as_longlong(SQ.gb2[b]) || __double_as_longlong(SQ.lb[b]) != __double_as_longlong(SQ.lb2[b])) atomicAdd(&SQ.ctl->n_mismatch, 1u);
}

// stores the sweep: elements of the tiles before upto_tile; fin: the whole sweep was settled here (else the walk finishes it and leaves the scalars)
static __global__ void __launch_bounds__(FR_BLOCK) k_fsq_commit(CompWork W, FksSeq *Q, FksSq SQ, uint32_t upto_tile, int fin) {
    if (!Q->go) return;
    const unsigned n_in = W.state[0].n_in;
    const size_t e = (size_t)blockIdx.x * FR_SQ_TILE + threadIdx.x;
    if (blockIdx.x < upto_tile && e < n_in && SQ.tany[blockIdx.x]) { W.keep[e] = SQ.nkp[e]; W.wt_remain[e] = SQ.nwr[e]; }
    if (fin && e == 0) { Q->G = SQ.ctl->G_end; Q->L = SQ.ctl->L_end; Q->K = SQ.ctl->K_tot; }
}
