// find_keep_sub replay, lane-per-element formulation (see comp_kernels.hpp for the idea).
//
// One replay = k_fks_sweep + k_fks_scan.
//   k_fks_sweep: lane <-> element, 8 consecutive lanes <-> one 8-block of the reference's sweep
//     (compress_utils.cpp:159-250).  For every sweep p the group's start state comes from the
//     previous replay's exclusive prefixes (xk8, xg8) and the sweep scalars; inside the group the
//     reference's order (element 0 updates the running norm before element 1 is examined, ...) is
//     reproduced by speculate-and-validate: every flagged lane decides with the group's start
//     norm, the exact running norm of each lane is then rebuilt from its predecessors' changes in
//     order with shuffles, and lanes whose decision could differ under their exact norm redo it.
//     Group deltas (dk8, dg8, ws8) are updated in place; any change raises the replay's flag.
//   k_fks_scan: per sweep, exclusive prefixes of (dk8, dg8) over the 8-blocks and the totals; the
//     last workgroup to finish turns the totals into the sweep scalars of the next replay
//     (compress_utils.cpp:153-158, 251-265).
// After the replay has settled, k_fks_sweep in final mode recomputes every wt_remain with the
// budget of the last sweep that flagged the element (compress_utils.cpp:243-245), bit for bit.
#pragma once
#include "comp_kernels.hpp"

struct FksScal {
    double psG[FR_FKS_PMAX];        // norm at the start of sweep p
    uint32_t psN[FR_FKS_PMAX];      // sample budget at the start of sweep p
    int n_pass;
    uint32_t zero_prefix;           // 1: replay 0, every prefix is zero
    double G0, G_last;
    uint32_t n0, n_last;
    uint32_t changed;               // raised by k_fks_sweep, cleared by k_fks_scan
    uint32_t done_ctr;              // workgroups of k_fks_scan that have finished
    uint32_t n_in;
    uint32_t overflow;              // the reference would run more sweeps than FR_FKS_PMAX
    int valid_upto;                 // sweeps 0..valid_upto have prefixes / chunk totals from the previous replay
    uint32_t warm;                  // replay 0 starts from the previous iteration's sweep structure (FksSaved)
    double warm_scale;              // this stage's norm / the saved stage's norm
};

// What the settled replay of this stage looked like in the previous FRI iteration: the sweep scalars and, per sweep,
// how many samples / how much norm each chunk of 2048 blocks consumed.  The walk is statistically stationary and vector
// positions persist, so this is a close first guess; any guess converges to the same (unique) consistent assignment.
struct FksSaved {
    int valid, n_pass;
    uint32_t n0, nchunk;
    double G0;
    double psG[FR_FKS_PMAX];
    uint32_t psN[FR_FKS_PMAX];
};

struct Fks2Work {
    uint32_t nb8_cap;
    uint32_t *dk8; double *dg8, *ws8;       // [FR_FKS_PMAX][nb8_cap] group deltas (in place)
    uint32_t *xk8; double *xg8;             // exclusive prefixes over groups, same shape
    uint32_t *ck; double *cg, *cw;          // [FR_FKS_PMAX][FR_FKS_MAXCHUNK] totals per chunk of 2048 groups
    FksScal *scal;
    uint32_t *hist;                         // [FR_MAX_ROUNDS] changed flag per replay, for the host
    uint32_t *dbg_cnt;                      // [FR_MAX_ROUNDS][4] FRIES_DBG=3 statistics
    FksSaved *saved; uint32_t *wk; double *wg;      // this stage's warm-start record and saved chunk totals (same shape as ck / cg)
};
#define FR_FKS_CHUNK 2048                   // groups per scan workgroup
#define FR_FKS_MAXCHUNK 1024

// What a rank tells the others about its shard: the norm entering sweep 0 and, per sweep of the replay that just
// ran, how many samples it preserved, by how much its norm dropped and what its wt_remain re-sums to.
struct FksMsg {
    double L0;
    uint32_t changed, pad;
    uint32_t totK[FR_FKS_PMAX];
    double totG[FR_FKS_PMAX], totW[FR_FKS_PMAX];
};
static_assert(sizeof(FksMsg) <= 2048, "FksMsg must fit FRIES_COMM_SMALL_BYTES");
#define FR_MAX_RANKS 64

// Sweep bookkeeping of compress_utils.cpp:153-158, 251-265 for the next replay from every rank's totals:
// glob_one_norm = sum_mpi(loc_one_norm) and glob_sampled = sum_mpi(loc_sampled), added in rank order.
__device__ __forceinline__ void fr_fks2_passes(FksScal *S, const FksMsg *msgs, int n_ranks, uint32_t *hist_it) {
    double L[FR_MAX_RANKS];
    uint32_t ch = 0;
    for (int r = 0; r < n_ranks; r++) { L[r] = msgs[r].L0; ch |= msgs[r].changed; }
    if (hist_it) *hist_it = ch;
    uint32_t n = S->n0;
    int last_pass = 0, p = 0;
    for (; p < FR_FKS_PMAX; p++) {
        double G = 0;
        for (int r = 0; r < n_ranks; r++) G += L[r];
        S->psG[p] = G; S->psN[p] = n;
        if (G < 0) break;
        uint32_t K = 0;
        for (int r = 0; r < n_ranks; r++) K += msgs[r].totK[p];
        n -= K;
        uint32_t gs = K;
        if (last_pass && gs) last_pass = 0;
        if (gs == 0 && !last_pass) { last_pass = 1; gs = 1; for (int r = 0; r < n_ranks; r++) L[r] = msgs[r].totW[p]; }
        else for (int r = 0; r < n_ranks; r++) L[r] = L[r] - msgs[r].totG[p];
        if (gs == 0) { p++; break; }
    }
    S->n_pass = p;
    if (p >= FR_FKS_PMAX) S->overflow = 1;
    S->G_last = S->psG[p > 0 ? p - 1 : 0]; S->n_last = n;
}

// Replay 0 from the previous iteration's record instead of "nothing kept anywhere" (psG[0] is already this stage's norm)
__device__ __forceinline__ void fr_fks2_warm(FksScal *S, const FksSaved *Wv, int enable) {
    S->warm = 0; S->warm_scale = 1.0;
    if (!enable || !Wv->valid || Wv->n0 != S->n0 || !(Wv->G0 > 0) || !(S->psG[0] > 0) || Wv->n_pass < 1) return;
    const double sc = S->psG[0] / Wv->G0;
    for (int p = 0; p < Wv->n_pass; p++) { S->psG[p] = Wv->psG[p] * sc; S->psN[p] = Wv->psN[p]; }
    S->n_pass = Wv->n_pass;
    S->warm = 1; S->warm_scale = sc;
}

// sets up replay 0: "nothing kept anywhere" (the stage's input norm comes from the prep kernel's tile partials)
static __global__ void __launch_bounds__(FR_BLOCK) k_fks_init(CompWork W, Fks2Work F, FksMsg *msg, int inline_passes, int warm) {
    __shared__ double shd[4];
    const CompState st0 = W.state[0];
    const double G0 = fr_sum_partials(W.psum[0], (st0.n_in + FR_TILE - 1) / FR_TILE, shd);
    if (threadIdx.x == 0) {
        FksScal *S = F.scal;
        S->G0 = G0; S->n0 = st0.n_rem; S->n_in = st0.n_in;
        msg->L0 = G0; msg->changed = 0; msg->pad = 0;
        for (int p = 0; p < FR_FKS_PMAX; p++) { msg->totG[p] = 0; msg->totK[p] = 0; msg->totW[p] = G0; }
        S->zero_prefix = 1; S->changed = 0; S->done_ctr = 0; S->overflow = 0; S->valid_upto = -1;
        for (int k = 0; k < FR_MAX_ROUNDS + 2; k++) F.hist[k] = 0;
        S->warm = 0; S->warm_scale = 1.0;
        if (inline_passes) { fr_fks2_passes(S, msg, 1, nullptr); fr_fks2_warm(S, F.saved, warm); }
    }
}

// after the all-gather of every rank's FksMsg (n_ranks > 1)
static __global__ void k_fks_passes(Fks2Work F, const FksMsg *msgs, int n_ranks, int it, uint32_t *err, int warm) {
    FksScal *S = F.scal;
    fr_fks2_passes(S, msgs, n_ranks, it >= 0 ? &F.hist[it] : nullptr);
    if (it < 0) fr_fks2_warm(S, F.saved, warm);
    if (S->overflow) atomicOr(err, FR_ERR_ROUNDS);
}

// One sub-weight row against a threshold: keeps, remaining weight, and the bounds needed to tell whether the
// decision would differ under a slightly smaller threshold.
template <int STAGE, bool NEW_HB>
__device__ __forceinline__ void fr_fks2_row(const HbTables &T, det_t det, uint32_t code, const RowInfo &ri, unsigned n_sub, double p_doub,
                                            double cwf, double gl, uint32_t kp_in, uint32_t *kp_out, uint32_t *add, double *sub_remain, double *max_unkept) {
    unsigned full = (n_sub / 8) * 8;
    uint32_t kk = kp_in, a = 0;
    double rem = 0, mu = 0;
    fr_row_visit<STAGE, NEW_HB>(T, det, code, ri, p_doub, [&](unsigned s, double w) {
        if (s >= n_sub || ((kk >> s) & 1u)) return;
        double sub_magn = cwf * w;
        double thr = s < full ? 1e-12 : 1e-10;      // compress_utils.cpp:213 / :233
        if (sub_magn >= gl && fabs(sub_magn) > thr) { kk |= 1u << s; a++; }
        else { rem += sub_magn; mu = sub_magn > mu ? sub_magn : mu; }
    });
    *kp_out = kk; *add = a; *sub_remain = rem; *max_unkept = mu;
}

template <int STAGE, bool NEW_HB>
__global__ void __launch_bounds__(FR_BLOCK) k_fks_sweep(CompWork W, Fks2Work F, VecDev V, const HbTables *Tg, int cur, int it, double p_doub, int final, int dbg = 0) {
    __shared__ HbTables T;
    __shared__ FksScal S;
    {
        const uint32_t *src = (const uint32_t *)F.scal;
        uint32_t *dst = (uint32_t *)&S;
        for (unsigned i = threadIdx.x; i < sizeof(FksScal) / 4; i += blockDim.x) dst[i] = src[i];
    }
    if (STAGE != 1) fr_stage_tables(&T, Tg); else __syncthreads();
    // offsets of my chunk of groups: sums of the earlier chunks' totals, sweep by sweep (one wave per sweep, round robin)
    __shared__ uint32_t s_offK[FR_FKS_PMAX], s_inK[FR_FKS_PMAX];
    __shared__ double s_offG[FR_FKS_PMAX], s_inG[FR_FKS_PMAX];
    const bool warm0 = !final && S.zero_prefix && S.warm;      // replay 0 of a warm start: prefixes from the saved chunk totals
    const unsigned my_chunk = (unsigned)(((size_t)blockIdx.x * FR_BLOCK / 8) / FR_FKS_CHUNK);
    {
        const int ln = fr_lane(), wv_ = threadIdx.x >> 6;
        int p_hi = (S.zero_prefix || S.valid_upto < 0) ? -1 : (S.valid_upto < FR_FKS_PMAX - 1 ? S.valid_upto : FR_FKS_PMAX - 1);
        const uint32_t *srck = F.ck; const double *srcg = F.cg;
        unsigned c_hi = my_chunk;
        if (warm0) { p_hi = S.n_pass - 1; srck = F.wk; srcg = F.wg; const unsigned nc = F.saved->nchunk; if (c_hi > nc) c_hi = nc; }
        for (int p = wv_; p <= p_hi; p += 4) {
            uint32_t k = 0; double g = 0;
            {
                for (unsigned c = ln; c < c_hi; c += 64) { k += srck[(size_t)p * FR_FKS_MAXCHUNK + c]; g += srcg[(size_t)p * FR_FKS_MAXCHUNK + c]; }
            }
            k = fr_wave_sum_u32(k); g = fr_wave_sum(g);
            if (ln == 0) {
                s_offK[p] = k; s_offG[p] = warm0 ? g * S.warm_scale : g;
                if (warm0) {
                    const bool in = my_chunk < F.saved->nchunk;
                    s_inK[p] = in ? srck[(size_t)p * FR_FKS_MAXCHUNK + my_chunk] : 0u;
                    s_inG[p] = in ? srcg[(size_t)p * FR_FKS_MAXCHUNK + my_chunk] * S.warm_scale : 0.0;
                }
            }
        }
        __syncthreads();
    }
    const unsigned n_in = S.n_in;
    const size_t e = (size_t)blockIdx.x * FR_BLOCK + threadIdx.x;
    const unsigned nb8 = n_in / 8 + 1;
    const size_t b = e >> 3;
    const int f = (int)(e & 7), lane = fr_lane(), gbase = lane & ~7;
    const bool in_grp = b < nb8;                 // the reference also visits the (possibly empty) tail group
    const bool live = e < n_in;
    const size_t stride = F.nb8_cap;
    const StageElems E = W.el[cur];
    const int n_pass = S.n_pass;
    const bool zp = S.zero_prefix != 0;
    // my element
    double v = live ? E.val[e] : 0.0;
    uint32_t nd = live ? E.ndiv[e] : 1u;
    double wr = v;
    uint32_t kp = (final && live) ? W.keep[e] : 0u;
    // could any sweep ever flag it?  the threshold never drops below G_last / n0-ish; be generous
    det_t det = 0; uint32_t code = 0; RowInfo ri; ri.inv_norm = 1; ri.aux = 0; ri.nsub = 2; ri.tot = 0;
    bool have_row = false;
    auto fetch_row = [&]() {
        if (have_row) return;
        have_row = true;
        if (STAGE != 1) { code = E.code[e]; det = V.dets[E.pos[e]]; ri = fr_row_cached(E, e); }
    };
    if (final) {
        double lastwf = 0;
        for (int p = 0; p < n_pass; p++) {
            const bool pv = !zp && in_grp && p <= S.valid_upto;
            double xg = pv ? s_offG[p] + F.xg8[(size_t)p * stride + b] : 0.0;
            uint32_t xk = pv ? s_offK[p] + F.xk8[(size_t)p * stride + b] : 0u;
            double glob = S.psG[p] - xg, wf = (double)(S.psN[p] - xk);
            if (live && nd == 0 && v > 0 && v * wf >= glob) lastwf = wf;
        }
        if (live) {
            double out = v;
            if (nd > 0) { if (kp & 1u) out = 0; }
            else if (lastwf > 0) {
                fetch_row();
                unsigned n_sub = fr_row_len<STAGE, NEW_HB>(T, ri.nsub);
                const double cwf = v * lastwf;
                double rem = 0;
                fr_row_visit<STAGE, NEW_HB>(T, det, code, ri, p_doub, [&](unsigned s, double w) {
                    if (s >= n_sub || ((kp >> s) & 1u)) return;
                    rem += cwf * w;
                });
                out = rem / lastwf;
            }
            W.wt_remain[e] = out;
        }
        return;
    }
    if (dbg == 1) return;
    float wmax = -1.0f;         // upper bound of the largest unpreserved normalised weight; < 0: row not looked at yet
    uint32_t diff = 0;
    for (int p = 0; p < n_pass; p++) {
        const bool pv = !zp && in_grp && p <= S.valid_upto;
        double xg = pv ? s_offG[p] + F.xg8[(size_t)p * stride + b] : 0.0;
        uint32_t xk = pv ? s_offK[p] + F.xk8[(size_t)p * stride + b] : 0u;
        if (warm0) {        // chunk offsets of the previous iteration, linear inside the chunk
            const double fr = (double)(b - (size_t)my_chunk * FR_FKS_CHUNK) * (1.0 / FR_FKS_CHUNK);
            xg = s_offG[p] + s_inG[p] * fr;
            xk = s_offK[p] + (uint32_t)((double)s_inK[p] * fr);
            if (xk >= S.psN[p]) xk = S.psN[p] - 1;
        }
        const double glob0 = S.psG[p] - xg, wf = (double)(S.psN[p] - xk);
        // flags are taken against the group's start norm (compress_utils.cpp:172-180)
        double cw = v * wf;
        if (nd > 0) cw /= nd;
        const bool flagged = live && wr > 0 && cw >= glob0;
        // --- speculate with the start norm, then validate against the exact running norm
        double change = 0, new_wr = wr, mu = 0;
        uint32_t add = 0, new_kp = kp;
        bool evaluated = false, skipped = false;
        double used_gl = glob0;
        if (flagged) {
            if (nd > 0) { new_kp = kp | 1u; new_wr = 0; add = nd; change = v; }
            else if (wmax >= 0 && cw * (double)wmax < glob0) skipped = true;
        }
        bool need_eval = flagged && nd == 0 && !skipped && dbg != 2;
        double gl_mine = glob0;
        for (int round = 0; round < 9; round++) {
            if (dbg == 3) {
                if (need_eval) atomicAdd(&F.hist[FR_MAX_ROUNDS], 1u);
                if (__any(need_eval) && lane == 0) atomicAdd(&F.hist[FR_MAX_ROUNDS + 1], 1u);
            }
            if (need_eval) {
                fetch_row();
                unsigned n_sub = fr_row_len<STAGE, NEW_HB>(T, ri.nsub);
                double rem;
                fr_fks2_row<STAGE, NEW_HB>(T, det, code, ri, n_sub, p_doub, cw, gl_mine, kp, &new_kp, &add, &rem, &mu);
                new_wr = rem / wf;
                change = wr - new_wr;
                evaluated = true; used_gl = gl_mine; skipped = false;
            }
            // exact running norm of each lane: start norm minus the changes of the flagged lanes before it, in order
            double g = glob0;
            for (int j = 0; j < 7; j++) {
                double cj = __shfl(change, gbase + j);
                int fj = __shfl((int)(flagged && !skipped), gbase + j);
                if (j < f && fj) g -= cj;
            }
            gl_mine = g;
            // would my decision differ under gl_mine?
            need_eval = false;
            if (flagged && nd == 0) {
                if (skipped) { if (cw * (double)wmax >= gl_mine) need_eval = true; }
                else if (evaluated && used_gl != gl_mine && mu >= gl_mine) need_eval = true;
            }
            if (!__any(need_eval)) break;
        }
        // commit
        if (flagged && !skipped) {
            if (evaluated) {
                // largest unpreserved normalised weight, rounded up
                wmax = (cw > 0) ? (float)((mu / cw) * 1.000002) : 0.0f;
            }
            kp = new_kp; wr = new_wr;
        }
        else { add = 0; change = 0; }
        // group totals in element order
        uint32_t gk = 0; double gg = 0, gw = 0;
        for (int j = 0; j < 8; j++) {
            gk += (uint32_t)__shfl((int)add, gbase + j);
            gg += __shfl(change, gbase + j);
            gw += __shfl(live ? wr : 0.0, gbase + j);
        }
        if (f == 0 && in_grp) {
            size_t ix = (size_t)p * stride + b;
            if (zp || p > S.valid_upto || F.dk8[ix] != gk || __double_as_longlong(F.dg8[ix]) != __double_as_longlong(gg) || __double_as_longlong(F.ws8[ix]) != __double_as_longlong(gw)) diff = 1;
            if (dbg == 3 && !zp && p <= S.valid_upto) {
                if (F.dk8[ix] != gk) atomicAdd(&F.dbg_cnt[it * 4 + 0], 1u);
                else if (__double_as_longlong(F.dg8[ix]) != __double_as_longlong(gg)) {
                    atomicAdd(&F.dbg_cnt[it * 4 + 1], 1u);
                    if (fabs(F.dg8[ix] - gg) > 1e-9 * fabs(gg)) atomicAdd(&F.dbg_cnt[it * 4 + 3], 1u);
                }
                else if (__double_as_longlong(F.ws8[ix]) != __double_as_longlong(gw)) atomicAdd(&F.dbg_cnt[it * 4 + 2], 1u);
            }
            F.dk8[ix] = gk; F.dg8[ix] = gg; F.ws8[ix] = gw;
        }
    }
    // one sweep beyond: keeps nothing, but its wt_remain sum is what a re-summed norm would be
    if (n_pass < FR_FKS_PMAX) {
        double gw = 0;
        for (int j = 0; j < 8; j++) gw += __shfl(live ? wr : 0.0, gbase + j);
        if (f == 0 && in_grp) {
            size_t ix = (size_t)n_pass * stride + b;
            if (zp || n_pass > S.valid_upto || F.dk8[ix] != 0 || __double_as_longlong(F.ws8[ix]) != __double_as_longlong(gw)) diff = 1;
            F.dk8[ix] = 0; F.dg8[ix] = 0; F.ws8[ix] = gw;
        }
    }
    if (live) { W.keep[e] = kp; W.wt_remain[e] = wr; }
    __shared__ uint32_t s_any;
    if (threadIdx.x == 0) s_any = 0;
    __syncthreads();
    if (__any(diff) && lane == 0) s_any = 1;
    __syncthreads();
    if (threadIdx.x == 0 && s_any && F.hist[it] == 0) atomicOr(&F.hist[it], 1u);
}

// Exclusive prefixes over the 8-blocks inside chunks of 2048 groups (grid: chunks x sweeps); chunk totals go to
// (ck, cg, cw).  The last workgroup to finish sums them into the sweep totals and derives the sweep scalars of the
// next replay (compress_utils.cpp:153-158, 251-265).
static __global__ void __launch_bounds__(FR_BLOCK) k_fks_scan(Fks2Work F, int it) {
    __shared__ double shd[12];
    __shared__ uint32_t shu[4];
    FksScal *S = F.scal;
    const int p = blockIdx.y;
    const unsigned c = blockIdx.x;
    const unsigned nb8 = S->n_in / 8 + 1;
    const unsigned nchunk = (nb8 + FR_FKS_CHUNK - 1) / FR_FKS_CHUNK;
    const int n_pass = S->n_pass;
    const size_t stride = F.nb8_cap;
    if (c < nchunk && p <= n_pass && p < FR_FKS_PMAX) {
        const size_t base = (size_t)p * stride + (size_t)c * FR_FKS_CHUNK + (size_t)threadIdx.x * 8;
        const size_t lim = (size_t)p * stride + nb8;
        uint32_t k[8]; double g[8];
        uint32_t tk = 0; double tg = 0, tw = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            bool ok = base + j < lim;
            k[j] = ok ? F.dk8[base + j] : 0u; g[j] = ok ? F.dg8[base + j] : 0.0;
            tk += k[j]; tg += g[j]; tw += ok ? F.ws8[base + j] : 0.0;
        }
        uint32_t totk;
        uint32_t ik = fr_block_scan_u32(tk, shu, &totk);
        double totg, totw;
        double eg = fr_block_excl_f64(tg, shd, &totg);
        fr_block_excl_f64(tw, shd, &totw);
        uint32_t ek = ik - tk;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (base + j < lim) { F.xk8[base + j] = ek; F.xg8[base + j] = eg; }
            ek += k[j]; eg += g[j];
        }
        if (threadIdx.x == 0) {
            F.ck[(size_t)p * FR_FKS_MAXCHUNK + c] = totk; F.cg[(size_t)p * FR_FKS_MAXCHUNK + c] = totg; F.cw[(size_t)p * FR_FKS_MAXCHUNK + c] = totw;
        }
    }
}

// Sweep totals from the chunk totals into this rank's FksMsg; with one rank, also the sweep scalars of the next replay
static __global__ void __launch_bounds__(FR_BLOCK) k_fks_totals(Fks2Work F, uint32_t *err, FksMsg *msg, int inline_passes, int it) {
    __shared__ double shd[4];
    __shared__ uint32_t shu[4];
    FksScal *S = F.scal;
    const unsigned nb8 = S->n_in / 8 + 1;
    const unsigned nchunk = (nb8 + FR_FKS_CHUNK - 1) / FR_FKS_CHUNK;
    const int n_pass = S->n_pass;
    for (int q = 0; q < FR_FKS_PMAX; q++) {
        uint32_t k = 0; double g = 0, w = 0;
        if (q > n_pass + 1) { if (threadIdx.x == 0) { msg->totK[q] = 0; msg->totG[q] = 0; msg->totW[q] = 0; } continue; }
        if (q <= n_pass) {
            k = fr_sum_partials_u32(F.ck + (size_t)q * FR_FKS_MAXCHUNK, nchunk, shu);
            g = fr_sum_partials(F.cg + (size_t)q * FR_FKS_MAXCHUNK, nchunk, shd);
            w = fr_sum_partials(F.cw + (size_t)q * FR_FKS_MAXCHUNK, nchunk, shd);
        }
        if (threadIdx.x == 0) { msg->totK[q] = k; msg->totG[q] = g; msg->totW[q] = w; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        S->zero_prefix = 0;
        S->valid_upto = n_pass;         // k_fks_scan covered sweeps 0..n_pass of the replay that just ran
        msg->L0 = S->G0; msg->changed = F.hist[it]; msg->pad = 0;
        if (inline_passes) {
            fr_fks2_passes(S, msg, 1, nullptr);
            if (S->overflow) atomicOr(err, FR_ERR_ROUNDS);
        }
    }
}

// records the settled replay for the next iteration's warm start (one workgroup)
static __global__ void __launch_bounds__(FR_BLOCK) k_fks_save(Fks2Work F) {
    const FksScal *S = F.scal;
    const unsigned nb8 = S->n_in / 8 + 1;
    const unsigned nchunk = (nb8 + FR_FKS_CHUNK - 1) / FR_FKS_CHUNK;
    const int n_pass = S->n_pass < FR_FKS_PMAX ? S->n_pass : FR_FKS_PMAX;
    for (int p = 0; p < n_pass; p++)
        for (unsigned c = threadIdx.x; c < nchunk; c += blockDim.x) {
            F.wk[(size_t)p * FR_FKS_MAXCHUNK + c] = F.ck[(size_t)p * FR_FKS_MAXCHUNK + c];
            F.wg[(size_t)p * FR_FKS_MAXCHUNK + c] = F.cg[(size_t)p * FR_FKS_MAXCHUNK + c];
        }
    if (threadIdx.x == 0) {
        FksSaved *V = F.saved;
        V->n_pass = n_pass; V->n0 = S->n0; V->nchunk = nchunk; V->G0 = S->psG[0];
        for (int p = 0; p < n_pass; p++) { V->psG[p] = S->psG[p]; V->psN[p] = S->psN[p]; }
        V->valid = (S->overflow || !(S->psG[0] > 0)) ? 0 : 1;
    }
}
