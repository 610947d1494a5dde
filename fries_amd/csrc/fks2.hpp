// find_keep_sub replay, lane-per-element formulation (see comp_kernels.hpp for the idea).
//
// One replay = k_fks_sweep + k_fks_scan + k_fks_totals.
//   k_fks_sweep: lane <-> element, 8 consecutive lanes <-> one 8-block of the reference's sweep
//     (compress_utils.cpp:159-250).  For every sweep p the group's start state comes from the
//     previous replay's exclusive prefixes (chunk offset + xk8 / xg8) and the sweep scalars; inside
//     the group the reference's order (element 0 updates the running norm before element 1 is
//     examined, ...) is reproduced by speculate-and-validate: every flagged lane decides with the
//     group's start norm, the running norm of each lane is then rebuilt from its predecessors'
//     changes in order (DPP row shifts), and lanes whose decision could differ under it redo it.
//     Workgroups are persistent (the 17.5 KB table block is staged into LDS once per workgroup) and
//     the prefixes of sweep p + 1 are fetched while sweep p is evaluated: the kernel is bound by
//     dependent global-load latency at 2 waves/SIMD, not by bandwidth.
//   k_fks_scan: per (sweep, chunk of 2048 groups) exclusive prefixes of the deltas and the chunk
//     totals; compares the deltas with the previous replay's (double-buffered) and raises the
//     replay's "changed" flag.
//   k_fks_totals: exclusive prefixes over the chunks, sweep totals, and (one rank) the sweep scalars
//     of the next replay (compress_utils.cpp:153-158, 251-265).
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
    double G_neg;                   // the negative norm that ended the sweeps (compress_utils.cpp:155-157), else +inf
};

// What the settled replay of this stage looked like in the previous FRI iteration: the sweep scalars and, per sweep,
// how many samples / how much norm each chunk of 2048 blocks consumed.  The walk is statistically stationary and vector
// positions persist, so this is a close first guess; any guess converges to the same (unique) consistent assignment.
struct FksSaved {
    int valid, n_pass;
    uint32_t n0, nchunk, nb8;
    double G0;
    double psG[FR_FKS_PMAX];
    uint32_t psN[FR_FKS_PMAX];
    // the settled replay before that one (same stage, one iteration earlier still), for the trend: right after a restart and while the shift
    // moves, the budget left after the late sweeps drifts by 2-3 % per iteration in one direction
    int valid2, n_pass2; uint32_t n02; double G02;
    double psG2[FR_FKS_PMAX];
    uint32_t psN2[FR_FKS_PMAX];
    float extrap;                   // FRIES_FKS_WARM_EXTRAP: fraction of the last change added to the guess (0: last iteration's values as they are)
};

// What the host looks at after a batch of replays, in host-coherent pinned memory the kernels write straight into (a device-to-host
// copy of a few bytes costs a blit kernel plus a staging copy per readback, ~27 of them per iteration before this)
struct FksHost {
    uint32_t hist[FR_MAX_ROUNDS];   // per replay: did anything change
    uint32_t overflow, pad;
    double G_last, psG0, G_neg;
    int n_pass;
};

// what a wave of 64 elements recorded about one sweep of the recording replay: the state it entered the sweep with, its tightest comparison, the
// smallest norm any of its comparisons saw, and by how much its own deltas moved afterwards
struct alignas(32) FksWRec { double G; uint32_t K; float R; float M; uint32_t dK; float dG; uint32_t pad; };
struct Fks2Work {
    FksHost *hm;                            // device-visible address of the host block
    int hm_close;                           // 1 (one rank): the closing pass raises hm->hist[it] itself -- no k_fks_close_flag launch behind it
    uint32_t nb8_cap;
    uint32_t *dk8; double *dg8, *ws8;       // [FR_FKS_PMAX][nb8_cap] group deltas of the latest evaluation of every group
    // Per (wave of 64 elements = 8 groups, sweep): the start state the wave was last evaluated with -- running norm and remaining budget at
    // its first element -- the smallest relative distance of any of its comparisons from flipping and the smallest norm one of them
    // compared against.  A later replay whose start state for the wave moved the comparisons' threshold by less cannot change any
    // decision of the wave and skips it ("light" replays).  wdK / wdG: by how much that evaluation moved the wave's own deltas (samples,
    // norm; summed over its groups) -- the prefixes INSIDE the wave have moved by at most that much since its groups were evaluated.
    // [FR_FKS_PMAX][nwv_cap]; wNp[nwv_cap] = sweeps the wave last ran.
    FksWRec *wrec;                          // [wave][sweep]: a wave's records of one replay lie side by side (the light test reads them as one or two cache lines)
    uint32_t *wNp; uint32_t nwv_cap;
    uint32_t *cdirty;                       // [FR_FKS_MAXCHUNK] it + 1 of the last replay that changed a delta inside the chunk
    uint32_t *xk8; double *xg8;             // exclusive prefixes over the groups of a chunk, same shape
    uint32_t *ck; double *cg, *cw;          // [FR_FKS_PMAX][FR_FKS_MAXCHUNK] totals per chunk of 2048 groups
    uint32_t *ckx; double *cgx;             // exclusive prefixes of (ck, cg) over the chunks
    FksScal *scal;
    uint32_t *hist;                         // [FR_MAX_ROUNDS] changed flag per replay, for the host
    uint32_t *dbg_cnt;                      // [FR_MAX_ROUNDS][4] FRIES_DBG=3 statistics
    FksSaved *saved; uint32_t *wk, *wkx; double *wg, *wgx;     // this stage's warm-start record: saved ck / cg / ckx / cgx
    uint32_t *sxk8; double *sxg8;           // [FR_FKS_SROWS][nb8_cap] the settled per-group prefixes inside the chunks, kept for the next iteration's first replay
};
#define FR_FKS_CHUNK 2048                   // groups per scan workgroup
#define FR_FKS_MAXCHUNK 1024
#define FR_FKS_PF 8                        // sweeps whose stored deltas a wave of k_fks_sweep<.., 1> prefetches into LDS
#define FR_FKS_WARM_EXTRAP 0.0f             // default trend factor of the first replay's sweep scalars (FksSaved::extrap)
#define FR_FKS_SROWS 16                    // sweeps of a stage whose settled per-group prefixes are kept for the next iteration's first replay

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
__device__ __forceinline__ void fr_fks2_passes(FksScal *S, const FksMsg *msgs, int n_ranks, uint32_t *hist_it, FksHost *hm = nullptr, int it = -1) {
    double L[FR_MAX_RANKS];
    uint32_t ch = 0;
    for (int r = 0; r < n_ranks; r++) { L[r] = msgs[r].L0; ch |= msgs[r].changed; }
    if (hist_it) *hist_it = ch;
    if (hm && it >= 0 && it < FR_MAX_ROUNDS) hm->hist[it] = ch;
    uint32_t n = S->n0;
    int last_pass = 0, p = 0;
    S->G_neg = INFINITY;
    for (; p < FR_FKS_PMAX; p++) {
        double G = 0;
        for (int r = 0; r < n_ranks; r++) G += L[r];
        S->psG[p] = G; S->psN[p] = n;
        if (G < 0) { S->G_neg = G; break; }
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
    if (hm) { hm->overflow = S->overflow; hm->G_last = S->G_last; hm->psG0 = S->psG[0]; hm->G_neg = S->G_neg; hm->n_pass = S->n_pass; }
}

// Replay 0 from the previous iteration's record instead of "nothing kept anywhere" (psG[0] is already this stage's norm)
__device__ __forceinline__ void fr_fks2_warm(FksScal *S, const FksSaved *Wv, int enable) {
    S->warm = 0; S->warm_scale = 1.0;
    if (!enable || !Wv->valid || Wv->n0 != S->n0 || !(Wv->G0 > 0) || !(S->psG[0] > 0) || Wv->n_pass < 1) return;
    const double sc = S->psG[0] / Wv->G0;
    const bool trend = Wv->extrap > 0.0f && Wv->valid2 && Wv->n_pass2 == Wv->n_pass && Wv->n02 == Wv->n0 && Wv->G02 > 0;
    for (int p = 0; p < Wv->n_pass; p++) {
        double g = Wv->psG[p] / Wv->G0, k = (double)Wv->psN[p];
        if (trend && p > 0) {
            const double g2 = Wv->psG2[p] / Wv->G02, k2 = (double)Wv->psN2[p];
            const double gn = g + (double)Wv->extrap * (g - g2), kn = k + (double)Wv->extrap * (k - k2);
            if (gn > 0 && kn >= 1.0 && kn <= (double)Wv->n0) { g = gn; k = kn; }
        }
        S->psG[p] = g * S->psG[0]; S->psN[p] = (uint32_t)(k + 0.5);
    }
    (void)sc;
    S->n_pass = Wv->n_pass;
    S->warm = 1; S->warm_scale = sc;
}

// sets up replay 0: "nothing kept anywhere" (the stage's input norm comes from the prep kernel's tile partials)
static __global__ void __launch_bounds__(FR_BLOCK) k_fks_init(CompWork W, Fks2Work F, FksMsg *msg, int inline_passes, int warm) {
    __shared__ double shd[4];
    const CompState st0 = W.state[0];
    const double G0 = fr_sum_partials(W.psum[0], (st0.n_in + FR_TILE - 1) / FR_TILE, shd);
    for (int k = threadIdx.x; k < FR_MAX_ROUNDS + 2; k += blockDim.x) F.hist[k] = 0;
    for (int k = threadIdx.x; k < FR_FKS_MAXCHUNK; k += blockDim.x) F.cdirty[k] = 0;
    if (threadIdx.x == 0) {
        FksScal *S = F.scal;
        S->G0 = G0; S->n0 = st0.n_rem; S->n_in = st0.n_in;
        msg->L0 = G0; msg->changed = 0; msg->pad = 0;
        for (int p = 0; p < FR_FKS_PMAX; p++) { msg->totG[p] = 0; msg->totK[p] = 0; msg->totW[p] = G0; }
        S->zero_prefix = 1; S->changed = 0; S->done_ctr = 0; S->overflow = 0; S->valid_upto = -1;
        S->warm = 0; S->warm_scale = 1.0;
        if (inline_passes) { fr_fks2_passes(S, msg, 1, nullptr); fr_fks2_warm(S, F.saved, warm); }
    }
}

// after the all-gather of every rank's FksMsg (n_ranks > 1)
static __global__ void __launch_bounds__(64) k_fks_passes(Fks2Work F, const FksMsg *msgs, int n_ranks, int it, uint32_t *err, int warm) {
    FksScal *S = F.scal;
    if (threadIdx.x != 0) return;
    fr_fks2_passes(S, msgs, n_ranks, it >= 0 ? &F.hist[it] : nullptr, F.hm, it);
    if (it < 0) fr_fks2_warm(S, F.saved, warm);
    if (S->overflow) atomicOr(err, FR_ERR_ROUNDS);
}

// ------------------------------------------------------------------ 8-lane group primitives (DPP, no LDS traffic)
#define FR_DPP_XOR1 0xB1          // quad_perm [1,0,3,2]
#define FR_DPP_XOR2 0x4E          // quad_perm [2,3,0,1]
#define FR_DPP_HMIRROR 0x141      // row_half_mirror: lane i <-> 7 - i inside each 8 lanes
#define FR_DPP_ROR8 0x128          // row_ror:8: lane i <- lane (i + 8) % 16 inside each row of 16
template <int CTRL> __device__ __forceinline__ uint32_t fr_dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
template <int CTRL> __device__ __forceinline__ double fr_dpp_f64(double v) {
    long long b = __double_as_longlong(v);
    uint32_t lo = fr_dpp_u32<CTRL>((uint32_t)b), hi = fr_dpp_u32<CTRL>((uint32_t)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// sum over the 8 lanes of a group; every lane gets the same bits (pairs (i, 7-i), then the quad butterfly)
__device__ __forceinline__ double fr_grp8_sum(double v) {
    v = v + fr_dpp_f64<FR_DPP_HMIRROR>(v);
    v = v + fr_dpp_f64<FR_DPP_XOR1>(v);
    v = v + fr_dpp_f64<FR_DPP_XOR2>(v);
    return v;
}
__device__ __forceinline__ uint32_t fr_grp8_sum_u32(uint32_t v) {
    v += fr_dpp_u32<FR_DPP_HMIRROR>(v);
    v += fr_dpp_u32<FR_DPP_XOR1>(v);
    v += fr_dpp_u32<FR_DPP_XOR2>(v);
    return v;
}
// start - (c of lane 0) - (c of lane 1) - ... - (c of lane f-1), subtracted in that order (f = lane & 7)
__device__ __forceinline__ double fr_grp8_running(double start, double c, int f) {
    double g = start;
    double t;
    t = fr_dpp_f64<0x117>(c); if (f >= 7) g -= t;        // row_shr:7 -> lane f-7
    t = fr_dpp_f64<0x116>(c); if (f >= 6) g -= t;
    t = fr_dpp_f64<0x115>(c); if (f >= 5) g -= t;
    t = fr_dpp_f64<0x114>(c); if (f >= 4) g -= t;
    t = fr_dpp_f64<0x113>(c); if (f >= 3) g -= t;
    t = fr_dpp_f64<0x112>(c); if (f >= 2) g -= t;
    t = fr_dpp_f64<0x111>(c); if (f >= 1) g -= t;
    return g;
}

// One sub-weight row against a threshold: keeps, remaining weight, and the bounds needed to tell whether the
// decision would differ under a slightly smaller threshold.
template <int STAGE, bool NEW_HB>
__device__ __forceinline__ void fr_fks2_row(const HbTables &T, det_t det, uint32_t code, const RowInfo &ri, unsigned n_sub, double p_doub,
                                            double cwf, double gl, uint32_t kp_in, uint32_t *kp_out, uint32_t *add, double *unkept_wt, double *max_unkept, double *min_kept) {
    unsigned full = (n_sub / 8) * 8;
    uint32_t kk = kp_in, a = 0;
    double wsum = 0, mu = 0, mk = INFINITY;
    fr_row_visit<STAGE, NEW_HB>(T, det, code, ri, p_doub, [&](unsigned s, double w) {
        if (s >= n_sub || ((kk >> s) & 1u)) return;
        double sub_magn = cwf * w;
        double thr = s < full ? 1e-12 : 1e-10;      // compress_utils.cpp:213 / :233
        if (sub_magn >= gl && fabs(sub_magn) > thr) { kk |= 1u << s; a++; mk = sub_magn < mk ? sub_magn : mk; }
        else { wsum += w; mu = sub_magn > mu ? sub_magn : mu; }
    });
    *kp_out = kk; *add = a; *unkept_wt = wsum; *max_unkept = mu; *min_kept = mk;
}

#define FR_FKS_TILES_PER_CHUNK (FR_FKS_CHUNK * 8 / FR_BLOCK)

// MODE 0: early replays -- every tile, prefixes zero / from the warm start / from the previous replay, deltas written without
//         comparison (the replay counts as changed), no margins recorded: the lean kernel.
// MODE 1: the first comparing replay -- every wave decides, margins recorded, deltas compared with the stored ones (hist[it] / cdirty
//         raised on a difference).
// MODE 3: light replays -- MODE 1 for the waves whose inputs moved by more than their tightest comparison tolerates, the others stand
//         (a template instantiation of its own: the test costs registers the recording replay does not have).  light == 2: no allowance
//         for a changed number of sweeps.
// MODE 2: final pass -- wt_remain with the budget of the last sweep that flagged the element.
// MODE 4: a light replay and the final pass in one launch ("closing pass"): the waves whose inputs moved beyond their margins decide again
//         (MODE 3), every tile then gets its final wt_remain (MODE 2).  If no delta changed -- hist[it] stays 0 -- the stage has settled and
//         the final values stand: the confirming replay (sweep + scan + totals, which only establishes that nothing changes) is not launched
//         at all.  If one did change, the host scans, adds up and launches the closing pass again; what this one wrote is overwritten.
#define FR_FKS_TILE_MAXK 8192u      // a tile of FR_BLOCK elements preserves at most 32 sub-weights per element
#define FR_FKS_GRP_MAXK 256u        // a group of 8 elements likewise

// records the settled replay for the next iteration's warm start (one workgroup: the last one of the final pass)
__device__ __forceinline__ void fr_fks_save(Fks2Work F) {
    const FksScal *S = F.scal;
    const unsigned nb8 = S->n_in / 8 + 1;
    const unsigned nchunk = (nb8 + FR_FKS_CHUNK - 1) / FR_FKS_CHUNK;
    const int n_pass = S->n_pass < FR_FKS_PMAX ? S->n_pass : FR_FKS_PMAX;
    for (int p = 0; p < n_pass; p++)
        for (unsigned c = threadIdx.x; c < nchunk; c += blockDim.x) {
            const size_t ix = (size_t)p * FR_FKS_MAXCHUNK + c;
            F.wk[ix] = F.ck[ix]; F.wg[ix] = F.cg[ix]; F.wkx[ix] = F.ckx[ix]; F.wgx[ix] = F.cgx[ix];
        }
    if (threadIdx.x == 0) {
        FksSaved *V = F.saved;
        V->valid2 = V->valid; V->n_pass2 = V->n_pass; V->n02 = V->n0; V->G02 = V->G0;
        for (int p = 0; p < V->n_pass && p < FR_FKS_PMAX; p++) { V->psG2[p] = V->psG[p]; V->psN2[p] = V->psN[p]; }
        V->n_pass = n_pass; V->n0 = S->n0; V->nchunk = nchunk; V->nb8 = nb8; V->G0 = S->psG[0];
        for (int p = 0; p < n_pass; p++) { V->psG[p] = S->psG[p]; V->psN[p] = S->psN[p]; }
        V->valid = (S->overflow || !(S->psG[0] > 0)) ? 0 : 1;
    }
}

// Occupancy targets: the lean replay fits 64 registers (8 waves per SIMD) at the price of a few spilled words; the comparing replays need ~100
#ifndef FR_FKS_WPE0
#define FR_FKS_WPE0 5
#endif
#ifndef FR_FKS_WPE1
#define FR_FKS_WPE1 5
#endif
template <int STAGE, bool NEW_HB, int MODE>
__global__ void __launch_bounds__(FR_BLOCK) __attribute__((amdgpu_waves_per_eu(MODE == 0 ? FR_FKS_WPE0 : FR_FKS_WPE1))) k_fks_sweep(CompWork W, Fks2Work F, const HbTables *Tg, int cur, int it, double p_doub, int light, int dbg) {
    constexpr bool M1 = MODE == 1 || MODE == 3 || MODE == 4, LIGHT = MODE == 3 || MODE == 4, FIN = MODE == 2 || MODE == 4;       // comparing replays; MODE 3 / 4 skip the waves that stand; MODE 2 / 4 end with the final wt_remain
    __shared__ HbTables T;
    __shared__ FksScal S;
    // MODE 1: the deltas my wave's 8 groups stored for sweeps 0 .. FR_FKS_PF-1, fetched in one go when the tile starts (compared with the new
    // ones sweep by sweep: fetched there, each comparison is a dependent global-load latency on the wave's critical path)
    __shared__ double sh_dg[M1 ? FR_BLOCK / 64 : 1][M1 ? FR_FKS_PF * 8 : 1], sh_ws[M1 ? FR_BLOCK / 64 : 1][M1 ? FR_FKS_PF * 8 : 1];
    __shared__ uint32_t sh_dk[M1 ? FR_BLOCK / 64 : 1][M1 ? FR_FKS_PF * 8 : 1];
    {
        const uint32_t *src = (const uint32_t *)F.scal;
        uint32_t *dst = (uint32_t *)&S;
        for (unsigned i = threadIdx.x; i < sizeof(FksScal) / 4; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    const unsigned n_in = S.n_in;
    const unsigned nb8 = n_in / 8 + 1;                       // the reference also visits the (possibly empty) tail group
    const unsigned ntile = (unsigned)(((size_t)nb8 * 8 + FR_BLOCK - 1) / FR_BLOCK);
    const size_t stride = F.nb8_cap;
    const size_t wstride = F.nwv_cap;
    const StageElems E = W.el[cur];
    const int n_pass = S.n_pass;
    const bool zp = S.zero_prefix != 0;
    const bool warm0 = MODE == 0 && zp && S.warm;            // replay 0 of a warm start: prefixes from the saved chunk totals
    const int vup = S.valid_upto;
    const unsigned n_chunk_saved = warm0 ? F.saved->nchunk : 0u;
    const unsigned saved_nb8 = warm0 ? F.saved->nb8 : 0u;
    const double wsc = S.warm_scale;
    uint32_t *const dk8 = F.dk8;
    double *const dg8 = F.dg8, *const ws8 = F.ws8;
    const int lane = fr_lane(), f = lane & 7;
    // a replay without comparison, or with a different number of sweeps than its predecessor, always counts as changed
    if (MODE != 2 && (MODE == 0 || S.n_pass != S.valid_upto) && blockIdx.x == 0 && threadIdx.x == 0) { F.hist[it] = 1u; if (MODE == 4 && F.hm_close && it < FR_MAX_ROUNDS) F.hm->hist[it] = 1u; }
    if (MODE != 2 && dbg == 3 && it < FR_MAX_ROUNDS && blockIdx.x == 0 && threadIdx.x == 0) F.dbg_cnt[it * 4] = (uint32_t)n_pass;

    // Light replay: has anything this WAVE's decisions depend on moved by more than its tightest comparison tolerates?
    // A group of 8 elements is evaluated exactly in the reference's order from its start state (running norm G and remaining budget wf
    // at its first element), and the start state of group j of a wave is the wave's minus what groups 0 .. j-1 of the wave removed
    // (gamma_j) and used (kappa_j) according to the stored deltas.  Every comparison reads
    //     c >= (G_w - gamma_j - g) / (wf_w - kappa_j),      g = what the group itself removed before that point,
    // so what matters is by how much the right-hand side moves when (G_w, wf_w) become (G_w', wf_w') with gamma, kappa, g unchanged --
    // they are unchanged as long as no decision of the wave changes, which is the induction: group 0 first, then group 1, ...
    //     new / old = (1 + a)(1 + bb),   a = (G_w' - G_w) / (old numerator),   bb = (wf_w - wf_w') / (wf_w' - kappa_j).
    // The old numerator lies in [gmin, G_w] (gmin = the smallest norm any comparison of the wave saw), kappa_j in [0, kappa_7], and
    // a + bb + a bb is bilinear: its extremes are at the four corners.  Decisions that flipped upstream sit at the threshold, i.e. they
    // remove norm and budget in the threshold's own proportion, so a and bb cancel to first order -- the ratio moves ~1e3 x less than
    // norm and budget do one by one (which is what the first version of this test added up).  No comparison whose two sides differ by
    // more than that, relative to the larger side, can flip.  The evaluation that wrote the records may itself have moved the wave's
    // deltas, i.e. the prefixes inside the wave are no longer the ones its groups were evaluated with: group j's state has moved by the
    // wave's shift -+ at most (wdK, wdG), the summed moves of the wave's deltas, and a, bb become intervals (still bilinear: corners).
    // Lane p checks sweep p.  Returns the same value in every lane.
    auto wave_stands = [&](unsigned tile) -> bool {
        const size_t wv = (size_t)tile * (FR_BLOCK / 64) + (threadIdx.x >> 6);
        const size_t b0 = wv * 8;
        if (b0 >= nb8) return true;                              // no group here
        const unsigned chunk = tile / FR_FKS_TILES_PER_CHUNK;
        // The stage may run another number of sweeps than the wave last ran (the late sweeps preserve a handful of samples; whether a last
        // one with a single sample exists changes from replay to replay until the early ones have settled):
        //  * fewer (n_pass < rec_np): the sweeps that remain are checked as usual; row n_pass of the wave's deltas, a real sweep when the
        //    wave ran, must already read "nothing preserved, nothing removed" -- it then is the row beyond the last sweep, as it stands;
        //  * more  (n_pass > rec_np): every comparison of a new sweep was also made in the wave's last recorded sweep q (an element that is
        //    still compared was compared then, with the same left-hand side c), so the new sweeps preserve nothing in this wave if their
        //    threshold G / wf lies within q's tightest margin of q's thresholds (old numerator in [gmin, G_w], old denominator
        //    wf_w - kappa, kappa in [0, kappa_7]; the new sweep removes nothing inside the wave).  The wave then gets its rows and records
        //    of the new sweeps written here: deltas (0, 0, remaining weight as after sweep q), margin = q's minus what was used up.
        const uint32_t rec_np = F.wNp[wv];
        int ok = 1;
        double ext_G = 0.0, ext_r = 0.0, ext_m = 0.0; uint32_t ext_K = 0u;       // lanes of new sweeps: what their record will say
        if (rec_np == 0u || rec_np > (uint32_t)FR_FKS_PMAX || n_pass == 0 || (rec_np < (uint32_t)n_pass && (int)rec_np - 1 > vup) || (light == 2 && rec_np != (uint32_t)n_pass) || rec_np > (uint32_t)n_pass + 1u) ok = 0;      // (two sweeps fewer: the wave's decisions of the first dropped sweep are not looked at below)
        else if (lane < n_pass) {
            const int p = lane;
            const bool ext = p >= (int)rec_np;                       // a sweep the wave has not run: checked against its last recorded sweep
            const int q = ext ? (int)rec_np - 1 : p;
            double xg = 0.0; uint32_t xk = 0u, k7 = 0u;
            const size_t bl = b0 + 7 < nb8 ? b0 + 7 : (size_t)nb8 - 1;
            if (p <= vup) {
                const size_t cx = (size_t)p * FR_FKS_MAXCHUNK + chunk;
                const uint32_t x0 = F.xk8[(size_t)p * stride + b0];
                xg = F.cgx[cx] + F.xg8[(size_t)p * stride + b0]; xk = F.ckx[cx] + x0; k7 = F.xk8[(size_t)p * stride + bl] - x0;
            }
            if (ext) k7 = F.xk8[(size_t)q * stride + bl] - F.xk8[(size_t)q * stride + b0];      // (q <= vup: checked above)
            const double G_in = S.psG[p] - xg; const uint32_t K_in = S.psN[p] - xk;
            const FksWRec rec = F.wrec[wv * FR_FKS_PMAX + (size_t)q];
            const uint32_t K_old = rec.K; const double G_old = rec.G;
            const double rmar = (double)rec.R;
            const double ik = (double)rec.dK, ig_ = (double)rec.dG;       // the wave's own deltas moved by this much after its groups were evaluated
            const double gmin = (double)rec.M - ig_;
            int okp = 0;
            if (rmar == INFINITY) okp = 1;                  // the wave compared nothing in this sweep
            else if ((double)K_in > 2.0 * ((double)k7 + ik) + 64.0 && (double)K_old > 2.0 * ((double)k7 + ik) + 64.0 && gmin > 0 && G_old >= gmin) {
                const double dK = (double)K_old - (double)K_in, dG = G_in - G_old;
                const double dG_lo = dG - ig_, dG_hi = dG + ig_;
                const double h1 = 1.0 / G_old, h2 = 1.0 / gmin;
                double b_lo, b_hi;
                if (!ext) {
                    const double dK_lo = dK - ik, dK_hi = dK + ik;
                    const double q1 = 1.0 / (double)K_in, q2 = 1.0 / ((double)K_in - (double)k7 - ik);
                    b_lo = fmin(dK_lo * q1, dK_lo * q2); b_hi = fmax(dK_hi * q1, dK_hi * q2);
                }
                else { const double q1 = 1.0 / (double)K_in; b_lo = (dK - (double)k7 - ik) * q1; b_hi = (dK + ik) * q1; }       // (wf_w - kappa) / wf' - 1
                // a in [a_lo, a_hi], bb in [b_lo, b_hi]; a + bb + a bb is bilinear: extremes at the corners
                const double a_lo = fmin(dG_lo * h1, dG_lo * h2), a_hi = fmax(dG_hi * h1, dG_hi * h2);
                double r = fabs(a_lo + b_lo + a_lo * b_lo), t;
                t = fabs(a_lo + b_hi + a_lo * b_hi); r = t > r ? t : r;
                t = fabs(a_hi + b_lo + a_hi * b_lo); r = t > r ? t : r;
                t = fabs(a_hi + b_hi + a_hi * b_hi); r = t > r ? t : r;
                const double noise = 1e-12 * S.psG[0] * h2 * (1.0 + fabs(b_lo) + fabs(b_hi)) + 1e-13;     // prefix sums carry ~1e-16 of the stage's norm
                const double used = (r + noise) * 1.0001;
                okp = used < rmar;
                ext_r = (rmar - used) * 0.999;
#ifdef FR_FKS_CHKDBG
                if (dbg == 3 && it == 2 && STAGE == 1 && wv % 1500 == 7) printf("[chk] st %d wv %u p %d ext %d ok %d: K_old %u K_in %u k7 %u ik %.0f | G_old %.9e dG/G %.3e ig/G %.3e gmin/G %.6f | a %.3e %.3e b %.3e %.3e r %.3e rmar %.3e\n", STAGE, (unsigned)wv, p, (int)ext, okp, K_old, K_in, k7, ik, G_old, dG / G_old, ig_ / G_old, gmin / G_old, a_lo, a_hi, b_lo, b_hi, r, rmar);
#endif
            }
            else if (rmar != INFINITY) ext_r = 0.0;
            if (rmar == INFINITY) ext_r = INFINITY;
            if (!ext && K_in == K_old && G_in == G_old && ik == 0.0 && ig_ == 0.0) okp = 1;         // nothing moved at all
            ok = okp;
            ext_G = G_in; ext_K = K_in; ext_m = G_in * 0.9999;
        }
        if (ok && rec_np > (uint32_t)n_pass && lane >= 32 && lane < 40 && n_pass < FR_FKS_PMAX) {       // fewer sweeps: is row n_pass of my group empty?
            const size_t bj = b0 + (size_t)(lane - 32);
            if (bj < nb8) { const size_t ix = (size_t)n_pass * stride + bj; if (F.dk8[ix] != 0u || F.dg8[ix] != 0.0) ok = 0; }
        }
        const bool stands = __all(ok) != 0;
        if (stands && rec_np != (uint32_t)n_pass) {
            if (rec_np < (uint32_t)n_pass) {
                if (lane >= 32 && lane < 40) {               // the rows of the new sweeps and the new row beyond them
                    const size_t bj = b0 + (size_t)(lane - 32);
                    if (bj < nb8) {
                        const double gw = F.ws8[(size_t)rec_np * stride + bj];
                        for (int p = (int)rec_np + 1; p <= n_pass && p < FR_FKS_PMAX; p++) { const size_t ix = (size_t)p * stride + bj; F.dk8[ix] = 0u; F.dg8[ix] = 0.0; F.ws8[ix] = gw; }
                    }
                }
                if (lane >= (int)rec_np && lane < n_pass) {
                    FksWRec rn; rn.G = ext_G; rn.K = ext_K; rn.R = (float)ext_r * 0.9999f; rn.M = (float)ext_m; rn.dK = 0u; rn.dG = 0.0f; rn.pad = 0u;
                    F.wrec[wv * FR_FKS_PMAX + (size_t)lane] = rn;
                }
            }
            if (lane == 0) F.wNp[wv] = (uint32_t)n_pass;
        }
        return stands;
    };
    // Which of my tiles decide again is settled before the tables are staged: a workgroup with nothing to decide leaves at once
    // (late replays: almost all of them).  Bit k of `decide` = the k-th tile of this workgroup's stride (tiles beyond 32 are asked again below).
    uint32_t decide = 0;
    if (LIGHT) {
        int any = 0;
        unsigned k = 0;
        for (unsigned tile = blockIdx.x; tile < ntile; tile += gridDim.x, k++) {
            const bool d = !wave_stands(tile);
            if (d) { any = 1; if (k < 32) decide |= 1u << k; }
            if (dbg == 3 && it < FR_MAX_ROUNDS && lane == 0 && (size_t)tile * FR_BLOCK + threadIdx.x < (size_t)nb8 * 8) { atomicAdd(&F.dbg_cnt[it * 4 + 3], 1u); if (d) atomicAdd(&F.dbg_cnt[it * 4 + 2], 1u); }
        }
        if (MODE == 3 && !__syncthreads_or(any)) return;
    }
    if (FIN && blockIdx.x == gridDim.x - 1) fr_fks_save(F);       // (reads the settled chunk totals and scalars; nobody writes them in this launch)
    if (STAGE != 1) fr_stage_tables(&T, Tg);

    unsigned tile_k = 0;
    bool any_chg = false;           // (wave-uniform) this wave changed a delta somewhere
    for (unsigned tile = blockIdx.x; tile < ntile; tile += gridDim.x, tile_k++) {
        const size_t e = (size_t)tile * FR_BLOCK + threadIdx.x;
        const size_t b = e >> 3;
        const bool in_grp = b < nb8;
        const bool live = e < n_in;
        const unsigned my_chunk = tile / FR_FKS_TILES_PER_CHUNK;                 // uniform over the workgroup
        const size_t my_wave = e >> 6;
        bool decide_this = MODE != 2;                   // (wave-uniform)
        if (LIGHT) {
            const bool d = tile_k < 32 ? ((decide >> tile_k) & 1u) != 0 : !wave_stands(tile);
            if (!d) { if (MODE == 3) continue; decide_this = false; }      // nothing to decide in this wave
        }
        const bool lv = live, ig = in_grp;
        const double chunk_frac = (double)(b - (size_t)my_chunk * FR_FKS_CHUNK) * (1.0 / FR_FKS_CHUNK);
        // group start state of sweep p: what the groups before mine removed (norm) and used (samples) in that sweep
        // The loads are issued one sweep ahead and only added up when the sweep starts: any arithmetic on them here would make the
        // wave wait for the data on the spot.  a + b = norm removed before my group, c + d = samples used before it.
        struct Pfx { double a, b; uint32_t c, d; };
        auto prefix_issue = [&](int p, Pfx *o) {
            o->a = 0.0; o->b = 0.0; o->c = 0u; o->d = 0u;
            if (MODE == 0 && warm0) {        // previous iteration's chunk profile, linear inside the chunk (replay 0 only: resolved on the spot)
                double xg = 0.0; uint32_t xk = 0u;
                if (F.sxk8 && p < FR_FKS_SROWS && b < saved_nb8 && my_chunk < n_chunk_saved) {       // the same (stage 1) or the corresponding elements sat in this group last time
                    const size_t cx = (size_t)p * FR_FKS_MAXCHUNK + my_chunk;
                    xg = (F.wgx[cx] + F.sxg8[(size_t)p * stride + b]) * wsc;
                    xk = F.wkx[cx] + F.sxk8[(size_t)p * stride + b];
                }
                else if (my_chunk < n_chunk_saved) {
                    const size_t cx = (size_t)p * FR_FKS_MAXCHUNK + my_chunk;
                    xg = (F.wgx[cx] + F.wg[cx] * chunk_frac) * wsc;
                    xk = F.wkx[cx] + (uint32_t)((double)F.wk[cx] * chunk_frac);
                }
                else if (n_chunk_saved) {
                    const size_t cx = (size_t)p * FR_FKS_MAXCHUNK + n_chunk_saved - 1;
                    xg = (F.wgx[cx] + F.wg[cx]) * wsc; xk = F.wkx[cx] + F.wk[cx];
                }
                if (xk >= S.psN[p]) xk = S.psN[p] - 1;
                o->a = xg; o->c = xk;
            }
            else if (!zp && ig && p <= vup) {
                const size_t cx = (size_t)p * FR_FKS_MAXCHUNK + my_chunk;
                o->a = F.cgx[cx]; o->b = F.xg8[(size_t)p * stride + b];
                o->c = F.ckx[cx]; o->d = F.xk8[(size_t)p * stride + b];
            }
        };
        // the deltas my wave stored before (lane = 8 x sweep + group)
        const int wslot = threadIdx.x >> 6;
        if (M1 && decide_this) {
            const int ps = lane >> 3;
            const size_t bj = (my_wave << 3) + (size_t)f;
            uint32_t ok_ = 0u; double og = 0.0, ow = 0.0;
            if (ps <= n_pass && ps < FR_FKS_PMAX && bj < nb8) { const size_t ix = (size_t)ps * stride + bj; ok_ = dk8[ix]; og = dg8[ix]; ow = ws8[ix]; }
            sh_dk[wslot][lane] = ok_; sh_dg[wslot][lane] = og; sh_ws[wslot][lane] = ow;
        }
        // my element
        double v = lv ? E.val[e] : 0.0;
        uint32_t nd = lv ? E.ndiv[e] : 1u;
        double wr = v;
        uint32_t kp = (FIN && !decide_this && lv) ? W.keep[e] : 0u;
        det_t det = 0; uint32_t code = 0; RowInfo ri = fr_row1(W.row1);
        if (MODE == 3) { if (STAGE != 1 && lv) { code = E.code[e]; det = E.det[e]; ri = fr_row_cached(E, e); } }       // few waves, alone on their CU: one round of loads instead of two
        else if (STAGE != 1 && lv && nd == 0 && v > 0) { code = E.code[e]; det = E.det[e]; ri = fr_row_cached(E, e); }
        auto final_part = [&]() {
            double lastwf = 0;
            Pfx nx;
            if (n_pass > 0) prefix_issue(0, &nx);
            for (int p = 0; p < n_pass; p++) {
                const double xg = nx.a + nx.b; const uint32_t xk = nx.c + nx.d;
                // the settled per-group prefixes inside the chunks, kept for the next iteration's first replay of this stage
                if (F.sxk8 && f == 0 && in_grp && p < FR_FKS_SROWS) { F.sxk8[(size_t)p * stride + b] = nx.d; F.sxg8[(size_t)p * stride + b] = nx.b; }
                if (p + 1 < n_pass) prefix_issue(p + 1, &nx);
                double glob = S.psG[p] - xg, wf = (double)(S.psN[p] - xk);
                if (live && nd == 0 && v > 0 && v * wf >= glob) lastwf = wf;
            }
            if (live) {
                double out = v;
                if (nd > 0) { if (kp & 1u) out = 0; }
                else if (lastwf > 0) {
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
        };
        if (!decide_this) { if (FIN) final_part(); continue; }
        bool out_changed = false;
        float wmax = -1.0f;         // upper bound of the largest unpreserved normalised weight; < 0: row not looked at yet
        // (in this loop the sums are formed at once: measured faster than deferring them -- 62 vs 69 us -- while the final loop
        // above gains a third from deferring)
        Pfx nx;
        double xg_n = 0.0; uint32_t xk_n = 0u;
        if (n_pass > 0) { prefix_issue(0, &nx); xg_n = nx.a + nx.b; xk_n = nx.c + nx.d; }
        double gw_last = fr_grp8_sum(lv ? wr : 0.0);        // the group's remaining weight as of the last sweep that touched this wave
        for (int p = 0; p < n_pass; p++) {
            const double xg = xg_n; const uint32_t xk = xk_n;
            if (p + 1 < n_pass) { prefix_issue(p + 1, &nx); xg_n = nx.a + nx.b; xk_n = nx.c + nx.d; }
            const double glob0 = S.psG[p] - xg, wf = (double)(S.psN[p] - xk);
            // flags are taken against the group's start norm (compress_utils.cpp:172-180)
            double cw = v * wf;
            if (nd > 0) cw /= nd;
            const bool cmp = lv && wr > 0;
            const bool flagged = cmp && cw >= glob0;
            // distance of my comparisons from flipping, relative to their larger side (float is plenty; rounded down)
            float mr = INFINITY;
            if (M1 && cmp) mr = (float)fabs(cw - glob0) * __frcp_rn((float)(cw > glob0 ? cw : glob0));
            // --- speculate with the start norm, then validate against the running norm
            double change = 0, new_wr = wr, mu = 0, mk = INFINITY;
            uint32_t add = 0, new_kp = kp;
            bool evaluated = false, skipped = false;
            double used_gl = glob0;
            if (flagged) {
                if (nd > 0) { new_kp = kp | 1u; new_wr = 0; add = nd; change = v; }
                else if (wmax >= 0 && cw * (double)wmax < glob0) skipped = true;
            }
            bool need_eval = flagged && nd == 0 && !skipped;
            double gl_mine = glob0;
            const bool any_flagged = __any(flagged);
            if (any_flagged) {
                for (int round = 0; round < 9; round++) {
                    if (dbg == 3) {
                        if (need_eval) atomicAdd(&F.hist[FR_MAX_ROUNDS], 1u);
                        if (__any(need_eval) && lane == 0) atomicAdd(&F.hist[FR_MAX_ROUNDS + 1], 1u);
                    }
                    if (need_eval) {
                        unsigned n_sub = fr_row_len<STAGE, NEW_HB>(T, ri.nsub);
                        double uw;
                        fr_fks2_row<STAGE, NEW_HB>(T, det, code, ri, n_sub, p_doub, cw, gl_mine, kp, &new_kp, &add, &uw, &mu, &mk);
                        // Remaining weight of the row inside the replay: value x (sum of the unpreserved normalised sub-weights).  The
                        // reference forms sum(value * budget * w_s) / budget (compress_utils.cpp:243-245), the same number up to rounding
                        // but dependent on the budget; that form is used for the final wt_remain (final pass, bit for bit), while the
                        // replay's running norms only steer comparisons and use the budget-free form so that they stop moving as soon as
                        // the decisions do (<= 1e-16 relative, inside the tolerance the tree-summed norms already have).
                        new_wr = v * uw;
                        change = wr - new_wr;
                        evaluated = true; used_gl = gl_mine; skipped = false;
                    }
                    // running norm of each lane: start norm minus the changes of the flagged lanes before it, in order
                    gl_mine = fr_grp8_running(glob0, (flagged && !skipped) ? change : 0.0, f);
                    // would my decision differ under gl_mine?
                    need_eval = false;
                    if (flagged && nd == 0) {
                        if (skipped) { if (cw * (double)wmax >= gl_mine) need_eval = true; }
                        else if (evaluated && used_gl != gl_mine && mu >= gl_mine) need_eval = true;
                    }
                    if (!__any(need_eval)) break;
                }
                // the row comparisons, against the running norm they were finally decided with
                if (M1 && flagged && nd == 0) {
                    float m2;
                    if (need_eval || !(gl_mine > 0)) m2 = 0.0f;          // the validation loop ran out of rounds / the norm is gone: never skip this tile
                    else if (skipped) m2 = (float)(gl_mine - cw * (double)wmax) * __frcp_rn((float)gl_mine);
                    else {
                        m2 = (float)(gl_mine - mu) * __frcp_rn((float)gl_mine);
                        if (mk < INFINITY) { const float m3 = (float)(mk - gl_mine) * __frcp_rn((float)mk); m2 = m3 < m2 ? m3 : m2; }
                    }
                    mr = m2 < mr ? m2 : mr;
                }
            }
            // commit
            if (flagged && !skipped) {
                if (evaluated) wmax = (cw > 0) ? (float)((mu / cw) * 1.000002) : 0.0f;     // largest unpreserved normalised weight, rounded up
                kp = new_kp; wr = new_wr;
            }
            else { add = 0; change = 0; }
            // the wave's tightest comparison and smallest norm of this sweep: minima over the 8 lanes of a group, over the two groups of a
            // 16-lane row (DPP), then over the four rows by lane reads -- the result is wave-uniform
            uint32_t rec_r = 0u, rec_g = 0u;
            if (M1) {
                float gm = cmp ? (float)gl_mine * 0.99999f : INFINITY;      // gl_mine <= glob0: the smaller of my two right-hand sides
                if (!(mr >= 0.0f)) mr = 0.0f;
                if (!(gm >= 0.0f)) gm = 0.0f;
                uint32_t ur = __float_as_uint(mr * 0.9999f), ug = __float_as_uint(gm), t;
                t = fr_dpp_u32<FR_DPP_HMIRROR>(ur); ur = t < ur ? t : ur; t = fr_dpp_u32<FR_DPP_XOR1>(ur); ur = t < ur ? t : ur; t = fr_dpp_u32<FR_DPP_XOR2>(ur); ur = t < ur ? t : ur;
                t = fr_dpp_u32<FR_DPP_HMIRROR>(ug); ug = t < ug ? t : ug; t = fr_dpp_u32<FR_DPP_XOR1>(ug); ug = t < ug ? t : ug; t = fr_dpp_u32<FR_DPP_XOR2>(ug); ug = t < ug ? t : ug;
                t = fr_dpp_u32<FR_DPP_ROR8>(ur); ur = t < ur ? t : ur;
                t = fr_dpp_u32<FR_DPP_ROR8>(ug); ug = t < ug ? t : ug;
                rec_r = (uint32_t)__builtin_amdgcn_readlane((int)ur, 0); rec_g = (uint32_t)__builtin_amdgcn_readlane((int)ug, 0);
#pragma unroll
                for (int q = 16; q < 64; q += 16) {
                    const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)ur, q), y = (uint32_t)__builtin_amdgcn_readlane((int)ug, q);
                    rec_r = x < rec_r ? x : rec_r; rec_g = y < rec_g ? y : rec_g;
                }
            }
            // group totals (a wave without a flagged lane, the usual case from the third sweep on, has nothing new to add up)
            uint32_t gk = 0; double gg = 0.0, gw = gw_last;
            if (any_flagged) {
                gk = fr_grp8_sum_u32(add);
                gg = fr_grp8_sum(change);
                gw = fr_grp8_sum(lv ? wr : 0.0);
                gw_last = gw;
            }
            uint32_t dki = 0u; float dgi = 0.0f;        // MODE 1: by how much my group's deltas of this sweep moved
            if (f == 0 && ig) {
                size_t ix = (size_t)p * stride + b;
                if (M1) {
                    uint32_t ok_; double og, ow;
                    if (p < FR_FKS_PF) { const int sl = (p << 3) + (lane >> 3); ok_ = sh_dk[wslot][sl]; og = sh_dg[wslot][sl]; ow = sh_ws[wslot][sl]; }
                    else { ok_ = dk8[ix]; og = dg8[ix]; ow = ws8[ix]; }
                    if (ok_ != gk || __double_as_longlong(og) != __double_as_longlong(gg) || __double_as_longlong(ow) != __double_as_longlong(gw)) {
                        dki = gk > ok_ ? gk - ok_ : ok_ - gk; dgi = (float)fabs(gg - og) * 1.0001f;
                        dk8[ix] = gk; dg8[ix] = gg; ws8[ix] = gw; out_changed = true;
                    }
                }
                else { dk8[ix] = gk; dg8[ix] = gg; ws8[ix] = gw; }
            }
            if (M1) {
                // the wave's record of this sweep (lane 0 stores): start state, tightest comparison, smallest norm, moves of its deltas
                uint32_t sk = 0u; float sg = 0.0f;
                if (__any(dki != 0u || dgi != 0.0f)) {      // the leaders sit in lanes 0, 8, ..., 56
                    dki += fr_dpp_u32<FR_DPP_ROR8>(dki);
                    dgi += __uint_as_float(fr_dpp_u32<FR_DPP_ROR8>(__float_as_uint(dgi)));
#pragma unroll
                    for (int q = 0; q < 64; q += 16) { sk += (uint32_t)__builtin_amdgcn_readlane((int)dki, q); sg += __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(dgi), q)); }
                    sg *= 1.0001f;
                }
                if (lane == 0 && ig) {
                    FksWRec rn; rn.G = glob0; rn.K = S.psN[p] - xk; rn.R = __uint_as_float(rec_r); rn.M = __uint_as_float(rec_g); rn.dK = sk; rn.dG = sg; rn.pad = 0u;
                    F.wrec[my_wave * FR_FKS_PMAX + (size_t)p] = rn;
                }
            }
        }
        // one sweep beyond: keeps nothing, but its wt_remain sum is what a re-summed norm would be
        if (n_pass < FR_FKS_PMAX) {
            const double gw = gw_last;
            if (f == 0 && ig) {
                size_t ix = (size_t)n_pass * stride + b;
                if (M1) {
                    uint32_t ok_; double og, ow;
                    if (n_pass < FR_FKS_PF) { const int sl = (n_pass << 3) + (lane >> 3); ok_ = sh_dk[wslot][sl]; og = sh_dg[wslot][sl]; ow = sh_ws[wslot][sl]; }
                    else { ok_ = dk8[ix]; og = dg8[ix]; ow = ws8[ix]; }
                    if (ok_ != 0u || __double_as_longlong(og) != 0ll || __double_as_longlong(ow) != __double_as_longlong(gw)) {
                        dk8[ix] = 0; dg8[ix] = 0; ws8[ix] = gw; out_changed = true;
                    }
                }
                else { dk8[ix] = 0; dg8[ix] = 0; ws8[ix] = gw; }
            }
        }
        if (M1) {
            const bool chg = __any(out_changed) != 0;
            if (lane == 0) {
                // (hist[it] only ever becomes 1: a plain store.  The guard `hist[it] == 0` in front of an atomicOr read the wave's own L1, which
                // never sees another CU's atomic -- in the recording replay, where most waves change something, that was an atomic per wave
                // on one address)
                if (chg) { any_chg = true; F.cdirty[my_chunk] = (uint32_t)it + 1u; }
                if (ig) F.wNp[my_wave] = (uint32_t)n_pass;
            }
        }
        if (lv) { W.keep[e] = kp; if (!FIN) W.wt_remain[e] = wr; }
        if (FIN) final_part();          // MODE 4: with the keep bits just decided
    }
    if (M1 && any_chg && lane == 0) { F.hist[it] = 1u; if (MODE == 4 && F.hm_close && it < FR_MAX_ROUNDS) F.hm->hist[it] = 1u; }
}

// Chunk totals cross from the workgroups of k_fks_scan to the one that finishes last (fused totals) as relaxed device-scope atomics:
// written through to memory and read past the reader's L2, so no release fence -- which on this part writes the whole dirty L2 back, the
// 10 MB of prefixes the scan has just stored included (68 us per replay when every workgroup fenced) -- is needed for them to be seen;
// the writer waits for its stores to be acknowledged before it takes its ticket.
__device__ __forceinline__ void fr_st_agent(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void fr_st_agent(double *p, double v) { __hip_atomic_store((long long *)p, __double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t fr_ld_agent(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double fr_ld_agent(const double *p) { return __longlong_as_double(__hip_atomic_load((const long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }

// Exclusive prefixes over the chunks and sweep totals (one wave per sweep) into this rank's FksMsg; with one rank, also the sweep
// scalars of the next replay.
__device__ __forceinline__ void fr_fks_totals(Fks2Work F, uint32_t *err, FksMsg *msg, int inline_passes, int it, FksMsg *sm) {
    FksScal *S = F.scal;
    const unsigned nb8 = S->n_in / 8 + 1;
    const unsigned nchunk = (nb8 + FR_FKS_CHUNK - 1) / FR_FKS_CHUNK;
    const int n_pass = S->n_pass;
    const int lane = fr_lane(), wv = threadIdx.x >> 6, n_wv = blockDim.x >> 6;
    for (int q = wv; q < FR_FKS_PMAX; q += n_wv) {
        uint32_t rk = 0; double rg = 0, rw = 0;
        if (q <= n_pass) {
            for (unsigned c0 = 0; c0 < nchunk; c0 += 64) {
                unsigned c = c0 + lane;
                size_t ix = (size_t)q * FR_FKS_MAXCHUNK + c;
                uint32_t k = c < nchunk ? fr_ld_agent(&F.ck[ix]) : 0u;
                double g = c < nchunk ? fr_ld_agent(&F.cg[ix]) : 0.0, w = c < nchunk ? fr_ld_agent(&F.cw[ix]) : 0.0;
                uint32_t tk; double tg, tw;
                uint32_t ek = fr_wave_excl_u32(k, &tk);
                double eg = fr_wave_excl_f64(g, &tg);
                fr_wave_excl_f64(w, &tw);
                if (c < nchunk) { F.ckx[ix] = rk + ek; F.cgx[ix] = rg + eg; }
                rk += tk; rg += tg; rw += tw;
            }
        }
        if (lane == 0) { msg->totK[q] = rk; msg->totG[q] = rg; msg->totW[q] = rw; sm->totK[q] = rk; sm->totG[q] = rg; sm->totW[q] = rw; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        S->zero_prefix = 0;
        S->valid_upto = n_pass;         // the scan covered sweeps 0..n_pass of the replay that just ran
        const double L0 = S->G0; const uint32_t ch = F.hist[it];
        msg->L0 = L0; msg->changed = ch; msg->pad = 0;
        sm->L0 = L0; sm->changed = ch;
        if (inline_passes) {
            fr_fks2_passes(S, sm, 1, nullptr, F.hm, it);
            if (S->overflow) atomicOr(err, FR_ERR_ROUNDS);
            if (F.hm && it + 1 < FR_MAX_ROUNDS) F.hm->hist[it + 1] = 0u;        // the next replay may be a closing pass, which only ever raises its flag
        }
    }
}

// The three block scans of k_fks_scan (samples, norm removed, remaining weight) behind two barriers instead of eleven.  The arithmetic is that of
// fr_block_scan_u32 / fr_block_excl_f64 operation for operation: Hillis-Steele inside a wave, the waves' sums added in order, the block total
// = (exclusive prefix of thread 255) + (its own sum).  (Lane 0 of wave w gets the waves' running sum directly: that is the inclusive value
// of the last lane before it, which fr_block_excl_f64 fetches through LDS behind a barrier of its own.)
struct FksScan3 { uint32_t k[4]; double g[4], w[4]; uint32_t tk; double tg, tw; };
__device__ __forceinline__ void fr_block_scan3(uint32_t tk, double tg, double tw, FksScan3 *sh, uint32_t *excl_k, double *excl_g, uint32_t *totk, double *totg, double *totw) {
    const int lane = fr_lane(), w = threadIdx.x >> 6;
    uint32_t ik = tk; double ig = tg, iw = tw;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t a = __shfl_up(ik, off); const double b = __shfl_up(ig, off), c = __shfl_up(iw, off);
        if (lane >= off) { ik += a; ig += b; iw += c; }
    }
    if (lane == 63) { sh->k[w] = ik; sh->g[w] = ig; sh->w[w] = iw; }
    __syncthreads();
    uint32_t bk = 0; double bg = 0, bw = 0;
    for (int q = 0; q < w; q++) { bk += sh->k[q]; bg += sh->g[q]; bw += sh->w[q]; }
    const double incl_g = w ? bg + ig : ig, incl_w = w ? bw + iw : iw;
    double xg = __shfl_up(incl_g, 1), xw = __shfl_up(incl_w, 1);
    if (lane == 0) { xg = w ? bg : 0.0; xw = w ? bw : 0.0; }
    if (threadIdx.x == FR_BLOCK - 1) { sh->tk = bk + ik; sh->tg = xg + tg; sh->tw = xw + tw; }
    __syncthreads();
    *excl_k = bk + ik - tk; *excl_g = xg;
    *totk = sh->tk; *totg = sh->tg; *totw = sh->tw;
}

// Exclusive prefixes over the 8-blocks inside chunks of 2048 groups (grid: chunks x a few sweep lanes, each looping over the
// sweeps); chunk totals go to (ck, cg, cw).
static __global__ void __launch_bounds__(FR_BLOCK) k_fks_scan(Fks2Work F, int it, int light, int fuse, uint32_t *err, FksMsg *msg, int inline_passes) {
    __shared__ FksScan3 sh3;
    __shared__ uint32_t last_wg;
    __shared__ FksMsg sm_tot;
    const FksScal *S = F.scal;
    const unsigned c = blockIdx.x;
    const unsigned nb8 = S->n_in / 8 + 1;
    const unsigned nchunk = (nb8 + FR_FKS_CHUNK - 1) / FR_FKS_CHUNK;
    const int n_pass = S->n_pass;
    const size_t stride = F.nb8_cap;            // a multiple of FR_FKS_CHUNK: every chunk of a sweep's row is fully addressable
    // no delta of this chunk moved in this replay (and the sweeps are the same as before): its prefixes and totals stand
    if (light && !fuse && n_pass == S->valid_upto && F.hist[it] == 0u) return;       // the replay changed no delta: nothing to scan anywhere
    const bool stands = light && n_pass == S->valid_upto && F.cdirty[c] != (uint32_t)it + 1u;
    if (!stands && c < nchunk)
    for (int p = blockIdx.y; p <= n_pass && p < FR_FKS_PMAX; p += gridDim.y) {
        // thread t owns groups [8t, 8t + 8) of the chunk: 32 / 64 contiguous bytes per array, fetched as 16-byte vectors
        const size_t base = (size_t)p * stride + (size_t)c * FR_FKS_CHUNK + (size_t)threadIdx.x * 8;
        const unsigned left = (size_t)c * FR_FKS_CHUNK + (size_t)threadIdx.x * 8 < nb8 ? (unsigned)(nb8 - ((size_t)c * FR_FKS_CHUNK + (size_t)threadIdx.x * 8)) : 0u;   // valid groups of mine
        uint32_t k[8]; double g[8], w[8];
        {
            const uint4 *pk = (const uint4 *)(F.dk8 + base);
            uint4 a = pk[0], bq = pk[1];
            k[0] = a.x; k[1] = a.y; k[2] = a.z; k[3] = a.w; k[4] = bq.x; k[5] = bq.y; k[6] = bq.z; k[7] = bq.w;
            const double2 *pg = (const double2 *)(F.dg8 + base), *pw = (const double2 *)(F.ws8 + base);
#pragma unroll
            for (int j = 0; j < 4; j++) { double2 x = pg[j], y = pw[j]; g[2 * j] = x.x; g[2 * j + 1] = x.y; w[2 * j] = y.x; w[2 * j + 1] = y.y; }
        }
        uint32_t tk = 0; double tg = 0, tw = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if ((unsigned)j >= left) { k[j] = 0; g[j] = 0; w[j] = 0; }
            tk += k[j]; tg += g[j]; tw += w[j];
        }
        uint32_t totk, ek;
        double totg, totw, eg;
        fr_block_scan3(tk, tg, tw, &sh3, &ek, &eg, &totk, &totg, &totw);
        uint32_t xk[8]; double xg[8];
#pragma unroll
        for (int j = 0; j < 8; j++) { xk[j] = ek; xg[j] = eg; ek += k[j]; eg += g[j]; }
        {
            uint4 *qk = (uint4 *)(F.xk8 + base);
            qk[0] = make_uint4(xk[0], xk[1], xk[2], xk[3]); qk[1] = make_uint4(xk[4], xk[5], xk[6], xk[7]);
            double2 *qg = (double2 *)(F.xg8 + base);
#pragma unroll
            for (int j = 0; j < 4; j++) qg[j] = make_double2(xg[2 * j], xg[2 * j + 1]);
        }
        if (threadIdx.x == 0) {
            const size_t cx = (size_t)p * FR_FKS_MAXCHUNK + c;
            fr_st_agent(&F.ck[cx], totk); fr_st_agent(&F.cg[cx], totg); fr_st_agent(&F.cw[cx], totw);
        }
        __syncthreads();        // sh3 is reused by the next sweep
    }
    if (!fuse) return;
    // the workgroup that finishes last turns the chunk totals into this replay's sweep totals and the next replay's sweep scalars
    // (k_fks_totals, one launch less per replay); everybody else has read the scalars it is about to overwrite before taking a ticket
    if (threadIdx.x == 0) {
        __builtin_amdgcn_s_waitcnt(0);          // my chunk totals have been acknowledged
        const uint32_t t = atomicAdd(&F.scal->done_ctr, 1u);
        last_wg = (t + 1u == gridDim.x * gridDim.y) ? 1u : 0u;
    }
    __syncthreads();
    if (!last_wg) return;
    if (threadIdx.x == 0) F.scal->done_ctr = 0u;
    fr_fks_totals(F, err, msg, inline_passes, it, &sm_tot);
}

// (a separate launch: folding it into the last workgroup of k_fks_scan needs a device-scope fence in every workgroup, which on this part
// writes the L2 back each time -- measured 68 us per replay instead of 9 + 9)
#define FR_FKS_TOTALS_THREADS 1024      // a wave per sweep (the sweeps are independent until the bookkeeping of one thread at the end)
static __global__ void __launch_bounds__(FR_FKS_TOTALS_THREADS) k_fks_totals(Fks2Work F, uint32_t *err, FksMsg *msg, int inline_passes, int it) {
    __shared__ FksMsg sm;        // the bookkeeping is one thread chasing ~20 values per sweep: keep them in LDS
    // a replay that changed no delta (the confirming one, and those the host enqueued beyond it) leaves every total and scalar as it is
    if (it > 0 && F.hist[it] == 0u && F.scal->n_pass == F.scal->valid_upto && !F.scal->zero_prefix) {
        if (threadIdx.x == 0) { msg->changed = 0u; if (inline_passes && F.hm && it < FR_MAX_ROUNDS) { F.hm->hist[it] = 0u; if (it + 1 < FR_MAX_ROUNDS) F.hm->hist[it + 1] = 0u; } }
        return;
    }
    fr_fks_totals(F, err, msg, inline_passes, it, &sm);
}

// after the closing pass (k_fks_sweep MODE 4): did any rank change a delta?  The host reads hm->hist[it].
static __global__ void k_fks_close_put(Fks2Work F, uint32_t *send, int it) { send[0] = F.hist[it]; send[1] = 0u; send[2] = 0u; send[3] = 0u; }
// ranks: the closing pass's flag and -- computed behind it, ahead of the host's look -- this rank's remaining norm in ONE 16-byte message
// (the norm is what k_put_norm hands over: 0 when the budget is spent, compress_utils.cpp:267-269; garbage when this rank's flag is set, and then unused)
static __global__ void k_fks_close_put2(Fks2Work F, uint32_t *send, int it, const double *loc_total) {
    const FksScal *S = F.scal;
    send[0] = F.hist[it]; send[1] = 0u;
    ((double *)send)[1] = (S->G_last / S->n_last < 1e-8) ? 0.0 : *loc_total;
}
static __global__ void k_fks_close_flag2(Fks2Work F, const uint32_t *all, int n_ranks, int it, double *norms_out) {
    uint32_t ch = F.hist[it];
    for (int r = 0; r < n_ranks; r++) { ch |= all[4 * r]; norms_out[r] = ((const double *)all)[2 * r + 1]; }
    F.hist[it] = ch;
    if (it < FR_MAX_ROUNDS) F.hm->hist[it] = ch;
}
static __global__ void k_fks_close_flag(Fks2Work F, const uint32_t *all, int n_ranks, int it) {
    uint32_t ch = F.hist[it];
    for (int r = 0; r < n_ranks; r++) ch |= all[4 * r];
    F.hist[it] = ch;
    if (it < FR_MAX_ROUNDS) F.hm->hist[it] = ch;
}

// tie statistics (optional): the smallest relative margin any comparison of the settled stage has (the tiles' records of their last evaluation)
static __global__ void __launch_bounds__(FR_BLOCK) k_fks_tie(Fks2Work F, uint32_t *tie) {
    const FksScal *S = F.scal;
    const unsigned nb8 = S->n_in / 8 + 1;
    const int n_pass = S->n_pass < FR_FKS_PMAX ? S->n_pass : FR_FKS_PMAX;
    const size_t stride = F.nwv_cap, nwv = ((size_t)nb8 + 7) / 8;
    float m = INFINITY;
    for (int p = 0; p < n_pass; p++)
        for (size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwv; w += (size_t)gridDim.x * blockDim.x) { const float r = F.wrec[w * FR_FKS_PMAX + (size_t)p].R; m = r < m ? r : m; }
    for (int off = 32; off > 0; off >>= 1) { const float t = __shfl_xor(m, off); m = t < m ? t : m; }
    if (fr_lane() == 0 && m < INFINITY) atomicMin(&tie[0], __float_as_uint(m));
}

// work arrays of the parallel form of the in-order sweep (fks_seq.hpp)
#define FR_SQ_TILE 256u             // elements per workgroup of k_fsq_spec = 32 blocks of 8
#define FR_SQ_INF 0xFFFFFFFFu
struct FksSqCtl { uint32_t first_changed; uint32_t K_tot; double G_end, L_end; uint32_t n_mismatch; uint32_t pad; unsigned long long n_fast, n_dense; };
struct FksSq {
    double *dl, *nwr; uint32_t *nkp;            // per element: change of this sweep, wt_remain and keep after it
    double *gb, *lb, *dgb; uint32_t *kb, *dk;   // per block of 8: norm (global, local) and count entering it; change and count of the block
    uint32_t *tk, *tkx; double *tg, *tgx;       // per tile: count and change, their exclusive prefixes (tgx: norm entering the tile, approximate)
    uint8_t *tany;                              // per tile: the sweep touches an element of it
    // the chain as integer arithmetic (k_fsq_maps): per tile and chain (G: global norm, L: local norm) the binade the tile is assumed to be entered in, the
    // tile's total decrement in units of that binade's ulp, the sum of |change| (bounds the excursion), usable flag; what the chain found: exact entry
    // norms per tile, and whether the tile went through as one integer step
    double *mG, *mL, *sabs, *gt, *lt; int32_t *eG, *eL; uint8_t *mflag, *fast;
    double *gb2, *lb2;                          // FRIES_FSQ_CHECK: the element-by-element chain's entries, compared with the integer form's
    FksSqCtl *ctl;
};

