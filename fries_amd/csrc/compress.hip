// Vector side of one FRI iteration: death/cloning + column add (FRIES_bin/frisys_mol.cpp:487-499),
// exact-preservation selection (find_preserve, FRIES/compress_utils.cpp:29-105), systematic
// resampling in storage order (sys_comp, :283-327) and the projected-energy dot products
// (DistVec::dot, FRIES/vec_utils.hpp:228-238).
#include "ctx.hpp"
#include <cstring>

void fr_vcomp_alloc(FriesCtx *c, uint32_t cap) {
    VcompBuf &B = c->vc;
    B.keep = fr_alloc<uint8_t>(cap); B.del = fr_alloc<uint8_t>(cap); B.S = fr_alloc<double>(cap);
    for (int h = 0; h < 2; h++) { B.psum[h] = fr_alloc<double>(FR_MAX_PART); B.pcnt[h] = fr_alloc<uint32_t>(FR_MAX_PART); }
    B.state = fr_alloc<CompState>(FR_MAX_ROUNDS + 2);
    B.teeth = fr_alloc<Teeth>(1);
    B.dots = fr_alloc<double>(2);
    B.fix_list = fr_alloc<uint32_t>(FR_MAX_FIX);
    B.seq.tiles = fr_alloc<SeqRec>(FR_MAX_PART); B.seq.subs = fr_alloc<SeqRec>((size_t)FR_MAX_PART * FR_SUBS_PER_TILE); B.seq.total = fr_alloc<double>(1); B.seq.tsum = fr_alloc<double>(FR_MAX_PART);
    B.gnorm = fr_alloc<double>(1);
    FR_HIP(hipMemsetAsync(B.keep, 0, cap, c->stream));
    FR_HIP(hipMemsetAsync(B.del, 0, cap, c->stream));
    FR_HIP(hipMemsetAsync(B.state, 0, sizeof(CompState) * (FR_MAX_ROUNDS + 2), c->stream));
    if (fr_blocks(cap, FR_TILE) > FR_MAX_PART) throw FriesError("vector capacity exceeds FR_MAX_PART tiles");
}

// v0 <- v0 * (1 - eps (H_ii - S)) for the elements that existed before the spawns were merged,
// then v0 += v1, v1 <- 0; publishes per-block sums of |v0| (round 0 of find_preserve).
// add_col1 == 0: column 1 is left alone and only the block sums of |v0| are published (frifull_mol keeps the previous vector there)
__global__ void __launch_bounds__(FR_BLOCK) k_death_clone(VecDev V, VcompBuf B, SysDev S, uint32_t vec_size_before, double eps, double shift, uint32_t n_samp, int add_col1) {
    __shared__ double shd[12];
    const uint32_t n = V.st->curr_size;
    const unsigned nblk = (n + FR_TILE - 1) / FR_TILE;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        CompState s{};
        s.n_rem = n_samp; s.n_in = n; s.done = 0; s.pbuf = 0;
        B.state[0] = s;
    }
    if (blockIdx.x >= nblk) return;
    size_t base = (size_t)blockIdx.x * FR_TILE + threadIdx.x;      // lane-contiguous: coalesced loads and stores
    // Determinants created by the last merges have no diagonal element yet (NaN).  A few per wave: evaluated where they stand, each would hold its whole
    // wave for the ~100 dependent integral reads of one Slater-Condon diagonal, item after item.  They are listed in LDS instead and evaluated side by
    // side, one lane each (the sum inside an element keeps its order: same bits).
    __shared__ uint32_t sh_n, sh_idx[FR_TILE];
    __shared__ double sh_d[FR_TILE];
    if (threadIdx.x == 0) sh_n = 0;
    __syncthreads();
    double vv[FR_ITEMS], dd[FR_ITEMS]; int slot[FR_ITEMS];
#pragma unroll
    for (int it = 0; it < FR_ITEMS; it++) {
        const size_t i = base + (size_t)it * FR_BLOCK;
        vv[it] = 0; dd[it] = 0; slot[it] = -1;
        if (i >= n) continue;
        vv[it] = V.v0[i];
        if (i < vec_size_before && vv[it] != 0) {
            dd[it] = V.diag[i];
            if (dd[it] != dd[it]) { slot[it] = (int)atomicAdd(&sh_n, 1u); sh_idx[slot[it]] = (uint32_t)i; }
        }
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < sh_n; k += FR_BLOCK) {
        const uint32_t i = sh_idx[k];
        const double d = fr_diag_matrel(V.dets[i], S.h_core, S.eris, S.n_orb) - S.hf_en;
        V.diag[i] = d; sh_d[k] = d;
    }
    __syncthreads();
    double sum = 0;
#pragma unroll
    for (int it = 0; it < FR_ITEMS; it++) {
        const size_t i = base + (size_t)it * FR_BLOCK;
        if (i >= n) break;
        double v = vv[it];
        if (i < vec_size_before && v != 0) {
            const double d = slot[it] >= 0 ? sh_d[slot[it]] : dd[it];
            v *= 1 - eps * (d - shift);
        }
        if (add_col1) { v += V.v1[i] * 1.0; V.v0[i] = v; V.v1[i] = 0; }
        if (i >= V.n_dense) sum += fabs(v);     // the dense space is not part of the array find_preserve is given
    }
    double bs;
    fr_block_excl_f64(sum, shd, &bs);
    if (threadIdx.x == 0) { B.psum[0][blockIdx.x] = bs; B.pcnt[0][blockIdx.x] = 0; }
}

void fr_death_clone(FriesCtx *c, uint32_t vec_size_before) {
    SysDev S; S.n_orb = c->n_orb; S.n_elec = c->n_elec; S.h_core = c->d_h; S.eris = c->d_eris; S.hb = c->d_hb; S.hf_en = c->hf_en; S.spin_parity = c->spin_parity;
    uint32_t bound = c->h_vst.curr_size;
    FR_LAUNCH(c, "k_death_clone", k_death_clone, dim3(fr_blocks(bound ? bound : 1, FR_TILE)), dim3(FR_BLOCK), c->vec, c->vc, S, vec_size_before, c->eps, c->en_shift, c->vec_nonz, 1);
}
// round 0 of find_preserve on column 0 as it stands
void fr_abs_sums(FriesCtx *c) {
    SysDev S; S.n_orb = c->n_orb; S.n_elec = c->n_elec; S.h_core = c->d_h; S.eris = c->d_eris; S.hb = c->d_hb; S.hf_en = c->hf_en; S.spin_parity = c->spin_parity;
    uint32_t bound = c->h_vst.curr_size;
    FR_LAUNCH(c, "k_abs_sums", k_death_clone, dim3(fr_blocks(bound ? bound : 1, FR_TILE)), dim3(FR_BLOCK), c->vec, c->vc, S, 0u, c->eps, c->en_shift, c->vec_nonz, 0);
}

// One round of the exact-preservation fixed point: keep every element with
// |v| >= remaining_norm / remaining_samples (compress_utils.cpp:58).
struct FpMsg { double G; uint32_t kept, pad; };     // a rank's unkept norm and the elements it preserved in the last round

// this rank's totals of the previous round, for the all-gather (n_ranks > 1)
__global__ void __launch_bounds__(FR_BLOCK) k_fp_reduce(VcompBuf B, int round, FpMsg *msg) {
    __shared__ double shd[12];
    __shared__ uint32_t shu[4];
    const CompState prev = B.state[round - 1];
    const unsigned nblk = (prev.n_in + FR_TILE - 1) / FR_TILE;
    const int pin = (round - 1) & 1;
    double G = fr_sum_partials(B.psum[pin], nblk, shd);
    uint32_t k = fr_sum_partials_u32(B.pcnt[pin], nblk, shu);
    if (threadIdx.x == 0) { msg->G = G; msg->kept = k; msg->pad = 0; }
}

__global__ void __launch_bounds__(FR_BLOCK) k_fp_round(VecDev V, VcompBuf B, int round, const FpMsg *all, int n_ranks, uint32_t *tie) {
    __shared__ double shd[12];
    __shared__ uint32_t shu[4];
    const CompState prev = B.state[round - 1];
    const unsigned n = prev.n_in;
    const unsigned nblk = (n + FR_TILE - 1) / FR_TILE;
    if (prev.done) { if (blockIdx.x == 0 && threadIdx.x == 0) B.state[round] = prev; return; }
    if (blockIdx.x >= nblk && blockIdx.x != 0) return;
    const int pin = (round - 1) & 1, pout = round & 1;
    double G; uint32_t kept_prev;
    {      // sum_mpi: rank order (compress_utils.cpp:53, :76)
        G = 0; kept_prev = 0;
        for (int p = 0; p < n_ranks; p++) { G += all[p].G; kept_prev += all[p].kept; }
    }
    uint32_t n_rem = prev.n_rem - kept_prev;
    bool done = (round > 1 && kept_prev == 0) || (n_ranks == 1 && nblk == 0);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        CompState s = prev;
        if (round == 1) s.loc_norm = G;          // *global_norm (compress_utils.cpp:50)
        s.G = G; s.n_rem = n_rem; s.done = done; s.pbuf = pin;
        if (done) s.n_pass = round;          // the round that found nothing left to preserve (later rounds copy this state)
        B.state[round] = s;
    }
    if (done || blockIdx.x >= nblk) return;
    const double thr = G / n_rem;
    size_t base = (size_t)blockIdx.x * FR_TILE + threadIdx.x;      // lane-contiguous: coalesced loads and stores
    double sum = 0;
    uint32_t kept = 0;
    float mrel = INFINITY;       // tie statistics (optional): distance of the closest |v| >= norm / budget comparison from flipping, relative
    for (int it = 0; it < FR_ITEMS; it++) {
        size_t i = base + (size_t)it * FR_BLOCK;
        if (i >= n) break;
        double a = fabs(V.v0[i]);
        if (!B.keep[i] && a != 0) {
            if (tie) { const float m = (float)fabs(a - thr) / (float)(a > thr ? a : thr); mrel = m < mrel ? m : mrel; }
            if (a >= thr) { B.keep[i] = 1; kept++; }
            else sum += a;
        }
    }
    if (tie) {
        for (int off = 32; off > 0; off >>= 1) { const float t = __shfl_xor(mrel, off); mrel = t < mrel ? t : mrel; }
        if (fr_lane() == 0 && mrel < INFINITY) atomicMin(&tie[1], __float_as_uint(mrel));
    }
    double bs;
    fr_block_excl_f64(sum, shd, &bs);
    uint32_t bk = fr_block_sum_u32(kept, shu);
    if (threadIdx.x == 0) { B.psum[pout][blockIdx.x] = bs; B.pcnt[pout][blockIdx.x] = bk; }
}

template <class Acc>
static void run_seq(FriesCtx *c, SeqWork Q, Acc acc, uint32_t n_bound) {
    unsigned grid = fr_blocks(n_bound ? n_bound : 1, FR_SEQ_TILE);
    FR_LAUNCH(c, "k_seq_sums", (k_seq_sums<Acc>), dim3(grid), dim3(FR_BLOCK), Q, acc);
    FR_LAUNCH(c, "k_seq_maps", (k_seq_maps<Acc>), dim3(grid), dim3(FR_BLOCK), Q, acc, fr_seq_from_zero());
    FR_LAUNCH(c, "k_seq_chain", (k_seq_chain<Acc>), dim3(1), dim3(FR_BLOCK), Q, acc, fr_seq_from_zero());
}
// the chain again, starting from the lbound this rank inherits (classification and maps depend on the running sum's binade)
template <class Acc>
static void run_seq_from(FriesCtx *c, SeqWork Q, Acc acc, uint32_t n_bound, SeqStart from) {
    unsigned grid = fr_blocks(n_bound ? n_bound : 1, FR_SEQ_TILE);
    FR_LAUNCH(c, "k_seq_maps", (k_seq_maps<Acc>), dim3(grid), dim3(FR_BLOCK), Q, acc, from);
    FR_LAUNCH(c, "k_seq_chain", (k_seq_chain<Acc>), dim3(1), dim3(FR_BLOCK), Q, acc, from);
}

// exact in-order sum of the unpreserved |v| into vc.seq.total (what find_preserve returns, compress_utils.cpp:98-101)
void fr_unkept_norm(FriesCtx *c, uint32_t bound) {
    AccUnkept au{c->vec.v0, c->vc.keep, c->vec.st};
    run_seq(c, c->vc.seq, au, bound);
}

static __global__ void k_put_double(const double *src, double *dst) { *dst = *src; }
// sum_mpi of one double: 0 + x_0 + x_1 + ... in rank order (compress_utils.hpp:177-187)
static __global__ void k_sum_ranks(const double *all, int n_ranks, double *out) {
    double g = 0;
    for (int p = 0; p < n_ranks; p++) g += all[p];
    *out = g;
}

void fr_find_preserve(FriesCtx *c, uint32_t *n_samp_io, double *glob_norm) {
    VcompBuf &B = c->vc;
    hipStream_t st = c->stream;
    uint32_t bound = c->h_vst.curr_size ? c->h_vst.curr_size : 1;
    unsigned grid = fr_blocks(bound, FR_TILE);
    // *global_norm: exact in-order sum of |v| (compress_utils.cpp:34-35, :50)
    // the dense space keeps its values whatever the compression does: marked as already preserved (no share of the norm, no samples, no
    // deletion; sys_comp clears the marks again)
    if (c->vec.n_dense) FR_HIP(hipMemsetAsync(B.keep, 1, c->vec.n_dense, st));
    AccAbs aa{c->vec.v0, c->vec.st, c->vec.n_dense};
    SeqWork Qg = B.seq; Qg.total = B.gnorm;
    run_seq(c, Qg, aa, bound);
    const int P = c->n_ranks;
    if (c->use_comm) {
        FR_LAUNCH(c, "k_put_double", k_put_double, dim3(1), dim3(1), B.gnorm, (double *)c->comm.small_send);
        const double *all = (const double *)fr_allgather(c, sizeof(double));
        FR_LAUNCH(c, "k_sum_ranks", k_sum_ranks, dim3(1), dim3(1), all, P, B.gnorm);
    }
    int r = 0, batch = c->rounds_hint[6] + 1;
    CompState hs{};
    hs.done = 0;
    while (!hs.done) {
        if (r + batch > FR_MAX_ROUNDS) batch = FR_MAX_ROUNDS - r;
        if (batch <= 0) throw FriesError("find_preserve did not converge within FR_MAX_ROUNDS rounds");
        for (int k = 0; k < batch; k++) {
            r++;
            // one workgroup reduces the previous round's partials (every workgroup of the round would otherwise redo it)
            FR_LAUNCH(c, "k_fp_reduce", k_fp_reduce, dim3(1), dim3(FR_BLOCK), B, r, (FpMsg *)c->comm.small_send);
            const FpMsg *all = (const FpMsg *)fr_allgather(c, sizeof(FpMsg));
            FR_LAUNCH(c, "k_fp_round", k_fp_round, dim3(grid), dim3(FR_BLOCK), c->vec, B, r, all, P, c->d_tie);
        }
        const void *h_cs = fr_readback(c, &B.state[r], sizeof(CompState));
        uint32_t tk = 0;
        const void *h_gn = fr_readback(c, B.gnorm, 8, false, &tk);
        fr_stream_wait_ticket(c, tk);
        memcpy(&hs, h_cs, sizeof(CompState)); memcpy(glob_norm, h_gn, 8);
        batch = 2;
    }
    c->rounds_hint[6] = hs.n_pass > 3 ? (int)hs.n_pass - 1 : 2;     // next iteration's first batch = the rounds this one needed
    c->rounds_hint[7] = r;      // state slot sys_comp reads
    uint32_t n_rem = hs.n_rem;
    if (hs.G < 1e-9) n_rem = 0;     // compress_utils.cpp:94-96
    *n_samp_io = n_rem;
}

// norms: every rank's find_preserve result in rank order (frisys_mol.cpp:529, compress_utils.cpp:293-300);
// keep receives a copy that outlives the staging block (zeros when no samples are left: every lbound then starts at 0)
__global__ void k_vc_teeth(VcompBuf B, int last_round, uint32_t n_samp, double rn, const double *norms, int rank, int n_ranks, double *keep) {
    CompState s = B.state[last_round];
    const double loc_norm = n_samp ? *B.seq.total : 0.0;      // exact in-order sum of the non-preserved |v|
    s.n_rem = n_samp; s.loc_norm = loc_norm; s.n_fix = 0; s.n_out = 0;
    double lbound0 = 0;
    for (int p = 0; p < rank; p++) { double x = n_samp ? norms[p] : 0.0; lbound0 += x; keep[p] = x; }
    double glob = lbound0;
    for (int p = rank; p < n_ranks; p++) glob += (p == rank) ? loc_norm : (n_samp ? norms[p] : 0.0);
    double unit = 0, r0 = INFINITY;
    if (n_samp > 0) r0 = fr_seed_sys(rn, lbound0, glob, n_samp, &unit);
    s.unit = glob / n_samp;
    B.state[FR_MAX_ROUNDS + 1] = s;
    if (n_samp > 0) fr_build_teeth(B.teeth, r0, unit, n_samp + 2, lbound0);
    else { B.teeth->nseg = 0; B.teeth->kmax = 0; B.teeth->unit = 0; B.teeth->lbound0 = lbound0; }
}

// sys_comp body for one element (compress_utils.cpp:307-325).  Returns the new value.
__device__ __forceinline__ void fr_sc_element(VecDev &V, VcompBuf &B, const Teeth *th, size_t i, double Se, uint32_t *k, double unit, bool write) {
    double v = V.v0[i];
    if (B.keep[i]) { if (write) B.keep[i] = 0; return; }
    if (v == 0) return;
    if (fr_tooth(th, *k) < Se) {
        if (write) V.v0[i] = unit * ((v > 0) - (v < 0));
        (*k)++;
    }
    else if (write) { V.v0[i] = 0; B.del[i] = 1; }
}

__global__ void __launch_bounds__(FR_BLOCK) k_sc_apply(VecDev V, VcompBuf B, uint32_t *kin_out) {
    __shared__ SeqShared seqsh;
    CompState *fin = &B.state[FR_MAX_ROUNDS + 1];
    const unsigned n = fin->n_in;
    const unsigned nblk = (n + FR_TILE - 1) / FR_TILE;
    if (blockIdx.x >= nblk) return;
    __shared__ Teeth Tsh;
    fr_stage_teeth(&Tsh, B.teeth);
    const Teeth *th = &Tsh;
    AccUnkept acc{V.v0, B.keep, V.st};
    double Sx[4], Sprev;
    fr_seq_prefix4(B.seq, acc, blockIdx.x, &seqsh, Sx, &Sprev);
    size_t base = (size_t)blockIdx.x * FR_TILE + (size_t)threadIdx.x * FR_ITEMS;
#pragma unroll
    for (int it = 0; it < FR_ITEMS; it++) {
        size_t i = base + it;
        if (i >= n) break;
        double Se = Sx[it];
        uint32_t kin = (i == 0) ? 0u : fr_teeth_below(th, Sprev);
        uint32_t k = kin;
        B.S[i] = Se; kin_out[i] = kin;
        bool kept = B.keep[i];
        double v = V.v0[i];
        bool sel = !kept && v != 0 && fr_tooth(th, k) < Se;
        if (sel) k++;
        if (k != fr_teeth_below(th, Se)) {
            uint32_t slot = atomicAdd(&fin->n_fix, 1u);
            if (slot < FR_MAX_FIX) B.fix_list[slot] = (uint32_t)i;
        }
        Sprev = Se;
    }
}

// walks forward from every flagged element, handing the lagging tooth index to its successors
__global__ void __launch_bounds__(FR_BLOCK) k_sc_fixup(VecDev V, VcompBuf B, uint32_t *kin, uint32_t *err) {
    CompState *fin = &B.state[FR_MAX_ROUNDS + 1];
    uint32_t nf = fin->n_fix;
    if (nf == 0) return;
    if (nf > FR_MAX_FIX) { if (threadIdx.x == 0) atomicOr(err, FR_ERR_BACKLOG); nf = FR_MAX_FIX; }
    const Teeth *th = B.teeth;
    const unsigned n = fin->n_in;
    fr_sort_fix_list(B.fix_list, nf);
    if (threadIdx.x != 0) return;
    size_t done_upto = 0;
    for (uint32_t q = 0; q < nf; q++) {
        size_t e = B.fix_list[q];
        if (e < done_upto) continue;
        uint32_t k = kin[e];
        { bool sel = !B.keep[e] && V.v0[e] != 0 && fr_tooth(th, k) < B.S[e]; if (sel) k++; }
        for (size_t e2 = e + 1; e2 < n; e2++) {
            kin[e2] = k;
            bool sel = !B.keep[e2] && V.v0[e2] != 0 && fr_tooth(th, k) < B.S[e2];
            if (sel) k++;
            done_upto = e2 + 1;
            if (k == fr_teeth_below(th, B.S[e2])) break;
        }
    }
}

__global__ void __launch_bounds__(FR_BLOCK) k_sc_write(VecDev V, VcompBuf B, const uint32_t *kin) {
    __shared__ Teeth Tsh;
    fr_stage_teeth(&Tsh, B.teeth);
    CompState *fin = &B.state[FR_MAX_ROUNDS + 1];
    const unsigned n = fin->n_in;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t k = kin[i];
    fr_sc_element(V, B, &Tsh, i, B.S[i], &k, fin->unit, true);
}

// find_preserve returns 0 when the budget is spent (compress_utils.cpp:94-96)
static __global__ void k_put_norm_vc(const double *total, uint32_t n_samp, double *out) { *out = n_samp ? *total : 0.0; }

void fr_sys_comp(FriesCtx *c, uint32_t n_samp, double rn) {
    VcompBuf &B = c->vc;
    hipStream_t st = c->stream;
    uint32_t bound = c->h_vst.curr_size ? c->h_vst.curr_size : 1;
    uint32_t *kin = c->W.kin;      // HB-PP scratch is idle here
    AccUnkept au{c->vec.v0, B.keep, c->vec.st};
    run_seq(c, B.seq, au, bound);
    const int P = c->n_ranks;
    const double *norms = B.seq.total;
    if (c->use_comm) {
        FR_LAUNCH(c, "k_put_norm", k_put_norm_vc, dim3(1), dim3(1), B.seq.total, n_samp, (double *)c->comm.small_send);
        norms = (const double *)fr_allgather(c, sizeof(double));
    }
    FR_LAUNCH(c, "k_vc_teeth", k_vc_teeth, dim3(1), dim3(1), B, c->rounds_hint[7], n_samp, rn, norms, c->rank, P, c->d_norms_keep);
    if (c->rank > 0) {
        SeqStart from; from.norms = c->d_norms_keep; from.n = c->rank;
        SeqWork Q2 = B.seq; Q2.total = c->d_seq_scratch;
        run_seq_from(c, Q2, au, bound, from);
    }
    FR_LAUNCH(c, "k_sc_apply", k_sc_apply, dim3(fr_blocks(bound, FR_TILE)), dim3(FR_BLOCK), c->vec, B, kin);
    FR_LAUNCH(c, "k_sc_fixup", k_sc_fixup, dim3(1), dim3(FR_BLOCK), c->vec, B, kin, c->d_err);
    FR_LAUNCH(c, "k_sc_write", k_sc_write, dim3(fr_blocks(bound, FR_BLOCK)), dim3(FR_BLOCK), c->vec, B, kin);
    if (c->hh_keep0) fr_hh_clear_pos0(c);        // frisys_hh.cpp:356
    fr_vec_delete_flagged(c, &c->vec, B.del, bound);
}

// ------------------------------------------------------------------ frimulti_mol: samples per column (frimulti_mol.cpp:301-322)
// One comb over the in-order running sum of |v|: first tooth from seed_sys with the norms the last compression left (one rank: the
// in-order sum itself), spacing = the one-norm BEFORE that compression / n_teeth -- two different numbers in the reference, kept apart.
// n_walk[i] = teeth in (S_{i-1}, S_i]... strictly: teeth below S_i minus teeth below S_{i-1} (`while (rn_sys < lbound)`).
// norms: every rank's in-order |v| sum in rank order (one rank: its own); keep receives the lower ranks' for the prefix sums
__global__ void k_mc_teeth(VcompBuf B, double rn, double prev_glob, uint32_t n_teeth, double *unit_out, const double *norms, int rank, int n_ranks, double *keep) {
    double lbound0 = 0;             // seed_sys (compress_utils.cpp:107-127)
    for (int p = 0; p < rank; p++) { lbound0 += norms[p]; keep[p] = norms[p]; }
    double T = lbound0;
    for (int p = rank; p < n_ranks; p++) T += norms[p];
    const double G = prev_glob < 0 ? T : prev_glob;
    double r0 = rn * (T / n_teeth);
    r0 += T / n_teeth * (int)(lbound0 * n_teeth / T);
    if (r0 < lbound0) r0 += T / n_teeth;
    const double unit = G / n_teeth;
    fr_build_teeth(B.teeth, r0, unit, n_teeth + 64, lbound0);
    unit_out[0] = unit; unit_out[1] = T;
}
__global__ void __launch_bounds__(FR_BLOCK) k_mc_walk(VecDev V, VcompBuf B, uint32_t *n_walk, uint32_t *err) {
    __shared__ SeqShared seqsh;
    __shared__ Teeth Tsh;
    const unsigned n = V.st->curr_size;
    const unsigned nblk = (n + FR_TILE - 1) / FR_TILE;
    if (blockIdx.x >= nblk) return;
    fr_stage_teeth(&Tsh, B.teeth);
    const Teeth *th = &Tsh;
    AccAbs acc{V.v0, V.st};
    double Sx[4], Sprev;
    fr_seq_prefix4(B.seq, acc, blockIdx.x, &seqsh, Sx, &Sprev);
    size_t base = (size_t)blockIdx.x * FR_TILE + (size_t)threadIdx.x * FR_ITEMS;
    uint32_t kprev = fr_teeth_below(th, Sprev);
#pragma unroll
    for (int it = 0; it < FR_ITEMS; it++) {
        size_t i = base + it;
        if (i >= n) break;
        uint32_t k = fr_teeth_below(th, Sx[it]);
        n_walk[i] = k - kprev;
        if (k >= th->kmax) atomicOr(err, FR_ERR_SPAWN_CAP);      // the comb ran past its table: the vector's norm moved by more than 64 sampling units
        kprev = k;
    }
}
// exact in-order sum of |v| over the stored vector (DistVec::local_norm), to the host
double fr_abs_norm(FriesCtx *c) {
    const uint32_t bound = c->h_vst.curr_size ? c->h_vst.curr_size : 1;
    AccAbs aa{c->vec.v0, c->vec.st};
    run_seq(c, c->vc.seq, aa, bound);
    double t = 0;
    FR_HIP(hipMemcpyAsync(&t, c->vc.seq.total, 8, hipMemcpyDeviceToHost, c->stream));
    FR_HIP(hipStreamSynchronize(c->stream));
    return t;
}
void fr_multi_walks(FriesCtx *c, double rn, double prev_glob_norm, uint32_t n_teeth, uint32_t *n_walk, double *unit_out) {
    VcompBuf &B = c->vc;
    const uint32_t bound = c->h_vst.curr_size ? c->h_vst.curr_size : 1;
    AccAbs aa{c->vec.v0, c->vec.st};
    run_seq(c, B.seq, aa, bound);
    const double *norms = B.seq.total;
    if (c->use_comm) {          // loc_norms of the last compression = every rank's in-order sum now (frimulti_mol.cpp:227-229, 414)
        FR_LAUNCH(c, "k_put_double", k_put_double, dim3(1), dim3(1), B.seq.total, (double *)c->comm.small_send);
        norms = (const double *)fr_allgather(c, sizeof(double));
    }
    FR_LAUNCH(c, "k_mc_teeth", k_mc_teeth, dim3(1), dim3(1), B, rn, prev_glob_norm, n_teeth, unit_out, norms, c->rank, c->n_ranks, c->d_norms_keep);
    if (c->rank > 0) {          // this rank's prefix sums continue the lower ranks' (the comb is global)
        SeqStart from; from.norms = c->d_norms_keep; from.n = c->rank;
        SeqWork Q2 = B.seq; Q2.total = c->d_seq_scratch;
        run_seq_from(c, Q2, aa, bound, from);
    }
    FR_LAUNCH(c, "k_mc_walk", k_mc_walk, dim3(fr_blocks(bound, FR_TILE)), dim3(FR_BLOCK), c->vec, B, n_walk, c->d_err);
}

// block 0: <H trial | v>, block 1: <trial | v>.  The products are formed in parallel, the additions by one lane IN LIST ORDER, absent
// determinants contributing +0: the doubles of the reference's loop (vec_utils.hpp:228-238), not a tree sum of them.
__global__ void __launch_bounds__(FR_BLOCK) k_dots(VecDev V, const det_t *hd, const double *hv, uint32_t nh, const det_t *td, const double *tv, uint32_t nt, double *out) {
    __shared__ double prod[FR_BLOCK];
    const det_t *d = blockIdx.x == 0 ? hd : td;
    const double *w = blockIdx.x == 0 ? hv : tv;
    const uint32_t n = blockIdx.x == 0 ? nh : nt;
    double acc = 0;
    for (uint32_t base = 0; base < n; base += FR_BLOCK) {
        const uint32_t i = base + threadIdx.x;
        double p = 0;
        if (i < n) {
            uint32_t s = fr_hash_find(V, d[i]);
            if (s != FR_NOPOS) { uint32_t pos = V.hs[s].val; if (pos < V.cap) p = w[i] * V.v0[pos]; }
        }
        prod[threadIdx.x] = p;
        __syncthreads();
        if (threadIdx.x == 0) { const uint32_t m = n - base < FR_BLOCK ? n - base : FR_BLOCK; for (uint32_t j = 0; j < m; j++) acc += prod[j]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

// the two halves of fr_dots for a caller that synchronises the stream anyway in between (frisys_iterate: find_preserve's rounds)
const void *fr_dots_enqueue(FriesCtx *c) {
    FR_LAUNCH(c, "k_dots", k_dots, dim3(2), dim3(FR_BLOCK), c->vec, c->htr_det, c->htr_val, c->n_htrial, c->tr_det, c->tr_val, c->n_trial, c->vc.dots);
    const double *src = c->vc.dots;
    if (c->use_comm) {
        FR_HIP(hipMemcpyAsync(c->comm.small_send, c->vc.dots, 16, hipMemcpyDeviceToDevice, c->stream));
        src = (const double *)fr_allgather(c, 16);
    }
    return fr_readback(c, src, 16 * (size_t)c->n_ranks, true);      // read after find_preserve's rounds, each of which takes ring slots
}
void fr_dots_collect(FriesCtx *c, const void *h_d, double *numer, double *denom) {
    const int P = c->n_ranks;
    double h[2 * FR_MAX_RANKS];
    memcpy(h, h_d, 16 * (size_t)P);
    double nu = 0, de = 0;          // sum_mpi in rank order (frisys_mol.cpp:512-517)
    if (c->dots_slot0_from_hf && P > 1) { h[0] = h[2 * c->hf_proc]; h[1] = h[2 * c->hf_proc + 1]; }
    for (int p = 0; p < P; p++) { nu += h[2 * p]; de += h[2 * p + 1]; }
    *numer = nu; *denom = de;
}
void fr_dots(FriesCtx *c, double *numer, double *denom) {
    FR_LAUNCH(c, "k_dots", k_dots, dim3(2), dim3(FR_BLOCK), c->vec, c->htr_det, c->htr_val, c->n_htrial, c->tr_det, c->tr_val, c->n_trial, c->vc.dots);
    const int P = c->n_ranks;
    double h[2 * FR_MAX_RANKS];
    const double *src = c->vc.dots;
    if (c->use_comm) {
        FR_HIP(hipMemcpyAsync(c->comm.small_send, c->vc.dots, 16, hipMemcpyDeviceToDevice, c->stream));
        src = (const double *)fr_allgather(c, 16);
    }
    const void *h_d = fr_readback(c, src, 16 * (size_t)P);
    FR_HIP(hipStreamSynchronize(c->stream));
    memcpy(h, h_d, 16 * (size_t)P);
    double nu = 0, de = 0;          // sum_mpi in rank order (frisys_mol.cpp:512-517)
    if (c->dots_slot0_from_hf && P > 1) { h[0] = h[2 * c->hf_proc]; h[1] = h[2 * c->hf_proc + 1]; }      // fciqmc_fp_mol.cpp:461-462: slot 0 overwritten by the gathering rank
    for (int p = 0; p < P; p++) { nu += h[2 * p]; de += h[2 * p + 1]; }
    *numer = nu; *denom = de;
}
