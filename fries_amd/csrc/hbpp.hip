// apply_HBPP_sys on the device (FRIES/Hamiltonians/heat_bathPP.cpp:686-992): five comp_sub
// stages with on-the-fly sub-weight rows, then weight / matrix element / parity and an
// order-preserving compaction of the surviving samples.
#include "ctx.hpp"
#include <cstring>
#include "fks2.hpp"
#include "fks_seq.hpp"

// ------------------------------------------------------------------ stage preparation
// Stage 1 elements are the stored vector elements (frisys_mol.cpp:414-420, heat_bathPP.cpp:714-727).
__global__ void __launch_bounds__(FR_BLOCK) k_prep1(CompWork W, VecDev V, int cur, uint32_t n_samp) {
    __shared__ double shd[12];
    const unsigned nd = V.n_dense;                      // the dense space in front is multiplied exactly, not compressed (frisys_mol.cpp:414-420)
    const unsigned n_in = V.st->curr_size - nd;
    const unsigned nblk = (n_in + FR_TILE - 1) / FR_TILE;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        CompState s{};
        s.n_rem = n_samp; s.n_in = n_in; s.done = 0; s.pbuf = 0;
        W.state[0] = s;
    }
    if (blockIdx.x >= nblk) return;
    const StageElems E = W.el[cur];
    size_t base = (size_t)blockIdx.x * FR_TILE + threadIdx.x;      // lane-contiguous: coalesced loads and stores
    double sum = 0;
    for (int it = 0; it < FR_ITEMS; it++) {
        size_t e = base + (size_t)it * FR_BLOCK;
        if (e >= n_in) break;
        double w = fabs(V.v0[e + nd]);
        E.val[e] = w; E.pos[e] = (uint32_t)(e + nd); E.code[e] = 0; E.ndiv[e] = (w > 0) ? 0u : 1u; E.nsub[e] = 2; E.rinv[e] = 1.0; E.raux[e] = 0;
        W.wt_remain[e] = w; W.keep[e] = 0;
        sum += w;
    }
    double bs;
    fr_block_excl_f64(sum, shd, &bs);
    if (threadIdx.x == 0) { W.psum[0][blockIdx.x] = bs; W.pcnt[0][blockIdx.x] = 0; }
}

// Stages 2..5: element e comes from emission e of the previous stage
// (heat_bathPP.cpp:739-762, 773-809, 821-857, 869-908).
template <int STAGE, bool NEW_HB>
__global__ void __launch_bounds__(FR_BLOCK) k_prep(CompWork W, VecDev V, const HbTables *Tg, int cur, uint32_t n_samp, double p_doub) {
    __shared__ HbTables T;
    __shared__ double shd[12];
    const unsigned n_in = W.state[FR_MAX_ROUNDS + 1].n_out;
    const unsigned nblk = (n_in + FR_TILE - 1) / FR_TILE;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        CompState s{};
        s.n_rem = n_samp; s.n_in = n_in; s.done = 0; s.pbuf = 0;
        W.state[0] = s;
    }
    if (blockIdx.x >= nblk) return;
    fr_stage_tables(&T, Tg);
    const StageElems E = W.el[cur], P = W.el[cur ^ 1];
    const unsigned n_elec = T.n_elec, n_orb = T.n_orb;
    size_t base = (size_t)blockIdx.x * FR_TILE + threadIdx.x;      // lane-contiguous: coalesced loads and stores
    double sum = 0;
    for (int it = 0; it < FR_ITEMS; it++) {
        size_t e = base + (size_t)it * FR_BLOCK;
        if (e >= n_in) break;
        uint32_t wi = W.e_wi[e], sub = W.e_sub[e];
        double val = W.e_val[e];
        uint32_t pos = P.pos[wi], pc = P.code[wi];
        det_t det = V.dets[pos];
        uint32_t code = 0, ndiv = 0, nsub = 0, raux = 0;
        double rinv = 1.0;
        if (STAGE == 2) {
            code = fr_code(sub, 0, 0, 0);
            if (sub == 0) {
                RowInfo ri = fr_row2_setup<NEW_HB>(T, det);
                if (NEW_HB) val *= ri.tot;
                nsub = ri.nsub; rinv = ri.inv_norm; raux = ri.aux;
            }
            else {
                unsigned n_occ = fr_count_sing_allowed(T, det);
                if (n_occ == 0) { ndiv = 1; val = 0; }
                else ndiv = n_occ;
            }
        }
        else if (STAGE == 3) {
            unsigned sd = fr_c(pc, 0), c1 = sub;
            if (c1 >= n_elec) { val = 0; ndiv = 1; code = fr_code(sd, c1, 0, 0); }
            else if (sd == 0) {
                if (NEW_HB) {
                    c1++;
                    RowInfo ri = fr_row3h_setup(T, det, c1);
                    nsub = c1; rinv = ri.inv_norm; raux = ri.aux;
                    val *= ri.tot;
                }
                else { RowInfo ri = fr_row3_setup(T, det, c1); nsub = n_elec; rinv = ri.inv_norm; raux = ri.aux; }
                code = fr_code(0, c1, 0, 0);
            }
            else {
                unsigned choice = c1;
                unsigned n_virt = fr_count_sing_virt(T, det, &choice);
                if (n_virt == 0) { ndiv = 1; val = 0; code = fr_code(1, choice, 0, 0); }
                else { ndiv = n_virt; code = fr_code(1, choice, 0, n_virt); }
            }
        }
        else if (STAGE == 4) {
            unsigned sd = fr_c(pc, 0), o1_idx = fr_c(pc, 1), o2u1 = sub;
            if (sd == 0) {
                if (o2u1 >= n_elec) { val = 0; ndiv = 1; code = fr_code(0, o1_idx, o2u1, 0); }
                else {
                    code = fr_code(0, o1_idx, o2u1, 0);
                    RowInfo ri = fr_row_setup<4, NEW_HB>(T, det, code, p_doub);
                    if (NEW_HB) val *= ri.tot;
                    nsub = ri.nsub; rinv = ri.inv_norm; raux = ri.aux;
                }
            }
            else { code = fr_code(1, o1_idx, o2u1, fr_c(pc, 3)); ndiv = 1; }
        }
        else {  // STAGE == 5
            unsigned sd = fr_c(pc, 0), o1_idx = fr_c(pc, 1), o2_idx = fr_c(pc, 2);
            if (sd == 0) {
                unsigned u1 = fr_find_nth_virt(det, o1_idx / (n_elec / 2), n_orb, sub);
                if (fr_bit(det, u1)) { val = 0; ndiv = 1; code = fr_code(0, o1_idx, o2_idx, u1); }
                else {
                    code = fr_code(0, o1_idx, o2_idx, u1);
                    RowInfo ri = fr_row_setup<5, NEW_HB>(T, det, code, p_doub);
                    nsub = ri.nsub; rinv = ri.inv_norm; raux = ri.aux;
                    if (NEW_HB || ri.tot == 0) val *= ri.tot;
                }
            }
            else { code = fr_code(1, o1_idx, o2_idx, fr_c(pc, 3)); ndiv = 1; }
        }
        E.val[e] = val; E.pos[e] = pos; E.code[e] = code; E.ndiv[e] = ndiv; E.nsub[e] = nsub; E.rinv[e] = rinv; E.raux[e] = raux; E.det[e] = det;
        W.wt_remain[e] = val; W.keep[e] = 0;
        sum += val;
    }
    double bs;
    fr_block_excl_f64(sum, shd, &bs);
    if (threadIdx.x == 0) { W.psum[0][blockIdx.x] = bs; W.pcnt[0][blockIdx.x] = 0; }
}

// ------------------------------------------------------------------ final evaluation (heat_bathPP.cpp:917-991)
// f_val[e] == 0 marks an unsuccessful sample.
template <bool NEW_HB>
__global__ void __launch_bounds__(FR_BLOCK) k_final_eval(CompWork W, VecDev V, SysDev S, int prev, double p_doub, int unit_matrel,
                                                         double *f_val, uint32_t *f_orbs, uint32_t *pcnt) {
    __shared__ HbTables T;
    __shared__ uint32_t shu[4];
    const unsigned n_in = W.state[FR_MAX_ROUNDS + 1].n_out;
    const unsigned nblk = (n_in + FR_TILE - 1) / FR_TILE;
    if (blockIdx.x >= nblk) return;
    fr_stage_tables(&T, S.hb);
    const StageElems P = W.el[prev];
    const unsigned n_orb = T.n_orb;
    size_t base = (size_t)blockIdx.x * FR_TILE + threadIdx.x;      // lane-contiguous: coalesced loads and stores
    uint32_t cnt = 0;
    // The single excitations (a few per wave) cost a loop over the occupied orbitals each -- the virtual orbital from its index, the allowed
    // occupied orbitals, the one-body element with its 2 n_elec integral reads -- and where they stand each holds its wave while the other lanes have
    // long finished their doubles.  They are listed in LDS and evaluated side by side afterwards, one lane each.
    __shared__ uint32_t sh_ns, sh_sing[FR_TILE];
    if (threadIdx.x == 0) sh_ns = 0;
    __syncthreads();
    auto single = [&](size_t e) {
        const uint32_t wi = W.e_wi[e];
        const double val = W.e_val[e];
        const uint32_t pos = P.pos[wi], pc = P.code[wi];
        const det_t det = V.dets[pos];
        double el = 0;
        uint32_t orbs = 0;
        unsigned o1 = fr_nth_bit(det, fr_c(pc, 1));
        unsigned u1 = fr_virt_from_idx(T, det, T.irrep[o1 % n_orb], n_orb * (o1 / n_orb), fr_c(pc, 2));
        if (u1 != 255) {
            orbs = fr_code(o1, u1, 0, 0);
            unsigned n_occ = fr_count_sing_allowed(T, det);
            el = unit_matrel ? 1.0 : fr_sing_matrel(det, o1, u1, S.h_core, S.eris, n_orb);
            el *= val / (1 - p_doub) * n_occ * fr_c(pc, 3);
            if (fabs(el) > 1e-9) el *= fr_sing_parity(det, o1, u1);
            else el = 0;
        }
        f_val[e] = el; f_orbs[e] = orbs;
        cnt += (el != 0);
    };
    for (int it = 0; it < FR_ITEMS; it++) {
        size_t e = base + (size_t)it * FR_BLOCK;
        if (e >= n_in) break;
        uint32_t wi = W.e_wi[e], sub = W.e_sub[e];
        double val = W.e_val[e];
        uint32_t pos = P.pos[wi], pc = P.code[wi];
        if (fr_c(pc, 0) != 0) { sh_sing[atomicAdd(&sh_ns, 1u)] = (uint32_t)(e - (size_t)blockIdx.x * FR_TILE); continue; }
        det_t det = V.dets[pos];
        unsigned o1_idx = fr_c(pc, 1);
        double el = 0;
        uint32_t orbs = 0;
        if (fr_c(pc, 0) == 0) {
            unsigned o1 = fr_nth_bit(det, o1_idx), o2 = fr_nth_bit(det, fr_c(pc, 2)), u1 = fr_c(pc, 3);
            unsigned ir = T.irrep[o1 % n_orb] ^ T.irrep[o2 % n_orb] ^ T.irrep[u1 % n_orb];
            unsigned u2 = T.lookup[ir][sub + 1] + n_orb * (o2 / n_orb);
            if (!fr_bit(det, u2) && u1 != u2) {
                if (u1 > u2) { unsigned t = u1; u1 = u2; u2 = t; }
                if (o1 > o2) { unsigned t = o1; o1 = o2; o2 = t; }
                orbs = fr_code(o1, o2, u1, u2);
                double tw = NEW_HB ? fr_unnorm_wt(T, o1, o2, u1, u2) : fr_norm_wt(T, det, o1, o2, u1, u2);
                double mel = unit_matrel ? 1.0 : fr_doub_matrel(o1, o2, u1, u2, S.eris, n_orb);
                el = mel * val / tw / p_doub;
                if (fabs(el) > 1e-9) el *= fr_doub_parity(det, o1, o2, u1, u2);
                else el = 0;
            }
        }
        f_val[e] = el; f_orbs[e] = orbs;
        cnt += (el != 0);
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < sh_ns; k += FR_BLOCK) single((size_t)blockIdx.x * FR_TILE + sh_sing[k]);
    uint32_t bc = fr_block_sum_u32(cnt, shu);
    if (threadIdx.x == 0) pcnt[blockIdx.x] = bc;
}

__global__ void __launch_bounds__(FR_BLOCK) k_final_compact(CompWork W, int prev, const double *f_val, const uint32_t *f_orbs, const uint32_t *pcnt,
                                                            uint32_t *c_pos, uint32_t *c_orbs, double *c_val, uint32_t *n_succ) {
    __shared__ uint32_t shu[4];
    const unsigned n_in = W.state[FR_MAX_ROUNDS + 1].n_out;
    const unsigned nblk = (n_in + FR_TILE - 1) / FR_TILE;
    if (blockIdx.x >= nblk) { if (blockIdx.x == 0 && threadIdx.x == 0) *n_succ = 0; return; }
    uint32_t off;
    { uint32_t x = 0; for (unsigned i = threadIdx.x; i < blockIdx.x; i += blockDim.x) x += pcnt[i]; off = fr_block_sum_u32(x, shu); }
    const StageElems P = W.el[prev];
    size_t base = (size_t)blockIdx.x * FR_TILE + (size_t)threadIdx.x * FR_ITEMS;
    uint32_t f[FR_ITEMS], tsum = 0;
#pragma unroll
    for (int it = 0; it < FR_ITEMS; it++) { size_t e = base + it; f[it] = (e < n_in && f_val[e] != 0) ? 1u : 0u; tsum += f[it]; }
    uint32_t tot;
    uint32_t incl = fr_block_scan_u32(tsum, shu, &tot);
    uint32_t o = off + incl - tsum;
#pragma unroll
    for (int it = 0; it < FR_ITEMS; it++) {
        size_t e = base + it;
        if (f[it]) { c_pos[o] = P.pos[W.e_wi[e]]; c_orbs[o] = f_orbs[e]; c_val[o] = f_val[e]; o++; }
    }
    if (blockIdx.x == nblk - 1 && threadIdx.x == FR_BLOCK - 1) *n_succ = o;
}

// find_keep_sub's epilogue (compress_utils.cpp:266-275), seed_sys and the comb, from the settled replay
// norms: every rank's find_keep_sub result in rank order (compress_utils.cpp:817-818); norms[rank] == *W.seq.total
static __global__ void __launch_bounds__(64) k_comp_finalize2(CompWork W, Fks2Work F, double rn, const double *norms, int rank, int n_ranks) {
    __shared__ Teeth Tsh;           // the comb is tabulated in LDS by one lane and copied out by the wave
    if (W.seq.skip && *W.seq.skip) return;
    if (threadIdx.x == 0) {
        const FksScal *S = F.scal;
        CompState s = W.state[0];
        double G = S->G_last;
        uint32_t n_rem = S->n_last;
        s.n_pass = S->n_pass;
        double loc_norm = 0;
        if (G / n_rem < 1e-8) n_rem = 0;
        else loc_norm = *W.seq.total;
        s.n_rem = n_rem; s.loc_norm = loc_norm; s.G = G; s.pbuf = 0;
        // seed_sys (compress_utils.cpp:113-120): lbound over the ranks before me, then the rest on top of it, in rank order.
        // n_rem == 0 is decided from replicated scalars, so every rank then reports a zero norm.
        double lbound0 = 0;
        for (int p = 0; p < rank; p++) lbound0 += n_rem ? norms[p] : 0.0;
        double glob = lbound0;
        for (int p = rank; p < n_ranks; p++) glob += (p == rank) ? loc_norm : (n_rem ? norms[p] : 0.0);
        double unit = 0, r0 = INFINITY;
        if (n_rem > 0) r0 = fr_seed_sys(rn, lbound0, glob, n_rem, &unit);
        s.unit = glob / n_rem;
        s.n_out = 0; s.n_fix = 0;
        W.state[FR_MAX_ROUNDS + 1] = s;
        if (n_rem > 0) fr_build_teeth(&Tsh, r0, unit, n_rem + 2, lbound0);
        else { Tsh.nseg = 0; Tsh.kmax = 0; Tsh.unit = 0; Tsh.lbound0 = lbound0; }
    }
    __syncthreads();
    const uint32_t ndw = (uint32_t)((offsetof(Teeth, seg) + (size_t)Tsh.nseg * sizeof(TeethSeg)) / 4);
    const uint32_t *src = (const uint32_t *)&Tsh;
    uint32_t *dst = (uint32_t *)W.teeth;
    for (uint32_t i = threadIdx.x; i < ndw; i += blockDim.x) dst[i] = src[i];
}

// this rank's find_keep_sub result, as the reference returns it (0 when the budget is spent, compress_utils.cpp:267-269)
static __global__ void k_put_norm(CompWork W, Fks2Work F, double *out) {
    const FksScal *S = F.scal;
    *out = (S->G_last / S->n_last < 1e-8) ? 0.0 : *W.seq.total;
}
// copies the gathered norms out of the staging block (the next collective overwrites it)
// (all zero when no samples are left: sys_sub then starts every rank's lbound at 0, compress_utils.cpp:720-726)
static __global__ void k_keep_norms(const double *norms, int n, double *keep, CompWork W, Fks2Work F) {
    const bool any = W.state[FR_MAX_ROUNDS + 1].n_rem > 0;
    for (int p = threadIdx.x; p < n; p += blockDim.x) keep[p] = any ? norms[p] : 0.0;
}

// ------------------------------------------------------------------ host orchestration
#ifdef FR_SYS_TIMING
// where a workgroup of k_sys_count / k_sys_write spends its time (wall_clock64: 100 MHz), and how the workgroups are spread over the launch
static void fr_sys_timing_dump(FriesCtx *c, const char *name, int stage, unsigned grid, int n_stamp) {
    static int calls = 0;
    if (!c->W.tdbg) return;
    calls++;
    if (calls < 200 || calls > 220) return;
    FR_HIP(hipStreamSynchronize(c->stream));
    const unsigned nb = grid < 8192 ? grid : 8192;
    std::vector<unsigned long long> h((size_t)nb * 8);
    FR_HIP(hipMemcpy(h.data(), c->W.tdbg, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0;
    for (unsigned b = 0; b < nb; b++) { if (h[b * 8] && h[b * 8] < t0) t0 = h[b * 8]; if (h[b * 8 + n_stamp - 1] > t1) t1 = h[b * 8 + n_stamp - 1]; }
    double ph[8] = {0}; unsigned cnt = 0;
    double start_hist[8] = {0};
    for (unsigned b = 0; b < nb; b++) {
        if (!h[b * 8]) continue;
        cnt++;
        for (int k = 1; k < n_stamp; k++) ph[k] += (double)(h[b * 8 + k] - h[b * 8 + k - 1]) * 0.01;
        const double rel = (double)(h[b * 8] - t0) / (double)(t1 - t0 + 1);
        start_hist[(int)(rel * 8)]++;
    }
    fprintf(stderr, "[%s stage %d] %u workgroups, span %.1f us; mean us per phase:", name, stage, cnt, (double)(t1 - t0) * 0.01);
    for (int k = 1; k < n_stamp; k++) fprintf(stderr, " %.2f", ph[k] / (cnt ? cnt : 1));
    if (c->tile_dirty_mem) { std::vector<uint8_t> td(nb); std::vector<uint32_t> sc(nb); FR_HIP(hipMemcpy(td.data(), c->tile_dirty_mem, nb, hipMemcpyDeviceToHost)); FR_HIP(hipMemcpy(sc.data(), c->spill_cnt_mem, 4 * (size_t)nb, hipMemcpyDeviceToHost));
        unsigned nd = 0; unsigned long long ns = 0; uint32_t mx = 0; for (unsigned b = 0; b < nb; b++) { nd += td[b]; ns += sc[b]; mx = sc[b] > mx ? sc[b] : mx; }
        fprintf(stderr, " | dirty tiles %u, spills per tile %.1f (max %u)", nd, (double)ns / nb, mx); }
    fprintf(stderr, " | starts by eighth of the span:");
    for (int k = 0; k < 8; k++) fprintf(stderr, " %.0f", start_hist[k]);
    fprintf(stderr, "\n");
    FR_HIP(hipMemset(c->W.tdbg, 0, h.size() * 8));
}
#endif
void fr_hbpp_alloc(FriesCtx *c, uint32_t cap) {
    CompWork &W = c->W;
    W.cap = cap;
    for (int h = 0; h < 2; h++) {
        W.el[h].val = fr_alloc<double>(cap); W.el[h].pos = fr_alloc<uint32_t>(cap); W.el[h].code = fr_alloc<uint32_t>(cap);
        W.el[h].ndiv = fr_alloc<uint32_t>(cap); W.el[h].nsub = fr_alloc<uint32_t>(cap);
        W.el[h].rinv = fr_alloc<double>(cap); W.el[h].raux = fr_alloc<uint32_t>(cap); W.el[h].det = fr_alloc<det_t>(cap);
        W.psum[h] = fr_alloc<double>(FR_MAX_PART); W.pcnt[h] = fr_alloc<uint32_t>(FR_MAX_PART);
    }
    W.wt_remain = fr_alloc<double>(cap); W.keep = fr_alloc<uint32_t>(cap); W.S = fr_alloc<double>(cap);
    W.kin = fr_alloc<uint32_t>(cap); W.cnt = fr_alloc<uint32_t>(cap);
    W.kend = nullptr; W.act[0] = W.act[1] = nullptr; W.act_n = nullptr; W.kstart = nullptr; W.prop = 0;
    W.e_wi = fr_alloc<uint32_t>(cap); W.e_sub = fr_alloc<uint32_t>(cap); W.e_val = fr_alloc<double>(cap);
    W.state = fr_alloc<CompState>(FR_MAX_ROUNDS + 2);
    W.stg = nullptr; W.spill = nullptr; W.spill_cnt = nullptr; W.tile_dirty = nullptr;
    if (!(getenv("FRIES_NO_STAGING") && atoi(getenv("FRIES_NO_STAGING")))) {       // emissions staged by k_sys_count (comp_kernels.hpp); FRIES_NO_STAGING=1: k_sys_write evaluates the rows again
        const size_t ntile = fr_blocks(cap, FR_TILE) + 1;
        c->stg_mem = fr_alloc<uint4>(ntile * FR_BLOCK * FR_STG_SLOTS); c->spill_mem = fr_alloc<uint4>(ntile * FR_STG_SPILL);
        c->spill_cnt_mem = fr_alloc<uint32_t>(ntile); c->tile_dirty_mem = fr_alloc<uint8_t>(ntile);
    }
    W.teeth = fr_alloc<Teeth>(1);
    W.fix_list = fr_alloc<uint32_t>(FR_MAX_FIX);
    W.seq.tiles = fr_alloc<SeqRec>(FR_MAX_PART); W.seq.subs = fr_alloc<SeqRec>((size_t)FR_MAX_PART * FR_SUBS_PER_TILE); W.seq.total = fr_alloc<double>(1); W.seq.tsum = fr_alloc<double>(FR_MAX_PART);
#ifdef FR_SEQ_TIMING
    W.seq.dbg = getenv("FRIES_SEQ_DBG") ? 1 : 0;
#endif
#ifdef FR_SYS_TIMING
    W.tdbg = getenv("FRIES_SYS_DBG") ? fr_alloc<unsigned long long>(8192 * 8) : nullptr;
#endif
    {
        Fks2Work &F = c->F2;
        F.nb8_cap = (uint32_t)(((size_t)cap / 8 + 2 + FR_FKS_CHUNK - 1) / FR_FKS_CHUNK * FR_FKS_CHUNK);     // whole chunks: k_fks_scan uses unguarded vector loads
        size_t n8 = (size_t)FR_FKS_PMAX * F.nb8_cap;
        F.dk8 = fr_alloc<uint32_t>(n8); F.dg8 = fr_alloc<double>(n8); F.ws8 = fr_alloc<double>(n8);
        F.cdirty = fr_alloc<uint32_t>(FR_FKS_MAXCHUNK);
        {   // the replay loop's readback block: pinned, host-coherent, mapped into the device's address space
            void *hp = nullptr, *dp = nullptr;
            FR_HIP(hipHostMalloc(&hp, sizeof(FksHost), hipHostMallocMapped | hipHostMallocCoherent));
            for (size_t q = 0; q < sizeof(FksHost); q++) ((volatile uint8_t *)hp)[q] = 0;
            FR_HIP(hipHostGetDevicePointer(&dp, hp, 0));
            c->h_fks = (FksHost *)hp; F.hm = (FksHost *)dp;
        }
        F.nwv_cap = F.nb8_cap / 8;
        const size_t nw = (size_t)FR_FKS_PMAX * F.nwv_cap;
        F.wrec = fr_alloc<FksWRec>(nw); F.wNp = fr_alloc<uint32_t>(F.nwv_cap);
        FR_HIP(hipMemset(F.wNp, 0xff, 4 * (size_t)F.nwv_cap));
        F.xk8 = fr_alloc<uint32_t>(n8); F.xg8 = fr_alloc<double>(n8);
        F.scal = fr_alloc<FksScal>(1); F.hist = fr_alloc<uint32_t>(FR_MAX_ROUNDS + 2);
        F.ck = fr_alloc<uint32_t>((size_t)FR_FKS_PMAX * FR_FKS_MAXCHUNK); F.cg = fr_alloc<double>((size_t)FR_FKS_PMAX * FR_FKS_MAXCHUNK); F.cw = fr_alloc<double>((size_t)FR_FKS_PMAX * FR_FKS_MAXCHUNK);
        F.ckx = fr_alloc<uint32_t>((size_t)FR_FKS_PMAX * FR_FKS_MAXCHUNK); F.cgx = fr_alloc<double>((size_t)FR_FKS_PMAX * FR_FKS_MAXCHUNK);
        c->fks_wkx = fr_alloc<uint32_t>((size_t)8 * FR_FKS_PMAX * FR_FKS_MAXCHUNK); c->fks_wgx = fr_alloc<double>((size_t)8 * FR_FKS_PMAX * FR_FKS_MAXCHUNK);
        FR_HIP(hipMemset(F.hist, 0, 4 * (FR_MAX_ROUNDS + 2)));
        F.dbg_cnt = fr_alloc<uint32_t>((size_t)FR_MAX_ROUNDS * 4);
        FR_HIP(hipMemset(F.dbg_cnt, 0, (size_t)FR_MAX_ROUNDS * 16));
        c->fks_sxk8 = fr_alloc<uint32_t>((size_t)6 * FR_FKS_SROWS * F.nb8_cap); c->fks_sxg8 = fr_alloc<double>((size_t)6 * FR_FKS_SROWS * F.nb8_cap);
        c->fks_saved = fr_alloc<FksSaved>(8);
        FR_HIP(hipMemset(c->fks_saved, 0, 8 * sizeof(FksSaved)));
        {
            const float ex = getenv("FRIES_FKS_WARM_EXTRAP") ? (float)atof(getenv("FRIES_FKS_WARM_EXTRAP")) : FR_FKS_WARM_EXTRAP;
            if (ex > 0.0f) { FksSaved hsv{}; hsv.extrap = ex; for (int k = 0; k < 8; k++) FR_HIP(hipMemcpy(c->fks_saved + k, &hsv, sizeof(FksSaved), hipMemcpyHostToDevice)); }
        }
        c->fks_wk = fr_alloc<uint32_t>((size_t)8 * FR_FKS_PMAX * FR_FKS_MAXCHUNK); c->fks_wg = fr_alloc<double>((size_t)8 * FR_FKS_PMAX * FR_FKS_MAXCHUNK);
        FR_HIP(hipMemset(F.scal, 0, sizeof(FksScal)));
    }
    c->d_norms_keep = fr_alloc<double>(FR_MAX_RANKS); c->d_seq_scratch = fr_alloc<double>(1); c->d_norms_all = fr_alloc<double>(FR_MAX_RANKS);
    c->fks_seq = fr_alloc<FksSeq>(1);
    {   // work arrays of the parallel form of the in-order sweep (fks_seq.hpp)
        FksSq &SQ = c->fsq;
        const size_t nt = fr_blocks(cap, FR_SQ_TILE) + 1, ne = nt * FR_SQ_TILE, nb = nt * 32;
        SQ.dl = fr_alloc<double>(ne); SQ.nwr = fr_alloc<double>(ne); SQ.nkp = fr_alloc<uint32_t>(ne);
        SQ.gb = fr_alloc<double>(nb); SQ.lb = fr_alloc<double>(nb); SQ.dgb = fr_alloc<double>(nb); SQ.kb = fr_alloc<uint32_t>(nb); SQ.dk = fr_alloc<uint32_t>(nb);
        SQ.tk = fr_alloc<uint32_t>(nt); SQ.tkx = fr_alloc<uint32_t>(nt); SQ.tg = fr_alloc<double>(nt); SQ.tgx = fr_alloc<double>(nt); SQ.tany = fr_alloc<uint8_t>(nt);
        SQ.ctl = fr_alloc<FksSqCtl>(1);
        FR_HIP(hipMemset(SQ.ctl, 0, sizeof(FksSqCtl)));
        SQ.mG = fr_alloc<double>(nt); SQ.mL = fr_alloc<double>(nt); SQ.sabs = fr_alloc<double>(nt); SQ.gt = fr_alloc<double>(nt); SQ.lt = fr_alloc<double>(nt);
        SQ.eG = fr_alloc<int32_t>(nt); SQ.eL = fr_alloc<int32_t>(nt); SQ.mflag = fr_alloc<uint8_t>(nt); SQ.fast = fr_alloc<uint8_t>(nt);
        c->fsq_check = getenv("FRIES_FSQ_CHECK") && atoi(getenv("FRIES_FSQ_CHECK"));
        if (getenv("FRIES_FSQ_MAPS")) c->fsq_use_maps = atoi(getenv("FRIES_FSQ_MAPS")) != 0;
        SQ.gb2 = c->fsq_check ? fr_alloc<double>(nb) : nullptr; SQ.lb2 = c->fsq_check ? fr_alloc<double>(nb) : nullptr;
        c->fsq_walk_only = getenv("FRIES_FKS_SEQ_WALK") && atoi(getenv("FRIES_FKS_SEQ_WALK"));
        if (getenv("FRIES_FSQ_GUESS_ROUNDS")) c->fsq_guess_rounds = atoi(getenv("FRIES_FSQ_GUESS_ROUNDS"));
        if (getenv("FRIES_FSQ_EXACT_ROUNDS")) c->fsq_exact_rounds = atoi(getenv("FRIES_FSQ_EXACT_ROUNDS"));
        if (getenv("FRIES_FSQ_SPARSE_MAX")) c->fsq_sparse_max = atoi(getenv("FRIES_FSQ_SPARSE_MAX"));
    }
    c->c_pos = fr_alloc<uint32_t>(cap); c->c_orbs = fr_alloc<uint32_t>(cap); c->c_val = fr_alloc<double>(cap);
    c->d_nsucc = fr_alloc<uint32_t>(1);
    FR_HIP(hipMemset(W.state, 0, sizeof(CompState) * (FR_MAX_ROUNDS + 2)));
    if (fr_blocks(cap, FR_TILE) > FR_MAX_PART) throw FriesError("work capacity exceeds FR_MAX_PART tiles");
}

// Stage 2 of frisys_hh (frisys_hh.cpp:209-220): element e comes from emission e of stage 1; sub 0 = electron hop (uniform over
// the possible hops), sub 1 = phonon move (uniform over 2 n_elec choices); the value is multiplied by the subdivision count.
__global__ void __launch_bounds__(FR_BLOCK) k_prep_hh2(CompWork W, VecDev V, int cur, uint32_t n_samp) {
    __shared__ double shd[12];
    const unsigned n_in = W.state[FR_MAX_ROUNDS + 1].n_out;
    const unsigned nblk = (n_in + FR_TILE - 1) / FR_TILE;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        CompState s{};
        s.n_rem = n_samp; s.n_in = n_in; s.done = 0; s.pbuf = 0;
        W.state[0] = s;
    }
    if (blockIdx.x >= nblk) return;
    const StageElems E = W.el[cur], P = W.el[cur ^ 1];
    const unsigned L = V.hh_sites;
    size_t base = (size_t)blockIdx.x * FR_TILE + threadIdx.x;
    double sum = 0;
    for (int it = 0; it < FR_ITEMS; it++) {
        size_t e = base + (size_t)it * FR_BLOCK;
        if (e >= n_in) break;
        uint32_t wi = W.e_wi[e], sub = W.e_sub[e];
        double val = W.e_val[e];
        uint32_t pos = P.pos[wi];
        uint32_t ndiv;
        if (sub) ndiv = 2 * V.hh_nelec;
        else {
            const det_t El = V.dets[pos] & ((1ull << (2 * L)) - 1ull);
            det_t r0 = El & ~(El >> 1); r0 &= ~(1ull << (L - 1)); r0 &= ~(1ull << (2 * L - 1));      // hh_vec.hpp:139-175
            det_t r1 = El & (~El << 1); r1 &= ~(1ull << L);
            ndiv = (uint32_t)__popcll(r0) + (uint32_t)__popcll(r1);
        }
        val *= ndiv;
        E.val[e] = val; E.pos[e] = pos; E.code[e] = sub; E.ndiv[e] = ndiv; E.nsub[e] = 0; E.rinv[e] = 1.0; E.raux[e] = 0;
        W.wt_remain[e] = val; W.keep[e] = 0;
        sum += val;
    }
    double bs;
    fr_block_excl_f64(sum, shd, &bs);
    if (threadIdx.x == 0) { W.psum[0][blockIdx.x] = bs; W.pcnt[0][blockIdx.x] = 0; }
}

// drops some flags of the device error word and leaves the others standing
static __global__ void k_err_clear(uint32_t *err, uint32_t bits) { atomicAnd(err, ~bits); }

// find_keep_sub of this stage in the reference's order (fks_seq.hpp): sweeps driven from the host, one sum_mpi before and one
// after each, as in compress_utils.cpp:153-265
// one sweep of it in the parallel form (fks_seq.hpp, second half): guesses settled with tree-summed norms, then the exact chain and
// the comparison against it, repeated from the first tile that differs; the one-wave walk from there if that does not close
template <int STAGE, bool NEW_HB>
static void run_fsq_sweep(FriesCtx *c, int cur, uint32_t n_tiles) {
    CompWork &W = c->W;
    hipStream_t st = c->stream;
    FksSeq *Q = c->fks_seq;
    FksSq &SQ = c->fsq;
    auto first_changed = [&]() {
        uint32_t fc;
        FR_HIP(hipMemcpyAsync(&fc, &SQ.ctl->first_changed, 4, hipMemcpyDeviceToHost, st));
        FR_HIP(hipStreamSynchronize(st));
        if (fc != FR_SQ_INF) FR_HIP(hipMemsetAsync(&SQ.ctl->first_changed, 0xff, 4, st));
        return fc;
    };
    auto spec = [&](uint32_t from) { FR_LAUNCH(c, "k_fsq_spec", (k_fsq_spec<STAGE, NEW_HB>), dim3(n_tiles - from), dim3(FR_BLOCK), W, c->d_hb, cur, c->p_doub, Q, SQ, from); };
    auto prefixes = [&](uint32_t from, int approx) {
        FR_LAUNCH(c, "k_fsq_scan", k_fsq_scan, dim3(1), dim3(1024), SQ, from, n_tiles);
        FR_LAUNCH(c, "k_fsq_expand", k_fsq_expand, dim3(fr_blocks(n_tiles - from, FR_BLOCK / 32)), dim3(FR_BLOCK), SQ, from, n_tiles, approx);
    };
    FR_LAUNCH(c, "k_fsq_init", k_fsq_init, dim3(n_tiles), dim3(FR_BLOCK), W, Q, SQ);
    uint32_t from = 0, fc = 0;
    int n_guess = 0;
    for (int r = 0; r < c->fsq_guess_rounds; r++) {
        spec(from);
        c->n_fsq_guess++; n_guess++;
        fc = first_changed();
        if (fc == FR_SQ_INF) {
            if (r == 0) {
                // Nothing moves under (the sweep's start norm, no sample taken) -- and with nothing moving that IS every block's entry
                // state: the sweep is settled (the two closing sweeps of every stage end here)
                FR_LAUNCH(c, "k_fsq_commit", k_fsq_commit, dim3(n_tiles), dim3(FR_BLOCK), W, Q, SQ, n_tiles, 1);
                if (c->dbg >= 2) fprintf(stderr, "[fries]   in-order sweep: %u tiles, untouched\n", n_tiles);
                return;
            }
            break;
        }
        from = fc / 32;
        prefixes(from, 1);
    }
    from = 0;
    for (int x = 0; ; x++) {
        if (c->fsq_use_maps) FR_LAUNCH(c, "k_fsq_maps", k_fsq_maps<0>, dim3(n_tiles - from), dim3(FR_BLOCK), Q, SQ, from);
        FR_LAUNCH(c, "k_fsq_chain", k_fsq_chain, dim3(1), dim3(64), Q, SQ, SQ.gb, SQ.lb, from, n_tiles, c->fsq_sparse_max, c->fsq_use_maps ? 1 : 0);
        FR_LAUNCH(c, "k_fsq_entries", k_fsq_maps<1>, dim3(n_tiles - from), dim3(FR_BLOCK), Q, SQ, from);
        if (c->fsq_check) {         // the same chain element by element into a second pair of arrays; every entry must agree bit for bit
            FR_LAUNCH(c, "k_fsq_chain", k_fsq_chain, dim3(1), dim3(64), Q, SQ, SQ.gb2, SQ.lb2, from, n_tiles, c->fsq_sparse_max, 0);
            FksSq S2 = SQ; S2.gb = SQ.gb2; S2.lb = SQ.lb2;
            FR_LAUNCH(c, "k_fsq_entries", k_fsq_maps<1>, dim3(n_tiles - from), dim3(FR_BLOCK), Q, S2, from);
            FR_LAUNCH(c, "k_fsq_compare", k_fsq_compare, dim3(fr_blocks((n_tiles - from) * 32, FR_BLOCK)), dim3(FR_BLOCK), SQ, from, n_tiles);
            uint32_t bad = 0;
            FR_HIP(hipMemcpyAsync(&bad, &SQ.ctl->n_mismatch, 4, hipMemcpyDeviceToHost, st));
            FR_HIP(hipStreamSynchronize(st));
            if (bad) throw FriesError("FRIES_FSQ_CHECK: the integer form of the chain differs from the element-by-element chain in " + std::to_string(bad) + " block entries");
        }
        spec(from);
        c->n_fsq_exact++; c->n_fsq_chain_tiles += n_tiles - from;
        if (c->dbg >= 2) fprintf(stderr, "[fries]   in-order sweep: %u tiles, %d guess rounds, exact round %d from tile %u\n", n_tiles, n_guess, x, from);
        fc = first_changed();
        if (fc == FR_SQ_INF) { FR_LAUNCH(c, "k_fsq_commit", k_fsq_commit, dim3(n_tiles), dim3(FR_BLOCK), W, Q, SQ, n_tiles, 1); return; }
        from = fc / 32;
        prefixes(from, 0);
        if (x >= c->fsq_exact_rounds) break;
    }
    // not closed: blocks before `from` are final; the walk goes on from there with the state the last chain and scan left at that tile
    FR_LAUNCH(c, "k_fsq_commit", k_fsq_commit, dim3(n_tiles), dim3(FR_BLOCK), W, Q, SQ, from, 0);
    FR_LAUNCH(c, "k_fks_seq_sweep", (k_fks_seq_sweep<STAGE, NEW_HB>), dim3(1), dim3(64), W, c->d_hb, cur, c->p_doub, Q, from, SQ.gb, SQ.lb, SQ.kb);
    c->n_fsq_walk++; c->n_fsq_walk_tiles += n_tiles - from;
}

template <int STAGE, bool NEW_HB>
static void run_fks_sequential(FriesCtx *c, int cur, unsigned grid, Fks2Work F, uint32_t n_bound) {
    CompWork &W = c->W;
    hipStream_t st = c->stream;
    const int P = c->n_ranks;
    FksSeq *Q = c->fks_seq;
    uint32_t n_tiles = fr_blocks(n_bound, FR_SQ_TILE);
    if (n_tiles == 0) n_tiles = 1;
    FR_LAUNCH(c, "k_fks_seq_reset", k_fks_seq_reset, dim3(grid < 1024 ? grid : 1024), dim3(FR_BLOCK), W, cur);
    AccVal av{W.el[cur].val, &W.state[0]};
    FR_LAUNCH(c, "k_seq_sums", (k_seq_sums<AccVal>), dim3(grid), dim3(FR_BLOCK), W.seq, av);
    FR_LAUNCH(c, "k_seq_maps", (k_seq_maps<AccVal>), dim3(grid), dim3(FR_BLOCK), W.seq, av, fr_seq_from_zero());
    FR_LAUNCH(c, "k_seq_chain", (k_seq_chain<AccVal>), dim3(1), dim3(FR_BLOCK), W.seq, av, fr_seq_from_zero());
    FR_LAUNCH(c, "k_fks_seq_begin", k_fks_seq_begin, dim3(1), dim3(1), W, Q, W.seq.total, (double *)c->comm.small_send);
    AccWt acc{W.wt_remain, &W.state[0]};
    for (int sweep = 0; ; sweep++) {
        if (sweep > 4096) throw FriesError("sequential find_keep_sub did not terminate");
        const double *alln = (const double *)fr_allgather(c, sizeof(double));
        FR_LAUNCH(c, "k_fks_seq_norm", k_fks_seq_norm, dim3(1), dim3(1), Q, alln, P);
        if (c->fsq_walk_only) FR_LAUNCH(c, "k_fks_seq_sweep", (k_fks_seq_sweep<STAGE, NEW_HB>), dim3(1), dim3(64), W, c->d_hb, cur, c->p_doub, Q, 0u, (const double *)nullptr, (const double *)nullptr, (const uint32_t *)nullptr);
        else run_fsq_sweep<STAGE, NEW_HB>(c, cur, n_tiles);
        FR_LAUNCH(c, "k_fks_seq_put_k", k_fks_seq_put_k, dim3(1), dim3(1), Q, (uint32_t *)c->comm.small_send);
        const uint32_t *allk = (const uint32_t *)fr_allgather(c, sizeof(uint32_t));
        FR_LAUNCH(c, "k_fks_seq_post", k_fks_seq_post, dim3(1), dim3(1), Q, allk, P, (double *)c->comm.small_send);
        FksSeq h;
        FR_HIP(hipMemcpyAsync(&h, Q, sizeof(h), hipMemcpyDeviceToHost, st));
        FR_HIP(hipStreamSynchronize(st));
        if (!h.go) break;
        if (h.resum) {      // loc_one_norm re-summed from wt_remain, in order (compress_utils.cpp:258-264)
            FR_LAUNCH(c, "k_seq_sums", (k_seq_sums<AccWt>), dim3(grid), dim3(FR_BLOCK), W.seq, acc);
            FR_LAUNCH(c, "k_seq_maps", (k_seq_maps<AccWt>), dim3(grid), dim3(FR_BLOCK), W.seq, acc, fr_seq_from_zero());
            FR_LAUNCH(c, "k_seq_chain", (k_seq_chain<AccWt>), dim3(1), dim3(FR_BLOCK), W.seq, acc, fr_seq_from_zero());
            FR_LAUNCH(c, "k_fks_seq_take_resum", k_fks_seq_take_resum, dim3(1), dim3(1), Q, W.seq.total, (double *)c->comm.small_send);
        }
    }
    FR_LAUNCH(c, "k_fks_seq_end", k_fks_seq_end, dim3(1), dim3(1), Q, F);
    c->n_fks_sequential++;
}

template <int STAGE, bool NEW_HB>
static void run_stage(FriesCtx *c, int cur, uint32_t n_bound, uint32_t n_samp, double rn, int out_slot, bool hh_stage2 = false) {
    CompWork &W = c->W;
    hipStream_t st = c->stream;
    unsigned grid = fr_blocks(n_bound, FR_TILE);
    if (grid == 0) grid = 1;
    if (STAGE == 1) FR_LAUNCH(c, "k_prep1", k_prep1, dim3(grid), dim3(FR_BLOCK), W, c->vec, cur, n_samp);
    else if (hh_stage2) FR_LAUNCH(c, "k_prep_hh2", k_prep_hh2, dim3(grid), dim3(FR_BLOCK), W, c->vec, cur, n_samp);
    else FR_LAUNCH(c, "k_prep", (k_prep<STAGE, NEW_HB>), dim3(grid), dim3(FR_BLOCK), W, c->vec, c->d_hb, cur, n_samp, c->p_doub);
    const bool stage_emissions = c->stg_mem && !W.prop;
    W.stg = stage_emissions ? c->stg_mem : nullptr; W.spill = stage_emissions ? c->spill_mem : nullptr;
    W.spill_cnt = stage_emissions ? c->spill_cnt_mem : nullptr; W.tile_dirty = stage_emissions ? c->tile_dirty_mem : nullptr;
    Fks2Work F = c->F2;
    F.saved = c->fks_saved + STAGE; F.wk = c->fks_wk + (size_t)STAGE * FR_FKS_PMAX * FR_FKS_MAXCHUNK; F.wg = c->fks_wg + (size_t)STAGE * FR_FKS_PMAX * FR_FKS_MAXCHUNK;
    F.wkx = c->fks_wkx + (size_t)STAGE * FR_FKS_PMAX * FR_FKS_MAXCHUNK; F.wgx = c->fks_wgx + (size_t)STAGE * FR_FKS_PMAX * FR_FKS_MAXCHUNK;
    // the settled per-group prefixes of this stage in the previous iteration (stage 1: vector positions persist; later stages: the head of
    // the stage holds the children of the same heavy determinants in the same order, and that is where the prefixes are least linear)
    F.sxk8 = ((STAGE == 1 || c->fks_group_warm_all) && !c->fks_no_group_warm) ? c->fks_sxk8 + (size_t)STAGE * FR_FKS_SROWS * F.nb8_cap : nullptr;
    F.sxg8 = F.sxk8 ? c->fks_sxg8 + (size_t)STAGE * FR_FKS_SROWS * F.nb8_cap : nullptr;
    const int warm = c->warm_start ? 1 : 0;
    unsigned gridE = fr_blocks(((size_t)n_bound / 8 + 1) * 8, FR_BLOCK);
    const unsigned gridL = c->fks_light_full_grid ? gridE : (gridE > c->fks_grid ? c->fks_grid : gridE);      // light replays: one tile per workgroup (the test of a tile is two dependent rounds of loads; a workgroup striding over three tiles pays them three times before it may leave)
    const unsigned gridE0 = gridE > c->fks_grid0 ? c->fks_grid0 : gridE;        // the lean replay's own count
    if (gridE > c->fks_grid) gridE = c->fks_grid;          // persistent workgroups (5 per CU), each strides over the tiles
    unsigned nchunk = fr_blocks((size_t)n_bound / 8 + 1, FR_FKS_CHUNK);
    if (nchunk > FR_FKS_MAXCHUNK) throw FriesError("stage too large for the find_keep_sub scan");
    const int P = c->n_ranks;
    FksMsg *msg = (FksMsg *)c->comm.small_send;
    const bool xr = c->use_comm;        // totals travel through the all-gather (also with one rank, when a comm was given)
    F.hm_close = xr ? 0 : 1;
    FR_LAUNCH(c, "k_fks_init", k_fks_init, dim3(1), dim3(FR_BLOCK), W, F, msg, xr ? 0 : 1, warm);
    if (xr) {
        const FksMsg *all = (const FksMsg *)fr_allgather(c, sizeof(FksMsg));
        FR_LAUNCH(c, "k_fks_passes", k_fks_passes, dim3(1), dim3(1), F, all, P, -1, c->d_err, warm);
    }
    // Replays are enqueued in batches (a host round trip costs a fraction of a replay); the per-replay "changed" flags tell
    // afterwards which replay was the first to reproduce its predecessor, and that count is the next iteration's first batch.
    int it = 0, batch = c->rounds_hint[STAGE], needed = 0;
    FksScal hscal{};
    bool sequential = c->fks_force_seq;
    bool closed = false;        // the closing pass has settled the stage and written the final wt_remain
    // replay 0 evaluates every tile; later replays only the tiles whose inputs moved by more than their tightest comparison allows
    // The replay needs about as many rounds as the reference runs sweeps (a sweep's start state is only right once the sweep
    // before it is): the early ones run lean (no margins, counted as changed), the round before the expected end records the
    // tiles' margins, and from then on only tiles whose inputs moved beyond their margin are evaluated.
    const int rec_at = c->fks_rec_at >= 1 && !c->fks_no_light ? c->fks_rec_at : 1;
    auto scan_totals = [&](int k) {
        FR_LAUNCH(c, "k_fks_scan", k_fks_scan, dim3(nchunk, 8), dim3(FR_BLOCK), F, k, k >= rec_at ? 1 : 0, c->fks_fuse_totals ? 1 : 0, c->d_err, msg, xr ? 0 : 1);
        if (!c->fks_fuse_totals) FR_LAUNCH(c, "k_fks_totals", k_fks_totals, dim3(1), dim3(FR_FKS_TOTALS_THREADS), F, c->d_err, msg, xr ? 0 : 1, k);
        if (xr) {
            const FksMsg *all = (const FksMsg *)fr_allgather(c, sizeof(FksMsg));
            FR_LAUNCH(c, "k_fks_passes", k_fks_passes, dim3(1), dim3(1), F, all, P, k, c->d_err, 0);
        }
    };
    auto replay = [&](int k) {
        const int light = (k > rec_at && !c->fks_no_light && !c->d_tie) ? (c->fks_no_ext ? 2 : 1) : 0;       // (tie statistics: every wave decides in every replay, so that the records are those of the settled state)       // FRIES_FKS_NO_LIGHT: every tile evaluated in every replay
        if (k < rec_at) FR_LAUNCH(c, "k_fks_sweep", (k_fks_sweep<STAGE, NEW_HB, 0>), dim3(gridE0), dim3(FR_BLOCK), W, F, c->d_hb, cur, k, c->p_doub, 0, c->dbg);
        else if (light) FR_LAUNCH(c, "k_fks_sweep_light", (k_fks_sweep<STAGE, NEW_HB, 3>), dim3(gridL), dim3(FR_BLOCK), W, F, c->d_hb, cur, k, c->p_doub, light, c->dbg);
        else FR_LAUNCH(c, "k_fks_sweep_rec", (k_fks_sweep<STAGE, NEW_HB, 1>), dim3(gridE), dim3(FR_BLOCK), W, F, c->d_hb, cur, k, c->p_doub, 0, c->dbg);
        scan_totals(k);
    };
    auto read_host = [&]() {
        const volatile FksHost *hm = c->h_fks;
        hscal.overflow = hm->overflow; hscal.G_last = hm->G_last; hscal.psG[0] = hm->psG0; hscal.G_neg = hm->G_neg; hscal.n_pass = hm->n_pass;
        if (hscal.overflow) sequential = true;
    };
    // The closing pass (k_fks_sweep MODE 4): the light replay that is expected to change nothing and the final pass in one launch.  When it
    // changes no delta the stage has settled and its output stands -- the confirming sweep + scan + totals are never launched.
    const bool closing = !c->fks_no_light && !c->d_tie && !c->fks_no_closing && !c->fks_fuse_totals && rec_at == 1 && !sequential;
    // what follows a settled stage up to the emission counts, for one rank without the propagation repair (the other cases exchange norms or
    // look at list lengths on the host in between): skip != nullptr = launched ahead of the host's look at *skip
    AccWt acc{W.wt_remain, &W.state[0]};
    auto launch_tail = [&](const uint32_t *skip, uint32_t ticket = 0u) {
        CompWork Wt = W;
        Wt.seq.skip = skip;
        SeqWork first = Wt.seq;
        if (ticket) { first.tk_word = c->d_misc(); first.tk = ticket; }      // the first kernel of the tail tells the host that the closing pass is through
        FR_LAUNCH(c, "k_seq_sums", (k_seq_sums<AccWt>), dim3(grid), dim3(FR_BLOCK), first, acc);
        FR_LAUNCH(c, "k_seq_maps", (k_seq_maps<AccWt>), dim3(grid), dim3(FR_BLOCK), Wt.seq, acc, fr_seq_from_zero());
        FR_LAUNCH(c, "k_seq_chain", (k_seq_chain<AccWt>), dim3(1), dim3(FR_BLOCK), Wt.seq, acc, fr_seq_from_zero());
        FR_LAUNCH(c, "k_comp_finalize", k_comp_finalize2, dim3(1), dim3(64), Wt, F, rn, (const double *)Wt.seq.total, c->rank, P);
        FR_LAUNCH(c, "k_sys_count", (k_sys_count<STAGE, NEW_HB>), dim3(grid), dim3(FR_BLOCK), Wt, c->vec, c->d_hb, cur, c->p_doub);
#ifdef FR_SYS_TIMING
        if (!skip) fr_sys_timing_dump(c, "k_sys_count", STAGE, grid, 6);
#endif
        FR_LAUNCH(c, "k_sys_fixup", (k_sys_fixup<STAGE, NEW_HB>), dim3(1), dim3(FR_BLOCK), Wt, c->vec, c->d_hb, cur, c->p_doub, c->d_err);
    };
    const bool simple_tail = !xr && c->rank == 0 && !W.prop;
    const bool speculate = simple_tail && !c->fks_no_speculation;
    bool tail_done = false;
    // ranks: the remaining norms already gathered with the closing pass's flag (FRIES_FKS_NO_MERGED_NORM=1: a message of their own, as before)
    const bool merge_norm = xr && !c->fks_no_merged_norm && !W.prop;
    bool norms_done = false;
    if (closing) {
        int plain = batch < 2 ? 2 : batch;       // replays before the closing pass (replay 1 writes the records the light test needs)
        while (!closed && !sequential) {
            if (plain > 47) { sequential = true; break; }       // does not settle (e.g. the reference's own 0/0 corner): walk the stage in order instead
            for (; it < plain; it++) replay(it);
            FR_LAUNCH(c, "k_fks_close", (k_fks_sweep<STAGE, NEW_HB, 4>), dim3(gridE), dim3(FR_BLOCK), W, F, c->d_hb, cur, it, c->p_doub, c->fks_no_ext ? 2 : 1, c->dbg);
            if (xr && merge_norm) {
                // every rank must see whether ANY rank changed a delta; the exact sum of this rank's wt_remain is formed at once (it leaves at once
                // if this rank's own flag is up) and travels with the flag: one all-gather per closing pass instead of two
                CompWork Wt = W;
                Wt.seq.skip = &F.hist[it];
                FR_LAUNCH(c, "k_seq_sums", (k_seq_sums<AccWt>), dim3(grid), dim3(FR_BLOCK), Wt.seq, acc);
                FR_LAUNCH(c, "k_seq_maps", (k_seq_maps<AccWt>), dim3(grid), dim3(FR_BLOCK), Wt.seq, acc, fr_seq_from_zero());
                FR_LAUNCH(c, "k_seq_chain", (k_seq_chain<AccWt>), dim3(1), dim3(FR_BLOCK), Wt.seq, acc, fr_seq_from_zero());
                FR_LAUNCH(c, "k_fks_close_put", k_fks_close_put2, dim3(1), dim3(1), F, (uint32_t *)c->comm.small_send, it, (const double *)W.seq.total);
                const uint32_t *all = (const uint32_t *)fr_allgather(c, 16);
                FR_LAUNCH(c, "k_fks_close_flag", k_fks_close_flag2, dim3(1), dim3(1), F, all, P, it, c->d_norms_all);
                norms_done = true;
            }
            else if (xr) {       // every rank must see whether ANY rank changed a delta
                FR_LAUNCH(c, "k_fks_close_put", k_fks_close_put, dim3(1), dim3(1), F, (uint32_t *)c->comm.small_send, it);
                const uint32_t *all = (const uint32_t *)fr_allgather(c, 16);
                FR_LAUNCH(c, "k_fks_close_flag", k_fks_close_flag, dim3(1), dim3(1), F, all, P, it);
            }
            // (one rank: the closing pass has written hm->hist[it] itself, Fks2Work::hm_close)
            if (speculate) {
                // The host needs ~10 us from the ticket to its next launch.  The closing pass settles the stage six times out of seven, so what
                // follows it -- the exact sum of wt_remain, the comb, the emission counts -- is enqueued behind the ticket and runs while the
                // host looks at the flag; if the pass did change a delta, these kernels see the same flag and leave at once.
                const uint32_t tk = fr_ticket_reserve(c);
                launch_tail(&F.hist[it], tk);
                fr_stream_wait_ticket(c, tk);
                tail_done = true;
            }
            else fr_stream_wait(c);
            read_host();
            if (sequential) break;
            const volatile FksHost *hm = c->h_fks;
            if (hm->hist[it] == 0) {
                closed = true;
                needed = it;
                for (int j = 2; j < it; j++) if (hm->hist[j] == 0) { needed = j; break; }      // a plain replay already reproduced its predecessor: the closing pass could have come there
            }
            else { tail_done = false; norms_done = false; scan_totals(it); it++; plain = it; }       // it was a replay like any other: scan, add up, close again
        }
        if (closed) c->rounds_hint[STAGE] = needed > 2 ? needed : 2;
    }
    else
    while (!needed && !sequential) {
        if (it + batch > FR_MAX_ROUNDS) batch = FR_MAX_ROUNDS - it;
        if (batch <= 0 || it >= 48) { sequential = true; break; }       // the replay does not settle (e.g. the reference's own 0/0 corner): walk the stage in order instead
        for (int k = 0; k < batch; k++) { replay(it); it++; }
        fr_stream_wait(c);       // the kernels wrote the flags and the stage's closing scalars into c->h_fks themselves
        const volatile FksHost *hm = c->h_fks;
        for (int j = 1; j < it && !needed; j++) if (hm->hist[j] == 0) needed = j + 1;      // replay 0 always counts as changed
        batch = 1;
        read_host();
    }
    // A stage that removes (almost) all of its norm: the reference's running norm is then its own rounding noise, which only the
    // in-order walk reproduces (fks_seq.hpp).  psG[0] / G_last are sums over the ranks, so every rank decides alike.
    // (a norm that went negative did so by rounding noise of the same kind)
    if (!sequential && !c->fks_no_collapse_walk && (!(hscal.G_last >= 1e-3 * hscal.psG[0]) || hscal.G_neg < 0) && hscal.psG[0] > 0) sequential = true;
    if (sequential) {
        tail_done = false; norms_done = false;
        if (hscal.overflow) FR_LAUNCH(c, "k_err_clear", k_err_clear, dim3(1), dim3(1), c->d_err, (uint32_t)FR_ERR_ROUNDS);     // FR_ERR_ROUNDS of the abandoned replay only: d_err also carries the flags of earlier stages and iterations of the batch
        run_fks_sequential<STAGE, NEW_HB>(c, cur, grid, F, n_bound);
    }
    else if (!closing) c->rounds_hint[STAGE] = needed > 2 ? needed : 2;
    c->fks_iters[STAGE] = it;
    if (c->dbg == 3) {
        FksScal hs; uint32_t hh[FR_MAX_ROUNDS + 2];
        FR_HIP(hipMemcpy(&hs, F.scal, sizeof(hs), hipMemcpyDeviceToHost));
        FR_HIP(hipMemcpy(hh, F.hist, sizeof(hh), hipMemcpyDeviceToHost));
        fprintf(stderr, "[fks] stage %d n_in %u replays %d needed %d n_pass %d lane_evals %u wave_eval_rounds %u (per replay %.0f / %.0f) psN:", STAGE, hs.n_in, it, needed, hs.n_pass,
                hh[FR_MAX_ROUNDS], hh[FR_MAX_ROUNDS + 1], hh[FR_MAX_ROUNDS] / (double)it, hh[FR_MAX_ROUNDS + 1] / (double)it);
        for (int p = 0; p < hs.n_pass; p++) fprintf(stderr, " %u", hs.psN[p]);
        fprintf(stderr, "\n   waves deciding / waves per light replay:");
        std::vector<uint32_t> dc((size_t)FR_MAX_ROUNDS * 4);
        FR_HIP(hipMemcpy(dc.data(), F.dbg_cnt, dc.size() * 4, hipMemcpyDeviceToHost));
        for (int k = 0; k < it; k++) fprintf(stderr, "  [%d] %u/%u np %u chg %u", k, dc[k * 4 + 2], dc[k * 4 + 3], dc[k * 4], hh[k]);
        fprintf(stderr, "\n");
        FR_HIP(hipMemset(F.dbg_cnt, 0, dc.size() * 4));
    }
    if (!sequential && !closed) {
        if (c->d_tie) FR_LAUNCH(c, "k_fks_tie", k_fks_tie, dim3(64), dim3(FR_BLOCK), F, c->d_tie);
        // settled: recompute every wt_remain with the budget of its last flagged sweep
        FR_LAUNCH(c, "k_fks_final", (k_fks_sweep<STAGE, NEW_HB, 2>), dim3(gridE), dim3(FR_BLOCK), W, F, c->d_hb, cur, it, c->p_doub, 0, 0);
    }
    if (tail_done) { }
    else if (simple_tail) launch_tail(nullptr);
    else {
        if (!norms_done) {
        FR_LAUNCH(c, "k_seq_sums", (k_seq_sums<AccWt>), dim3(grid), dim3(FR_BLOCK), W.seq, acc);
        FR_LAUNCH(c, "k_seq_maps", (k_seq_maps<AccWt>), dim3(grid), dim3(FR_BLOCK), W.seq, acc, fr_seq_from_zero());
        FR_LAUNCH(c, "k_seq_chain", (k_seq_chain<AccWt>), dim3(1), dim3(FR_BLOCK), W.seq, acc, fr_seq_from_zero());
        }
        const double *norms = norms_done ? c->d_norms_all : W.seq.total;
        if (xr && !norms_done) {
            // every rank's remaining norm (compress_utils.cpp:817-818), then the in-order lbound chain again from this
            // rank's offset: a floating-point running sum depends on where it starts
            FR_LAUNCH(c, "k_put_norm", k_put_norm, dim3(1), dim3(1), W, F, (double *)c->comm.small_send);
            norms = (const double *)fr_allgather(c, sizeof(double));
        }
        FR_LAUNCH(c, "k_comp_finalize", k_comp_finalize2, dim3(1), dim3(64), W, F, rn, norms, c->rank, P);
        if (c->rank > 0) {
            SeqStart from; from.norms = c->d_norms_keep; from.n = c->rank;
            SeqWork Q2 = W.seq; Q2.total = c->d_seq_scratch;
            FR_LAUNCH(c, "k_keep_norms", k_keep_norms, dim3(1), dim3(64), norms, P, c->d_norms_keep, W, F);
            FR_LAUNCH(c, "k_seq_maps", (k_seq_maps<AccWt>), dim3(grid), dim3(FR_BLOCK), Q2, acc, from);
            FR_LAUNCH(c, "k_seq_chain", (k_seq_chain<AccWt>), dim3(1), dim3(FR_BLOCK), Q2, acc, from);
        }
        if (W.prop) FR_HIP(hipMemsetAsync(W.act_n, 0, 8, st));
        FR_LAUNCH(c, "k_sys_count", (k_sys_count<STAGE, NEW_HB>), dim3(grid), dim3(FR_BLOCK), W, c->vec, c->d_hb, cur, c->p_doub);
    #ifdef FR_SYS_TIMING
        fr_sys_timing_dump(c, "k_sys_count", STAGE, grid, 6);
    #endif
        if (W.prop) {
            // chains of repairs, each walked by one lane (k_sys_walk); a round ends where chains ran into one another, and those are walked
            // on in the next round.  This is the frisys_hh path, where one stage needs ~1e4 repairs in chains of up to a few hundred
            // elements; the molecular path keeps the short sequential fix-up below.  Two rounds per host look at the list length.
            int in = 0;
            uint32_t na = 0;
            FR_HIP(hipMemcpyAsync(&na, &W.act_n[0], 4, hipMemcpyDeviceToHost, st));
            FR_HIP(hipStreamSynchronize(st));
            if (na > W.cap) throw FriesError("comb repair list overflow");
            for (int round = 0; na != 0; ) {
                if (round > 100000) throw FriesError("comb repair did not settle");
                unsigned gp = fr_blocks(na, FR_BLOCK);
                if (gp > 1024) gp = 1024;
                for (int k = 0; k < 2; k++, round++) {      // the next list is never longer than this one
                    const uint32_t tag = ++c->prop_tag;
                    if (tag == 0) throw FriesError("repair tags exhausted");
                    FR_HIP(hipMemsetAsync(&W.act_n[in ^ 1], 0, 4, st));
                    FR_LAUNCH(c, "k_sys_mark", k_sys_mark, dim3(gp), dim3(FR_BLOCK), W, in, tag);
                    FR_LAUNCH(c, "k_sys_walk", (k_sys_walk<STAGE, NEW_HB>), dim3(gp), dim3(FR_BLOCK), W, c->vec, c->d_hb, cur, c->p_doub, in, tag);
                    in ^= 1;
                }
                FR_HIP(hipMemcpyAsync(&na, &W.act_n[in], 4, hipMemcpyDeviceToHost, st));
                FR_HIP(hipStreamSynchronize(st));
            }
        }
        else FR_LAUNCH(c, "k_sys_fixup", (k_sys_fixup<STAGE, NEW_HB>), dim3(1), dim3(FR_BLOCK), W, c->vec, c->d_hb, cur, c->p_doub, c->d_err);
    }
    // the stage's emission count also goes to the host block (word 16 + slot): a copy into pageable memory would hold the host until it is done
    fr_rb_init(c);
    FR_LAUNCH(c, "k_sys_write", (k_sys_write<STAGE, NEW_HB>), dim3(grid), dim3(FR_BLOCK), W, c->vec, c->d_hb, cur, c->p_doub, c->d_err, out_slot >= 0 ? c->d_misc() + 16 + out_slot : nullptr);
#ifdef FR_SYS_TIMING
    fr_sys_timing_dump(c, "k_sys_write", STAGE, grid, 5);
#endif
    if (c->dbg == 5) {
        CompState fs; FksScal hs;
        FR_HIP(hipMemcpy(&fs, &W.state[FR_MAX_ROUNDS + 1], sizeof(fs), hipMemcpyDeviceToHost));
        FR_HIP(hipMemcpy(&hs, F.scal, sizeof(hs), hipMemcpyDeviceToHost));
        fprintf(stderr, "[stage %d rank %d] n_in %u n_rem %u unit %.17g loc_norm %.17g G %.6g n_fix %u n_out %u n_pass %d replays %d\n", STAGE, c->rank, fs.n_in, fs.n_rem, fs.unit, fs.loc_norm, fs.G, fs.n_fix, fs.n_out, hs.n_pass, it);
    }
}

template <bool NEW_HB>
static void hbpp_apply_t(FriesCtx *c, uint32_t n_samp, const double rn[5], int unit_matrel) {
    CompWork &W = c->W;
    hipStream_t st = c->stream;
    uint32_t bound1 = c->h_vst.curr_size - c->vec.n_dense;
    uint32_t bound = n_samp + 64 < W.cap ? n_samp + 64 : W.cap;     // a stage never emits more than n_samp entries
    if (bound1 > W.cap) throw FriesError("vector larger than HB-PP work capacity");
    // with ranks, n_samp is the global budget and a shard usually emits ~1/n_ranks of it: size the next stage's
    // launches from the emission count (one host sync per stage; the replay loop syncs anyway)
    const uint32_t bound_max = bound;
    auto next_bound = [&](int k) {
        if (!c->use_comm) return bound_max;
        fr_stream_wait(c);
        c->comp_len[k] = c->h_misc()[16 + k];
        uint32_t b = c->comp_len[k] + 64;
        return b < bound_max ? b : bound_max;
    };
    run_stage<1, NEW_HB>(c, 0, bound1, n_samp, rn[0], 0);
    bound = next_bound(0);
    run_stage<2, NEW_HB>(c, 1, bound, n_samp, rn[1], 1);
    bound = next_bound(1);
    run_stage<3, NEW_HB>(c, 0, bound, n_samp, rn[2], 2);
    bound = next_bound(2);
    run_stage<4, NEW_HB>(c, 1, bound, n_samp, rn[3], 3);
    bound = next_bound(3);
    run_stage<5, NEW_HB>(c, 0, bound, n_samp, rn[4], 4);
    bound = next_bound(4);
    unsigned grid = fr_blocks(bound, FR_TILE);
    SysDev S; S.n_orb = c->n_orb; S.n_elec = c->n_elec; S.h_core = c->d_h; S.eris = c->d_eris; S.hb = c->d_hb; S.hf_en = c->hf_en; S.spin_parity = c->spin_parity;
    // f_val / f_orbs reuse the S / kin arrays of the (finished) last stage
    double *f_val = W.S; uint32_t *f_orbs = W.kin;
    FR_LAUNCH(c, "k_final_eval", (k_final_eval<NEW_HB>), dim3(grid), dim3(FR_BLOCK), W, c->vec, S, 0, c->p_doub, unit_matrel, f_val, f_orbs, W.pcnt[0]);
    FR_LAUNCH(c, "k_final_compact", k_final_compact, dim3(grid), dim3(FR_BLOCK), W, 0, f_val, f_orbs, W.pcnt[0], c->c_pos, c->c_orbs, c->c_val, c->d_nsucc);
    {
        uint32_t tk = 0;
        const void *h_ns = fr_readback(c, c->d_nsucc, 4, false, &tk);
        fr_stream_wait_ticket(c, tk);
        memcpy(&c->num_success, h_ns, 4);
        for (int k = 0; k < 5; k++) c->comp_len[k] = c->h_misc()[16 + k];
    }
}

void fr_hbpp_apply(FriesCtx *c, uint32_t n_samp, const double rn[5]) {
    if (c->new_hb) hbpp_apply_t<true>(c, n_samp, rn, 0);
    else hbpp_apply_t<false>(c, n_samp, rn, 0);
}
void fr_hbpp_apply_unit(FriesCtx *c, uint32_t n_samp, const double rn[5]) {
    if (c->new_hb) hbpp_apply_t<true>(c, n_samp, rn, 1);
    else hbpp_apply_t<false>(c, n_samp, rn, 1);
}

// frisys_hh.cpp:187-224: hop-vs-phonon (two sub-weights per element), then which hop / which phonon move (uniform)
void fr_hh_apply(FriesCtx *c, uint32_t n_samp, const double rn[2]) {
    CompWork &W = c->W;
    uint32_t bound1 = c->h_vst.curr_size;
    uint32_t bound = n_samp + 64 < W.cap ? n_samp + 64 : W.cap;
    if (bound1 > W.cap) throw FriesError("vector larger than the compression work capacity");
    run_stage<1, true>(c, 0, bound1, n_samp, rn[0], 0);
    if (c->use_comm) { fr_stream_wait(c); c->comp_len[0] = c->h_misc()[16]; uint32_t b = c->comp_len[0] + 64; if (b < bound) bound = b; }
    run_stage<2, true>(c, 1, bound, n_samp, rn[1], 1, true);
    fr_stream_wait(c);
    c->comp_len[0] = c->h_misc()[16]; c->comp_len[1] = c->h_misc()[17];
    c->num_success = c->comp_len[1];
}

// one of the two comp_sub calls of frisys_hh.cpp:187-224 on its own (the reference's driver source behind include/FRIES calls them one by
// one): stage 1 = hop / phonon on the stored vector, stage 2 = which hop / which phonon move on stage 1's emissions.  Emissions stay in
// W.e_wi / e_sub / e_val; comp_len[stage - 1] = their number.
void fr_hh_stage(FriesCtx *c, int stage, uint32_t n_samp, double rn) {
    CompWork &W = c->W;
    if (stage == 1) {
        const uint32_t bound1 = c->h_vst.curr_size;
        if (bound1 > W.cap) throw FriesError("vector larger than the compression work capacity");
        run_stage<1, true>(c, 0, bound1, n_samp, rn, 0);
    }
    else {
        uint32_t bound = n_samp + 64 < W.cap ? n_samp + 64 : W.cap;
        const uint32_t b = c->comp_len[0] + 64;
        if (b < bound) bound = b;
        run_stage<2, true>(c, 1, bound, n_samp, rn, 1, true);
    }
    fr_stream_wait(c);
    c->comp_len[stage - 1] = c->h_misc()[16 + stage - 1];
}

// ------------------------------------------------------------------ apply_HBPP_piv (heat_bathPP.cpp:1014-1419, spin_parity 0)
// Every factor of the HB-PP factorisation is multiplied out -- element e of the short vector becomes a group of values in
// the long vector (long_vec) -- the long vector is compressed by piv_comp_parallel (find_preserve + pivotal sampling,
// pivotal.hip) and collapsed back to the elements that were not zeroed (collapse_long_, :994-1012).  The short vector of a stage
// is the same StageElems the systematic path builds (k_prep1 / k_prep: value incl. the factor's total weight, orbital code, row
// cache): the products are the same numbers (p * (v * tot) == (v * tot) * p) and a group's size is the row length comp_sub sees,
// or 0 where the reference leaves the group empty (:1061, :1108, :1133, :1167, :1222).
struct PvLong { double *vals; uint32_t *parent; uint32_t *goff; uint8_t *del; uint32_t *total; };

template <int STAGE, bool NEW_HB>
__device__ __forceinline__ unsigned fr_pv_gsize(unsigned n_elec, unsigned n_orb, double val, uint32_t ndiv, uint32_t nsub) {
    if (ndiv) return (val == 0 && ndiv == 1) ? 0u : ndiv;     // ndiv == 1 with a zero value is how k_prep marks an empty group
    if (STAGE == 1) return 2;
    if (STAGE == 2) return n_elec - (NEW_HB ? 1 : 0);
    if (STAGE == 3) return NEW_HB ? nsub : n_elec;
    if (STAGE == 4) return n_orb - n_elec / 2;
    return nsub;
}

template <int STAGE, bool NEW_HB>
__global__ void __launch_bounds__(FR_BLOCK) k_pv_count(CompWork W, const HbTables *Tg, int cur, uint32_t *pcnt, uint32_t *total) {
    __shared__ uint32_t shu[4];
    const unsigned n_in = W.state[0].n_in;
    const unsigned nblk = (n_in + FR_TILE - 1) / FR_TILE;
    if (blockIdx.x >= nblk) return;
    const StageElems E = W.el[cur];
    const unsigned n_elec = Tg->n_elec, n_orb = Tg->n_orb;
    size_t base = (size_t)blockIdx.x * FR_TILE + (size_t)threadIdx.x * FR_ITEMS;
    uint32_t cnt = 0;
    for (int it = 0; it < FR_ITEMS; it++) {
        size_t e = base + it;
        if (e >= n_in) break;
        cnt += fr_pv_gsize<STAGE, NEW_HB>(n_elec, n_orb, E.val[e], E.ndiv[e], E.nsub[e]);
    }
    uint32_t bc = fr_block_sum_u32(cnt, shu);
    if (threadIdx.x == 0) { pcnt[blockIdx.x] = bc; if (bc) atomicAdd(total, bc); }
}

template <int STAGE, bool NEW_HB>
__global__ void __launch_bounds__(FR_BLOCK) k_pv_expand(CompWork W, VecDev V, const HbTables *Tg, int cur, double p_doub, const uint32_t *pcnt, PvLong L) {
    __shared__ HbTables T;
    __shared__ uint32_t shu[4];
    const unsigned n_in = W.state[0].n_in;
    const unsigned nblk = (n_in + FR_TILE - 1) / FR_TILE;
    if (blockIdx.x >= nblk) return;
    fr_stage_tables(&T, Tg);
    const StageElems E = W.el[cur];
    uint32_t off;
    { uint32_t x = 0; for (unsigned i = threadIdx.x; i < blockIdx.x; i += blockDim.x) x += pcnt[i]; off = fr_block_sum_u32(x, shu); }
    const unsigned n_elec = T.n_elec, n_orb = T.n_orb;
    size_t base = (size_t)blockIdx.x * FR_TILE + (size_t)threadIdx.x * FR_ITEMS;
    uint32_t g[FR_ITEMS], tsum = 0;
#pragma unroll
    for (int it = 0; it < FR_ITEMS; it++) {
        size_t e = base + it;
        g[it] = e < n_in ? fr_pv_gsize<STAGE, NEW_HB>(n_elec, n_orb, E.val[e], E.ndiv[e], E.nsub[e]) : 0u;
        tsum += g[it];
    }
    uint32_t tot;
    uint32_t incl = fr_block_scan_u32(tsum, shu, &tot);
    uint32_t o = off + incl - tsum;
#pragma unroll
    for (int it = 0; it < FR_ITEMS; it++) {
        size_t e = base + it;
        if (e >= n_in) break;
        L.goff[e] = o;
        if (g[it] == 0) continue;
        const double val = E.val[e];
        const uint32_t nd = E.ndiv[e];
        if (nd) {
            const double part = val / nd;
            for (uint32_t j = 0; j < nd; j++) { L.vals[o + j] = part; L.parent[o + j] = (uint32_t)e; }
        }
        else {
            RowInfo ri = (STAGE == 1) ? fr_row1(W.row1) : fr_row_cached(E, e);
            const det_t det = (STAGE == 1) ? 0ull : V.dets[E.pos[e]];
            const uint32_t code = (STAGE == 1) ? 0u : E.code[e];
            const uint32_t ge = g[it];
            fr_row_visit<STAGE, NEW_HB>(T, det, code, ri, p_doub, [&](unsigned s, double w) {
                if (s < ge) { L.vals[o + s] = w * val; L.parent[o + s] = (uint32_t)e; }
            });
        }
        o += g[it];
    }
}

// collapse_long_: the elements the compression did not zero become the next stage's emissions (source element, index in its group, value)
__global__ void __launch_bounds__(FR_BLOCK) k_pv_ccount(PvLong L, uint32_t n_long, uint32_t *pcnt) {
    __shared__ uint32_t shu[4];
    size_t base = (size_t)blockIdx.x * FR_TILE + (size_t)threadIdx.x * FR_ITEMS;
    uint32_t cnt = 0;
    for (int it = 0; it < FR_ITEMS; it++) { size_t l = base + it; if (l < n_long && !L.del[l]) cnt++; }
    uint32_t bc = fr_block_sum_u32(cnt, shu);
    if (threadIdx.x == 0) pcnt[blockIdx.x] = bc;
}
__global__ void __launch_bounds__(FR_BLOCK) k_pv_cwrite(CompWork W, PvLong L, uint32_t n_long, const uint32_t *pcnt, uint32_t out_cap, uint32_t *err) {
    __shared__ uint32_t shu[4];
    uint32_t off;
    { uint32_t x = 0; for (unsigned i = threadIdx.x; i < blockIdx.x; i += blockDim.x) x += pcnt[i]; off = fr_block_sum_u32(x, shu); }
    size_t base = (size_t)blockIdx.x * FR_TILE + (size_t)threadIdx.x * FR_ITEMS;
    uint32_t f[FR_ITEMS], tsum = 0;
#pragma unroll
    for (int it = 0; it < FR_ITEMS; it++) { size_t l = base + it; f[it] = (l < n_long && !L.del[l]) ? 1u : 0u; tsum += f[it]; }
    uint32_t tot;
    uint32_t incl = fr_block_scan_u32(tsum, shu, &tot);
    uint32_t o = off + incl - tsum;
    bool over = false;
#pragma unroll
    for (int it = 0; it < FR_ITEMS; it++) {
        size_t l = base + it;
        if (l >= n_long) break;
        if (f[it]) {
            if (o < out_cap) { uint32_t e = L.parent[l]; W.e_wi[o] = e; W.e_sub[o] = (uint32_t)l - L.goff[e]; W.e_val[o] = L.vals[l]; }
            else over = true;
            o++;
        }
        L.del[l] = 0;
    }
    if (over) atomicOr(err, FR_ERR_SPAWN_CAP);
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == FR_BLOCK - 1) { W.state[FR_MAX_ROUNDS + 1].n_out = o < out_cap ? o : out_cap; L.total[1] = o; }
}

// the last factor's survivors: orbitals, weight, matrix element with its sign, value (:1254-1417); f_val == 0 marks a dropped sample
template <bool NEW_HB>
__global__ void __launch_bounds__(FR_BLOCK) k_final_eval_piv(CompWork W, VecDev V, SysDev S, int prev, double p_doub, int unit_matrel,
                                                             double *f_val, uint32_t *f_orbs, uint32_t *pcnt) {
    __shared__ HbTables T;
    __shared__ uint32_t shu[4];
    const unsigned n_in = W.state[FR_MAX_ROUNDS + 1].n_out;
    const unsigned nblk = (n_in + FR_TILE - 1) / FR_TILE;
    if (blockIdx.x >= nblk) return;
    fr_stage_tables(&T, S.hb);
    const StageElems P = W.el[prev];
    const unsigned n_orb = T.n_orb;
    size_t base = (size_t)blockIdx.x * FR_TILE + threadIdx.x;
    uint32_t cnt = 0;
    for (int it = 0; it < FR_ITEMS; it++) {
        size_t e = base + (size_t)it * FR_BLOCK;
        if (e >= n_in) break;
        uint32_t wi = W.e_wi[e], sub = W.e_sub[e];
        double val = W.e_val[e];
        uint32_t pos = P.pos[wi], pc = P.code[wi];
        det_t det = V.dets[pos];
        unsigned o1_idx = fr_c(pc, 1);
        double el = 0;
        uint32_t orbs = 0;
        if (fr_c(pc, 0) == 0) {
            unsigned o1 = fr_nth_bit(det, o1_idx), o2 = fr_nth_bit(det, fr_c(pc, 2)), u1 = fr_c(pc, 3);
            unsigned ir = T.irrep[o1 % n_orb] ^ T.irrep[o2 % n_orb] ^ T.irrep[u1 % n_orb];
            unsigned u2 = T.lookup[ir][sub + 1] + n_orb * (o2 / n_orb);
            if (!fr_bit(det, u2) && u1 != u2) {
                if (u1 > u2) { unsigned t = u1; u1 = u2; u2 = t; }
                if (o1 > o2) { unsigned t = o1; o1 = o2; o2 = t; }
                orbs = fr_code(o1, o2, u1, u2);
                double tw = NEW_HB ? fr_unnorm_wt(T, o1, o2, u1, u2) : fr_norm_wt(T, det, o1, o2, u1, u2);
                tw *= p_doub;
                double mel = unit_matrel ? 1.0 : fr_doub_matrel(o1, o2, u1, u2, S.eris, n_orb);
                mel *= fr_doub_parity(det, o1, o2, u1, u2);
                bool keep = true;
                if (S.spin_parity) { det_t tgt; keep = fr_adjust_tr(T, S, det, (det & ~(1ull << o1) & ~(1ull << o2)) | (1ull << u1) | (1ull << u2), &mel, S.spin_parity, &tgt, unit_matrel, &tw, p_doub); }
                el = keep ? val * mel / tw : 0.0;
            }
        }
        else {
            unsigned o1 = fr_nth_bit(det, o1_idx);
            unsigned u1 = fr_virt_from_idx(T, det, T.irrep[o1 % n_orb], n_orb * (o1 / n_orb), fr_c(pc, 2));
            if (u1 != 255) {
                orbs = fr_code(o1, u1, 0, 0);
                unsigned n_occ = fr_count_sing_allowed(T, det);
                double tw = (1 - p_doub) / n_occ / fr_c(pc, 3);
                double mel = unit_matrel ? 1.0 : fr_sing_matrel(det, o1, u1, S.h_core, S.eris, n_orb);
                mel *= fr_sing_parity(det, o1, u1);
                bool keep = true;
                if (S.spin_parity) { det_t tgt; keep = fr_adjust_tr(T, S, det, (det & ~(1ull << o1)) | (1ull << u1), &mel, S.spin_parity, &tgt, unit_matrel, &tw, p_doub); }
                el = keep ? val * mel / tw : 0.0;
            }
        }
        if (!(fabs(el) > 1e-12)) el = 0;
        f_val[e] = el; f_orbs[e] = orbs;
        cnt += (el != 0);
    }
    uint32_t bc = fr_block_sum_u32(cnt, shu);
    if (threadIdx.x == 0) pcnt[blockIdx.x] = bc;
}

template <int STAGE, bool NEW_HB>
static uint32_t pv_stage(FriesCtx *c, int cur, uint32_t n_in, uint32_t n_samp) {
    CompWork &W = c->W;
    FlatPiv &F = c->flat;
    hipStream_t st = c->stream;
    const unsigned grid = fr_blocks(n_in ? n_in : 1, FR_TILE);
    if (STAGE == 1) FR_LAUNCH(c, "k_prep1", k_prep1, dim3(grid), dim3(FR_BLOCK), W, c->vec, cur, n_samp);
    else FR_LAUNCH(c, "k_prep", (k_prep<STAGE, NEW_HB>), dim3(grid), dim3(FR_BLOCK), W, c->vec, c->d_hb, cur, n_samp, c->p_doub);
    FR_HIP(hipMemsetAsync(F.total, 0, 8, st));
    FR_LAUNCH(c, "k_pv_count", (k_pv_count<STAGE, NEW_HB>), dim3(grid), dim3(FR_BLOCK), W, c->d_hb, cur, W.pcnt[0], F.total);
    uint32_t n_long = 0;
    FR_HIP(hipMemcpyAsync(&n_long, F.total, 4, hipMemcpyDeviceToHost, st));
    FR_HIP(hipStreamSynchronize(st));
    if (n_long == 0) {          // nothing on this rank (or at all)
        FR_HIP(hipMemsetAsync(&W.state[FR_MAX_ROUNDS + 1].n_out, 0, 4, st));
        if (c->use_comm) fr_piv_comp_flat(c, 0, n_samp);     // the other ranks' compression still needs this one in its collectives
        return 0;
    }
    if (n_long > F.cap) fr_piv_flat_reserve(c, n_long + n_long / 4 + 1024);
    PvLong L{F.vals, F.parent, c->pv_goff, F.vc.del, F.total};
    FR_LAUNCH(c, "k_pv_expand", (k_pv_expand<STAGE, NEW_HB>), dim3(grid), dim3(FR_BLOCK), W, c->vec, c->d_hb, cur, c->p_doub, W.pcnt[0], L);
    fr_piv_comp_flat(c, n_long, n_samp);
    L.del = F.vc.del;
    const unsigned gl = fr_blocks(n_long, FR_TILE);
    FR_LAUNCH(c, "k_pv_ccount", k_pv_ccount, dim3(gl), dim3(FR_BLOCK), L, n_long, W.pcnt[0]);
    FR_LAUNCH(c, "k_pv_cwrite", k_pv_cwrite, dim3(gl), dim3(FR_BLOCK), W, L, n_long, W.pcnt[0], W.cap, c->d_err);
    uint32_t n_out = 0;
    FR_HIP(hipMemcpyAsync(&n_out, F.total + 1, 4, hipMemcpyDeviceToHost, st));
    FR_HIP(hipStreamSynchronize(st));
    if (n_out > W.cap) throw FriesError("pivotal matrix compression: more survivors than the work arrays hold");
    return n_out;
}

template <bool NEW_HB>
static void hbpp_piv_t(FriesCtx *c, uint32_t n_samp, int unit_matrel, uint32_t stage_len[5]) {
    CompWork &W = c->W;
    hipStream_t st = c->stream;
    fr_vec_sync_state(c, &c->vec, &c->h_vst);
    uint32_t n = c->h_vst.curr_size;
    if (n > W.cap) throw FriesError("vector larger than HB-PP work capacity");
    if (!c->pv_goff) c->pv_goff = fr_alloc<uint32_t>(W.cap);
    fr_piv_flat_reserve(c, 2 * (n > n_samp ? n : n_samp) + 1024);
    n = pv_stage<1, NEW_HB>(c, 0, n, n_samp); stage_len[0] = n;
    n = pv_stage<2, NEW_HB>(c, 1, n, n_samp); stage_len[1] = n;
    n = pv_stage<3, NEW_HB>(c, 0, n, n_samp); stage_len[2] = n;
    n = pv_stage<4, NEW_HB>(c, 1, n, n_samp); stage_len[3] = n;
    n = pv_stage<5, NEW_HB>(c, 0, n, n_samp); stage_len[4] = n;
    const unsigned grid = fr_blocks(n ? n : 1, FR_TILE);
    SysDev S; S.n_orb = c->n_orb; S.n_elec = c->n_elec; S.h_core = c->d_h; S.eris = c->d_eris; S.hb = c->d_hb; S.hf_en = c->hf_en; S.spin_parity = c->spin_parity;
    double *f_val = W.S; uint32_t *f_orbs = W.kin;
    FR_LAUNCH(c, "k_final_eval_piv", (k_final_eval_piv<NEW_HB>), dim3(grid), dim3(FR_BLOCK), W, c->vec, S, 0, c->p_doub, unit_matrel, f_val, f_orbs, W.pcnt[0]);
    FR_LAUNCH(c, "k_final_compact", k_final_compact, dim3(grid), dim3(FR_BLOCK), W, 0, f_val, f_orbs, W.pcnt[0], c->c_pos, c->c_orbs, c->c_val, c->d_nsucc);
    FR_HIP(hipMemcpyAsync(&c->num_success, c->d_nsucc, 4, hipMemcpyDeviceToHost, st));
    FR_HIP(hipStreamSynchronize(st));
}

void fr_hbpp_piv_apply(FriesCtx *c, uint32_t n_samp, int unit_matrel, uint32_t stage_len[5]) {
    if (c->new_hb) hbpp_piv_t<true>(c, n_samp, unit_matrel, stage_len);
    else hbpp_piv_t<false>(c, n_samp, unit_matrel, stage_len);
}
