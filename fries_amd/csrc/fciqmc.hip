// fciqmc_mol on the device (FRIES_bin/fciqmc_mol.cpp:331-412, near-uniform excitation generator, near_uniform.cpp).
//
// The reference consumes one sequential mt19937 stream whose length per determinant depends on the values drawn (rejection
// loops), which no parallel sampler can replay.  The device draws every uniform from a counter-based stream
//     u = hash(seed, iteration, determinant, attempt, purpose, n-th draw of that attempt) / 2^32
// -- the same i.i.d. uniforms, a different stream.  The tests' CPU restatement of the reference has that stream as a second mode,
// so the device trajectory is checked bit for bit against it, while its sampling functions and loop are pinned against the
// reference itself on the reference's own mt19937 stream.  Walker numbers are integers held exactly in the vector's doubles.
#include "ctx.hpp"
#include "hbpp_rows.hpp"

struct FqRng {
    unsigned long long key; uint32_t ctr;
    __device__ __forceinline__ void begin(unsigned long long seed, unsigned long long iter, det_t det, uint32_t attempt, uint32_t purpose) {
        unsigned long long h = fr_mix64(seed ^ 0x9e3779b97f4a7c15ull);
        h = fr_mix64(h ^ (iter * 0xd1b54a32d192ed03ull));
        h = fr_mix64(h ^ det);
        h = fr_mix64(h ^ (((unsigned long long)attempt << 8) | purpose));
        key = h; ctr = 0;
    }
    __device__ __forceinline__ double uni() {
        unsigned long long h = fr_mix64(key + (unsigned long long)(++ctr) * 0x9e3779b97f4a7c15ull);
        return (uint32_t)(h >> 32) / (1. + 4294967295.0);
    }
    __device__ __forceinline__ unsigned choose(unsigned nmax) { return (unsigned)(uni() * nmax); }     // near_uniform.cpp:41-44
};
#define FQ_HEAVY 128u
enum { FQ_BIN = 0, FQ_DOUB = 1, FQ_SING = 2, FQ_ROUND_D = 3, FQ_ROUND_S = 4, FQ_DEATH = 5, FQ_HB_O1 = 6, FQ_HB_O2 = 7, FQ_HB_U1 = 8, FQ_HB_U2 = 9, FQ_NWALK = 10, FQ_COMP = 11 };
// FqWork::multi: 0 fciqmc_mol (integer walkers), 1 frimulti_mol, 2 fciqmc_fp_mol (real-valued walkers)

// alias method (compress_utils.cpp:823-877) on a probability row of at most 32 states
struct FqAlias { uint8_t alias[32]; double prob[32]; };
__device__ inline void fq_setup_alias(FqAlias &A, const double *probs, unsigned n) {
    unsigned n_s = 0, n_b = 0;
    uint8_t smaller[32], bigger[32];
    for (unsigned i = 0; i < n; i++) {
        A.alias[i] = (uint8_t)i;
        A.prob[i] = n * probs[i];
        if (A.prob[i] < 1) smaller[n_s++] = (uint8_t)i; else bigger[n_b++] = (uint8_t)i;
    }
    while (n_s > 0 && n_b > 0) {
        unsigned sm = smaller[n_s - 1], b = bigger[n_b - 1];
        A.alias[sm] = (uint8_t)b;
        A.prob[b] += A.prob[sm] - 1;
        if (A.prob[b] < 1) { smaller[n_s - 1] = (uint8_t)b; n_b--; }
        else n_s--;
    }
}
__device__ __forceinline__ unsigned fq_sample_alias(const FqAlias &A, unsigned n, FqRng &rng) {
    unsigned chosen = (unsigned)(uint8_t)(rng.uni() * n);
    return rng.uni() < A.prob[chosen] ? chosen : A.alias[chosen];
}
// calc_o1_probs (heat_bathPP.cpp:182-200): normalised s_tens of the occupied orbitals
__device__ inline void fq_o1_probs(const HbTables &T, det_t det, double *p) {
    RowInfo ri = fr_row2_setup<false>(T, det);
    fr_row2_visit<false>(T, det, [&](unsigned s, double w) { p[s] = w * ri.inv_norm; });
}

// per stored determinant: walkers split into double / single attempts (bin_sample, :352), death / cloning (:396-403)
__global__ void __launch_bounds__(FR_BLOCK) k_fq_count(VecDev V, SysDev S, FqWork Q, unsigned long long seed, unsigned long long iter, double p_doub,
                                                       double eps, double shift, uint32_t init_thresh) {
    __shared__ HbTables T;
    __shared__ uint32_t shu[4];
    fr_stage_tables(&T, S.hb);
    const uint32_t n = V.st->curr_size;
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t att = 0, nz = 0, ini = 0;
    if (d < n) {
        const double cur = V.v0[d];
        const int cur_i = (int)cur;
        // frimulti_mol: real weights; the column's sample number comes from the comb (k_mc_walk) and may be zero
        unsigned n_walk = Q.multi == 1 ? (cur != 0 ? Q.n_walk[d] : 0u) : (unsigned)(cur_i < 0 ? -cur_i : cur_i);
        if (Q.multi == 2) {      // fciqmc_fp_mol.cpp:342: |v| rounded stochastically to the number of spawning attempts
            n_walk = 0;
            if (cur != 0) {
                FqRng r0; r0.begin(seed, iter, V.dets[d], 0, FQ_NWALK);
                const double a = fabs(cur); const int flr = (int)floor(a);
                n_walk = (unsigned)(flr + (r0.uni() < a - flr ? 1 : 0));
            }
            Q.n_walk[d] = n_walk;
        }
        uint32_t n_doub = 0, n_sing = 0;
        double new_val = Q.multi == 2 ? cur : 0;        // a real-valued walker without attempts is left alone (:343-345)
        if (Q.multi == 1 ? cur != 0 : n_walk != 0) {
            nz = 1; ini = Q.multi == 1 ? fabs(cur) > Q.init_f : n_walk > init_thresh;
            const int sign = cur < 0 ? -1 : 1;
            const det_t det = V.dets[d];
            double dg = V.diag[d];
            if (dg != dg) { dg = fr_diag_matrel(det, S.h_core, S.eris, S.n_orb) - S.hf_en; V.diag[d] = dg; }
            if (n_walk > FQ_HEAVY) {
                // a determinant with many walkers draws one uniform per walker, twice: handed to a whole workgroup (k_fq_heavy);
                // the counter-based stream is addressable by draw index, so the draws can be split over lanes
                uint32_t slot = atomicAdd(&Q.totals[3], 1u);
                Q.heavy[slot] = d;
                n_doub = 0; n_sing = 0; new_val = Q.multi ? cur * (1 - eps * (dg - shift)) : cur;
            }
            else {
                FqRng rng;
                rng.begin(seed, iter, det, 0, FQ_BIN);
                for (unsigned i = 0; i < n_walk; i++) n_doub += rng.uni() < p_doub;
                n_sing = n_walk - n_doub;
                // sing_multin returns nothing when no electron has a symmetry-allowed excitation (near_uniform.cpp:293-295)
                if (fr_count_sing_allowed(T, det) == 0) n_sing = 0;
                if (Q.multi) new_val = cur * (1 - eps * (dg - shift));       // frimulti_mol.cpp:379-380
                else {
                    const double m = (1 - eps * (dg - shift)) * sign;
                    const int flr = (int)floor(m);
                    const double prob = m - flr;
                    int ret = flr * (int)n_walk;
                    rng.begin(seed, iter, det, 0, FQ_DEATH);
                    for (unsigned i = 0; i < n_walk; i++) ret += rng.uni() < prob;
                    new_val = (double)ret;
                }
                if (Q.o1cnt) {          // heat-bath doubles: how many samples chose each electron as o1 (heat_bathPP.cpp:614-625)
                    const unsigned ne = T.n_elec;
                    double p1[32]; FqAlias A;
                    uint32_t cnt[32];
                    for (unsigned e = 0; e < ne; e++) cnt[e] = 0;
                    if (n_doub) { fq_o1_probs(T, det, p1); fq_setup_alias(A, p1, ne); }
                    for (unsigned i = 0; i < n_doub; i++) { rng.begin(seed, iter, det, i, FQ_HB_O1); cnt[fq_sample_alias(A, ne, rng)]++; }
                    for (unsigned e = 0; e < ne; e++) Q.o1cnt[(size_t)d * ne + e] = cnt[e];
                }
            }
        }
        Q.n_doub[d] = n_doub; Q.n_att[d] = n_doub + n_sing; Q.new_val[d] = new_val;
    }
    uint32_t b_nz = fr_block_sum_u32(nz, shu);
    __syncthreads();
    uint32_t b_ini = fr_block_sum_u32(ini, shu);
    if (threadIdx.x == 0) { Q.blk_nz[blockIdx.x] = b_nz; Q.blk_ini[blockIdx.x] = b_ini; }
}

// Determinants with more than FQ_HEAVY walkers: the same draws as the lane loop above (draw i of a stream is hash(key + (i + 1) * golden),
// whoever evaluates it), spread over FQ_HSLICES workgroups per determinant; the counts are integers, so the atomic adds commute.
// Pass A: the binomial split and the death draws (n_doub / n_att of a heavy determinant are zero when it starts and serve as the
// accumulators).  Pass B: samples per first occupied electron (heat-bath generator), then the attempt number and the new value.
#define FQ_HSLICES 32
// workgroups sharing one heavy determinant: one per 8192 walkers (the per-workgroup set-up -- alias table, tables in LDS -- is not free)
__device__ __forceinline__ unsigned fq_heavy_slices(unsigned n_walk) { unsigned s = (n_walk + 8191u) / 8192u; return s < 1u ? 1u : (s > FQ_HSLICES ? FQ_HSLICES : s); }
__global__ void __launch_bounds__(FR_BLOCK) k_fq_heavy_a(VecDev V, SysDev S, FqWork Q, unsigned long long seed, unsigned long long iter, double p_doub,
                                                         double eps, double shift) {
    __shared__ uint32_t shu[4];
    const uint32_t nh = Q.totals[3];
    for (uint32_t h = blockIdx.x; h < nh; h += gridDim.x) {
        const uint32_t d = Q.heavy[h];
        const double cur = V.v0[d];
        const int cur_i = (int)cur;
        const unsigned n_walk = Q.multi ? Q.n_walk[d] : (unsigned)(cur_i < 0 ? -cur_i : cur_i);
        const int sign = cur < 0 ? -1 : 1;
        const unsigned ns = fq_heavy_slices(n_walk);
        if (blockIdx.y >= ns) continue;
        const det_t det = V.dets[d];
        FqRng rb, rd;
        rb.begin(seed, iter, det, 0, FQ_BIN);
        rd.begin(seed, iter, det, 0, FQ_DEATH);
        const double m = (1 - eps * (V.diag[d] - shift)) * sign;
        const int flr = (int)floor(m);
        const double prob = m - flr;
        uint32_t c_doub = 0, c_live = 0;
        for (unsigned i = blockIdx.y * blockDim.x + threadIdx.x; i < n_walk; i += ns * blockDim.x) {
            rb.ctr = i; rd.ctr = i;         // uni() pre-increments: draw number i + 1 of each stream
            c_doub += rb.uni() < p_doub;
            if (!Q.multi) c_live += rd.uni() < prob;
        }
        __syncthreads();
        const uint32_t b_doub = fr_block_sum_u32(c_doub, shu);
        __syncthreads();
        const uint32_t b_live = fr_block_sum_u32(c_live, shu);
        if (threadIdx.x == 0) { if (b_doub) atomicAdd(&Q.n_doub[d], b_doub); if (b_live) atomicAdd(&Q.n_att[d], b_live); }
        if (Q.o1cnt && blockIdx.y == 0 && threadIdx.x < S.n_elec) Q.o1cnt[(size_t)d * S.n_elec + threadIdx.x] = 0;
        __syncthreads();
    }
}
__global__ void __launch_bounds__(FR_BLOCK) k_fq_heavy_b(VecDev V, SysDev S, FqWork Q, unsigned long long seed, unsigned long long iter,
                                                         double eps, double shift) {
    __shared__ HbTables T;
    __shared__ FqAlias As;
    __shared__ uint32_t hist[32];
    fr_stage_tables(&T, S.hb);
    const uint32_t nh = Q.totals[3];
    const unsigned ne = T.n_elec;
    for (uint32_t h = blockIdx.x; h < nh; h += gridDim.x) {
        const uint32_t d = Q.heavy[h];
        const unsigned ns = fq_heavy_slices(Q.multi ? Q.n_walk[d] : (unsigned)fabs(V.v0[d]));
        if (blockIdx.y >= ns) continue;
        const det_t det = V.dets[d];
        const uint32_t n_doub = Q.n_doub[d];
        if (Q.o1cnt) {
            if (threadIdx.x == 0) { double p1[32]; fq_o1_probs(T, det, p1); fq_setup_alias(As, p1, ne); }
            if (threadIdx.x < 32) hist[threadIdx.x] = 0;
            __syncthreads();
            FqRng r1;
            // a lane per sample; the wave counts its samples per electron with ballots (lane k keeps the count of electron k): a few
            // hundred thousand atomic adds onto n_elec LDS words serialise otherwise
            uint32_t mine = 0;
            const int lane = fr_lane();
            for (unsigned i0 = blockIdx.y * blockDim.x; i0 < n_doub; i0 += ns * blockDim.x) {
                const unsigned i = i0 + threadIdx.x;
                unsigned e = 0xffu;
                if (i < n_doub) { r1.begin(seed, iter, det, i, FQ_HB_O1); e = fq_sample_alias(As, ne, r1); }
                for (unsigned k = 0; k < ne; k++) {
                    const unsigned long long mk = __ballot(e == k);
                    if ((unsigned)lane == k) mine += (uint32_t)__popcll(mk);
                }
            }
            if ((unsigned)lane < ne && mine) atomicAdd(&hist[lane], mine);
            __syncthreads();
            if (threadIdx.x < ne && hist[threadIdx.x]) atomicAdd(&Q.o1cnt[(size_t)d * ne + threadIdx.x], hist[threadIdx.x]);
        }
        if (blockIdx.y == 0 && threadIdx.x == 0) {       // nobody else reads n_att / new_val of this determinant in this kernel
            const double cur = V.v0[d];
            const int cur_i = (int)cur;
            const unsigned n_walk = Q.multi ? Q.n_walk[d] : (unsigned)(cur_i < 0 ? -cur_i : cur_i);
            const int sign = cur < 0 ? -1 : 1;
            const uint32_t n_live = Q.n_att[d];
            uint32_t n_sing = n_walk - n_doub;
            if (fr_count_sing_allowed(T, det) == 0) n_sing = 0;
            Q.n_att[d] = n_doub + n_sing;
            if (!Q.multi) {
                const double m = (1 - eps * (V.diag[d] - shift)) * sign;
                const int flr = (int)floor(m);
                Q.new_val[d] = (double)(flr * (int)n_walk + (int)n_live);
            }
        }
        __syncthreads();
    }
}

// attempts per workgroup of determinants (after the heavy ones are known)
__global__ void __launch_bounds__(FR_BLOCK) k_fq_blocksum(VecDev V, FqWork Q) {
    __shared__ uint32_t shu[4];
    const uint32_t n = V.st->curr_size;
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t b = fr_block_sum_u32(d < n ? Q.n_att[d] : 0u, shu);
    if (threadIdx.x == 0) Q.blk_att[blockIdx.x] = b;
}

// exclusive offsets of every determinant's attempts; totals (one workgroup per FR_BLOCK determinants, block prefix re-reduced)
__global__ void __launch_bounds__(FR_BLOCK) k_fq_offsets(VecDev V, FqWork Q) {
    __shared__ uint32_t shu[4];
    const uint32_t n = V.st->curr_size;
    const unsigned nblk = (n + FR_BLOCK - 1) / FR_BLOCK;
    uint32_t off;
    { uint32_t x = 0; for (unsigned i = threadIdx.x; i < blockIdx.x; i += blockDim.x) x += Q.blk_att[i]; off = fr_block_sum_u32(x, shu); }
    __syncthreads();
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t a = d < n ? Q.n_att[d] : 0u, tot;
    uint32_t incl = fr_block_scan_u32(a, shu, &tot);
    if (d < n) Q.att_off[d] = off + incl - a;
    if (blockIdx.x == nblk - 1 && threadIdx.x == FR_BLOCK - 1) {
        Q.totals[0] = off + incl;       // attempts
    }
    if (blockIdx.x == 0) {
        __syncthreads();
        uint32_t x = 0, y = 0;
        for (unsigned i = threadIdx.x; i < nblk; i += blockDim.x) { x += Q.blk_nz[i]; y += Q.blk_ini[i]; }
        uint32_t sx = fr_block_sum_u32(x, shu);
        __syncthreads();
        uint32_t sy = fr_block_sum_u32(y, shu);
        if (threadIdx.x == 0) { Q.totals[1] = sx; Q.totals[2] = sy; }
    }
}

// one spawning attempt per lane: excitation (doub_multin / sing_multin for one sample), matrix element, stochastic rounding
__global__ void __launch_bounds__(FR_BLOCK) k_fq_attempt(VecDev V, SysDev S, FqWork Q, unsigned long long seed, unsigned long long iter, double p_doub,
                                                         double eps, uint32_t init_thresh) {
    __shared__ HbTables T;
    __shared__ uint32_t shu[4];
    fr_stage_tables(&T, S.hb);
    const uint32_t n = V.st->curr_size, A = Q.totals[0];
    const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
    double sp_val = 0; det_t sp_det = 0; uint32_t sp_ini = 0;
    if (a < A) {
        // owner determinant: last d with att_off[d] <= a (empty determinants share their successor's offset)
        uint32_t lo = 0, hi = n - 1;
        while (lo < hi) { uint32_t mid = (lo + hi + 1) >> 1; if (Q.att_off[mid] <= a) lo = mid; else hi = mid - 1; }
        while (Q.n_att[lo] == 0 && lo > 0) lo--;       // cannot happen (an empty determinant never owns an attempt), kept for safety
        const uint32_t d = lo, i = a - Q.att_off[d], n_doub = Q.n_doub[d];
        const det_t det = V.dets[d];
        const double cur_f = V.v0[d];
        const int cur_i = (int)cur_f;
        const unsigned n_walk = Q.multi ? Q.n_walk[d] : (unsigned)(cur_i < 0 ? -cur_i : cur_i);
        const int sign = cur_f < 0 ? -1 : 1;
        sp_ini = Q.multi == 1 ? fabs(cur_f) > Q.init_f : n_walk > init_thresh;
        // frimulti_mol.cpp:317-320: the column's weight relative to one sampling unit, at most 1
        double colw = 1;
        if (Q.multi == 1) { colw = fabs(cur_f) / Q.samp_unit; if (colw > 1) colw = 1; }
        const bool keep_real = Q.multi == 2;           // fciqmc_fp_mol.cpp:385-390: only spawns below 0.01 are rounded
        const unsigned n_orb = T.n_orb, n_elec = T.n_elec;
        SymCounts sc; fr_count_symm_virt(sc, T, det);
        FqRng rng;
        if (i < n_doub && Q.o1cnt) {       // ---- heat-bath double (heat_bathPP.cpp:601-683): group = electron chosen as o1, k-th sample of the group
            const uint32_t *oc = Q.o1cnt + (size_t)d * n_elec;
            unsigned e = 0, off = 0;
            while (e + 1 < n_elec && off + oc[e] <= i) { off += oc[e]; e++; }
            const unsigned k = i - off;
            const uint32_t att = (e << 20) | k;
            const unsigned n_virt = n_orb - n_elec / 2;
            const unsigned o1 = fr_nth_bit(det, e);
            double pr[32]; FqAlias A;
            {   // second occupied orbital: calc_o2_probs (:203-233)
                RowInfo ri = fr_row3_setup(T, det, e);
                fr_row3_visit(T, det, e, ri.aux, [&](unsigned s2, double w) { pr[s2] = w * ri.inv_norm; });
                fq_setup_alias(A, pr, n_elec);
            }
            rng.begin(seed, iter, det, att, FQ_HB_O2);
            const unsigned o2 = fr_nth_bit(det, fq_sample_alias(A, n_elec, rng));
            {   // first virtual: calc_u1_probs (:273-319)
                RowInfo ri = fr_row4_setup(T, det, o1, false);
                for (unsigned q = 0; q < n_virt; q++) pr[q] = 0;
                fr_row4_visit(T, det, o1, false, [&](unsigned s2, double w) { pr[s2] = w * ri.inv_norm; });
                fq_setup_alias(A, pr, n_virt);
            }
            rng.begin(seed, iter, det, att, FQ_HB_U1);
            const unsigned u1 = fr_find_nth_virt(det, (int)(o1 / n_orb), n_orb, fq_sample_alias(A, n_virt, rng));
            {   // second virtual: calc_u2_probs (:322-365)
                const unsigned u2_symm = T.irrep[o1 % n_orb] ^ T.irrep[o2 % n_orb] ^ T.irrep[u1 % n_orb];
                double norm = 0;
                const unsigned num_u2 = fr_row5_visit<false>(T, det, o1, o2, u1, [&](unsigned s2, double w) { pr[s2] = w; norm += w; });
                if (norm != 0) {
                    const double inv = 1 / norm;
                    for (unsigned q = 0; q < num_u2; q++) pr[q] *= inv;
                    fq_setup_alias(A, pr, num_u2);
                    rng.begin(seed, iter, det, att, FQ_HB_U2);
                    unsigned u2 = fq_sample_alias(A, num_u2, rng);
                    u2 = T.lookup[u2_symm][u2 + 1] + n_orb * (o2 / n_orb);
                    if (!fr_bit(det, u2)) {
                        const unsigned a1 = o1 < o2 ? o1 : o2, a2 = o1 < o2 ? o2 : o1, b1 = u1 < u2 ? u1 : u2, b2 = u1 < u2 ? u2 : u1;
                        const double prob = fr_norm_wt(T, det, a1, a2, b1, b2);
                        double m = fr_doub_matrel(a1, a2, b1, b2, S.eris, n_orb);
                        if (Q.multi == 1) {     // a real-valued weight instead of a rounded walker (frimulti_mol.cpp:350-358)
                            if (fabs(m) > 1e-9) {
                                m *= -eps / prob / p_doub / n_walk * cur_f * fr_doub_parity(det, a1, a2, b1, b2) / colw;
                                sp_val = m;
                                sp_det = (det & ~(1ull << a1) & ~(1ull << a2)) | (1ull << b1) | (1ull << b2);
                            }
                        }
                        else {
                        m *= eps / prob / p_doub;
                        double sp = m;
                        if (!keep_real || fabs(m) < 0.01) {
                            rng.begin(seed, iter, det, att, FQ_ROUND_D);
                            const int flr = (int)floor(m);
                            sp = (double)(flr + (rng.uni() < m - flr ? 1 : 0));
                        }
                        if (sp != 0) {
                            sp *= -fr_doub_parity(det, a1, a2, b1, b2) * sign;
                            sp_val = sp;
                            sp_det = (det & ~(1ull << a1) & ~(1ull << a2)) | (1ull << b1) | (1ull << b2);
                        }
                        }
                    }
                }
            }
        }
        else if (i < n_doub) {       // ---- near-uniform double (near_uniform.cpp:193-245)
            rng.begin(seed, iter, det, i, FQ_DOUB);
            unsigned tri = rng.choose(n_elec * (n_elec - 1) / 2);
            unsigned i1 = (unsigned)((sqrt(tri * 8. + 1) - 1) / 2);
            unsigned i2 = (unsigned)(tri - i1 * (i1 + 1.) / 2);
            i1 += 1;
            unsigned orb1 = fr_nth_bit(det, i1), orb2 = fr_nth_bit(det, i2);
            unsigned spin1 = i1 / (n_elec / 2), spin2 = i2 / (n_elec / 2);
            unsigned sym_prod = T.irrep[orb1 % n_orb] ^ T.irrep[orb2 % n_orb];
            unsigned same_symm = (sym_prod == 0 && spin1 == spin2) ? 1u : 0u;
            unsigned n_allow = spin1 == spin2 ? n_orb - n_elec / 2 : 2 * n_orb - n_elec;
            for (unsigned k = 0; k < 8; k++) {
                if (sc.c[k ^ sym_prod][spin2] == same_symm) n_allow -= sc.c[k][spin1];
                if (spin1 != spin2 && sc.c[k ^ sym_prod][spin1] == same_symm) n_allow -= sc.c[k][spin2];
            }
            if (n_allow != 0) {
                int virt_choice;
                unsigned a_spin, b_spin, n_virt2, orbital;
                if (n_allow <= 3) {
                    virt_choice = (int)rng.choose(n_allow);
                    if (spin1 == spin2) { a_spin = spin1; b_spin = a_spin; } else { a_spin = 0; b_spin = 1; }
                    orbital = 0;
                    while (virt_choice >= 0 && orbital < n_orb) {
                        if (!fr_bit(det, orbital + a_spin * n_orb)) {
                            n_virt2 = sc.c[sym_prod ^ T.irrep[orbital]][b_spin] - ((sym_prod == 0 && a_spin == b_spin) ? 1u : 0u);
                            if (n_virt2 != 0) virt_choice -= 1;
                        }
                        orbital += 1;
                    }
                    if (virt_choice >= 0) {
                        a_spin = 1; b_spin = 0;
                        while (virt_choice >= 0 && orbital < 2 * n_orb) {
                            if (!fr_bit(det, orbital)) {
                                n_virt2 = sc.c[sym_prod ^ T.irrep[orbital - n_orb]][b_spin] - ((sym_prod == 0 && a_spin == b_spin) ? 1u : 0u);
                                if (n_virt2 != 0) virt_choice -= 1;
                            }
                            orbital += 1;
                        }
                        orbital -= n_orb;
                    }
                    virt_choice = (int)(orbital - 1 + a_spin * n_orb);
                }
                else {
                    n_virt2 = 0;
                    while (n_virt2 == 0) {
                        if (spin1 == spin2) { a_spin = spin1; b_spin = a_spin; virt_choice = (int)(rng.choose(n_orb) + a_spin * n_orb); }
                        else { virt_choice = (int)rng.choose(2 * n_orb); a_spin = (unsigned)virt_choice / n_orb; b_spin = 1 - a_spin; }
                        if (!fr_bit(det, (unsigned)virt_choice))
                            n_virt2 = sc.c[sym_prod ^ T.irrep[(unsigned)virt_choice % n_orb]][b_spin] - ((sym_prod == 0 && a_spin == b_spin) ? 1u : 0u);
                    }
                }
                const unsigned unocc1 = (unsigned)virt_choice;
                a_spin = unocc1 / n_orb;
                b_spin = spin1 ^ spin2 ^ a_spin;
                const unsigned a_symm = T.irrep[unocc1 % n_orb], b_symm = sym_prod ^ a_symm;
                const unsigned m_a_b = sc.c[b_symm][b_spin] - ((sym_prod == 0 && a_spin == b_spin) ? 1u : 0u);
                int orb_idx = (int)rng.choose(m_a_b);
                unsigned unocc2 = 0, symm_idx = 1;
                while (orb_idx >= 0) {
                    unocc2 = T.lookup[b_symm][symm_idx] + b_spin * n_orb;
                    if (!fr_bit(det, unocc2) && unocc2 != unocc1) orb_idx -= 1;
                    symm_idx += 1;
                }
                const unsigned m_b_a = sc.c[a_symm][a_spin] - ((sym_prod == 0 && a_spin == b_spin) ? 1u : 0u);
                const double prob = 2. / n_elec / (n_elec - 1) / n_allow * (1. / m_a_b + 1. / m_b_a);
                const unsigned o1 = orb2, o2 = orb1, u1 = unocc1 < unocc2 ? unocc1 : unocc2, u2 = unocc1 < unocc2 ? unocc2 : unocc1;
                double m = fr_doub_matrel(o1, o2, u1, u2, S.eris, n_orb);
                m *= eps / prob / p_doub;
                double sp = m;
                if (!keep_real || fabs(m) < 0.01) {
                    rng.begin(seed, iter, det, i, FQ_ROUND_D);
                    const int flr = (int)floor(m);
                    sp = (double)(flr + (rng.uni() < m - flr ? 1 : 0));
                }
                if (sp != 0) {
                    sp *= -fr_doub_parity(det, o1, o2, u1, u2) * sign;
                    sp_val = sp;
                    sp_det = (det & ~(1ull << o1) & ~(1ull << o2)) | (1ull << u1) | (1ull << u2);
                }
            }
        }
        else {                  // ---- single (near_uniform.cpp:277-313)
            const uint32_t j = i - n_doub;
            unsigned m_allow[64], delta_s = 0, e = 0;
            for (det_t b = det; b; b &= b - 1, e++) {
                unsigned o = __ffsll((long long)b) - 1;
                unsigned na = sc.c[T.irrep[o % n_orb]][e / (n_elec / 2)];
                m_allow[e] = na;
                if (na == 0) delta_s++;
            }
            rng.begin(seed, iter, det, j, FQ_SING);
            unsigned elec = 0, na = 0;
            while (na == 0) { elec = rng.choose(n_elec); na = m_allow[elec]; }
            const unsigned occ_orb = fr_nth_bit(det, elec), occ_symm = T.irrep[occ_orb % n_orb], spin = occ_orb / n_orb;
            int symm_idx = -1;
            unsigned orbital = 0;
            while (symm_idx == -1) {
                symm_idx = (int)rng.choose(T.lookup[occ_symm][0]);
                orbital = spin * n_orb + T.lookup[occ_symm][symm_idx + 1];
                if (fr_bit(det, orbital)) symm_idx = -1;
            }
            const double prob = 1. / m_allow[elec] / (n_elec - delta_s);
            double m = fr_sing_matrel(det, occ_orb, orbital, S.h_core, S.eris, n_orb);
            if (Q.multi == 1) {         // frimulti_mol.cpp:365-374
                if (fabs(m) > 1e-9) {
                    m *= -eps / prob / (1 - p_doub) / n_walk * cur_f * fr_sing_parity(det, occ_orb, orbital) / colw;
                    sp_val = m;
                    sp_det = (det & ~(1ull << occ_orb)) | (1ull << orbital);
                }
            }
            else {
            m *= eps / prob / (1 - p_doub);
            double sp = m;
            if (!keep_real || fabs(m) < 0.01) {
                rng.begin(seed, iter, det, j, FQ_ROUND_S);
                const int flr = (int)floor(m);
                sp = (double)(flr + (rng.uni() < m - flr ? 1 : 0));
            }
            if (sp != 0) {
                sp *= -fr_sing_parity(det, occ_orb, orbital) * sign;
                sp_val = sp;
                sp_det = (det & ~(1ull << occ_orb)) | (1ull << orbital);
            }
            }
        }
        Q.sp_val[a] = sp_val; Q.sp_det[a] = sp_det; Q.sp_ini[a] = (uint8_t)sp_ini;
    }
    uint32_t bc = fr_block_sum_u32(sp_val != 0 ? 1u : 0u, shu);
    if (threadIdx.x == 0) Q.blk_sp[blockIdx.x] = bc;
}

// ordered compaction of the non-zero spawns into the spawn list (the order of the reference's add() calls)
__global__ void __launch_bounds__(FR_BLOCK) k_fq_compact(FqWork Q, SpawnBuf S) {
    __shared__ uint32_t shu[4];
    const uint32_t A = Q.totals[0];
    const unsigned nblk = (A + FR_BLOCK - 1) / FR_BLOCK;
    if (nblk == 0) { if (blockIdx.x == 0 && threadIdx.x == 0) *S.n_spawn = 0; return; }
    if (blockIdx.x >= nblk) return;
    uint32_t off;
    { uint32_t x = 0; for (unsigned i = threadIdx.x; i < blockIdx.x; i += blockDim.x) x += Q.blk_sp[i]; off = fr_block_sum_u32(x, shu); }
    __syncthreads();
    const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t f = (a < A && Q.sp_val[a] != 0) ? 1u : 0u;
    uint32_t tot;
    uint32_t incl = fr_block_scan_u32(f, shu, &tot);
    if (f) { uint32_t o = off + incl - 1; if (o < S.cap) { S.det[o] = Q.sp_det[a]; S.val[o] = Q.sp_val[a]; S.ini[o] = Q.sp_ini[a]; } }
    if (blockIdx.x == nblk - 1 && threadIdx.x == FR_BLOCK - 1) *S.n_spawn = off + incl;
}

// the post-death walker numbers become the stored values (:403) and |walkers| is summed for the shift update (:416-417)
__global__ void __launch_bounds__(FR_BLOCK) k_fq_apply(VecDev V, FqWork Q) {
    const uint32_t n = V.st->curr_size;
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d < n) V.v0[d] = Q.new_val[d];
}
__global__ void __launch_bounds__(FR_BLOCK) k_fq_norm(VecDev V, double *out) {
    __shared__ double shd[4];
    const uint32_t n = V.st->curr_size;
    double acc = 0;      // integers: any summation order is exact
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) acc += fabs(V.v0[i]);
    double r = fr_block_sum(acc, shd);
    if (threadIdx.x == 0) atomicAdd(out, r);
}

// fciqmc_fp_mol.cpp:428-441: every |v| < 1 becomes -1, 0 or 1; what became 0 is deleted
__global__ void __launch_bounds__(FR_BLOCK) k_fq_fp_round(VecDev V, uint8_t *del, unsigned long long seed, unsigned long long iter) {
    const uint32_t n = V.st->curr_size;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = V.v0[i];
    if (v == 0 || !(fabs(v) < 1)) return;
    FqRng r; r.begin(seed, iter, V.dets[i], 0, FQ_COMP);
    const int flr = (int)floor(v);
    const int nv = flr + (r.uni() < v - flr ? 1 : 0);
    V.v0[i] = (double)nv;
    if (nv == 0) del[i] = 1;
}

void fr_fq_setup(FriesCtx *c, const fries_fciqmc_params *p) {
    if (!c->d_eris) throw FriesError("fries_set_molecule must be called first");
    if (p->max_dets == 0 || p->target_walkers == 0) throw FriesError("max_dets and target_walkers must be positive");
    c->fq = *p; c->fq_mode = true;
    c->eps = p->epsilon; c->target_norm = p->target_walkers; c->en_shift = 0; c->last_one_norm = 0; c->iterat = 0;
    c->mt.seed(p->seed);
    c->proc_scr.resize(2 * c->n_orb); c->vec_scr.resize(2 * c->n_orb);
    if (c->in_proc_scr.size() == c->proc_scr.size()) c->proc_scr = c->in_proc_scr;     // --load_dir: load_proc_hash (fciqmc_mol.cpp:122-124)
    else for (auto &x : c->proc_scr) x = c->mt();   // fciqmc_mol.cpp:126-128
    for (auto &x : c->vec_scr) x = c->mt();         // :134-136
    if (!c->comm.small_send) { c->own_small = fr_alloc<uint8_t>(2048); c->comm.small_send = c->own_small; }
    uint32_t spawn_length = p->target_walkers * 2;      // what a rank can spawn and receive; the reference's Adder holds
    c->adder_cap = p->target_walkers / c->n_ranks / c->n_ranks * 2;      // target / n_procs^2 * 2 per destination (:107)
    if (!c->d_proc_scr) c->d_proc_scr = fr_alloc<uint32_t>(64);
    FR_HIP(hipMemcpyAsync(c->d_proc_scr, c->proc_scr.data(), 4 * c->proc_scr.size(), hipMemcpyHostToDevice, c->stream));
    c->hf_proc = fr_host_idx_to_proc(c, c->hf_det);
    fr_vec_alloc(c, &c->vec, p->max_dets);
    fr_spawn_alloc(c, spawn_length + 4096);
    fr_xch_alloc(c, spawn_length + 4096);
    fr_vcomp_alloc(c, p->max_dets);                 // dots
    FqWork &Q = c->fqw;
    Q.cap_d = p->max_dets; Q.cap_a = spawn_length + 4096;
    Q.n_doub = fr_alloc<uint32_t>(Q.cap_d); Q.n_att = fr_alloc<uint32_t>(Q.cap_d); Q.att_off = fr_alloc<uint32_t>(Q.cap_d); Q.new_val = fr_alloc<double>(Q.cap_d);
    unsigned nb = fr_blocks(Q.cap_d, FR_BLOCK) + 1, nba = fr_blocks(Q.cap_a, FR_BLOCK) + 1;
    Q.blk_att = fr_alloc<uint32_t>(nb); Q.blk_nz = fr_alloc<uint32_t>(nb); Q.blk_ini = fr_alloc<uint32_t>(nb); Q.blk_sp = fr_alloc<uint32_t>(nba);
    Q.totals = fr_alloc<uint32_t>(4); Q.norm = fr_alloc<double>(1); Q.heavy = fr_alloc<uint32_t>(Q.cap_d);
    FR_HIP(hipMemsetAsync(Q.totals, 0, 16, c->stream));
    Q.sp_val = fr_alloc<double>(Q.cap_a); Q.sp_det = fr_alloc<det_t>(Q.cap_a); Q.sp_ini = fr_alloc<uint8_t>(Q.cap_a);
    Q.o1cnt = p->heat_bath ? fr_alloc<uint32_t>((size_t)Q.cap_d * c->n_elec) : nullptr;
    Q.multi = 0; Q.n_walk = nullptr; Q.samp_unit = 1; Q.init_f = 0;
    if (p->real_walkers) {             // fciqmc_fp_mol
        if (c->use_comm && c->n_ranks > 1 && 2 * c->n_orb > 63) throw FriesError("fciqmc_fp_mol over ranks needs bit 63 of the index for the initiator flag (at most 31 orbitals)");
        c->dots_slot0_from_hf = true;      // fciqmc_fp_mol.cpp:461-462
        Q.multi = 2; Q.n_walk = fr_alloc<uint32_t>(Q.cap_d);
    }
    if (p->heat_bath && (c->n_elec > 32 || c->n_orb - c->n_elec / 2 > 32)) throw FriesError("heat-bath sampling supports at most 32 electrons / 32 virtual orbitals per spin");
    fr_h_trial_setup(c);        // HF trial vector, H * trial, p_doub (:139-191, as in frisys_mol)
    if (!c->in_ini_det.empty()) {                            // --ini_vec (:226-237): entries add()ed in file order from rank 0; integer walkers (the reader
        std::vector<det_t> d; std::vector<double> v;         // fills an int array), real values in fciqmc_fp_mol.cpp:233-246 and frimulti_mol.cpp:205-215
        const bool real_valued = p->real_walkers || c->fq_ini_real;
        for (size_t i = 0; i < c->in_ini_det.size(); i++) {
            const double w = real_valued ? c->in_ini_val[i] : (double)(int)c->in_ini_val[i];
            if (w != 0 && fr_host_idx_to_proc(c, c->in_ini_det[i]) == c->rank) { d.push_back(c->in_ini_det[i]); v.push_back(w); }
        }
        uint32_t m = (uint32_t)d.size();
        if (m >= c->adder_cap) throw FriesError("the initial vector fills the Adder (the reference would store the filling entry twice): raise --target");
        if (m) {
            std::vector<uint8_t> f(m, 1);
            FR_HIP(hipMemcpyAsync(c->sp.det, d.data(), 8 * (size_t)m, hipMemcpyHostToDevice, c->stream));
            FR_HIP(hipMemcpyAsync(c->sp.val, v.data(), 8 * (size_t)m, hipMemcpyHostToDevice, c->stream));
            FR_HIP(hipMemcpyAsync(c->sp.ini, f.data(), m, hipMemcpyHostToDevice, c->stream));
            FR_HIP(hipMemcpyAsync(c->sp.n_spawn, &m, 4, hipMemcpyHostToDevice, c->stream));
            fr_vec_merge(c, &c->vec, m, true);
            FR_HIP(hipStreamSynchronize(c->stream));
        }
    }
    else if (c->rank == c->hf_proc) {                        // :239-243
        double v = 100; uint8_t one = 1; uint32_t n1 = 1;
        FR_HIP(hipMemcpyAsync(c->sp.det, &c->hf_det, 8, hipMemcpyHostToDevice, c->stream));
        FR_HIP(hipMemcpyAsync(c->sp.val, &v, 8, hipMemcpyHostToDevice, c->stream));
        FR_HIP(hipMemcpyAsync(c->sp.ini, &one, 1, hipMemcpyHostToDevice, c->stream));
        FR_HIP(hipMemcpyAsync(c->sp.n_spawn, &n1, 4, hipMemcpyHostToDevice, c->stream));
        fr_vec_merge(c, &c->vec, 1, true);
    }
    fr_vec_sync_state(c, &c->vec, &c->h_vst);
}

void fr_fq_iterate(FriesCtx *c, fries_fciqmc_log *lg) {
    hipStream_t st = c->stream;
    const fries_fciqmc_params &P = c->fq;
    FqWork &Q = c->fqw;
    fr_vec_sync_state(c, &c->vec, &c->h_vst);
    fr_vec_maybe_rebuild(c, &c->vec);
    const uint32_t n = c->h_vst.curr_size;
    if (n > Q.cap_d) throw FriesError("vector larger than the FCIQMC work arrays");
    SysDev S; S.n_orb = c->n_orb; S.n_elec = c->n_elec; S.h_core = c->d_h; S.eris = c->d_eris; S.hb = c->d_hb; S.hf_en = c->hf_en; S.spin_parity = c->spin_parity;
    const unsigned gd = fr_blocks(n ? n : 1, FR_BLOCK);
    FR_HIP(hipMemsetAsync(&Q.totals[3], 0, 4, st));
    FR_LAUNCH(c, "k_fq_count", k_fq_count, dim3(gd), dim3(FR_BLOCK), c->vec, S, Q, (unsigned long long)P.seed, (unsigned long long)c->iterat, c->p_doub, c->eps, c->en_shift, P.initiator);
    FR_LAUNCH(c, "k_fq_heavy_a", k_fq_heavy_a, dim3(128, FQ_HSLICES), dim3(FR_BLOCK), c->vec, S, Q, (unsigned long long)P.seed, (unsigned long long)c->iterat, c->p_doub, c->eps, c->en_shift);
    FR_LAUNCH(c, "k_fq_heavy_b", k_fq_heavy_b, dim3(128, FQ_HSLICES), dim3(FR_BLOCK), c->vec, S, Q, (unsigned long long)P.seed, (unsigned long long)c->iterat, c->eps, c->en_shift);
    FR_LAUNCH(c, "k_fq_blocksum", k_fq_blocksum, dim3(gd), dim3(FR_BLOCK), c->vec, Q);
    FR_LAUNCH(c, "k_fq_offsets", k_fq_offsets, dim3(gd), dim3(FR_BLOCK), c->vec, Q);
    uint32_t tot[3];
    FR_HIP(hipMemcpyAsync(tot, Q.totals, 12, hipMemcpyDeviceToHost, st));
    FR_HIP(hipStreamSynchronize(st));
    const uint32_t A = tot[0];
    if (A > Q.cap_a) throw FriesError("Insufficient memory allocated in adder");       // more attempts than 2 x target walkers
    const unsigned ga = fr_blocks(A ? A : 1, FR_BLOCK);
    FR_LAUNCH(c, "k_fq_attempt", k_fq_attempt, dim3(ga), dim3(FR_BLOCK), c->vec, S, Q, (unsigned long long)P.seed, (unsigned long long)c->iterat, c->p_doub, c->eps, P.initiator);
    FR_LAUNCH(c, "k_fq_compact", k_fq_compact, dim3(ga), dim3(FR_BLOCK), Q, c->sp);
    FR_LAUNCH(c, "k_fq_apply", k_fq_apply, dim3(gd), dim3(FR_BLOCK), c->vec, Q);
    uint32_t n_spawn = 0;
    FR_HIP(hipMemcpyAsync(&n_spawn, c->sp.n_spawn, 4, hipMemcpyDeviceToHost, st));
    FR_HIP(hipStreamSynchronize(st));
    uint32_t n_merge = n_spawn;
    if (c->use_comm) n_merge = fr_spawn_exchange(c, n_spawn, Q.multi == 2 ? 2 : 1);  // one all-to-all per iteration; the spawns keep their order (:413)
    else if (n_spawn >= c->adder_cap) throw FriesError("Insufficient memory allocated in adder");
    if (n_merge) fr_vec_merge(c, &c->vec, n_merge, true);           // one perform_add into the column itself (:413)
    fr_vec_sync_state(c, &c->vec, &c->h_vst);
    if (c->h_vst.err) throw FriesError("device error in the FCIQMC merge (capacity, hash table or electron count)");
    if (Q.multi == 2) {
        const uint32_t nb = c->h_vst.curr_size ? c->h_vst.curr_size : 1;
        FR_LAUNCH(c, "k_fq_fp_round", k_fq_fp_round, dim3(fr_blocks(nb, FR_BLOCK)), dim3(FR_BLOCK), c->vec, c->vc.del, (unsigned long long)P.seed, (unsigned long long)c->iterat);
        fr_vec_delete_flagged(c, &c->vec, c->vc.del, nb);
        fr_vec_sync_state(c, &c->vec, &c->h_vst);
    }
    double norm = 0;
    if ((c->iterat + 1) % 10 == 0 && Q.multi == 2) {                // real values: the reference's in-order sum, exactly (local_norm, vec_utils.hpp:683-689)
        norm = fr_abs_norm(c);
        if (c->use_comm) {          // sum_mpi of the local norms, in rank order
            FR_HIP(hipMemcpyAsync(c->comm.small_send, &norm, 8, hipMemcpyHostToDevice, st));
            const double *all = (const double *)fr_allgather(c, 8);
            double h[FR_MAX_RANKS];
            FR_HIP(hipMemcpyAsync(h, all, 8 * (size_t)c->n_ranks, hipMemcpyDeviceToHost, st));
            FR_HIP(hipStreamSynchronize(st));
            norm = 0;
            for (int q = 0; q < c->n_ranks; q++) norm += h[q];
        }
        double damp = 0.05 / c->eps / 10;
        if (c->last_one_norm) { c->en_shift -= damp * log(norm / c->last_one_norm); c->last_one_norm = norm; }
        if (c->last_one_norm == 0 && norm > c->target_norm) c->last_one_norm = norm;
    }
    else if ((c->iterat + 1) % 10 == 0) {                           // :415-427
        FR_HIP(hipMemsetAsync(Q.norm, 0, 8, st));
        unsigned gn = fr_blocks(c->h_vst.curr_size ? c->h_vst.curr_size : 1, FR_BLOCK);
        if (gn > 1024) gn = 1024;
        FR_LAUNCH(c, "k_fq_norm", k_fq_norm, dim3(gn), dim3(FR_BLOCK), c->vec, Q.norm);
        if (c->use_comm) {          // sum_mpi of the local walker numbers, in rank order (:417)
            FR_HIP(hipMemcpyAsync(c->comm.small_send, Q.norm, 8, hipMemcpyDeviceToDevice, st));
            const double *all = (const double *)fr_allgather(c, 8);
            double h[FR_MAX_RANKS];
            FR_HIP(hipMemcpyAsync(h, all, 8 * (size_t)c->n_ranks, hipMemcpyDeviceToHost, st));
            FR_HIP(hipStreamSynchronize(st));
            for (int q = 0; q < c->n_ranks; q++) norm += h[q];
        }
        else {
            FR_HIP(hipMemcpyAsync(&norm, Q.norm, 8, hipMemcpyDeviceToHost, st));
            FR_HIP(hipStreamSynchronize(st));
        }
        double damp = 0.05 / c->eps / 10;
        if (c->last_one_norm) { c->en_shift -= damp * log(norm / c->last_one_norm); c->last_one_norm = norm; }
        if (c->last_one_norm == 0 && norm > c->target_norm) c->last_one_norm = norm;
    }
    fr_dots(c, &c->numer, &c->denom);
    c->iterat++;
    c->tot_iters++; c->tot_spawns += n_spawn;
    if (lg) {
        lg->numer = c->numer; lg->denom = c->denom; lg->shift = c->en_shift; lg->norm = norm;
        lg->n_nonz = (int32_t)tot[1]; lg->n_ini = tot[2]; lg->curr_size = c->h_vst.curr_size; lg->n_spawn = n_spawn;
        uint32_t e = 0;
        FR_HIP(hipMemcpy(&e, c->d_err, 4, hipMemcpyDeviceToHost));
        lg->err = e | c->h_vst.err; lg->n_attempts = A;
    }
}


// ------------------------------------------------------------------ frimulti_mol (FRIES_bin/frimulti_mol.cpp:84-425), --distribution HB, one rank
void fr_multi_setup(FriesCtx *c, const fries_frimulti_params *p) {
    if (c->use_comm && c->n_ranks > 1 && 2 * c->n_orb > 63) throw FriesError("frimulti_mol over ranks needs bit 63 of the index for the initiator flag (at most 31 orbitals)");
    if (p->vec_nonz == 0 || p->mat_nonz < 10 || p->max_dets == 0) throw FriesError("vec_nonz, max_dets must be positive and mat_nonz at least 10 (the first iterations use a tenth of it)");
    // --trial_vec: frimulti_mol.cpp:149-157 throws when an add() reports a full Adder, and trial_vec's Adder holds exactly n_trial entries: on one rank the
    // reference refuses EVERY trial file with this message (over several ranks its non-root ranks build zero-sized vectors, io_utils.cpp:410-444, and abort
    // inside MPI).  Same behaviour here; the HF trial vector is the one this driver can use.
    if (!c->in_trial_det.empty()) throw FriesError("Insufficient memory allocated in adder");
    c->fq_ini_real = true;              // --ini_vec: real values (:205-215)
    fries_fciqmc_params q{};
    q.epsilon = p->epsilon; q.target_walkers = p->mat_nonz; q.initiator = 0; q.max_dets = p->max_dets; q.seed = p->seed; q.heat_bath = 1;
    fr_fq_setup(c, &q);                 // scramblers, vector, work arrays, H * trial, 100 x HF -- as there (frimulti_mol.cpp:84-233)
    c->fm = *p;
    c->target_norm = p->target_norm; c->init_thresh = p->initiator; c->vec_nonz = p->vec_nonz; c->mat_nonz = p->mat_nonz;
    c->adder_cap = p->mat_nonz * 2 / c->n_ranks / c->n_ranks;     // :89
    FqWork &Q = c->fqw;
    Q.multi = 1; Q.n_walk = fr_alloc<uint32_t>(Q.cap_d); Q.init_f = p->initiator; Q.samp_unit = 1;
    if (!c->W.kin) c->W.kin = fr_alloc<uint32_t>(p->max_dets);          // fr_sys_comp's tooth indices
    if (!c->d_norms_keep) { c->d_norms_keep = fr_alloc<double>(FR_MAX_RANKS); c->d_seq_scratch = fr_alloc<double>(1); }
    c->glob_norm = -1;                  // no compression yet: the first comb is spaced by the norm of the start vector (:227-233)
}

void fr_multi_iterate(FriesCtx *c, fries_fciqmc_log *lg) {
    hipStream_t st = c->stream;
    const fries_frimulti_params &P = c->fm;
    FqWork &Q = c->fqw;
    fr_vec_sync_state(c, &c->vec, &c->h_vst);
    fr_vec_maybe_rebuild(c, &c->vec);
    const uint32_t n = c->h_vst.curr_size;
    if (n > Q.cap_d) throw FriesError("vector larger than the work arrays");
    SysDev S; S.n_orb = c->n_orb; S.n_elec = c->n_elec; S.h_core = c->d_h; S.eris = c->d_eris; S.hb = c->d_hb; S.hf_en = c->hf_en; S.spin_parity = c->spin_parity;
    // samples per column (:301-322)
    double rn_sys = c->mt() / (1. + UINT32_MAX);
    const uint32_t curr_mat_samp = c->iterat < 10 ? P.mat_nonz / 10 : P.mat_nonz;
    fr_multi_walks(c, rn_sys, c->glob_norm, curr_mat_samp, Q.n_walk, c->vc.dots);
    const unsigned gd = fr_blocks(n ? n : 1, FR_BLOCK);
    FR_HIP(hipMemsetAsync(&Q.totals[3], 0, 4, st));
    const unsigned long long seed = c->fq.seed, iter = c->iterat;
    FR_LAUNCH(c, "k_fq_count", k_fq_count, dim3(gd), dim3(FR_BLOCK), c->vec, S, Q, seed, iter, c->p_doub, c->eps, c->en_shift, 0u);
    FR_LAUNCH(c, "k_fq_heavy_a", k_fq_heavy_a, dim3(128, FQ_HSLICES), dim3(FR_BLOCK), c->vec, S, Q, seed, iter, c->p_doub, c->eps, c->en_shift);
    FR_LAUNCH(c, "k_fq_heavy_b", k_fq_heavy_b, dim3(128, FQ_HSLICES), dim3(FR_BLOCK), c->vec, S, Q, seed, iter, c->eps, c->en_shift);
    FR_LAUNCH(c, "k_fq_blocksum", k_fq_blocksum, dim3(gd), dim3(FR_BLOCK), c->vec, Q);
    FR_LAUNCH(c, "k_fq_offsets", k_fq_offsets, dim3(gd), dim3(FR_BLOCK), c->vec, Q);
    uint32_t tot[3];
    double unit_T[2];
    FR_HIP(hipMemcpyAsync(tot, Q.totals, 12, hipMemcpyDeviceToHost, st));
    FR_HIP(hipMemcpyAsync(unit_T, c->vc.dots, 16, hipMemcpyDeviceToHost, st));
    FR_HIP(hipStreamSynchronize(st));
    const uint32_t A = tot[0];
    if (A > Q.cap_a) throw FriesError("Insufficient memory allocated in adder");
    FqWork Qa = Q;
    Qa.samp_unit = unit_T[0];
    const unsigned ga = fr_blocks(A ? A : 1, FR_BLOCK);
    FR_LAUNCH(c, "k_fq_attempt", k_fq_attempt, dim3(ga), dim3(FR_BLOCK), c->vec, S, Qa, seed, iter, c->p_doub, c->eps, 0u);
    FR_LAUNCH(c, "k_fq_compact", k_fq_compact, dim3(ga), dim3(FR_BLOCK), Q, c->sp);
    FR_LAUNCH(c, "k_fq_apply", k_fq_apply, dim3(gd), dim3(FR_BLOCK), c->vec, Q);
    uint32_t n_spawn = 0;
    FR_HIP(hipMemcpyAsync(&n_spawn, c->sp.n_spawn, 4, hipMemcpyDeviceToHost, st));
    FR_HIP(hipStreamSynchronize(st));
    uint32_t n_merge = n_spawn;
    if (c->use_comm) n_merge = fr_spawn_exchange(c, n_spawn, 2);     // one all-to-all per iteration, arrival order = (source rank, add order)
    else if (n_spawn >= c->adder_cap) throw FriesError("Insufficient memory allocated in adder.");
    if (n_merge) fr_vec_merge(c, &c->vec, n_merge, true);           // perform_add(0) into the column itself (:382)
    fr_vec_sync_state(c, &c->vec, &c->h_vst);
    if (c->h_vst.err) throw FriesError("device error in the merge (capacity, hash table or electron count)");
    // compression (:385-421)
    fr_abs_sums(c);
    uint32_t n_samp = P.vec_nonz;
    double glob_norm = 0;
    fr_find_preserve(c, &n_samp, &glob_norm);
    c->glob_norm = glob_norm;
    c->nkept = P.vec_nonz - n_samp;
    if ((c->iterat + 1) % 10 == 0) {
        double damp = 0.05 / 10 / c->eps;
        if (c->last_one_norm) { c->en_shift -= damp * log(glob_norm / c->last_one_norm); c->last_one_norm = glob_norm; }
        if (c->last_one_norm == 0 && glob_norm > c->target_norm) c->last_one_norm = glob_norm;
    }
    fr_dots(c, &c->numer, &c->denom);
    rn_sys = c->mt() / (1. + UINT32_MAX);
    fr_sys_comp(c, n_samp, rn_sys);        // incl. the deletes; the reference's HF test compares addresses (:417), so HF goes like any other
    c->iterat++;
    c->tot_iters++; c->tot_spawns += n_spawn;
    if (lg) {
        fr_vec_sync_state(c, &c->vec, &c->h_vst);
        lg->numer = c->numer; lg->denom = c->denom; lg->shift = c->en_shift; lg->norm = glob_norm;
        lg->n_nonz = c->h_vst.n_nonz; lg->n_ini = tot[2]; lg->curr_size = c->h_vst.curr_size; lg->n_spawn = n_spawn;
        uint32_t e = 0;
        FR_HIP(hipMemcpy(&e, c->d_err, 4, hipMemcpyDeviceToHost));
        lg->err = e | c->h_vst.err; lg->n_attempts = A;
    }
}
