// frisys_hh on the device (FRIES_bin/frisys_hh.cpp:27-380): FRI with systematic matrix compression for the 1-D
// Hubbard-Holstein model, built from the same compression / merge / vector kernels as frisys_mol.
// Index = [alpha sites | beta sites | 3 bits per phonon] (hh_vec.hpp, hub_holstein.cpp:139-171), n_sites <= 12.
#include "ctx.hpp"

__device__ __forceinline__ unsigned fr_hh_ph(det_t d, unsigned L, unsigned site) { return (unsigned)(d >> (2 * L + FR_HH_PH_BITS * site)) & ((1u << FR_HH_PH_BITS) - 1u); }
__device__ __forceinline__ unsigned fr_hh_tot_ph(det_t d, unsigned L) { unsigned t = 0; for (unsigned s = 0; s < L; s++) t += fr_hh_ph(d, L, s); return t; }
// hub_diag (hub_holstein.cpp:101-136): doubly occupied sites
__device__ __forceinline__ unsigned fr_hub_diag(det_t d, unsigned L) { return (unsigned)__popcll(d & (d >> L) & ((1ull << L) - 1ull)); }

// Spawn of sample e of stage 2 (frisys_hh.cpp:243-290): new index and signed matrix element; 0 = dropped.
__global__ void __launch_bounds__(FR_BLOCK) k_hh_eval(CompWork W, VecDev V, int prev, double eps, double *f_val, det_t *f_det, uint32_t *pcnt) {
    __shared__ uint32_t shu[4];
    const unsigned n_in = W.state[FR_MAX_ROUNDS + 1].n_out;
    const unsigned nblk = (n_in + FR_TILE - 1) / FR_TILE;
    if (blockIdx.x >= nblk) return;
    const StageElems P = W.el[prev];
    const unsigned L = V.hh_sites, n_elec = V.hh_nelec;
    size_t base = (size_t)blockIdx.x * FR_TILE + threadIdx.x;
    uint32_t cnt = 0;
    for (int it = 0; it < FR_ITEMS; it++) {
        size_t e = base + (size_t)it * FR_BLOCK;
        if (e >= n_in) break;
        uint32_t wi = W.e_wi[e], exc = W.e_sub[e] & 0xffu;       // the reference narrows the excitation index to 8 bits (:245)
        double el = W.e_val[e] * -eps;
        uint32_t pos = P.pos[wi], ph_ex = P.code[wi];
        double cur_val = V.v0[pos];
        if (cur_val < 0) el *= -1;
        det_t cur = V.dets[pos], nd = cur;
        if (ph_ex) {
            unsigned orb = fr_nth_bit(cur & ((1ull << (2 * L)) - 1ull), exc % n_elec);
            unsigned site = orb % L;
            unsigned pn = fr_hh_ph(cur, L, site);
            const unsigned sh = 2 * L + FR_HH_PH_BITS * site;
            if (exc < n_elec && pn > 0) { nd = cur - (1ull << sh); el *= sqrt((double)pn); }
            else if (exc >= n_elec && pn + 1 < (1u << FR_HH_PH_BITS)) { nd = cur + (1ull << sh); el *= sqrt((double)(pn + 1)); }
            else el = 0;
        }
        else {
            const det_t El = cur & ((1ull << (2 * L)) - 1ull);
            det_t r0 = El & ~(El >> 1); r0 &= ~(1ull << (L - 1)); r0 &= ~(1ull << (2 * L - 1));
            det_t r1 = El & (~El << 1); r1 &= ~(1ull << L);
            unsigned n0 = (unsigned)__popcll(r0);
            unsigned orig, dest;
            if (exc < n0) { orig = fr_nth_bit(r0, exc); dest = orig + 1; }
            else { orig = fr_nth_bit(r1, exc - n0); dest = orig - 1; }
            nd = (cur & ~(1ull << orig)) | (1ull << dest);
            el *= -1;       // hub_t
        }
        if (!(fabs(el) > 1e-9)) el = 0;
        f_val[e] = el; f_det[e] = nd;
        cnt += (el != 0);
    }
    uint32_t bc = fr_block_sum_u32(cnt, shu);
    if (threadIdx.x == 0) pcnt[blockIdx.x] = bc;
}

// ordered compaction of the surviving spawns into the spawn list
__global__ void __launch_bounds__(FR_BLOCK) k_hh_compact(CompWork W, VecDev V, SpawnBuf S, int prev, const double *f_val, const det_t *f_det, const uint32_t *pcnt, double init_thresh) {
    __shared__ uint32_t shu[4];
    const unsigned n_in = W.state[FR_MAX_ROUNDS + 1].n_out;
    const unsigned nblk = (n_in + FR_TILE - 1) / FR_TILE;
    if (blockIdx.x >= nblk) { if (blockIdx.x == 0 && threadIdx.x == 0) *S.n_spawn = 0; return; }
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
        if (f[it]) {
            uint32_t pos = P.pos[W.e_wi[e]];
            S.det[o] = f_det[e]; S.val[o] = f_val[e]; S.ini[o] = fabs(V.v0[pos]) >= init_thresh;
            o++;
        }
    }
    if (blockIdx.x == nblk - 1 && threadIdx.x == FR_BLOCK - 1) *S.n_spawn = o;
}

// frifull_hh (FRIES_bin/frifull_hh.cpp:187-263): the off-diagonal part of the Hamiltonian applied in full.  The adds of one
// stored state, in the reference's order: hops to the right, hops to the left (eps * t * value), then for every spin-up electron the
// phonon moves -1 / +1 on its site (twice the coupling on a doubly occupied site), then the same for the spin-down electrons on
// sites without a spin-up one.  Returns the number of adds written (zero values never reach the Adder, vec_utils.hpp:418-431);
// *tried counts what the reference's num_added counts.
template <bool WRITE>
__device__ __forceinline__ uint32_t fr_hhf_emit(det_t cur, double cur_el, unsigned L, unsigned n_elec, double eps, double g, det_t *o_det, double *o_val, uint32_t *tried) {
    uint32_t n = 0, t = 0;
    const det_t El = cur & ((1ull << (2 * L)) - 1ull);
    det_t r0 = El & ~(El >> 1); r0 &= ~(1ull << (L - 1)); r0 &= ~(1ull << (2 * L - 1));
    det_t r1 = El & (~El << 1); r1 &= ~(1ull << L); r1 &= ~1ull;
    const double hop = eps * 1.0 * cur_el;
    for (det_t m = r0; m; m &= m - 1) {
        unsigned o = (unsigned)__ffsll((long long)m) - 1;
        t++;
        if (hop != 0) { if (WRITE) { o_det[n] = (cur & ~(1ull << o)) | (1ull << (o + 1)); o_val[n] = hop; } n++; }
    }
    for (det_t m = r1; m; m &= m - 1) {
        unsigned o = (unsigned)__ffsll((long long)m) - 1;
        t++;
        if (hop != 0) { if (WRITE) { o_det[n] = (cur & ~(1ull << o)) | (1ull << (o - 1)); o_val[n] = hop; } n++; }
    }
    const det_t up = El & ((1ull << L) - 1ull), dn = El >> L;
    for (int sp = 0; sp < 2; sp++) {
        for (det_t m = sp ? (dn & ~up) : up; m; m &= m - 1) {
            unsigned site = (unsigned)__ffsll((long long)m) - 1;
            unsigned pn = fr_hh_ph(cur, L, site);
            const unsigned sh = 2 * L + FR_HH_PH_BITS * site;
            const int mult = sp ? 1 : (int)((dn >> site) & 1) + 1;
            if (pn > 0) {
                double v = sp ? -eps * g * sqrt((double)pn) * cur_el : -eps * g * sqrt((double)pn) * mult * cur_el;
                t++;
                if (v != 0) { if (WRITE) { o_det[n] = cur - (1ull << sh); o_val[n] = v; } n++; }
            }
            if (pn + 1 < (1u << FR_HH_PH_BITS)) {
                double v = sp ? -eps * g * sqrt((double)(pn + 1)) * cur_el : -eps * g * sqrt((double)(pn + 1)) * mult * cur_el;
                t++;
                if (v != 0) { if (WRITE) { o_det[n] = cur + (1ull << sh); o_val[n] = v; } n++; }
            }
        }
    }
    if (tried) *tried = t;
    return n;
}

__global__ void __launch_bounds__(FR_BLOCK) k_hhf_count(VecDev V, double eps, double g, uint32_t *pcnt, unsigned long long *totals) {
    __shared__ uint32_t shu[4];
    const uint32_t n = V.st->curr_size;
    const unsigned L = V.hh_sites, n_elec = V.hh_nelec;
    size_t base = (size_t)blockIdx.x * FR_TILE + (size_t)threadIdx.x * FR_ITEMS;
    uint32_t cnt = 0, tried = 0;
    for (int it = 0; it < FR_ITEMS; it++) {
        size_t i = base + it;
        if (i >= n) break;
        double v = V.v0[i];
        if (v == 0) continue;
        uint32_t t;
        cnt += fr_hhf_emit<false>(V.dets[i], v, L, n_elec, eps, g, nullptr, nullptr, &t);
        tried += t;
    }
    uint32_t bc = fr_block_sum_u32(cnt, shu);
    uint32_t bt = fr_block_sum_u32(tried, shu);
    if (threadIdx.x == 0) { pcnt[blockIdx.x] = bc; if (bt) atomicAdd(&totals[0], (unsigned long long)bt); if (bc) atomicAdd(&totals[1], (unsigned long long)bc); }
}

__global__ void __launch_bounds__(FR_BLOCK) k_hhf_write(VecDev V, SpawnBuf S, double eps, double g, double init_thresh, const uint32_t *pcnt) {
    __shared__ uint32_t shu[4];
    const uint32_t n = V.st->curr_size;
    const unsigned L = V.hh_sites, n_elec = V.hh_nelec;
    uint32_t off;
    { uint32_t x = 0; for (unsigned i = threadIdx.x; i < blockIdx.x; i += blockDim.x) x += pcnt[i]; off = fr_block_sum_u32(x, shu); }
    size_t base = (size_t)blockIdx.x * FR_TILE + (size_t)threadIdx.x * FR_ITEMS;
    uint32_t tsum = 0;
    for (int it = 0; it < FR_ITEMS; it++) {
        size_t i = base + it;
        if (i >= n) break;
        double v = V.v0[i];
        if (v != 0) tsum += fr_hhf_emit<false>(V.dets[i], v, L, n_elec, eps, g, nullptr, nullptr, nullptr);
    }
    uint32_t tot;
    uint32_t incl = fr_block_scan_u32(tsum, shu, &tot);
    uint32_t o = off + incl - tsum;
    for (int it = 0; it < FR_ITEMS; it++) {
        size_t i = base + it;
        if (i >= n) break;
        double v = V.v0[i];
        if (v == 0) continue;
        uint32_t k = fr_hhf_emit<true>(V.dets[i], v, L, n_elec, eps, g, S.det + o, S.val + o, nullptr);     // (the host checked the total against the capacity)
        const uint8_t ini = fabs(v) > init_thresh;          // strict here (frifull_hh.cpp:201), >= in frisys_hh
        for (uint32_t q = 0; q < k; q++) S.ini[o + q] = ini;
        o += k;
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == FR_BLOCK - 1) *S.n_spawn = o;
}

// v0 <- v0 (1 - eps (U n_double + omega n_phonon - E_ref - S)) for the elements that existed before the merge, v0 += v1
// (frisys_hh.cpp:311-321); publishes per-block sums of |v0| like k_death_clone
__global__ void __launch_bounds__(FR_BLOCK) k_hh_death_clone(VecDev V, VcompBuf B, uint32_t vec_size_before, double eps, double shift, double hub_u, double omega, double hf_en, uint32_t n_samp) {
    __shared__ double shd[12];
    const uint32_t n = V.st->curr_size;
    const unsigned nblk = (n + FR_TILE - 1) / FR_TILE;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        CompState s{};
        s.n_rem = n_samp; s.n_in = n; s.done = 0; s.pbuf = 0;
        B.state[0] = s;
    }
    if (blockIdx.x >= nblk) return;
    const unsigned L = V.hh_sites;
    size_t base = (size_t)blockIdx.x * FR_TILE + threadIdx.x;
    double sum = 0;
    for (int it = 0; it < FR_ITEMS; it++) {
        size_t i = base + (size_t)it * FR_BLOCK;
        if (i >= n) break;
        double v = V.v0[i];
        if (i < vec_size_before && v != 0) {
            const det_t d = V.dets[i];
            double diag_el = (double)fr_hub_diag(d, L);
            double phonon_diag = fr_hh_tot_ph(d, L) * omega;
            v *= 1 - eps * (diag_el * hub_u + phonon_diag - hf_en - shift);
        }
        v += V.v1[i] * 1.0;
        V.v0[i] = v;        // column 1 keeps the spawn sums until the next iteration zeroes it (:227-228): an element that was
                            // spawned onto and then compressed to zero therefore survives the deletes of :355-359 as a zero entry
        sum += fabs(v);
    }
    double bs;
    fr_block_excl_f64(sum, shd, &bs);
    if (threadIdx.x == 0) { B.psum[0][blockIdx.x] = bs; B.pcnt[0][blockIdx.x] = 0; }
}

// calc_ref_ovlp (hub_holstein.hpp:93-186) of this shard: sum over the stored states of their off-diagonal coupling (over t) to
// the Neel state.  Two byte-level details of the reference are behaviour and are kept: the orbital to the right of bit 7 of every byte
// counts as empty (integer promotion of ~byte >> 1, :150), and the open-boundary mask lands in byte ceil(L / 8) (:165-167).
#define FR_HH_OVLP_BLOCKS 256
// (fixed grid, one partial per workgroup, added up in workgroup order by k_hh_ref_ovlp_sum: the same bits whatever the device does)
__global__ void __launch_bounds__(FR_BLOCK) k_hh_ref_ovlp(VecDev V, det_t ref, double g_over_t, double *part) {
    __shared__ double shd[4];
    const uint32_t n = V.st->curr_size;
    const unsigned L = V.hh_sites, n_elec = V.hh_nelec;
    const unsigned nbytes = (2 * L + 7) / 8;
    const det_t emask = (1ull << (2 * L)) - 1ull, byte_mask = (1ull << (8 * nbytes)) - 1ull;
    double acc = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double val = V.v0[i];
        if (val == 0) continue;
        const det_t cur = V.dets[i];
        if (((cur ^ ref) & emask) == 0) {
            unsigned found = 0, site_elecs = 0;
            for (unsigned s = 0; s < L && found < 2; s++) {
                unsigned ph = fr_hh_ph(cur, L, s);
                unsigned n_occ = (unsigned)((ref >> s) & 1ull) + (unsigned)((ref >> (s + L)) & 1ull);
                if (ph > 1 || (ph == 1 && n_occ == 0)) { site_elecs = 0; break; }
                else if (ph == 1) { site_elecs = n_occ; found++; }
            }
            if (found == 2) site_elecs = 0;
            acc -= val * g_over_t * site_elecs;
        }
        else {
            if (fr_hh_tot_ph(cur, L) != 0) continue;
            det_t c = cur & byte_mask, r = ref & byte_mask;
            det_t not_occ = c & ~r;
            det_t ref_left = c & (r >> 1);
            det_t not_occ_left = ((~c & byte_mask) >> 1) | 0x8080808080808080ull;
            det_t ref_right = c & (r << 1);
            det_t not_occ_right = (~c << 1);
            unsigned ob = (L + 7) / 8;
            if (ob < nbytes) ref_left &= ~(1ull << (8 * ob + (L - 1) % 8));
            det_t mask = not_occ & ((ref_left & not_occ_left) | (ref_right & not_occ_right)) & byte_mask;
            if ((2 * L) % 8 != 0) mask &= ~((0xffull << (8 * (nbytes - 1))) & ~emask);
            unsigned n_hop = 0, n_common = 0;
            for (unsigned b = 0; b < nbytes && n_hop <= 1; b++) {
                n_hop += (unsigned)__popcll((mask >> (8 * b)) & 0xffull);
                if (n_hop > 1) break;
                n_common += (unsigned)__popcll(((r & c) >> (8 * b)) & 0xffull);
            }
            if (n_hop == 1 && n_common == n_elec - 1) acc += val;
        }
    }
    double r = fr_block_sum(acc, shd);
    if (threadIdx.x == 0) part[blockIdx.x] = r;
}
__global__ void __launch_bounds__(64) k_hh_ref_ovlp_sum(VecDev V, const double *part, double *out) {
    if (threadIdx.x != 0) return;
    double r = 0;
    for (int k = 0; k < FR_HH_OVLP_BLOCKS; k++) r += part[k];
    out[0] = r; out[1] = V.v0[0]; out[2] = (double)fr_hub_diag(V.dets[0], V.hh_sites);
}

// sys_comp zeroes elements and flags them for deletion; frisys_hh never deletes position 0 of rank 0 (:356)
static __global__ void k_hh_keep_pos0(uint8_t *del) { del[0] = 0; }

void fr_hh_setup(FriesCtx *c, const fries_hh_params *p) {
    if (p->n_sites < 2 || p->n_sites > 12) throw FriesError("n_sites must be 2..12 (5 bits per site in one 64-bit index)");
    if (p->n_elec < 2 || p->n_elec > 2 * p->n_sites || (p->n_elec & 1)) throw FriesError("n_elec must be even and fit the lattice");
    if (p->max_dets == 0 || p->vec_nonz == 0) throw FriesError("max_dets and vec_nonz must be positive");
    c->hh = *p;
    c->hh_mode = true;
    c->n_orb = p->n_sites; c->n_elec = p->n_elec;
    c->eps = p->eps; c->target_norm = p->target_norm; c->init_thresh = p->initiator;
    c->vec_nonz = p->vec_nonz; c->mat_nonz = p->vec_nonz;
    c->en_shift = 0; c->last_one_norm = 0; c->iterat = 0;
    c->mt.seed(p->seed);
    const unsigned L = p->n_sites;
    c->proc_scr.resize(2 * L); c->vec_scr.resize(2 * L);
    // frisys_hh.cpp:80-83, :88-91; a caller that drew them itself (the reference's driver behind include/FRIES) hands them in
    if (c->in_proc_scr.size() == c->proc_scr.size()) c->proc_scr = c->in_proc_scr; else for (auto &x : c->proc_scr) x = c->mt();
    if (c->in_vec_scr.size() == c->vec_scr.size()) c->vec_scr = c->in_vec_scr; else for (auto &x : c->vec_scr) x = c->mt();
    uint32_t wcap = p->max_dets > p->vec_nonz + 4096 ? p->max_dets : p->vec_nonz + 4096;
    if (!c->comm.small_send) { c->own_small = fr_alloc<uint8_t>(2048); c->comm.small_send = c->own_small; }
    if (!c->d_proc_scr) c->d_proc_scr = fr_alloc<uint32_t>(64);
    if (!c->d_vec_scr) c->d_vec_scr = fr_alloc<uint32_t>(64);
    FR_HIP(hipMemcpyAsync(c->d_proc_scr, c->proc_scr.data(), 4 * c->proc_scr.size(), hipMemcpyHostToDevice, c->stream));
    FR_HIP(hipMemcpyAsync(c->d_vec_scr, c->vec_scr.data(), 4 * c->vec_scr.size(), hipMemcpyHostToDevice, c->stream));
    if (!c->d_hb) { c->d_hb = fr_alloc<HbTables>(1); FR_HIP(hipMemsetAsync(c->d_hb, 0, sizeof(HbTables), c->stream)); }     // unused by uniform stages, but staged
    c->adder_cap = (uint32_t)((uint64_t)p->vec_nonz * 4 / c->n_ranks);      // the Adder gets spawn_length here (:94, :100)
    if (p->full) { uint64_t sl = (uint64_t)p->n_elec * 4 * p->max_dets / c->n_ranks; c->adder_cap = sl > 200000 ? 200000u : (uint32_t)sl; }      // frifull_hh.cpp:91-95
    fr_vec_alloc(c, &c->vec, p->max_dets);
    c->vec.hh_sites = L; c->vec.hh_nelec = p->n_elec; c->vec.hh_buckets = p->max_dets; c->vec.hh_scr = c->d_vec_scr;
    fr_hbpp_alloc(c, wcap);
    // un-normalised stage-1 rows make comb repairs the rule rather than the exception (comp_kernels.hpp: k_sys_prop)
    c->W.prop = 1; c->W.kend = fr_alloc<uint32_t>(wcap); c->W.act[0] = fr_alloc<uint32_t>(wcap + 1); c->W.act[1] = fr_alloc<uint32_t>(wcap + 1); c->W.act_n = fr_alloc<uint32_t>(2);
    c->W.kstart = fr_alloc<uint32_t>(wcap + 1); FR_HIP(hipMemset(c->W.kstart, 0, 4 * ((size_t)wcap + 1)));
    // frifull_hh: at most 4 n_elec adds per stored state (2 hops and 2 phonon moves per electron), vec_nonz states after a compression
    const uint64_t sp_need = p->full ? (uint64_t)4 * p->n_elec * ((uint64_t)p->vec_nonz + 64) + 4096 : (uint64_t)p->vec_nonz + 4096;
    if (sp_need > 0x7fffffffull) throw FriesError("vec_nonz too large for the spawn list");
    fr_spawn_alloc(c, (uint32_t)sp_need);
    fr_xch_alloc(c, (uint32_t)sp_need);
    if (p->full) c->hhf_cnt = fr_alloc<unsigned long long>(2);
    fr_vcomp_alloc(c, p->max_dets);
    c->hh_fdet = fr_alloc<det_t>(wcap);
    c->hh_ovlp = fr_alloc<double>(4 + FR_HH_OVLP_BLOCKS);
    c->W.row1[0] = 1.0; c->W.row1[1] = p->g;        // {hub_t, elec_ph}, :191-192
    // Neel state: alpha electrons on the even sites, beta on the odd ones (hub_holstein.cpp:139-171)
    det_t neel = 0;
    for (unsigned k = 0; k < p->n_elec / 2; k++) { neel |= 1ull << (2 * k); neel |= 1ull << (L + 2 * k + 1); }
    c->hf_det = neel;
    c->hf_proc = fr_host_idx_to_proc(c, neel);
    if (c->rank == c->hf_proc) {
        double v = 100; uint8_t one = 1; uint32_t n1 = 1;
        FR_HIP(hipMemcpyAsync(c->sp.det, &c->hf_det, 8, hipMemcpyHostToDevice, c->stream));
        FR_HIP(hipMemcpyAsync(c->sp.val, &v, 8, hipMemcpyHostToDevice, c->stream));
        FR_HIP(hipMemcpyAsync(c->sp.ini, &one, 1, hipMemcpyHostToDevice, c->stream));
        FR_HIP(hipMemcpyAsync(c->sp.n_spawn, &n1, 4, hipMemcpyHostToDevice, c->stream));
        fr_vec_merge(c, &c->vec, 1, true);
    }
    fr_vec_sync_state(c, &c->vec, &c->h_vst);
}

static inline double hh_uni(std::mt19937 &mt) { return mt() / (1. + UINT32_MAX); }

void fr_hh_iterate(FriesCtx *c, fries_iter_log *lg) {
    hipStream_t st = c->stream;
    const fries_hh_params &P = c->hh;
    fr_vec_sync_state(c, &c->vec, &c->h_vst);
    fr_vec_maybe_rebuild(c, &c->vec);
    uint32_t vec_size = c->h_vst.curr_size;
    uint32_t n_spawn = 0;
    if (P.full) {
        // frifull_hh.cpp:187-263: the whole off-diagonal action, one ordered add list (the reference's Adder flushes only cut it)
        unsigned long long tot[2] = {0, 0};
        if (vec_size) {
            const unsigned gt = fr_blocks(vec_size, FR_TILE);
            FR_HIP(hipMemsetAsync(c->hhf_cnt, 0, 16, st));
            FR_LAUNCH(c, "k_hhf_count", k_hhf_count, dim3(gt), dim3(FR_BLOCK), c->vec, c->eps, P.g, c->sp.pcnt, c->hhf_cnt);
            FR_HIP(hipMemcpyAsync(tot, c->hhf_cnt, 16, hipMemcpyDeviceToHost, st));
            FR_HIP(hipStreamSynchronize(st));
            if (tot[1] > c->sp.cap) throw FriesError("frifull_hh: more adds than the spawn list holds (more than vec_nonz + 64 stored states?)");
            if (tot[1]) FR_LAUNCH(c, "k_hhf_write", k_hhf_write, dim3(gt), dim3(FR_BLOCK), c->vec, c->sp, c->eps, P.g, c->init_thresh, c->sp.pcnt);
        }
        n_spawn = (uint32_t)tot[1];
        c->num_success = (uint32_t)tot[0];
        if (c->use_comm) {
            // Over ranks the reference ships its adds in rounds of about one Adder (frifull_hh.cpp:91-95, 193, 258-262) and a receiver sees
            // (round, source rank, order).  One round -- every rank below its Adder size -- is (source rank, order), what one exchange delivers.
            uint64_t sl = (uint64_t)P.n_elec * 4 * P.max_dets / c->n_ranks;
            if (sl > 200000) sl = 200000;
            if (tot[0] + (uint64_t)P.n_elec * 4 >= sl) throw FriesError("frifull_hh over ranks: this shard fills its Adder (200000 adds) in one iteration; several rounds per iteration are not provided");
        }
        c->comp_len[0] = c->comp_len[1] = 0;
    }
    else {
        double rn[2];
        rn[0] = hh_uni(c->mt); rn[1] = hh_uni(c->mt);       // every rank seeds alike; the reference broadcasts rank 0's draws
        fr_hh_apply(c, c->vec_nonz, rn);
        // spawning (:226-300)
        CompWork &W = c->W;
        uint32_t bound = c->num_success ? c->num_success : 1;
        unsigned grid = fr_blocks(bound, FR_TILE);
        double *f_val = W.S;
        FR_LAUNCH(c, "k_hh_eval", k_hh_eval, dim3(grid), dim3(FR_BLOCK), W, c->vec, 1, c->eps, f_val, c->hh_fdet, W.pcnt[0]);
        FR_LAUNCH(c, "k_hh_compact", k_hh_compact, dim3(grid), dim3(FR_BLOCK), W, c->vec, c->sp, 1, f_val, c->hh_fdet, W.pcnt[0], c->init_thresh);
        FR_HIP(hipMemcpyAsync(&n_spawn, c->sp.n_spawn, 4, hipMemcpyDeviceToHost, st));
        FR_HIP(hipStreamSynchronize(st));
        if (n_spawn > c->sp.cap) throw FriesError("spawn buffer too small");
    }
    if (vec_size) FR_HIP(hipMemsetAsync(c->vec.v1, 0, 8 * (size_t)vec_size, st));      // set_curr_vec_idx(1); zero_vec() (:227-228)
    uint32_t n_merge = n_spawn;
    if (c->use_comm) n_merge = fr_spawn_exchange(c, n_spawn, P.full ? 2 : 0);
    if (n_merge) fr_vec_merge(c, &c->vec, n_merge, false, P.full != 0);
    fr_vec_sync_state(c, &c->vec, &c->h_vst);
    if (c->h_vst.err) throw FriesError("device error in the Hubbard-Holstein merge (capacity, hash table or electron count)");
    // diagonal (:309-321)
    uint32_t nb = c->h_vst.curr_size ? c->h_vst.curr_size : 1;
    FR_LAUNCH(c, "k_hh_death_clone", k_hh_death_clone, dim3(fr_blocks(nb, FR_TILE)), dim3(FR_BLOCK), c->vec, c->vc, vec_size, c->eps, c->en_shift, P.U, P.omega, P.gs_energy, c->vec_nonz);
    // compression (:323-361)
    uint32_t n_samp = c->vec_nonz;
    double glob_norm = 0;
    fr_find_preserve(c, &n_samp, &glob_norm);
    c->glob_norm = glob_norm;
    c->nkept = c->vec_nonz - n_samp;
    if ((c->iterat + 1) % 10 == 0) {
        double damp = 0.05 / 10 / c->eps;
        if (c->last_one_norm) { c->en_shift -= damp * log(glob_norm / c->last_one_norm); c->last_one_norm = glob_norm; }
        if (c->last_one_norm == 0 && glob_norm > c->target_norm) c->last_one_norm = glob_norm;
    }
    // energy estimate (:336-349): every shard's overlap with the Neel state, gathered to the rank that owns it
    FR_LAUNCH(c, "k_hh_ref_ovlp", k_hh_ref_ovlp, dim3(FR_HH_OVLP_BLOCKS), dim3(FR_BLOCK), c->vec, c->hf_det, P.g / 1.0, c->hh_ovlp + 4);
    FR_LAUNCH(c, "k_hh_ref_ovlp_sum", k_hh_ref_ovlp_sum, dim3(1), dim3(64), c->vec, c->hh_ovlp + 4, c->hh_ovlp);
    {
        const int R = c->n_ranks;
        double h[3 * FR_MAX_RANKS];
        const double *src = c->hh_ovlp;
        if (c->use_comm) {
            FR_HIP(hipMemcpyAsync(c->comm.small_send, c->hh_ovlp, 24, hipMemcpyDeviceToDevice, st));
            src = (const double *)fr_allgather(c, 24);
        }
        FR_HIP(hipMemcpyAsync(h, src, 24 * (size_t)R, hipMemcpyDeviceToHost, st));
        FR_HIP(hipStreamSynchronize(st));
        c->numer = 0; c->denom = 0;
        if (c->rank == c->hf_proc) {
            double ref_el = h[3 * c->rank + 1], diag_el = h[3 * c->rank + 2];
            double nu = (diag_el * P.U - P.gs_energy) * ref_el;
            for (int p = 0; p < R; p++) nu += h[3 * p] * -1.0;
            c->numer = nu; c->denom = ref_el;
        }
    }
    double rn_sys = hh_uni(c->mt);
    c->hh_keep0 = c->rank == 0;
    fr_sys_comp(c, n_samp, rn_sys);
    c->iterat++;
    c->tot_iters++; c->tot_spawns += c->num_success;
    if (lg) {
        fr_vec_sync_state(c, &c->vec, &c->h_vst);
        lg->numer = c->numer; lg->denom = c->denom; lg->shift = c->en_shift; lg->norm = c->glob_norm;
        lg->nkept = c->nkept; lg->n_nonz = c->h_vst.n_nonz; lg->curr_size = c->h_vst.curr_size;
        lg->num_success = c->num_success;
        for (int k = 0; k < 5; k++) lg->comp_len[k] = k < 2 ? c->comp_len[k] : 0;
        uint32_t e = 0;
        FR_HIP(hipMemcpy(&e, c->d_err, 4, hipMemcpyDeviceToHost));
        lg->err = e | c->h_vst.err;
    }
}

// calc_ref_ovlp of this rank's shard (hub_holstein.hpp:93-186): out = {overlap sum, value at position 0, diagonal element at position 0}
void fr_hh_ref_ovlp(FriesCtx *c, double out[3]) {
    const fries_hh_params &P = c->hh;
    FR_LAUNCH(c, "k_hh_ref_ovlp", k_hh_ref_ovlp, dim3(FR_HH_OVLP_BLOCKS), dim3(FR_BLOCK), c->vec, c->hf_det, P.g / 1.0, c->hh_ovlp + 4);
    FR_LAUNCH(c, "k_hh_ref_ovlp_sum", k_hh_ref_ovlp_sum, dim3(1), dim3(64), c->vec, c->hh_ovlp + 4, c->hh_ovlp);
    FR_HIP(hipMemcpyAsync(out, c->hh_ovlp, 24, hipMemcpyDeviceToHost, c->stream));
    FR_HIP(hipStreamSynchronize(c->stream));
}
void fr_hh_clear_pos0(FriesCtx *c) { FR_LAUNCH(c, "k_hh_keep_pos0", k_hh_keep_pos0, dim3(1), dim3(1), c->vc.del); }
