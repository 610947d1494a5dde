// Molecular system on the device: integrals, HB-PP tensors (set_up,
// FRIES/Hamiltonians/heat_bathPP.cpp:99-179), Hartree-Fock reference quantities and the
// full single/double enumeration used for H * trial (FRIES/Hamiltonians/molecule.cpp:108-203,
// 448-665).  Every sum below runs in the reference's loop order inside one lane, so the
// tables are bit-identical to the CPU ones.
#include "ctx.hpp"
#include "hbpp_rows.hpp"
#include <cstring>

__global__ void k_hb_pairs(HbTables *T, const double *eris, unsigned n) {
    unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n * n) {
        unsigned i = t / n, j = t % n;
        double s = 0;
        for (unsigned a = 0; a < n; a++) for (unsigned b = 0; b < n; b++)
            if (i != a && j != b) s += fabs(fr_phys(eris, i, j, a, b));
        T->d_diff[i * n + j] = s;
        if (i < j) {
            double d = 0;
            for (unsigned a = 0; a < n; a++) for (unsigned b = 0; b < a; b++)
                if (a != j && a != i && b != j && b != i) d += 2 * fabs(fr_phys(eris, i, j, a, b) - fr_phys(eris, i, j, b, a));
            T->d_same[fr_tri_nodiag(i, j)] = d;
            T->exch_sqrt[fr_tri_nodiag(i, j)] = sqrt(fabs(fr_phys(eris, i, j, j, i)));
        }
        if (i == j) T->diag_sqrt[j] = sqrt(fabs(fr_phys(eris, j, j, j, j)));
    }
}
__global__ void k_hb_rows(HbTables *T, unsigned n) {
    unsigned i = threadIdx.x;
    if (i < n) {
        double s = 0;
        for (unsigned j = 0; j < i; j++) s += T->d_same[fr_tri_nodiag(j, i)];
        for (unsigned j = i + 1; j < n; j++) s += T->d_same[fr_tri_nodiag(i, j)];
        for (unsigned j = 0; j < n; j++) s += T->d_diff[i * n + j];
        T->s_tens[i] = s;
        double e = 0;
        for (unsigned j = 0; j < i; j++) e += T->exch_sqrt[fr_tri_nodiag(j, i)];
        e += T->diag_sqrt[i];
        for (unsigned j = i + 1; j < n; j++) e += T->exch_sqrt[fr_tri_nodiag(i, j)];
        T->exch_norms[i] = e;
    }
    __syncthreads();
    if (i == 0) { double s = 0; for (unsigned k = 0; k < n; k++) s += T->s_tens[k]; T->s_norm = s; }
}

__global__ void k_diag_list(const det_t *dets, uint32_t n, const double *h, const double *eris, unsigned n_orb, double sub, double *out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fr_diag_matrel(dets[i], h, eris, n_orb) - sub;
}

void fr_system_upload(FriesCtx *c, uint32_t n_orb, uint32_t n_elec, const uint8_t *irreps, const double *h_core, const double *eris) {
    if (n_orb > FR_MAX_ORB || n_orb < 2) throw FriesError("n_orb must be in [2, 32]");
    if (n_elec % 2 || n_elec < 2 || n_elec / 2 >= n_orb) throw FriesError("n_elec must be even and leave at least one virtual orbital per spin");
    c->n_orb = n_orb; c->n_elec = n_elec;
    size_t np = (size_t)n_orb * (n_orb + 1) / 2, ne = np * (np + 1) / 2;
    c->d_h = fr_alloc<double>((size_t)n_orb * n_orb);
    c->d_eris = fr_alloc<double>(ne);
    FR_HIP(hipMemcpyAsync(c->d_h, h_core, 8 * (size_t)n_orb * n_orb, hipMemcpyHostToDevice, c->stream));
    FR_HIP(hipMemcpyAsync(c->d_eris, eris, 8 * ne, hipMemcpyHostToDevice, c->stream));
    HbTables &H = c->h_hb;
    memset(&H, 0, sizeof(H));
    H.n_orb = n_orb; H.n_elec = n_elec;
    for (unsigned i = 0; i < n_orb; i++) {
        if (irreps[i] >= 8) throw FriesError("irrep label out of range");
        H.irrep[i] = irreps[i];
        unsigned s = irreps[i], cnt = H.lookup[s][0];      // molecule.cpp:1050-1065
        H.lookup[s][1 + cnt] = (uint8_t)i;
        H.lookup[s][0] = (uint8_t)(cnt + 1);
    }
    H.max_n_symm = 0;
    for (unsigned s = 0; s < 8; s++) if (H.lookup[s][0] > H.max_n_symm) H.max_n_symm = H.lookup[s][0];
    c->d_hb = fr_alloc<HbTables>(1);
    FR_HIP(hipMemcpyAsync(c->d_hb, &H, sizeof(H), hipMemcpyHostToDevice, c->stream));
    FR_LAUNCH(c, "k_hb_pairs", k_hb_pairs, dim3(fr_blocks(n_orb * n_orb, 64)), dim3(64), c->d_hb, c->d_eris, n_orb);
    FR_LAUNCH(c, "k_hb_rows", k_hb_rows, dim3(1), dim3(64), c->d_hb, n_orb);
    FR_HIP(hipMemcpyAsync(&c->h_hb, c->d_hb, sizeof(H), hipMemcpyDeviceToHost, c->stream));
    // Hartree-Fock determinant and its diagonal element (frisys_mol.cpp:90-101)
    det_t half = (1ull << (n_elec / 2)) - 1ull;
    c->hf_det = half | (half << n_orb);
    det_t *dd = fr_alloc<det_t>(1); double *de = fr_alloc<double>(1);
    FR_HIP(hipMemcpyAsync(dd, &c->hf_det, 8, hipMemcpyHostToDevice, c->stream));
    FR_LAUNCH(c, "k_diag_list", k_diag_list, dim3(1), dim3(64), dd, 1u, c->d_h, c->d_eris, n_orb, 0.0, de);
    FR_HIP(hipMemcpyAsync(&c->hf_en, de, 8, hipMemcpyDeviceToHost, c->stream));
    FR_HIP(hipStreamSynchronize(c->stream));
    FR_HIP(hipFree(dd)); FR_HIP(hipFree(de));
}

// ------------------------------------------------------------------ full excitation enumeration
// One workgroup per source determinant; candidates are visited in the reference's loop order in
// chunks of 256 and compacted with a block scan, so the output order is the reference's.
// mode 0: singles (sing_ex_symm), 1: doubles (doub_ex_symm).  pass 0 counts, pass 1 writes.
struct EnumOut { det_t *det; double *val; uint32_t *orbs; };

__device__ __forceinline__ bool fr_enum_candidate(const HbTables &T, det_t det, int mode, uint32_t idx, unsigned *o1, unsigned *o2, unsigned *u1, unsigned *u2) {
    const unsigned n = T.n_orb, ne = T.n_elec, h = ne / 2;
    if (mode == 0) {
        // (electron i, orbital a of the same spin)
        unsigned i = idx / n, a = idx % n;
        if (i >= ne) return false;
        unsigned sp = i / h, io = fr_nth_bit(det, i), ao = a + sp * n;
        if (fr_bit(det, ao) || T.irrep[io % n] != T.irrep[a]) return false;
        *o1 = io; *u1 = ao; *o2 = 0; *u2 = 0;
        return true;
    }
    const uint32_t n_os = h * h * n * n, npair = h * (h - 1) / 2, n_ss = npair * n * n;
    if (idx < n_os) {
        unsigned l = idx % n, k = (idx / n) % n, j = (idx / (n * n)) % h, i = idx / (n * n * h);
        unsigned io = fr_nth_bit(det, i), jo = fr_nth_bit(det, h + j), ko = k, lo = l + n;
        if (fr_bit(det, ko) || fr_bit(det, lo)) return false;
        if ((T.irrep[io] ^ T.irrep[jo - n] ^ T.irrep[k] ^ T.irrep[l]) != 0) return false;
        *o1 = io; *o2 = jo; *u1 = ko; *u2 = lo;
        return true;
    }
    idx -= n_os;
    unsigned sp = 0;
    if (idx >= n_ss) { idx -= n_ss; sp = 1; if (idx >= n_ss) return false; }
    unsigned l = idx % n, k = (idx / n) % n, pr = idx / (n * n);
    if (l <= k) return false;
    // pair index -> (i < j) in the order i outer, j inner
    unsigned i = 0, rem = pr;
    while (rem >= h - 1 - i) { rem -= h - 1 - i; i++; }
    unsigned j = i + 1 + rem;
    unsigned io = fr_nth_bit(det, sp * h + i), jo = fr_nth_bit(det, sp * h + j), ko = k + sp * n, lo = l + sp * n;
    if (fr_bit(det, ko) || fr_bit(det, lo)) return false;
    if ((T.irrep[io % n] ^ T.irrep[jo % n] ^ T.irrep[k] ^ T.irrep[l]) != 0) return false;
    *o1 = io; *o2 = jo; *u1 = ko; *u2 = lo;
    return true;
}

// counts[2*d + mode] = symmetry-allowed excitations; nz[2*d+mode] = those with a non-zero element
__global__ void __launch_bounds__(FR_BLOCK) k_enum(const det_t *src, const double *src_val, uint32_t n_src, SysDev S, int mode, int pass,
                                                   uint32_t *counts, uint32_t *nz, const uint32_t *offsets, EnumOut out, double h_fac, const uint32_t *src_idx = nullptr) {
    __shared__ HbTables T;
    __shared__ uint32_t shu[4];
    fr_stage_tables(&T, S.hb);
    const unsigned n = T.n_orb, ne = T.n_elec, h = ne / 2;
    const uint32_t d = blockIdx.x;              // counts / nz / offsets are indexed by list entry
    if (d >= n_src) return;
    const uint32_t at = src_idx ? src_idx[d] : d;
    const det_t det = src[at];
    const double cur = src_val[at];
    uint32_t n_cand = mode == 0 ? ne * n : h * h * n * n + 2 * (h * (h - 1) / 2) * n * n;
    uint32_t n_allowed = 0, n_written = 0;
    uint32_t obase = pass ? offsets[2 * d + mode] : 0;
    for (uint32_t c0 = 0; c0 < n_cand; c0 += FR_BLOCK) {
        uint32_t idx = c0 + threadIdx.x;
        unsigned o1, o2, u1, u2;
        bool ok = idx < n_cand && cur != 0 && fr_enum_candidate(T, det, mode, idx, &o1, &o2, &u1, &u2);
        double m = 0;
        det_t nd = det;
        if (ok) {
            if (mode == 0) {
                m = fr_sing_matrel(det, o1, u1, S.h_core, S.eris, n);
                nd = det & ~(1ull << o1);
                int sgn = (fr_bits_between(nd, o1, u1) & 1) ? -1 : 1;       // sing_det_parity, fci_utils.c:46-51
                nd |= 1ull << u1;
                m *= sgn;
            }
            else {
                m = fr_doub_matrel(o1, o2, u1, u2, S.eris, n);
                m *= fr_doub_parity(det, o1, o2, u1, u2);                   // doub_det_parity, fci_utils.c:66-75
                nd = (det & ~(1ull << o1) & ~(1ull << o2)) | (1ull << u1) | (1ull << u2);
            }
            // time-reversal symmetrised vectors: the element between the symmetrised functions, added to the pair's representative
            if (S.spin_parity) { det_t tgt = nd; if (fr_adjust_tr(T, S, det, nd, &m, S.spin_parity, &tgt, 0, nullptr, 0.0)) nd = tgt; else m = 0; }
            m *= cur * h_fac;
        }
        uint32_t f = (ok && m != 0) ? 1u : 0u, tot, tot_ok;
        uint32_t incl_ok = fr_block_scan_u32(ok ? 1u : 0u, shu, &tot_ok);
        (void)incl_ok;
        uint32_t incl = fr_block_scan_u32(f, shu, &tot);
        if (pass && f) {
            uint32_t o = obase + n_written + incl - 1;
            out.det[o] = nd; out.val[o] = m; if (out.orbs) out.orbs[o] = fr_code(o1, o2, u1, u2);
        }
        n_allowed += tot_ok; n_written += tot;
    }
    if (!pass && threadIdx.x == 0) { counts[2 * d + mode] = n_allowed; nz[2 * d + mode] = n_written; }
}

// H restricted to the dense space, times -eps (frisys_mol.cpp:347-397): for every dense determinant, in position order, its
// symmetry-allowed singles and then its doubles as (from position, to determinant, <to|H|from> * parity * -eps).  Excitations whose
// element is zero are counted (they are part of tot_dense_h, which the matrix sample budget is reduced by, :421) but not stored:
// the reference's add() drops a zero value.
void fr_dense_h_setup(FriesCtx *c) {
    hipStream_t st = c->stream;
    const uint32_t ns = c->vec.n_dense;
    c->n_dense_h = c->n_dense_h_nz = 0;
    if (!c->d_dense_norm) c->d_dense_norm = fr_alloc<double>(1);
    if (!ns) return;
    SysDev S; S.n_orb = c->n_orb; S.n_elec = c->n_elec; S.h_core = c->d_h; S.eris = c->d_eris; S.hb = c->d_hb; S.hf_en = c->hf_en; S.spin_parity = c->spin_parity;
    std::vector<double> ones(ns, 1.0);
    double *d_val = fr_alloc<double>(ns);
    uint32_t *d_cnt = fr_alloc<uint32_t>(2 * ns), *d_nz = fr_alloc<uint32_t>(2 * ns), *d_off = fr_alloc<uint32_t>(2 * ns);
    FR_HIP(hipMemcpyAsync(d_val, ones.data(), 8 * (size_t)ns, hipMemcpyHostToDevice, st));
    EnumOut eo{nullptr, nullptr, nullptr};
    for (int mode = 0; mode < 2; mode++)
        FR_LAUNCH(c, "k_enum", k_enum, dim3(ns), dim3(FR_BLOCK), c->vec.dets, d_val, ns, S, mode, 0, d_cnt, d_nz, d_off, eo, -c->eps);
    std::vector<uint32_t> cnt(2 * ns), nz(2 * ns), off(2 * ns);
    FR_HIP(hipMemcpyAsync(cnt.data(), d_cnt, 8 * (size_t)ns, hipMemcpyDeviceToHost, st));
    FR_HIP(hipMemcpyAsync(nz.data(), d_nz, 8 * (size_t)ns, hipMemcpyDeviceToHost, st));
    FR_HIP(hipStreamSynchronize(st));
    uint64_t tot = 0; uint32_t o = 0;
    std::vector<uint32_t> from;
    for (uint32_t d = 0; d < ns; d++)
        for (int mode = 0; mode < 2; mode++) { tot += cnt[2 * d + mode]; off[2 * d + mode] = o; o += nz[2 * d + mode]; from.insert(from.end(), nz[2 * d + mode], d); }
    if (tot > 0xffffffffull) throw FriesError("dense block of H too large");
    c->n_dense_h = (uint32_t)tot; c->n_dense_h_nz = o;
    c->d_dh_from = fr_alloc<uint32_t>(o ? o : 1); c->d_dh_to = fr_alloc<det_t>(o ? o : 1); c->d_dh_el = fr_alloc<double>(o ? o : 1);
    if (o) {
        FR_HIP(hipMemcpyAsync(d_off, off.data(), 8 * (size_t)ns, hipMemcpyHostToDevice, st));
        FR_HIP(hipMemcpyAsync(c->d_dh_from, from.data(), 4 * (size_t)o, hipMemcpyHostToDevice, st));
        EnumOut wo{c->d_dh_to, c->d_dh_el, nullptr};
        for (int mode = 0; mode < 2; mode++)
            FR_LAUNCH(c, "k_enum", k_enum, dim3(ns), dim3(FR_BLOCK), c->vec.dets, d_val, ns, S, mode, 1, d_cnt, d_nz, d_off, wo, -c->eps);
    }
    FR_HIP(hipStreamSynchronize(st));
    FR_HIP(hipFree(d_val)); FR_HIP(hipFree(d_cnt)); FR_HIP(hipFree(d_nz)); FR_HIP(hipFree(d_off));
}

// H * trial with trial = the list (src, src_val): returns on the host the merged (det, value)
// list in the reference's storage order, i.e. what htrial_vec holds after
// h_op_offdiag / h_op_diag / add_vecs (frisys_mol.cpp:205-210).
void fr_h_apply_list(FriesCtx *c, const std::vector<det_t> &src, const std::vector<double> &val,
                     std::vector<det_t> &out_det, std::vector<double> &out_val, uint32_t *n_sing0, uint32_t *n_doub0, bool with_diag) {
    hipStream_t st = c->stream;
    uint32_t ns = (uint32_t)src.size();
    SysDev S; S.n_orb = c->n_orb; S.n_elec = c->n_elec; S.h_core = c->d_h; S.eris = c->d_eris; S.hb = c->d_hb; S.hf_en = c->hf_en; S.spin_parity = c->spin_parity;
    det_t *d_src = fr_alloc<det_t>(ns); double *d_val = fr_alloc<double>(ns);
    uint32_t *d_cnt = fr_alloc<uint32_t>(2 * ns), *d_nz = fr_alloc<uint32_t>(2 * ns), *d_off = fr_alloc<uint32_t>(2 * ns);
    FR_HIP(hipMemcpyAsync(d_src, src.data(), 8 * (size_t)ns, hipMemcpyHostToDevice, st));
    FR_HIP(hipMemcpyAsync(d_val, val.data(), 8 * (size_t)ns, hipMemcpyHostToDevice, st));
    EnumOut eo{nullptr, nullptr, nullptr};
    for (int mode = 0; mode < 2; mode++)
        FR_LAUNCH(c, "k_enum", k_enum, dim3(ns), dim3(FR_BLOCK), d_src, d_val, ns, S, mode, 0, d_cnt, d_nz, d_off, eo, 1.0);
    std::vector<uint32_t> cnt(2 * ns), nz(2 * ns), off(2 * ns);
    FR_HIP(hipMemcpyAsync(cnt.data(), d_cnt, 8 * (size_t)ns, hipMemcpyDeviceToHost, st));
    FR_HIP(hipMemcpyAsync(nz.data(), d_nz, 8 * (size_t)ns, hipMemcpyDeviceToHost, st));
    FR_HIP(hipStreamSynchronize(st));
    // output order: source dets first (they are added before h_op_offdiag), then all singles, then all doubles
    uint32_t o = ns;
    for (uint32_t d = 0; d < ns; d++) { off[2 * d] = o; o += nz[2 * d]; }
    for (uint32_t d = 0; d < ns; d++) { off[2 * d + 1] = o; o += nz[2 * d + 1]; }
    uint32_t total = o;
    if (n_sing0) *n_sing0 = cnt[0];
    if (n_doub0) *n_doub0 = cnt[1];
    FR_HIP(hipMemcpyAsync(d_off, off.data(), 8 * (size_t)ns, hipMemcpyHostToDevice, st));
    // temporary vector with the merge machinery's own spawn buffers
    VecDev hv{};
    fr_vec_alloc(c, &hv, total + 16);
    SpawnBuf saved = c->sp;
    SpawnBuf tmp{};
    c->sp = tmp;
    fr_spawn_alloc(c, total + 16);
    uint32_t *d_orbs = fr_alloc<uint32_t>(total + 16);
    eo.det = c->sp.det; eo.val = c->sp.val; eo.orbs = d_orbs;
    FR_HIP(hipMemcpyAsync(c->sp.det, src.data(), 8 * (size_t)ns, hipMemcpyHostToDevice, st));
    FR_HIP(hipMemsetAsync(c->sp.val, 0, 8 * (size_t)ns, st));      // the sources enter with value 0 in column 1 ...
    FR_HIP(hipMemsetAsync(c->sp.ini, 1, total + 16, st));
    for (int mode = 0; mode < 2; mode++)
        FR_LAUNCH(c, "k_enum", k_enum, dim3(ns), dim3(FR_BLOCK), d_src, d_val, ns, S, mode, 1, d_cnt, d_nz, d_off, eo, 1.0);
    FR_HIP(hipMemcpyAsync(c->sp.n_spawn, &total, 4, hipMemcpyHostToDevice, st));
    fr_vec_merge(c, &hv, total, false);
    // ... and their own amplitude times the diagonal element in column 0 (h_op_diag with id_fac 0, h_fac 1)
    std::vector<det_t> hd(total); std::vector<double> v1(total);
    VecState hs;
    fr_vec_sync_state(c, &hv, &hs);
    if (hs.err) throw FriesError("H * trial merge failed");
    uint32_t nout = hs.curr_size;
    FR_HIP(hipMemcpy(hd.data(), hv.dets, 8 * (size_t)nout, hipMemcpyDeviceToHost));
    FR_HIP(hipMemcpy(v1.data(), hv.v1, 8 * (size_t)nout, hipMemcpyDeviceToHost));
    double *d_diag = fr_alloc<double>(ns);
    FR_LAUNCH(c, "k_diag_list", k_diag_list, dim3(fr_blocks(ns, 64)), dim3(64), d_src, ns, c->d_h, c->d_eris, c->n_orb, c->hf_en, d_diag);
    std::vector<double> dg(ns);
    FR_HIP(hipMemcpyAsync(dg.data(), d_diag, 8 * (size_t)ns, hipMemcpyDeviceToHost, st));
    FR_HIP(hipStreamSynchronize(st));
    out_det.assign(hd.begin(), hd.begin() + nout);
    out_val.assign(nout, 0.0);
    // positions 0..ns-1 are the sources in order (they arrived first); duplicates among sources are not supported
    for (uint32_t i = 0; i < nout; i++) {
        double c0 = 0;
        if (with_diag && i < ns && val[i] != 0) c0 = val[i] * (0 + 1 * dg[i]);
        out_val[i] = c0 + v1[i] * 1.0;
    }
    // release temporaries
    SpawnBuf used = c->sp;
    c->sp = saved;
    hipFree(used.det); hipFree(used.val); hipFree(used.ini); hipFree(used.slot); hipFree(used.flag);
    for (int h2 = 0; h2 < 2; h2++) { hipFree(used.key[h2]); hipFree(used.pay[h2]); }
    hipFree(used.hist); hipFree(used.pcnt); hipFree(used.n_spawn);
    hipFree(hv.dets); hipFree(hv.v0); hipFree(hv.v1); hipFree(hv.diag); hipFree(hv.active); hipFree(hv.free_stack); hipFree(hv.hs); hipFree(hv.stat_part); hipFree(hv.st);
    hipFree(d_src); hipFree(d_val); hipFree(d_cnt); hipFree(d_nz); hipFree(d_off); hipFree(d_orbs); hipFree(d_diag);
}

void fr_h_trial_setup(FriesCtx *c) {
    std::vector<det_t> src{c->hf_det}, od;
    std::vector<double> val{1.0}, ov;
    uint32_t n_sing = 0, n_doub = 0;
    fr_h_apply_list(c, src, val, od, ov, &n_sing, &n_doub, true);
    c->p_doub = (double)n_doub / (n_sing + n_doub);        // frisys_mol.cpp:216-220 (always from the HF determinant)
    c->W.row1[0] = c->p_doub; c->W.row1[1] = 1 - c->p_doub;    // heat_bathPP.cpp:714-727
    if (!c->in_trial_det.empty()) {
        // --trial_vec (frisys_mol.cpp:157-181): the entries are add()ed in file order, so zero values never arrive and a repeated
        // determinant sums into its first position; then H * trial by h_op_offdiag / h_op_diag / add_vecs (:205-210)
        src.clear(); val.clear();
        for (size_t i = 0; i < c->in_trial_det.size(); i++) {
            if (c->in_trial_val[i] == 0) continue;
            size_t j = 0;
            while (j < src.size() && src[j] != c->in_trial_det[i]) j++;
            if (j == src.size()) { src.push_back(c->in_trial_det[i]); val.push_back(c->in_trial_val[i]); }
            else val[j] += c->in_trial_val[i];
        }
        if (src.empty()) throw FriesError("the trial vector holds no non-zero element");
        fr_h_apply_list(c, src, val, od, ov, nullptr, nullptr, true);
        if (c->fq_mode) {
            // fciqmc_mol.cpp:163-170 stores every entry with `while (!trial_vec.add(...)) trial_vec.perform_add(0)`; add() reports a full
            // Adder AFTER storing, so the entry that fills it is stored again.  trial_vec's Adder holds exactly as many entries as the
            // file has, so its last entry counts twice in the trial vector (but once in H * trial): the reference's denominators.
            size_t stored = 0;
            for (size_t i = 0; i < c->in_trial_det.size(); i++) {
                if (c->in_trial_val[i] == 0) continue;
                stored++;
                if (stored == c->in_trial_det.size()) {
                    size_t j = 0;
                    while (src[j] != c->in_trial_det[i]) j++;
                    val[j] += c->in_trial_val[i];
                    stored = 1;
                }
            }
        }
    }
    c->n_trial = (uint32_t)src.size(); c->n_htrial = (uint32_t)od.size();
    c->tr_det = fr_alloc<det_t>(src.size()); c->tr_val = fr_alloc<double>(src.size());
    c->htr_det = fr_alloc<det_t>(od.size()); c->htr_val = fr_alloc<double>(od.size());
    FR_HIP(hipMemcpy(c->tr_det, src.data(), 8 * src.size(), hipMemcpyHostToDevice));
    FR_HIP(hipMemcpy(c->tr_val, val.data(), 8 * val.size(), hipMemcpyHostToDevice));
    FR_HIP(hipMemcpy(c->htr_det, od.data(), 8 * od.size(), hipMemcpyHostToDevice));
    FR_HIP(hipMemcpy(c->htr_val, ov.data(), 8 * ov.size(), hipMemcpyHostToDevice));
}

// ------------------------------------------------------------------ deterministic H on the stored vector (frifull_mol)
// h_op_diag (molecule.cpp:205-219): column 1 <- column 0 * (id_fac + h_fac * H_ii), zero where column 0 is zero
__global__ void __launch_bounds__(FR_BLOCK) k_hop_diag(VecDev V, SysDev S, double id_fac, double h_fac) {
    const uint32_t n = V.st->curr_size;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = V.v0[i], o = 0;
    if (v != 0) {
        double d = V.diag[i];
        if (d != d) { d = fr_diag_matrel(V.dets[i], S.h_core, S.eris, S.n_orb) - S.hf_en; V.diag[i] = d; }
        o = v * (id_fac + h_fac * d);
    }
    V.v1[i] = o;
}
void fr_h_diag_vec(FriesCtx *c, double id_fac, double h_fac) {
    SysDev S; S.n_orb = c->n_orb; S.n_elec = c->n_elec; S.h_core = c->d_h; S.eris = c->d_eris; S.hb = c->d_hb; S.hf_en = c->hf_en; S.spin_parity = c->spin_parity;
    uint32_t bound = c->h_vst.curr_size ? c->h_vst.curr_size : 1;
    FR_LAUNCH(c, "k_hop_diag", k_hop_diag, dim3(fr_blocks(bound, FR_BLOCK)), dim3(FR_BLOCK), c->vec, S, id_fac, h_fac);
}

// positions with a non-zero value in column 0, ascending (the determinants h_op_offdiag visits, molecule.cpp:569-574)
__global__ void __launch_bounds__(FR_BLOCK) k_src_count(VecDev V, uint32_t *pcnt) {
    __shared__ uint32_t shu[4];
    const uint32_t n = V.st->curr_size;
    size_t base = (size_t)blockIdx.x * FR_TILE + (size_t)threadIdx.x * FR_ITEMS;
    uint32_t cnt = 0;
    for (int it = 0; it < FR_ITEMS; it++) { size_t i = base + it; if (i < n && V.v0[i] != 0) cnt++; }
    uint32_t bc = fr_block_sum_u32(cnt, shu);
    if (threadIdx.x == 0) pcnt[blockIdx.x] = bc;
}
__global__ void __launch_bounds__(FR_BLOCK) k_src_write(VecDev V, const uint32_t *pcnt, uint32_t *list, uint32_t *n_out) {
    __shared__ uint32_t shu[4];
    const uint32_t n = V.st->curr_size;
    uint32_t off;
    { uint32_t x = 0; for (unsigned i = threadIdx.x; i < blockIdx.x; i += blockDim.x) x += pcnt[i]; off = fr_block_sum_u32(x, shu); }
    size_t base = (size_t)blockIdx.x * FR_TILE + (size_t)threadIdx.x * FR_ITEMS;
    uint32_t f[FR_ITEMS], tsum = 0;
#pragma unroll
    for (int it = 0; it < FR_ITEMS; it++) { size_t i = base + it; f[it] = (i < n && V.v0[i] != 0) ? 1u : 0u; tsum += f[it]; }
    uint32_t tot;
    uint32_t incl = fr_block_scan_u32(tsum, shu, &tot);
    uint32_t r = off + incl - tsum;
#pragma unroll
    for (int it = 0; it < FR_ITEMS; it++) if (f[it]) list[r++] = (uint32_t)(base + it);
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *n_out = off + tot;
}

// h_op_offdiag (molecule.cpp:448-665, spin_parity 0) from column 0 into column 1: every symmetry-allowed single excitation of
// every stored determinant in storage order, then every double, each worth value * h_fac * <j|H|i>; the annihilating
// merge takes them in that order.  The spawn buffer holds sp.cap entries, so the list is produced and merged in chunks of
// whole determinants (the reference's Adder does the same every 1e6 adds).  Returns the number of non-zero adds.
// With ranks (frifull_mol under mpiexec, molecule.cpp:553-660): the singles of every stored determinant are one pass of adds, the doubles
// another; each pass is shipped to the owners like frisys_mol's spawns -- in several perform_add rounds when a destination's Adder buffer
// fills (fr_spawn_exchange, vec.hip) -- and every rank takes part in both exchanges, also with nothing to send.
static uint64_t fr_h_offdiag_vec_ranks(FriesCtx *c, double h_fac) {
    hipStream_t st = c->stream;
    fr_vec_sync_state(c, &c->vec, &c->h_vst);
    const uint32_t ns = c->h_vst.curr_size;
    SysDev S; S.n_orb = c->n_orb; S.n_elec = c->n_elec; S.h_core = c->d_h; S.eris = c->d_eris; S.hb = c->d_hb; S.hf_en = c->hf_en; S.spin_parity = c->spin_parity;
    if (c->full_cap < ns || !c->full_cnt) {
        if (c->full_cnt) { FR_HIP(hipFree(c->full_cnt)); FR_HIP(hipFree(c->full_nz)); FR_HIP(hipFree(c->full_off)); FR_HIP(hipFree(c->full_list)); }
        c->full_cap = c->vec.cap;
        c->full_cnt = fr_alloc<uint32_t>(2 * (size_t)c->full_cap); c->full_nz = fr_alloc<uint32_t>(2 * (size_t)c->full_cap); c->full_off = fr_alloc<uint32_t>(2 * (size_t)c->full_cap);
        c->full_list = fr_alloc<uint32_t>(c->full_cap);
    }
    uint32_t nl = 0;
    if (ns) {
        const unsigned gt = fr_blocks(ns, FR_TILE);
        FR_LAUNCH(c, "k_src_count", k_src_count, dim3(gt), dim3(FR_BLOCK), c->vec, c->sp.pcnt);
        FR_LAUNCH(c, "k_src_write", k_src_write, dim3(gt), dim3(FR_BLOCK), c->vec, c->sp.pcnt, c->full_list, c->full_off);
        FR_HIP(hipMemcpyAsync(&nl, c->full_off, 4, hipMemcpyDeviceToHost, st));
        FR_HIP(hipStreamSynchronize(st));
    }
    EnumOut eo{nullptr, nullptr, nullptr};
    std::vector<uint32_t> nz(2 * (size_t)nl + 2), off(2 * (size_t)nl + 2);
    if (nl) {
        for (int mode = 0; mode < 2; mode++)
            FR_LAUNCH(c, "k_enum", k_enum, dim3(nl), dim3(FR_BLOCK), c->vec.dets, c->vec.v0, nl, S, mode, 0, c->full_cnt, c->full_nz, c->full_off, eo, h_fac, c->full_list);
        FR_HIP(hipMemcpyAsync(nz.data(), c->full_nz, 8 * (size_t)nl, hipMemcpyDeviceToHost, st));
        FR_HIP(hipStreamSynchronize(st));
    }
    eo.det = c->sp.det; eo.val = c->sp.val;
    uint64_t n_add = 0;
    for (int mode = 0; mode < 2; mode++) {
        uint64_t tot64 = 0;
        for (uint32_t d = 0; d < nl; d++) { off[2 * (size_t)d + mode] = (uint32_t)tot64; tot64 += nz[2 * (size_t)d + mode]; }
        // (a rank whose pass does not fit raises here and the others wait in the exchange: size spawn_cap for the largest pass)
        if (tot64 > c->sp.cap) throw FriesError("frifull_mol over ranks: a pass of the Hamiltonian (all singles, or all doubles, of this rank's determinants) must fit the spawn buffer: raise spawn_cap");
        const uint32_t tot = (uint32_t)tot64;
        FR_HIP(hipMemsetAsync(c->sp.ini, 1, c->sp.cap, st));        // every add of H * v carries the initiator flag (molecule.cpp:599, :649)
        if (tot) {
            FR_HIP(hipMemcpyAsync(c->full_off, off.data(), 8 * (size_t)nl, hipMemcpyHostToDevice, st));
            FR_LAUNCH(c, "k_enum", k_enum, dim3(nl), dim3(FR_BLOCK), c->vec.dets, c->vec.v0, nl, S, mode, 1, c->full_cnt, c->full_nz, c->full_off, eo, h_fac, c->full_list);
        }
        FR_HIP(hipMemcpyAsync(c->sp.n_spawn, &tot, 4, hipMemcpyHostToDevice, st));
        FR_HIP(hipStreamSynchronize(st));       // tot and off are host temporaries
        bool merged = false;
        const uint32_t n_recv = fr_spawn_exchange(c, tot, 0, &merged);
        if (n_recv && !merged) fr_vec_merge(c, &c->vec, n_recv, false);
        fr_vec_sync_state(c, &c->vec, &c->h_vst);
        if (c->h_vst.err) return n_add;
        fr_vec_maybe_rebuild(c, &c->vec);
        n_add += tot;
    }
    return n_add;
}

uint64_t fr_h_offdiag_vec(FriesCtx *c, double h_fac) {
    if (c->use_comm) return fr_h_offdiag_vec_ranks(c, h_fac);
    hipStream_t st = c->stream;
    fr_vec_sync_state(c, &c->vec, &c->h_vst);
    const uint32_t ns = c->h_vst.curr_size;
    if (ns == 0) return 0;
    SysDev S; S.n_orb = c->n_orb; S.n_elec = c->n_elec; S.h_core = c->d_h; S.eris = c->d_eris; S.hb = c->d_hb; S.hf_en = c->hf_en; S.spin_parity = c->spin_parity;
    if (c->full_cap < ns) {
        if (c->full_cnt) { FR_HIP(hipFree(c->full_cnt)); FR_HIP(hipFree(c->full_nz)); FR_HIP(hipFree(c->full_off)); FR_HIP(hipFree(c->full_list)); }
        c->full_cap = c->vec.cap;
        c->full_cnt = fr_alloc<uint32_t>(2 * (size_t)c->full_cap); c->full_nz = fr_alloc<uint32_t>(2 * (size_t)c->full_cap); c->full_off = fr_alloc<uint32_t>(2 * (size_t)c->full_cap);
        c->full_list = fr_alloc<uint32_t>(c->full_cap);
    }
    // the determinants with a non-zero value, in storage order
    const unsigned gt = fr_blocks(ns, FR_TILE);
    FR_LAUNCH(c, "k_src_count", k_src_count, dim3(gt), dim3(FR_BLOCK), c->vec, c->sp.pcnt);
    FR_LAUNCH(c, "k_src_write", k_src_write, dim3(gt), dim3(FR_BLOCK), c->vec, c->sp.pcnt, c->full_list, c->full_off);
    uint32_t nl = 0;
    FR_HIP(hipMemcpyAsync(&nl, c->full_off, 4, hipMemcpyDeviceToHost, st));
    FR_HIP(hipStreamSynchronize(st));
    if (nl == 0) return 0;
    EnumOut eo{nullptr, nullptr, nullptr};
    for (int mode = 0; mode < 2; mode++)
        FR_LAUNCH(c, "k_enum", k_enum, dim3(nl), dim3(FR_BLOCK), c->vec.dets, c->vec.v0, nl, S, mode, 0, c->full_cnt, c->full_nz, c->full_off, eo, h_fac, c->full_list);
    std::vector<uint32_t> nz(2 * (size_t)nl), off(2 * (size_t)nl);
    FR_HIP(hipMemcpyAsync(nz.data(), c->full_nz, 8 * (size_t)nl, hipMemcpyDeviceToHost, st));
    FR_HIP(hipStreamSynchronize(st));
    FR_HIP(hipMemsetAsync(c->sp.ini, 1, c->sp.cap, st));
    eo.det = c->sp.det; eo.val = c->sp.val;
    uint64_t n_add = 0;
    // (the source column does not move while its excitations are merged into the other one: positions are stable)
    for (int mode = 0; mode < 2; mode++) {
        uint32_t d0 = 0;
        while (d0 < nl) {
            uint32_t d1 = d0, tot = 0;
            while (d1 < nl && (uint64_t)tot + nz[2 * (size_t)d1 + mode] <= c->sp.cap) { off[2 * (size_t)d1 + mode] = tot; tot += nz[2 * (size_t)d1 + mode]; d1++; }
            if (d1 == d0) throw FriesError("spawn buffer smaller than one determinant's excitation list");
            if (tot) {
                FR_HIP(hipMemcpyAsync(c->full_off + 2 * (size_t)d0, off.data() + 2 * (size_t)d0, 8 * (size_t)(d1 - d0), hipMemcpyHostToDevice, st));
                FR_LAUNCH(c, "k_enum", k_enum, dim3(d1 - d0), dim3(FR_BLOCK), c->vec.dets, c->vec.v0, d1 - d0, S, mode, 1, c->full_cnt + 2 * (size_t)d0,
                          c->full_nz + 2 * (size_t)d0, c->full_off + 2 * (size_t)d0, eo, h_fac, c->full_list + d0);
                FR_HIP(hipMemcpyAsync(c->sp.n_spawn, &tot, 4, hipMemcpyHostToDevice, st));
                FR_HIP(hipStreamSynchronize(st));       // tot and off are host temporaries
                fr_vec_merge(c, &c->vec, tot, false);
                fr_vec_sync_state(c, &c->vec, &c->h_vst);
                if (c->h_vst.err) return n_add;
                fr_vec_maybe_rebuild(c, &c->vec);
                n_add += tot;
            }
            d0 = d1;
        }
    }
    return n_add;
}
