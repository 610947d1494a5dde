/*! \file  FRIES/Hamiltonians/heat_bathPP.hpp for the MI355X build: hb_info + set_up, the compression work structures and
 * apply_HBPP_sys with the reference's signatures (FRIES/Hamiltonians/heat_bathPP.hpp:25-34, 52-54, 250-297, 329-333).
 *
 * apply_HBPP_sys is the hot path: it draws the five uniforms the reference draws (one before each comp_sub, heat_bathPP.cpp:729,
 * 765, 811, 859, 910), moves the vector that owns `all_dets` to the device the first time, runs fries_apply_hbpp_sys (five
 * find_keep_sub + sys_sub stages, fries_amd/csrc/hbpp.hip) on the vector's column 0 and hands the surviving matrix elements back in
 * comp_scratch->{vec_len, det_indices2, orb_indices1, vec1}, which is everything the drivers read afterwards.  The work arrays of the
 * reference's CPU algorithm (subwts, keep_sub, ndiv, wt_remain, comp_idx, vec2, ...) exist as members but are not sized: nothing in
 * this build touches them.  The sing_mat_fxn / doub_mat_fxn callbacks are not called -- the device evaluates the same Slater-Condon
 * elements on the integrals parse_fcidump uploaded. */
#ifndef heat_bathPP_h
#define heat_bathPP_h
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <random>
#include <sstream>
#include <string>
#include <stdexcept>
#include <vector>
#include <FRIES/ndarr.hpp>
#include <FRIES/vec_utils.hpp>
#include <FRIES/compress_utils.hpp>
#include <FRIES/fci_utils.h>
#include <FRIES/Hamiltonians/near_uniform.hpp>
#include <FRIES/Hamiltonians/molecule.hpp>
#include <FRIES/backend.hpp>

struct hb_info {
    size_t n_orb;
    double *s_tens;     ///< single-electron components, length n_orb
    double s_norm;
    double *d_same;     ///< same-spin double components, n_orb choose 2
    double *d_diff;     ///< opposite-spin double components, n_orb x n_orb
    double *exch_sqrt;  ///< sqrt |<ia|ai>|, n_orb choose 2
    double *diag_sqrt;  ///< sqrt |<pp|pp>|, n_orb
    double *exch_norms; ///< row sums of the exchange square roots
};

/* the tensors as the engine computed them from the uploaded integrals (heat_bathPP.cpp:99-179 restated in csrc/system.hip) */
inline hb_info *set_up(uint32_t tot_orb, uint32_t n_orb, const SymmERIs &eris) {
    fries_hip::check_mol(&eris, nullptr, 2 * (tot_orb - n_orb));
    fries_ctx *cx = fries_hip::Backend::get().ctx();
    hb_info *hb = (hb_info *)malloc(sizeof(hb_info));
    hb->n_orb = n_orb;
    auto fetch = [&](int which, size_t len) {
        double *p = (double *)malloc(sizeof(double) * (len ? len : 1));
        size_t got = 0;
        fries_hip::ck(fries_get_hb_tensor(cx, which, p, len, &got));
        return p;
    };
    const size_t n = n_orb, tri = n * (n - 1) / 2;
    hb->s_tens = fetch(0, n); hb->d_same = fetch(1, tri); hb->d_diff = fetch(2, n * n); hb->exch_sqrt = fetch(3, tri);
    hb->diag_sqrt = fetch(4, n); hb->exch_norms = fetch(5, n);
    size_t got = 0;
    fries_hip::ck(fries_get_hb_tensor(cx, 6, &hb->s_norm, 1, &got));
    return hb;
}

/* ---- the factorisation's probability rows and weights on the host (heat_bathPP.cpp:182-598), for callers that sample from them one
 * determinant at a time (hb_doub_multi below; the reference's fciqmc_mol / frimulti_mol loops).  Same sums in the same order as the
 * reference; the device forms the same rows per lane from the tables in LDS (csrc/hbpp_rows.hpp).  Each returns the row's weight
 * relative to the factor above it. */
namespace fries_hip {
inline size_t tri_lt(size_t i, size_t j) { return j * (j - 1) / 2 + i; }                    // i < j
inline double hb_pair_sqrt(const hb_info *t, unsigned o, unsigned u) {                      // sqrt |<ou|uo>|, or the diagonal one for o == u
    if (o == u) return t->diag_sqrt[o];
    return t->exch_sqrt[o < u ? tri_lt(o, u) : tri_lt(u, o)];
}
}
/* first occupied orbital: the single-electron weights of the occupied orbitals (exclude_first drops the first one) */
inline double calc_o1_probs(hb_info *tens, double *prob_arr, unsigned int n_elec, uint8_t *occ_orbs, int exclude_first) {
    const unsigned int n = (unsigned int)tens->n_orb, skip = exclude_first > 0 ? 1u : 0u;
    double norm = 0;
    for (unsigned int k = skip; k < n_elec; k++) { const double w = tens->s_tens[occ_orbs[k] % n]; prob_arr[k - skip] = w; norm += w; }
    const double inv = 1. / norm;
    for (unsigned int k = 0; k < n_elec - skip; k++) prob_arr[k] *= inv;
    return norm / tens->s_norm;
}
/* second occupied orbital given the first: opposite-spin electrons first, then the same-spin ones below and above it */
inline double calc_o2_probs(hb_info *tens, double *prob_arr, unsigned int n_elec, uint8_t *occ_orbs, uint8_t o1_idx) {
    const unsigned int n = (unsigned int)tens->n_orb, half = n_elec / 2;
    const unsigned int o1 = occ_orbs[o1_idx] % n, sp = occ_orbs[o1_idx] / n;
    double norm = 0;
    const unsigned int opp = (1 - sp) * half, own = sp * half;
    for (unsigned int k = opp; k < opp + half; k++) { prob_arr[k] = tens->d_diff[o1 * n + occ_orbs[k] % n]; norm += prob_arr[k]; }
    for (unsigned int k = own; k < o1_idx; k++) { prob_arr[k] = tens->d_same[fries_hip::tri_lt(occ_orbs[k] % n, o1)]; norm += prob_arr[k]; }
    for (unsigned int k = o1_idx + 1u; k < own + half; k++) { prob_arr[k] = tens->d_same[fries_hip::tri_lt(o1, occ_orbs[k] % n)]; norm += prob_arr[k]; }
    prob_arr[o1_idx] = 0;
    const double inv = 1. / norm;
    for (unsigned int k = 0; k < n_elec; k++) prob_arr[k] *= inv;
    return norm / tens->s_tens[o1];
}
/* the same restricted to the electrons in front of the first one (unnormalised factorisation) */
inline double calc_o2_probs_half(hb_info *tens, double *prob_arr, unsigned int n_elec, uint8_t *occ_orbs, uint8_t o1_idx) {
    const unsigned int n = (unsigned int)tens->n_orb, half = n_elec / 2;
    const unsigned int o1 = occ_orbs[o1_idx], sp = o1 / n;
    double norm = 0;
    const unsigned int upto = half > o1_idx ? o1_idx : half;
    for (unsigned int k = 0; k < upto; k++) { prob_arr[k] = sp == 0 ? tens->d_same[fries_hip::tri_lt(occ_orbs[k], o1)] : tens->d_diff[(o1 - n) * n + occ_orbs[k]]; norm += prob_arr[k]; }
    for (unsigned int k = half; k < o1_idx; k++) { prob_arr[k] = sp == 0 ? tens->d_diff[o1 * n + occ_orbs[k] - n] : tens->d_same[fries_hip::tri_lt(occ_orbs[k] - n, o1 - n)]; norm += prob_arr[k]; }
    const double inv = 1. / norm;
    for (unsigned int k = 0; k < o1_idx; k++) prob_arr[k] *= inv;
    return norm / tens->s_tens[o1 % n];
}
/* first virtual orbital: the unoccupied orbitals of the first electron's spin in ascending order.  (When the first electron is the last
 * one, the reference looks one entry past the occupied list, heat_bathPP.cpp:300-301; inside a DistVec that byte is the next row's first
 * alpha orbital and can never match: past the list nothing is occupied.) */
inline double calc_u1_probs(hb_info *tens, double *prob_arr, uint8_t o1_orb, uint8_t *occ_orbs, uint8_t n_elec, int exclude_first) {
    const unsigned int n = (unsigned int)tens->n_orb, sp = o1_orb / n, o1 = o1_orb % n, shift = sp * n;
    auto occ_at = [&](unsigned int k) -> unsigned int { return k < n_elec ? occ_orbs[k] : 255u; };
    unsigned int at = (n_elec / 2) * sp, n_out = 0;
    double norm = 0;
    for (unsigned int k = 0; k < o1; k++) {
        if (k + shift == occ_at(at)) at++;
        else { prob_arr[n_out] = tens->exch_sqrt[fries_hip::tri_lt(k, o1)]; norm += prob_arr[n_out]; n_out++; }
    }
    at++;
    for (unsigned int k = o1 + 1; k < n; k++) {
        if (k + shift == occ_at(at)) { if (at < (unsigned int)n_elec - 1) at++; }
        else { prob_arr[n_out] = tens->exch_sqrt[fries_hip::tri_lt(o1, k)]; norm += prob_arr[n_out]; n_out++; }
    }
    if (exclude_first) { norm -= prob_arr[0]; prob_arr[0] = 0; }
    const double inv = 1. / norm;
    for (unsigned int k = 0; k < n_out; k++) prob_arr[k] *= inv;
    return norm / tens->exch_norms[o1];
}
/* second virtual orbital: the orbitals of the irrep that closes the symmetry product, in the order of the symmetry table (occupied ones
 * included: the caller rejects them) */
inline double calc_u2_probs(hb_info *tens, double *prob_arr, uint8_t o1_orb, uint8_t o2_orb, uint8_t u1_orb, SymmInfo *symm, uint16_t *prob_len) {
    const unsigned int n = (unsigned int)tens->n_orb, o2 = o2_orb % n, u1 = u1_orb % n;
    const bool par = (o1_orb / n) == (o2_orb / n);
    const unsigned int ir = symm->symm_vec[o1_orb % n] ^ symm->symm_vec[o2] ^ symm->symm_vec[u1];
    const unsigned int len = symm->symm_lookup(ir, 0);
    *prob_len = (uint16_t)len;
    double norm = 0;
    for (unsigned int k = 0; k < len; k++) {
        const unsigned int u2 = symm->symm_lookup(ir, k + 1);
        if (par && u2 == u1) prob_arr[k] = 0;
        else { prob_arr[k] = fries_hip::hb_pair_sqrt(tens, o2, u2); norm += prob_arr[k]; }
    }
    if (norm != 0) {
        const double inv = 1 / norm;
        for (unsigned int k = 0; k < len; k++) if (!(par && symm->symm_lookup(ir, k + 1) == u1)) prob_arr[k] *= inv;
    }
    return norm / tens->exch_norms[o2];
}
/* the same restricted to unoccupied orbitals below the first virtual (parallel spins) */
inline double calc_u2_probs_half(hb_info *tens, double *prob_arr, uint8_t o1_orb, uint8_t o2_orb, uint8_t u1_orb, uint8_t *det, SymmInfo *symm, uint16_t *prob_len) {
    const unsigned int n = (unsigned int)tens->n_orb, o2 = o2_orb % n, u1 = u1_orb % n, sp2 = o2_orb / n;
    const bool par = (o1_orb / n) == sp2;
    const unsigned int ir = symm->symm_vec[o1_orb % n] ^ symm->symm_vec[o2] ^ symm->symm_vec[u1];
    const unsigned int len = symm->symm_lookup(ir, 0);
    double norm = 0;
    unsigned int k = 0;
    for (; k < len; k++) {
        const unsigned int u2 = symm->symm_lookup(ir, k + 1);
        if (par && u2 >= u1) break;
        if (fries_hip::det_bit(det, u2 + n * sp2)) prob_arr[k] = 0;
        else { prob_arr[k] = fries_hip::hb_pair_sqrt(tens, o2, u2); norm += prob_arr[k]; }
    }
    *prob_len = (uint16_t)k;
    if (norm != 0) { const double inv = 1 / norm; for (unsigned int q = 0; q < k; q++) prob_arr[q] *= inv; }
    return norm / tens->exch_norms[o2];
}
/* weight of a double excitation in the unnormalised factorisation (heat_bathPP.cpp:414-437) */
inline double calc_unnorm_wt(hb_info *tens, uint8_t *orbs) {
    const unsigned int n = (unsigned int)tens->n_orb;
    const unsigned int o1 = orbs[0] % n, o2 = orbs[1] % n, u1 = orbs[2] % n, u2 = orbs[3] % n;
    const double e1 = tens->exch_sqrt[o1 < u1 ? fries_hip::tri_lt(o1, u1) : fries_hip::tri_lt(u1, o1)], e2 = tens->exch_sqrt[o2 < u2 ? fries_hip::tri_lt(o2, u2) : fries_hip::tri_lt(u2, o2)];
    if ((orbs[0] / n) == (orbs[1] / n)) return tens->d_same[fries_hip::tri_lt(o1, o2)] * (e1 * e2) / tens->s_norm / tens->exch_norms[o1] / tens->exch_norms[o2];
    return tens->d_diff[o2 * n + o1] * e1 * e2 / tens->s_norm / tens->exch_norms[o1] / tens->exch_norms[o2];
}
/* probability with which the normalised factorisation proposes the excitation orbs = (o1 < o2, u1 < u2) from this determinant, summed
 * over the orders in which its four orbitals can be drawn (heat_bathPP.cpp:440-598) */
inline double calc_norm_wt(hb_info *tens, uint8_t *orbs, uint8_t *occ, unsigned int n_elec, uint8_t *det, SymmInfo *symm) {
    const unsigned int n = (unsigned int)tens->n_orb, half = n_elec / 2;
    const unsigned int o1 = orbs[0] % n, o2 = orbs[1] % n, u1 = orbs[2] % n, u2 = orbs[3] % n;
    const unsigned int sp1 = orbs[0] / n, sp2 = orbs[1] / n;
    const bool par = sp1 == sp2;
    uint8_t os[257];
    for (unsigned int k = 0; k < n_elec; k++) os[k] = (uint8_t)(occ[k] % n);
    os[n_elec] = 255;
    double s_all = 0;
    for (unsigned int k = 0; k < n_elec; k++) s_all += tens->s_tens[os[k]];
    auto pair_sum = [&](unsigned int o, unsigned int sp) {            // sum over the partners of an electron in orbital o with spin sp
        double d = 0;
        unsigned int k, off = (1 - sp) * half;
        for (k = off; k < off + half; k++) d += tens->d_diff[o * n + os[k]];
        off = sp * half;
        for (k = off; os[k] < o; k++) d += tens->d_same[fries_hip::tri_lt(os[k], o)];
        for (k++; k < off + half; k++) d += tens->d_same[fries_hip::tri_lt(o, os[k])];
        return d;
    };
    const double d1 = pair_sum(o1, sp1), d2 = pair_sum(o2, sp2);
    auto virt_sum = [&](unsigned int o, unsigned int sp) {            // sum over the unoccupied orbitals of spin sp
        double e = 0;
        const unsigned int off = sp * n;
        for (unsigned int k = 0; k < o; k++) if (!fries_hip::det_bit(det, k + off)) e += tens->exch_sqrt[fries_hip::tri_lt(k, o)];
        for (unsigned int k = o + 1; k < n; k++) if (!fries_hip::det_bit(det, k + off)) e += tens->exch_sqrt[fries_hip::tri_lt(o, k)];
        return e;
    };
    const double e1v = virt_sum(o1, sp1), e2v = virt_sum(o2, sp2);
    const unsigned int ir1 = symm->symm_vec[u1], ir2 = symm->symm_vec[u2];
    double e2s_no1 = 0, e1s_no1 = 0, e2s_no2 = 0, e1s_no2 = 0;
    for (unsigned int k = 0; k < symm->symm_lookup(ir2, 0); k++) {
        const unsigned int so = symm->symm_lookup(ir2, k + 1);
        if (par && so == u1) continue;
        e2s_no1 += fries_hip::hb_pair_sqrt(tens, o2, so);
        e1s_no1 += fries_hip::hb_pair_sqrt(tens, o1, so);
    }
    for (unsigned int k = 0; k < symm->symm_lookup(ir1, 0); k++) {
        const unsigned int so = symm->symm_lookup(ir1, k + 1);
        if (par && so == u2) continue;
        e2s_no2 += fries_hip::hb_pair_sqrt(tens, o2, so);
        e1s_no2 += fries_hip::hb_pair_sqrt(tens, o1, so);
    }
    const double x11 = fries_hip::hb_pair_sqrt(tens, o1, u1), x22 = fries_hip::hb_pair_sqrt(tens, o2, u2);
    if (par) {
        const double x12 = fries_hip::hb_pair_sqrt(tens, o1, u2), x21 = fries_hip::hb_pair_sqrt(tens, o2, u1);
        return tens->d_same[fries_hip::tri_lt(o1, o2)] / s_all * (
            tens->s_tens[o1] / d1 / e1v * (x11 * x22 / e2s_no1 + x12 * x21 / e2s_no2) +
            tens->s_tens[o2] / d2 / e2v * (x21 * x12 / e1s_no1 + x22 * x11 / e1s_no2));
    }
    return (tens->s_tens[o1] * tens->d_diff[o1 * n + o2] / d1 / e1v / e2s_no1 + tens->s_tens[o2] * tens->d_diff[o2 * n + o1] / d2 / e2v / e1s_no2) * x11 * x22 / s_all;
}

/* num_sampl double excitations from the normalised heat-bath factorisation (heat_bathPP.cpp:601-683): the first occupied orbital of every
 * sample, then -- electron by electron -- the second occupied and the first virtual orbital of all samples that chose it, then the second
 * virtual one sample at a time; a sample whose last factor has no weight, or whose second virtual is occupied, is dropped.  Alias
 * sampling, two draws each.  The samples are compacted towards the front of chosen_orbs as the reference does. */
inline unsigned int hb_doub_multi(uint8_t *det, uint8_t *occ_orbs, unsigned int num_elec, SymmInfo *symm, hb_info *tens,
                                  unsigned int num_sampl, std::mt19937 &mt_obj, uint8_t (*chosen_orbs)[4], double *prob_vec) {
    const unsigned int n_orb = (unsigned int)tens->n_orb, n_virt = n_orb - num_elec / 2;
    const unsigned int width = num_elec > n_virt ? num_elec : n_virt;
    std::vector<unsigned int> alias(width > 256 ? width : 256);
    std::vector<double> alias_p(alias.size()), row(alias.size());
    calc_o1_probs(tens, row.data(), num_elec, occ_orbs, 0);
    setup_alias(row.data(), alias.data(), alias_p.data(), num_elec);
    sample_alias(alias.data(), alias_p.data(), num_elec, chosen_orbs[0], num_sampl, 4, mt_obj);
    std::vector<unsigned int> per_elec(num_elec, 0);
    for (unsigned int k = 0; k < num_sampl; k++) per_elec[chosen_orbs[k][0]]++;
    unsigned int n_out = 0;
    for (unsigned int e = 0; e < num_elec; e++) {
        const unsigned int here = per_elec[e];
        if (!here) continue;
        const unsigned int first = n_out;
        calc_o2_probs(tens, row.data(), num_elec, occ_orbs, (uint8_t)e);
        const uint8_t o1 = occ_orbs[e];
        setup_alias(row.data(), alias.data(), alias_p.data(), num_elec);
        sample_alias(alias.data(), alias_p.data(), num_elec, &chosen_orbs[first][1], here, 4, mt_obj);
        calc_u1_probs(tens, row.data(), o1, occ_orbs, (uint8_t)num_elec, 0);
        setup_alias(row.data(), alias.data(), alias_p.data(), n_virt);
        sample_alias(alias.data(), alias_p.data(), n_virt, &chosen_orbs[first][2], here, 4, mt_obj);
        for (unsigned int k = first; k < first + here; k++) {
            const uint8_t o2 = occ_orbs[chosen_orbs[k][1]];
            const uint8_t u1 = find_nth_virt(occ_orbs, o1 / n_orb, (uint8_t)num_elec, (uint8_t)n_orb, chosen_orbs[k][2]);
            const unsigned int ir = symm->symm_vec[o1 % n_orb] ^ symm->symm_vec[o2 % n_orb] ^ symm->symm_vec[u1 % n_orb];
            uint16_t len = 0;
            if (calc_u2_probs(tens, row.data(), o1, o2, u1, symm, &len) == 0) continue;
            setup_alias(row.data(), alias.data(), alias_p.data(), len);
            uint8_t pick;
            sample_alias(alias.data(), alias_p.data(), len, &pick, 1, 1, mt_obj);
            const uint8_t u2 = (uint8_t)(symm->symm_lookup(ir, pick + 1) + n_orb * (o2 / n_orb));
            if (read_bit(det, u2)) continue;
            uint8_t *out = chosen_orbs[n_out];
            out[0] = o1 < o2 ? o1 : o2; out[1] = o1 < o2 ? o2 : o1;
            out[2] = u1 < u2 ? u1 : u2; out[3] = u1 < u2 ? u2 : u1;
            prob_vec[n_out] = calc_norm_wt(tens, out, occ_orbs, num_elec, det, symm);
            n_out++;
        }
    }
    return n_out;
}

struct HBCompress {
    std::vector<double> vec1;               ///< in: |values| of the vector; out: the surviving matrix elements
    size_t vec_len;                         ///< in: length of the vector; out: number of surviving elements
    std::vector<size_t> det_indices1;       ///< in: position of each input element in the vector
    std::vector<size_t> det_indices2;       ///< out: position of the origin determinant of each surviving element
    uint8_t (*orb_indices1)[4];             ///< out: (o1, o2, u1, u2) or (o, u, 0, 0)
    uint8_t (*orb_indices2)[4];
    std::vector<uint16_t> group_sizes;
    HBCompress(size_t length) : vec1(length), vec_len(0), det_indices1(length), det_indices2(length) {
        orb_indices1 = (uint8_t (*)[4])malloc(sizeof(uint8_t) * 4 * (length ? length : 1));
        orb_indices2 = nullptr;
    }
    HBCompress(const HBCompress &) = delete;
    HBCompress &operator=(const HBCompress &) = delete;
    ~HBCompress() { free(orb_indices1); }
};
struct HBCompressSys : HBCompress {
    std::vector<double> vec2;
    Matrix<double> subwts;
    std::vector<uint32_t> ndiv;
    Matrix<bool> keep_sub;
    std::vector<double> wt_remain;
    size_t (*comp_idx)[2];
    std::vector<uint32_t> pos32;            // staging of the device's 32-bit positions
    HBCompressSys(size_t length, size_t /*n_subwt*/) : HBCompress(length), subwts(1, 1), keep_sub(1, 1), comp_idx(nullptr), pos32(length) {}
};

inline void apply_HBPP_sys(Matrix<uint8_t> & /*all_orbs*/, Matrix<uint8_t> &all_dets, HBCompressSys *comp_scratch,
                           hb_info * /*hb_probs*/, SymmInfo * /*symm*/, double p_doub, bool new_hb,
                           std::mt19937 &mt_obj, uint32_t n_samp,
                           std::function<double(uint8_t *, uint8_t *)> /*sing_mat_fxn*/,
                           std::function<double(uint8_t *)> /*doub_mat_fxn*/) {
    fries_hip::DeviceVecBase *v = fries_hip::Backend::get().by_indices(&all_dets);
    if (!v) throw std::runtime_error("apply_HBPP_sys: all_dets must be the indices() matrix of the solution DistVec (this build runs the operator on the device-resident vector; there is no host implementation)");
    const size_t spawn_length = comp_scratch->vec1.size();
    // rank 0 draws the five uniforms (one before each comp_sub) and everybody uses them (heat_bathPP.cpp:729; compress_utils.cpp:806 broadcasts)
    double rn[5] = {0, 0, 0, 0, 0};
    if (fries_hip::mpi_rank() == 0) for (int k = 0; k < 5; k++) rn[k] = mt_obj() / (1. + UINT32_MAX);
    MPI_Bcast(rn, 5, MPI_DOUBLE, 0, MPI_COMM_WORLD);
    if (!v->bound()) {
        // spawn_length = mat_nonz * 4 / n_procs in the drivers (frisys_mol.cpp:109): the global budget sizes the device's work arrays
        const size_t glob = spawn_length / 4 * (size_t)fries_hip::mpi_size();
        const uint32_t mat_nonz = (uint32_t)(glob > n_samp ? glob : n_samp);
        v->bind(mat_nonz, new_hb);
        if (fries_p_doub(v->ctx()) != p_doub) throw std::runtime_error("apply_HBPP_sys: p_doub differs from the Hartree-Fock excitation counts the device computed");
    }
    v->before_device_op();
    size_t n_out = 0;
    uint32_t stage_len[5];
    fries_hip::ck(fries_apply_hbpp_sys(v->ctx(), n_samp, rn, 0, comp_scratch->pos32.data(), (uint8_t *)comp_scratch->orb_indices1, comp_scratch->vec1.data(), spawn_length, &n_out, stage_len));
    for (size_t i = 0; i < n_out; i++) comp_scratch->det_indices2[i] = comp_scratch->pos32[i];
    comp_scratch->vec_len = n_out;
}
/* pivotal variant (heat_bathPP.hpp:303-310, 353-357): every factor multiplied out into long_vec, compressed by piv_comp_parallel and
 * collapsed.  The number of uniforms drawn depends on the data, so the caller's generator is lent to the device context (its state
 * travels as text, the standard's operator<< format) and taken back afterwards.  spin_parity = +-1: fries_set_spin_parity around the call. */
struct HBCompressPiv : HBCompress {
    std::vector<double> long_vec;
    std::vector<bool> keep_idx;
    std::vector<size_t> cmp_srt;
    std::vector<uint32_t> pos32;
    HBCompressPiv(size_t length, size_t /*n_subwt*/) : HBCompress(length), pos32(length) {}
};
inline void apply_HBPP_piv(Matrix<uint8_t> & /*all_orbs*/, Matrix<uint8_t> &all_dets, HBCompressPiv *comp_scratch,
                           hb_info * /*hb_probs*/, SymmInfo * /*symm*/, double p_doub, bool new_hb,
                           std::mt19937 &mt_obj, uint32_t n_samp,
                           std::function<double(uint8_t *, uint8_t *)> /*sing_mat_fxn*/,
                           std::function<double(uint8_t *)> /*doub_mat_fxn*/, int spin_parity) {
    fries_hip::DeviceVecBase *v = fries_hip::Backend::get().by_indices(&all_dets);
    if (!v) throw std::runtime_error("apply_HBPP_piv: all_dets must be the indices() matrix of the solution DistVec (this build runs the operator on the device-resident vector; there is no host implementation)");
    const size_t cap = comp_scratch->vec1.size();
    if (!v->bound()) {
        const size_t glob = cap / 4 * (size_t)fries_hip::mpi_size();
        v->bind((uint32_t)(glob > n_samp ? glob : n_samp), new_hb);
        if (fries_p_doub(v->ctx()) != p_doub) throw std::runtime_error("apply_HBPP_piv: p_doub differs from the Hartree-Fock excitation counts the device computed");
    }
    v->before_device_op();
    std::ostringstream os; os << mt_obj;
    fries_hip::ck(fries_rng_set_state(v->ctx(), os.str().c_str()));
    size_t n_out = 0;
    uint32_t stage_len[5];
    fries_hip::ck(fries_set_spin_parity(v->ctx(), spin_parity));       // time-reversal symmetry (heat_bathPP.cpp:1326-1407) for this call
    const int rc = fries_apply_hbpp_piv(v->ctx(), n_samp, 0, comp_scratch->pos32.data(), (uint8_t *)comp_scratch->orb_indices1, comp_scratch->vec1.data(), cap, &n_out, stage_len);
    fries_set_spin_parity(v->ctx(), 0);
    fries_hip::ck(rc);
    size_t need = 0;
    fries_hip::ck(fries_rng_get_state(v->ctx(), nullptr, 0, &need));
    std::string st(need + 1, '\0');
    fries_hip::ck(fries_rng_get_state(v->ctx(), &st[0], st.size(), &need));
    std::istringstream is(st.c_str()); is >> mt_obj;
    for (size_t i = 0; i < n_out; i++) comp_scratch->det_indices2[i] = comp_scratch->pos32[i];
    comp_scratch->vec_len = n_out;
}
#endif /* heat_bathPP_h */
