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
    double rn[5];
    for (int k = 0; k < 5; k++) rn[k] = mt_obj() / (1. + UINT32_MAX);
    if (!v->bound()) {
        const uint32_t mat_nonz = (uint32_t)(spawn_length / 4 > n_samp ? spawn_length / 4 : n_samp);
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
 * travels as text, the standard's operator<< format) and taken back afterwards.  spin_parity != 0 is not supported. */
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
    if (spin_parity) throw std::runtime_error("apply_HBPP_piv: time-reversal symmetrised vectors (spin_parity != 0) are not supported by this build");
    fries_hip::DeviceVecBase *v = fries_hip::Backend::get().by_indices(&all_dets);
    if (!v) throw std::runtime_error("apply_HBPP_piv: all_dets must be the indices() matrix of the solution DistVec (this build runs the operator on the device-resident vector; there is no host implementation)");
    const size_t cap = comp_scratch->vec1.size();
    if (!v->bound()) {
        v->bind((uint32_t)(cap / 4 > n_samp ? cap / 4 : n_samp), new_hb);
        if (fries_p_doub(v->ctx()) != p_doub) throw std::runtime_error("apply_HBPP_piv: p_doub differs from the Hartree-Fock excitation counts the device computed");
    }
    v->before_device_op();
    std::ostringstream os; os << mt_obj;
    fries_hip::ck(fries_rng_set_state(v->ctx(), os.str().c_str()));
    size_t n_out = 0;
    uint32_t stage_len[5];
    fries_hip::ck(fries_apply_hbpp_piv(v->ctx(), n_samp, 0, comp_scratch->pos32.data(), (uint8_t *)comp_scratch->orb_indices1, comp_scratch->vec1.data(), cap, &n_out, stage_len));
    size_t need = 0;
    fries_hip::ck(fries_rng_get_state(v->ctx(), nullptr, 0, &need));
    std::string st(need + 1, '\0');
    fries_hip::ck(fries_rng_get_state(v->ctx(), &st[0], st.size(), &need));
    std::istringstream is(st.c_str()); is >> mt_obj;
    for (size_t i = 0; i < n_out; i++) comp_scratch->det_indices2[i] = comp_scratch->pos32[i];
    comp_scratch->vec_len = n_out;
}
#endif /* heat_bathPP_h */
