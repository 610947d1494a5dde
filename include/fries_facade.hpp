// C++ facade over the C ABI of libfries_hip.so (include/fries_hip.h): the reference's names and call shapes for the
// objects and free functions its drivers use on the frisys_mol path, so that a driver loop written against
// <FRIES/vec_utils.hpp>, <FRIES/compress_utils.hpp> and <FRIES/Hamiltonians/heat_bathPP.hpp> reads the same here
// (tests/cpp/test_facade.cpp is FRIES_bin/frisys_mol.cpp:405-552 written this way).  Header only; errors surface as
// std::runtime_error like the reference's (frisys_mol.cpp:562-565).  The truth lives in HBM: values() / indices() are
// host mirrors refreshed on demand, for inspection and for building spawns on the host as the reference's loop does.
#pragma once
#include "fries_hip.h"
#include <cmath>
#include <cstring>
#include <random>
#include <sstream>
#include <stdexcept>
#include <vector>

namespace fries_hip {

inline void ck(int rc) { if (rc) throw std::runtime_error(fries_last_error()); }

// FRIES/ndarr.hpp:24-150 (the members the drivers touch)
template <class T> class Matrix {
    size_t rows_ = 0, cols_ = 0;
    std::vector<T> data_;
public:
    Matrix() {}
    Matrix(size_t rows, size_t cols) : rows_(rows), cols_(cols), data_(rows * cols) {}
    void reshape(size_t rows, size_t cols) { rows_ = rows; cols_ = cols; data_.resize(rows * cols); }
    size_t rows() const { return rows_; }
    size_t cols() const { return cols_; }
    T &operator()(size_t r, size_t c) { return data_[r * cols_ + c]; }
    T *operator[](size_t r) { return &data_[r * cols_]; }
    const T *operator[](size_t r) const { return &data_[r * cols_]; }
    T *data() { return data_.data(); }
};

// FRIES/fci_utils.c:60-64, 77-83 and 9-43: excited determinants and the Hartree-Fock string on byte strings
inline void zero_bit(uint8_t *s, unsigned b) { s[b / 8] &= (uint8_t)~(1u << (b % 8)); }
inline void set_bit(uint8_t *s, unsigned b) { s[b / 8] |= (uint8_t)(1u << (b % 8)); }
inline void sing_det(uint8_t *det, const uint8_t *orbs) { zero_bit(det, orbs[0]); set_bit(det, orbs[1]); }
inline void doub_det(uint8_t *det, const uint8_t *orbs) { zero_bit(det, orbs[0]); zero_bit(det, orbs[1]); set_bit(det, orbs[2]); set_bit(det, orbs[3]); }
inline void gen_hf_bitstring(unsigned n_orb, unsigned n_elec, uint8_t *det) {
    memset(det, 0, (2 * n_orb + 7) / 8);
    for (unsigned sp = 0; sp < 2; sp++) for (unsigned i = 0; i < n_elec / 2; i++) set_bit(det, i + sp * n_orb);
}
// FRIES/compress_utils.cpp:684-693
inline void adjust_shift(double *shift, double one_norm, double *last_norm, double target_norm, double damp_factor) {
    if (*last_norm) { *shift -= damp_factor * log(one_norm / *last_norm); *last_norm = one_norm; }
    if (*last_norm == 0 && one_norm > target_norm) *last_norm = one_norm;
}

// FRIES/Hamiltonians/heat_bathPP.hpp:250-297: what apply_HBPP_sys hands back to the driver
struct HBCompressSys {
    std::vector<double> vec1;
    std::vector<size_t> det_indices1, det_indices2;
    Matrix<uint8_t> orb_indices1;
    size_t vec_len = 0;
    HBCompressSys(size_t length, size_t /*n_states*/) : vec1(length), det_indices1(length), det_indices2(length), orb_indices1(length, 4) {}
};

template <class el_type> class DistVec;

// One device, one molecule, one solution vector: what a FRIES main builds between parse_fcidump and its iteration loop
// (frisys_mol.cpp:76-346).  setup() stands for: the scramblers drawn from the seed, the DistVec and Adder, the HF trial
// vector and H * trial, p_doub, the HB-PP tensors and the start from 100 x HF.
class Engine {
    fries_ctx *ctx_ = nullptr;
    unsigned n_orb_ = 0, n_elec_ = 0;
public:
    explicit Engine(int device = 0) { ck(fries_ctx_create(&ctx_, device)); }
    ~Engine() { if (ctx_) fries_ctx_destroy(ctx_); }
    Engine(const Engine &) = delete;
    Engine &operator=(const Engine &) = delete;
    void set_molecule(unsigned n_orb, unsigned n_elec, const uint8_t *symm, const double *h_core, const double *eris_packed) {
        ck(fries_set_molecule(ctx_, n_orb, n_elec, symm, h_core, eris_packed));
        n_orb_ = n_orb; n_elec_ = n_elec;
    }
    void setup(const fries_frisys_params &p) { ck(fries_frisys_setup(ctx_, &p)); }
    fries_ctx *ctx() { return ctx_; }
    unsigned n_orb() const { return n_orb_; }
    unsigned n_elec() const { return n_elec_; }
    double p_doub() { return fries_p_doub(ctx_); }
    // the fused loop body (frisys_mol.cpp:405-552), for comparison with one assembled from the operators below
    fries_iter_log iterate() { fries_iter_log lg; ck(fries_frisys_iterate(ctx_, 1, &lg)); return lg; }
};

// FRIES/vec_utils.hpp:121-953 for el_type = double, the members the frisys_mol loop uses
template <> class DistVec<double> {
    Engine &eng_;
    size_t add_size_;
    uint8_t n_bytes_, curr_vec_idx_ = 0;
    std::vector<uint64_t> a_det_; std::vector<double> a_val_; std::vector<uint8_t> a_ini_;     // the Adder (vec_utils.hpp:957-1019)
    std::vector<double> values_; Matrix<uint8_t> indices_; bool fresh_ = false;
    void refresh() {
        if (fresh_) return;
        uint32_t n; int32_t nz; uint32_t nf;
        ck(fries_vec_info(eng_.ctx(), &n, &nz, &nf));
        std::vector<uint64_t> d(n ? n : 1);
        values_.assign(n ? n : 1, 0.0);
        size_t got = 0;
        ck(fries_vec_download(eng_.ctx(), d.data(), values_.data(), d.size(), &got));
        indices_.reshape(n ? n : 1, n_bytes_);
        for (size_t i = 0; i < n; i++) memcpy(indices_[i], &d[i], n_bytes_);      // byte string = little-endian index
        fresh_ = true;
    }
public:
    DistVec(Engine &eng, size_t add_size) : eng_(eng), add_size_(add_size), n_bytes_((uint8_t)((2 * eng.n_orb() + 7) / 8)) {}
    uint8_t n_bits() const { return (uint8_t)(2 * eng_.n_orb()); }
    uint8_t curr_vec_idx() const { return curr_vec_idx_; }
    void set_curr_vec_idx(uint8_t i) { if (i > 1) throw std::runtime_error("this DistVec has two value columns"); curr_vec_idx_ = i; }
    // vec_utils.hpp:418-431: zero values never reach the adder; false when the adder is full
    bool add(const uint8_t *idx, double val, uint8_t ini_flag) {
        if (val != 0) {
            uint64_t d = 0; memcpy(&d, idx, n_bytes_);
            a_det_.push_back(d); a_val_.push_back(val); a_ini_.push_back(ini_flag);
        }
        return a_det_.size() < add_size_;
    }
    void perform_add(size_t /*origin*/ = 0) {     // vec_utils.hpp:438-440
        ck(fries_vec_add_to(eng_.ctx(), curr_vec_idx_, a_det_.data(), a_val_.data(), a_ini_.data(), a_det_.size()));
        a_det_.clear(); a_val_.clear(); a_ini_.clear();
        fresh_ = false;
    }
    size_t curr_size() { uint32_t n; int32_t nz; uint32_t nf; ck(fries_vec_info(eng_.ctx(), &n, &nz, &nf)); return n; }
    int n_nonz() { uint32_t n; int32_t nz; uint32_t nf; ck(fries_vec_info(eng_.ctx(), &n, &nz, &nf)); return nz; }
    // host mirror of column 0 (column 1 only ever holds the spawns of the iteration in flight and is not mirrored)
    double *values() { if (curr_vec_idx_ != 0) throw std::runtime_error("values() mirrors column 0"); refresh(); return values_.data(); }
    Matrix<uint8_t> &indices() { refresh(); return indices_; }
    double *operator[](size_t pos) { return values() + pos; }
    // the device keeps column 1 zero between iterations (fries_death_clone ends with zero_vec on it, frisys_mol.cpp:497-498)
    void zero_vec() { if (curr_vec_idx_ != 1) throw std::runtime_error("zero_vec is provided for the spawn column"); }
    void invalidate() { fresh_ = false; }
    Engine &engine() { return eng_; }
};

// FRIES/Hamiltonians/heat_bathPP.cpp:686-992: draws its five uniforms from the caller's generator (:729, :765, :811, :859, :910)
inline void apply_HBPP_sys(DistVec<double> &vec, HBCompressSys *comp_vecs, std::mt19937 &mt_obj, uint32_t n_samp) {
    double rn[5];
    for (int k = 0; k < 5; k++) rn[k] = mt_obj() / (1. + UINT32_MAX);
    const size_t cap = comp_vecs->vec1.size();
    std::vector<uint32_t> pos(cap);
    size_t n_out = 0;
    uint32_t comp_len[5];
    ck(fries_apply_hbpp_sys(vec.engine().ctx(), n_samp, rn, 0, pos.data(), comp_vecs->orb_indices1.data(), comp_vecs->vec1.data(), cap, &n_out, comp_len));
    for (size_t i = 0; i < n_out; i++) comp_vecs->det_indices2[i] = pos[i];
    comp_vecs->vec_len = n_out;
}
// FRIES/Hamiltonians/heat_bathPP.hpp:303-311
struct HBCompressPiv {
    std::vector<double> vec1;
    std::vector<size_t> det_indices1, det_indices2;
    Matrix<uint8_t> orb_indices1;
    size_t vec_len = 0;
    size_t stage_len[5] = {0, 0, 0, 0, 0};      // elements after each of the five compressions (not in the reference's struct)
    HBCompressPiv(size_t length, size_t /*n_states*/) : vec1(length), det_indices1(length), det_indices2(length), orb_indices1(length, 4) {}
};
// the caller's generator travels to the device context and back (the number of draws depends on the data)
inline void lend_generator(Engine &eng, std::mt19937 &mt_obj) { std::ostringstream os; os << mt_obj; ck(fries_rng_set_state(eng.ctx(), os.str().c_str())); }
inline void return_generator(Engine &eng, std::mt19937 &mt_obj) {
    size_t need = 0;
    ck(fries_rng_get_state(eng.ctx(), nullptr, 0, &need));
    std::vector<char> buf(need);
    ck(fries_rng_get_state(eng.ctx(), buf.data(), buf.size(), nullptr));
    std::istringstream is(buf.data());
    is >> mt_obj;
}
// FRIES/Hamiltonians/heat_bathPP.cpp:1014-1419 (spin_parity 0) on the vector's column 0
inline void apply_HBPP_piv(DistVec<double> &vec, HBCompressPiv *comp_vecs, std::mt19937 &mt_obj, uint32_t n_samp) {
    const size_t cap = comp_vecs->vec1.size();
    std::vector<uint32_t> pos(cap);
    size_t n_out = 0;
    uint32_t sl[5];
    lend_generator(vec.engine(), mt_obj);
    ck(fries_apply_hbpp_piv(vec.engine().ctx(), n_samp, 0, pos.data(), comp_vecs->orb_indices1.data(), comp_vecs->vec1.data(), cap, &n_out, sl));
    return_generator(vec.engine(), mt_obj);
    for (size_t i = 0; i < n_out; i++) comp_vecs->det_indices2[i] = pos[i];
    for (int k = 0; k < 5; k++) comp_vecs->stage_len[k] = sl[k];
    comp_vecs->vec_len = n_out;
}
// FRIES/compress_utils.cpp:29-105: the preserved set stays on the device until sys_comp
inline void find_preserve(DistVec<double> &vec, unsigned int *n_samp, double *global_norm) {
    uint32_t ns = *n_samp;
    ck(fries_find_preserve(vec.engine().ctx(), &ns, global_norm));
    *n_samp = ns;
}
// FRIES/compress_utils.cpp:283-327 followed by the del_at_pos loop of frisys_mol.cpp:534-539
inline void sys_comp(DistVec<double> &vec, unsigned int n_samp, double rand_num) {
    ck(fries_sys_comp(vec.engine().ctx(), n_samp, rand_num));
    vec.invalidate();
}
// frisys_mol.cpp:487-499: matr_el_at_pos / death-cloning loop, add_vecs(0, 1), zero_vec on column 1
inline void death_clone_and_add(DistVec<double> &vec, double eps, double en_shift, size_t vec_size) {
    ck(fries_death_clone(vec.engine().ctx(), eps, en_shift, (uint32_t)vec_size));
    vec.invalidate();
}
// DistVec::dot against H * trial and the trial vector (frisys_mol.cpp:511-517)
inline void proj_dots(DistVec<double> &vec, double *numer, double *denom) { ck(fries_dots(vec.engine().ctx(), numer, denom)); }

}  // namespace fries_hip
