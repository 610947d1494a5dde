/*! \file  FRIES/vec_utils.hpp for the MI355X build: Adder<el_type> and DistVec<el_type> with the reference's public members
 * (FRIES/vec_utils.hpp:50-118, 120-952), so that a driver written against the reference compiles unchanged.
 *
 * A DistVec starts on the host and behaves as in the reference (hash of determinant -> position, LIFO stack of freed positions,
 * buffered adds under the initiator rule) -- that is what trial vectors and H * trial use.  A DistVec<double> with two value columns
 * becomes DEVICE-BOUND at the first apply_HBPP_sys that receives its indices() matrix (FRIES/Hamiltonians/heat_bathPP.hpp of this
 * build); from then on the determinants, both value columns, the diagonal elements, the hash table and the free stack live in HBM
 * (fries_amd/csrc/vec.hip) and every member below is a call through include/fries_hip.h:
 *
 *   perform_add              fries_vec_add_to        (annihilating merge, same positions and the same order of additions)
 *   add_vecs / zero_vec      fries_vec_add_vecs / fries_vec_column_zero
 *   dot                      fries_vec_dot_list      (sum in list order)
 *   values() / operator[]    host mirror of the column, downloaded when stale, uploaded before the next device call
 *   indices()                host mirror of the determinants, downloaded when stale
 *   matr_el_at_pos           fries_vec_diag_download (diagonal elements evaluated on the device)
 *   curr_size / n_nonz       fries_vec_info
 *   save                     the reference's dets<rank>.dat / vals<rank>.dat / dense.txt
 *
 * Not carried over to a bound vector: del_at_pos (sys_comp already performs the drivers' deletes on the device), expand (the device
 * capacity is the max_size given at construction), the dense (semi-stochastic) prefix, writes through indices().  This host surface is
 * one rank (include/FRIES/compat/mpi.h); ranks of the engine go through fries_comm. */
#ifndef vec_utils_h
#define vec_utils_h
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <sstream>
#include <stack>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>
#include <sys/types.h>
#include <mpi.h>
#include <FRIES/det_store.h>
#include <FRIES/det_hash.hpp>
#include <FRIES/ndarr.hpp>
#include <FRIES/io_utils.hpp>
#include <FRIES/compress_utils.hpp>
#include <FRIES/backend.hpp>

template <class el_type> class DistVec;

/*! buffered adds (vec_utils.hpp:50-118); one destination in this build */
template <class el_type>
class Adder {
    size_t size_;
    uint8_t n_bytes_;
    std::vector<uint8_t> idx_;       // size_ x n_bytes_, initiator flag in bit n_bits of each index
    std::vector<el_type> vals_;
    int count_ = 0;
public:
    Adder(size_t size, int /*n_procs*/, uint8_t n_bits) : size_(size), n_bytes_((uint8_t)CEILING(n_bits + 1, 8)), idx_(size * CEILING(n_bits + 1, 8)), vals_(size) {}
    void perform_add(DistVec<el_type> *parent_vec, size_t origin);
    /* false once the buffer is full (vec_utils.hpp:956-971) */
    bool add(uint8_t *idx, uint8_t idx_bits, el_type val, int /*proc_idx*/, uint8_t ini_flag) {
        if ((size_t)count_ >= size_) throw std::runtime_error("Too many elements added to Adder - must call perform_add() more frequently.");
        uint8_t *dst = &idx_[(size_t)count_ * n_bytes_];
        dst[n_bytes_ - 1] = 0;
        memcpy(dst, idx, CEILING(idx_bits, 8));
        if (ini_flag) set_bit(dst, idx_bits);
        vals_[count_] = val;
        count_++;
        return (size_t)count_ < size_;
    }
    size_t size() const { return size_; }
    int pending() const { return count_; }
    uint8_t *idx_data() { return idx_.data(); }
    el_type *val_data() { return vals_.data(); }
    uint8_t n_bytes() const { return n_bytes_; }
    void clear() { count_ = 0; }
};

template <class el_type>
class DistVec : public fries_hip::DeviceVecBase {
    Matrix<el_type> values_;
    uint8_t curr_vec_idx_;
    size_t n_dense_;
    std::stack<size_t> vec_stack_;
    int n_nonz_;
    Adder<el_type> *adder_;
    size_t min_del_idx_;
    // device binding (DistVec<double>, two columns)
    bool bound_ = false;
    fries_ctx *ctx_ = nullptr;
    bool val_fresh_[2] = {false, false}, val_dirty_[2] = {false, false};
    size_t mirror_n_[2] = {0, 0};       // positions each value mirror covered when it was downloaded
    bool idx_fresh_ = false, diag_fresh_ = false, info_fresh_ = false;
    std::vector<uint32_t> rns_common_;
protected:
    Matrix<uint8_t> indices_;
    size_t max_size_;
    size_t curr_size_;
    Matrix<uint8_t> occ_orbs_;
    uint8_t n_bits_;
    HashTable<ssize_t> vec_hash_;
    HashTable<ssize_t> proc_hash_;
    uint64_t nonini_occ_add;
    std::vector<double> matr_el_;
    std::function<double(const uint8_t *)> diag_calc_;
    std::vector<bool> active_pos_;

    virtual void initialize_at_pos(size_t pos, uint8_t *orbs) {
        for (uint8_t v = 0; v < values_.rows(); v++) values_(v, pos) = 0;
        matr_el_[pos] = NAN;
        active_pos_[pos] = true;
        memcpy(occ_orbs_[pos], orbs, occ_orbs_.cols());
    }
    uint64_t word_at(size_t pos) const { return fries_word_of(indices_[pos], (uint8_t)indices_.cols()); }
    void refresh_info() {
        if (!bound_ || info_fresh_) return;
        uint32_t cs = 0, nf = 0; int32_t nn = 0;
        fries_hip::ck(fries_vec_info(ctx_, &cs, &nn, &nf));
        curr_size_ = cs; n_nonz_ = nn;
        info_fresh_ = true;
    }
    void refresh_col(uint8_t col) {
        if (!bound_ || val_fresh_[col]) return;
        if constexpr (std::is_same<el_type, double>::value) {
            refresh_info();
            size_t n = 0;
            fries_hip::ck(fries_vec_column_download(ctx_, col, values_[col], max_size_, &n));
            mirror_n_[col] = n;
            val_fresh_[col] = true;
        }
    }
    void refresh_idx() {
        if (!bound_ || idx_fresh_) return;
        refresh_info();
        std::vector<uint64_t> w(curr_size_ ? curr_size_ : 1);
        size_t n = 0;
        fries_hip::ck(fries_vec_download(ctx_, w.data(), nullptr, w.size(), &n));
        const size_t nb = indices_.cols();
        for (size_t i = 0; i < n; i++) memcpy(indices_[i], &w[i], nb);
        idx_fresh_ = true;
    }
public:
    DistVec(size_t size, Adder<el_type> *adder, uint8_t n_bits, unsigned int n_elec,
            std::function<double(const uint8_t *)> diag_fxn, uint8_t n_vecs,
            std::vector<uint32_t> rns_common, std::vector<uint32_t> rns_distinct) :
    values_(n_vecs, size), curr_vec_idx_(0), n_dense_(0), n_nonz_(0), adder_(adder), min_del_idx_(0), rns_common_(rns_common),
    indices_(size, CEILING(n_bits, 8)), max_size_(size), curr_size_(0), occ_orbs_(size, n_elec), n_bits_(n_bits),
    vec_hash_(size, rns_distinct), proc_hash_(0, rns_common), nonini_occ_add(0), matr_el_(size), diag_calc_(diag_fxn), active_pos_(size) {
        if (n_bits > 64) throw std::runtime_error("this build stores determinants in 64 bits (2 * n_orb <= 64)");
        fries_hip::Backend::get().add(this);
    }
    DistVec(size_t size, Adder<el_type> *adder, uint8_t n_bits, unsigned int n_elec,
            std::vector<uint32_t> rns_common, std::vector<uint32_t> rns_distinct) :
    DistVec(size, adder, n_bits, n_elec, nullptr, 1, rns_common, rns_distinct) {}
    DistVec(size_t size, size_t add_size, uint8_t n_bits, unsigned int n_elec, int n_procs,
            std::function<double(const uint8_t *)> diag_fxn, uint8_t n_vecs,
            std::vector<uint32_t> rns_common, std::vector<uint32_t> rns_distinct) :
    DistVec(size, new Adder<el_type>(add_size, n_procs, n_bits), n_bits, n_elec, diag_fxn, n_vecs, rns_common, rns_distinct) {}
    DistVec(size_t size, size_t add_size, uint8_t n_bits, unsigned int n_elec, int n_procs,
            std::vector<uint32_t> rns_common, std::vector<uint32_t> rns_distinct) :
    DistVec(size, add_size, n_bits, n_elec, n_procs, nullptr, 1, rns_common, rns_distinct) {}
    DistVec(const DistVec &) = delete;
    DistVec &operator=(const DistVec &) = delete;
    virtual ~DistVec() { fries_hip::Backend::get().remove(this); }

    // ---------------------------------------------------------------- device binding (fries_hip::DeviceVecBase)
    bool owns(const void *p) const override {
        const char *q = (const char *)p;
        const char *b = (const char *)values_.data();
        return q >= b && q < b + sizeof(el_type) * values_.rows() * values_.cols();
    }
    size_t offset_of(const void *p) const override { return (size_t)((const el_type *)p - values_.data()) % values_.cols(); }
    const void *indices_key() const override { return &indices_; }
    bool bound() const override { return bound_; }
    fries_ctx *ctx() override { return ctx_; }
    /* moves the vector to the device: positions, both columns and the rank hash are kept; mat_nonz sizes the operators' work arrays */
    void bind(uint32_t mat_nonz, bool new_hb) override {
        if (bound_) return;
        if constexpr (!std::is_same<el_type, double>::value) throw std::runtime_error("only a DistVec<double> can be bound to the device");
        else {
            fries_hip::Backend &B = fries_hip::Backend::get();
            if (values_.rows() != 2) throw std::runtime_error("the device vector has two value columns (n_vecs = 2)");
            if (B.ctx_taken) throw std::runtime_error("one device-bound solution vector per process");
            if (n_dense_) throw std::runtime_error("a dense (semi-stochastic) prefix is not supported on the device");
            if (adder_->pending()) throw std::runtime_error("perform_add() must run before the vector moves to the device");
            for (size_t i = 0; i < curr_size_; i++) {
                if (!active_pos_[i]) throw std::runtime_error("a vector with freed positions cannot be moved to the device (bind it before deleting)");
                if (values_(1, i) != 0) throw std::runtime_error("column 1 must be zero when the vector moves to the device");
            }
            fries_ctx *cx = B.ctx();
            if (rns_common_.size() != 2 * (size_t)B.n_orb) throw std::runtime_error("rns_common must hold 2 * n_orb numbers");
            fries_hip::ck(fries_set_proc_scrambler(cx, rns_common_.data(), rns_common_.size()));
            // diag_fxn is "diag_matrel(occ) - hf_en" in every driver: recover hf_en from the HF determinant (exact when it is within a
            // factor two of the HF energy, Sterbenz) and verify the device's diagonal elements against diag_fxn below
            std::vector<uint8_t> hf_occ(occ_orbs_.cols());
            for (unsigned k = 0; k < B.n_elec / 2; k++) { hf_occ[k] = (uint8_t)k; hf_occ[k + B.n_elec / 2] = (uint8_t)(k + B.n_orb); }
            const double d_hf = fries_hf_energy(cx);
            if (diag_calc_) {
                const double lam = diag_calc_(hf_occ.data());
                if (lam != 0) fries_hip::ck(fries_set_ham_shift(cx, d_hf - lam));
            }
            fries_frisys_params p{};
            p.epsilon = 0; p.target_norm = 0; p.initiator = 0;
            p.max_dets = (uint32_t)max_size_; p.vec_nonz = (uint32_t)max_size_; p.mat_nonz = mat_nonz; p.seed = 0; p.hb_unnorm = new_hb ? 1 : 0;
            fries_hip::ck(fries_frisys_setup(cx, &p));
            std::vector<uint64_t> w(curr_size_ ? curr_size_ : 1);
            for (size_t i = 0; i < curr_size_; i++) w[i] = word_at(i);
            fries_hip::ck(fries_vec_load(cx, w.data(), values_[0], curr_size_));
            ctx_ = cx; bound_ = true; B.ctx_taken = true;
            val_fresh_[0] = true; val_fresh_[1] = true; val_dirty_[0] = val_dirty_[1] = false;
            mirror_n_[0] = mirror_n_[1] = curr_size_;
            idx_fresh_ = true; diag_fresh_ = false; info_fresh_ = false;
            if (diag_calc_ && curr_size_) {
                const size_t n_chk = curr_size_ < 8 ? curr_size_ : 8;
                for (size_t i = 0; i < n_chk; i++) {
                    const double host = diag_calc_(occ_orbs_[i]), dev = matr_el_at_pos(i);
                    if (host != dev) throw std::runtime_error("the diagonal elements of diag_fxn are not diag_matrel(occ) - const: this build cannot evaluate them on the device");
                }
            }
        }
    }
    void before_device_op() override {
        if (!bound_) return;
        if constexpr (std::is_same<el_type, double>::value) {
            for (int col = 0; col < 2; col++) if (val_dirty_[col]) {
                fries_hip::ck(fries_vec_column_upload(ctx_, col, values_[col], mirror_n_[col]));
                val_dirty_[col] = false;
            }
        }
    }
    void after_device_op(bool col0, bool col1, bool layout) override {
        if (col0) val_fresh_[0] = false;
        if (col1) val_fresh_[1] = false;
        if (layout) { idx_fresh_ = false; diag_fresh_ = false; info_fresh_ = false; }
    }

    // ---------------------------------------------------------------- the reference's members
    uint8_t n_bits() { return n_bits_; }
    virtual uint8_t gen_orb_list(uint8_t *det, uint8_t *occ_orbs) { return find_bits(det, occ_orbs, (uint8_t)indices_.cols()); }

    /* vec_utils.hpp:228-252 */
    double dot(Matrix<uint8_t> &idx2, double *vals2, size_t num2, std::vector<uintmax_t> & /*hashes2*/) { return dot(idx2, vals2, num2); }
    double dot(Matrix<uint8_t> &idx2, double *vals2, size_t num2) {
        if (bound_) {
            if constexpr (std::is_same<el_type, double>::value) {
                before_device_op();
                std::vector<uint64_t> w(num2 ? num2 : 1);
                for (size_t i = 0; i < num2; i++) w[i] = fries_word_of(idx2[i], (uint8_t)idx2.cols());
                double r = 0;
                fries_hip::ck(fries_vec_dot_list(ctx_, curr_vec_idx_, w.data(), vals2, num2, &r));
                return r;
            }
        }
        double numer = 0;
        for (size_t i = 0; i < num2; i++) {
            ssize_t *ht_ptr = vec_hash_.read(idx2[i], 0, false);
            if (ht_ptr) numer += vals2[i] * values_(curr_vec_idx_, *ht_ptr);
        }
        return numer;
    }
    /* vec_utils.hpp:324-340 */
    double internal_dot(uint8_t idx1, uint8_t idx2) {
        if (idx1 >= values_.rows() || idx2 >= values_.rows()) throw std::runtime_error("Error: argument to internal_dot exceeds bounds of value matrix");
        refresh_col(idx1); refresh_col(idx2); refresh_info();
        double dprod = 0;
        for (size_t i = 0; i < curr_size_; i++) dprod += values_(idx1, i) * values_(idx2, i);
        return dprod;
    }

    virtual void expand() {
        if (bound_) throw std::runtime_error("the device vector cannot grow beyond the max_size it was constructed with");
        size_t new_max = max_size_ * 2;
        std::cout << "Increasing storage capacity in vector to " << new_max << "\n";
        indices_.reshape(new_max, indices_.cols());
        active_pos_.resize(new_max);
        matr_el_.resize(new_max);
        occ_orbs_.reshape(new_max, occ_orbs_.cols());
        values_.enlarge_cols(new_max, (int)curr_size_);
        max_size_ = new_max;
    }

    virtual int idx_to_proc(uint8_t *idx) {
        uint8_t orbs[256];
        gen_orb_list(idx, orbs);
        return idx_to_proc(idx, orbs);
    }
    virtual int idx_to_proc(uint8_t * /*idx*/, uint8_t *orbs) {
        uintmax_t hash_val = proc_hash_.hash_fxn(orbs, (uint8_t)occ_orbs_.cols(), NULL, 0);
        int n_procs = 1;
        MPI_Comm_size(MPI_COMM_WORLD, &n_procs);
        return (int)(hash_val % n_procs);
    }
    virtual uintmax_t idx_to_hash(uint8_t *idx, uint8_t *orbs) {
        unsigned int n_elec = (unsigned int)occ_orbs_.cols();
        if (gen_orb_list(idx, orbs) != n_elec) {
            char det_txt[2 * 8 + 1];
            print_str(idx, (uint8_t)indices_.cols(), det_txt);
            std::stringstream error;
            error << "Determinant " << det_txt << " created with an incorrect number of electrons";
            throw std::runtime_error(error.str());
        }
        return vec_hash_.hash_fxn(orbs, (uint8_t)n_elec, NULL, 0);
    }
    void print_ht() { vec_hash_.print_ht(); }

    /* vec_utils.hpp:418-436 */
    bool add(uint8_t *idx, el_type val, uint8_t ini_flag) {
        if (val != 0) return adder_->add(idx, n_bits_, val, 0, ini_flag);
        return true;
    }
    bool add(uint8_t *idx, uint8_t * /*orbs*/, el_type val, uint8_t ini_flag) { return adder_->add(idx, n_bits_, val, 0, ini_flag); }
    void perform_add(size_t origin) { adder_->perform_add(this, origin); }

    ssize_t pop_stack() {
        if (vec_stack_.empty()) return -1;
        ssize_t ret_idx = (ssize_t)vec_stack_.top();
        vec_stack_.pop();
        return ret_idx;
    }
    /* vec_utils.hpp:458-476 */
    void del_at_pos(size_t pos) {
        if (bound_) throw std::runtime_error("del_at_pos on a device-bound vector: sys_comp() of this build already deletes the zeroed elements on the device");
        if (!active_pos_[pos]) return;
        bool all_zero = true;
        for (uint8_t v = 0; v < values_.rows(); v++) if (values_(v, pos) != 0) all_zero = false;
        if (pos >= min_del_idx_ && all_zero) {
            uint8_t *idx = indices_[pos];
            vec_stack_.push(pos);
            vec_hash_.del_entry(idx, 0);
            n_nonz_--;
            active_pos_[pos] = false;
        }
    }
    void cleanup() {
        for (size_t pos = min_del_idx_; pos < curr_size_; pos++) {
            bool all_zero = true;
            for (uint8_t v = 0; v < values_.rows(); v++) if (values_(v, pos) != 0) all_zero = false;
            if (all_zero) del_at_pos(pos);
        }
    }
    void fix_min_del_idx() { min_del_idx_ = curr_size_; }
    void set_min_del_idx(size_t idx) { min_del_idx_ = idx; }

    /*! the value array of the current column; for a bound vector the host mirror, refreshed now and written back before the next
     * device call (the pointer is not const, so it is taken to be written through) */
    el_type *values() const {
        DistVec *self = const_cast<DistVec *>(this);
        if (bound_) { self->refresh_col(curr_vec_idx_); self->val_dirty_[curr_vec_idx_] = true; }
        return (el_type *)values_[curr_vec_idx_];
    }
    uint8_t num_vecs() const { return (uint8_t)values_.rows(); }
    Matrix<uint8_t> &indices() { refresh_idx(); return indices_; }
    size_t curr_size() const { const_cast<DistVec *>(this)->refresh_info(); return curr_size_; }
    size_t adder_size() const { return adder_->size(); }
    size_t max_size() const { return max_size_; }
    int n_nonz() const { const_cast<DistVec *>(this)->refresh_info(); return n_nonz_; }
    uint64_t tot_sgn_coh() const { return nonini_occ_add; }     // not counted on the device

    void add_vecs(uint8_t idx1, uint8_t idx2) { add_vecs(idx1, idx2, 1); }
    void add_vecs(uint8_t idx1, uint8_t idx2, el_type c) {
        if (bound_) {
            if constexpr (std::is_same<el_type, double>::value) {
                before_device_op();
                fries_hip::ck(fries_vec_add_vecs(ctx_, idx1, idx2, c));
                after_device_op(idx1 == 0, idx1 == 1, false);
                return;
            }
        }
        for (size_t i = 0; i < curr_size_; i++) values_(idx1, i) += values_(idx2, i) * c;
    }
    void copy_vec(uint8_t src, uint8_t dst) {
        refresh_col(src); refresh_info();
        for (size_t i = 0; i < curr_size_; i++) values_(dst, i) = values_(src, i);
        if (bound_) { val_fresh_[dst] = true; val_dirty_[dst] = true; mirror_n_[dst] = curr_size_; }
    }
    void weight_vec(uint8_t idx1, uint8_t idx2, double expo) {
        refresh_col(idx1); refresh_col(idx2); refresh_info();
        for (size_t i = 0; i < curr_size_; i++) values_(idx1, i) *= pow(1 + fabs(values_(idx2, i)), expo);
        if (bound_) val_dirty_[idx1] = true;
    }
    void zero_vec() {
        std::fill(values_[curr_vec_idx_], values_[curr_vec_idx_] + values_.cols(), (el_type)0);
        if (bound_) {
            fries_hip::ck(fries_vec_column_zero(ctx_, curr_vec_idx_));
            refresh_info();
            val_fresh_[curr_vec_idx_] = true; val_dirty_[curr_vec_idx_] = false; mirror_n_[curr_vec_idx_] = curr_size_;
        }
    }
    void set_curr_vec_idx(uint8_t new_idx) {
        if (new_idx < values_.rows()) curr_vec_idx_ = new_idx;
        else {
            std::stringstream error;
            error << "Argument to set_curr_vec_idx (" << (unsigned int)new_idx << ") is out of bounds";
            throw std::runtime_error(error.str());
        }
    }
    uint8_t curr_vec_idx() const { return curr_vec_idx_; }

    /* the received adds, in buffer order (vec_utils.hpp:606-641); on the host */
    void add_elements(uint8_t *indices, el_type *vals, size_t count, size_t origin) {
        const uint8_t add_n_bytes = (uint8_t)CEILING(n_bits_ + 1, 8);
        const uint8_t vec_n_bytes = (uint8_t)indices_.cols();
        uint8_t tmp_occ[256];
        for (size_t el = 0; el < count; el++) {
            uint8_t *new_idx = &indices[el * add_n_bytes];
            int ini_flag = read_bit(new_idx, n_bits_);
            if (ini_flag) zero_bit(new_idx, n_bits_);
            uintmax_t hash_val = idx_to_hash(new_idx, tmp_occ);
            ssize_t *idx_ptr = vec_hash_.read(new_idx, hash_val, ini_flag);
            if (idx_ptr && *idx_ptr == -1) {
                *idx_ptr = pop_stack();
                if (*idx_ptr == -1) {
                    if (curr_size_ >= max_size_) expand();
                    *idx_ptr = (ssize_t)curr_size_;
                    curr_size_++;
                }
                memcpy(indices_[*idx_ptr], new_idx, vec_n_bytes);
                initialize_at_pos((size_t)*idx_ptr, tmp_occ);
                n_nonz_++;
            }
            if (idx_ptr) {
                bool nonz = values_(origin, *idx_ptr) != 0;
                bool should_add = ini_flag || nonz;
                nonini_occ_add += !ini_flag && nonz;
                if (should_add) values_(curr_vec_idx_, *idx_ptr) += vals[el];
                vals[el] = 0;
            }
        }
    }
    /* the same on the device */
    void add_elements_device(uint8_t *indices, el_type *vals, size_t count, size_t origin) {
        if constexpr (std::is_same<el_type, double>::value) {
            if (origin != 0) throw std::runtime_error("perform_add(origin) on the device judges the initiator rule against column 0 only");
            if (curr_vec_idx_ > 1) throw std::runtime_error("column must be 0 or 1");
            const uint8_t add_n_bytes = (uint8_t)CEILING(n_bits_ + 1, 8);
            std::vector<uint64_t> w(count ? count : 1);
            std::vector<uint8_t> ini(count ? count : 1);
            const uint64_t mask = n_bits_ >= 64 ? ~0ull : ((1ull << n_bits_) - 1ull);
            for (size_t el = 0; el < count; el++) {
                const uint8_t *src = &indices[el * add_n_bytes];
                ini[el] = (uint8_t)read_bit(src, n_bits_);
                w[el] = fries_word_of(src, add_n_bytes) & mask;
            }
            before_device_op();
            fries_hip::ck(fries_vec_add_to(ctx_, curr_vec_idx_, w.data(), vals, ini.data(), count));
            after_device_op(curr_vec_idx_ == 0, curr_vec_idx_ == 1, true);
            // a new position starts at zero in both columns: the other column's mirror only has to grow
            const uint8_t other = (uint8_t)(1 - curr_vec_idx_);
            if (val_fresh_[other]) {
                refresh_info();
                for (size_t i = mirror_n_[other]; i < curr_size_; i++) values_(other, i) = 0;
                // positions re-used from the free stack held zero in both columns already (they were deleted because of that)
                mirror_n_[other] = curr_size_;
            }
        }
    }

    el_type *operator[](size_t pos) { return values() + pos; }
    el_type *operator()(size_t vec_idx, size_t pos) {
        if (bound_) { refresh_col((uint8_t)vec_idx); val_dirty_[vec_idx] = true; }
        return &values_(vec_idx, pos);
    }
    uint8_t *orbs_at_pos(size_t pos) {
        if (bound_) { refresh_idx(); find_bits(indices_[pos], occ_orbs_[pos], (uint8_t)indices_.cols()); }
        return occ_orbs_[pos];
    }
    /*! for a bound vector the rows are filled by orbs_at_pos() on demand; apply_HBPP_sys of this build does not read them */
    Matrix<uint8_t> &occ_orbs() { return occ_orbs_; }
    /* vec_utils.hpp:672-677 */
    double matr_el_at_pos(size_t pos) {
        if (bound_) {
            if (!diag_fresh_) {
                size_t n = 0;
                fries_hip::ck(fries_vec_diag_download(ctx_, matr_el_.data(), matr_el_.size(), &n));
                diag_fresh_ = true;
            }
            return matr_el_[pos];
        }
        if (std::isnan(matr_el_[pos])) matr_el_[pos] = diag_calc_(occ_orbs_[pos]);
        return matr_el_[pos];
    }
    double local_norm() const {
        DistVec *self = const_cast<DistVec *>(this);
        self->refresh_col(curr_vec_idx_); self->refresh_info();
        double norm = 0;
        for (size_t i = 0; i < curr_size_; i++) norm += fabs((double)values_(curr_vec_idx_, i));
        return norm;
    }
    double two_norm() const {
        DistVec *self = const_cast<DistVec *>(this);
        self->refresh_col(curr_vec_idx_); self->refresh_info();
        double norm = 0;
        for (size_t i = 0; i < curr_size_; i++) norm += (double)values_(curr_vec_idx_, i) * (double)values_(curr_vec_idx_, i);
        return norm;
    }

    /* dets<rank>.dat, vals<rank>.dat (n_vecs columns one after the other) and dense.txt (vec_utils.hpp:713-750) */
    void save(const std::string &path, uint8_t start_idx, uint8_t n_vecs) {
        int my_rank = 0;
        MPI_Comm_rank(MPI_COMM_WORLD, &my_rank);
        refresh_idx(); refresh_info();
        for (uint8_t v = 0; v < n_vecs; v++) refresh_col((uint8_t)(start_idx + v));
        std::stringstream buffer;
        buffer << path << "dets" << my_rank << ".dat";
        std::ofstream file_p(buffer.str(), std::ios::binary);
        file_p.write((const char *)indices_.data(), curr_size_ * indices_.cols());
        file_p.close();
        buffer.str("");
        buffer << path << "vals" << my_rank << ".dat";
        file_p.open(buffer.str(), std::ios::binary);
        for (uint8_t v = 0; v < n_vecs; v++) file_p.write((const char *)values_[start_idx + v], sizeof(el_type) * curr_size_);
        file_p.close();
        if (my_rank == 0) {
            buffer.str("");
            buffer << path << "dense.txt";
            file_p.open(buffer.str());
            file_p << n_dense_ << '\n';
        }
    }
    void save(const std::string &path) { save(path, 0, (uint8_t)values_.rows()); }

    /* vec_utils.hpp:761-850: entries with |value| <= 1e-9 in every column are dropped, the rest compacted to the front */
    size_t load(const std::string &path, uint8_t n_vecs) {
        if (bound_) throw std::runtime_error("load() must run before the vector moves to the device");
        if (curr_vec_idx_ + n_vecs > values_.rows()) throw std::runtime_error("Number of vectors requested in load function will put you over the capacity of this DistVec object");
        int my_rank = 0;
        MPI_Comm_rank(MPI_COMM_WORLD, &my_rank);
        std::stringstream buffer;
        int dense_sizes[1] = {0};
        buffer << path << "dense.txt";
        read_csv(dense_sizes, buffer.str());
        n_dense_ = (size_t)dense_sizes[0];
        if (n_dense_) throw std::runtime_error("a dense (semi-stochastic) prefix is not supported by this build");
        const size_t n_bytes = indices_.cols();
        buffer.str("");
        buffer << path << "dets" << my_rank << ".dat";
        std::ifstream file_p(buffer.str(), std::ios::binary | std::ios::ate);
        if (!file_p.is_open()) throw std::runtime_error("Could not open saved binary vector file at path " + buffer.str());
        const size_t n_dets = (size_t)file_p.tellg() / n_bytes;
        while (n_dets > max_size_) expand();
        file_p.seekg(0, std::ios::beg);
        file_p.read((char *)indices_.data(), n_dets * n_bytes);
        file_p.close();
        buffer.str("");
        buffer << path << "vals" << my_rank << ".dat";
        file_p.open(buffer.str(), std::ios::binary);
        if (!file_p.is_open()) throw std::runtime_error("Could not open saved binary vector file at path " + buffer.str());
        for (uint8_t v = 0; v < n_vecs; v++) file_p.read((char *)values_[curr_vec_idx_ + v], sizeof(el_type) * n_dets);
        file_p.close();
        n_nonz_ = 0;
        uint8_t tmp_orbs[256];
        std::vector<el_type> tmp_vals(n_vecs);
        for (size_t det_idx = 0; det_idx < n_dets; det_idx++) {
            bool is_nonz = det_idx < min_del_idx_;
            for (uint8_t v = 0; v < n_vecs && !is_nonz; v++) if (fabs((double)values_(curr_vec_idx_ + v, det_idx)) > 1e-9) is_nonz = true;
            if (!is_nonz) continue;
            uint8_t *new_idx = indices_[det_idx];
            uintmax_t hash_val = idx_to_hash(new_idx, tmp_orbs);
            ssize_t *idx_ptr = vec_hash_.read(new_idx, hash_val, true);
            *idx_ptr = n_nonz_;
            memmove(indices_[n_nonz_], new_idx, n_bytes);
            for (uint8_t v = 0; v < n_vecs; v++) tmp_vals[v] = values_(curr_vec_idx_ + v, det_idx);
            initialize_at_pos((size_t)n_nonz_, tmp_orbs);
            for (uint8_t v = 0; v < n_vecs; v++) values_(curr_vec_idx_ + v, n_nonz_) = tmp_vals[v];
            n_nonz_++;
        }
        curr_size_ = (size_t)n_nonz_;
        return n_dense_;
    }
    size_t load(const std::string &path) { return load(path, (uint8_t)values_.rows()); }

    size_t init_dense(const std::string & /*read_path*/, const std::string & /*save_dir*/) {
        throw std::runtime_error("init_dense: the semi-stochastic dense subspace is not supported by this build");
    }
    el_type dense_norm() { return 0; }
    /* one rank: every element is here already (vec_utils.hpp:920-952) */
    void collect_procs() {}

    Adder<el_type> *adder() { return adder_; }
    friend class Adder<el_type>;
};

/* vec_utils.hpp:990-1019 with one rank: the buffer goes to the vector in the order of the add() calls */
template <class el_type>
void Adder<el_type>::perform_add(DistVec<el_type> *parent_vec, size_t origin) {
    if (parent_vec->bound()) { if (count_) parent_vec->add_elements_device(idx_.data(), vals_.data(), (size_t)count_, origin); }
    else parent_vec->add_elements(idx_.data(), vals_.data(), (size_t)count_, origin);
    count_ = 0;
}
#endif /* vec_utils_h */
