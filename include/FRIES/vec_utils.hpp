/*! \file  FRIES/vec_utils.hpp for the MI355X build: Adder<el_type> and DistVec<el_type> with the reference's public members
 * (FRIES/vec_utils.hpp:50-118, 120-952), so that a driver written against the reference compiles unchanged.
 *
 * A DistVec starts on the host (hash of determinant -> position, LIFO of released positions, buffered adds under the initiator rule) --
 * that is what trial vectors and H * trial use.  The SOLUTION vector becomes device-bound: a DistVec<double> with two value columns at
 * the first apply_HBPP_sys that receives its indices() matrix, a DistVec<int> (fciqmc_mol's walker numbers, exact integers in the
 * device's doubles) at its first perform_add.  From then on the determinants, the value columns, the diagonal elements, the hash table and
 * the free stack live in HBM (fries_amd/csrc/vec.hip) and every member below is a call through include/fries_hip.h:
 *
 *   perform_add              MPI_Alltoallv of the buffered adds (as in the reference), then fries_vec_add_to on what this rank received
 *                            (annihilating merge, the reference's positions and order of additions)
 *   add_vecs / zero_vec      fries_vec_add_vecs / fries_vec_column_zero
 *   dot                      fries_vec_dot_list      (sum in list order)
 *   values() / operator[]    host mirror of the column, downloaded when stale, uploaded before the next device call
 *   indices()                host mirror of the determinants, downloaded when stale
 *   matr_el_at_pos           fries_vec_diag_download (diagonal elements evaluated on the device)
 *   curr_size / n_nonz       fries_vec_info
 *   init_dense / load        the dense (semi-stochastic) prefix is declared to the device when the vector is bound (fries_vec_set_dense)
 *   save                     the reference's dets<rank>.dat / vals<rank>.dat / dense.txt
 *
 * Not carried over to a bound vector: del_at_pos of a position that would really be released (sys_comp already performs the drivers'
 * deletes on the device; fciqmc_mol's call never releases, fciqmc_mol.cpp:400-403), expand (the device capacity is the max_size given at
 * construction), writes through indices().  Ranks are the program's MPI ranks (FRIES/backend.hpp). */
#ifndef vec_utils_h
#define vec_utils_h
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <sstream>
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

/*! buffered adds, one buffer per destination rank (vec_utils.hpp:50-118) */
template <class el_type>
class Adder {
    int n_dest_;
    size_t cap_;                        // elements per destination
    uint8_t stride_;                    // bytes per buffered index: the index bits and one flag bit above them
    std::vector<uint8_t> out_idx_, in_idx_;
    std::vector<el_type> out_val_, in_val_;
    std::vector<int> out_n_, in_n_;
public:
    Adder(size_t size, int n_procs, uint8_t n_bits) : n_dest_(n_procs < 1 ? 1 : n_procs), cap_(size), stride_((uint8_t)CEILING(n_bits + 1, 8)),
        out_idx_((size_t)n_dest_ * size * stride_), in_idx_((size_t)n_dest_ * size * stride_), out_val_((size_t)n_dest_ * size), in_val_((size_t)n_dest_ * size),
        out_n_(n_dest_, 0), in_n_(n_dest_, 0) {}
    /* ships every buffer to its rank and hands what arrives to the vector, source rank by source rank (vec_utils.hpp:991-1019) */
    void perform_add(DistVec<el_type> *parent_vec, size_t origin);
    /* false once the destination's buffer is full (vec_utils.hpp:956-971) */
    bool add(uint8_t *idx, uint8_t idx_bits, el_type val, int proc_idx, uint8_t ini_flag) {
        int &n = out_n_[proc_idx];
        if ((size_t)n >= cap_) throw std::runtime_error("Too many elements added to Adder - must call perform_add() more frequently.");
        uint8_t *slot = &out_idx_[((size_t)proc_idx * cap_ + (size_t)n) * stride_];
        memset(slot, 0, stride_);
        memcpy(slot, idx, CEILING(idx_bits, 8));
        if (ini_flag) set_bit(slot, idx_bits);
        out_val_[(size_t)proc_idx * cap_ + (size_t)n] = val;
        return (size_t)++n < cap_;
    }
    size_t size() const { return cap_; }
    int pending() const { int s = 0; for (int c : out_n_) s += c; return s; }
    uint8_t n_bytes() const { return stride_; }
};

template <class el_type>
class DistVec : public fries_hip::DeviceVecBase {
    Matrix<el_type> vals_;              // one row per value column
    uint8_t col_;                       // curr_vec_idx
    size_t n_dense_;
    std::vector<size_t> holes_;         // released positions, last released first out
    int n_stored_;                      // n_nonz: positions in use
    Adder<el_type> *adder_;
    size_t del_floor_;                  // positions below it are never released (min_del_idx)
    // device binding
    bool bound_ = false;
    fries_ctx *ctx_ = nullptr;
    bool col_fresh_[2] = {false, false}, col_dirty_[2] = {false, false};
    size_t col_len_[2] = {0, 0};        // positions each value mirror covered when it was downloaded
    bool idx_fresh_ = false, diag_fresh_ = false, info_fresh_ = false;
    std::vector<uint32_t> proc_rns_;
    std::vector<double> xfer_;          // staging of a column in the device's doubles (DistVec<int>)
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
        for (uint8_t c = 0; c < vals_.rows(); c++) vals_(c, pos) = 0;
        active_pos_[pos] = true;
        matr_el_[pos] = NAN;
        memcpy(occ_orbs_[pos], orbs, occ_orbs_.cols());
    }
    /* (MI355X build) what a subclass has to re-derive for positions [0, n) after the determinant mirror was refreshed from the device */
    virtual void mirror_refreshed(size_t /*n*/) {}
    /* (MI355X build) the set-up call a subclass's model needs on the context instead of fries_frisys_setup */
    virtual void device_setup(fries_ctx *cx, uint32_t mat_nonz, bool new_hb) {
        fries_hip::Backend &B = fries_hip::Backend::get();
        if (proc_rns_.size() != 2 * (size_t)B.n_orb) throw std::runtime_error("rns_common must hold 2 * n_orb numbers");
        fries_hip::ck(fries_set_proc_scrambler(cx, proc_rns_.data(), proc_rns_.size()));
        // diag_fxn is "diag_matrel(occ) - hf_en" in every driver: recover hf_en from the HF determinant (exact when it is within a
        // factor two of the HF energy, Sterbenz) and verify the device's diagonal elements against diag_fxn after the load
        if (diag_calc_) {
            std::vector<uint8_t> hf_occ(occ_orbs_.cols());
            for (unsigned k = 0; k < B.n_elec / 2; k++) { hf_occ[k] = (uint8_t)k; hf_occ[k + B.n_elec / 2] = (uint8_t)(k + B.n_orb); }
            const double lam = diag_calc_(hf_occ.data());
            if (lam != 0) fries_hip::ck(fries_set_ham_shift(cx, fries_hf_energy(cx) - lam));
        }
        fries_frisys_params p{};
        p.max_dets = (uint32_t)max_size_; p.vec_nonz = (uint32_t)max_size_; p.mat_nonz = mat_nonz; p.hb_unnorm = new_hb ? 1 : 0;
        fries_hip::ck(fries_frisys_setup(cx, &p));
    }
    uint64_t word_at(size_t pos) const { return fries_word_of(indices_[pos], (uint8_t)indices_.cols()); }
    void pull_info() {
        if (!bound_ || info_fresh_) return;
        uint32_t cs = 0, nf = 0; int32_t nn = 0;
        fries_hip::ck(fries_vec_info(ctx_, &cs, &nn, &nf));
        curr_size_ = cs; n_stored_ = nn;
        info_fresh_ = true;
    }
    void pull_col(uint8_t c) {
        if (!bound_ || col_fresh_[c]) return;
        pull_info();
        size_t n = 0;
        if constexpr (std::is_same<el_type, double>::value) fries_hip::ck(fries_vec_column_download(ctx_, c, vals_[c], max_size_, &n));
        else {
            xfer_.resize(max_size_);
            fries_hip::ck(fries_vec_column_download(ctx_, c, xfer_.data(), max_size_, &n));
            for (size_t i = 0; i < n; i++) vals_(c, i) = (el_type)xfer_[i];
        }
        fries_hip::Backend::get().bytes_to_host += 8 * n;
        col_len_[c] = n;
        col_fresh_[c] = true;
    }
    void pull_idx() {
        if (!bound_ || idx_fresh_) return;
        pull_info();
        std::vector<uint64_t> w(curr_size_ ? curr_size_ : 1);
        size_t n = 0;
        fries_hip::ck(fries_vec_download(ctx_, w.data(), nullptr, w.size(), &n));
        fries_hip::Backend::get().bytes_to_host += 8 * n;
        const size_t nb = indices_.cols();
        for (size_t i = 0; i < n; i++) memcpy(indices_[i], &w[i], nb);
        idx_fresh_ = true;
        mirror_refreshed(n);
    }
public:
    DistVec(size_t size, Adder<el_type> *adder, uint8_t n_bits, unsigned int n_elec,
            std::function<double(const uint8_t *)> diag_fxn, uint8_t n_vecs,
            std::vector<uint32_t> rns_common, std::vector<uint32_t> rns_distinct) :
    vals_(n_vecs, size), col_(0), n_dense_(0), n_stored_(0), adder_(adder), del_floor_(0), proc_rns_(rns_common),
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
        const char *q = (const char *)p, *b = (const char *)vals_.data();
        return q >= b && q < b + sizeof(el_type) * vals_.rows() * vals_.cols();
    }
    size_t offset_of(const void *p) const override { return (size_t)((const el_type *)p - vals_.data()) % vals_.cols(); }
    const void *indices_key() const override { return &indices_; }
    bool bound() const override { return bound_; }
    size_t dense_size() const override { return n_dense_; }
    fries_ctx *ctx() override { return ctx_; }
    /* moves the vector to the device: positions, the value columns and the rank hash are kept; mat_nonz sizes the operators' work arrays */
    void bind(uint32_t mat_nonz, bool new_hb) override {
        if (bound_) return;
        fries_hip::Backend &B = fries_hip::Backend::get();
        if (vals_.rows() > 2) throw std::runtime_error("the device vector has at most two value columns");
        if (B.ctx_taken) throw std::runtime_error("one device-bound solution vector per process");
        if (adder_->pending()) throw std::runtime_error("perform_add() must run before the vector moves to the device");
        for (size_t i = 0; i < curr_size_; i++) {
            if (!active_pos_[i]) throw std::runtime_error("a vector with released positions cannot be moved to the device (bind it before deleting)");
            if (vals_.rows() == 2 && vals_(1, i) != 0) throw std::runtime_error("column 1 must be zero when the vector moves to the device");
        }
        fries_ctx *cx = B.ctx();
        B.attach_comm(mat_nonz);
        device_setup(cx, mat_nonz, new_hb);
        std::vector<uint64_t> w(curr_size_ ? curr_size_ : 1);
        xfer_.assign(curr_size_ ? curr_size_ : 1, 0.0);
        for (size_t i = 0; i < curr_size_; i++) { w[i] = word_at(i); xfer_[i] = (double)vals_(0, i); }
        fries_hip::ck(fries_vec_load(cx, w.data(), xfer_.data(), curr_size_));
        fries_hip::Backend::get().bytes_to_device += 16 * curr_size_;
        const int n_procs = fries_hip::mpi_size();
        if (n_dense_ || n_procs > 1) {       // every rank declares its share of the dense space, also an empty one (collective)
            const int my_rank = fries_hip::mpi_rank();
            const int tot = sum_mpi((int)n_dense_, my_rank, n_procs);
            if (tot) fries_hip::ck(fries_vec_set_dense(cx, (uint32_t)n_dense_));
        }
        ctx_ = cx; bound_ = true; B.ctx_taken = true;
        for (int c = 0; c < 2; c++) { col_fresh_[c] = true; col_dirty_[c] = false; col_len_[c] = curr_size_; }
        idx_fresh_ = true; diag_fresh_ = false; info_fresh_ = false;
        if (diag_calc_ && B.have_mol && curr_size_) {
            const size_t n_chk = curr_size_ < 8 ? curr_size_ : 8;
            for (size_t i = 0; i < n_chk; i++)
                if (diag_calc_(occ_orbs_[i]) != matr_el_at_pos(i)) throw std::runtime_error("the diagonal elements of diag_fxn are not diag_matrel(occ) - const: this build cannot evaluate them on the device");
        }
    }
    void before_device_op() override {
        if (!bound_) return;
        for (int c = 0; c < (int)vals_.rows(); c++) if (col_dirty_[c]) {
            if constexpr (std::is_same<el_type, double>::value) fries_hip::ck(fries_vec_column_upload(ctx_, c, vals_[c], col_len_[c]));
            else {
                xfer_.resize(col_len_[c] ? col_len_[c] : 1);
                for (size_t i = 0; i < col_len_[c]; i++) xfer_[i] = (double)vals_(c, i);
                fries_hip::ck(fries_vec_column_upload(ctx_, c, xfer_.data(), col_len_[c]));
            }
            col_dirty_[c] = false;
            fries_hip::Backend::get().bytes_to_device += 8 * col_len_[c];
        }
    }
    void after_device_op(bool col0, bool col1, bool layout) override {
        if (col0) col_fresh_[0] = false;
        if (col1) col_fresh_[1] = false;
        if (layout) { idx_fresh_ = false; diag_fresh_ = false; info_fresh_ = false; }
    }

    // ---------------------------------------------------------------- the reference's members
    uint8_t n_bits() { return n_bits_; }
    virtual uint8_t gen_orb_list(uint8_t *det, uint8_t *occ_orbs) { return find_bits(det, occ_orbs, (uint8_t)indices_.cols()); }

    /* sum over the listed elements of vals2[i] * this[idx2[i]], in list order (vec_utils.hpp:228-252) */
    double dot(Matrix<uint8_t> &idx2, double *vals2, size_t num2, std::vector<uintmax_t> & /*hashes2*/) { return dot(idx2, vals2, num2); }
    double dot(Matrix<uint8_t> &idx2, double *vals2, size_t num2) {
        if (bound_) {
            before_device_op();
            std::vector<uint64_t> w(num2 ? num2 : 1);
            for (size_t i = 0; i < num2; i++) w[i] = fries_word_of(idx2[i], (uint8_t)idx2.cols());
            double r = 0;
            fries_hip::ck(fries_vec_dot_list(ctx_, col_, w.data(), vals2, num2, &r));
            fries_hip::Backend::get().bytes_to_device += 16 * num2;
            return r;
        }
        double acc = 0;
        for (size_t i = 0; i < num2; i++) if (ssize_t *at = vec_hash_.read(idx2[i], 0, false)) acc += vals2[i] * vals_(col_, *at);
        return acc;
    }
    /* vec_utils.hpp:324-340 */
    double internal_dot(uint8_t idx1, uint8_t idx2) {
        if (idx1 >= vals_.rows() || idx2 >= vals_.rows()) throw std::runtime_error("Error: argument to internal_dot exceeds bounds of value matrix");
        pull_col(idx1); pull_col(idx2); pull_info();
        double acc = 0;
        for (size_t i = 0; i < curr_size_; i++) acc += vals_(idx1, i) * vals_(idx2, i);
        return acc;
    }

    virtual void expand() {
        if (bound_) throw std::runtime_error("the device vector cannot grow beyond the max_size it was constructed with");
        const size_t grown = 2 * max_size_;
        std::cout << "Increasing storage capacity in vector to " << grown << "\n";
        indices_.reshape(grown, indices_.cols());
        occ_orbs_.reshape(grown, occ_orbs_.cols());
        vals_.enlarge_cols(grown, (int)curr_size_);
        matr_el_.resize(grown);
        active_pos_.resize(grown);
        max_size_ = grown;
    }

    virtual int idx_to_proc(uint8_t *idx) {
        uint8_t orbs[256];
        gen_orb_list(idx, orbs);
        return idx_to_proc(idx, orbs);
    }
    virtual int idx_to_proc(uint8_t * /*idx*/, uint8_t *orbs) {
        const int n_procs = fries_hip::mpi_size();
        return (int)(proc_hash_.hash_fxn(orbs, (uint8_t)occ_orbs_.cols(), NULL, 0) % (uintmax_t)n_procs);
    }
    virtual uintmax_t idx_to_hash(uint8_t *idx, uint8_t *orbs) {
        const unsigned int want = (unsigned int)occ_orbs_.cols();
        if (gen_orb_list(idx, orbs) != want) {
            char txt[2 * 8 + 1];
            print_str(idx, (uint8_t)indices_.cols(), txt);
            throw std::runtime_error(std::string("Determinant ") + txt + " created with an incorrect number of electrons");
        }
        return vec_hash_.hash_fxn(orbs, (uint8_t)want, NULL, 0);
    }
    void print_ht() { vec_hash_.print_ht(); }

    /* vec_utils.hpp:418-436: a zero value is not buffered; the element goes to the buffer of the rank that owns idx */
    bool add(uint8_t *idx, el_type val, uint8_t ini_flag) { return val != 0 ? adder_->add(idx, n_bits_, val, idx_to_proc(idx), ini_flag) : true; }
    bool add(uint8_t *idx, uint8_t *orbs, el_type val, uint8_t ini_flag) { return adder_->add(idx, n_bits_, val, idx_to_proc(idx, orbs), ini_flag); }
    void perform_add(size_t origin) { adder_->perform_add(this, origin); }

    ssize_t pop_stack() {
        if (holes_.empty()) return -1;
        const size_t p = holes_.back();
        holes_.pop_back();
        return (ssize_t)p;
    }
    bool zero_everywhere(size_t pos) const { for (uint8_t c = 0; c < vals_.rows(); c++) if (vals_(c, pos) != 0) return false; return true; }
    /* releases pos when it holds zero in every column and lies at or beyond the delete floor (vec_utils.hpp:458-476) */
    void del_at_pos(size_t pos) {
        if (bound_) {
            for (uint8_t c = 0; c < vals_.rows(); c++) pull_col(c);
            if (!zero_everywhere(pos) || pos < del_floor_) return;      // the reference would not release it either
            throw std::runtime_error("del_at_pos on a device-bound vector: sys_comp() of this build already releases the zeroed elements on the device");
        }
        if (!active_pos_[pos] || pos < del_floor_ || !zero_everywhere(pos)) return;
        vec_hash_.del_entry(indices_[pos], 0);
        holes_.push_back(pos);
        active_pos_[pos] = false;
        n_stored_--;
    }
    void cleanup() { for (size_t pos = del_floor_; pos < curr_size_; pos++) if (zero_everywhere(pos)) del_at_pos(pos); }
    void fix_min_del_idx() { del_floor_ = curr_size_; }
    void set_min_del_idx(size_t idx) { del_floor_ = idx; }

    /*! the value array of the current column; for a bound vector the host mirror, refreshed now and written back before the next
     * device call (the pointer is not const, so it is taken to be written through) */
    el_type *values() const {
        DistVec *self = const_cast<DistVec *>(this);
        if (bound_) { self->pull_col(col_); self->col_dirty_[col_] = true; }
        return (el_type *)vals_[col_];
    }
    uint8_t num_vecs() const { return (uint8_t)vals_.rows(); }
    Matrix<uint8_t> &indices() { pull_idx(); return indices_; }
    size_t curr_size() const { const_cast<DistVec *>(this)->pull_info(); return curr_size_; }
    size_t adder_size() const { return adder_->size(); }
    size_t max_size() const { return max_size_; }
    int n_nonz() const { const_cast<DistVec *>(this)->pull_info(); return n_stored_; }
    uint64_t tot_sgn_coh() const { return nonini_occ_add; }     // not counted on the device

    void add_vecs(uint8_t idx1, uint8_t idx2) { add_vecs(idx1, idx2, 1); }
    void add_vecs(uint8_t idx1, uint8_t idx2, el_type c) {
        if (bound_) {
            before_device_op();
            fries_hip::ck(fries_vec_add_vecs(ctx_, idx1, idx2, (double)c));
            after_device_op(idx1 == 0, idx1 == 1, false);
            return;
        }
        for (size_t i = 0; i < curr_size_; i++) vals_(idx1, i) += c * vals_(idx2, i);
    }
    void copy_vec(uint8_t src, uint8_t dst) {
        pull_col(src); pull_info();
        std::copy(vals_[src], vals_[src] + curr_size_, vals_[dst]);
        if (bound_) { col_fresh_[dst] = true; col_dirty_[dst] = true; col_len_[dst] = curr_size_; }
    }
    void weight_vec(uint8_t idx1, uint8_t idx2, double expo) {
        pull_col(idx1); pull_col(idx2); pull_info();
        for (size_t i = 0; i < curr_size_; i++) vals_(idx1, i) *= pow(1 + fabs((double)vals_(idx2, i)), expo);
        if (bound_) col_dirty_[idx1] = true;
    }
    void zero_vec() {
        std::fill(vals_[col_], vals_[col_] + vals_.cols(), (el_type)0);
        if (bound_) {
            fries_hip::ck(fries_vec_column_zero(ctx_, col_));
            pull_info();
            col_fresh_[col_] = true; col_dirty_[col_] = false; col_len_[col_] = curr_size_;
        }
    }
    void set_curr_vec_idx(uint8_t new_idx) {
        if (new_idx >= vals_.rows()) throw std::runtime_error("Argument to set_curr_vec_idx (" + std::to_string((unsigned)new_idx) + ") is out of bounds");
        col_ = new_idx;
    }
    uint8_t curr_vec_idx() const { return col_; }

    /*! what a rank received, in arrival order (vec_utils.hpp:606-641): an element whose index is not stored yet takes a position only when it
     * carries the initiator flag (a released position first, else the end); it is added when flagged or when the origin column is non-zero there */
    void add_elements(uint8_t *indices, el_type *vals, size_t count, size_t origin) {
        const uint8_t in_stride = (uint8_t)CEILING(n_bits_ + 1, 8);
        uint8_t occ[256];
        for (size_t k = 0; k < count; k++) {
            uint8_t *key = indices + k * in_stride;
            const bool flagged = read_bit(key, n_bits_) != 0;
            zero_bit(key, n_bits_);
            const uintmax_t hv = idx_to_hash(key, occ);
            ssize_t *where = vec_hash_.read(key, hv, flagged);
            if (!where) { vals[k] = 0; continue; }                      // unknown index, no initiator behind it: dropped (counts nowhere)
            if (*where < 0) {
                ssize_t p = pop_stack();
                if (p < 0) { if (curr_size_ >= max_size_) expand(); p = (ssize_t)curr_size_++; }
                *where = p;
                memcpy(indices_[p], key, indices_.cols());
                initialize_at_pos((size_t)p, occ);
                n_stored_++;
            }
            const bool occupied = vals_(origin, *where) != 0;
            if (!flagged && occupied) nonini_occ_add++;
            if (flagged || occupied) vals_(col_, *where) += vals[k];
            vals[k] = 0;
        }
    }
    /* the same on the device */
    void add_elements_device(uint8_t *indices, el_type *vals, size_t count, size_t origin) {
        if (origin != 0) throw std::runtime_error("perform_add(origin) on the device judges the initiator rule against column 0 only");
        if (col_ > 1) throw std::runtime_error("column must be 0 or 1");
        const uint8_t in_stride = (uint8_t)CEILING(n_bits_ + 1, 8);
        std::vector<uint64_t> w(count ? count : 1);
        std::vector<uint8_t> ini(count ? count : 1);
        std::vector<double> incoming(count ? count : 1);      // (not xfer_: before_device_op() stages the column mirrors there)
        const uint64_t mask = n_bits_ >= 64 ? ~0ull : ((1ull << n_bits_) - 1ull);
        for (size_t k = 0; k < count; k++) {
            const uint8_t *src = indices + k * in_stride;
            ini[k] = (uint8_t)read_bit(src, n_bits_);
            w[k] = fries_word_of(src, in_stride) & mask;
            incoming[k] = (double)vals[k];
        }
        before_device_op();
        fries_hip::ck(fries_vec_add_to(ctx_, col_, w.data(), incoming.data(), ini.data(), count));
        fries_hip::Backend::get().bytes_to_device += 17 * count;
        after_device_op(col_ == 0, col_ == 1, true);
        if (vals_.rows() == 2) {
            // a new position starts at zero in both columns: the other column's mirror only has to grow (positions re-used from the free
            // stack held zero in both columns already: they were released because of that)
            const uint8_t other = (uint8_t)(1 - col_);
            if (col_fresh_[other]) {
                pull_info();
                for (size_t i = col_len_[other]; i < curr_size_; i++) vals_(other, i) = 0;
                col_len_[other] = curr_size_;
            }
        }
    }
    /* (MI355X build) DistVec<int> is the solution vector of fciqmc_mol: it moves to the device when its first adds are performed */
    bool binds_on_add() const { return !bound_ && std::is_same<el_type, int>::value && vals_.rows() == 1 && (bool)diag_calc_ && fries_hip::Backend::get().have_mol && !fries_hip::Backend::get().ctx_taken; }

    el_type *operator[](size_t pos) { return values() + pos; }
    el_type *operator()(size_t vec_idx, size_t pos) {
        if (bound_) { pull_col((uint8_t)vec_idx); col_dirty_[vec_idx] = true; }
        return &vals_(vec_idx, pos);
    }
    uint8_t *orbs_at_pos(size_t pos) {
        if (bound_) { pull_idx(); gen_orb_list(indices_[pos], occ_orbs_[pos]); }
        return occ_orbs_[pos];
    }
    /*! for a bound vector the rows are filled by orbs_at_pos() on demand; apply_HBPP_sys of this build does not read them */
    Matrix<uint8_t> &occ_orbs() { return occ_orbs_; }
    /* vec_utils.hpp:672-677 */
    double matr_el_at_pos(size_t pos) {
        if (bound_ && fries_hip::Backend::get().have_mol) {
            if (!diag_fresh_) {
                size_t n = 0;
                fries_hip::ck(fries_vec_diag_download(ctx_, matr_el_.data(), matr_el_.size(), &n));
                fries_hip::Backend::get().bytes_to_host += 8 * n;
                diag_fresh_ = true;
            }
            return matr_el_[pos];
        }
        if (std::isnan(matr_el_[pos])) matr_el_[pos] = diag_calc_(occ_orbs_[pos]);
        return matr_el_[pos];
    }
    double local_norm() const {
        DistVec *self = const_cast<DistVec *>(this);
        self->pull_col(col_); self->pull_info();
        double s = 0;
        for (size_t i = 0; i < curr_size_; i++) s += fabs((double)vals_(col_, i));
        return s;
    }
    double two_norm() const {
        DistVec *self = const_cast<DistVec *>(this);
        self->pull_col(col_); self->pull_info();
        double s = 0;
        for (size_t i = 0; i < curr_size_; i++) { const double x = (double)vals_(col_, i); s += x * x; }
        return s;
    }

    /* dets<rank>.dat (raw index bytes), vals<rank>.dat (the n_vecs columns one after the other), dense.txt: every rank's dense-space size
     * on one line, written by rank 0 (vec_utils.hpp:713-750) */
    void save(const std::string &path, uint8_t start_idx, uint8_t n_vecs) {
        const int my_rank = fries_hip::mpi_rank(), n_procs = fries_hip::mpi_size();
        pull_idx(); pull_info();
        for (uint8_t v = 0; v < n_vecs; v++) pull_col((uint8_t)(start_idx + v));
        const std::string tag = std::to_string(my_rank) + ".dat";
        {
            std::ofstream out(path + "dets" + tag, std::ios::binary);
            out.write((const char *)indices_.data(), (std::streamsize)(curr_size_ * indices_.cols()));
        }
        {
            std::ofstream out(path + "vals" + tag, std::ios::binary);
            for (uint8_t v = 0; v < n_vecs; v++) out.write((const char *)vals_[start_idx + v], (std::streamsize)(sizeof(el_type) * curr_size_));
        }
        std::vector<int> per_rank((size_t)n_procs, 0);
        int mine = (int)n_dense_;
        MPI_Gather(&mine, 1, MPI_INT, per_rank.data(), 1, MPI_INT, 0, MPI_COMM_WORLD);
        if (my_rank == 0) {
            std::ofstream out(path + "dense.txt");
            for (int p = 0; p < n_procs; p++) out << per_rank[p] << (p + 1 < n_procs ? "," : "\n");
        }
    }
    void save(const std::string &path) { save(path, 0, (uint8_t)vals_.rows()); }

    /* vec_utils.hpp:761-850: this rank's files; the first n_dense entries stay whatever their values, of the others those with a magnitude
     * above 1e-9 in some column; the survivors move to the front in file order and are hashed again.  Returns the dense-space size. */
    size_t load(const std::string &path, uint8_t n_vecs) {
        if (bound_) throw std::runtime_error("load() must run before the vector moves to the device");
        if (col_ + n_vecs > vals_.rows()) throw std::runtime_error("Number of vectors requested in load function will put you over the capacity of this DistVec object");
        const int my_rank = fries_hip::mpi_rank(), n_procs = fries_hip::mpi_size();
        std::vector<int> per_rank((size_t)n_procs + 64, 0);
        if (my_rank == 0) read_csv(per_rank.data(), path + "dense.txt");
        int mine = 0;
        MPI_Scatter(per_rank.data(), 1, MPI_INT, &mine, 1, MPI_INT, 0, MPI_COMM_WORLD);
        n_dense_ = (size_t)mine;
        const std::string tag = std::to_string(my_rank) + ".dat";
        const size_t width = indices_.cols();
        std::ifstream fd(path + "dets" + tag, std::ios::binary | std::ios::ate);
        if (!fd.is_open()) throw std::runtime_error("Could not open saved binary vector file at path " + path + "dets" + tag);
        const size_t n_file = (size_t)fd.tellg() / width;
        while (n_file > max_size_) expand();
        fd.seekg(0, std::ios::beg);
        fd.read((char *)indices_.data(), (std::streamsize)(n_file * width));
        std::ifstream fv(path + "vals" + tag, std::ios::binary);
        if (!fv.is_open()) throw std::runtime_error("Could not open saved binary vector file at path " + path + "vals" + tag);
        for (uint8_t v = 0; v < n_vecs; v++) fv.read((char *)vals_[col_ + v], (std::streamsize)(sizeof(el_type) * n_file));
        size_t kept = 0;
        uint8_t occ[256];
        std::vector<el_type> row(n_vecs);
        for (size_t src = 0; src < n_file; src++) {
            bool keep = src < n_dense_ || src < del_floor_;
            for (uint8_t v = 0; v < n_vecs && !keep; v++) keep = fabs((double)vals_(col_ + v, src)) > 1e-9;
            if (!keep) continue;
            for (uint8_t v = 0; v < n_vecs; v++) row[v] = vals_(col_ + v, src);
            uint8_t *key = indices_[src];
            const uintmax_t hv = idx_to_hash(key, occ);
            *vec_hash_.read(key, hv, true) = (ssize_t)kept;
            memmove(indices_[kept], key, width);
            initialize_at_pos(kept, occ);
            for (uint8_t v = 0; v < n_vecs; v++) vals_(col_ + v, kept) = row[v];
            kept++;
        }
        n_stored_ = (int)kept;
        curr_size_ = kept;
        return n_dense_;
    }
    size_t load(const std::string &path) { return load(path, (uint8_t)vals_.rows()); }

    /* the dense (semi-stochastic) space (vec_utils.hpp:858-897): rank 0 reads the determinants (integers, one per line) and adds each with
     * value 1 and the initiator flag; after the adds have travelled, what this rank stores is its dense space -- values zeroed, positions kept for
     * good.  dense.txt in save_dir lists every rank's size. */
    size_t init_dense(const std::string &read_path, const std::string &save_dir) {
        if (bound_) throw std::runtime_error("init_dense() must run before the vector moves to the device");
        const int my_rank = fries_hip::mpi_rank(), n_procs = fries_hip::mpi_size();
        const size_t n_read = my_rank == 0 ? read_dets(read_path, indices_) : 0;
        for (size_t i = 0; i < n_read; i++) add(indices_[i], (el_type)1, 1);
        perform_add(0);
        n_dense_ = curr_size_;
        for (uint8_t c = 0; c < vals_.rows(); c++) std::fill(vals_[c], vals_[c] + n_dense_, (el_type)0);
        std::vector<int> per_rank((size_t)n_procs, 0);
        per_rank[my_rank] = (int)n_dense_;
        MPI_Allgather(MPI_IN_PLACE, 0, MPI_INT, per_rank.data(), 1, MPI_INT, MPI_COMM_WORLD);
        if (my_rank == 0) {
            std::ofstream out(save_dir + "dense.txt");
            if (!out.is_open()) throw std::runtime_error("Could not load deterministic subspace from file at path " + save_dir + "dense.txt");
            for (int p = 0; p < n_procs; p++) out << per_rank[p] << ",";
            out << "\n";
        }
        return n_dense_;
    }
    /* one-norm of the dense space over all ranks (vec_utils.hpp:903-918) */
    el_type dense_norm() {
        pull_col(col_);
        el_type s = 0;
        for (size_t i = 0; i < n_dense_; i++) { const el_type x = vals_(col_, i); s += x >= 0 ? x : -x; }
        const int my_rank = fries_hip::mpi_rank(), n_procs = fries_hip::mpi_size();
        return sum_mpi(s, my_rank, n_procs);
    }
    /* every rank ends up with every rank's elements, appended in rank order (vec_utils.hpp:920-952); small host vectors only */
    void collect_procs() {
        const int my_rank = fries_hip::mpi_rank(), n_procs = fries_hip::mpi_size();
        if (n_procs == 1) return;
        if (bound_) throw std::runtime_error("collect_procs() is for host vectors (trial vectors)");
        std::vector<int> cnt((size_t)n_procs, 0), off((size_t)n_procs, 0);
        cnt[my_rank] = (int)curr_size_;
        MPI_Allgather(MPI_IN_PLACE, 0, MPI_INT, cnt.data(), 1, MPI_INT, MPI_COMM_WORLD);
        int tot = 0;
        for (int p = 0; p < n_procs; p++) { off[p] = tot; tot += cnt[p]; }
        while ((size_t)tot > max_size_) expand();
        const int width = (int)indices_.cols();
        std::vector<uint8_t> all_idx((size_t)tot * width);
        std::vector<int> bc((size_t)n_procs), bo((size_t)n_procs);
        for (int p = 0; p < n_procs; p++) { bc[p] = cnt[p] * width; bo[p] = off[p] * width; }
        {
            std::vector<uint8_t> mine(indices_.data(), indices_.data() + curr_size_ * width);
            fries_allgatherv(mine.data(), (int)mine.size(), all_idx.data(), bc.data(), bo.data(), 1);
        }
        std::vector<el_type> all_val((size_t)tot);
        for (uint8_t c = 0; c < vals_.rows(); c++) {
            std::vector<el_type> mine(vals_[c], vals_[c] + curr_size_);
            for (int p = 0; p < n_procs; p++) { bc[p] = cnt[p] * (int)sizeof(el_type); bo[p] = off[p] * (int)sizeof(el_type); }
            fries_allgatherv(mine.data(), (int)(mine.size() * sizeof(el_type)), all_val.data(), bc.data(), bo.data(), 1);
            std::copy(all_val.begin(), all_val.end(), vals_[c]);
        }
        memcpy(indices_.data(), all_idx.data(), all_idx.size());
        curr_size_ = (size_t)tot;
    }

    Adder<el_type> *adder() { return adder_; }
    friend class Adder<el_type>;
private:
    /* MPI_Allgatherv through MPI_Alltoallv (the compat header does not carry every collective): every rank sends its block to everybody */
    static void fries_allgatherv(const void *send, int send_bytes, void *recv, const int *recv_bytes, const int *recv_off, int /*unit*/) {
        const int n_procs = fries_hip::mpi_size();
        std::vector<int> sc((size_t)n_procs, send_bytes), sd((size_t)n_procs, 0);
        MPI_Alltoallv((void *)send, sc.data(), sd.data(), MPI_BYTE, recv, (int *)recv_bytes, (int *)recv_off, MPI_BYTE, MPI_COMM_WORLD);
    }
};

template <class el_type>
void Adder<el_type>::perform_add(DistVec<el_type> *parent_vec, size_t origin) {
    MPI_Alltoall(out_n_.data(), 1, MPI_INT, in_n_.data(), 1, MPI_INT, MPI_COMM_WORLD);
    std::vector<int> sc((size_t)n_dest_), rc((size_t)n_dest_), disp((size_t)n_dest_);
    for (int p = 0; p < n_dest_; p++) { sc[p] = out_n_[p] * stride_; rc[p] = in_n_[p] * stride_; disp[p] = (int)((size_t)p * cap_ * stride_); }
    MPI_Alltoallv(out_idx_.data(), sc.data(), disp.data(), MPI_BYTE, in_idx_.data(), rc.data(), disp.data(), MPI_BYTE, MPI_COMM_WORLD);
    for (int p = 0; p < n_dest_; p++) { sc[p] = out_n_[p] * (int)sizeof(el_type); rc[p] = in_n_[p] * (int)sizeof(el_type); disp[p] = (int)((size_t)p * cap_ * sizeof(el_type)); }
    MPI_Alltoallv(out_val_.data(), sc.data(), disp.data(), MPI_BYTE, in_val_.data(), rc.data(), disp.data(), MPI_BYTE, MPI_COMM_WORLD);
    if (parent_vec->bound()) {
        // one merge for everything that arrived, source rank by source rank: the order the reference's add_elements calls see
        size_t tot = 0;
        for (int p = 0; p < n_dest_; p++) tot += (size_t)in_n_[p];
        std::vector<uint8_t> idx(tot * stride_ + 1);
        std::vector<el_type> val(tot + 1);
        size_t o = 0;
        for (int p = 0; p < n_dest_; p++) {
            memcpy(&idx[o * stride_], &in_idx_[(size_t)p * cap_ * stride_], (size_t)in_n_[p] * stride_);
            std::copy(&in_val_[(size_t)p * cap_], &in_val_[(size_t)p * cap_] + in_n_[p], &val[o]);
            o += (size_t)in_n_[p];
        }
        if (tot) parent_vec->add_elements_device(idx.data(), val.data(), tot, origin);
    }
    else for (int p = 0; p < n_dest_; p++) parent_vec->add_elements(&in_idx_[(size_t)p * cap_ * stride_], &in_val_[(size_t)p * cap_], (size_t)in_n_[p], origin);
    std::fill(out_n_.begin(), out_n_.end(), 0);
    if (parent_vec->binds_on_add()) {
        // fciqmc_mol: the walker vector moves to the device once its first adds (the starting vector) are in; from the next perform_add on
        // the merge runs there.  One iteration's spawn list is at most n_dest_ * cap_ received elements.
        const size_t cap = (size_t)n_dest_ * cap_;
        parent_vec->bind((uint32_t)(cap > 4096 ? cap : 4096), false);
    }
}
#endif /* vec_utils_h */
