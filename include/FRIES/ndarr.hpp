/*! \file  FRIES/ndarr.hpp for the MI355X build: the reference's dense containers with the same names and members
 * (FRIES/ndarr.hpp:15-357) -- Matrix<T> (row-major, operator()(r, c), operator[](r) -> row pointer, reshape that keeps the leading
 * elements), the bit-packed Matrix<bool>, FourDArr, and SymmERIs (8-fold packed two-electron integrals: tri(tri(i,j), tri(k,l))).
 * Host-side only; the device keeps its own copies (fries_set_molecule). */
#ifndef ndarr_h
#define ndarr_h
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <FRIES/math_utils.h>

template <class mat_type>
class Matrix {
    size_t rows_, cols_, tot_size_;
    std::vector<mat_type> data_;
public:
    Matrix(size_t rows, size_t cols) : rows_(rows), cols_(cols), tot_size_(rows * cols), data_(rows * cols) {}
    mat_type &operator()(size_t row, size_t col) { return data_[cols_ * row + col]; }
    mat_type operator()(size_t row, size_t col) const { return data_[cols_ * row + col]; }
    void zero() { std::fill(data_.begin(), data_.end(), 0); }
    mat_type *operator[](size_t row) { return &data_[cols_ * row]; }
    const mat_type *operator[](size_t row) const { return &data_[cols_ * row]; }
    /* more columns; the first n_keep[row] (or n_keep) elements of every row stay where they belong */
    void enlarge_cols(size_t new_col, int *n_keep) {
        if (new_col <= cols_) return;
        std::vector<mat_type> nd(rows_ * new_col);
        for (size_t r = 0; r < rows_; r++) for (int c = 0; c < n_keep[r]; c++) nd[r * new_col + c] = data_[r * cols_ + c];
        data_.swap(nd); cols_ = new_col; tot_size_ = rows_ * new_col;
    }
    void enlarge_cols(size_t new_col, int n_keep) {
        if (new_col <= cols_) return;
        std::vector<mat_type> nd(rows_ * new_col);
        for (size_t r = 0; r < rows_; r++) for (int c = 0; c < n_keep; c++) nd[r * new_col + c] = data_[r * cols_ + c];
        data_.swap(nd); cols_ = new_col; tot_size_ = rows_ * new_col;
    }
    /* new shape; storage only grows, the flat prefix of the data is kept */
    void reshape(size_t new_rows, size_t new_cols) {
        size_t new_size = new_rows * new_cols;
        if (new_size > tot_size_) { tot_size_ = new_size; data_.resize(tot_size_); }
        rows_ = new_rows; cols_ = new_cols;
    }
    size_t rows() const { return rows_; }
    size_t cols() const { return cols_; }
    mat_type *data() const { return (mat_type *)data_.data(); }
    void copy_from(Matrix<mat_type> &mat) { std::copy(mat.data_.begin(), mat.data_.begin() + std::min(mat.tot_size_, tot_size_), data_.begin()); }
};

class FourDArr {
    size_t len1_, len2_, len3_, len4_;
    double *data_;
public:
    FourDArr(size_t len1, size_t len2, size_t len3, size_t len4) : len1_(len1), len2_(len2), len3_(len3), len4_(len4) {
        data_ = (double *)malloc(sizeof(double) * len1 * len2 * len3 * len4);
    }
    FourDArr(const FourDArr &) = delete;
    FourDArr &operator=(const FourDArr &) = delete;
    double &operator()(size_t i1, size_t i2, size_t i3, size_t i4) { return data_[i1 * len2_ * len3_ * len4_ + i2 * len3_ * len4_ + i3 * len4_ + i4]; }
    double operator()(size_t i1, size_t i2, size_t i3, size_t i4) const { return data_[i1 * len2_ * len3_ * len4_ + i2 * len3_ * len4_ + i3 * len4_ + i4]; }
    ~FourDArr() { free(data_); }
    double *data() { return data_; }
};

class SymmERIs {
    double *data_;
    size_t len_;
public:
    SymmERIs(size_t len) {
        size_t n_pair = len * (len + 1) / 2;
        len_ = len;
        size_t vec_size = n_pair * (n_pair + 1) / 2;
        data_ = (double *)malloc(vec_size * sizeof(double));
        std::fill(data_, data_ + vec_size, 0);
    }
    SymmERIs(const SymmERIs &) = delete;
    SymmERIs &operator=(const SymmERIs &) = delete;
    double chemist(size_t i1, size_t i2, size_t i3, size_t i4) const {
        size_t min1 = i1 < i2 ? i1 : i2, max1 = i1 < i2 ? i2 : i1;
        size_t p1_idx = I_J_TO_TRI_WDIAG(min1, max1);
        size_t min2 = i3 < i4 ? i3 : i4, max2 = i3 < i4 ? i4 : i3;
        size_t p2_idx = I_J_TO_TRI_WDIAG(min2, max2);
        size_t min_p = p1_idx < p2_idx ? p1_idx : p2_idx, max_p = p1_idx < p2_idx ? p2_idx : p1_idx;
        return data_[I_J_TO_TRI_WDIAG(min_p, max_p)];
    }
    double &chemist_ordered(size_t i1, size_t i2, size_t i3, size_t i4) {
        size_t p1_idx = I_J_TO_TRI_WDIAG(i1, i2), p2_idx = I_J_TO_TRI_WDIAG(i3, i4);
        return data_[I_J_TO_TRI_WDIAG(p1_idx, p2_idx)];
    }
    double physicist(size_t i1, size_t i2, size_t i3, size_t i4) const { return chemist(i1, i3, i2, i4); }
    ~SymmERIs() { free(data_); }
    /* MI355X build: the packed array as the device takes it (fries_set_molecule) */
    const double *packed() const { return data_; }
    size_t n_orb() const { return len_; }
};

template <> class Matrix<bool> {
    size_t rows_, cols_, cols_coarse_, tot_size_;
    std::vector<uint8_t> data_;
public:
    class BoolReference {
        uint8_t *value_; uint8_t mask_;
    public:
        BoolReference(uint8_t &value, uint8_t nbit) : value_(&value), mask_(uint8_t(0x1) << nbit) {}
        BoolReference &operator=(bool b) noexcept { if (b) *value_ |= mask_; else *value_ &= (uint8_t)~mask_; return *this; }
        operator bool() const noexcept { return *value_ & mask_; }
    };
    Matrix(size_t rows, size_t cols) : rows_(rows), cols_(cols), cols_coarse_(CEILING(cols, 8)), tot_size_(rows * CEILING(cols, 8)), data_(rows * CEILING(cols, 8), 0) {}
    BoolReference operator()(size_t row, size_t col) { return BoolReference(data_[cols_coarse_ * row + col / 8], col % 8); }
    void reshape(size_t new_rows, size_t new_cols) {
        size_t new_coarse = CEILING(new_cols, 8), new_size = new_rows * new_coarse;
        if (new_size > tot_size_) { tot_size_ = new_size; data_.resize(tot_size_, 0); }
        rows_ = new_rows; cols_ = new_cols; cols_coarse_ = new_coarse;
    }
    size_t cols() const { return cols_; }
    size_t rows() const { return rows_; }
    class RowReference {
        size_t row_idx_; Matrix<bool> *mat_; Matrix<uint8_t> *other_mat_;
    public:
        RowReference(Matrix<bool> &mat, size_t row) : row_idx_(row), mat_(&mat), other_mat_(nullptr) {}
        RowReference(Matrix<uint8_t> *mat, size_t row) : row_idx_(row), mat_(nullptr), other_mat_(mat) {}
        BoolReference operator[](size_t idx) {
            if (mat_) return BoolReference(mat_->row_ptr(row_idx_)[idx / 8], idx % 8);
            return BoolReference((*other_mat_)(row_idx_, idx / 8), idx % 8);
        }
    };
    RowReference operator[](size_t row) { return RowReference(*this, row); }
    uint8_t *row_ptr(size_t row) const { return (uint8_t *)&data_[cols_coarse_ * row]; }
    uint8_t *data() const { return (uint8_t *)data_.data(); }
};
#endif /* ndarr_h */
