/*! \file  FRIES/compress_utils.hpp for the MI355X build: the compression entry points the drivers call, with the reference's
 * signatures (FRIES/compress_utils.hpp:52, 72, 170-231; compress_utils.cpp:29-105, 283-327, 684-693).  find_preserve and sys_comp take
 * the raw value pointer DistVec::values() handed out, as in the reference; the pointer is mapped back to its (device-bound) vector and
 * the work is done there -- the host scratch arguments (srt_idx, loc_norms) are accepted and left alone, keep_idx stays all-false
 * because fries_sys_comp also performs the deletes the driver would do from it (frisys_mol.cpp:534-539). */
#ifndef compress_utils_h
#define compress_utils_h
#include <cmath>
#include <cstdint>
#include <random>
#include <stdexcept>
#include <vector>
#include <mpi.h>
#include <FRIES/ndarr.hpp>
#include <FRIES/backend.hpp>

/* sum over the ranks in rank order (compress_utils.hpp:170-231); one rank in this build's host shim */
inline double sum_mpi(double local, int /*my_rank*/, int n_procs) { if (n_procs != 1) throw std::runtime_error("the FRIES/*.hpp host surface of this build is one-rank; ranks go through fries_comm"); return local; }
inline int sum_mpi(int local, int /*my_rank*/, int n_procs) { if (n_procs != 1) throw std::runtime_error("the FRIES/*.hpp host surface of this build is one-rank; ranks go through fries_comm"); return local; }
inline uint64_t sum_mpi(uint64_t local, int /*my_rank*/, int n_procs) { if (n_procs != 1) throw std::runtime_error("the FRIES/*.hpp host surface of this build is one-rank; ranks go through fries_comm"); return local; }

/* compress_utils.cpp:684-693 */
inline void adjust_shift(double *shift, double one_norm, double *last_norm, double target_norm, double damp_factor) {
    if (*last_norm) {
        *shift -= damp_factor * log(one_norm / *last_norm);
        *last_norm = one_norm;
    }
    if (*last_norm == 0 && one_norm > target_norm) *last_norm = one_norm;
}

/* compress_utils.cpp:29-105 on the device-bound vector that owns `values`; returns the norm of the unpreserved part.  The elements
 * before `values` (the dense space) are not supported: values must be the start of the column. */
inline double find_preserve(double *values, std::vector<size_t> & /*srt_idx*/, std::vector<bool> & /*keep_idx*/, size_t count, unsigned int *n_samp, double *global_norm) {
    fries_hip::DeviceVecBase *v = fries_hip::Backend::get().owner_of(values);
    if (!v) throw std::runtime_error("find_preserve: the value pointer does not belong to a device-bound DistVec");
    if (v->offset_of(values) != 0) throw std::runtime_error("find_preserve: a dense (semi-stochastic) prefix is not supported by this build");
    (void)count;
    v->before_device_op();
    uint32_t ns = *n_samp;
    double gn = 0;
    fries_hip::ck(fries_find_preserve(v->ctx(), &ns, &gn));
    v->after_device_op(false, false, false);
    *n_samp = ns;
    *global_norm = gn;
    return 0;       // the local remaining norm stays on the device: sys_comp picks it up there
}
/* compress_utils.cpp:283-327 + the del_at_pos loop of the drivers */
inline void sys_comp(double *vec_vals, size_t /*vec_len*/, double * /*loc_norms*/, unsigned int n_samp, std::vector<bool> & /*keep_exact*/, double rand_num) {
    fries_hip::DeviceVecBase *v = fries_hip::Backend::get().owner_of(vec_vals);
    if (!v) throw std::runtime_error("sys_comp: the value pointer does not belong to a device-bound DistVec");
    v->before_device_op();
    fries_hip::ck(fries_sys_comp(v->ctx(), n_samp, rand_num));
    v->after_device_op(true, false, true);
}
#endif /* compress_utils_h */
