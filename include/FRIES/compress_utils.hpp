/*! \file  FRIES/compress_utils.hpp for the MI355X build: the compression entry points the drivers call, with the reference's
 * signatures (FRIES/compress_utils.hpp:28, 52, 72, 170-231, 354, 374, 392-428; compress_utils.cpp:19-27, 29-105, 283-327, 684-693, 797-877).
 * find_preserve and sys_comp take the raw value pointer DistVec::values() handed out, as in the reference; the pointer is mapped back to
 * its (device-bound) vector and the work is done there -- the host scratch arguments (srt_idx, loc_norms) are accepted and left alone,
 * keep_idx stays all-false because fries_sys_comp also performs the deletes the driver would do from it (frisys_mol.cpp:534-539).
 * sum_mpi is the reference's: MPI_Allgather, then the sum in rank order.  The small samplers (round_binomially, the alias method) are
 * host functions on the caller's std::mt19937, draw for draw the reference's. */
#ifndef compress_utils_h
#define compress_utils_h
#include <cmath>
#include <cstdint>
#include <limits>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>
#include <mpi.h>
#include <FRIES/ndarr.hpp>
#include <FRIES/backend.hpp>

/* every rank's value, added up in rank order on every rank (compress_utils.hpp:170-231) */
template <class T> inline T fries_rank_sum(T local, int my_rank, int n_procs, MPI_Datatype ty, MPI_Comm comm) {
    std::vector<T> all((size_t)(n_procs > 0 ? n_procs : 1));
    all[(size_t)my_rank] = local;
    MPI_Allgather(MPI_IN_PLACE, 0, ty, all.data(), 1, ty, comm);
    T total = 0;
    for (int p = 0; p < n_procs; p++) total += all[(size_t)p];
    return total;
}
inline double sum_mpi(double local, int my_rank, int n_procs, MPI_Comm comm) { return fries_rank_sum(local, my_rank, n_procs, MPI_DOUBLE, comm); }
inline double sum_mpi(double local, int my_rank, int n_procs) { return fries_rank_sum(local, my_rank, n_procs, MPI_DOUBLE, MPI_COMM_WORLD); }
inline int sum_mpi(int local, int my_rank, int n_procs) { return fries_rank_sum(local, my_rank, n_procs, MPI_INT, MPI_COMM_WORLD); }
inline uint64_t sum_mpi(uint64_t local, int my_rank, int n_procs) { return fries_rank_sum(local, my_rank, n_procs, MPI_UINT64_T, MPI_COMM_WORLD); }

/* compress_utils.cpp:684-693 */
inline void adjust_shift(double *shift, double one_norm, double *last_norm, double target_norm, double damp_factor) {
    if (*last_norm) {
        *shift -= damp_factor * log(one_norm / *last_norm);
        *last_norm = one_norm;
    }
    if (*last_norm == 0 && one_norm > target_norm) *last_norm = one_norm;
}

/* n Bernoulli(p - floor(p)) trials on top of n * floor(p) (compress_utils.cpp:19-27): one draw per trial */
inline int round_binomially(double p, unsigned int n, std::mt19937 &mt_obj) {
    const int whole = (int)floor(p);
    const double frac = p - whole;
    int out = whole * (int)n;
    for (unsigned int t = 0; t < n; t++) if (mt_obj() / (1. + UINT32_MAX) < frac) out++;
    return out;
}

/* Walker's alias tables for n_states probabilities (compress_utils.cpp:823-853): small and big columns are paired from the back of two
 * stacks, a big one that drops below 1 becomes the small one on top */
inline void setup_alias(double *probs, unsigned int *aliases, double *alias_probs, size_t n_states) {
    std::vector<unsigned int> under, over;
    under.reserve(n_states); over.reserve(n_states);
    for (unsigned int i = 0; i < n_states; i++) {
        aliases[i] = i;
        alias_probs[i] = n_states * probs[i];
        (alias_probs[i] < 1 ? under : over).push_back(i);
    }
    while (!under.empty() && !over.empty()) {
        const unsigned int s = under.back(), b = over.back();
        aliases[s] = b;
        alias_probs[b] += alias_probs[s] - 1;
        if (alias_probs[b] < 1) { under.back() = b; over.pop_back(); }
        else under.pop_back();
    }
}
/* n_samp draws from the tables, two uniforms each, written samp_int bytes apart (compress_utils.cpp:856-877) */
inline void sample_alias(unsigned int *aliases, double *alias_probs, size_t n_states, uint8_t *samples, unsigned int n_samp, size_t samp_int, std::mt19937 &mt_obj) {
    if (n_states > std::numeric_limits<uint8_t>::max()) throw std::runtime_error("Number of states that can be sampled (" + std::to_string(n_states) + ") exceeds 255 in sample_alias");
    for (unsigned int k = 0; k < n_samp; k++) {
        const uint8_t col = (uint8_t)(mt_obj() / (1. + UINT32_MAX) * n_states);
        const bool stay = mt_obj() / (1. + UINT32_MAX) < alias_probs[col];
        samples[k * samp_int] = stay ? col : (uint8_t)aliases[col];
    }
}

namespace fries_hip {
/* the device-bound vector behind a value pointer; the pointer may skip the vector's dense (semi-stochastic) space, nothing else */
inline DeviceVecBase *vec_behind(const double *values, const char *who) {
    DeviceVecBase *v = Backend::get().owner_of(values);
    if (!v || !v->bound()) throw std::runtime_error(std::string(who) + ": the value pointer does not belong to a device-bound DistVec (this build has no host implementation of the compression)");
    const size_t off = v->offset_of(values);
    if (off != 0 && off != v->dense_size()) throw std::runtime_error(std::string(who) + ": the values must start at the vector's first position or right behind its dense space");
    return v;
}
}

/* compress_utils.cpp:29-105 on the device-bound vector that owns `values`; the preserved set and the local remaining norm stay on the
 * device, where sys_comp picks them up.  Returns 0: the drivers only store the return value in loc_norms, which sys_comp of this build ignores. */
inline double find_preserve(double *values, std::vector<size_t> & /*srt_idx*/, std::vector<bool> & /*keep_idx*/, size_t /*count*/, unsigned int *n_samp, double *global_norm) {
    fries_hip::DeviceVecBase *v = fries_hip::vec_behind(values, "find_preserve");
    v->before_device_op();
    uint32_t ns = *n_samp;
    double gn = 0;
    fries_hip::ck(fries_find_preserve(v->ctx(), &ns, &gn));
    v->after_device_op(false, false, false);
    *n_samp = ns;
    *global_norm = gn;
    return 0;
}
/* compress_utils.cpp:283-327 + the del_at_pos loop of the drivers */
inline void sys_comp(double *vec_vals, size_t /*vec_len*/, double * /*loc_norms*/, unsigned int n_samp, std::vector<bool> & /*keep_exact*/, double rand_num) {
    fries_hip::DeviceVecBase *v = fries_hip::vec_behind(vec_vals, "sys_comp");
    MPI_Bcast(&rand_num, 1, MPI_DOUBLE, 0, MPI_COMM_WORLD);        // rank 0's uniform (compress_utils.cpp:291)
    v->before_device_op();
    fries_hip::ck(fries_sys_comp(v->ctx(), n_samp, rand_num));
    v->after_device_op(true, false, true);
}
/* comp_sub = find_keep_sub + sys_sub (compress_utils.cpp:130-276, 702-820) on the device.  The general routine takes any matrix of
 * sub-weights; what the engine runs are the two shapes frisys_hh's loop passes (frisys_hh.cpp:187-224), both on the Hubbard-Holstein
 * solution vector: (1) one element per stored state, every non-zero one with the same two sub-weights (the rows of sub_weights; the zero
 * ones marked n_div = 1) -- the magnitudes are those of the vector itself, which moves to the device here the first time; (2) the
 * emissions of (1), every one with a uniform subdivision n_div > 0 (hops of the state / 2 n_elec phonon moves), which the device
 * derives again from the states.  Other shapes are refused.  new_vals / new_idx as in the reference: value, (input element, sub-index). */
inline size_t comp_sub(double *values, size_t count, unsigned int *n_div, Matrix<double> &sub_weights, Matrix<bool> & /*keep_idx*/, uint16_t *sub_sizes,
                       unsigned int n_samp, double * /*wt_remain*/, double rand_num, double *new_vals, size_t new_idx[][2]) {
    fries_hip::Backend &B = fries_hip::Backend::get();
    if (!B.hh_mode || sub_sizes) throw std::runtime_error("comp_sub: this build compresses the two sub-weight shapes of frisys_hh on the device; general sub-weight matrices are not supported");
    static fries_hip::DeviceVecBase *hh = nullptr;
    static size_t stage1_len = 0;
    bool uniform = count > 0;
    for (size_t i = 0; i < count && uniform; i++) uniform = n_div[i] > 0;
    int stage;
    if (!uniform || !hh || count != stage1_len) {
        // shape (1): find the vector these magnitudes were taken from
        fries_hip::DeviceVecBase *v = B.bound_vec();
        if (!v) for (auto *c : B.vecs) if (c->hh_candidate()) { v = c; break; }
        if (!v) throw std::runtime_error("comp_sub: no HubHolVec to run on");
        if (sub_weights.cols() != 2) throw std::runtime_error("comp_sub: the element rows must hold two sub-weights (hop, phonon)");
        if (!v->bound()) { v->hh_budget(n_samp); v->bind(n_samp, true); }
        hh = v; stage = 1;
    }
    else stage = 2;
    MPI_Bcast(&rand_num, 1, MPI_DOUBLE, 0, MPI_COMM_WORLD);        // rank 0's uniform (compress_utils.cpp:806)
    hh->before_device_op();
    const size_t cap = sub_weights.rows();
    std::vector<uint32_t> i0(cap ? cap : 1), i1(cap ? cap : 1);
    size_t n_out = 0;
    fries_hip::ck(fries_hh_comp_sub(hh->ctx(), stage, n_samp, rand_num, i0.data(), i1.data(), new_vals, cap, &n_out));
    for (size_t k = 0; k < n_out; k++) { new_idx[k][0] = i0[k]; new_idx[k][1] = i1[k]; }
    if (stage == 1) stage1_len = n_out; else stage1_len = 0;
    (void)values;
    return n_out;
}
#endif /* compress_utils_h */
