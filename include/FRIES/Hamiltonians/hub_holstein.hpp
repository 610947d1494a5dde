/*! \file  FRIES/Hamiltonians/hub_holstein.hpp for the MI355X build: the Hubbard-Holstein helpers of the reference's drivers, same names and
 * signatures (FRIES/Hamiltonians/hub_holstein.hpp:23-186, hub_holstein.cpp:10-171).  A basis state is the bit string
 * [alpha sites | beta sites | ph_bits bits per site] (1-D chain, open ends).  The neighbour lists a hop is chosen from are
 * [count, orbitals that can hop right ..., (n_elec + 1:) count, orbitals that can hop left ...], as HubHolVec::find_neighbors_1D fills them.
 * calc_ref_ovlp of the solution vector runs on the device (csrc/hh.hip: k_hh_ref_ovlp, the reference's byte-wise neighbour test included). */
#ifndef hub_holstein_h
#define hub_holstein_h
#include <cstdint>
#include <cstring>
#include <random>
#include <stdexcept>
#include <FRIES/fci_utils.h>
#include <FRIES/math_utils.h>
#include <FRIES/det_store.h>
#include <FRIES/ndarr.hpp>
#include <FRIES/backend.hpp>

namespace fries_hip {
inline uint64_t hh_word(const uint8_t *det, unsigned n_bits) { uint64_t w = 0; memcpy(&w, det, (n_bits + 7) / 8 > 8 ? 8 : (n_bits + 7) / 8); return n_bits >= 64 ? w : w & ((1ull << n_bits) - 1ull); }
/* the k-th hop of a neighbour list: (from, to) */
inline void hh_hop(unsigned int k, unsigned int n_elec, const uint8_t *neighbors, uint8_t *orbs) {
    const unsigned int n_right = neighbors[0], n_left = neighbors[n_elec + 1];
    if (k < n_right) { orbs[0] = neighbors[1 + k]; orbs[1] = (uint8_t)(orbs[0] + 1); }
    else if (k < n_right + n_left) { orbs[0] = neighbors[n_elec + 2 + (k - n_right)]; orbs[1] = (uint8_t)(orbs[0] - 1); }
    else throw std::runtime_error("Excitation index selected for a Hubbard determinant exceeds the possible number of excitations from that determinant");
}
}

/* num_sampl hops drawn uniformly from the state's possible hops (hub_holstein.cpp:10-19) */
inline void hub_multin(unsigned int n_elec, const uint8_t *neighbors, unsigned int num_sampl, std::mt19937 &mt_obj, uint8_t chosen_orbs[][2]) {
    for (unsigned int s = 0; s < num_sampl; s++) {
        const unsigned int n_hops = neighbors[0] + neighbors[n_elec + 1];
        fries_hip::hh_hop((unsigned int)(mt_obj() / (1. + UINT32_MAX) * n_hops), n_elec, neighbors, chosen_orbs[s]);
    }
}
/* every possible hop: the right-hops, then the left-hops (hub_holstein.cpp:83-98) */
inline size_t hub_all(unsigned int n_elec, uint8_t *neighbors, uint8_t chosen_orbs[][2]) {
    const size_t n_hops = (size_t)neighbors[0] + neighbors[n_elec + 1];
    for (size_t k = 0; k < n_hops; k++) fries_hip::hh_hop((unsigned int)k, n_elec, neighbors, chosen_orbs[k]);
    return n_hops;
}
/* number of doubly occupied sites (hub_holstein.cpp:100-136) */
inline unsigned int hub_diag(uint8_t *det, unsigned int n_sites) {
    const uint64_t w = fries_hip::hh_word(det, 2 * n_sites);
    return (unsigned int)__builtin_popcountll(w & (w >> n_sites) & ((1ull << n_sites) - 1ull));
}
/* the Neel state: spin-up electrons on sites 0, 2, 4, ..., spin-down on 1, 3, 5, ..., no phonons (hub_holstein.cpp:139-171) */
inline void gen_neel_det_1D(unsigned int n_sites, unsigned int n_elec, uint8_t ph_bits, uint8_t *det) {
    const size_t n_bytes = CEILING((2 + ph_bits) * n_sites, 8);
    memset(det, 0, n_bytes);
    for (unsigned int k = 0; k < n_elec / 2; k++) { set_bit(det, (uint8_t)(2 * k)); set_bit(det, (uint8_t)(n_sites + 2 * k + 1)); }
}
/* sum over the stored states connected to the reference (one hop, or one phonon on an occupied site of it) of value x coupling
 * (hub_holstein.hpp:93-186).  `dets` must be the indices() of the device-bound HubHolVec: the sum is formed on the device. */
template <typename T>
double calc_ref_ovlp(Matrix<uint8_t> &dets, T * /*vals*/, Matrix<uint8_t> & /*phonons*/, size_t /*n_dets*/, uint8_t * /*ref_det*/, uint8_t * /*occ_ref*/,
                     uint8_t /*n_elec*/, unsigned int /*n_sites*/, double /*g_over_t*/) {
    fries_hip::DeviceVecBase *v = fries_hip::Backend::get().by_indices(&dets);
    if (!v || !v->bound()) throw std::runtime_error("calc_ref_ovlp: dets must be the indices() matrix of the device-bound HubHolVec (this build has no host implementation)");
    v->before_device_op();
    double r = 0;
    fries_hip::ck(fries_hh_ref_ovlp(v->ctx(), &r));
    return r;
}
#endif /* hub_holstein_h */
