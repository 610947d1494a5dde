/*! \file  FRIES/Hamiltonians/near_uniform.hpp for the MI355X build: the near-uniform excitation generator with the reference's names and
 * signatures (FRIES/Hamiltonians/near_uniform.hpp:33-180; near_uniform.cpp:14-433).
 *
 * These are HOST functions on the caller's std::mt19937: the reference's fciqmc_mol / frimulti_mol drivers call them once per stored
 * determinant from their own loops, and the number of draws a call consumes depends on the values drawn (rejection loops), so they
 * reproduce the reference draw for draw -- a driver compiled against these headers with the reference's seed walks the reference's
 * trajectory.  The engine's own FCIQMC loop (fries_fciqmc_iterate, csrc/fciqmc.hip) runs the same samplers one attempt per lane on a
 * counter-based stream; that is the path to use for throughput, this is the path that makes the reference's driver source run. */
#ifndef near_uniform_h
#define near_uniform_h
#include <cmath>
#include <cstdint>
#include <random>
#include <FRIES/ndarr.hpp>
#include <FRIES/Hamiltonians/molecule.hpp>

struct orb_pair {
    uint8_t orb1;   ///< first spin orbital, 0 .. 2 n_orb - 1
    uint8_t orb2;   ///< second spin orbital
    uint8_t spin1;  ///< spin of the first (0 or 1)
    uint8_t spin2;  ///< spin of the second
};

namespace fries_hip {
inline double uni01(std::mt19937 &mt) { return mt() / (1. + UINT32_MAX); }
inline unsigned int pick_below(std::mt19937 &mt, unsigned int n) { return (unsigned int)(uni01(mt) * n); }     // uniform on [0, n)
inline bool det_bit(const uint8_t *det, unsigned int b) { return (det[b >> 3] >> (b & 7)) & 1; }
}

/* unoccupied orbitals per irrep and spin (near_uniform.cpp:14-28): every irrep starts with its orbital count, each electron takes one away */
inline void count_symm_virt(unsigned int counts[][2], uint8_t *occ_orbs, unsigned int n_elec, SymmInfo *symm) {
    const unsigned int n_orb = (unsigned int)symm->symm_vec.size(), half = n_elec / 2;
    for (unsigned int s = 0; s < n_irreps; s++) counts[s][0] = counts[s][1] = symm->symm_lookup(s, 0);
    for (unsigned int e = 0; e < n_elec; e++) counts[symm->symm_vec[occ_orbs[e] % n_orb]][e < half ? 0 : 1]--;
}
/* successes among n Bernoulli(p) draws (near_uniform.cpp:31-39) */
inline unsigned int bin_sample(unsigned int n, double p, std::mt19937 &mt_obj) {
    unsigned int hits = 0;
    for (unsigned int t = 0; t < n; t++) if (fries_hip::uni01(mt_obj) < p) hits++;
    return hits;
}

/* One double excitation: an electron pair from the triangular index of one draw, the number of allowed first virtuals, the first virtual
 * (by enumeration when at most three are allowed, by rejection otherwise), the second among the allowed orbitals of the complementary irrep
 * (near_uniform.cpp:46-191).  Null excitations (no allowed virtual) take one draw and emit nothing. */
inline unsigned int doub_multin(uint8_t *det, uint8_t *occ_orbs, unsigned int num_elec, SymmInfo *symm, unsigned int (*unocc_sym_counts)[2],
                                unsigned int num_sampl, std::mt19937 &mt_obj, uint8_t (*chosen_orbs)[4], double *prob_vec) {
    using fries_hip::pick_below; using fries_hip::det_bit;
    const unsigned int n_orb = (unsigned int)symm->symm_vec.size(), half = num_elec / 2;
    const std::vector<uint8_t> &irr = symm->symm_vec;
    unsigned int n_out = 0;
    for (unsigned int smp = 0; smp < num_sampl; smp++) {
        const unsigned int tri = pick_below(mt_obj, num_elec * (num_elec - 1) / 2);
        unsigned int hi = (unsigned int)((sqrt(tri * 8. + 1) - 1) / 2);
        const unsigned int lo = (unsigned int)(tri - hi * (hi + 1.) / 2);
        hi++;
        const unsigned int e_hi = occ_orbs[hi], e_lo = occ_orbs[lo], s_hi = hi / half, s_lo = lo / half;
        const unsigned int prod = irr[e_hi % n_orb] ^ irr[e_lo % n_orb];
        const bool par = s_hi == s_lo;
        const unsigned int self = (prod == 0 && par) ? 1u : 0u;         // a virtual cannot pair with itself
        unsigned int n_first = par ? n_orb - half : 2 * n_orb - num_elec;
        for (unsigned int s = 0; s < n_irreps; s++) {
            if (unocc_sym_counts[s ^ prod][s_lo] == self) n_first -= unocc_sym_counts[s][s_hi];
            if (!par && unocc_sym_counts[s ^ prod][s_hi] == self) n_first -= unocc_sym_counts[s][s_lo];
        }
        if (n_first == 0) continue;
        unsigned int u1;
        if (n_first <= 3) {
            int left = (int)pick_below(mt_obj, n_first);
            unsigned int sa = par ? s_hi : 0, sb = par ? s_hi : 1, orb = 0;
            for (; left >= 0 && orb < n_orb; orb++)
                if (!det_bit(det, orb + sa * n_orb) && unocc_sym_counts[prod ^ irr[orb]][sb] - (prod == 0 && sa == sb) != 0) left--;
            if (left >= 0) {
                sa = 1; sb = 0;
                for (; left >= 0 && orb < 2 * n_orb; orb++)
                    if (!det_bit(det, orb) && unocc_sym_counts[prod ^ irr[orb - n_orb]][sb] - (prod == 0 && sa == sb) != 0) left--;
                orb -= n_orb;
            }
            u1 = orb - 1 + sa * n_orb;
        }
        else {
            for (;;) {
                unsigned int cand, sa, sb;
                if (par) { sa = sb = s_hi; cand = pick_below(mt_obj, n_orb) + sa * n_orb; }
                else { cand = pick_below(mt_obj, 2 * n_orb); sa = cand / n_orb; sb = 1 - sa; }
                if (det_bit(det, cand)) continue;
                if (unocc_sym_counts[prod ^ irr[cand % n_orb]][sb] - (prod == 0 && sa == sb) != 0) { u1 = cand; break; }
            }
        }
        const unsigned int sa = u1 / n_orb, sb = s_hi ^ s_lo ^ sa;
        const unsigned int ia = irr[u1 % n_orb], ib = prod ^ ia;
        const unsigned int n_b_given_a = unocc_sym_counts[ib][sb] - (prod == 0 && sa == sb);
        int left = (int)pick_below(mt_obj, n_b_given_a);
        unsigned int u2 = 0;
        for (unsigned int k = 1; left >= 0; k++) {
            u2 = symm->symm_lookup(ib, k) + sb * n_orb;
            if (!det_bit(det, u2) && u2 != u1) left--;
        }
        const unsigned int n_a_given_b = unocc_sym_counts[ia][sa] - (prod == 0 && sa == sb);
        prob_vec[n_out] = 2. / num_elec / (num_elec - 1) / n_first * (1. / n_b_given_a + 1. / n_a_given_b);
        chosen_orbs[n_out][0] = (uint8_t)e_lo; chosen_orbs[n_out][1] = (uint8_t)e_hi;
        chosen_orbs[n_out][2] = (uint8_t)(u1 < u2 ? u1 : u2); chosen_orbs[n_out][3] = (uint8_t)(u1 < u2 ? u2 : u1);
        n_out++;
    }
    return n_out;
}

/* One single excitation: an electron that has allowed virtuals (rejection), then an unoccupied orbital of its irrep and spin (rejection)
 * (near_uniform.cpp:248-313) */
inline unsigned int sing_multin(uint8_t *det, uint8_t *occ_orbs, unsigned int num_elec, SymmInfo *symm, unsigned int (*unocc_sym_counts)[2],
                                unsigned int num_sampl, std::mt19937 &mt_obj, uint8_t (*chosen_orbs)[2], double *prob_vec) {
    using fries_hip::pick_below; using fries_hip::det_bit;
    const unsigned int n_orb = (unsigned int)symm->symm_vec.size(), half = num_elec / 2;
    std::vector<unsigned int> n_virt(num_elec);
    unsigned int stuck = 0;
    for (unsigned int e = 0; e < num_elec; e++) {
        n_virt[e] = unocc_sym_counts[symm->symm_vec[occ_orbs[e] % n_orb]][e / half];
        if (n_virt[e] == 0) stuck++;
    }
    if (stuck == num_elec) return 0;
    for (unsigned int smp = 0; smp < num_sampl; smp++) {
        unsigned int e;
        do e = pick_below(mt_obj, num_elec); while (n_virt[e] == 0);
        const unsigned int from = occ_orbs[e], ir = symm->symm_vec[from % n_orb], shift = (from / n_orb) * n_orb;
        unsigned int to;
        do to = shift + symm->symm_lookup(ir, 1 + pick_below(mt_obj, symm->symm_lookup(ir, 0))); while (det_bit(det, to));
        prob_vec[smp] = 1. / n_virt[e] / (num_elec - stuck);
        chosen_orbs[smp][0] = (uint8_t)from; chosen_orbs[smp][1] = (uint8_t)to;
    }
    return num_sampl;
}

/* electrons with at least one symmetry-allowed single excitation (near_uniform.cpp:316-327) */
inline unsigned int count_sing_allowed(uint8_t *occ_orbs, unsigned int num_elec, uint8_t *orb_symm, unsigned int num_orb, unsigned int (*unocc_sym_counts)[2]) {
    unsigned int n = 0;
    for (unsigned int e = 0; e < num_elec; e++) if (unocc_sym_counts[orb_symm[occ_orbs[e] % num_orb]][e / (num_elec / 2)] != 0) n++;
    return n;
}
/* the *occ_choice-th such electron: its index goes back through occ_choice, its number of allowed virtuals is returned (near_uniform.cpp:330-347) */
inline unsigned int count_sing_virt(uint8_t *occ_orbs, uint8_t num_elec, uint8_t *orb_symm, uint8_t num_orb, unsigned int (*unocc_sym_counts)[2], uint8_t *occ_choice) {
    unsigned int seen = 0;
    for (unsigned int e = 0; e < num_elec; e++) {
        const unsigned int nv = unocc_sym_counts[orb_symm[occ_orbs[e] % num_orb]][e / (num_elec / 2)];
        if (nv == 0) continue;
        if (seen == *occ_choice) { *occ_choice = (uint8_t)e; return nv; }
        seen++;
    }
    return 0;
}
/* For the electron pair with triangular index *occ_choice (near_uniform.cpp:348-416): occ_choice[0..1] receive its two orbitals (lower
 * electron first), virt_weights[k] the probability that the first virtual falls in the k-th representative irrep and virt_counts[k] the
 * number of virtual pairs there.  With parallel spins and a non-trivial product every unordered irrep pair {s, s ^ prod} is represented by
 * its member that has the product's highest bit set, in ascending order; otherwise all eight irreps stand for themselves.  A pair without
 * any allowed excitation gets occ_choice = (0, 0) and zero weights. */
inline void symm_pair_wt(uint8_t *occ_orbs, unsigned int num_elec, uint8_t *orb_symm, unsigned int num_orb, unsigned int (*unocc_sym_counts)[2],
                         uint8_t *occ_choice, double *virt_weights, uint8_t *virt_counts) {
    const unsigned int half = num_elec / 2, tri = *occ_choice;
    unsigned int hi = (unsigned int)((sqrt(tri * 8. + 1) - 1) / 2);
    const unsigned int lo = (unsigned int)(tri - hi * (hi + 1.) / 2);
    hi++;
    const unsigned int e_hi = occ_orbs[hi], e_lo = occ_orbs[lo], s_hi = hi / half, s_lo = lo / half;
    const unsigned int prod = orb_symm[e_hi % num_orb] ^ orb_symm[e_lo % num_orb];
    const bool par = s_hi == s_lo;
    const unsigned int self = (prod == 0 && par) ? 1u : 0u;
    unsigned int n_first = par ? num_orb - half : 2 * num_orb - num_elec;
    for (unsigned int s = 0; s < n_irreps; s++) {
        if (unocc_sym_counts[s ^ prod][s_lo] == self) n_first -= unocc_sym_counts[s][s_hi];
        if (!par && unocc_sym_counts[s ^ prod][s_hi] == self) n_first -= unocc_sym_counts[s][s_lo];
    }
    for (unsigned int k = 0; k < n_irreps; k++) virt_weights[k] = 0;
    if (n_first == 0) { occ_choice[0] = 0; occ_choice[1] = 0; return; }
    occ_choice[0] = (uint8_t)e_lo; occ_choice[1] = (uint8_t)e_hi;
    unsigned int top = 0;                                               // the product's highest bit, when the irreps pair up
    if (par && prod != 0) for (top = 4; !(prod & top); top >>= 1) {}
    unsigned int k = 0;
    for (unsigned int s = 0; s < n_irreps; s++) {
        if (top && !(s & top)) continue;
        const unsigned int na = unocc_sym_counts[s][s_hi];
        if (self) { if (na > 1) { virt_weights[k] = na * 1. / n_first; virt_counts[k] = (uint8_t)(na * (na - 1) / 2); } }
        else {
            const unsigned int nb = unocc_sym_counts[prod ^ s][s_lo];
            if (na != 0 && nb != 0) { virt_weights[k] = 1. * (na + nb) / n_first; virt_counts[k] = (uint8_t)(nb * na); }
        }
        k++;
    }
}
/* the index-th unoccupied orbital (0-based) of a symmetry row, as a spin orbital (near_uniform.cpp:419-433); 255 if there is none */
inline uint8_t virt_from_idx(uint8_t *det, uint8_t *lookup_row, uint8_t spin_shift, unsigned int index) {
    for (unsigned int k = 0; k < lookup_row[0]; k++) {
        const unsigned int orb = spin_shift + lookup_row[1 + k];
        if (fries_hip::det_bit(det, orb)) continue;
        if (index == 0) return (uint8_t)orb;
        index--;
    }
    return 255;
}
#endif /* near_uniform_h */
