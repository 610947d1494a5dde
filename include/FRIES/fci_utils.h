/* FRIES/fci_utils.h for the MI355X build (host, header-only): Hartree-Fock string, fermionic signs, excited determinants
 * (FRIES/fci_utils.c:14-148), and the time-reversal helpers flip_spins / tr_doub_connect (:158-204, :310-359). */
#ifndef fci_utils_h
#define fci_utils_h
#include "det_store.h"
#include "math_utils.h"
static inline void gen_hf_bitstring(unsigned int n_orb, unsigned int n_elec, uint8_t *det) {
    for (unsigned b = 0; b < CEILING(2 * n_orb, 8); b++) det[b] = 0;
    for (unsigned k = 0; k < n_elec / 2; k++) { set_bit(det, (uint8_t)k); set_bit(det, (uint8_t)(k + n_orb)); }
}
static inline int excite_sign(uint8_t cre_op, uint8_t des_op, uint8_t *det) { return (bits_between(det, cre_op, des_op) % 2 == 0) ? 1 : -1; }
static inline int sing_det_parity(uint8_t *det, uint8_t *orbs) { zero_bit(det, orbs[0]); int s = excite_sign(orbs[0], orbs[1], det); set_bit(det, orbs[1]); return s; }
static inline int sing_parity(uint8_t *det, uint8_t *orbs) { return excite_sign(orbs[0], orbs[1], det); }
static inline void sing_det(uint8_t *det, uint8_t *orbs) { zero_bit(det, orbs[0]); set_bit(det, orbs[1]); }
static inline int doub_det_parity(uint8_t *det, uint8_t *orbs) {
    zero_bit(det, orbs[0]); zero_bit(det, orbs[1]);
    int s = excite_sign(orbs[2], orbs[0], det); s *= excite_sign(orbs[3], orbs[1], det);
    set_bit(det, orbs[2]); set_bit(det, orbs[3]);
    return s;
}
static inline void doub_det(uint8_t *det, uint8_t *orbs) { zero_bit(det, orbs[0]); zero_bit(det, orbs[1]); set_bit(det, orbs[2]); set_bit(det, orbs[3]); }
static inline int doub_parity(uint8_t *det, uint8_t *orbs) {
    zero_bit(det, orbs[0]); zero_bit(det, orbs[1]);
    int s = excite_sign(orbs[2], orbs[0], det); s *= excite_sign(orbs[3], orbs[1], det);
    set_bit(det, orbs[0]); set_bit(det, orbs[1]);
    return s;
}
static inline int excite_sign_occ(uint8_t occ_idx, uint8_t virt_orb, const uint8_t *occ_orbs, uint32_t n_elec) {
    uint32_t n_perm = 1;
    if (occ_orbs[occ_idx] < virt_orb) { while (occ_idx + n_perm < n_elec && occ_orbs[occ_idx + n_perm] < virt_orb) n_perm++; }
    else { while (n_perm <= occ_idx && occ_orbs[occ_idx - n_perm] > virt_orb) n_perm++; }
    n_perm++;
    return (n_perm % 2 == 0) ? 1 : -1;
}
/* the n-th (0-based) unoccupied orbital of the given spin, as a spin-orbital index (fci_utils.c:138-148) */
static inline uint8_t find_nth_virt(uint8_t *occ_orbs, int spin, uint8_t n_elec, uint8_t n_orb, uint8_t n) {
    uint8_t virt_orb = (uint8_t)(n_orb * spin + n);
    for (size_t i = (size_t)(n_elec / 2) * spin; i < (size_t)(n_elec / 2) * (spin + 1) && occ_orbs[i] <= virt_orb; i++) virt_orb++;
    return virt_orb;
}
/* the determinant with the alpha and beta strings exchanged (fci_utils.c:158-204).  For strings that are whole bytes long and at least three
 * of them (n_orb = 24, 32) the reference's byte loop is off by one -- bytes mid + 1 .. n_bytes - 2 receive alpha byte b - mid - 1 -- and that
 * is what this returns too: a vector symmetrised by the reference is symmetrised under the reference's map. */
static inline void flip_spins(uint8_t *det_in, uint8_t *det_out, uint8_t n_orb) {
    const uint8_t n_bytes = (uint8_t)CEILING(2 * n_orb, 8);
    uint64_t d = 0;
    memcpy(&d, det_in, n_bytes > 8 ? 8 : n_bytes);
    const uint64_t half = n_orb >= 64 ? ~0ull : (1ull << n_orb) - 1ull;
    uint64_t out = n_orb >= 32 ? (d >> 32) | (d << 32) : ((d >> n_orb) & half) | ((d & half) << n_orb);
    if (n_orb % 8 == 0 && n_orb >= 24) {
        const unsigned mid = n_orb / 8u, nb = 2 * mid;
        for (unsigned b = mid + 1; b + 1 < nb; b++) out = (out & ~(0xffull << (8 * b))) | (((d >> (8 * (b - mid - 1))) & 0xffull) << (8 * b));
    }
    memcpy(det_out, &out, n_bytes > 8 ? 8 : n_bytes);
}
/* how a determinant is connected to its spin-flipped image (fci_utils.c:310-359): 0 = they are the same, 1 = one orbital of either spin
 * differs (diff_idx = the alpha electron's index, the beta electron's index), 2 = more */
static inline int tr_doub_connect(const uint8_t *occ_orbs, uint32_t n_orb, uint32_t n_elec, uint8_t *diff_idx) {
    const uint32_t half = n_elec / 2;
    uint32_t ia = 0, ib = 0;
    int extra_a = 0, extra_b = 0, same = 1;
    for (uint32_t k = 0; k < half && same; k++) same = occ_orbs[k] == occ_orbs[half + k] - n_orb;
    if (same) return 0;
    while (ia < half && ib < half) {
        const int a = occ_orbs[ia], b = (int)occ_orbs[half + ib] - (int)n_orb;
        if (a == b) { ia++; ib++; }
        else if (a > b) { if (extra_b) return 2; extra_b = 1; diff_idx[1] = (uint8_t)(half + ib); ib++; }
        else { if (extra_a) return 2; extra_a = 1; diff_idx[0] = (uint8_t)ia; ia++; }
    }
    if (ia < half) diff_idx[0] = (uint8_t)ia;
    else if (ib < half) diff_idx[1] = (uint8_t)(half + ib);
    return 1;
}
static inline void sing_ex_orbs(uint8_t *curr_orbs, uint8_t *new_orbs, uint8_t *ex_orbs, uint8_t n_elec) {
    memcpy(new_orbs, curr_orbs, n_elec);
    uint8_t shift = (uint8_t)((ex_orbs[0] / (n_elec / 2)) * (n_elec / 2));
    new_sorted(curr_orbs + shift, new_orbs + shift, (uint8_t)(n_elec / 2), (uint8_t)(ex_orbs[0] - shift), ex_orbs[1]);
}
#endif
