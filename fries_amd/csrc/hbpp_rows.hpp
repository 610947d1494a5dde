// Heat-bath Power-Pitzer sub-weight rows, recomputed on the fly from LDS-resident tables.
//
// The reference materialises a (4*mat_nonz x n_sub) matrix of normalised probabilities per
// stage (FRIES/Hamiltonians/heat_bathPP.hpp:289-291, rows filled by calc_*_probs,
// heat_bathPP.cpp:182-412).  Here a row is a pure function of (determinant, orbital code),
// evaluated per lane whenever a kernel needs it: first the norm in the reference's own
// summation order, then the entries raw * (1 / norm) in ascending sub-index order.
//
// Stage numbering (heat_bathPP.cpp:713-915): 1 singles|doubles, 2 first occupied,
// 3 second occupied (double) / virtual (single), 4 first virtual, 5 second virtual.
// Orbital code bytes c[0..3] follow the reference's orb_indices{1,2}: c[0] = 1 for a single
// excitation, c[1] = index of the first occupied electron, c[2] = index of the second
// occupied electron (double) or virtual index (single), c[3] = first virtual orbital (double)
// or number of allowed virtuals (single).
#pragma once
#include "fries_dev.hpp"

struct RowInfo {
    double inv_norm;    // multiply raw weights by this
    double tot;         // the value calc_*_probs returns
    uint32_t nsub;      // row length seen by comp_sub (sub_sizes or column count)
    uint32_t aux;       // stage specific (exclude_first / irrep / same-spin flags)
};

__device__ __forceinline__ uint32_t fr_code(unsigned c0, unsigned c1, unsigned c2, unsigned c3) { return c0 | (c1 << 8) | (c2 << 16) | (c3 << 24); }
__device__ __forceinline__ unsigned fr_c(uint32_t code, int i) { return (code >> (8 * i)) & 0xffu; }

// FRIES/Hamiltonians/near_uniform.cpp:14-28: virtual orbitals per irrep and spin, packed as
// 16 nibbles-wide bytes: cnt[ir][spin].
struct SymCounts { uint8_t c[8][2]; };
__device__ inline void fr_count_symm_virt(SymCounts &sc, const HbTables &T, det_t det) {
    for (int i = 0; i < 8; i++) { sc.c[i][0] = T.lookup[i][0]; sc.c[i][1] = T.lookup[i][0]; }
    unsigned n = T.n_orb;
    for (det_t a = det; a; a &= a - 1) {
        unsigned o = __ffsll((long long)a) - 1;
        sc.c[T.irrep[o % n]][o / n] -= 1;
    }
}
// FRIES/Hamiltonians/near_uniform.cpp:316-327
__device__ inline unsigned fr_count_sing_allowed(const HbTables &T, det_t det) {
    SymCounts sc; fr_count_symm_virt(sc, T, det);
    unsigned n = T.n_orb, cnt = 0;
    for (det_t a = det; a; a &= a - 1) {
        unsigned o = __ffsll((long long)a) - 1;
        if (sc.c[T.irrep[o % n]][o / n] != 0) cnt++;
    }
    return cnt;
}
// FRIES/Hamiltonians/near_uniform.cpp:330-347: choice -> electron index, returns #virtuals
__device__ inline unsigned fr_count_sing_virt(const HbTables &T, det_t det, unsigned *choice) {
    SymCounts sc; fr_count_symm_virt(sc, T, det);
    unsigned n = T.n_orb, cnt = 0, e = 0;
    for (det_t a = det; a; a &= a - 1, e++) {
        unsigned o = __ffsll((long long)a) - 1;
        unsigned va = sc.c[T.irrep[o % n]][o / n];
        if (va != 0) {
            if (cnt == *choice) { *choice = e; return va; }
            cnt++;
        }
    }
    return 0;
}
// FRIES/Hamiltonians/near_uniform.cpp:419-433
__device__ inline unsigned fr_virt_from_idx(const HbTables &T, det_t det, unsigned irrep, unsigned spin_shift, unsigned index) {
    for (unsigned k = 0; k < T.lookup[irrep][0]; k++) {
        unsigned orb = spin_shift + T.lookup[irrep][1 + k];
        if (!fr_bit(det, orb)) {
            if (index == 0) return orb;
            index--;
        }
    }
    return 255;
}
// FRIES/fci_utils.c:138-148: n-th virtual orbital of a spin channel
__device__ inline unsigned fr_find_nth_virt(det_t det, int spin, unsigned n_orb, unsigned n) {
    det_t chan = (det >> (spin * n_orb)) & ((1ull << n_orb) - 1ull);
    unsigned virt = n;
    for (det_t a = chan; a; a &= a - 1) {
        unsigned o = __ffsll((long long)a) - 1;
        if (o <= virt) virt++; else break;
    }
    return virt + spin * n_orb;
}

__device__ __forceinline__ double fr_exch_or_diag(const HbTables &T, unsigned o, unsigned u) {
    if (o == u) return T.diag_sqrt[o];
    unsigned mn = o < u ? o : u, mx = o > u ? o : u;
    return T.exch_sqrt[fr_tri_nodiag(mn, mx)];
}

// ------------------------------------------------------------------ row generators
// visit<STAGE, NEW_HB>(T, det, code, f): calls f(sub_index, raw_weight) for sub_index ascending.
// setup<STAGE, NEW_HB>(...): norm in the reference's order -> RowInfo.

// ---- stage 1: {p_doub, 1 - p_doub}
// ---- stage 2: calc_o1_probs (heat_bathPP.cpp:182-200)
template <bool NEW_HB, class F>
__device__ __forceinline__ void fr_row2_visit(const HbTables &T, det_t det, F f) {
    unsigned n = T.n_orb, k = 0;
    for (det_t a = det; a; a &= a - 1, k++) {
        if (NEW_HB && k == 0) continue;
        unsigned o = __ffsll((long long)a) - 1;
        f(k - (NEW_HB ? 1 : 0), T.s_tens[o % n]);
    }
}
template <bool NEW_HB>
__device__ inline RowInfo fr_row2_setup(const HbTables &T, det_t det) {
    double norm = 0;
    fr_row2_visit<NEW_HB>(T, det, [&](unsigned, double w) { norm += w; });
    RowInfo r; r.inv_norm = 1. / norm; r.tot = norm / T.s_norm; r.nsub = T.n_elec - (NEW_HB ? 1 : 0); r.aux = 0;
    return r;
}

// ---- stage 3, HB: calc_o2_probs (heat_bathPP.cpp:203-233); sub = electron index
template <class F>
__device__ __forceinline__ void fr_row3_visit(const HbTables &T, det_t det, unsigned o1_idx, unsigned o1, F f) {
    unsigned n = T.n_orb, k = 0, o1s = o1 % n, sp = o1 / n;
    for (det_t a = det; a; a &= a - 1, k++) {
        unsigned o = __ffsll((long long)a) - 1, os = o % n;
        double w;
        if (k == o1_idx) w = 0;
        else if (o / n != sp) w = T.d_diff[o1s * n + os];
        else w = (k < o1_idx) ? T.d_same[fr_tri_nodiag(os, o1s)] : T.d_same[fr_tri_nodiag(o1s, os)];
        f(k, w);
    }
}
__device__ inline RowInfo fr_row3_setup(const HbTables &T, det_t det, unsigned o1_idx) {
    unsigned n = T.n_orb, o1 = fr_nth_bit(det, o1_idx), sp = o1 / n;
    double norm = 0;
    // opposite-spin block first, then same spin (the reference's accumulation order)
    fr_row3_visit(T, det, o1_idx, o1, [&](unsigned k, double w) { unsigned spk = k / (T.n_elec / 2); if (spk != sp) norm += w; });
    fr_row3_visit(T, det, o1_idx, o1, [&](unsigned k, double w) { unsigned spk = k / (T.n_elec / 2); if (spk == sp && k != o1_idx) norm += w; });
    RowInfo r; r.inv_norm = 1. / norm; r.tot = norm / T.s_tens[o1 % n]; r.nsub = T.n_elec; r.aux = o1;
    return r;
}
// ---- stage 3, HB_unnorm: calc_o2_probs_half (:236-270); sub = electron index < o1_idx
template <class F>
__device__ __forceinline__ void fr_row3h_visit(const HbTables &T, det_t det, unsigned o1_idx, unsigned o1, F f) {
    unsigned n = T.n_orb, k = 0, sp = o1 / n, half = T.n_elec / 2;
    for (det_t a = det; a && k < o1_idx; a &= a - 1, k++) {
        unsigned o = __ffsll((long long)a) - 1;
        double w;
        if (k < half) w = (sp == 0) ? T.d_same[fr_tri_nodiag(o, o1)] : T.d_diff[(o1 - n) * n + o];
        else w = (sp == 0) ? T.d_diff[o1 * n + o - n] : T.d_same[fr_tri_nodiag(o - n, o1 - n)];
        f(k, w);
    }
}
__device__ inline RowInfo fr_row3h_setup(const HbTables &T, det_t det, unsigned o1_idx) {
    unsigned o1 = fr_nth_bit(det, o1_idx);
    double norm = 0;
    fr_row3h_visit(T, det, o1_idx, o1, [&](unsigned, double w) { norm += w; });
    RowInfo r; r.inv_norm = 1. / norm; r.tot = norm / T.s_tens[o1 % T.n_orb]; r.nsub = o1_idx; r.aux = o1;
    return r;
}

// ---- stage 4: calc_u1_probs (:273-319); sub = index among the virtuals of o1's spin
template <class F>
__device__ __forceinline__ void fr_row4_visit(const HbTables &T, det_t det, unsigned o1, bool excl_first, F f) {
    unsigned n = T.n_orb, sp = o1 / n, o1s = o1 % n, pi = 0;
    det_t chan = (det >> (sp * n)) & ((1ull << n) - 1ull);
    for (unsigned k = 0; k < n; k++) {
        if (k == o1s || fr_bit(chan, k)) continue;
        double w = (k < o1s) ? T.exch_sqrt[fr_tri_nodiag(k, o1s)] : T.exch_sqrt[fr_tri_nodiag(o1s, k)];
        if (excl_first && pi == 0) w = 0;
        f(pi, w);
        pi++;
    }
}
__device__ inline RowInfo fr_row4_setup(const HbTables &T, det_t det, unsigned o1, bool excl_first) {
    double norm = 0, first = 0;
    fr_row4_visit(T, det, o1, false, [&](unsigned pi, double w) { norm += w; if (pi == 0) first = w; });
    if (excl_first) norm -= first;
    RowInfo r; r.inv_norm = 1. / norm; r.tot = norm / T.exch_norms[o1 % T.n_orb]; r.nsub = T.n_orb - T.n_elec / 2; r.aux = excl_first;
    return r;
}

// ---- stage 5: calc_u2_probs (:322-365) / calc_u2_probs_half (:368-412); sub = index in the irrep's orbital list
template <bool NEW_HB, class F>
__device__ __forceinline__ unsigned fr_row5_visit(const HbTables &T, det_t det, unsigned o1, unsigned o2, unsigned u1, F f) {
    unsigned n = T.n_orb, o2s = o2 % n, u1s = u1 % n, u2_spin = o2 / n;
    bool same = (o1 / n) == u2_spin;
    unsigned ir = T.irrep[o1 % n] ^ T.irrep[o2s] ^ T.irrep[u1s];
    unsigned num = T.lookup[ir][0], k;
    for (k = 0; k < num; k++) {
        unsigned u2 = T.lookup[ir][k + 1];
        if (NEW_HB && same && u2 >= u1s) break;
        bool ok = (same && u2 != u1s) || !same;
        if (NEW_HB) ok = ok && !fr_bit(det, u2 + n * u2_spin);
        f(k, ok ? fr_exch_or_diag(T, o2s, u2) : 0.0);
    }
    return k;
}
template <bool NEW_HB>
__device__ inline RowInfo fr_row5_setup(const HbTables &T, det_t det, unsigned o1, unsigned o2, unsigned u1) {
    double norm = 0;
    unsigned len = fr_row5_visit<NEW_HB>(T, det, o1, o2, u1, [&](unsigned, double w) { norm += w; });
    RowInfo r;
    r.inv_norm = (norm != 0) ? 1 / norm : 1.0;   // rows with zero norm stay all-zero
    r.tot = norm / T.exch_norms[o2 % T.n_orb];
    r.nsub = len; r.aux = 0;
    return r;
}

// ------------------------------------------------------------------ final weights
// FRIES/Hamiltonians/heat_bathPP.cpp:414-439
__device__ inline double fr_unnorm_wt(const HbTables &T, unsigned O1, unsigned O2, unsigned U1, unsigned U2) {
    unsigned n = T.n_orb;
    unsigned o1 = O1 % n, o2 = O2 % n, u1 = U1 % n, u2 = U2 % n;
    unsigned mn11 = o1 < u1 ? o1 : u1, mx11 = o1 > u1 ? o1 : u1;
    unsigned mn22 = o2 < u2 ? o2 : u2, mx22 = o2 > u2 ? o2 : u2;
    bool same = (O1 / n) == (O2 / n);
    double w;
    if (same)
        w = T.d_same[fr_tri_nodiag(o1, o2)] * (T.exch_sqrt[fr_tri_nodiag(mn11, mx11)] * T.exch_sqrt[fr_tri_nodiag(mn22, mx22)]) / T.s_norm / T.exch_norms[o1] / T.exch_norms[o2];
    else
        w = (T.d_diff[o2 * n + o1]) * T.exch_sqrt[fr_tri_nodiag(mn11, mx11)] * T.exch_sqrt[fr_tri_nodiag(mn22, mx22)] / T.s_norm / T.exch_norms[o1] / T.exch_norms[o2];
    return w;
}

// FRIES/Hamiltonians/heat_bathPP.cpp:442-598
__device__ inline double fr_norm_wt(const HbTables &T, det_t det, unsigned O1, unsigned O2, unsigned U1, unsigned U2) {
    unsigned n = T.n_orb;
    unsigned o1 = O1 % n, o2 = O2 % n, u1 = U1 % n, u2 = U2 % n;
    unsigned o1_spin = O1 / n, o2_spin = O2 / n;
    unsigned mn11 = o1 < u1 ? o1 : u1, mx11 = o1 > u1 ? o1 : u1;
    unsigned mn22 = o2 < u2 ? o2 : u2, mx22 = o2 > u2 ? o2 : u2;
    bool same = o1_spin == o2_spin;
    det_t lowmask = (1ull << n) - 1ull;
    det_t chan[2] = {det & lowmask, det >> n};
    double s_denom = 0;
    for (det_t a = chan[0]; a; a &= a - 1) s_denom += T.s_tens[__ffsll((long long)a) - 1];
    for (det_t a = chan[1]; a; a &= a - 1) s_denom += T.s_tens[__ffsll((long long)a) - 1];
    auto d_denom = [&](unsigned o, unsigned sp) {
        double d = 0;
        for (det_t a = chan[1 - sp]; a; a &= a - 1) d += T.d_diff[o * n + (__ffsll((long long)a) - 1)];
        for (det_t a = chan[sp]; a; a &= a - 1) {
            unsigned k = __ffsll((long long)a) - 1;
            if (k < o) d += T.d_same[fr_tri_nodiag(k, o)];
            else if (k > o) d += T.d_same[fr_tri_nodiag(o, k)];
        }
        return d;
    };
    double d1 = d_denom(o1, o1_spin), d2 = d_denom(o2, o2_spin);
    auto e_virt = [&](unsigned o, unsigned sp) {
        double e = 0;
        for (unsigned k = 0; k < o; k++) if (!fr_bit(chan[sp], k)) e += T.exch_sqrt[fr_tri_nodiag(k, o)];
        for (unsigned k = o + 1; k < n; k++) if (!fr_bit(chan[sp], k)) e += T.exch_sqrt[fr_tri_nodiag(o, k)];
        return e;
    };
    double e1v = e_virt(o1, o1_spin), e2v = e_virt(o2, o2_spin);
    unsigned u1_ir = T.irrep[u1], u2_ir = T.irrep[u2];
    double e2s_no1 = 0, e2s_no2 = 0, e1s_no1 = 0, e1s_no2 = 0;
    for (unsigned k = 0; k < T.lookup[u2_ir][0]; k++) {
        unsigned so = T.lookup[u2_ir][k + 1];
        if ((same && so != u1) || !same) { e2s_no1 += fr_exch_or_diag(T, o2, so); e1s_no1 += fr_exch_or_diag(T, o1, so); }
    }
    for (unsigned k = 0; k < T.lookup[u1_ir][0]; k++) {
        unsigned so = T.lookup[u1_ir][k + 1];
        if ((same && so != u2) || !same) { e2s_no2 += fr_exch_or_diag(T, o2, so); e1s_no2 += fr_exch_or_diag(T, o1, so); }
    }
    unsigned o1u1 = fr_tri_nodiag(mn11, mx11), o2u2 = fr_tri_nodiag(mn22, mx22);
    double w;
    if (same) {
        unsigned mn12 = o1 < u2 ? o1 : u2, mx12 = o1 > u2 ? o1 : u2;
        unsigned mn21 = o2 < u1 ? o2 : u1, mx21 = o2 > u1 ? o2 : u1;
        unsigned o1o2 = fr_tri_nodiag(o1, o2), o1u2 = fr_tri_nodiag(mn12, mx12), o2u1 = fr_tri_nodiag(mn21, mx21);
        w = T.d_same[o1o2] / s_denom * (
            T.s_tens[o1] / d1 / e1v * (T.exch_sqrt[o1u1] * T.exch_sqrt[o2u2] / e2s_no1 + T.exch_sqrt[o1u2] * T.exch_sqrt[o2u1] / e2s_no2) +
            T.s_tens[o2] / d2 / e2v * (T.exch_sqrt[o2u1] * T.exch_sqrt[o1u2] / e1s_no1 + T.exch_sqrt[o2u2] * T.exch_sqrt[o1u1] / e1s_no2));
    }
    else {
        w = (T.s_tens[o1] * T.d_diff[o1 * n + o2] / d1 / e1v / e2s_no1 + T.s_tens[o2] * T.d_diff[o2 * n + o1] / d2 / e2v / e1s_no2) * T.exch_sqrt[o1u1] * T.exch_sqrt[o2u2] / s_denom;
    }
    return w;
}

// ------------------------------------------------------------------ time-reversal symmetry
// FRIES/fci_utils.c:158-204: the alpha and the beta string trade places -- with the reference's own slip for strings that are whole bytes
// long and at least three of them (n_orb = 24, 32): bytes mid + 1 .. n_bytes - 2 of the result receive alpha byte b - mid - 1.
__device__ __forceinline__ det_t fr_flip_spins(det_t det, unsigned n) {
    const det_t half = (1ull << n) - 1ull;
    det_t out = n >= 32 ? (det >> 32) | (det << 32) : ((det >> n) & half) | ((det & half) << n);
    if ((n & 7u) == 0 && n >= 24) {
        const unsigned mid = n / 8, nb = 2 * mid;
        for (unsigned b = mid + 1; b + 1 < nb; b++) out = (out & ~(0xffull << (8 * b))) | (((det >> (8 * (b - mid - 1))) & 0xffull) << (8 * b));
    }
    return out;
}
// memcmp over the little-endian byte strings
__device__ __forceinline__ int fr_det_memcmp(det_t a, det_t b) {
    const det_t x = a ^ b;
    if (!x) return 0;
    const int byte = (__ffsll((long long)x) - 1) >> 3;
    return ((a >> (8 * byte)) & 255ull) > ((b >> (8 * byte)) & 255ull) ? 1 : -1;
}
// The adjust_tr lambda of h_op_offdiag (molecule.cpp:298-369, 472-552) and the same block of apply_HBPP_piv (heat_bathPP.cpp:1326-1407):
// <new|H|cur> -> the element between the symmetrised functions.  false: no contribution.  *target = the representative the element goes
// to (the byte-wise smaller of new and its image).  tw != nullptr selects apply_HBPP_piv's form: the image's selection probability is added to
// *tw and the "two excitations" factor of h_op_offdiag is left out.  (`a ^ b ^ c ^ d == 0` in the reference parses as a ^ b ^ c ^ (d == 0): kept.)
__device__ inline bool fr_adjust_tr(const HbTables &T, const SysDev &S, det_t cur, det_t nd, double *matr_el, int spin_parity, det_t *target,
                                    int unit_matrel, double *tw, double p_doub) {
    const unsigned n = T.n_orb;
    double norm = fr_flip_spins(cur, n) == cur ? 1.4142135623730951 : 1.0;           // sqrt(2)
    const det_t img = fr_flip_spins(nd, n);
    if (img == cur) { *matr_el = 0; return false; }
    const int cmp = fr_det_memcmp(nd, img);
    if (cmp == 0) {
        if (spin_parity == -1) { *matr_el = 0; return false; }
        *matr_el *= 2;
        norm *= 1.4142135623730951;
    }
    else {
        const det_t x = cur ^ img;
        const int n_diff = __popcll(x);
        unsigned d[4] = {0, 0, 0, 0};
        if (n_diff <= 4) { int k = 0; for (det_t y = x; y; y &= y - 1) d[k++] = (unsigned)(__ffsll((long long)y) - 1); }
        if (n_diff == 2) {
            if (T.irrep[d[0] % n] == T.irrep[d[1] % n]) {
                if (fr_bit(cur, d[1])) { const unsigned t = d[0]; d[0] = d[1]; d[1] = t; }
                if (tw) { SymCounts sc; fr_count_symm_virt(sc, T, cur); *tw += (1 - p_doub) / fr_count_sing_allowed(T, cur) / sc.c[T.irrep[d[0] % n]][0]; }
                double rev = unit_matrel ? 1.0 : fr_sing_matrel(cur, d[0], d[1], S.h_core, S.eris, n);
                rev *= fr_sing_parity(cur, d[0], d[1]);
                *matr_el += rev * spin_parity;
                if (!tw) norm *= 2;
            }
        }
        else if (n_diff == 4) {
            if ((T.irrep[d[0] % n] ^ T.irrep[d[1] % n] ^ T.irrep[d[2] % n] ^ (unsigned)(T.irrep[d[3] % n] == 0)) != 0) {
                unsigned t;
                if (fr_bit(cur, d[2])) { if (fr_bit(cur, d[0])) { t = d[1]; d[1] = d[2]; d[2] = t; } else { t = d[0]; d[0] = d[2]; d[2] = t; } }
                if (fr_bit(cur, d[3])) { if (fr_bit(cur, d[0])) { t = d[1]; d[1] = d[3]; d[3] = t; } else { t = d[0]; d[0] = d[3]; d[3] = t; } }
                if (d[0] > d[1]) { t = d[0]; d[0] = d[1]; d[1] = t; }
                if (d[2] > d[3]) { t = d[2]; d[2] = d[3]; d[3] = t; }
                if (tw) *tw += fr_unnorm_wt(T, d[0], d[1], d[2], d[3]) * p_doub;
                double rev = unit_matrel ? 1.0 : fr_doub_matrel(d[0], d[1], d[2], d[3], S.eris, n);
                rev *= fr_doub_parity(cur, d[0], d[1], d[2], d[3]);
                *matr_el += rev * spin_parity;
                if (!tw) norm *= 2;
            }
        }
    }
    if (cmp > 0) norm *= spin_parity;
    *matr_el /= norm;
    *target = cmp > 0 ? img : nd;
    return true;
}

