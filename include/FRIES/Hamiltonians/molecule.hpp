/*! \file  FRIES/Hamiltonians/molecule.hpp for the MI355X build: the molecular-Hamiltonian helpers the drivers call, with the reference's
 * names and signatures (FRIES/Hamiltonians/molecule.hpp:43-301).
 *
 *   - diag_matrel (molecule.cpp:983-1029) is evaluated by the device function the engine itself uses (fries_matrel_batch); the single /
 *     double elements without sign (molecule.cpp:26-105), which the reference's drivers call once per sampled excitation, are host
 *     arithmetic on the same integrals.  The eris / h_core arguments must be the objects parse_fcidump returned; frozen orbitals are
 *     not supported;
 *   - symmetry-allowed excitation lists (sing_ex_symm, doub_ex_symm, count_singex, SymmInfo; molecule.cpp:108-203, 914-933) are host
 *     integer loops in the reference's enumeration order (they size buffers and define the order of H * trial);
 *   - h_op_offdiag / h_op_diag (molecule.cpp:205-219, 448-665) for HOST vectors -- the trial vector times H at start-up -- with the
 *     matrix elements of each determinant's excitations computed in one device batch.  spin_parity = +-1 (time-reversal symmetrised vectors): fries_hip::tr_adjust below. */
#ifndef molecule_h
#define molecule_h
#include <cmath>
#include <cstdint>
#include <cstring>
#include <utility>
#include <sstream>
#include <stdexcept>
#include <vector>
#include <FRIES/fci_utils.h>
#include <FRIES/ndarr.hpp>
#include <FRIES/vec_utils.hpp>
#include <FRIES/compress_utils.hpp>
#include <FRIES/backend.hpp>

#define n_irreps 8

/* which factorisation compresses the Hamiltonian (molecule.hpp:23-27) */
typedef enum { near_uni, heat_bath, unnorm_heat_bath } h_dist;

namespace fries_hip {
inline void check_mol(const void *eris, const void *hcore, unsigned n_frozen) {
    Backend &B = Backend::get();
    if (n_frozen) throw std::runtime_error("frozen orbitals are not supported by this build");
    if ((eris && eris != B.eris_obj) || (hcore && hcore != B.hcore_obj)) throw std::runtime_error("matrix elements are evaluated on the integrals parse_fcidump returned; other arrays are not supported");
}
inline uint64_t det_of_occ(const uint8_t *occ_orbs, unsigned n_elec) { uint64_t d = 0; for (unsigned i = 0; i < n_elec; i++) d |= 1ull << occ_orbs[i]; return d; }
}

/* <o1 o2 || u1 u2> without its sign: the direct integral, minus the exchange one for parallel spins (molecule.cpp:26-42).  Host
 * arithmetic on the integrals parse_fcidump read -- the reference's drivers call this once per sampled excitation; the engine's own
 * loops evaluate the same expression per lane (csrc/fries_dev.hpp: fr_doub_matrel). */
inline double doub_matr_el_nosgn(uint8_t *chosen_orbs, unsigned int n_orbs, const SymmERIs &eris, unsigned int n_frozen) {
    fries_hip::check_mol(&eris, nullptr, n_frozen);
    const unsigned int a = chosen_orbs[0] % n_orbs, b = chosen_orbs[1] % n_orbs, c = chosen_orbs[2] % n_orbs, d = chosen_orbs[3] % n_orbs;
    double el = eris.physicist(a, b, c, d);
    if (chosen_orbs[0] / n_orbs == chosen_orbs[1] / n_orbs) el -= eris.physicist(a, b, d, c);
    return el;
}
/* <D| H |D(o -> u)> without its sign: the one-electron integral plus, electron by electron in the order of the occupied list, the
 * Coulomb integral and (same spin) minus the exchange integral (molecule.cpp:76-105) */
inline double sing_matr_el_nosgn(uint8_t *chosen_orbs, uint8_t *occ_orbs, unsigned int n_orbs, const SymmERIs &eris, const Matrix<double> &h_core, unsigned int n_frozen, unsigned int n_elec) {
    fries_hip::check_mol(&eris, &h_core, n_frozen);
    const unsigned int o = chosen_orbs[0] % n_orbs, u = chosen_orbs[1] % n_orbs, spin = chosen_orbs[0] / n_orbs;
    double el = h_core(o, u);
    for (unsigned int j = 0; j < n_elec; j++) {
        const unsigned int p = occ_orbs[j] % n_orbs;
        el += eris.physicist(o, p, u, p);
        if (occ_orbs[j] / n_orbs == spin) el -= eris.physicist(o, p, p, u);
    }
    return el;
}
inline double diag_matrel(const uint8_t *occ_orbs, unsigned int /*n_orbs*/, const SymmERIs &eris, const Matrix<double> &h_core, unsigned int n_frozen, unsigned int n_elec) {
    fries_hip::check_mol(&eris, &h_core, n_frozen);
    uint64_t det = fries_hip::det_of_occ(occ_orbs, n_elec - n_frozen); double out = 0;
    fries_hip::ck(fries_matrel_batch(fries_hip::Backend::get().ctx(), 0, &det, nullptr, 1, &out, nullptr));
    return out;
}

/* opposite-spin pairs first, then up-up, then down-down; within each the loops run occupied pair, then virtual pair, ascending
 * (molecule.cpp:108-175) */
inline size_t doub_ex_symm(uint8_t *det, uint8_t *occ_orbs, unsigned int num_elec, unsigned int num_orb, uint8_t res_arr[][4], uint8_t *symm) {
    size_t n = 0;
    const unsigned half = num_elec / 2;
    auto irrep = [&](unsigned so) { return symm[so % num_orb]; };
    auto emit = [&](uint8_t a, uint8_t b, unsigned c, unsigned d) { res_arr[n][0] = a; res_arr[n][1] = b; res_arr[n][2] = (uint8_t)c; res_arr[n][3] = (uint8_t)d; n++; };
    for (unsigned i = 0; i < half; i++) for (unsigned j = half; j < num_elec; j++) {
        const uint8_t io = occ_orbs[i], jo = occ_orbs[j];
        const uint8_t occ_sym = (uint8_t)(irrep(io) ^ irrep(jo));
        for (unsigned k = 0; k < num_orb; k++) {
            if (read_bit(det, (uint8_t)k)) continue;
            for (unsigned l = num_orb; l < 2 * num_orb; l++) if (!read_bit(det, (uint8_t)l) && (occ_sym ^ irrep(k) ^ irrep(l)) == 0) emit(io, jo, k, l);
        }
    }
    for (unsigned spin = 0; spin < 2; spin++) {
        const unsigned e0 = spin * half, e1 = e0 + half, o0 = spin * num_orb, o1 = o0 + num_orb;
        for (unsigned i = e0; i < e1; i++) for (unsigned j = i + 1; j < e1; j++) {
            const uint8_t io = occ_orbs[i], jo = occ_orbs[j];
            const uint8_t occ_sym = (uint8_t)(irrep(io) ^ irrep(jo));
            for (unsigned k = o0; k < o1; k++) {
                if (read_bit(det, (uint8_t)k)) continue;
                for (unsigned l = k + 1; l < o1; l++) if (!read_bit(det, (uint8_t)l) && (occ_sym ^ irrep(k) ^ irrep(l)) == 0) emit(io, jo, k, l);
            }
        }
    }
    return n;
}
/* every electron in occ order, its spin's unoccupied orbitals of the same irrep ascending (molecule.cpp:178-203) */
inline size_t sing_ex_symm(uint8_t *det, uint8_t *occ_orbs, unsigned int num_elec, unsigned int num_orb, uint8_t res_arr[][2], uint8_t *symm) {
    size_t n = 0;
    for (unsigned i = 0; i < num_elec; i++) {
        const uint8_t io = occ_orbs[i];
        const unsigned o0 = (i < num_elec / 2) ? 0 : num_orb;
        for (unsigned a = o0; a < o0 + num_orb; a++) if (!read_bit(det, (uint8_t)a) && symm[io - o0] == symm[a - o0]) { res_arr[n][0] = io; res_arr[n][1] = (uint8_t)a; n++; }
    }
    return n;
}

/* row s: [count, orbitals of irrep s ...] (molecule.cpp gen_symm_lookup) */
inline void gen_symm_lookup(uint8_t *orb_symm, Matrix<uint8_t> &lookup_tabl) {
    const size_t n_orb = lookup_tabl.cols() - 1;
    for (unsigned s = 0; s < n_irreps; s++) lookup_tabl(s, 0) = 0;
    for (size_t orb = 0; orb < n_orb; orb++) { const uint8_t s = orb_symm[orb]; lookup_tabl(s, 0)++; lookup_tabl(s, lookup_tabl(s, 0)) = (uint8_t)orb; }
}
struct SymmInfo {
    std::vector<uint8_t> symm_vec;
    Matrix<uint8_t> symm_lookup;
    uint32_t max_n_symm;
    SymmInfo(uint8_t *symm, uint32_t n_orb) : symm_vec(symm, symm + n_orb), symm_lookup(n_irreps, n_orb + 1), max_n_symm(0) {
        gen_symm_lookup(symm, symm_lookup);
        for (uint8_t s = 0; s < n_irreps; s++) if (symm_lookup(s, 0) > max_n_symm) max_n_symm = symm_lookup(s, 0);
    }
};
inline size_t count_singex(uint8_t *det, const uint8_t *occ_orbs, uint32_t num_elec, SymmInfo *symm) {
    size_t n = 0;
    const size_t num_orb = symm->symm_vec.size();
    for (uint32_t e = 0; e < num_elec; e++) {
        const uint8_t orb = occ_orbs[e], s = symm->symm_vec[orb % num_orb];
        const size_t off = num_orb * (orb / num_orb);
        for (unsigned k = 0; k < symm->symm_lookup(s, 0); k++) if (!read_bit(det, (uint8_t)(symm->symm_lookup(s, k + 1) + off))) n++;
    }
    return n;
}

namespace fries_hip {
/* time-reversal symmetrised vectors (the adjust_tr lambda of the reference's h_op_offdiag, molecule.cpp:298-369, 472-552): the element
 * <new|H|cur> becomes the one between the symmetrised functions and *nw the representative it is added to (the byte-wise smaller of new and
 * its spin-flipped image).  false: no contribution.  Host arithmetic for host vectors; the engine's enumeration kernels apply the same rule
 * per candidate (csrc/hbpp_rows.hpp: fr_adjust_tr). */
inline bool tr_adjust(uint64_t cur, uint64_t *nw, uint8_t *occ_orbs, double *matr_el, int spin_parity, uint8_t *symm, unsigned int n_orbs,
                      const SymmERIs &eris, const Matrix<double> &h_core, unsigned int n_elec, uint8_t n_bytes) {
    auto flipped = [&](uint64_t d) { uint8_t in[8], out[8] = {0, 0, 0, 0, 0, 0, 0, 0}; memcpy(in, &d, 8); flip_spins(in, out, (uint8_t)n_orbs); uint64_t r = 0; memcpy(&r, out, n_bytes); return r; };
    auto cmp_bytes = [&](uint64_t a, uint64_t b) { return memcmp(&a, &b, n_bytes); };
    double norm = flipped(cur) == cur ? sqrt(2) : 1;
    const uint64_t img = flipped(*nw);
    if (img == cur) { *matr_el = 0; return false; }
    const int cmp = cmp_bytes(*nw, img);
    if (cmp == 0) {
        if (spin_parity == -1) { *matr_el = 0; return false; }
        *matr_el *= 2;
        norm *= sqrt(2);
    }
    else {
        uint8_t d[4], cb[8], ib[8];
        memcpy(cb, &cur, 8); memcpy(ib, &img, 8);
        const uint8_t n_diff = find_diff_bits(cb, ib, d, n_bytes);
        if (n_diff == 2 && symm[d[0] % n_orbs] == symm[d[1] % n_orbs]) {
            if (read_bit(cb, d[1])) std::swap(d[0], d[1]);
            double rev = sing_matr_el_nosgn(d, occ_orbs, n_orbs, eris, h_core, 0, n_elec);
            rev *= sing_parity(cb, d);
            *matr_el += rev * spin_parity;
            norm *= 2;
        }
        // the reference writes `a ^ b ^ c ^ d == 0`, which C++ reads as a ^ b ^ c ^ (d == 0): kept as written
        else if (n_diff == 4 && (symm[d[0] % n_orbs] ^ symm[d[1] % n_orbs] ^ symm[d[2] % n_orbs] ^ (unsigned)(symm[d[3] % n_orbs] == 0)) != 0) {
            if (read_bit(cb, d[2])) std::swap(read_bit(cb, d[0]) ? d[1] : d[0], d[2]);
            if (read_bit(cb, d[3])) std::swap(read_bit(cb, d[0]) ? d[1] : d[0], d[3]);
            if (d[0] > d[1]) std::swap(d[0], d[1]);
            if (d[2] > d[3]) std::swap(d[2], d[3]);
            double rev = doub_matr_el_nosgn(d, n_orbs, eris, 0);
            rev *= doub_parity(cb, d);
            *matr_el += rev * spin_parity;
            norm *= 2;
        }
    }
    if (cmp > 0) norm *= spin_parity;
    *matr_el /= norm;
    if (cmp > 0) *nw = img;
    return true;
}
}

/* vec[dest_idx] = (id_fac + h_fac * H_ii) vec[curr]  (molecule.cpp:205-219) */
inline void h_op_diag(DistVec<double> &vec, uint8_t dest_idx, double id_fac, double h_fac) {
    double *vals_before_mult = vec.values();
    vec.set_curr_vec_idx(dest_idx);
    for (size_t det_idx = 0; det_idx < vec.curr_size(); det_idx++) {
        double *target_val = vec[det_idx];
        const double curr_val = vals_before_mult[det_idx];
        *target_val = curr_val != 0 ? curr_val * (id_fac + h_fac * vec.matr_el_at_pos(det_idx)) : 0;
    }
}

/* vec[dest_idx] += h_fac * offdiag(H) vec[curr] for the first vec_size positions: all single excitations of every determinant in
 * position order, then all doubles, each pass cut into adder-sized pieces (molecule.cpp:448-665) */
inline void h_op_offdiag(DistVec<double> &vec, size_t vec_size, uint8_t *symm, unsigned int n_orbs, const SymmERIs &eris, const Matrix<double> &h_core,
                         uint8_t *orbs_scratch, size_t scratch_size, unsigned int n_frozen, unsigned int n_elec, uint8_t dest_idx, double h_fac, int spin_parity) {
    fries_hip::check_mol(&eris, &h_core, n_frozen);
    if (vec.bound()) throw std::runtime_error("h_op_offdiag on the device-bound vector: use the engine's frifull path (fries_frifull_iterate)");
    if (vec.num_vecs() <= dest_idx) throw std::runtime_error("The dest_idx argument exceeds the number of vectors stored in this object.");
    fries_ctx *cx = fries_hip::Backend::get().ctx();
    const uint8_t n_bytes = (uint8_t)CEILING(vec.n_bits(), 8);
    const uint8_t origin_idx = vec.curr_vec_idx();
    std::vector<uint64_t> dets; std::vector<double> els; std::vector<int32_t> sgn; std::vector<uint8_t> orbs4;
    for (int pass = 0; pass < 2; pass++) {
        size_t det_idx = 0, ex_idx = 0, n_ex_det = 0;
        uint8_t *occ_now = nullptr;
        double curr_el = 0;
        uint64_t curr_word = 0;
        int keep_going = 1;
        while (keep_going) {
            keep_going = 0;
            vec.set_curr_vec_idx(origin_idx);
            double *vals_before_mult = vec.values();
            vec.set_curr_vec_idx(dest_idx);
            while (true) {
                if (ex_idx >= n_ex_det) {
                    if (det_idx >= vec_size) break;
                    curr_el = vals_before_mult[det_idx];
                    if (curr_el == 0) { det_idx++; continue; }
                    uint8_t *curr_det = vec.indices()[det_idx];
                    uint8_t *occ_orbs = vec.orbs_at_pos(det_idx);
                    occ_now = occ_orbs;
                    n_ex_det = pass == 0 ? sing_ex_symm(curr_det, occ_orbs, n_elec, n_orbs, (uint8_t (*)[2])orbs_scratch, symm)
                                         : doub_ex_symm(curr_det, occ_orbs, n_elec, n_orbs, (uint8_t (*)[4])orbs_scratch, symm);
                    if (n_ex_det * (pass == 0 ? 2 : 4) > scratch_size) {
                        std::stringstream msg;
                        msg << "The memory passed via the orbs_scratch argument is not enough for " << (pass == 0 ? "single" : "double") << " excitations (" << scratch_size << " bytes provided)";
                        throw std::runtime_error(msg.str());
                    }
                    det_idx++;
                    if (n_ex_det == 0) continue;
                    ex_idx = 0;
                    curr_word = fries_word_of(curr_det, n_bytes);
                    // this determinant's matrix elements and signs in one device batch
                    dets.assign(n_ex_det, curr_word); els.resize(n_ex_det); sgn.resize(n_ex_det); orbs4.assign(4 * n_ex_det, 0);
                    for (size_t e = 0; e < n_ex_det; e++) for (int k = 0; k < (pass == 0 ? 2 : 4); k++) orbs4[4 * e + k] = orbs_scratch[(pass == 0 ? 2 : 4) * e + k];
                    fries_hip::ck(fries_matrel_batch(cx, pass == 0 ? 1 : 2, dets.data(), orbs4.data(), n_ex_det, els.data(), sgn.data()));
                }
                double matr_el = els[ex_idx];
                matr_el *= sgn[ex_idx];
                uint64_t new_word = curr_word;
                const uint8_t *o = &orbs4[4 * ex_idx];
                if (pass == 0) new_word = (new_word & ~(1ull << o[0])) | (1ull << o[1]);
                else new_word = (new_word & ~(1ull << o[0]) & ~(1ull << o[1])) | (1ull << o[2]) | (1ull << o[3]);
                ex_idx++;
                keep_going = 1;
                if (spin_parity && !fries_hip::tr_adjust(curr_word, &new_word, occ_now, &matr_el, spin_parity, symm, n_orbs, eris, h_core, n_elec, n_bytes)) continue;
                uint8_t new_det[8];
                memcpy(new_det, &new_word, 8);
                matr_el *= curr_el * h_fac;
                if (!vec.add(new_det, matr_el, 1)) break;
            }
            keep_going = sum_mpi(keep_going, fries_hip::mpi_rank(), fries_hip::mpi_size());     // every rank takes part in every round (molecule.cpp:607, 661)
            vec.perform_add(0);
        }
    }
    vec.set_curr_vec_idx(dest_idx);
}
inline void h_op_offdiag(DistVec<double> &vec, uint8_t *symm, unsigned int n_orbs, const SymmERIs &eris, const Matrix<double> &h_core,
                         uint8_t *orbs_scratch, size_t scratch_size, unsigned int n_frozen, unsigned int n_elec, uint8_t dest_idx, double h_fac, int spin_parity) {
    h_op_offdiag(vec, vec.curr_size(), symm, n_orbs, eris, h_core, orbs_scratch, scratch_size, n_frozen, n_elec, dest_idx, h_fac, spin_parity);
}
#endif /* molecule_h */
