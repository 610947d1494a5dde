// TEST INFRASTRUCTURE ONLY.  Pins oracle/fries_oracle.cpp against the REAL
// reference: this file is our code, but it #includes the reference's headers
// from /root/reference and links oracle/_ref/libfries_ref.so (the reference's own
// sources compiled in place by oracle/Makefile).  It only builds where
// /root/reference exists; its outputs (tests/golden/*) are what travels.
//
//   ref_harness unit                       function-by-function bit comparison on random inputs
//   ref_harness hbpp_all <out>             tests/test_hamiltonian.cpp:454-520 ([new_hb_all]) through both
//   ref_harness frisys <fcidump> <pg> <n_iter> <seed> <eps> <vec_nonz> <mat_nonz> <max_dets> <init> <target> <HB|HB_unnorm> <out> [snap_every]
//                                          FRIES_bin/frisys_mol.cpp loop (1 rank, HF start, HF trial) vs fo::Frisys, lockstep
//   ref_harness dump_ints <fcidump> <pg> <out>   what the reference's parse_fcidump read (binary)
//   ref_harness time <fcidump> <pg> <n_iter> ...same as frisys...   reference loop only, prints seconds
#include <FRIES/Hamiltonians/near_uniform.hpp>
#include <FRIES/io_utils.hpp>
#include <FRIES/compress_utils.hpp>
#include <FRIES/Hamiltonians/heat_bathPP.hpp>
#include <FRIES/Hamiltonians/molecule.hpp>
#include <FRIES/vec_utils.hpp>
#include <FRIES/hh_vec.hpp>
#include <FRIES/Hamiltonians/hub_holstein.hpp>
#include <chrono>
#include <fstream>
#include <cstdio>
#include <cstring>
#include <cinttypes>
#include "fries_oracle.hpp"

static int n_fail = 0, n_chk = 0;
#define CHECK(cond, ...) do { n_chk++; if (!(cond)) { n_fail++; if (n_fail < 30) { printf("MISMATCH %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } } while (0)

static inline bool same_bits(double a, double b) { return memcmp(&a, &b, 8) == 0; }
static fo::det_t to_u64(const uint8_t *bytes, size_t n) { fo::det_t d = 0; memcpy(&d, bytes, n); return d; }
static void from_u64(fo::det_t d, uint8_t *bytes, size_t n) { memcpy(bytes, &d, n); }

static void fill_oracle_ints(fo::Integrals &oi, const SymmERIs &eris, const Matrix<double> &h, unsigned n) {
    oi.n_orb = n;
    oi.h.resize((size_t)n * n);
    for (unsigned i = 0; i < n; i++) for (unsigned j = 0; j < n; j++) oi.h[i * n + j] = h(i, j);
    oi.eri.assign(fo::Integrals::packed_len(n), 0.0);
    for (unsigned j = 0; j < n; j++) for (unsigned i = 0; i <= j; i++)
        for (unsigned l = 0; l < n; l++) for (unsigned k = 0; k <= l; k++) {
            size_t p1 = (size_t)j * (j + 1) / 2 + i, p2 = (size_t)l * (l + 1) / 2 + k;
            if (p1 <= p2) oi.eri[p2 * (p2 + 1) / 2 + p1] = eris.chemist(i, j, k, l);
        }
}

static fo::det_t rand_det(std::mt19937 &mt, unsigned n_orb, unsigned n_elec) {
    fo::det_t d = 0;
    for (int sp = 0; sp < 2; sp++) {
        unsigned placed = 0;
        while (placed < n_elec / 2) {
            unsigned o = mt() % n_orb + sp * n_orb;
            if (!((d >> o) & 1)) { d |= 1ull << o; placed++; }
        }
    }
    return d;
}

// ------------------------------------------------------------------ unit
static int run_unit() {
    std::mt19937 mt(1234);
    const unsigned n_orb = 26, n_elec = 10, nb = CEILING(2 * n_orb, 8);
    // random symmetric integrals through the reference's own container
    SymmERIs eris(n_orb);
    Matrix<double> h(n_orb, n_orb);
    uint8_t symm[26];
    for (unsigned i = 0; i < n_orb; i++) symm[i] = mt() % 8;
    for (unsigned j = 0; j < n_orb; j++) for (unsigned i = 0; i <= j; i++) h(i, j) = h(j, i) = mt() / (1. + UINT32_MAX) - 0.5;
    for (unsigned j = 0; j < n_orb; j++) for (unsigned i = 0; i <= j; i++)
        for (unsigned l = 0; l < n_orb; l++) for (unsigned k = 0; k <= l; k++) {
            size_t p1 = (size_t)j * (j + 1) / 2 + i, p2 = (size_t)l * (l + 1) / 2 + k;
            if (p1 <= p2) eris.chemist_ordered(i, j, k, l) = mt() / (1. + UINT32_MAX) - 0.5;
        }
    fo::MolSys sys; sys.n_orb = n_orb; sys.n_elec = n_elec;
    fill_oracle_ints(sys.ints, eris, h, n_orb);
    sys.symm.init(symm, n_orb);
    SymmInfo rsymm(symm, n_orb);
    CHECK(rsymm.max_n_symm == sys.symm.max_n_symm, "max_n_symm");
    for (unsigned s = 0; s < 8; s++) for (unsigned c = 0; c <= n_orb; c++) if (c <= rsymm.symm_lookup(s, 0)) CHECK(rsymm.symm_lookup(s, c) == sys.symm.lk(s, c), "lookup");

    // HF string
    for (unsigned no = 4; no <= 32; no++) for (unsigned ne = 2; ne <= 2 * no && ne <= 24; ne += 2) {
        uint8_t b[8] = {0};
        gen_hf_bitstring(no, ne, b);
        CHECK(to_u64(b, CEILING(2 * no, 8)) == fo::gen_hf_det(no, ne), "hf det no=%u ne=%u", no, ne);
    }
    std::vector<uint32_t> scr(2 * n_orb);
    for (auto &x : scr) x = mt();
    HashTable<ssize_t> rht(16, scr);

    hb_info *rhb = set_up(n_orb, n_orb, eris);
    sys.hb.set_up(sys.ints);
    CHECK(same_bits(rhb->s_norm, sys.hb.s_norm), "s_norm");
    for (unsigned i = 0; i < n_orb; i++) {
        CHECK(same_bits(rhb->s_tens[i], sys.hb.s_tens[i]), "s_tens");
        CHECK(same_bits(rhb->diag_sqrt[i], sys.hb.diag_sqrt[i]), "diag_sqrt");
        CHECK(same_bits(rhb->exch_norms[i], sys.hb.exch_norms[i]), "exch_norms");
    }
    for (unsigned i = 0; i < n_orb * n_orb; i++) CHECK(same_bits(rhb->d_diff[i], sys.hb.d_diff[i]), "d_diff");
    for (unsigned i = 0; i < n_orb * (n_orb - 1) / 2; i++) { CHECK(same_bits(rhb->d_same[i], sys.hb.d_same[i]), "d_same"); CHECK(same_bits(rhb->exch_sqrt[i], sys.hb.exch_sqrt[i]), "exch_sqrt"); }

    for (int trial = 0; trial < 400; trial++) {
        fo::det_t d = rand_det(mt, n_orb, n_elec);
        uint8_t b[8] = {0}, occ_r[32], occ_o[32];
        from_u64(d, b, 8);
        CHECK(find_bits(b, occ_r, nb) == n_elec, "find_bits count");
        fo::occ_list(d, occ_o);
        CHECK(memcmp(occ_r, occ_o, n_elec) == 0, "occ list");
        CHECK(rht.hash_fxn(occ_r, n_elec, NULL, 0) == fo::hash_fxn(occ_o, n_elec, scr.data()), "hash_fxn");
        for (int k = 0; k < 20; k++) {
            unsigned a = mt() % (2 * n_orb), c = mt() % (2 * n_orb);
            if (a != c) CHECK(bits_between(b, a, c) == fo::bits_between(d, a, c), "bits_between %u %u", a, c);
        }
        CHECK(same_bits(diag_matrel(occ_r, n_orb, eris, h, 0, n_elec), fo::diag_matrel(occ_o, sys.ints, n_elec)), "diag_matrel");
        // excitation lists
        std::vector<uint8_t> so, dbo;
        static uint8_t sr[4096][2]; static uint8_t dr[200000][4];
        size_t ns_r = sing_ex_symm(b, occ_r, n_elec, n_orb, sr, symm);
        size_t ns_o = fo::sing_ex_symm(d, occ_o, n_elec, n_orb, so, symm);
        CHECK(ns_r == ns_o && memcmp(sr, so.data(), 2 * ns_r) == 0, "sing_ex_symm");
        CHECK(count_singex(b, occ_r, n_elec, &rsymm) == fo::count_singex(d, occ_o, n_elec, sys.symm), "count_singex");
        size_t nd_r = doub_ex_symm(b, occ_r, n_elec, n_orb, dr, symm);
        size_t nd_o = fo::doub_ex_symm(d, occ_o, n_elec, n_orb, dbo, symm);
        CHECK(nd_r == nd_o && memcmp(dr, dbo.data(), 4 * nd_r) == 0, "doub_ex_symm");
        for (size_t e = 0; e < ns_r; e++) {
            CHECK(same_bits(sing_matr_el_nosgn(sr[e], occ_r, n_orb, eris, h, 0, n_elec), fo::sing_matrel_nosgn(sr[e], occ_o, sys.ints, n_elec)), "sing matrel");
            uint8_t b2[8]; memcpy(b2, b, 8); fo::det_t d2 = d;
            CHECK(sing_det_parity(b2, sr[e]) == fo::sing_det_parity(&d2, sr[e]) && to_u64(b2, 8) == d2, "sing_det_parity");
            CHECK(sing_parity(b, sr[e]) == fo::sing_parity(d, sr[e]), "sing_parity");
        }
        for (size_t e = 0; e < nd_r; e += 7) {
            CHECK(same_bits(doub_matr_el_nosgn(dr[e], n_orb, eris, 0), fo::doub_matrel_nosgn(dr[e], sys.ints)), "doub matrel");
            uint8_t b2[8]; memcpy(b2, b, 8); fo::det_t d2 = d;
            CHECK(doub_det_parity(b2, dr[e]) == fo::doub_det_parity(&d2, dr[e]) && to_u64(b2, 8) == d2, "doub_det_parity");
            memcpy(b2, b, 8);
            CHECK(doub_parity(b2, dr[e]) == fo::doub_parity(d, dr[e]), "doub_parity");
            CHECK(same_bits(calc_unnorm_wt(rhb, dr[e]), fo::calc_unnorm_wt(sys.hb, dr[e])), "unnorm_wt");
            CHECK(same_bits(calc_norm_wt(rhb, dr[e], occ_r, n_elec, b, &rsymm), fo::calc_norm_wt(sys.hb, dr[e], occ_o, n_elec, d, sys.symm)), "norm_wt");
        }
        // symmetry counters
        unsigned cr[8][2], co[8][2];
        count_symm_virt(cr, occ_r, n_elec, &rsymm); fo::count_symm_virt(co, occ_o, n_elec, sys.symm);
        CHECK(memcmp(cr, co, sizeof(cr)) == 0, "count_symm_virt");
        CHECK(count_sing_allowed(occ_r, n_elec, symm, n_orb, cr) == fo::count_sing_allowed(occ_o, n_elec, sys.symm, co), "count_sing_allowed");
        for (uint8_t c = 0; c < n_elec; c++) {
            uint8_t c1 = c, c2 = c;
            CHECK(count_sing_virt(occ_r, n_elec, symm, n_orb, cr, &c1) == fo::count_sing_virt(occ_o, n_elec, sys.symm, co, &c2) && c1 == c2, "count_sing_virt");
        }
        for (unsigned ir = 0; ir < 8; ir++) for (unsigned ix = 0; ix < 4; ix++) for (unsigned sp = 0; sp < 2; sp++)
            CHECK(virt_from_idx(b, rsymm.symm_lookup[ir], n_orb * sp, ix) == fo::virt_from_idx(d, sys.symm, ir, n_orb * sp, ix), "virt_from_idx");
        for (unsigned sp = 0; sp < 2; sp++) for (unsigned k = 0; k < n_orb - n_elec / 2; k++)
            CHECK(find_nth_virt(occ_r, sp, n_elec, n_orb, k) == fo::find_nth_virt(occ_o, sp, n_elec, n_orb, k), "find_nth_virt");
        // probability rows
        double pr[64], po[64];
        for (int ex = 0; ex < 2; ex++) {
            memset(pr, 0, sizeof pr); memset(po, 0, sizeof po);
            double a = calc_o1_probs(rhb, pr, n_elec, occ_r, ex), c = fo::calc_o1_probs(sys.hb, po, n_elec, occ_o, ex);
            CHECK(same_bits(a, c) && memcmp(pr, po, sizeof pr) == 0, "o1_probs");
        }
        for (unsigned o1 = 0; o1 < n_elec; o1++) {
            memset(pr, 0, sizeof pr); memset(po, 0, sizeof po);
            double a = calc_o2_probs(rhb, pr, n_elec, occ_r, o1), c = fo::calc_o2_probs(sys.hb, po, n_elec, occ_o, o1);
            CHECK(same_bits(a, c) && memcmp(pr, po, sizeof pr) == 0, "o2_probs");
            if (o1 > 0) {
                memset(pr, 0, sizeof pr); memset(po, 0, sizeof po);
                a = calc_o2_probs_half(rhb, pr, n_elec, occ_r, o1); c = fo::calc_o2_probs_half(sys.hb, po, n_elec, occ_o, o1);
                CHECK(same_bits(a, c) && memcmp(pr, po, sizeof pr) == 0, "o2_probs_half");
            }
            for (int ex = 0; ex < 2; ex++) {
                memset(pr, 0, sizeof pr); memset(po, 0, sizeof po);
                a = calc_u1_probs(rhb, pr, occ_r[o1], occ_r, n_elec, ex); c = fo::calc_u1_probs(sys.hb, po, occ_o[o1], occ_o, n_elec, ex);
                CHECK(same_bits(a, c) && memcmp(pr, po, sizeof pr) == 0, "u1_probs o1=%u ex=%d", o1, ex);
            }
            for (unsigned o2 = 0; o2 < n_elec; o2++) if (o2 != o1) {
                unsigned u1 = find_nth_virt(occ_r, occ_r[o1] / n_orb, n_elec, n_orb, mt() % (n_orb - n_elec / 2));
                uint16_t lr = 0, lo = 0;
                memset(pr, 0, sizeof pr); memset(po, 0, sizeof po);
                a = calc_u2_probs(rhb, pr, occ_r[o1], occ_r[o2], u1, &rsymm, &lr); c = fo::calc_u2_probs(sys.hb, po, occ_o[o1], occ_o[o2], u1, sys.symm, &lo);
                CHECK(same_bits(a, c) && lr == lo && memcmp(pr, po, sizeof pr) == 0, "u2_probs");
                memset(pr, 0, sizeof pr); memset(po, 0, sizeof po);
                a = calc_u2_probs_half(rhb, pr, occ_r[o1], occ_r[o2], u1, b, &rsymm, &lr); c = fo::calc_u2_probs_half(sys.hb, po, occ_o[o1], occ_o[o2], u1, d, sys.symm, &lo);
                CHECK((same_bits(a, c) || (a != a && c != c)) && lr == lo && memcmp(pr, po, sizeof pr) == 0, "u2_probs_half");
            }
        }
    }
    // compression kernels on random vectors
    for (int trial = 0; trial < 60; trial++) {
        size_t n = 50 + mt() % 3000;
        std::vector<double> v(n), v2;
        for (auto &x : v) { double u = mt() / (1. + UINT32_MAX); x = (mt() & 1 ? 1 : -1) * exp(6 * u) * ((mt() % 7) ? 1 : 0); }
        v2 = v;
        unsigned ns_r = 1 + mt() % n, ns_o = ns_r;
        std::vector<size_t> srt_r(n), srt_o(n);
        std::vector<bool> keep_r(n, false); std::vector<uint8_t> keep_o(n, 0);
        double gn_r, gn_o;
        double ln_r = find_preserve(v.data(), srt_r, keep_r, n, &ns_r, &gn_r);
        double ln_o = fo::find_preserve(v2.data(), srt_o, keep_o, n, &ns_o, &gn_o);
        CHECK(same_bits(ln_r, ln_o) && same_bits(gn_r, gn_o) && ns_r == ns_o, "find_preserve scalars");
        for (size_t i = 0; i < n; i++) CHECK((bool)keep_r[i] == (bool)keep_o[i], "find_preserve keep");
        double rn = mt() / (1. + UINT32_MAX);
        double norms[1] = {ln_r};
        sys_comp(v.data(), n, norms, ns_r, keep_r, rn);
        double on = ln_o; fo::sys_comp(v2.data(), n, &on, ns_o, keep_o, rn);
        CHECK(same_bits(norms[0], on), "sys_comp norm");
        for (size_t i = 0; i < n; i++) CHECK(same_bits(v[i], v2[i]) && (bool)keep_r[i] == (bool)keep_o[i], "sys_comp out");
    }
    for (int trial = 0; trial < 60; trial++) {
        size_t n = 20 + mt() % 2000, ncol = 2 + mt() % 25;
        bool jag = mt() & 1;
        std::vector<double> v(n), wr_r(n), wr_o(n);
        std::vector<uint32_t> nd(n);
        std::vector<uint16_t> ss(n);
        Matrix<double> rsw(n, ncol); Matrix<bool> rk(n, ncol);
        fo::SubWts osw; osw.reshape(n, ncol);
        for (size_t i = 0; i < n; i++) {
            double u = mt() / (1. + UINT32_MAX);
            v[i] = (mt() % 9) ? exp(5 * u) : 0;
            nd[i] = (mt() % 3 == 0) ? 1 + mt() % 12 : 0;
            ss[i] = 1 + mt() % ncol;
            size_t lim = jag ? ss[i] : ncol;
            double tot = 0;
            for (size_t c = 0; c < ncol; c++) { double w = (c < lim && (mt() % 5)) ? mt() / (1. + UINT32_MAX) : 0; rsw(i, c) = w; tot += w; }
            if (tot == 0) { rsw(i, 0) = 1; tot = 1; }
            for (size_t c = 0; c < ncol; c++) { rsw(i, c) /= tot; osw.row(i)[c] = rsw(i, c); }
        }
        unsigned n_samp = 1 + mt() % (2 * n);
        double rn = mt() / (1. + UINT32_MAX);
        std::vector<double> nv_r(n * ncol + n * 12 + n_samp), nv_o(nv_r.size());
        std::vector<size_t> ni_r(2 * nv_r.size()), ni_o(2 * nv_r.size());
        size_t c_r = comp_sub(v.data(), n, nd.data(), rsw, rk, jag ? ss.data() : NULL, n_samp, wr_r.data(), rn, nv_r.data(), (size_t (*)[2])ni_r.data());
        size_t c_o = fo::comp_sub(v.data(), n, nd.data(), osw, jag ? ss.data() : nullptr, n_samp, wr_o.data(), rn, nv_o.data(), (size_t (*)[2])ni_o.data());
        CHECK(c_r == c_o, "comp_sub count %zu %zu", c_r, c_o);
        for (size_t i = 0; i < n; i++) CHECK(same_bits(wr_r[i], wr_o[i]), "wt_remain");
        for (size_t i = 0; i < std::min(c_r, c_o); i++) CHECK(same_bits(nv_r[i], nv_o[i]) && ni_r[2 * i] == ni_o[2 * i] && ni_r[2 * i + 1] == ni_o[2 * i + 1], "comp_sub out %zu", i);
    }
    printf("UNIT checks=%d fails=%d\n", n_chk, n_fail);
    return n_fail != 0;
}

// ------------------------------------------------------------------ pivotal compression (compress_utils.cpp:354-681)
static void piv_random_vec(std::mt19937 &mt, std::vector<double> &v, int style) {
    for (auto &x : v) {
        double u = mt() / (1. + UINT32_MAX);
        double a = style == 0 ? exp(6 * u) : (style == 1 ? u : exp(14 * u));
        x = (mt() & 1 ? 1 : -1) * a * ((mt() % 7) ? 1 : 0);
    }
}
static int run_piv(const char *out) {
    std::mt19937 mt(4321);
    // piv_comp_parallel on random vectors, the two generators starting from the same state
    for (int trial = 0; trial < 400; trial++) {
        size_t n = 30 + mt() % 4000;
        std::vector<double> v(n), v2;
        piv_random_vec(mt, v, trial % 3);
        v2 = v;
        uint32_t cs = 1 + mt() % (trial % 5 == 0 ? 2 * n : n);
        uint32_t seed = mt();
        std::mt19937 g_r(seed), g_o(seed);
        std::vector<size_t> srt_r(n), srt_o(n);
        std::vector<bool> k_r(n, false); std::vector<uint8_t> k_o(n, 0);
        piv_comp_parallel(v.data(), n, cs, srt_r, k_r, g_r);
        fo::piv_comp_parallel(v2.data(), n, cs, srt_o, k_o, g_o);
        size_t bad = 0, nz = 0;
        for (size_t i = 0; i < n; i++) { if (!same_bits(v[i], v2[i]) || (bool)k_r[i] != (bool)k_o[i]) bad++; if (v[i] != 0) nz++; }
        CHECK(bad == 0, "piv_comp_parallel trial %d: %zu mismatches", trial, bad);
        CHECK(nz <= cs, "piv_comp_parallel trial %d: %zu non-zeros for a budget of %u", trial, nz, cs);
        CHECK(g_r() == g_o(), "piv_comp_parallel generator state, trial %d", trial);
    }
    // piv_samp_serial alone: arbitrary preserved flags, budgets and segment norms (elements may exceed the sampling unit)
    for (int trial = 0; trial < 400; trial++) {
        size_t n = 10 + mt() % 3000;
        std::vector<double> v(n), v2;
        piv_random_vec(mt, v, trial % 3);
        std::vector<bool> k_r(n); std::vector<uint8_t> k_o(n);
        double norm = 0;
        for (size_t i = 0; i < n; i++) { bool k = mt() % 5 == 0; k_r[i] = k; k_o[i] = k; if (!k) norm += fabs(v[i]); }
        v2 = v;
        uint32_t ns = trial % 11 == 0 ? 0 : 1 + mt() % (n / 2);
        if (trial % 4 == 1) norm *= 0.5 + mt() / (1. + UINT32_MAX);
        uint32_t seed = mt();
        std::mt19937 g_r(seed), g_o(seed);
        piv_samp_serial(v.data(), n, norm, ns, k_r, g_r);
        fo::piv_samp_serial(v2.data(), n, norm, ns, k_o, g_o);
        size_t bad = 0;
        for (size_t i = 0; i < n; i++) if (!same_bits(v[i], v2[i]) || (bool)k_r[i] != (bool)k_o[i]) bad++;
        CHECK(bad == 0, "piv_samp_serial trial %d: %zu mismatches", trial, bad);
        CHECK(g_r() == g_o(), "piv_samp_serial generator state, trial %d", trial);
    }
    // adjust_probs with fractional local budgets, rounded up and down
    for (int trial = 0; trial < 400; trial++) {
        size_t n = 10 + mt() % 2000;
        std::vector<double> v(n), v2;
        piv_random_vec(mt, v, 1);
        std::vector<bool> k_r(n); std::vector<uint8_t> k_o(n);
        double norm = 0;
        for (size_t i = 0; i < n; i++) { bool k = mt() % 9 == 0; k_r[i] = k; k_o[i] = k; if (!k) norm += fabs(v[i]); }
        v2 = v;
        uint32_t tot = 2 + mt() % n;
        double share = 0.2 + 0.6 * (mt() / (1. + UINT32_MAX));
        double tot_norm = norm / share;
        double exp_loc = tot * norm / tot_norm;
        uint32_t nl_r = (uint32_t)exp_loc + (mt() & 1), nl_o = nl_r;
        double a = adjust_probs(v.data(), n, &nl_r, exp_loc, tot, tot_norm, k_r);
        double b = fo::adjust_probs(v2.data(), n, &nl_o, exp_loc, tot, tot_norm, k_o);
        size_t bad = 0;
        for (size_t i = 0; i < n; i++) if (!same_bits(v[i], v2[i]) || (bool)k_r[i] != (bool)k_o[i]) bad++;
        CHECK(bad == 0 && same_bits(a, b) && nl_r == nl_o, "adjust_probs trial %d: %zu mismatches, %a %a, %u %u", trial, bad, a, b, nl_r, nl_o);
    }
    // known answers for tests/golden
    FILE *f = fopen(out, "w");
    fprintf(f, "# piv_comp_parallel of the reference (compress_utils.cpp:354-386), one rank.  case <len> <compress_size> <mt19937 seed>, then per element:\n"
               "# <input> <output> <flag after>, then the generator's next draw.  Doubles are C99 hex floats.\n");
    const size_t lens[4] = {12, 300, 900, 1500};
    const uint32_t css[4] = {5, 40, 700, 333};
    for (int k = 0; k < 4; k++) {
        size_t n = lens[k];
        std::vector<double> v(n), in;
        piv_random_vec(mt, v, k % 2);
        in = v;
        uint32_t seed = 100 + k;
        std::mt19937 g(seed);
        std::vector<size_t> srt(n); std::vector<bool> kp(n, false);
        piv_comp_parallel(v.data(), n, css[k], srt, kp, g);
        fprintf(f, "case %zu %u %u\n", n, css[k], seed);
        for (size_t i = 0; i < n; i++) fprintf(f, "%a %a %d\n", in[i], v[i], (int)kp[i]);
        fprintf(f, "next %u\n", (unsigned)g());
    }
    fclose(f);
    printf("PIV checks=%d fails=%d\n", n_chk, n_fail);
    return n_fail != 0;
}

// ------------------------------------------------------------------ reference frisys loop (1 rank)
struct RefRun {
    fcidump_input *in;
    unsigned n_orb, n_elec;
    size_t det_size;
    SymmInfo *basis_symm;
    hb_info *hb;
    std::mt19937 mt;
    DistVec<double> *sol, *trial, *htrial;
    HBCompressSys *comp;
    std::vector<uintmax_t> trial_hashes, htrial_hashes;
    double p_doub, hf_en, en_shift = 0, last_one_norm = 0, eps, target, init_thresh;
    uint32_t vec_nonz, mat_nonz;
    int new_hb;
    std::vector<size_t> srt; std::vector<bool> keep;
    unsigned iterat = 0;
    int n_procs = 1, proc_rank = 0;
    unsigned hf_proc = 0;
    std::function<double(uint8_t *, uint8_t *)> sing_sc;
    std::function<double(uint8_t *)> doub_sc;
    std::vector<uint32_t> proc_scr, vec_scr;
    std::vector<fo::det_t> trial_in_det, ini_in_det; std::vector<double> trial_in_val, ini_in_val;      // what the text files held
    size_t max_dets_ = 0, adder_size_ = 0;
    size_t adder_size_override = 0;
    bool skip_init_dense = false;              // the restarted run of `reload`: frisys_mol --load_dir takes the dense space from the checkpoint (:234, :258)
    unsigned long long n_perform_add = 0;      // perform_add calls of the spawning loop (2 passes x (rounds + the empty closing one) per iteration)
    std::function<double(const uint8_t *)> diag_sc_;
    // --det_space (frisys_mol.cpp:236-239, 347-401): the dense subspace through the reference's own init_dense, and H inside it
    size_t n_determ = 0, determ_h_size = 0;
    unsigned n_determ_h = 0, tot_dense_h = 0;
    size_t *determ_from = nullptr; Matrix<uint8_t> *determ_to = nullptr; double *determ_matr_el = nullptr;
    std::vector<fo::det_t> det_space_in;
    // per-iteration record
    double numer, denom, glob_norm; unsigned nkept; size_t num_success;

    void setup(const char *path, const char *pg, uint32_t seed, double eps_, uint32_t vnz, uint32_t mnz, size_t max_dets, double ini, double tgt, int nhb) {
        in = parse_fcidump(path, pg);
        n_orb = in->n_orb_; n_elec = in->n_elec; det_size = CEILING(2 * n_orb, 8);
        eps = eps_; vec_nonz = vnz; mat_nonz = mnz; init_thresh = ini; target = tgt; new_hb = nhb;
        Matrix<double> *h_core = in->hcore; SymmERIs *eris = &in->eris;
        uint8_t tmp_orbs[64], hf_det[8] = {0};
        gen_hf_bitstring(n_orb, n_elec, hf_det);
        find_bits(hf_det, tmp_orbs, det_size);
        hf_en = diag_matrel(tmp_orbs, n_orb, *eris, *h_core, 0, n_elec);
        if (getenv("FRIES_HAM_SHIFT")) hf_en = atof(getenv("FRIES_HAM_SHIFT")) - in->core_en;      // --ham_shift (frisys_mol.cpp:95-98)
        mt.seed(seed);
        MPI_Comm_size(MPI_COMM_WORLD, &n_procs);
        MPI_Comm_rank(MPI_COMM_WORLD, &proc_rank);
        unsigned spawn_length = mat_nonz * 4 / n_procs;
        size_t adder_size = spawn_length > 1000000 ? 1000000 : spawn_length;
        if (adder_size_override) adder_size = adder_size_override;       // frifull_mol.cpp:68-70 sizes its Adder differently
        if (getenv("FRIES_ADDER_SIZE")) adder_size = (size_t)atol(getenv("FRIES_ADDER_SIZE"));      // a smaller Adder (the DistVec constructor's argument): the loop's early perform_add rounds at test sizes
        unsigned no = n_orb, ne = n_elec; double hfe = hf_en;
        std::function<double(const uint8_t *)> diag_sc = [no, eris, h_core, ne, hfe](const uint8_t *occ) { return diag_matrel(occ, no, *eris, *h_core, 0, ne) - hfe; };
        sing_sc = [no, eris, h_core, ne](uint8_t *ex, uint8_t *occ) { return sing_matr_el_nosgn(ex, occ, no, *eris, *h_core, 0, ne); };
        doub_sc = [no, eris](uint8_t *ex) { return doub_matr_el_nosgn(ex, no, *eris, 0); };
        basis_symm = new SymmInfo(in->symm, n_orb);
        proc_scr.resize(2 * n_orb); vec_scr.resize(2 * n_orb);
        for (auto &x : proc_scr) x = mt();
        for (auto &x : vec_scr) x = mt();
        sol = new DistVec<double>(max_dets, adder_size, n_orb * 2, n_elec, n_procs, diag_sc, 2, proc_scr, vec_scr);
        max_dets_ = max_dets; adder_size_ = adder_size; diag_sc_ = diag_sc;
        hf_proc = sol->idx_to_proc(hf_det);
        size_t n_states = n_elec > (n_orb - n_elec / 2) ? n_elec : n_orb - n_elec / 2;
        comp = new HBCompressSys(spawn_length, n_states);
        size_t n_ex = (size_t)n_orb * n_orb * n_elec * n_elec;
        // --trial_vec (frisys_mol.cpp:157-181): the reference's own text reader, through the solution vector's arrays as there
        size_t n_trial = 1;
        const char *trial_prefix = getenv("FRIES_TRIAL");
        Matrix<uint8_t> &load_dets = sol->indices();
        double *load_vals = (double *)sol->values();
        if (trial_prefix) n_trial = load_vec_txt(std::string(trial_prefix), load_dets, load_vals);
        unsigned tot_trial = sum_mpi((int)n_trial, proc_rank, n_procs);
        tot_trial = CEILING(tot_trial * 2, n_procs);
        trial = new DistVec<double>(tot_trial + 2, tot_trial + 2, n_orb * 2, n_elec, n_procs, proc_scr, vec_scr);
        htrial = new DistVec<double>((size_t)tot_trial * n_ex / n_procs + 2 * n_ex, (size_t)tot_trial * n_ex / n_procs + 2 * n_ex, n_orb * 2, n_elec, n_procs, diag_sc, 2, proc_scr, vec_scr);
        if (trial_prefix) {
            for (size_t i = 0; i < n_trial; i++) { trial->add(load_dets[i], load_vals[i], 1); htrial->add(load_dets[i], load_vals[i], 1); }
            trial_in_det.clear(); trial_in_val.clear();
            for (size_t i = 0; i < n_trial; i++) { trial_in_det.push_back(to_u64(load_dets[i], det_size)); trial_in_val.push_back(load_vals[i]); }
            bzero(load_vals, (n_trial + 1) * sizeof(double));
        }
        else if ((int)hf_proc == proc_rank) { trial->add(hf_det, 1, 1); htrial->add(hf_det, 1, 1); }
        trial->perform_add(0); htrial->perform_add(0);
        trial->collect_procs();
        trial_hashes.resize(trial->curr_size());
        for (size_t i = 0; i < trial->curr_size(); i++) trial_hashes[i] = sol->idx_to_hash(trial->indices()[i], tmp_orbs);
        std::vector<uint8_t> scratch(4 * n_ex);
        h_op_offdiag(*htrial, in->symm, n_orb, *eris, *h_core, scratch.data(), scratch.size(), 0, n_elec, 1, 1, 0);
        htrial->set_curr_vec_idx(0);
        h_op_diag(*htrial, 0, 0, 1);
        htrial->add_vecs(0, 1);
        htrial->collect_procs();
        htrial_hashes.resize(htrial->curr_size());
        for (size_t i = 0; i < htrial->curr_size(); i++) htrial_hashes[i] = sol->idx_to_hash(htrial->indices()[i], tmp_orbs);
        sol->gen_orb_list(hf_det, tmp_orbs);
        size_t n_hf_doub = doub_ex_symm(hf_det, tmp_orbs, n_elec, n_orb, (uint8_t (*)[4])scratch.data(), in->symm);
        size_t n_hf_sing = count_singex(hf_det, tmp_orbs, n_elec, basis_symm);
        p_doub = (double)n_hf_doub / (n_hf_sing + n_hf_doub);
        if (getenv("FRIES_DETSPACE") && !skip_init_dense) {  // --det_space (:236-239); not with --load_dir (:234)
            std::string dir = getenv("FRIES_DETSPACE_DIR") ? getenv("FRIES_DETSPACE_DIR") : "/tmp/";
            n_determ = sol->init_dense(std::string(getenv("FRIES_DETSPACE")), dir);
            for (size_t i = 0; i < n_determ; i++) det_space_in.push_back(to_u64(sol->indices()[i], det_size));
        }
        if (getenv("FRIES_INI")) {       // --ini_vec (:264-274)
            Matrix<uint8_t> ini_dets(sol->max_size(), det_size);
            double *lv = sol->values();
            size_t n_dets = load_vec_txt(std::string(getenv("FRIES_INI")), ini_dets, lv);
            ini_in_det.clear(); ini_in_val.clear();
            for (size_t i = 0; i < n_dets; i++) { ini_in_det.push_back(to_u64(ini_dets[i], det_size)); ini_in_val.push_back(lv[i]); }
            for (size_t i = 0; i < n_dets; i++) sol->add(ini_dets[i], lv[i], 1);
            n_dets++;
            bzero(lv, n_dets * sizeof(double));
        }
        else if ((int)hf_proc == proc_rank) sol->add(hf_det, 100, 1);
        sol->perform_add(0);
        hb = set_up(n_orb, n_orb, *eris);
        srt.resize(sol->max_size()); keep.assign(sol->max_size(), false);
        build_dense_h();
    }

    // H inside the dense space (:347-401), with the reference's allocation sizes and its scratch use of orb_indices1
    void build_dense_h() {
        Matrix<double> *h_core = in->hcore; SymmERIs *eris = &in->eris;
        determ_h_size = n_determ * n_elec * n_elec * (n_orb - n_elec / 2) * (n_orb - n_elec / 2);
        n_determ_h = 0;
        determ_from = (size_t *)malloc(determ_h_size * sizeof(size_t));
        determ_to = new Matrix<uint8_t>(determ_h_size, det_size);
        determ_matr_el = (double *)malloc(determ_h_size * sizeof(double));
        for (size_t det_idx = 0; det_idx < n_determ; det_idx++) {
            uint8_t *curr_det = sol->indices()[det_idx];
            uint8_t *occ_orbs = sol->orbs_at_pos(det_idx);
            uint8_t (*sing_ex)[2] = (uint8_t (*)[2])comp->orb_indices1;
            size_t n_sing = sing_ex_symm(curr_det, occ_orbs, n_elec, n_orb, sing_ex, in->symm);
            if (n_sing + n_determ_h > determ_h_size) { fprintf(stderr, "dense H larger than its allocation\n"); exit(3); }
            for (size_t e = 0; e < n_sing; e++) {
                double m = sing_matr_el_nosgn(sing_ex[e], occ_orbs, n_orb, *eris, *h_core, 0, n_elec);
                uint8_t *new_det = (*determ_to)[n_determ_h];
                memcpy(new_det, curr_det, det_size);
                m *= sing_det_parity(new_det, sing_ex[e]) * -eps;
                determ_from[n_determ_h] = det_idx; determ_matr_el[n_determ_h] = m; n_determ_h++;
            }
            uint8_t (*doub_ex)[4] = (uint8_t (*)[4])comp->orb_indices1;
            size_t n_doub = doub_ex_symm(curr_det, occ_orbs, n_elec, n_orb, doub_ex, in->symm);
            if (n_doub + n_determ_h > determ_h_size) { fprintf(stderr, "dense H larger than its allocation\n"); exit(3); }
            for (size_t e = 0; e < n_doub; e++) {
                double m = doub_matr_el_nosgn(doub_ex[e], n_orb, *eris, 0);
                uint8_t *new_det = (*determ_to)[n_determ_h];
                memcpy(new_det, curr_det, det_size);
                m *= doub_det_parity(new_det, doub_ex[e]) * -eps;
                determ_from[n_determ_h] = det_idx; determ_matr_el[n_determ_h] = m; n_determ_h++;
            }
        }
        tot_dense_h = sum_mpi((int)n_determ_h, proc_rank, n_procs);
    }

    // A fresh solution vector holding (dets, vals) in positions 0..n-1 -- what DistVec::load / fries_vec_load leave behind
    // (the old one's free-position stack would hand the positions out in reverse).  Every entry must belong to this rank.
    void replace_vector(const std::vector<uint64_t> &dets, const std::vector<double> &vals) {
        sol = new DistVec<double>(max_dets_, adder_size_, n_orb * 2, n_elec, n_procs, diag_sc_, 2, proc_scr, vec_scr);      // the old one is leaked, like the reference's own objects
        size_t i = 0;
        int more = 1;
        while (more) {
            while (i < dets.size()) {
                uint8_t d[8]; memcpy(d, &dets[i], 8);
                i++;
                if (!sol->add(d, vals[i - 1], 1)) break;
            }
            sol->perform_add(0);
            more = sum_mpi((int)(i < dets.size()), proc_rank, n_procs);
        }
        srt.resize(sol->max_size()); keep.assign(sol->max_size(), false);
    }
    uint64_t digest() const {
        uint64_t hsh = 1469598103934665603ull;
        for (size_t i = 0; i < sol->curr_size(); i++) {
            double rv = sol->values()[i];
            if (rv != 0) { uint64_t vb; memcpy(&vb, &rv, 8); fo::det_t rd = to_u64(sol->indices()[i], det_size); hsh = (hsh ^ rd) * 1099511628211ull; hsh = (hsh ^ vb) * 1099511628211ull; hsh = (hsh ^ i) * 1099511628211ull; }
        }
        return hsh;
    }

    void iterate() {
        DistVec<double> &sol_vec = *sol;
        HBCompressSys &cv = *comp;
        std::copy(sol_vec.values() + n_determ, sol_vec.values() + sol_vec.curr_size(), cv.vec1.begin());
        for (size_t i = n_determ; i < sol_vec.curr_size(); i++) cv.det_indices1[i - n_determ] = i;
        cv.vec_len = sol_vec.curr_size() - n_determ;
        apply_HBPP_sys(sol_vec.occ_orbs(), sol_vec.indices(), &cv, hb, basis_symm, p_doub, new_hb, mt, mat_nonz - tot_dense_h, sing_sc, doub_sc);
        size_t comp_len = cv.vec_len;
        num_success = comp_len;
        double *before = sol_vec.values();
        sol_vec.set_curr_vec_idx(1);
        sol_vec.zero_vec();
        size_t vec_size = sol_vec.curr_size();
        for (int add_ini = 0; add_ini < 2; add_ini++) {
            int num_added = 1;
            size_t s = 0;
            while (num_added > 0) {
                num_added = 0;
                while (s < comp_len) {
                    size_t d = cv.det_indices2[s];
                    double c = before[d];
                    uint8_t ini = fabs(c) >= init_thresh;
                    if (ini != add_ini) { s++; continue; }
                    uint8_t nd[8];
                    double add_el = -eps * cv.vec1[s];
                    if (c < 0) add_el *= -1;
                    memcpy(nd, sol_vec.indices()[d], det_size);
                    if (!(cv.orb_indices1[s][2] == 0 && cv.orb_indices1[s][3] == 0)) doub_det(nd, cv.orb_indices1[s]);
                    else sing_det(nd, cv.orb_indices1[s]);
                    num_added++; s++;
                    if (!sol_vec.add(nd, add_el, ini)) break;
                }
                sol_vec.perform_add(0);
                n_perform_add++;
                sol_vec.set_curr_vec_idx(0);
                before = sol_vec.values();
                sol_vec.set_curr_vec_idx(1);
                num_added = sum_mpi(num_added, proc_rank, n_procs);
            }
        }
        if (sol_vec.max_size() > srt.size()) { srt.resize(sol_vec.max_size()); keep.resize(sol_vec.max_size(), false); }
        // the dense block of H (:480-485), with the reference's loop bound: the ALLOCATED size.  The entries beyond the n_determ_h
        // filled ones are whatever malloc returned -- zero pages for an allocation of this size -- and add() drops zero values.
        for (size_t k = 0; k < determ_h_size; k++) {
            size_t d = determ_from[k];
            double mat_vec = before[d] * determ_matr_el[k];
            sol_vec.add((*determ_to)[k], mat_vec, 1);
        }
        sol_vec.perform_add(0);
        sol_vec.set_curr_vec_idx(0);
        for (size_t i = 0; i < vec_size; i++) {
            double *c = sol_vec[i];
            if (*c != 0) { double de = sol_vec.matr_el_at_pos(i); *c *= 1 - eps * (de - en_shift); }
        }
        sol_vec.add_vecs(0, 1);
        sol_vec.set_curr_vec_idx(1); sol_vec.zero_vec(); sol_vec.set_curr_vec_idx(0);
        unsigned n_samp = vec_nonz;
        double loc_norms[64];
        loc_norms[proc_rank] = find_preserve(&(sol_vec.values()[n_determ]), srt, keep, sol_vec.curr_size() - n_determ, &n_samp, &glob_norm);
        glob_norm += sol_vec.dense_norm();
        nkept = vec_nonz - n_samp;
        if ((iterat + 1) % 10 == 0) adjust_shift(&en_shift, glob_norm, &last_one_norm, target, 0.05 / 10 / eps);
        numer = sol_vec.dot(htrial->indices(), htrial->values(), htrial->curr_size(), htrial_hashes);
        denom = sol_vec.dot(trial->indices(), trial->values(), trial->curr_size(), trial_hashes);
        numer = sum_mpi(numer, proc_rank, n_procs);
        denom = sum_mpi(denom, proc_rank, n_procs);
        double rn_sys = 0;
        if (proc_rank == 0) rn_sys = mt() / (1. + UINT32_MAX);
        MPI_Allgather(MPI_IN_PLACE, 0, MPI_DOUBLE, loc_norms, 1, MPI_DOUBLE, MPI_COMM_WORLD);
        sys_comp(&(sol_vec.values()[n_determ]), sol_vec.curr_size() - n_determ, loc_norms, n_samp, keep, rn_sys);
        for (size_t i = 0; i < sol_vec.curr_size() - n_determ; i++) if (keep[i]) { sol_vec.del_at_pos(i + n_determ); keep[i] = 0; }
        iterat++;
    }
};

// ------------------------------------------------------------------ reference frifull_mol loop (FRIES_bin/frifull_mol.cpp:258-304)
struct RefFull {
    RefRun rr;
    int vec_idx = 0;
    std::vector<uint8_t> scratch;
    double numer = 0, denom = 0, glob_norm = 0; unsigned nkept = 0;
    void setup(const char *path, const char *pg, uint32_t seed, double eps, uint32_t vnz, size_t max_dets, double tgt, unsigned n_orb_hint = 0, unsigned n_elec_hint = 0) {
        if (n_orb_hint) {      // frifull_mol.cpp:68-70: num_ex = n_elec^2 (n_orb - n_elec / 2)^2, spawn_len = target_nonz / n_procs * num_ex / n_procs / 4
            int np = 1; MPI_Comm_size(MPI_COMM_WORLD, &np);
            const unsigned num_ex = n_elec_hint * n_elec_hint * (n_orb_hint - n_elec_hint / 2) * (n_orb_hint - n_elec_hint / 2);
            const unsigned spawn_len = vnz / np * num_ex / np / 4;
            rr.adder_size_override = spawn_len > 1000000 ? 1000000 : spawn_len;
        }
        rr.setup(path, pg, seed, eps, vnz, vnz, max_dets, 0.0, tgt, 1);
        scratch.resize(4 * (size_t)rr.n_orb * rr.n_orb * rr.n_elec * rr.n_elec);
    }
    void iterate() {
        DistVec<double> &sol_vec = *rr.sol;
        const double eps = rr.eps;
        denom = sol_vec.dot(rr.trial->indices(), rr.trial->values(), rr.trial->curr_size(), rr.trial_hashes);
        denom = sum_mpi(denom, rr.proc_rank, rr.n_procs);
        unsigned n_samp = rr.vec_nonz;
        double loc_norms[64];
        if (sol_vec.max_size() > rr.srt.size()) { rr.srt.resize(sol_vec.max_size()); rr.keep.resize(sol_vec.max_size(), false); }
        loc_norms[rr.proc_rank] = find_preserve(sol_vec.values(), rr.srt, rr.keep, sol_vec.curr_size(), &n_samp, &glob_norm);
        MPI_Allgather(MPI_IN_PLACE, 0, MPI_DOUBLE, loc_norms, 1, MPI_DOUBLE, MPI_COMM_WORLD);       // :264
        nkept = rr.vec_nonz - n_samp;
        if ((rr.iterat + 1) % 10 == 0) adjust_shift(&rr.en_shift, glob_norm, &rr.last_one_norm, rr.target, 0.05 / 10 / eps);
        double rn_sys = 0;
        if (rr.proc_rank == 0) rn_sys = rr.mt() / (1. + UINT32_MAX);       // :279-281 (sys_comp broadcasts rank 0's)
        sys_comp(sol_vec.values(), sol_vec.curr_size(), loc_norms, n_samp, rr.keep, rn_sys);
        for (size_t i = 0; i < sol_vec.curr_size(); i++) if (rr.keep[i]) { sol_vec.del_at_pos(i); rr.keep[i] = 0; }
        h_op_diag(sol_vec, !vec_idx, 1 + eps * rr.en_shift, -eps);
        sol_vec.set_curr_vec_idx(vec_idx);
        h_op_offdiag(sol_vec, rr.in->symm, rr.n_orb, rr.in->eris, *rr.in->hcore, scratch.data(), scratch.size(), 0, rr.n_elec, !vec_idx, -eps, 0);
        vec_idx = !vec_idx;
        numer = sol_vec.dot(rr.trial->indices(), rr.trial->values(), rr.trial->curr_size(), rr.trial_hashes);
        numer = sum_mpi(numer, rr.proc_rank, rr.n_procs);
        numer = ((1 + eps * rr.en_shift) * denom - numer) / eps;
        rr.iterat++;
    }
};

static int run_frifull(int argc, char **argv) {
    if (argc < 11) { fprintf(stderr, "usage: frifull <fcidump> <pg> <n_iter> <seed> <eps> <vec_nonz> <max_dets> <target> <out>\n"); return 2; }
    const char *path = argv[2], *pg = argv[3];
    unsigned n_iter = atoi(argv[4]); uint32_t seed = strtoul(argv[5], 0, 10);
    double eps = atof(argv[6]); uint32_t vnz = strtoul(argv[7], 0, 10);
    size_t max_dets = strtoull(argv[8], 0, 10); double tgt = atof(argv[9]);
    const char *out = argv[10];
    RefFull rf;
    rf.setup(path, pg, seed, eps, vnz, max_dets, tgt);
    RefRun &rr = rf.rr;
    fo::Frifull fr;
    fr.sys.n_orb = rr.n_orb; fr.sys.n_elec = rr.n_elec;
    fill_oracle_ints(fr.sys.ints, rr.in->eris, *rr.in->hcore, rr.n_orb);
    fr.sys.symm.init(rr.in->symm, rr.n_orb);
    fr.par.eps = eps; fr.par.target_norm = tgt; fr.par.vec_nonz = vnz; fr.par.max_dets = max_dets; fr.par.seed = seed;
    fr.setup();
    FILE *f = fopen(out, "w");
    fprintf(f, "# golden trajectory from the reference (frifull_mol.cpp loop, 1 rank, symmetry-packed integrals); cols: it numer denom norm shift nkept n_nonz curr_size n_add hash (doubles as C99 hex floats)\n");
    fprintf(f, "# args:");
    for (int i = 2; i < 10; i++) fprintf(f, " %s", i == 2 ? "<fcidump>" : argv[i]);
    fprintf(f, "\n# p_doub %a hf_en %a n_htrial %zu\n", 0.0, rr.hf_en, (size_t)0);
    for (unsigned it = 0; it < n_iter; it++) {
        rf.iterate();
        fr.iterate(1);
        const fo::IterLog &lg = fr.log.back();
        CHECK(same_bits(lg.numer, rf.numer) && same_bits(lg.denom, rf.denom), "it %u numer/denom %a %a | %a %a", it, lg.numer, rf.numer, lg.denom, rf.denom);
        CHECK(same_bits(lg.norm, rf.glob_norm) && same_bits(lg.shift, rr.en_shift), "it %u norm/shift", it);
        CHECK(lg.nkept == rf.nkept && lg.n_nonz == rr.sol->n_nonz() && lg.curr_size == rr.sol->curr_size(), "it %u counts nkept %u/%u nnz %d/%d size %zu/%zu", it, lg.nkept, rf.nkept, lg.n_nonz, rr.sol->n_nonz(), lg.curr_size, (size_t)rr.sol->curr_size());
        size_t n = std::min(lg.curr_size, (size_t)rr.sol->curr_size());
        size_t bad = 0;
        uint64_t hsh = 1469598103934665603ull;
        for (size_t i = 0; i < n; i++) {
            double rv = rr.sol->values()[i];          // the column the next iteration starts from
            fo::det_t rd = to_u64(rr.sol->indices()[i], rr.det_size);
            if (!same_bits(rv, fr.sol.vals[fr.vec_idx][i])) bad++;
            if (rv != 0 && rd != fr.sol.dets[i]) bad++;
            if (rv != 0) { uint64_t vb; memcpy(&vb, &rv, 8); hsh = (hsh ^ rd) * 1099511628211ull; hsh = (hsh ^ vb) * 1099511628211ull; hsh = (hsh ^ i) * 1099511628211ull; }
        }
        CHECK(bad == 0, "it %u vector mismatch in %zu slots", it, bad);
        fprintf(f, "%u %a %a %a %a %u %d %zu %zu %016" PRIx64 "\n", it, rf.numer, rf.denom, rf.glob_norm, rr.en_shift, rf.nkept, rr.sol->n_nonz(), (size_t)rr.sol->curr_size(), lg.num_success, hsh);
    }
    fclose(f);
    printf("FRIFULL iters=%u checks=%d fails=%d final n_nonz=%d\n", n_iter, n_chk, n_fail, rr.sol->n_nonz());
    return n_fail != 0;
}

// mpiexec -n P ref_harness frifull_mpi <fcidump> <pg> <n_iter> <seed> <eps> <vec_nonz> <max_dets> <target> <n_orb> <n_elec> <out>: the reference's frifull_mol loop
// on P ranks with frifull_mol.cpp's Adder size; every rank writes <out>.r<rank> (rows as in the frisys_mpi files, num_success = 0)
static int run_frifull_mpi(int argc, char **argv) {
    if (argc < 13) { fprintf(stderr, "usage: see header\n"); return 2; }
    const char *path = argv[2], *pg = argv[3];
    unsigned n_iter = atoi(argv[4]); uint32_t seed = strtoul(argv[5], 0, 10);
    double eps = atof(argv[6]); uint32_t vnz = strtoul(argv[7], 0, 10);
    size_t max_dets = strtoull(argv[8], 0, 10); double tgt = atof(argv[9]);
    RefFull rf;
    rf.setup(path, pg, seed, eps, vnz, max_dets, tgt, atoi(argv[10]), atoi(argv[11]));
    RefRun &rr = rf.rr;
    char name[1024];
    snprintf(name, sizeof name, "%s.r%d", argv[12], rr.proc_rank);
    FILE *f = fopen(name, "w");
    fprintf(f, "# golden trajectory from the reference's frifull_mol loop under mpiexec -n %d, rank %d; cols: it numer denom norm shift nkept n_nonz curr_size num_success digest\n", rr.n_procs, rr.proc_rank);
    fprintf(f, "# p_doub %a hf_en %a n_htrial %zu hf_proc %u\n", 0.0, rr.hf_en, (size_t)0, rr.hf_proc);
    for (unsigned it = 0; it < n_iter; it++) {
        rf.iterate();
        fprintf(f, "%u %a %a %a %a %u %d %zu %zu %016" PRIx64 "\n", it, rf.numer, rf.denom, rf.glob_norm, rr.en_shift, rf.nkept, rr.sol->n_nonz(), (size_t)rr.sol->curr_size(), (size_t)0, rr.digest());
    }
    fclose(f);
    if (rr.proc_rank == 0) printf("FRIFULL_MPI ranks=%d iters=%u\n", rr.n_procs, n_iter);
    return 0;
}

static void setup_oracle_from_ref(fo::Frisys &fr, RefRun &rr, uint32_t seed, size_t max_dets) {
    fr.sys.n_orb = rr.n_orb; fr.sys.n_elec = rr.n_elec;
    fill_oracle_ints(fr.sys.ints, rr.in->eris, *rr.in->hcore, rr.n_orb);
    fr.sys.symm.init(rr.in->symm, rr.n_orb);
    fr.par.eps = rr.eps; fr.par.target_norm = rr.target; fr.par.init_thresh = rr.init_thresh;
    fr.par.vec_nonz = rr.vec_nonz; fr.par.mat_nonz = rr.mat_nonz; fr.par.max_dets = max_dets;
    fr.par.new_hb = rr.new_hb; fr.par.seed = seed;
    fr.trial_in_det = rr.trial_in_det; fr.trial_in_val = rr.trial_in_val;
    fr.ini_det = rr.ini_in_det; fr.ini_val = rr.ini_in_val;
    if (getenv("FRIES_HAM_SHIFT")) { fr.has_ham_shift = true; fr.ham_shift = rr.hf_en; }
    fr.det_space = rr.det_space_in;
    fr.setup();
}

static int run_frisys(int argc, char **argv, bool time_only) {
    if (argc < 14) { fprintf(stderr, "usage: see header\n"); return 2; }
    const char *path = argv[2], *pg = argv[3];
    unsigned n_iter = atoi(argv[4]); uint32_t seed = strtoul(argv[5], 0, 10);
    double eps = atof(argv[6]); uint32_t vnz = strtoul(argv[7], 0, 10), mnz = strtoul(argv[8], 0, 10);
    size_t max_dets = strtoull(argv[9], 0, 10); double ini = atof(argv[10]), tgt = atof(argv[11]);
    int nhb = !strcmp(argv[12], "HB_unnorm");
    const char *out = argv[13];
    unsigned snap_every = argc > 14 ? atoi(argv[14]) : 0;
    RefRun rr;
    rr.setup(path, pg, seed, eps, vnz, mnz, max_dets, ini, tgt, nhb);
    if (time_only) {
        unsigned warm = argc > 14 ? atoi(argv[14]) : 0;
        for (unsigned i = 0; i < warm; i++) rr.iterate();
        auto t0 = std::chrono::steady_clock::now();
        size_t spawns = 0;
        for (unsigned i = 0; i < n_iter; i++) { rr.iterate(); spawns += rr.num_success; }
        double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("{\"kind\": \"reference\", \"iters\": %u, \"seconds\": %.6f, \"iters_per_s\": %.6f, \"spawns_per_s\": %.3f, \"n_nonz\": %d}\n", n_iter, s, n_iter / s, spawns / s, rr.sol->n_nonz());
        return 0;
    }
    fo::Frisys fr;
    setup_oracle_from_ref(fr, rr, seed, max_dets);
    CHECK(same_bits(fr.p_doub, rr.p_doub), "p_doub");
    CHECK(same_bits(fr.sys.hf_en, rr.hf_en), "hf_en");
    CHECK(fr.htrial_det.size() == rr.htrial->curr_size(), "htrial size");
    for (size_t i = 0; i < fr.htrial_det.size() && i < rr.htrial->curr_size(); i++)
        CHECK(fr.htrial_det[i] == to_u64(rr.htrial->indices()[i], rr.det_size) && same_bits(fr.htrial_val[i], rr.htrial->values()[i]), "htrial el %zu", i);
    FILE *f = fopen(out, "w");
    fprintf(f, "# golden trajectory from the reference (frisys_mol.cpp loop, 1 rank); cols: it numer denom norm shift nkept n_nonz curr_size num_success (doubles as C99 hex floats)\n");
    fprintf(f, "# args:");
    for (int i = 2; i < 13; i++) fprintf(f, " %s", i == 2 ? "<fcidump>" : argv[i]);
    fprintf(f, "\n# p_doub %a hf_en %a n_htrial %zu\n", rr.p_doub, rr.hf_en, (size_t)rr.htrial->curr_size());
    for (unsigned it = 0; it < n_iter; it++) {
        rr.iterate();
        fr.iterate(1);
        const fo::IterLog &lg = fr.log.back();
        CHECK(same_bits(lg.numer, rr.numer) && same_bits(lg.denom, rr.denom), "it %u numer/denom %a %a | %a %a", it, lg.numer, rr.numer, lg.denom, rr.denom);
        CHECK(same_bits(lg.norm, rr.glob_norm) && same_bits(lg.shift, rr.en_shift), "it %u norm/shift", it);
        CHECK(lg.nkept == rr.nkept && lg.n_nonz == rr.sol->n_nonz() && lg.curr_size == rr.sol->curr_size() && lg.num_success == rr.num_success, "it %u counts nkept %u/%u nnz %d/%d size %zu/%zu succ %zu/%zu", it, lg.nkept, rr.nkept, lg.n_nonz, rr.sol->n_nonz(), lg.curr_size, (size_t)rr.sol->curr_size(), lg.num_success, rr.num_success);
        size_t n = std::min(lg.curr_size, (size_t)rr.sol->curr_size());
        size_t bad = 0;
        uint64_t hsh = 1469598103934665603ull;
        for (size_t i = 0; i < n; i++) {
            double rv = rr.sol->values()[i];
            fo::det_t rd = to_u64(rr.sol->indices()[i], rr.det_size);
            if (!same_bits(rv, fr.sol.vals[0][i])) bad++;
            if (rv != 0 && rd != fr.sol.dets[i]) bad++;
            if (rv != 0) { uint64_t vb; memcpy(&vb, &rv, 8); hsh = (hsh ^ rd) * 1099511628211ull; hsh = (hsh ^ vb) * 1099511628211ull; hsh = (hsh ^ i) * 1099511628211ull; }
        }
        CHECK(bad == 0, "it %u vector mismatch in %zu slots", it, bad);
        fprintf(f, "%u %a %a %a %a %u %d %zu %zu %016" PRIx64 "\n", it, rr.numer, rr.denom, rr.glob_norm, rr.en_shift, rr.nkept, rr.sol->n_nonz(), (size_t)rr.sol->curr_size(), rr.num_success, hsh);
        if (getenv("FRIES_SAVE_DIR") && getenv("FRIES_SAVE_AT") && it + 1 == (unsigned)atoi(getenv("FRIES_SAVE_AT")))
            rr.sol->save(std::string(getenv("FRIES_SAVE_DIR")));       // the reference's own checkpoint writer (vec_utils.hpp:713-746)
        if (snap_every && ((it + 1) % snap_every == 0 || it + 1 == n_iter)) {
            fprintf(f, "SNAP %u %zu\n", it, (size_t)rr.sol->curr_size());
            for (size_t i = 0; i < rr.sol->curr_size(); i++) {
                double rv = rr.sol->values()[i];
                if (rv != 0) fprintf(f, "%zu %016" PRIx64 " %a\n", i, (uint64_t)to_u64(rr.sol->indices()[i], rr.det_size), rv);
            }
            fprintf(f, "ENDSNAP\n");
        }
    }
    fclose(f);
    printf("FRISYS iters=%u checks=%d fails=%d final n_nonz=%d\n", n_iter, n_chk, n_fail, rr.sol->n_nonz());
    return n_fail != 0;
}

// ------------------------------------------------------------------ [new_hb_all]
static int run_hbpp_all(const char *out) {
    std::mt19937 mt_obj(0);
    uint32_t n_orb = 22, n_frz = 2, n_elec = 10, n_elec_unf = n_elec - n_frz, tot_orb = n_orb + n_frz / 2;
    size_t det_size = CEILING(2 * n_orb, 8);
    FourDArr eris(tot_orb, tot_orb, tot_orb, tot_orb);
    for (size_t i = 0; i < tot_orb; i++) for (size_t j = 0; j < tot_orb; j++) for (size_t a = 0; a < tot_orb; a++) for (size_t b = 0; b < tot_orb; b++)
        eris(i, j, a, b) = mt_obj() / (1. + UINT32_MAX) - 0.5;
    hb_info *hbtens = set_up(tot_orb, n_orb, eris);
    Matrix<uint8_t> dets(1, det_size), occ_orbs(1, n_elec_unf);
    gen_hf_bitstring(n_orb, n_elec_unf, dets[0]);
    find_bits(dets[0], occ_orbs[0], det_size);
    uint8_t symm[] = {0, 5, 6, 7, 0, 5, 6, 7, 0, 0, 1, 2, 3, 5, 6, 7, 0, 0, 0, 1, 2, 3};
    SymmInfo basis_symm(symm, n_orb);
    uint32_t n_ex = n_orb * n_orb * n_elec_unf * n_elec_unf;
    size_t n_states = n_elec_unf > (n_orb - n_elec_unf / 2) ? n_elec_unf : n_orb - n_elec_unf / 2;
    HBCompressSys sv(n_ex, n_states);
    sv.vec_len = 1; sv.det_indices1[0] = 0; sv.vec1[0] = 1;
    std::function<double(uint8_t *, uint8_t *)> s1 = [](uint8_t *, uint8_t *) { return 1; };
    std::function<double(uint8_t *)> d1 = [](uint8_t *) { return 1; };
    std::mt19937 mt_copy = mt_obj;
    apply_HBPP_sys(occ_orbs, dets, &sv, hbtens, &basis_symm, 0.95, true, mt_obj, n_ex, s1, d1);

    // oracle side: copy the tensors the reference built from the non-symmetric FourDArr
    fo::MolSys sys; sys.n_orb = n_orb; sys.n_elec = n_elec_unf;
    sys.symm.init(symm, n_orb);
    fo::HBInfo &hb = sys.hb; hb.n_orb = n_orb; hb.s_norm = hbtens->s_norm;
    hb.s_tens.assign(hbtens->s_tens, hbtens->s_tens + n_orb);
    hb.d_diff.assign(hbtens->d_diff, hbtens->d_diff + n_orb * n_orb);
    hb.d_same.assign(hbtens->d_same, hbtens->d_same + n_orb * (n_orb - 1) / 2);
    hb.exch_sqrt.assign(hbtens->exch_sqrt, hbtens->exch_sqrt + n_orb * (n_orb - 1) / 2);
    hb.diag_sqrt.assign(hbtens->diag_sqrt, hbtens->diag_sqrt + n_orb);
    hb.exch_norms.assign(hbtens->exch_norms, hbtens->exch_norms + n_orb);
    fo::Vec v; v.init(4, 4, n_elec_unf, 1);
    v.add(fo::gen_hf_det(n_orb, n_elec_unf), 1, 1); v.perform_add(0);
    fo::HBScratch sc; sc.init(n_ex, n_states);
    sc.vec_len = 1; sc.det_idx1[0] = 0; sc.vec1[0] = 1;
    double rn[5];
    for (int k = 0; k < 5; k++) rn[k] = mt_copy() / (1. + UINT32_MAX);
    fo::apply_HBPP_sys(v, sc, sys, 0.95, true, rn, n_ex, true);
    CHECK(sc.vec_len == sv.vec_len, "hbpp_all len %zu %zu", sc.vec_len, (size_t)sv.vec_len);
    FILE *f = fopen(out, "w");
    fprintf(f, "# [new_hb_all] tests/test_hamiltonian.cpp:454-520 through the reference: rn[5], then o1 o2 u1 u2 value per sample\n");
    fprintf(f, "RN %a %a %a %a %a\nN %zu\n", rn[0], rn[1], rn[2], rn[3], rn[4], (size_t)sv.vec_len);
    size_t bad1 = 0;
    for (size_t s = 0; s < sv.vec_len; s++) {
        if (fabs(fabs(sv.vec1[s]) - 1) > 1e-7) bad1++;
        CHECK(memcmp(sv.orb_indices1[s], &sc.orb1[4 * s], 4) == 0 && same_bits(sv.vec1[s], sc.vec1[s]) && sv.det_indices2[s] == sc.det_idx2[s], "hbpp_all sample %zu", s);
        fprintf(f, "%u %u %u %u %a\n", sv.orb_indices1[s][0], sv.orb_indices1[s][1], sv.orb_indices1[s][2], sv.orb_indices1[s][3], sv.vec1[s]);
    }
    CHECK(bad1 == 0, "|value| != 1 in %zu samples", bad1);
    // second half of [new_hb_all] (:507-519): the pivotal variant with the same budget returns the same excitations
    {
        HBCompressPiv pv(n_ex, n_states);
        pv.vec_len = 1; pv.det_indices1[0] = 0; pv.vec1[0] = 1;
        std::mt19937 mt_piv = mt_obj;
        apply_HBPP_piv(occ_orbs, dets, &pv, hbtens, &basis_symm, 0.95, true, mt_obj, n_ex, s1, d1, 0);
        fo::HBPivScratch ps; ps.init(n_ex, n_states);
        ps.vec_len = 1; ps.det_idx1[0] = 0; ps.vec1[0] = 1;
        fo::apply_HBPP_piv(v, ps, sys, 0.95, true, mt_piv, n_ex, true);
        CHECK(ps.vec_len == pv.vec_len && pv.vec_len == sv.vec_len, "hbpp_all piv len %zu %zu %zu", ps.vec_len, (size_t)pv.vec_len, (size_t)sv.vec_len);
        CHECK(mt_piv() == mt_obj(), "hbpp_all piv generator state");
        size_t badp = 0;
        for (size_t s = 0; s < pv.vec_len && s < sv.vec_len; s++) {
            if (fabs(fabs(pv.vec1[s]) - 1) > 1e-7 || memcmp(pv.orb_indices1[s], sv.orb_indices1[s], 4)) badp++;
            CHECK(memcmp(pv.orb_indices1[s], &ps.orb1[4 * s], 4) == 0 && same_bits(pv.vec1[s], ps.vec1[s]) && pv.det_indices2[s] == ps.det_idx2[s], "hbpp_all piv sample %zu", s);
        }
        CHECK(badp == 0, "[new_hb_all] piv vs sys: %zu samples differ", badp);
        fprintf(f, "PIV %zu\n", (size_t)pv.vec_len);
        for (size_t s = 0; s < pv.vec_len; s++)
            fprintf(f, "%u %u %u %u %a\n", pv.orb_indices1[s][0], pv.orb_indices1[s][1], pv.orb_indices1[s][2], pv.orb_indices1[s][3], pv.vec1[s]);
    }
    // tensors, so the GPU test can run the same case without the FourDArr
    fprintf(f, "HB %a\n", hb.s_norm);
    auto dump = [&](const char *nm, const std::vector<double> &x) { fprintf(f, "%s %zu", nm, x.size()); for (double y : x) fprintf(f, " %a", y); fprintf(f, "\n"); };
    dump("s_tens", hb.s_tens); dump("d_diff", hb.d_diff); dump("d_same", hb.d_same); dump("exch_sqrt", hb.exch_sqrt); dump("diag_sqrt", hb.diag_sqrt); dump("exch_norms", hb.exch_norms);
    fclose(f);
    printf("HBPP_ALL n=%zu checks=%d fails=%d\n", (size_t)sv.vec_len, n_chk, n_fail);
    return n_fail != 0;
}

// ------------------------------------------------------------------ apply_HBPP_piv on a populated vector
// hbpiv <fcidump> <pg> <n_iter> <seed> <eps> <vec_nonz> <mat_nonz> <max_dets> <ini> <target> <dist> <out> <n_samp> <piv_seed> [<n_samp> <piv_seed> ...]
// n_iter iterations of the reference's frisys_mol loop (oracle in lockstep), then for every (n_samp, piv_seed) the reference's
// apply_HBPP_piv (heat_bathPP.cpp:1014-1419, spin_parity 0) and the restatement on that vector with generators seeded alike.
static int run_hbpiv(int argc, char **argv) {
    if (argc < 16) { fprintf(stderr, "usage: see header\n"); return 2; }
    const char *path = argv[2], *pg = argv[3];
    unsigned n_iter = atoi(argv[4]); uint32_t seed = strtoul(argv[5], 0, 10);
    double eps = atof(argv[6]); uint32_t vnz = strtoul(argv[7], 0, 10), mnz = strtoul(argv[8], 0, 10);
    size_t max_dets = strtoull(argv[9], 0, 10); double ini = atof(argv[10]), tgt = atof(argv[11]);
    int nhb = !strcmp(argv[12], "HB_unnorm");
    const char *out = argv[13];
    RefRun rr;
    rr.setup(path, pg, seed, eps, vnz, mnz, max_dets, ini, tgt, nhb);
    fo::Frisys fr;
    setup_oracle_from_ref(fr, rr, seed, max_dets);
    for (unsigned it = 0; it < n_iter; it++) { rr.iterate(); fr.iterate(1); }
    size_t n = rr.sol->curr_size();
    CHECK(n == fr.sol.curr_size, "hbpiv vector sizes");
    for (size_t i = 0; i < n; i++) CHECK(same_bits(rr.sol->values()[i], fr.sol.vals[0][i]), "hbpiv vector el %zu", i);
    FILE *f = fopen(out, "w");
    fprintf(f, "# apply_HBPP_piv through the reference after %u frisys_mol iterations; per case: CASE n_samp piv_seed n_out stage lengths, then pos o1 o2 u1 u2 value\n", n_iter);
    const size_t n_states = rr.n_elec > (rr.n_orb - rr.n_elec / 2) ? rr.n_elec : rr.n_orb - rr.n_elec / 2;
    for (int a = 14; a + 1 < argc; a += 2) {
        uint32_t n_samp = strtoul(argv[a], 0, 10), pseed = strtoul(argv[a + 1], 0, 10);
        size_t len = (n > (size_t)n_samp ? n : (size_t)n_samp) * 2 + 64;
        HBCompressPiv pv(len, n_states);
        std::copy(rr.sol->values(), rr.sol->values() + n, pv.vec1.begin());
        for (size_t i = 0; i < n; i++) pv.det_indices1[i] = i;
        pv.vec_len = n;
        std::mt19937 m1(pseed), m2(pseed);
        const int spin_parity = getenv("FRIES_SPIN_PARITY") ? atoi(getenv("FRIES_SPIN_PARITY")) : 0;       // time-reversal symmetry (heat_bathPP.cpp:1326-1407)
        apply_HBPP_piv(rr.sol->occ_orbs(), rr.sol->indices(), &pv, rr.hb, rr.basis_symm, rr.p_doub, nhb, m1, n_samp, rr.sing_sc, rr.doub_sc, spin_parity);
        fo::HBPivScratch ps; ps.init(len, n_states);
        std::copy(fr.sol.vals[0].begin(), fr.sol.vals[0].begin() + n, ps.vec1.begin());
        for (size_t i = 0; i < n; i++) ps.det_idx1[i] = i;
        ps.vec_len = n;
        fo::apply_HBPP_piv(fr.sol, ps, fr.sys, fr.p_doub, nhb, m2, n_samp, false, fo::Comm::self(), spin_parity);
        CHECK(ps.vec_len == pv.vec_len, "hbpiv n_samp %u len %zu %zu", n_samp, ps.vec_len, (size_t)pv.vec_len);
        CHECK(m1() == m2(), "hbpiv n_samp %u generator state", n_samp);
        size_t bad = 0;
        for (size_t s = 0; s < pv.vec_len && s < ps.vec_len; s++)
            if (memcmp(pv.orb_indices1[s], &ps.orb1[4 * s], 4) || !same_bits(pv.vec1[s], ps.vec1[s]) || pv.det_indices2[s] != ps.det_idx2[s]) bad++;
        CHECK(bad == 0, "hbpiv n_samp %u: %zu samples differ", n_samp, bad);
        fprintf(f, "CASE %u %u %zu %zu %zu %zu %zu %zu\n", n_samp, pseed, (size_t)pv.vec_len, ps.stage_len[0], ps.stage_len[1], ps.stage_len[2], ps.stage_len[3], ps.stage_len[4]);
        for (size_t s = 0; s < pv.vec_len; s++)
            fprintf(f, "%zu %u %u %u %u %a\n", (size_t)pv.det_indices2[s], pv.orb_indices1[s][0], pv.orb_indices1[s][1], pv.orb_indices1[s][2], pv.orb_indices1[s][3], pv.vec1[s]);
        printf("HBPIV n_samp=%u seed=%u n_out=%zu stages %zu %zu %zu %zu %zu\n", n_samp, pseed, (size_t)pv.vec_len, ps.stage_len[0], ps.stage_len[1], ps.stage_len[2], ps.stage_len[3], ps.stage_len[4]);
    }
    fclose(f);
    printf("HBPIV checks=%d fails=%d\n", n_chk, n_fail);
    return n_fail != 0;
}

// ------------------------------------------------------------------ time-reversal symmetry (spin_parity = +-1)
// tr <fcidump> <pg> <seed> <n_src> <out>
// Function level: flip_spins (fci_utils.c:158-204) on random determinants for every n_orb in 5 .. 32, tr_doub_connect (:310-359) on random
// occupied lists.  Operator level: h_op_offdiag (molecule.cpp:448-665) with spin_parity +1 and -1 applied to a vector of n_src determinants
// (HF and the first determinants H connects it to, random values) -- reference against restatement, result written to <out>:
// SRC lines (determinant value), then for each parity a PARITY line and the stored (determinant value) pairs of column 1 in position order.
static int run_tr(int argc, char **argv) {
    if (argc < 7) { fprintf(stderr, "usage: tr <fcidump> <pg> <seed> <n_src> <out>\n"); return 2; }
    const char *path = argv[2], *pg = argv[3];
    uint32_t seed = strtoul(argv[4], 0, 10); size_t n_src = strtoull(argv[5], 0, 10);
    std::mt19937 rg(seed);
    for (unsigned no = 5; no <= 32; no++) {
        const unsigned nb = CEILING(2 * no, 8);
        for (int trial = 0; trial < 400; trial++) {
            fo::det_t d = ((fo::det_t)rg() << 32 | rg());
            if (2 * no < 64) d &= ((fo::det_t)1 << (2 * no)) - 1;
            uint8_t in[9] = {0}, out[9] = {0};
            memcpy(in, &d, 8);
            flip_spins(in, out, (uint8_t)no);
            fo::det_t r = 0; memcpy(&r, out, nb);
            if (2 * no < 64) r &= ((fo::det_t)1 << (2 * no)) - 1;
            CHECK(r == fo::flip_spins(d, no), "flip_spins n_orb %u det %" PRIx64 ": %" PRIx64 " vs %" PRIx64, no, (uint64_t)d, (uint64_t)r, (uint64_t)fo::flip_spins(d, no));
        }
    }
    fcidump_input *in = parse_fcidump(path, pg);
    const unsigned n_orb = in->n_orb_, n_elec = in->n_elec;
    const size_t det_size = CEILING(2 * n_orb, 8);
    for (int trial = 0; trial < 4000; trial++) {
        uint8_t occ[64]; unsigned k = 0;
        for (int sp = 0; sp < 2; sp++) { std::vector<unsigned> o; while (o.size() < n_elec / 2) { unsigned c = rg() % n_orb; if (std::find(o.begin(), o.end(), c) == o.end()) o.push_back(c); }
            if (sp == 1 && trial % 3 == 0) { for (unsigned q = 0; q < n_elec / 2; q++) o[q] = occ[q]; if (trial % 6 == 0) o[rg() % (n_elec / 2)] = rg() % n_orb; std::sort(o.begin(), o.end()); o.erase(std::unique(o.begin(), o.end()), o.end()); while (o.size() < n_elec / 2) { unsigned c = rg() % n_orb; if (std::find(o.begin(), o.end(), c) == o.end()) o.push_back(c); } }
            std::sort(o.begin(), o.end()); for (unsigned q : o) occ[k++] = (uint8_t)(q + sp * n_orb); }
        uint8_t d1[2] = {77, 77}, d2[2] = {77, 77};
        int r1 = tr_doub_connect(occ, n_orb, n_elec, d1), r2 = fo::tr_doub_connect(occ, n_orb, n_elec, d2);
        CHECK(r1 == r2 && (r1 != 1 || (d1[0] == d2[0] && d1[1] == d2[1])), "tr_doub_connect %d %d", r1, r2);
    }
    // the source vector: HF and what H connects it to, random values
    Matrix<double> *h_core = in->hcore; SymmERIs *eris = &in->eris;
    std::vector<uint32_t> pscr(2 * n_orb), vscr(2 * n_orb);
    for (auto &x : pscr) x = rg();
    for (auto &x : vscr) x = rg();
    fo::MolSys sys; sys.n_orb = n_orb; sys.n_elec = n_elec;
    fill_oracle_ints(sys.ints, *eris, *h_core, n_orb);
    sys.symm.init(in->symm, n_orb);
    uint8_t hf_det[8] = {0}, occ[64];
    gen_hf_bitstring(n_orb, n_elec, hf_det);
    find_bits(hf_det, occ, det_size);
    std::vector<uint8_t> ex(4 * (size_t)n_orb * n_orb * n_elec * n_elec);
    std::vector<fo::det_t> src; std::vector<double> val;
    src.push_back(to_u64(hf_det, det_size));
    size_t nd = doub_ex_symm(hf_det, occ, n_elec, n_orb, (uint8_t (*)[4])ex.data(), in->symm);
    for (size_t e = 0; e < nd && src.size() < n_src; e += 3) { uint8_t nw[8]; memcpy(nw, hf_det, 8); doub_det(nw, &ex[4 * e]); src.push_back(to_u64(nw, det_size)); }
    for (size_t i = 0; i < src.size(); i++) val.push_back((rg() % 2001 - 1000.0) / 250.0 + 0.013);
    FILE *f = fopen(argv[6], "w");
    fprintf(f, "# time-reversal symmetry through the reference: h_op_offdiag (molecule.cpp:448-665) with spin_parity +1 / -1 on the SRC vector; doubles as C99 hex floats\n");
    for (size_t i = 0; i < src.size(); i++) fprintf(f, "SRC %" PRIu64 " %a\n", (uint64_t)src[i], val[i]);
    const size_t cap = src.size() * (size_t)n_orb * n_orb * n_elec * n_elec + 64;
    for (int sp = 1; sp >= -1; sp -= 2) {
        std::function<double(const uint8_t *)> diag_sc = [n_orb, eris, h_core, n_elec](const uint8_t *o) { return diag_matrel(o, n_orb, *eris, *h_core, 0, n_elec); };
        DistVec<double> rv(cap, cap, n_orb * 2, n_elec, 1, diag_sc, 2, pscr, vscr);
        fo::Vec ov; ov.init(cap, cap, n_elec, 2);
        for (size_t i = 0; i < src.size(); i++) { uint8_t d[8]; memcpy(d, &src[i], 8); rv.add(d, val[i], 1); ov.add(src[i], val[i], 1); }
        rv.perform_add(0); ov.perform_add(0);
        std::vector<uint8_t> scratch(4 * (size_t)n_orb * n_orb * n_elec * n_elec);
        h_op_offdiag(rv, in->symm, n_orb, *eris, *h_core, scratch.data(), scratch.size(), 0, n_elec, 1, 1.0, sp);
        fo::h_op_offdiag(ov, ov.curr_size, sys, 1, 1.0, sp);
        CHECK(rv.curr_size() == ov.curr_size, "tr h_op parity %d sizes %zu %zu", sp, (size_t)rv.curr_size(), ov.curr_size);
        rv.set_curr_vec_idx(1);
        size_t bad = 0, n = std::min((size_t)rv.curr_size(), ov.curr_size);
        for (size_t i = 0; i < n; i++) if (to_u64(rv.indices()[i], det_size) != ov.dets[i] || !same_bits(rv.values()[i], ov.vals[1][i])) bad++;
        CHECK(bad == 0, "tr h_op parity %d: %zu entries differ", sp, bad);
        fprintf(f, "PARITY %d %zu\n", sp, (size_t)rv.curr_size());
        for (size_t i = 0; i < rv.curr_size(); i++) fprintf(f, "%" PRIu64 " %a\n", (uint64_t)to_u64(rv.indices()[i], det_size), rv.values()[i]);
        printf("TR parity %d: %zu stored, %zu differ\n", sp, (size_t)rv.curr_size(), bad);
    }
    fclose(f);
    printf("TR checks=%d fails=%d\n", n_chk, n_fail);
    return n_fail != 0;
}

// ref_harness hfdir <dir/> <out>: what the reference's parse_hf_input (io_utils.cpp:98-187) reads from a legacy HF directory:
// header (n_orb, n_elec, n_frz as u32), symm, eps, hf_en, h_core, then the FourDArr <ij|kl> re-packed as chemists' (pq|rs) = <pr|qs>
static int run_hfdir(const char *dir, const char *out) {
    hf_input in;
    parse_hf_input(dir, &in);
    const unsigned n = in.n_orb + in.n_frz / 2;
    FILE *f = fopen(out, "wb");
    uint32_t hdr[3] = {in.n_orb, in.n_elec, in.n_frz};
    fwrite(hdr, 4, 3, f);
    fwrite(in.symm - in.n_frz / 2, 1, n, f);
    fwrite(&in.eps, 8, 1, f); fwrite(&in.hf_en, 8, 1, f);
    fwrite(in.hcore->data(), 8, (size_t)n * n, f);
    const size_t np = (size_t)n * (n + 1) / 2;
    std::vector<double> packed(np * (np + 1) / 2, 0.0);
    for (unsigned q = 0; q < n; q++) for (unsigned p = 0; p <= q; p++) for (unsigned s2 = 0; s2 < n; s2++) for (unsigned r = 0; r <= s2; r++) {
        size_t p1 = (size_t)q * (q + 1) / 2 + p, p2 = (size_t)s2 * (s2 + 1) / 2 + r;
        if (p1 <= p2) packed[p2 * (p2 + 1) / 2 + p1] = (*in.eris)(p, r, q, s2);
    }
    fwrite(packed.data(), 8, packed.size(), f);
    fclose(f);
    return 0;
}

static int run_dump_ints(const char *path, const char *pg, const char *out) {
    fcidump_input *in = parse_fcidump(path, pg);
    fo::Integrals oi;
    fill_oracle_ints(oi, in->eris, *in->hcore, in->n_orb_);
    FILE *f = fopen(out, "wb");
    uint32_t hdr[2] = {in->n_orb_, in->n_elec};
    fwrite(hdr, 4, 2, f);
    fwrite(in->symm, 1, in->n_orb_, f);
    fwrite(&in->core_en, 8, 1, f);
    fwrite(oi.h.data(), 8, oi.h.size(), f);
    fwrite(oi.eri.data(), 8, oi.eri.size(), f);
    fclose(f);
    return 0;
}

// mpiexec -n P ref_harness frisys_mpi <same args as frisys> : every rank writes <out>.r<rank>
static int run_frisys_mpi(int argc, char **argv) {
    if (argc < 14) { fprintf(stderr, "usage: see header\n"); return 2; }
    const char *path = argv[2], *pg = argv[3];
    unsigned n_iter = atoi(argv[4]); uint32_t seed = strtoul(argv[5], 0, 10);
    double eps = atof(argv[6]); uint32_t vnz = strtoul(argv[7], 0, 10), mnz = strtoul(argv[8], 0, 10);
    size_t max_dets = strtoull(argv[9], 0, 10); double ini = atof(argv[10]), tgt = atof(argv[11]);
    int nhb = !strcmp(argv[12], "HB_unnorm");
    RefRun rr;
    rr.setup(path, pg, seed, eps, vnz, mnz, max_dets, ini, tgt, nhb);
    char fn[1024];
    snprintf(fn, sizeof fn, "%s.r%d", argv[13], rr.proc_rank);
    FILE *f = fopen(fn, "w");
    fprintf(f, "# golden trajectory from the reference under mpiexec -n %d, rank %d; cols: it numer denom norm shift nkept n_nonz curr_size num_success digest\n", rr.n_procs, rr.proc_rank);
    fprintf(f, "# p_doub %a hf_en %a n_htrial %zu hf_proc %u\n", rr.p_doub, rr.hf_en, (size_t)rr.htrial->curr_size(), rr.hf_proc);
    for (unsigned it = 0; it < n_iter; it++) {
        rr.iterate();
        uint64_t hsh = 1469598103934665603ull;
        for (size_t i = 0; i < rr.sol->curr_size(); i++) {
            double rv = rr.sol->values()[i];
            if (rv != 0) { uint64_t vb; memcpy(&vb, &rv, 8); fo::det_t rd = to_u64(rr.sol->indices()[i], rr.det_size); hsh = (hsh ^ rd) * 1099511628211ull; hsh = (hsh ^ vb) * 1099511628211ull; hsh = (hsh ^ i) * 1099511628211ull; }
        }
        fprintf(f, "%u %a %a %a %a %u %d %zu %zu %016" PRIx64 "\n", it, rr.numer, rr.denom, rr.glob_norm, rr.en_shift, rr.nkept, rr.sol->n_nonz(), (size_t)rr.sol->curr_size(), rr.num_success, hsh);
    }
    fclose(f);
    if (rr.proc_rank == 0) printf("FRISYS_MPI ranks=%d iters=%u perform_add=%llu\n", rr.n_procs, n_iter, rr.n_perform_add);
    return 0;
}

// [mpiexec -n P] ref_harness restart <fcidump> <pg> <n_iter> <seed> <eps> <vnz> <mnz> <max_dets> <ini> <tgt> <dist> <state.bin> <run_seed> [log]
// The REAL reference advanced from a given vector: state.bin = u64 n, u64 dets[n], f64 vals[n] (storage order).  Every rank keeps
// the determinants it owns (idx_to_proc), in file order, so that positions equal what DistVec::load / fries_vec_load produce.
// Prints one JSON line with the timing of the iteration loop (rank 0); `log` receives rank 0's per-iteration scalars.
static int run_restart(int argc, char **argv) {
    if (argc < 15) { fprintf(stderr, "usage: see header\n"); return 2; }
    const char *path = argv[2], *pg = argv[3];
    unsigned n_iter = atoi(argv[4]); uint32_t seed = strtoul(argv[5], 0, 10);
    double eps = atof(argv[6]); uint32_t vnz = strtoul(argv[7], 0, 10), mnz = strtoul(argv[8], 0, 10);
    size_t max_dets = strtoull(argv[9], 0, 10); double ini = atof(argv[10]), tgt = atof(argv[11]);
    int nhb = !strcmp(argv[12], "HB_unnorm");
    uint32_t run_seed = strtoul(argv[14], 0, 10);
    RefRun rr;
    rr.setup(path, pg, seed, eps, vnz, mnz, max_dets, ini, tgt, nhb);
    // empty the vector (setup left 100 |HF> on its owner), then add the state in file order
    DistVec<double> &sv = *rr.sol;
    for (size_t i = 0; i < sv.curr_size(); i++) if (*sv[i] != 0) { *sv[i] = 0; sv.del_at_pos(i); }
    FILE *f = fopen(argv[13], "rb");
    if (!f) { fprintf(stderr, "cannot open state\n"); return 2; }
    uint64_t n = 0;
    if (fread(&n, 8, 1, f) != 1) return 2;
    std::vector<uint64_t> dets(n); std::vector<double> vals(n);
    if (fread(dets.data(), 8, n, f) != n || fread(vals.data(), 8, n, f) != n) return 2;
    fclose(f);
    size_t i = 0;
    int more = 1;
    while (more) {
        while (i < n) {
            uint8_t d[8]; memcpy(d, &dets[i], 8);
            bool mine = sv.idx_to_proc(d) == (unsigned)rr.proc_rank;
            i++;
            if (mine && !sv.add(d, vals[i - 1], 1)) break;
        }
        sv.perform_add(0);
        more = sum_mpi((int)(i < n), rr.proc_rank, rr.n_procs);
    }
    rr.mt.seed(run_seed); rr.en_shift = 0; rr.last_one_norm = 0; rr.iterat = 0;
    FILE *lg = (argc > 15 && rr.proc_rank == 0) ? fopen(argv[15], "w") : NULL;
    MPI_Barrier(MPI_COMM_WORLD);
    auto t0 = std::chrono::steady_clock::now();
    size_t spawns = 0;
    for (unsigned it = 0; it < n_iter; it++) {
        rr.iterate();
        spawns += rr.num_success;
        if (lg) fprintf(lg, "%u %a %a %a %a %u %d %zu %zu\n", it, rr.numer, rr.denom, rr.glob_norm, rr.en_shift, rr.nkept, rr.sol->n_nonz(), (size_t)rr.sol->curr_size(), rr.num_success);
    }
    MPI_Barrier(MPI_COMM_WORLD);
    double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (lg) fclose(lg);
    int nn = sum_mpi(rr.sol->n_nonz(), rr.proc_rank, rr.n_procs);
    int sp = sum_mpi((int)(spawns / (n_iter ? n_iter : 1)), rr.proc_rank, rr.n_procs);
    if (rr.proc_rank == 0)
        printf("{\"kind\": \"reference\", \"ranks\": %d, \"iters\": %u, \"seconds\": %.6f, \"iters_per_s\": %.6f, \"spawns_per_iter\": %d, \"n_nonz\": %d}\n", rr.n_procs, n_iter, s, n_iter / s, sp, nn);
    return 0;
}


// [mpiexec -n P] ref_harness pin <fcidump> <pg> <seed> <eps> <m> <max_dets> <HB|HB_unnorm> <run_seed> <n_iter> <out>
// The BASELINE sizes, pinned: exactly what bench.py does, by the REAL reference.  (1) filler: frisys_mol from 100 x HF with
// vec_nonz = mat_nonz = m, initiator 0, no norm control, in blocks of 5 iterations until the (global) number of non-zeros
// reaches m, then 10 more; (2) restart: the non-zero entries, in storage order, scaled by m / one-norm (local norms summed in
// storage order, then over the ranks in rank order), become positions 0..n-1 of a fresh vector, initiator 1, target norm m,
// generator re-seeded with run_seed, shift 0; (3) n_iter iterations.  Every rank writes <out>[.r<rank>]: one row per
// iteration of both phases with the digest of its shard, and the restart's scale factor.
static int run_pin(int argc, char **argv) {
    if (argc < 12) { fprintf(stderr, "usage: see header\n"); return 2; }
    const char *path = argv[2], *pg = argv[3];
    uint32_t seed = strtoul(argv[4], 0, 10); double eps = atof(argv[5]);
    uint32_t m = strtoul(argv[6], 0, 10); size_t max_dets = strtoull(argv[7], 0, 10);
    int nhb = !strcmp(argv[8], "HB_unnorm");
    uint32_t run_seed = strtoul(argv[9], 0, 10); unsigned n_iter = atoi(argv[10]);
    RefRun rr;
    rr.setup(path, pg, seed, eps, m, m, max_dets, 0.0, 0.0, nhb);
    char fn[1024];
    if (rr.n_procs > 1) snprintf(fn, sizeof fn, "%s.r%d", argv[11], rr.proc_rank); else snprintf(fn, sizeof fn, "%s", argv[11]);
    FILE *f = fopen(fn, "w");
    fprintf(f, "# pinned run of the reference (ref_harness pin), %d rank(s), rank %d; rows: phase it numer denom norm shift nkept n_nonz curr_size num_success digest\n", rr.n_procs, rr.proc_rank);
    fprintf(f, "# args: <fcidump> %s seed %u eps %s m %u max_dets %zu %s run_seed %u n_iter %u\n", pg, seed, argv[5], m, max_dets, argv[8], run_seed, n_iter);
    fprintf(f, "# p_doub %a hf_en %a n_htrial %zu hf_proc %u\n", rr.p_doub, rr.hf_en, (size_t)rr.htrial->curr_size(), rr.hf_proc);
    auto row = [&](const char *ph, unsigned it) {
        fprintf(f, "%s %u %a %a %a %a %u %d %zu %zu %016" PRIx64 "\n", ph, it, rr.numer, rr.denom, rr.glob_norm, rr.en_shift, rr.nkept, rr.sol->n_nonz(), (size_t)rr.sol->curr_size(), rr.num_success, rr.digest());
        fflush(f);
    };
    unsigned it = 0;
    bool full = false;
    for (int blk = 0; blk < 200 && !full; blk++) {
        for (int k = 0; k < 5; k++, it++) { rr.iterate(); row("F", it); }
        full = sum_mpi(rr.sol->n_nonz(), rr.proc_rank, rr.n_procs) >= (int)m;
    }
    for (int k = 0; k < 10; k++, it++) { rr.iterate(); row("F", it); }
    std::vector<uint64_t> dets; std::vector<double> vals;
    double loc = 0;
    for (size_t i = 0; i < rr.sol->curr_size(); i++) {
        double v = rr.sol->values()[i];
        if (v != 0) { dets.push_back(to_u64(rr.sol->indices()[i], rr.det_size)); vals.push_back(v); loc += fabs(v); }
    }
    const double glob = sum_mpi(loc, rr.proc_rank, rr.n_procs);
    const double scale = (double)m / glob;
    for (auto &v : vals) v = v * scale;
    fprintf(f, "RESTART %zu %a %a %a\n", dets.size(), loc, glob, scale);
    rr.replace_vector(dets, vals);
    rr.init_thresh = 1.0; rr.target = (double)m;
    rr.mt.seed(run_seed); rr.en_shift = 0; rr.last_one_norm = 0; rr.iterat = 0;
    MPI_Barrier(MPI_COMM_WORLD);
    auto t0 = std::chrono::steady_clock::now();
    for (unsigned k = 0; k < n_iter; k++) { rr.iterate(); row("R", k); }
    MPI_Barrier(MPI_COMM_WORLD);
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    fclose(f);
    if (rr.proc_rank == 0) printf("{\"kind\": \"reference\", \"mode\": \"pin\", \"ranks\": %d, \"filler_iters\": %u, \"iters\": %u, \"seconds\": %.3f, \"iters_per_s\": %.4f}\n", rr.n_procs, it, n_iter, sec, n_iter / sec);
    return 0;
}


// ref_harness reload <fcidump> <pg> <n1> <seed> <eps> <vnz> <mnz> <max_dets> <ini> <tgt> <dist> <out> <n2> <tmpdir/>
// --load_dir, pinned: n1 iterations of the reference's loop (n1 a multiple of 10), DistVec::save (vec_utils.hpp:713-746) into
// tmpdir, then what frisys_mol does when restarted with --load_dir tmpdir and the same seed flag: a fresh vector,
// DistVec::load (:761-844: entries with |v| <= 1e-9 dropped, the rest compacted into positions 0.., re-hashed), the shift of
// S.txt's last line, last_one_norm = 0 (frisys_mol.cpp:337: the loaded norm goes into `last_norm`, which nothing reads), the
// generator seeded and advanced past the 2 n_orb draws of the vec scrambler (:141-144; the proc scrambler comes from hash.dat) --
// and n2 more iterations.  <out>: a LOADED line (entries kept, digest right after the load), then one row per iteration.
static int run_reload(int argc, char **argv) {
    if (argc < 16) { fprintf(stderr, "usage: see header\n"); return 2; }
    const char *path = argv[2], *pg = argv[3];
    unsigned n1 = atoi(argv[4]); uint32_t seed = strtoul(argv[5], 0, 10);
    double eps = atof(argv[6]); uint32_t vnz = strtoul(argv[7], 0, 10), mnz = strtoul(argv[8], 0, 10);
    size_t max_dets = strtoull(argv[9], 0, 10); double ini = atof(argv[10]), tgt = atof(argv[11]);
    int nhb = !strcmp(argv[12], "HB_unnorm");
    unsigned n2 = atoi(argv[14]);
    std::string dir = argv[15];
    RefRun r1;
    r1.setup(path, pg, seed, eps, vnz, mnz, max_dets, ini, tgt, nhb);
    for (unsigned it = 0; it < n1; it++) r1.iterate();
    r1.sol->save(dir);
    RefRun rr;
    rr.skip_init_dense = true;
    rr.setup(path, pg, seed, eps, vnz, mnz, max_dets, ini, tgt, nhb);
    rr.replace_vector({}, {});                 // a fresh, empty DistVec with the same scramblers
    rr.n_determ = rr.sol->load(dir);           // frisys_mol.cpp:258: the dense space is the one dense.txt records
    if (rr.n_determ) { rr.build_dense_h(); for (size_t i = 0; i < rr.n_determ; i++) rr.det_space_in.push_back(to_u64(rr.sol->indices()[i], rr.det_size)); }
    rr.en_shift = r1.en_shift; rr.last_one_norm = 0; rr.iterat = 0;
    rr.mt.seed(seed); rr.mt.discard(2 * rr.n_orb);
    FILE *f = fopen(argv[13], "w");
    fprintf(f, "# --load_dir through the reference: %u iterations, DistVec::save, DistVec::load into a fresh vector, %u iterations; rows: it numer denom norm shift nkept n_nonz curr_size num_success digest\n", n1, n2);
    fprintf(f, "# p_doub %a hf_en %a n_htrial %zu hf_proc %u\n", rr.p_doub, rr.hf_en, (size_t)rr.htrial->curr_size(), rr.hf_proc);
    fprintf(f, "LOADED %zu %d %016" PRIx64 " saved_size %zu shift %a\n", (size_t)rr.sol->curr_size(), rr.sol->n_nonz(), rr.digest(), (size_t)r1.sol->curr_size(), rr.en_shift);
    for (unsigned it = 0; it < n2; it++) {
        rr.iterate();
        fprintf(f, "%u %a %a %a %a %u %d %zu %zu %016" PRIx64 "\n", it, rr.numer, rr.denom, rr.glob_norm, rr.en_shift, rr.nkept, rr.sol->n_nonz(), (size_t)rr.sol->curr_size(), rr.num_success, rr.digest());
    }
    fclose(f);
    printf("RELOAD n1=%u n2=%u loaded=%zu of %zu\n", n1, n2, (size_t)rr.sol->curr_size(), (size_t)r1.sol->curr_size());
    return 0;
}

// ------------------------------------------------------------------ Hubbard-Holstein: reference frisys_hh loop vs the oracle
// [mpiexec -n P] ref_harness hh <n_iter> <seed> <n_elec> <n_sites> <eps> <U> <omega> <g> <gs_energy> <vec_nonz> <max_dets> <initiator> <target> <out>
// One rank: lockstep against fo::FrisysHH + unit checks of the bit-string helpers, writes <out>.  P ranks: every rank writes <out>.r<rank>.
static int run_hh(int argc, char **argv) {
    if (argc < 16) { fprintf(stderr, "usage: see header\n"); return 2; }
    unsigned n_iter = atoi(argv[2]); uint32_t seed = strtoul(argv[3], 0, 10);
    unsigned n_elec = atoi(argv[4]), hub_len = atoi(argv[5]);
    double eps = atof(argv[6]), hub_u = atof(argv[7]), ph_freq = atof(argv[8]), elec_ph = atof(argv[9]), hf_en = atof(argv[10]);
    uint32_t target_nonz = strtoul(argv[11], 0, 10); size_t max_n_dets = strtoull(argv[12], 0, 10);
    double init_thresh = atof(argv[13]), target_norm = atof(argv[14]);
    const double hub_t = 1;
    int n_procs = 1, proc_rank = 0;
    MPI_Comm_size(MPI_COMM_WORLD, &n_procs);
    MPI_Comm_rank(MPI_COMM_WORLD, &proc_rank);
    unsigned n_orb = hub_len;
    std::mt19937 mt_obj(seed);
    std::vector<uint32_t> proc_scrambler(2 * n_orb), vec_scrambler(2 * n_orb);
    for (auto &x : proc_scrambler) x = mt_obj();
    for (auto &x : vec_scrambler) x = mt_obj();
    const bool full = getenv("FRIES_HH_FULL") != nullptr;        // frifull_hh.cpp instead of frisys_hh.cpp
    unsigned spawn_length = target_nonz * 4 / n_procs;
    if (full) { size_t sl = (size_t)n_elec * 4 * max_n_dets / n_procs; spawn_length = sl > 200000 ? 200000u : (unsigned)sl; }      // frifull_hh.cpp:91-95
    uint8_t ph_bits = 3;
    std::function<double(const uint8_t *)> diag_shortcut = [hub_len](const uint8_t *det) { return hub_diag((uint8_t *)det, hub_len); };
    HubHolVec<double> sol_vec(max_n_dets, spawn_length, hub_len, ph_bits, n_elec, n_procs, diag_shortcut, 2, proc_scrambler, vec_scrambler);
    size_t det_size = CEILING(2 * n_orb + ph_bits * n_orb, 8);
    uint8_t neel_det[16] = {0};
    gen_neel_det_1D(n_orb, n_elec, ph_bits, neel_det);
    unsigned ref_proc = sol_vec.idx_to_proc(neel_det);
    uint8_t neel_occ[64];
    sol_vec.gen_orb_list(neel_det, neel_occ);
    if ((int)ref_proc == proc_rank) sol_vec.add(neel_det, 100, 1);
    sol_vec.perform_add(0);
    double en_shift = 0, last_one_norm = 0, glob_norm = 0;
    double loc_norms[64]; double rn_sys = 0;
    std::vector<size_t> srt_arr(max_n_dets); std::vector<bool> keep_exact(max_n_dets, false);
    Matrix<double> subwt_mem(spawn_length, 2); Matrix<bool> keep_idx(spawn_length, 2);
    std::vector<double> wt_remain(spawn_length, 0);
    std::vector<unsigned int> ndiv_vec(spawn_length);
    std::vector<double> comp_vec1(spawn_length), comp_vec2(spawn_length);
    std::vector<size_t> comp_idx_v(2 * (size_t)spawn_length), det_indices(spawn_length);
    size_t (*comp_idx)[2] = (size_t (*)[2])comp_idx_v.data();
    std::vector<bool> ph_ex(spawn_length, false);
    uint8_t new_det[16] = {0};
    double recv_nums[64];
    Matrix<uint8_t> &neighb_orbs = sol_vec.neighb();

    // oracle twin (one rank only)
    fo::FrisysHH fr;
    const bool lock = n_procs == 1;
    if (lock) {
        fr.par.n_elec = n_elec; fr.par.n_sites = hub_len; fr.par.ph_bits = ph_bits; fr.par.eps = eps; fr.par.U = hub_u; fr.par.omega = ph_freq; fr.par.g = elec_ph;
        fr.full = full;
        fr.par.hf_en = hf_en; fr.par.target_norm = target_norm; fr.par.init_thresh = init_thresh; fr.par.vec_nonz = target_nonz; fr.par.max_dets = max_n_dets; fr.par.seed = seed;
        fr.setup();
        CHECK(fr.neel == to_u64(neel_det, det_size), "neel det %" PRIx64 " %" PRIx64, (uint64_t)fr.neel, (uint64_t)to_u64(neel_det, det_size));
        // helper functions on random bit strings
        std::mt19937 rg(12345 + seed);
        for (int trial = 0; trial < 20000; trial++) {
            fo::det_t d = 0;
            for (int sp = 0; sp < 2; sp++) { unsigned placed = 0; while (placed < n_elec / 2) { unsigned o = rg() % hub_len; if (!((d >> (o + sp * hub_len)) & 1)) { d |= (fo::det_t)1 << (o + sp * hub_len); placed++; } } }
            bool with_ph = (trial % 3) != 0;
            if (with_ph) for (unsigned st = 0; st < hub_len; st++) if (rg() % 4 == 0) d |= (fo::det_t)(rg() % 8) << (2 * hub_len + 3 * st);
            uint8_t db[16] = {0}; memcpy(db, &d, 8);
            CHECK(hub_diag(db, hub_len) == fo::hub_diag(d, hub_len), "hub_diag %" PRIx64, (uint64_t)d);
            uint8_t nb_r[2 * 65] = {0}, nb_o[2 * 65] = {0};
            sol_vec.find_neighbors_1D(db, nb_r);
            fo::find_neighbors_1D(d, hub_len, n_elec, nb_o);
            bool same = nb_r[0] == nb_o[0] && nb_r[n_elec + 1] == nb_o[n_elec + 1];
            for (unsigned k = 0; same && k < nb_r[0]; k++) same = nb_r[1 + k] == nb_o[1 + k];
            for (unsigned k = 0; same && k < nb_r[n_elec + 1]; k++) same = nb_r[n_elec + 2 + k] == nb_o[n_elec + 2 + k];
            CHECK(same, "neighbors %" PRIx64, (uint64_t)d);
            uint8_t ph_r[64], ph_o[64];
            sol_vec.decode_phonons(db, ph_r); fo::decode_phonons(d, hub_len, ph_bits, ph_o);
            CHECK(!memcmp(ph_r, ph_o, hub_len), "phonons %" PRIx64, (uint64_t)d);
            unsigned st = rg() % hub_len; int chg = (rg() & 1) ? 1 : -1;
            uint8_t nd_r[16] = {0}; fo::det_t nd_o = 0;
            bool ok_o = fo::det_from_ph(d, &nd_o, hub_len, ph_bits, st, chg);
            bool at_max = ph_r[st] == 7 && chg == 1;
            if (!at_max) { int ok_r = sol_vec.det_from_ph(db, nd_r, st, chg); CHECK((ok_r != 0) == ok_o && (!ok_o || to_u64(nd_r, det_size) == nd_o), "det_from_ph"); }
            else CHECK(!ok_o, "det_from_ph at max");
            // reference overlap of a single element
            Matrix<uint8_t> dm(1, det_size); memcpy(dm[0], db, det_size);
            Matrix<uint8_t> pm(1, hub_len); memcpy(pm[0], ph_r, hub_len);
            double v1 = 1.25;
            double r_ref = calc_ref_ovlp(dm, &v1, pm, 1, neel_det, neel_occ, n_elec, hub_len, elec_ph / hub_t);
            double r_orc = fo::calc_ref_ovlp(&d, &v1, 1, fr.neel, n_elec, hub_len, ph_bits, elec_ph / hub_t);
            CHECK(same_bits(r_ref, r_orc), "calc_ref_ovlp %" PRIx64 " %a %a", (uint64_t)d, r_ref, r_orc);
            if (trial < 2000) {      // states one hop / one phonon away from the Neel state exercise the non-zero branches
                fo::det_t e = fr.neel;
                uint8_t nbn[2 * 65]; fo::find_neighbors_1D(e, hub_len, n_elec, nbn);
                if (trial & 1) { unsigned tot = nbn[0] + nbn[n_elec + 1]; if (tot) { unsigned x = rg() % tot; unsigned og, ds; if (x < nbn[0]) { og = nbn[1 + x]; ds = og + 1; } else { og = nbn[n_elec + 2 + x - nbn[0]]; ds = og - 1; } e = (e & ~((fo::det_t)1 << og)) | ((fo::det_t)1 << ds); } }
                else { e |= (fo::det_t)(1 + rg() % 2) << (2 * hub_len + 3 * (rg() % hub_len)); if (rg() & 1) e |= (fo::det_t)1 << (2 * hub_len + 3 * (rg() % hub_len)); }
                uint8_t eb[16] = {0}; memcpy(eb, &e, 8);
                uint8_t phe[64]; sol_vec.decode_phonons(eb, phe);
                memcpy(dm[0], eb, det_size); memcpy(pm[0], phe, hub_len);
                double a = calc_ref_ovlp(dm, &v1, pm, 1, neel_det, neel_occ, n_elec, hub_len, elec_ph / hub_t);
                double b = fo::calc_ref_ovlp(&e, &v1, 1, fr.neel, n_elec, hub_len, ph_bits, elec_ph / hub_t);
                CHECK(same_bits(a, b), "calc_ref_ovlp near neel %" PRIx64 " %a %a", (uint64_t)e, a, b);
            }
        }
    }
    char fn[1024];
    if (lock) snprintf(fn, sizeof fn, "%s", argv[15]); else snprintf(fn, sizeof fn, "%s.r%d", argv[15], proc_rank);
    FILE *f = fopen(fn, "w");
    fprintf(f, "# golden trajectory from the reference (frisys_hh.cpp loop, %d rank(s), rank %d); cols: it numer denom norm shift nkept n_nonz curr_size num_success digest\n", n_procs, proc_rank);
    fprintf(f, "# ref_proc %u neel %016" PRIx64 "\n", ref_proc, (uint64_t)to_u64(neel_det, det_size));
    uint8_t (*spawn_orbs)[2] = (uint8_t (*)[2])malloc(sizeof(uint8_t) * n_elec * 2 * 2);
    for (unsigned iterat = 0; iterat < n_iter; iterat++) {
        size_t det_idx;
        size_t num_success = 0, vec_size = sol_vec.curr_size();
        if (full) {         // frifull_hh.cpp:187-263
            det_idx = 0;
            int num_added = 1;
            size_t adder_size = sol_vec.adder_size() - n_elec * 4;
            double *vals_before_mult = sol_vec.values();
            sol_vec.set_curr_vec_idx(1);
            sol_vec.zero_vec();
            while (num_added > 0) {
                num_added = 0;
                while (det_idx < vec_size && (size_t)num_added < adder_size) {
                    double curr_el = vals_before_mult[det_idx];
                    if (curr_el == 0) { det_idx++; continue; }
                    uint8_t *curr_det = sol_vec.indices()[det_idx];
                    int ini_flag = fabs(curr_el) > init_thresh;
                    size_t n_success = hub_all(n_elec, neighb_orbs[det_idx], spawn_orbs);
                    for (size_t ex_idx = 0; ex_idx < n_success; ex_idx++) {
                        memcpy(new_det, curr_det, det_size);
                        zero_bit(new_det, spawn_orbs[ex_idx][0]);
                        set_bit(new_det, spawn_orbs[ex_idx][1]);
                        sol_vec.add(new_det, eps * hub_t * curr_el, ini_flag);
                    }
                    num_added += n_success;
                    uint8_t *curr_occ = sol_vec.orbs_at_pos(det_idx);
                    uint8_t *curr_phonons = sol_vec.phonons_at_pos(det_idx);
                    for (size_t elec_idx = 0; elec_idx < n_elec / 2; elec_idx++) {
                        uint8_t site = curr_occ[elec_idx];
                        uint8_t phonon_num = curr_phonons[site];
                        int doubly_occ = read_bit(curr_det, site + hub_len);
                        if (phonon_num > 0) { sol_vec.det_from_ph(curr_det, new_det, site, -1); sol_vec.add(new_det, -eps * elec_ph * sqrt(phonon_num) * (doubly_occ + 1) * curr_el, ini_flag); num_added++; }
                        if (phonon_num + 1 < (1 << ph_bits)) { sol_vec.det_from_ph(curr_det, new_det, site, +1); sol_vec.add(new_det, -eps * elec_ph * sqrt(phonon_num + 1) * (doubly_occ + 1) * curr_el, ini_flag); num_added++; }
                    }
                    for (size_t elec_idx = n_elec / 2; elec_idx < n_elec; elec_idx++) {
                        uint8_t site = curr_occ[elec_idx] - n_orb;
                        int doubly_occ = read_bit(curr_det, site);
                        if (!doubly_occ) {
                            uint8_t phonon_num = curr_phonons[site];
                            if (phonon_num > 0) { sol_vec.det_from_ph(curr_det, new_det, site, -1); sol_vec.add(new_det, -eps * elec_ph * sqrt(phonon_num) * curr_el, ini_flag); num_added++; }
                            if (phonon_num + 1 < (1 << ph_bits)) { sol_vec.det_from_ph(curr_det, new_det, site, +1); sol_vec.add(new_det, -eps * elec_ph * sqrt(phonon_num + 1) * curr_el, ini_flag); num_added++; }
                        }
                    }
                    det_idx++;
                }
                num_success += (size_t)num_added;
                num_added = sum_mpi(num_added, proc_rank, n_procs);
                sol_vec.perform_add(0);
                sol_vec.set_curr_vec_idx(0);
                vals_before_mult = sol_vec.values();
                sol_vec.set_curr_vec_idx(1);
            }
        }
        else {
        for (det_idx = 0; det_idx < sol_vec.curr_size(); det_idx++) {
            double *curr_el = sol_vec[det_idx];
            double weight = fabs(*curr_el);
            comp_vec1[det_idx] = weight;
            if (weight > 0) { subwt_mem(det_idx, 0) = hub_t; subwt_mem(det_idx, 1) = elec_ph; ndiv_vec[det_idx] = 0; }
            else ndiv_vec[det_idx] = 1;
        }
        if (proc_rank == 0) rn_sys = mt_obj() / (1. + UINT32_MAX);
        size_t comp_len = comp_sub(comp_vec1.data(), sol_vec.curr_size(), ndiv_vec.data(), subwt_mem, keep_idx, NULL, target_nonz, wt_remain.data(), rn_sys, comp_vec2.data(), comp_idx);
        for (size_t samp_idx = 0; samp_idx < comp_len; samp_idx++) {
            det_idx = comp_idx[samp_idx][0];
            det_indices[samp_idx] = det_idx;
            ph_ex[samp_idx] = comp_idx[samp_idx][1];
            if (ph_ex[samp_idx]) ndiv_vec[samp_idx] = 2 * n_elec;
            else ndiv_vec[samp_idx] = neighb_orbs(det_idx, 0) + neighb_orbs(det_idx, n_elec + 1);
            comp_vec2[samp_idx] *= ndiv_vec[samp_idx];
        }
        if (proc_rank == 0) rn_sys = mt_obj() / (1. + UINT32_MAX);
        comp_len = comp_sub(comp_vec2.data(), comp_len, ndiv_vec.data(), subwt_mem, keep_idx, NULL, target_nonz, wt_remain.data(), rn_sys, comp_vec1.data(), comp_idx);
        num_success = comp_len;
        double *vals_before_mult = sol_vec.values();
        sol_vec.set_curr_vec_idx(1);
        sol_vec.zero_vec();
        vec_size = sol_vec.curr_size();
        for (int add_ini = 0; add_ini < 2; add_ini++) {
            int num_added = 1;
            size_t samp_idx = 0;
            while (num_added > 0) {
                num_added = 0;
                while (samp_idx < comp_len) {
                    size_t prev_idx = comp_idx[samp_idx][0];
                    size_t d_idx = det_indices[prev_idx];
                    double curr_val = vals_before_mult[d_idx];
                    uint8_t ini_flag = fabs(curr_val) >= init_thresh;
                    if (ini_flag != add_ini) { samp_idx++; continue; }
                    uint8_t exc_idx = comp_idx[samp_idx][1];
                    uint8_t *curr_det = sol_vec.indices()[d_idx];
                    double matr_el = comp_vec1[samp_idx] * -eps;
                    if (curr_val < 0) matr_el *= -1;
                    if (ph_ex[prev_idx]) {
                        uint8_t *curr_ph = sol_vec.phonons_at_pos(d_idx);
                        uint8_t *curr_occ = sol_vec.orbs_at_pos(d_idx);
                        uint8_t site = curr_occ[exc_idx % n_elec] % hub_len;
                        uint8_t phonon_num = curr_ph[site];
                        if (exc_idx < n_elec && phonon_num > 0) { sol_vec.det_from_ph(curr_det, new_det, site, -1); matr_el *= sqrt(phonon_num); }
                        else if (exc_idx >= n_elec && phonon_num + 1 < (1 << ph_bits)) { sol_vec.det_from_ph(curr_det, new_det, site, +1); matr_el *= sqrt(phonon_num + 1); }
                        else matr_el = 0;
                    }
                    else {
                        std::copy(curr_det, curr_det + det_size, new_det);
                        uint8_t *curr_neighb = neighb_orbs[d_idx];
                        uint8_t orig_orb, dest_orb;
                        if (exc_idx < curr_neighb[0]) { orig_orb = curr_neighb[exc_idx + 1]; dest_orb = orig_orb + 1; }
                        else { orig_orb = curr_neighb[n_elec + 1 + exc_idx - curr_neighb[0] + 1]; dest_orb = orig_orb - 1; }
                        zero_bit(new_det, orig_orb);
                        set_bit(new_det, dest_orb);
                        matr_el *= -1;
                    }
                    samp_idx++;
                    if (fabs(matr_el) > 1e-9) { num_added++; if (!sol_vec.add(new_det, matr_el, ini_flag)) break; }
                }
                sol_vec.perform_add(0);
                sol_vec.set_curr_vec_idx(0);
                vals_before_mult = sol_vec.values();
                sol_vec.set_curr_vec_idx(1);
                num_added = sum_mpi(num_added, proc_rank, n_procs);
            }
        }
        }
        size_t new_max_dets = sol_vec.max_size();
        if (new_max_dets > max_n_dets) { keep_exact.resize(new_max_dets, false); srt_arr.resize(new_max_dets); max_n_dets = new_max_dets; }
        sol_vec.set_curr_vec_idx(0);
        for (det_idx = 0; det_idx < vec_size; det_idx++) {
            double *curr_el = sol_vec[det_idx];
            if (*curr_el != 0) {
                double diag_el = sol_vec.matr_el_at_pos(det_idx);
                double phonon_diag = sol_vec.total_ph(det_idx) * ph_freq;
                *curr_el *= 1 - eps * (diag_el * hub_u + phonon_diag - hf_en - en_shift);
            }
        }
        sol_vec.add_vecs(0, 1);
        unsigned int n_samp = target_nonz;
        loc_norms[proc_rank] = find_preserve(sol_vec.values(), srt_arr, keep_exact, sol_vec.curr_size(), &n_samp, &glob_norm);
        unsigned nkept = target_nonz - n_samp;
        if ((iterat + 1) % 10 == 0) adjust_shift(&en_shift, glob_norm, &last_one_norm, target_norm, 0.05 / 10 / eps);
        double numer = calc_ref_ovlp(sol_vec.indices(), sol_vec.values(), sol_vec.phonon_nums(), sol_vec.curr_size(), neel_det, neel_occ, n_elec, hub_len, elec_ph / hub_t);
        MPI_Gather(&numer, 1, MPI_DOUBLE, recv_nums, 1, MPI_DOUBLE, ref_proc, MPI_COMM_WORLD);
        double ref_element = 0;
        numer = 0;
        if (proc_rank == (int)ref_proc) {
            double diag_el = sol_vec.matr_el_at_pos(0);
            ref_element = *(sol_vec[0]);
            numer = (diag_el * hub_u - hf_en) * ref_element;
            for (int proc_idx = 0; proc_idx < n_procs; proc_idx++) numer += recv_nums[proc_idx] * -hub_t;
        }
        if (proc_rank == 0) rn_sys = mt_obj() / (1. + UINT32_MAX);
        MPI_Allgather(MPI_IN_PLACE, 0, MPI_DOUBLE, loc_norms, 1, MPI_DOUBLE, MPI_COMM_WORLD);
        sys_comp(sol_vec.values(), sol_vec.curr_size(), loc_norms, n_samp, keep_exact, rn_sys);
        for (det_idx = 0; det_idx < sol_vec.curr_size(); det_idx++) {
            if (keep_exact[det_idx] && !(proc_rank == 0 && det_idx == 0)) { sol_vec.del_at_pos(det_idx); keep_exact[det_idx] = 0; }
        }
        uint64_t hsh = 1469598103934665603ull;
        for (size_t i = 0; i < sol_vec.curr_size(); i++) {
            double rv = sol_vec.values()[i];
            if (rv != 0) { uint64_t vb; memcpy(&vb, &rv, 8); fo::det_t rd = to_u64(sol_vec.indices()[i], det_size); hsh = (hsh ^ rd) * 1099511628211ull; hsh = (hsh ^ vb) * 1099511628211ull; hsh = (hsh ^ i) * 1099511628211ull; }
        }
        fprintf(f, "%u %a %a %a %a %u %d %zu %zu %016" PRIx64 "\n", iterat, numer, ref_element, glob_norm, en_shift, nkept, sol_vec.n_nonz(), (size_t)sol_vec.curr_size(), num_success, hsh);
        if (lock) {
            fr.iterate(1);
            const fo::HHLog &lg = fr.log.back();
            CHECK(same_bits(lg.numer, numer) && same_bits(lg.denom, ref_element), "hh it %u numer/denom %a %a | %a %a", iterat, lg.numer, numer, lg.denom, ref_element);
            CHECK(same_bits(lg.norm, glob_norm) && same_bits(lg.shift, en_shift), "hh it %u norm/shift", iterat);
            CHECK(lg.nkept == nkept && lg.n_nonz == sol_vec.n_nonz() && lg.curr_size == sol_vec.curr_size() && lg.num_success == num_success, "hh it %u counts nkept %u/%u nnz %d/%d size %zu/%zu succ %zu/%zu",
                  iterat, lg.nkept, nkept, lg.n_nonz, sol_vec.n_nonz(), lg.curr_size, (size_t)sol_vec.curr_size(), lg.num_success, num_success);
            size_t bad = 0, nmin = std::min(lg.curr_size, (size_t)sol_vec.curr_size());
            for (size_t i = 0; i < nmin; i++) {
                double rv = sol_vec.values()[i];
                if (!same_bits(rv, fr.sol.vals[0][i])) bad++;
                if (rv != 0 && to_u64(sol_vec.indices()[i], det_size) != fr.sol.dets[i]) bad++;
            }
            CHECK(bad == 0, "hh it %u vector mismatch in %zu slots", iterat, bad);
        }
    }
    fclose(f);
    if (proc_rank == 0) printf("HH ranks=%d iters=%u checks=%d fails=%d final n_nonz=%d\n", n_procs, n_iter, n_chk, n_fail, sol_vec.n_nonz());
    return n_fail != 0;
}


// ------------------------------------------------------------------ fciqmc_mol (NU): reference loop vs the oracle in mt mode
// ref_harness fciqmc <fcidump> <pg> <n_iter> <seed> <eps> <target_walkers> <max_dets> <initiator> <out> [NU|HB]
static int run_fciqmc(int argc, char **argv) {
    if (argc < 11) { fprintf(stderr, "usage: see header\n"); return 2; }
    const char *path = argv[2], *pg = argv[3];
    unsigned n_iter = atoi(argv[4]); uint32_t seed = strtoul(argv[5], 0, 10);
    double eps = atof(argv[6]); uint32_t target_walkers = strtoul(argv[7], 0, 10);
    uint32_t max_n_dets = strtoul(argv[8], 0, 10); uint32_t init_thresh = strtoul(argv[9], 0, 10);
    const bool heat_bath = argc > 11 && !strcmp(argv[11], "HB");
    int n_procs = 1, proc_rank = 0;
    MPI_Comm_size(MPI_COMM_WORLD, &n_procs);
    MPI_Comm_rank(MPI_COMM_WORLD, &proc_rank);
    fcidump_input *in_data = parse_fcidump(path, pg);
    unsigned n_elec = in_data->n_elec, n_frz = 0, n_orb = in_data->n_orb_;
    size_t det_size = CEILING(2 * n_orb, 8);
    unsigned n_elec_unf = n_elec, tot_orb = n_orb;
    uint8_t *symm = in_data->symm;
    Matrix<double> *h_core = in_data->hcore; SymmERIs *eris = &in_data->eris;
    uint8_t tmp_orbs[64], hf_det[8] = {0};
    gen_hf_bitstring(n_orb, n_elec, hf_det);
    find_bits(hf_det, tmp_orbs, det_size);
    double hf_en = diag_matrel(tmp_orbs, tot_orb, *eris, *h_core, n_frz, n_elec);
    std::mt19937 mt_obj(seed + (uint32_t)proc_rank);      // the reference seeds every process from its own clock (fciqmc_mol.cpp:102-104)
    unsigned spawn_length = target_walkers / n_procs / n_procs * 2;      // :107
    std::function<double(const uint8_t *)> diag_shortcut = [tot_orb, eris, h_core, n_frz, n_elec, hf_en](const uint8_t *occ) { return diag_matrel(occ, tot_orb, *eris, *h_core, n_frz, n_elec) - hf_en; };
    SymmInfo symm_basis(symm, n_orb);
    unsigned unocc_symm_cts[n_irreps][2];
    std::vector<uint32_t> proc_scrambler(2 * n_orb), vec_scrambler(2 * n_orb);
    if (proc_rank == 0) for (auto &x : proc_scrambler) x = mt_obj();      // :123-131
    MPI_Bcast(proc_scrambler.data(), 2 * n_orb, MPI_UNSIGNED, 0, MPI_COMM_WORLD);
    for (auto &x : vec_scrambler) x = mt_obj();
    DistVec<int> sol_vec(max_n_dets, spawn_length, n_orb * 2, n_elec_unf, n_procs, diag_shortcut, 1, proc_scrambler, vec_scrambler);
    unsigned hf_proc = sol_vec.idx_to_proc(hf_det);
    unsigned max_spawn = 500000;
    std::vector<uint8_t> spawn_orbs_v(4 * (size_t)max_spawn); std::vector<double> spawn_probs(max_spawn);
    uint8_t (*sing_orbs)[2] = (uint8_t (*)[2])spawn_orbs_v.data();
    uint8_t (*doub_orbs)[4] = (uint8_t (*)[4])spawn_orbs_v.data();
    size_t n_ex = (size_t)n_orb * n_orb * n_elec_unf * n_elec_unf;
    // --trial_vec (fciqmc_mol.cpp:150-177): the reference's own text reader and its `while (!add) perform_add` loops
    const char *trial_prefix = getenv("FRIES_TRIAL");
    size_t n_trial = 1;
    std::vector<fo::det_t> tin_det; std::vector<double> tin_val;
    {
        Matrix<uint8_t> &load_dets = sol_vec.indices();
        double *load_vals = (double *)sol_vec.values();
        if (trial_prefix) {
            n_trial = load_vec_txt(std::string(trial_prefix), load_dets, load_vals);
            for (size_t i = 0; i < n_trial; i++) { tin_det.push_back(to_u64(load_dets[i], det_size)); tin_val.push_back(load_vals[i]); }
        }
    }
    DistVec<double> trial_vec(trial_prefix ? n_trial : 4, trial_prefix ? n_trial : 4, n_orb * 2, n_elec_unf, n_procs, proc_scrambler, vec_scrambler);
    DistVec<double> htrial_vec(trial_prefix ? n_trial * n_ex / n_procs : 2 * n_ex, trial_prefix ? n_trial * n_ex / n_procs : 2 * n_ex, n_orb * 2, n_elec_unf, n_procs, diag_shortcut, 2, proc_scrambler, vec_scrambler);
    if (trial_prefix) {
        Matrix<uint8_t> &load_dets = sol_vec.indices();
        double *load_vals = (double *)sol_vec.values();
        for (size_t i = 0; i < n_trial; i++) {
            while (!trial_vec.add(load_dets[i], load_vals[i], 1)) trial_vec.perform_add(0);
            while (!htrial_vec.add(load_dets[i], load_vals[i], 1)) htrial_vec.perform_add(0);
        }
        memset(sol_vec.values(), 0, (n_trial + 1) * sizeof(int));
    }
    else if ((int)hf_proc == proc_rank) { trial_vec.add(hf_det, 1, 1); htrial_vec.add(hf_det, 1, 1); }
    trial_vec.perform_add(0); htrial_vec.perform_add(0);
    trial_vec.collect_procs();
    std::vector<uintmax_t> trial_hashes(trial_vec.curr_size());
    for (size_t i = 0; i < trial_vec.curr_size(); i++) trial_hashes[i] = sol_vec.idx_to_hash(trial_vec.indices()[i], tmp_orbs);
    h_op_offdiag(htrial_vec, symm, tot_orb, *eris, *h_core, spawn_orbs_v.data(), 4 * max_spawn, n_frz, n_elec_unf, 1, 1, 0);
    htrial_vec.set_curr_vec_idx(0);
    h_op_diag(htrial_vec, 0, 0, 1);
    htrial_vec.add_vecs(0, 1);
    htrial_vec.collect_procs();
    std::vector<uintmax_t> htrial_hashes(htrial_vec.curr_size());
    for (size_t i = 0; i < htrial_vec.curr_size(); i++) htrial_hashes[i] = sol_vec.idx_to_hash(htrial_vec.indices()[i], tmp_orbs);
    sol_vec.gen_orb_list(hf_det, tmp_orbs);
    size_t n_hf_doub = doub_ex_symm(hf_det, tmp_orbs, n_elec_unf, n_orb, doub_orbs, symm);
    size_t n_hf_sing = count_singex(hf_det, tmp_orbs, n_elec_unf, &symm_basis);
    double p_doub = (double)n_hf_doub / (n_hf_sing + n_hf_doub);
    std::vector<fo::det_t> iin_det; std::vector<int> iin_val;
    if (getenv("FRIES_INI")) {       // --ini_vec (:226-237)
        Matrix<uint8_t> &load_dets = sol_vec.indices();
        int *load_vals = sol_vec.values();
        size_t n_dets = load_vec_txt(std::string(getenv("FRIES_INI")), load_dets, load_vals);
        std::vector<std::vector<uint8_t>> keep_d(n_dets, std::vector<uint8_t>(det_size)); std::vector<int> keep_v(n_dets);
        for (size_t i = 0; i < n_dets; i++) { memcpy(keep_d[i].data(), load_dets[i], det_size); keep_v[i] = load_vals[i]; iin_det.push_back(to_u64(load_dets[i], det_size)); iin_val.push_back(load_vals[i]); }
        memset(load_vals, 0, (n_dets + 1) * sizeof(int));       // the reference adds straight out of the vector's own arrays; keep the harness well defined
        for (size_t i = 0; i < n_dets; i++) while (!sol_vec.add(keep_d[i].data(), keep_v[i], 1)) sol_vec.perform_add(0);
    }
    else if ((int)hf_proc == proc_rank) sol_vec.add(hf_det, 100, 1);
    sol_vec.perform_add(0);
    double en_shift = 0, last_norm = 0, glob_norm = 0;
    const bool lockstep = n_procs == 1;

    fo::Fciqmc fq;
    fq.sys.n_orb = n_orb; fq.sys.n_elec = n_elec;
    fill_oracle_ints(fq.sys.ints, *eris, *h_core, n_orb);
    fq.sys.symm.init(symm, n_orb);
    fq.par.eps = eps; fq.par.target_walkers = target_walkers; fq.par.init_thresh = init_thresh; fq.par.max_dets = max_n_dets; fq.par.seed = seed; fq.par.counter_rng = false;
    fq.par.heat_bath = heat_bath;
    fq.trial_in_det = tin_det; fq.trial_in_val = tin_val; fq.ini_det = iin_det; fq.ini_val.assign(iin_val.begin(), iin_val.end());
    if (lockstep) fq.setup();
    hb_info *hb_probs = heat_bath ? set_up(tot_orb, n_orb, *eris) : NULL;
    if (lockstep) CHECK(same_bits(fq.p_doub, p_doub), "fciqmc p_doub");
    // the sampling functions on their own, from identical mt19937 states
    if (lockstep) {
        std::mt19937 ga(777 + seed), gb(777 + seed);
        fo::Rng rb; rb.mt = &gb;
        // The reference's calc_u1_probs reads occ_orbs[n_elec] -- one entry past the list -- when the first occupied orbital is the last
        // electron (heat_bathPP.cpp:300-301: `occ_idx++; curr_occ = occ_orbs[occ_idx];`).  Inside a DistVec that byte is the next row's
        // first (alpha) orbital, which can never equal the beta index it is compared with, so the overrun is harmless there; a stack array
        // with an unwritten tail made this comparison depend on whatever the stack held (the "intermittent" mismatches of round 1).  The
        // tail is therefore filled with a value no orbital index takes, which is also what the restatement assumes (fo::calc_u1_probs).
        uint8_t occ[64];
        memset(occ, 0xff, sizeof occ);
        for (int trial = 0; trial < 4000; trial++) {
            fo::det_t d = 0;
            std::mt19937 &rg = ga;
            for (int sp = 0; sp < 2; sp++) { unsigned placed = 0; while (placed < n_elec / 2) { unsigned o = rg() % n_orb; if (!((d >> (o + sp * n_orb)) & 1)) { d |= (fo::det_t)1 << (o + sp * n_orb); placed++; } } }
            for (int sp = 0; sp < 2; sp++) { unsigned placed = 0; fo::det_t dd = 0; while (placed < n_elec / 2) { unsigned o = gb() % n_orb; if (!((dd >> (o + sp * n_orb)) & 1)) { dd |= (fo::det_t)1 << (o + sp * n_orb); placed++; } } }
            uint8_t db[8]; memcpy(db, &d, 8);
            fo::occ_list(d, occ);
            unsigned cr[n_irreps][2], co[fo::N_IRREPS][2];
            count_symm_virt(cr, occ, n_elec, &symm_basis);
            fo::count_symm_virt(co, occ, n_elec, fq.sys.symm);
            unsigned ns = 1 + trial % 7;
            uint8_t ro[64][4]; double rp[64];
            unsigned nr = doub_multin(db, occ, n_elec, &symm_basis, cr, ns, ga, ro, rp);
            unsigned no = 0; uint8_t oo[64][4]; double op[64];
            for (unsigned i = 0; i < ns; i++) if (fo::nu_doub_sample(d, occ, n_elec, fq.sys.symm, co, rb, oo[no], &op[no])) no++;
            CHECK(nr == no, "doub_multin count %u %u", nr, no);
            for (unsigned i = 0; i < std::min(nr, no); i++) CHECK(!memcmp(ro[i], oo[i], 4) && same_bits(rp[i], op[i]), "doub_multin sample");
            uint8_t rs[64][2]; double rsp[64];
            unsigned nsr = sing_multin(db, occ, n_elec, &symm_basis, cr, ns, ga, rs, rsp);
            unsigned m_allow[64], delta_s;
            fo::nu_sing_setup(occ, n_elec, fq.sys.symm, co, m_allow, &delta_s);
            unsigned nso = delta_s == n_elec ? 0 : ns;
            CHECK(nsr == nso, "sing_multin count");
            for (unsigned i = 0; i < nso; i++) { uint8_t so[2]; double sp2; fo::nu_sing_sample(d, occ, n_elec, fq.sys.symm, m_allow, delta_s, rb, so, &sp2); CHECK(!memcmp(rs[i], so, 2) && same_bits(rsp[i], sp2), "sing_multin sample"); }
            if (heat_bath && trial < 1500) {
                unsigned nh = 1 + trial % 23;
                uint8_t hr[64][4]; double hp[64];
                unsigned cr2 = hb_doub_multi(db, occ, n_elec, &symm_basis, hb_probs, nh, ga, hr, hp);
                uint8_t ho[64 * 4]; double hq[64]; uint32_t hatt[64];
                unsigned co2 = fo::hb_doub_multi(d, occ, n_elec, fq.sys.symm, fq.sys.hb, nh, rb, 0, ho, hq, hatt);
                if (getenv("FRIES_HB_DEBUG") && cr2 != co2) {
                    static int once = 0;
                    if (!once++) {
                        fprintf(stderr, "HBDBG trial %d det %016llx nh %u ref %u oracle %u\n", trial, (unsigned long long)d, nh, cr2, co2);
                        for (unsigned i = 0; i < cr2; i++) fprintf(stderr, "  ref %u: %u %u %u %u p %a\n", i, hr[i][0], hr[i][1], hr[i][2], hr[i][3], hp[i]);
                        for (unsigned i = 0; i < co2; i++) fprintf(stderr, "  orc %u: %u %u %u %u p %a att %x\n", i, ho[4 * i], ho[4 * i + 1], ho[4 * i + 2], ho[4 * i + 3], hq[i], hatt[i]);
                    }
                }
                CHECK(cr2 == co2, "hb_doub_multi count %u %u", cr2, co2);
                for (unsigned i = 0; i < std::min(cr2, co2); i++) CHECK(!memcmp(hr[i], &ho[4 * i], 4) && same_bits(hp[i], hq[i]), "hb_doub_multi sample %u", i);
            }
            unsigned nb = 1 + trial % 50; double pp = (trial % 97) / 97.0;
            CHECK(bin_sample(nb, pp, ga) == fo::bin_sample(nb, pp, rb), "bin_sample");
            double pr = -1.7 + (trial % 41) * 0.1;
            CHECK(round_binomially(pr, nb, ga) == fo::round_binomially(pr, nb, rb), "round_binomially");
        }
    }
    std::string out_name(argv[10]);
    if (n_procs > 1) out_name += ".r" + std::to_string(proc_rank);
    FILE *f = fopen(out_name.c_str(), "w");
    fprintf(f, "# golden trajectory from the reference (fciqmc_mol.cpp loop, %d rank(s), hf_proc %u); cols: it numer denom norm shift n_nonz n_ini curr_size n_spawn digest\n", n_procs, hf_proc);
    for (unsigned iterat = 0; iterat < n_iter; iterat++) {
        int n_nonz = 0; size_t n_ini = 0, n_spawn = 0;
        for (size_t det_idx = 0; det_idx < sol_vec.curr_size(); det_idx++) {
            int *curr_el = sol_vec[det_idx];
            uint8_t *curr_det = sol_vec.indices()[det_idx];
            unsigned n_walk = abs(*curr_el);
            if (n_walk == 0) continue;
            n_nonz++;
            int ini_flag = n_walk > init_thresh;
            n_ini += ini_flag;
            int walk_sign = 1 - ((*curr_el >> (sizeof(int) * 8 - 1)) & 2);
            uint8_t *occ_orbs = sol_vec.orbs_at_pos(det_idx);
            count_symm_virt(unocc_symm_cts, occ_orbs, n_elec_unf, &symm_basis);
            unsigned n_doub = bin_sample(n_walk, p_doub, mt_obj);
            unsigned n_sing = n_walk - n_doub;
            if (n_doub > max_spawn || n_sing > max_spawn) { fprintf(stderr, "harness: max_spawn exceeded\n"); return 2; }
            if (heat_bath) n_doub = hb_doub_multi(curr_det, occ_orbs, n_elec_unf, &symm_basis, hb_probs, n_doub, mt_obj, doub_orbs, spawn_probs.data());
            else n_doub = doub_multin(curr_det, occ_orbs, n_elec_unf, &symm_basis, unocc_symm_cts, n_doub, mt_obj, doub_orbs, spawn_probs.data());
            uint8_t new_det[8];
            for (size_t w = 0; w < n_doub; w++) {
                double matr_el = doub_matr_el_nosgn(doub_orbs[w], tot_orb, *eris, n_frz);
                matr_el *= eps / spawn_probs[w] / p_doub;
                int spawn_walker = round_binomially(matr_el, 1, mt_obj);
                if (spawn_walker != 0) {
                    memcpy(new_det, curr_det, det_size);
                    spawn_walker *= -doub_det_parity(new_det, doub_orbs[w]) * walk_sign;
                    if (!sol_vec.add(new_det, spawn_walker, ini_flag)) { fprintf(stderr, "adder full\n"); return 2; }
                    n_spawn++;
                }
            }
            n_sing = sing_multin(curr_det, occ_orbs, n_elec_unf, &symm_basis, unocc_symm_cts, n_sing, mt_obj, sing_orbs, spawn_probs.data());
            for (size_t w = 0; w < n_sing; w++) {
                double matr_el = sing_matr_el_nosgn(sing_orbs[w], occ_orbs, tot_orb, *eris, *h_core, n_frz, n_elec_unf);
                matr_el *= eps / spawn_probs[w] / (1 - p_doub);
                int spawn_walker = round_binomially(matr_el, 1, mt_obj);
                if (spawn_walker != 0) {
                    memcpy(new_det, curr_det, det_size);
                    spawn_walker *= -sing_det_parity(new_det, sing_orbs[w]) * walk_sign;
                    if (!sol_vec.add(new_det, spawn_walker, ini_flag)) { fprintf(stderr, "adder full\n"); return 2; }
                    n_spawn++;
                }
            }
            double diag_el = sol_vec.matr_el_at_pos(det_idx);
            double matr_el = (1 - eps * (diag_el - en_shift)) * walk_sign;
            int new_val = round_binomially(matr_el, n_walk, mt_obj);
            if (new_val == 0 && sol_vec.indices()[det_idx] != hf_det) sol_vec.del_at_pos(det_idx);
            *curr_el = new_val;
        }
        sol_vec.perform_add(0);
        double norm_out = 0;
        if ((iterat + 1) % 10 == 0) {
            glob_norm = sum_mpi(sol_vec.local_norm(), proc_rank, n_procs);
            (void)sum_mpi((int)n_nonz, proc_rank, n_procs);      // glob_nnonz for nnonz.txt (:422)
            adjust_shift(&en_shift, glob_norm, &last_norm, target_walkers, 0.05 / eps / 10);
            norm_out = glob_norm;
        }
        double numer = sol_vec.dot(htrial_vec.indices(), htrial_vec.values(), htrial_vec.curr_size(), htrial_hashes);
        double denom = sol_vec.dot(trial_vec.indices(), trial_vec.values(), trial_vec.curr_size(), trial_hashes);
        if (n_procs > 1) {      // gathered and added up in rank order on the rank that owns HF (:433-441); the others keep their own terms
            double rn[64], rd[64];
            MPI_Gather(&numer, 1, MPI_DOUBLE, rn, 1, MPI_DOUBLE, hf_proc, MPI_COMM_WORLD);
            MPI_Gather(&denom, 1, MPI_DOUBLE, rd, 1, MPI_DOUBLE, hf_proc, MPI_COMM_WORLD);
            if ((int)hf_proc == proc_rank) { numer = 0; denom = 0; for (int q = 0; q < n_procs; q++) { numer += rn[q]; denom += rd[q]; } }
        }
        uint64_t hsh = 1469598103934665603ull;
        for (size_t i = 0; i < sol_vec.curr_size(); i++) {
            double rv = sol_vec.values()[i];
            if (rv != 0) { uint64_t vb; memcpy(&vb, &rv, 8); fo::det_t rd = to_u64(sol_vec.indices()[i], det_size); hsh = (hsh ^ rd) * 1099511628211ull; hsh = (hsh ^ vb) * 1099511628211ull; hsh = (hsh ^ i) * 1099511628211ull; }
        }
        fprintf(f, "%u %a %a %a %a %d %zu %zu %zu %016" PRIx64 "\n", iterat, numer, denom, norm_out, en_shift, n_nonz, n_ini, (size_t)sol_vec.curr_size(), n_spawn, hsh);
        if (!lockstep) continue;
        fq.iterate(1);
        const fo::FciqmcLog &lg = fq.log.back();
        CHECK(same_bits(lg.numer, numer) && same_bits(lg.denom, denom), "fciqmc it %u numer/denom %a %a | %a %a", iterat, lg.numer, numer, lg.denom, denom);
        CHECK(same_bits(lg.norm, norm_out) && same_bits(lg.shift, en_shift), "fciqmc it %u norm/shift", iterat);
        CHECK(lg.n_nonz == n_nonz && lg.n_ini == n_ini && lg.curr_size == sol_vec.curr_size() && lg.n_spawn == n_spawn, "fciqmc it %u counts nnz %d/%d ini %u/%zu size %zu/%zu spawn %zu/%zu",
              iterat, lg.n_nonz, n_nonz, lg.n_ini, n_ini, lg.curr_size, (size_t)sol_vec.curr_size(), lg.n_spawn, n_spawn);
        size_t bad = 0, nmin = std::min(lg.curr_size, (size_t)sol_vec.curr_size());
        for (size_t i = 0; i < nmin; i++) {
            int rv = sol_vec.values()[i];
            if ((double)rv != fq.sol.vals[0][i]) bad++;
            if (rv != 0 && to_u64(sol_vec.indices()[i], det_size) != fq.sol.dets[i]) bad++;
        }
        CHECK(bad == 0, "fciqmc it %u vector mismatch in %zu slots", iterat, bad);
        if (getenv("FRIES_SAVE_DIR") && getenv("FRIES_SAVE_AT") && iterat + 1 == (unsigned)atoi(getenv("FRIES_SAVE_AT"))) {
            // what fciqmc_mol leaves in --result_dir: DistVec<int>::save (vec_utils.hpp:713-746), hash.dat (io_utils.cpp:589-606), S.txt
            const std::string dir(getenv("FRIES_SAVE_DIR"));
            sol_vec.save(dir);
            save_proc_hash(dir, proc_scrambler.data(), 2 * n_orb);
            std::ofstream sf(dir + "S.txt");
            sf << en_shift << "\n";
            double wn = 0; int nz = 0;
            for (size_t i = 0; i < sol_vec.curr_size(); i++) { wn += abs(sol_vec.values()[i]); nz += sol_vec.values()[i] != 0; }
            std::ofstream mf(dir + "meta.txt");
            mf << "iterations " << iterat + 1 << " curr_size " << sol_vec.curr_size() << " nonzero " << nz << " walkers " << (long long)wn << " shift_hex " << std::hexfloat << en_shift << "\n";
        }
    }
    fclose(f);
    if (n_procs > 1) { if (proc_rank == 0) printf("FCIQMC_MPI ranks=%d iters=%u hf_proc=%u\n", n_procs, n_iter, hf_proc); return 0; }
    printf("FCIQMC iters=%u checks=%d fails=%d final n_nonz=%d\n", n_iter, n_chk, n_fail, sol_vec.n_nonz());
    return n_fail != 0;
}

// ------------------------------------------------------------------ frimulti_mol: reference loop vs the oracle in mt mode
// ref_harness frimulti <fcidump> <pg> <n_iter> <seed> <eps> <vec_nonz> <mat_nonz> <max_dets> <initiator> <target> <out>
// FRIES_bin/frimulti_mol.cpp:84-425 with the FCIDUMP reader and SymmERIs in place of its legacy --hf_path / FourDArr inputs (the same
// library functions otherwise); --distribution HB, the only one its argument check lets through (:38-46).  One rank.
static int run_frimulti(int argc, char **argv) {
    if (argc < 13) { fprintf(stderr, "usage: see header\n"); return 2; }
    const char *path = argv[2], *pg = argv[3];
    unsigned n_iter = atoi(argv[4]); uint32_t seed = strtoul(argv[5], 0, 10);
    double eps = atof(argv[6]); uint32_t target_nonz = strtoul(argv[7], 0, 10), matr_samp = strtoul(argv[8], 0, 10);
    uint32_t max_n_dets = strtoul(argv[9], 0, 10); double init_thresh = atof(argv[10]), target_norm = atof(argv[11]);
    int n_procs = 1, proc_rank = 0;
    MPI_Comm_size(MPI_COMM_WORLD, &n_procs);
    MPI_Comm_rank(MPI_COMM_WORLD, &proc_rank);
    fcidump_input *in_data = parse_fcidump(path, pg);
    unsigned n_elec = in_data->n_elec, n_frz = 0, n_orb = in_data->n_orb_;
    size_t det_size = CEILING(2 * n_orb, 8);
    unsigned n_elec_unf = n_elec, tot_orb = n_orb;
    uint8_t *symm = in_data->symm;
    Matrix<double> *h_core = in_data->hcore; SymmERIs *eris = &in_data->eris;
    uint8_t tmp_orbs[64], hf_det[8] = {0};
    gen_hf_bitstring(n_orb, n_elec, hf_det);
    find_bits(hf_det, tmp_orbs, det_size);
    double hf_en = diag_matrel(tmp_orbs, tot_orb, *eris, *h_core, n_frz, n_elec);
    std::mt19937 mt_obj(seed + (uint32_t)proc_rank);       // one generator per process (:84-86)
    unsigned spawn_length = matr_samp * 2 / n_procs / n_procs;       // :89
    std::function<double(const uint8_t *)> diag_shortcut = [tot_orb, eris, h_core, n_frz, n_elec, hf_en](const uint8_t *occ) { return diag_matrel(occ, tot_orb, *eris, *h_core, n_frz, n_elec) - hf_en; };
    SymmInfo symm_basis(symm, n_orb);
    unsigned unocc_symm_cts[n_irreps][2];
    std::vector<uint32_t> proc_scrambler(2 * n_orb), vec_scrambler(2 * n_orb);
    if (proc_rank == 0) for (auto &x : proc_scrambler) x = mt_obj();
    MPI_Bcast(proc_scrambler.data(), 2 * n_orb, MPI_UNSIGNED, 0, MPI_COMM_WORLD);
    for (auto &x : vec_scrambler) x = mt_obj();
    DistVec<double> sol_vec(max_n_dets, spawn_length, n_orb * 2, n_elec_unf, n_procs, diag_shortcut, 1, proc_scrambler, vec_scrambler);
    unsigned max_spawn = matr_samp;
    std::vector<uint8_t> spawn_orbs_v(4 * (size_t)max_spawn + 64); std::vector<double> spawn_probs(max_spawn + 16);
    uint8_t (*sing_orbs)[2] = (uint8_t (*)[2])spawn_orbs_v.data();
    uint8_t (*doub_orbs)[4] = (uint8_t (*)[4])spawn_orbs_v.data();
    size_t n_ex = (size_t)n_orb * n_orb * n_elec_unf * n_elec_unf;
    // --trial_vec (frimulti_mol.cpp:139-163): the reference's own text reader, through the solution vector's arrays; a full Adder is an error there
    const char *trial_prefix = getenv("FRIES_TRIAL");
    size_t n_trial = 1;
    std::vector<fo::det_t> tin_det; std::vector<double> tin_val;
    if (trial_prefix) {
        n_trial = load_vec_txt(std::string(trial_prefix), sol_vec.indices(), sol_vec.values());
        for (size_t i = 0; i < n_trial; i++) { tin_det.push_back(to_u64(sol_vec.indices()[i], det_size)); tin_val.push_back(sol_vec.values()[i]); }
    }
    DistVec<double> trial_vec(trial_prefix ? n_trial : 4, trial_prefix ? n_trial : 4, n_orb * 2, n_elec_unf, n_procs, proc_scrambler, vec_scrambler);
    DistVec<double> htrial_vec(trial_prefix ? n_trial * n_ex / n_procs : 2 * n_ex, trial_prefix ? n_trial * n_ex / n_procs : 2 * n_ex, n_orb * 2, n_elec_unf, n_procs, diag_shortcut, 2, proc_scrambler, vec_scrambler);
    unsigned hf_proc = sol_vec.idx_to_proc(hf_det);
    if (trial_prefix) {
        Matrix<uint8_t> &load_dets = sol_vec.indices();
        double *load_vals = sol_vec.values();
        for (size_t i = 0; i < n_trial; i++) {
            if (!trial_vec.add(load_dets[i], load_vals[i], 1)) throw std::runtime_error("Insufficient memory allocated in adder");
            if (!htrial_vec.add(load_dets[i], load_vals[i], 1)) throw std::runtime_error("Insufficient memory allocated in adder");
        }
        memset(sol_vec.values(), 0, (n_trial + 1) * sizeof(double));
    }
    else if ((int)hf_proc == proc_rank) { trial_vec.add(hf_det, 1, 1); htrial_vec.add(hf_det, 1, 1); }
    trial_vec.perform_add(0); htrial_vec.perform_add(0);
    trial_vec.collect_procs();
    std::vector<uintmax_t> trial_hashes(trial_vec.curr_size());
    for (size_t i = 0; i < trial_vec.curr_size(); i++) trial_hashes[i] = sol_vec.idx_to_hash(trial_vec.indices()[i], tmp_orbs);
    std::vector<uint8_t> scratch(4 * n_ex);
    h_op_offdiag(htrial_vec, symm, tot_orb, *eris, *h_core, scratch.data(), scratch.size(), n_frz, n_elec_unf, 1, 1, 0);
    htrial_vec.set_curr_vec_idx(0);
    h_op_diag(htrial_vec, 0, 0, 1);
    htrial_vec.add_vecs(0, 1);
    htrial_vec.collect_procs();
    std::vector<uintmax_t> htrial_hashes(htrial_vec.curr_size());
    for (size_t i = 0; i < htrial_vec.curr_size(); i++) htrial_hashes[i] = sol_vec.idx_to_hash(htrial_vec.indices()[i], tmp_orbs);
    sol_vec.gen_orb_list(hf_det, tmp_orbs);
    size_t n_hf_doub = doub_ex_symm(hf_det, tmp_orbs, n_elec_unf, n_orb, (uint8_t (*)[4])scratch.data(), symm);
    size_t n_hf_sing = count_singex(hf_det, tmp_orbs, n_elec_unf, &symm_basis);
    double p_doub = (double)n_hf_doub / (n_hf_sing + n_hf_doub);
    std::vector<fo::det_t> iin_det; std::vector<double> iin_val;
    if (getenv("FRIES_INI")) {       // --ini_vec (frimulti_mol.cpp:205-215): plain add() calls
        Matrix<uint8_t> &load_dets = sol_vec.indices();
        double *load_vals = sol_vec.values();
        size_t n_dets = load_vec_txt(std::string(getenv("FRIES_INI")), load_dets, load_vals);
        std::vector<std::vector<uint8_t>> keep_d(n_dets, std::vector<uint8_t>(det_size)); std::vector<double> keep_v(n_dets);
        for (size_t i = 0; i < n_dets; i++) { memcpy(keep_d[i].data(), load_dets[i], det_size); keep_v[i] = load_vals[i]; iin_det.push_back(to_u64(load_dets[i], det_size)); iin_val.push_back(load_vals[i]); }
        memset(load_vals, 0, (n_dets + 1) * sizeof(double));       // the reference adds straight out of the vector's own arrays; keep the harness well defined
        for (size_t i = 0; i < n_dets; i++) sol_vec.add(keep_d[i].data(), keep_v[i], 1);
    }
    else if ((int)hf_proc == proc_rank) sol_vec.add(hf_det, 100, 1);
    sol_vec.perform_add(0);
    double loc_norms[64], glob_norm;
    loc_norms[proc_rank] = sol_vec.local_norm();
    MPI_Allgather(MPI_IN_PLACE, 0, MPI_DOUBLE, loc_norms, 1, MPI_DOUBLE, MPI_COMM_WORLD);
    glob_norm = 0;
    for (int q = 0; q < n_procs; q++) glob_norm += loc_norms[q];
    const bool lockstep = n_procs == 1;
    hb_info *hb_probs = set_up(tot_orb, n_orb, *eris);

    fo::Fciqmc fq;
    fq.sys.n_orb = n_orb; fq.sys.n_elec = n_elec;
    fill_oracle_ints(fq.sys.ints, *eris, *h_core, n_orb);
    fq.sys.symm.init(symm, n_orb);
    fq.par.eps = eps; fq.par.max_dets = max_n_dets; fq.par.seed = seed; fq.par.counter_rng = false; fq.par.heat_bath = true;
    fq.par.multi = true; fq.par.vec_nonz = target_nonz; fq.par.mat_nonz = matr_samp; fq.par.target_norm = target_norm; fq.par.init_thresh_f = init_thresh;
    fq.trial_in_det = tin_det; fq.trial_in_val = tin_val; fq.ini_det = iin_det; fq.ini_val = iin_val;
    if (lockstep) { fq.setup(); CHECK(same_bits(fq.p_doub, p_doub), "frimulti p_doub"); }

    double en_shift = 0, last_one_norm = 0;
    const double shift_damping = 0.05; const unsigned shift_interval = 10;
    std::vector<size_t> srt_arr(max_n_dets); std::vector<bool> keep_exact(max_n_dets, false);
    std::string out_name(argv[12]);
    if (n_procs > 1) out_name += ".r" + std::to_string(proc_rank);
    FILE *f = fopen(out_name.c_str(), "w");
    fprintf(f, "# golden trajectory from the reference's frimulti_mol loop (HB; per rank under mpiexec); cols: it numer denom norm shift nkept n_nonz curr_size n_spawn n_ini digest\n");
    for (unsigned iterat = 0; iterat < n_iter; iterat++) {
        size_t n_ini = 0, n_spawn = 0;
        double rn_sys = 0;
        if (proc_rank == 0) rn_sys = mt_obj() / (1. + UINT32_MAX);
        MPI_Bcast(&rn_sys, 1, MPI_DOUBLE, 0, MPI_COMM_WORLD);
        unsigned curr_mat_samp = (iterat < 10) ? matr_samp / 10 : matr_samp;
        double lbound = seed_sys(loc_norms, &rn_sys, curr_mat_samp);
        for (size_t det_idx = 0; det_idx < sol_vec.curr_size(); det_idx++) {
            double *curr_el = sol_vec[det_idx];
            uint8_t *curr_det = sol_vec.indices()[det_idx];
            double weight = fabs(*curr_el);
            if (weight == 0) continue;
            unsigned n_walk = 0;
            lbound += weight;
            while (rn_sys < lbound) { n_walk++; rn_sys += glob_norm / (curr_mat_samp); }
            double colsamp_wt = weight / (glob_norm / curr_mat_samp);
            if (colsamp_wt > 1) colsamp_wt = 1;
            int ini_flag = weight > init_thresh;
            n_ini += ini_flag;
            uint8_t *occ_orbs = sol_vec.orbs_at_pos(det_idx);
            count_symm_virt(unocc_symm_cts, occ_orbs, n_elec_unf, &symm_basis);
            unsigned n_doub = bin_sample(n_walk, p_doub, mt_obj);
            unsigned n_sing = n_walk - n_doub;
            if (n_doub > max_spawn || n_sing / 2 > max_spawn) { fprintf(stderr, "harness: max_spawn exceeded\n"); return 2; }
            n_doub = hb_doub_multi(curr_det, occ_orbs, n_elec_unf, &symm_basis, hb_probs, n_doub, mt_obj, doub_orbs, spawn_probs.data());
            uint8_t new_det[8];
            for (size_t w = 0; w < n_doub; w++) {
                double matr_el = doub_matr_el_nosgn(doub_orbs[w], tot_orb, *eris, n_frz);
                if (fabs(matr_el) > 1e-9) {
                    memcpy(new_det, curr_det, det_size);
                    matr_el *= -eps / spawn_probs[w] / p_doub / n_walk * (*curr_el) * doub_det_parity(new_det, doub_orbs[w]) / colsamp_wt;
                    if (!sol_vec.add(new_det, matr_el, ini_flag)) { fprintf(stderr, "harness: adder full\n"); return 2; }
                    n_spawn++;
                }
            }
            n_sing = sing_multin(curr_det, occ_orbs, n_elec_unf, &symm_basis, unocc_symm_cts, n_sing, mt_obj, sing_orbs, spawn_probs.data());
            for (size_t w = 0; w < n_sing; w++) {
                double matr_el = sing_matr_el_nosgn(sing_orbs[w], occ_orbs, tot_orb, *eris, *h_core, n_frz, n_elec_unf);
                if (fabs(matr_el) > 1e-9) {
                    memcpy(new_det, curr_det, det_size);
                    matr_el *= -eps / spawn_probs[w] / (1 - p_doub) / n_walk * (*curr_el) * sing_det_parity(new_det, sing_orbs[w]) / colsamp_wt;
                    if (!sol_vec.add(new_det, matr_el, ini_flag)) { fprintf(stderr, "harness: adder full\n"); return 2; }
                    n_spawn++;
                }
            }
            double diag_el = sol_vec.matr_el_at_pos(det_idx);
            *curr_el *= 1 - eps * (diag_el - en_shift);
        }
        sol_vec.perform_add(0);
        unsigned n_samp = target_nonz;
        loc_norms[proc_rank] = find_preserve(sol_vec.values(), srt_arr, keep_exact, sol_vec.curr_size(), &n_samp, &glob_norm);
        unsigned nkept = target_nonz - n_samp;
        if ((iterat + 1) % shift_interval == 0) adjust_shift(&en_shift, glob_norm, &last_one_norm, target_norm, shift_damping / shift_interval / eps);
        double numer = sol_vec.dot(htrial_vec.indices(), htrial_vec.values(), htrial_vec.curr_size(), htrial_hashes);
        double denom = sol_vec.dot(trial_vec.indices(), trial_vec.values(), trial_vec.curr_size(), trial_hashes);
        if (n_procs > 1) {      // :399-409: gathered and added up in rank order on the rank that owns HF
            double rnm[64], rdn[64];
            MPI_Gather(&numer, 1, MPI_DOUBLE, rnm, 1, MPI_DOUBLE, hf_proc, MPI_COMM_WORLD);
            MPI_Gather(&denom, 1, MPI_DOUBLE, rdn, 1, MPI_DOUBLE, hf_proc, MPI_COMM_WORLD);
            if ((int)hf_proc == proc_rank) { numer = 0; denom = 0; for (int q = 0; q < n_procs; q++) { numer += rnm[q]; denom += rdn[q]; } }
        }
        if (proc_rank == 0) rn_sys = mt_obj() / (1. + UINT32_MAX);
        MPI_Bcast(&rn_sys, 1, MPI_DOUBLE, 0, MPI_COMM_WORLD);        // (the reference relies on sys_comp's own broadcast, compress_utils.cpp:291)
        MPI_Allgather(MPI_IN_PLACE, 0, MPI_DOUBLE, loc_norms, 1, MPI_DOUBLE, MPI_COMM_WORLD);
        sys_comp(sol_vec.values(), sol_vec.curr_size(), loc_norms, n_samp, keep_exact, rn_sys);
        for (size_t det_idx = 0; det_idx < sol_vec.curr_size(); det_idx++) {
            if (keep_exact[det_idx] && sol_vec.indices()[det_idx] != hf_det) {        // an address comparison, as in the reference (:417)
                sol_vec.del_at_pos(det_idx);
                keep_exact[det_idx] = 0;
            }
        }
        if (!lockstep) {
            uint64_t hs2 = 1469598103934665603ull;
            for (size_t i = 0; i < sol_vec.curr_size(); i++) {
                double rv = sol_vec.values()[i];
                if (rv != 0) { uint64_t vb; memcpy(&vb, &rv, 8); fo::det_t rd = to_u64(sol_vec.indices()[i], det_size); hs2 = (hs2 ^ rd) * 1099511628211ull; hs2 = (hs2 ^ vb) * 1099511628211ull; hs2 = (hs2 ^ i) * 1099511628211ull; }
            }
            fprintf(f, "%u %a %a %a %a %u %d %zu %zu %zu %016" PRIx64 "\n", iterat, numer, denom, glob_norm, en_shift, nkept, sol_vec.n_nonz(), (size_t)sol_vec.curr_size(), n_spawn, n_ini, hs2);
            continue;
        }
        fq.iterate_multi(1);
        const fo::FciqmcLog &lg = fq.log.back();
        CHECK(same_bits(lg.numer, numer) && same_bits(lg.denom, denom), "frimulti it %u numer/denom %a %a | %a %a", iterat, lg.numer, numer, lg.denom, denom);
        CHECK(same_bits(lg.norm, glob_norm) && same_bits(lg.shift, en_shift), "frimulti it %u norm/shift", iterat);
        CHECK(fq.nkept == nkept && lg.n_nonz == sol_vec.n_nonz() && lg.curr_size == sol_vec.curr_size() && lg.n_spawn == n_spawn && lg.n_ini == n_ini,
              "frimulti it %u counts nkept %u/%u nnz %d/%d size %zu/%zu spawn %zu/%zu ini %u/%zu", iterat, fq.nkept, nkept, lg.n_nonz, sol_vec.n_nonz(), lg.curr_size,
              (size_t)sol_vec.curr_size(), lg.n_spawn, n_spawn, lg.n_ini, n_ini);
        size_t bad = 0, nmin = std::min(lg.curr_size, (size_t)sol_vec.curr_size());
        uint64_t hsh = 1469598103934665603ull;
        for (size_t i = 0; i < nmin; i++) {
            double rv = sol_vec.values()[i];
            fo::det_t rd = to_u64(sol_vec.indices()[i], det_size);
            if (!same_bits(rv, fq.sol.vals[0][i])) bad++;
            if (rv != 0 && rd != fq.sol.dets[i]) bad++;
            if (rv != 0) { uint64_t vb; memcpy(&vb, &rv, 8); hsh = (hsh ^ rd) * 1099511628211ull; hsh = (hsh ^ vb) * 1099511628211ull; hsh = (hsh ^ i) * 1099511628211ull; }
        }
        CHECK(bad == 0, "frimulti it %u vector mismatch in %zu slots", iterat, bad);
        fprintf(f, "%u %a %a %a %a %u %d %zu %zu %zu %016" PRIx64 "\n", iterat, numer, denom, glob_norm, en_shift, nkept, sol_vec.n_nonz(), (size_t)sol_vec.curr_size(), n_spawn, n_ini, hsh);
    }
    fclose(f);
    if (proc_rank == 0) printf("FRIMULTI ranks=%d hf_proc=%u iters=%u checks=%d fails=%d final n_nonz=%d\n", n_procs, hf_proc, n_iter, n_chk, n_fail, sol_vec.n_nonz());
    return n_fail != 0;
}

// ------------------------------------------------------------------ fciqmc_fp_mol: reference loop vs the oracle in mt mode
// ref_harness fciqmc_fp <fcidump> <pg> <n_iter> <seed> <eps> <target_walkers> <max_dets> <initiator> <out> [NU|HB]
// FRIES_bin/fciqmc_fp_mol.cpp:100-480 (real-valued walkers), HF trial vector, start from 100 x HF, one rank.
static int run_fciqmc_fp(int argc, char **argv) {
    if (argc < 11) { fprintf(stderr, "usage: see header\n"); return 2; }
    const char *path = argv[2], *pg = argv[3];
    unsigned n_iter = atoi(argv[4]); uint32_t seed = strtoul(argv[5], 0, 10);
    double eps = atof(argv[6]); uint32_t target_walkers = strtoul(argv[7], 0, 10);
    uint32_t max_n_dets = strtoul(argv[8], 0, 10); uint32_t init_thresh = strtoul(argv[9], 0, 10);
    const bool heat_bath = argc > 11 && !strcmp(argv[11], "HB");
    int n_procs = 1, proc_rank = 0;
    MPI_Comm_size(MPI_COMM_WORLD, &n_procs);
    MPI_Comm_rank(MPI_COMM_WORLD, &proc_rank);
    fcidump_input *in_data = parse_fcidump(path, pg);
    unsigned n_elec = in_data->n_elec, n_frz = 0, n_orb = in_data->n_orb_;
    size_t det_size = CEILING(2 * n_orb, 8);
    unsigned n_elec_unf = n_elec, tot_orb = n_orb;
    uint8_t *symm = in_data->symm;
    Matrix<double> *h_core = in_data->hcore; SymmERIs *eris = &in_data->eris;
    uint8_t tmp_orbs[64], hf_det[8] = {0};
    gen_hf_bitstring(n_orb, n_elec, hf_det);
    find_bits(hf_det, tmp_orbs, det_size);
    double hf_en = diag_matrel(tmp_orbs, tot_orb, *eris, *h_core, n_frz, n_elec);
    std::mt19937 mt_obj(seed + (uint32_t)proc_rank);      // one generator per process (fciqmc_fp_mol.cpp:108-110 seeds each from its clock)
    unsigned spawn_length = target_walkers / n_procs / n_procs * 2;
    std::function<double(const uint8_t *)> diag_shortcut = [tot_orb, eris, h_core, n_frz, n_elec, hf_en](const uint8_t *occ) { return diag_matrel(occ, tot_orb, *eris, *h_core, n_frz, n_elec) - hf_en; };
    SymmInfo symm_basis(symm, n_orb);
    unsigned unocc_symm_cts[n_irreps][2];
    std::vector<uint32_t> proc_scrambler(2 * n_orb), vec_scrambler(2 * n_orb);
    if (proc_rank == 0) for (auto &x : proc_scrambler) x = mt_obj();
    MPI_Bcast(proc_scrambler.data(), 2 * n_orb, MPI_UNSIGNED, 0, MPI_COMM_WORLD);
    for (auto &x : vec_scrambler) x = mt_obj();
    DistVec<double> sol_vec(max_n_dets, spawn_length, n_orb * 2, n_elec_unf, n_procs, diag_shortcut, 1, proc_scrambler, vec_scrambler);
    unsigned max_spawn = 500000;
    std::vector<uint8_t> spawn_orbs_v(4 * (size_t)max_spawn); std::vector<double> spawn_probs(max_spawn);
    uint8_t (*sing_orbs)[2] = (uint8_t (*)[2])spawn_orbs_v.data();
    uint8_t (*doub_orbs)[4] = (uint8_t (*)[4])spawn_orbs_v.data();
    size_t n_ex = (size_t)n_orb * n_orb * n_elec_unf * n_elec_unf;
    // --trial_vec (fciqmc_fp_mol.cpp:157-185): the reference's own text reader, through the solution vector's arrays, `while (!add) perform_add`
    const char *trial_prefix = getenv("FRIES_TRIAL");
    size_t n_trial = 1;
    std::vector<fo::det_t> tin_det; std::vector<double> tin_val;
    if (trial_prefix) {
        n_trial = load_vec_txt(std::string(trial_prefix), sol_vec.indices(), sol_vec.values());
        for (size_t i = 0; i < n_trial; i++) { tin_det.push_back(to_u64(sol_vec.indices()[i], det_size)); tin_val.push_back(sol_vec.values()[i]); }
    }
    DistVec<double> trial_vec(trial_prefix ? n_trial : 4, trial_prefix ? n_trial : 4, n_orb * 2, n_elec_unf, n_procs, proc_scrambler, vec_scrambler);
    DistVec<double> htrial_vec(trial_prefix ? n_trial * n_ex / n_procs : 2 * n_ex, trial_prefix ? n_trial * n_ex / n_procs : 2 * n_ex, n_orb * 2, n_elec_unf, n_procs, diag_shortcut, 2, proc_scrambler, vec_scrambler);
    unsigned hf_proc = sol_vec.idx_to_proc(hf_det);
    if (trial_prefix) {
        Matrix<uint8_t> &load_dets = sol_vec.indices();
        double *load_vals = sol_vec.values();
        for (size_t i = 0; i < n_trial; i++) {
            while (!trial_vec.add(load_dets[i], load_vals[i], 1)) trial_vec.perform_add(0);
            while (!htrial_vec.add(load_dets[i], load_vals[i], 1)) htrial_vec.perform_add(0);
        }
        memset(sol_vec.values(), 0, (n_trial + 1) * sizeof(double));
    }
    else if ((int)hf_proc == proc_rank) { trial_vec.add(hf_det, 1, 1); htrial_vec.add(hf_det, 1, 1); }
    trial_vec.perform_add(0); htrial_vec.perform_add(0);
    trial_vec.collect_procs();
    std::vector<uintmax_t> trial_hashes(trial_vec.curr_size());
    for (size_t i = 0; i < trial_vec.curr_size(); i++) trial_hashes[i] = sol_vec.idx_to_hash(trial_vec.indices()[i], tmp_orbs);
    h_op_offdiag(htrial_vec, symm, tot_orb, *eris, *h_core, spawn_orbs_v.data(), 4 * max_spawn, n_frz, n_elec_unf, 1, 1, 0);
    htrial_vec.set_curr_vec_idx(0);
    h_op_diag(htrial_vec, 0, 0, 1);
    htrial_vec.add_vecs(0, 1);
    htrial_vec.collect_procs();
    std::vector<uintmax_t> htrial_hashes(htrial_vec.curr_size());
    for (size_t i = 0; i < htrial_vec.curr_size(); i++) htrial_hashes[i] = sol_vec.idx_to_hash(htrial_vec.indices()[i], tmp_orbs);
    sol_vec.gen_orb_list(hf_det, tmp_orbs);
    size_t n_hf_doub = doub_ex_symm(hf_det, tmp_orbs, n_elec_unf, n_orb, doub_orbs, symm);
    size_t n_hf_sing = count_singex(hf_det, tmp_orbs, n_elec_unf, &symm_basis);
    double p_doub = (double)n_hf_doub / (n_hf_sing + n_hf_doub);
    std::vector<fo::det_t> iin_det; std::vector<double> iin_val;
    if (getenv("FRIES_INI")) {       // --ini_vec (fciqmc_fp_mol.cpp:233-246): real values, `while (!add) perform_add`
        Matrix<uint8_t> &load_dets = sol_vec.indices();
        double *load_vals = sol_vec.values();
        size_t n_dets = load_vec_txt(std::string(getenv("FRIES_INI")), load_dets, load_vals);
        std::vector<std::vector<uint8_t>> keep_d(n_dets, std::vector<uint8_t>(det_size)); std::vector<double> keep_v(n_dets);
        for (size_t i = 0; i < n_dets; i++) { memcpy(keep_d[i].data(), load_dets[i], det_size); keep_v[i] = load_vals[i]; iin_det.push_back(to_u64(load_dets[i], det_size)); iin_val.push_back(load_vals[i]); }
        memset(load_vals, 0, (n_dets + 1) * sizeof(double));       // the reference adds straight out of the vector's own arrays; keep the harness well defined
        for (size_t i = 0; i < n_dets; i++) while (!sol_vec.add(keep_d[i].data(), keep_v[i], 1)) sol_vec.perform_add(0);
    }
    else if ((int)hf_proc == proc_rank) sol_vec.add(hf_det, 100, 1);
    sol_vec.perform_add(0);
    double en_shift = 0, last_norm = 0, glob_norm = 0;
    const bool lockstep = n_procs == 1;
    hb_info *hb_probs = heat_bath ? set_up(tot_orb, n_orb, *eris) : NULL;

    fo::Fciqmc fq;
    fq.sys.n_orb = n_orb; fq.sys.n_elec = n_elec;
    fill_oracle_ints(fq.sys.ints, *eris, *h_core, n_orb);
    fq.sys.symm.init(symm, n_orb);
    fq.par.eps = eps; fq.par.target_walkers = target_walkers; fq.par.init_thresh = init_thresh; fq.par.max_dets = max_n_dets; fq.par.seed = seed; fq.par.counter_rng = false;
    fq.par.heat_bath = heat_bath; fq.par.fp = true;
    fq.trial_in_det = tin_det; fq.trial_in_val = tin_val; fq.ini_det = iin_det; fq.ini_val = iin_val;
    if (lockstep) { fq.setup(); CHECK(same_bits(fq.p_doub, p_doub), "fciqmc_fp p_doub"); }

    std::string out_name(argv[10]);
    if (n_procs > 1) out_name += ".r" + std::to_string(proc_rank);
    FILE *f = fopen(out_name.c_str(), "w");
    fprintf(f, "# golden trajectory from the reference's fciqmc_fp_mol loop (%d rank(s), %s); cols: it numer denom norm shift n_nonz n_ini curr_size n_spawn digest\n", n_procs, heat_bath ? "HB" : "NU");
    for (unsigned iterat = 0; iterat < n_iter; iterat++) {
        int n_nonz = 0; size_t n_ini = 0, n_spawn = 0;
        for (size_t det_idx = 0; det_idx < sol_vec.curr_size(); det_idx++) {
            double *curr_el = sol_vec[det_idx];
            uint8_t *curr_det = sol_vec.indices()[det_idx];
            unsigned n_walk = round_binomially(fabs(*curr_el), 1, mt_obj);
            if (n_walk == 0) continue;
            n_nonz++;
            int ini_flag = n_walk > init_thresh;
            n_ini += ini_flag;
            int walk_sign = 1 - 2 * (*curr_el < 0);
            uint8_t *occ_orbs = sol_vec.orbs_at_pos(det_idx);
            count_symm_virt(unocc_symm_cts, occ_orbs, n_elec_unf, &symm_basis);
            unsigned n_doub = bin_sample(n_walk, p_doub, mt_obj);
            unsigned n_sing = n_walk - n_doub;
            if (n_doub > max_spawn || n_sing > max_spawn) { fprintf(stderr, "harness: max_spawn exceeded\n"); return 2; }
            if (heat_bath) n_doub = hb_doub_multi(curr_det, occ_orbs, n_elec_unf, &symm_basis, hb_probs, n_doub, mt_obj, doub_orbs, spawn_probs.data());
            else n_doub = doub_multin(curr_det, occ_orbs, n_elec_unf, &symm_basis, unocc_symm_cts, n_doub, mt_obj, doub_orbs, spawn_probs.data());
            uint8_t new_det[8];
            for (size_t w = 0; w < n_doub; w++) {
                double matr_el = doub_matr_el_nosgn(doub_orbs[w], tot_orb, *eris, n_frz);
                matr_el *= eps / spawn_probs[w] / p_doub;
                double spawn_walker;
                if (fabs(matr_el) < 0.01) spawn_walker = round_binomially(matr_el, 1, mt_obj);
                else spawn_walker = matr_el;
                if (spawn_walker != 0) {
                    memcpy(new_det, curr_det, det_size);
                    spawn_walker *= -doub_det_parity(new_det, doub_orbs[w]) * walk_sign;
                    if (!sol_vec.add(new_det, spawn_walker, ini_flag)) { fprintf(stderr, "adder full\n"); return 2; }
                    n_spawn++;
                }
            }
            n_sing = sing_multin(curr_det, occ_orbs, n_elec_unf, &symm_basis, unocc_symm_cts, n_sing, mt_obj, sing_orbs, spawn_probs.data());
            for (size_t w = 0; w < n_sing; w++) {
                double matr_el = sing_matr_el_nosgn(sing_orbs[w], occ_orbs, tot_orb, *eris, *h_core, n_frz, n_elec_unf);
                matr_el *= eps / spawn_probs[w] / (1 - p_doub);
                double spawn_walker;
                if (fabs(matr_el) < 0.01) spawn_walker = round_binomially(matr_el, 1, mt_obj);
                else spawn_walker = matr_el;
                if (spawn_walker != 0) {
                    memcpy(new_det, curr_det, det_size);
                    spawn_walker *= -sing_det_parity(new_det, sing_orbs[w]) * walk_sign;
                    if (!sol_vec.add(new_det, spawn_walker, ini_flag)) { fprintf(stderr, "adder full\n"); return 2; }
                    n_spawn++;
                }
            }
            double diag_el = sol_vec.matr_el_at_pos(det_idx);
            *curr_el *= 1 - eps * (diag_el - en_shift);
        }
        sol_vec.perform_add(0);
        for (size_t det_idx = 0; det_idx < sol_vec.curr_size(); det_idx++) {       // :428-441
            double *curr_el = sol_vec[det_idx];
            if (*curr_el == 0) continue;
            if (fabs(*curr_el) < 1) *curr_el = round_binomially(*curr_el, 1, mt_obj);
            if (*curr_el == 0) sol_vec.del_at_pos(det_idx);
        }
        double norm_out = 0;
        if ((iterat + 1) % 10 == 0) {
            glob_norm = sum_mpi(sol_vec.local_norm(), proc_rank, n_procs);
            adjust_shift(&en_shift, glob_norm, &last_norm, target_walkers, 0.05 / eps / 10);
            norm_out = glob_norm;
        }
        double numer = sol_vec.dot(htrial_vec.indices(), htrial_vec.values(), htrial_vec.curr_size(), htrial_hashes);
        double denom = sol_vec.dot(trial_vec.indices(), trial_vec.values(), trial_vec.curr_size(), trial_hashes);
        if (n_procs > 1) {      // :456-470: gathered on the rank that owns HF -- and slot 0 overwritten with that rank's own terms (:461-462)
            double rn[64], rd[64];
            MPI_Gather(&numer, 1, MPI_DOUBLE, rn, 1, MPI_DOUBLE, hf_proc, MPI_COMM_WORLD);
            MPI_Gather(&denom, 1, MPI_DOUBLE, rd, 1, MPI_DOUBLE, hf_proc, MPI_COMM_WORLD);
            rn[0] = numer; rd[0] = denom;
            if ((int)hf_proc == proc_rank) { numer = 0; denom = 0; for (int q = 0; q < n_procs; q++) { numer += rn[q]; denom += rd[q]; } }
        }
        uint64_t hsh = 1469598103934665603ull;
        for (size_t i = 0; i < sol_vec.curr_size(); i++) {
            double rv = sol_vec.values()[i];
            if (rv != 0) { uint64_t vb; memcpy(&vb, &rv, 8); fo::det_t rd = to_u64(sol_vec.indices()[i], det_size); hsh = (hsh ^ rd) * 1099511628211ull; hsh = (hsh ^ vb) * 1099511628211ull; hsh = (hsh ^ i) * 1099511628211ull; }
        }
        fprintf(f, "%u %a %a %a %a %d %zu %zu %zu %016" PRIx64 "\n", iterat, numer, denom, norm_out, en_shift, n_nonz, n_ini, (size_t)sol_vec.curr_size(), n_spawn, hsh);
        if (!lockstep) continue;
        fq.iterate(1);
        const fo::FciqmcLog &lg = fq.log.back();
        CHECK(same_bits(lg.numer, numer) && same_bits(lg.denom, denom), "fciqmc_fp it %u numer/denom %a %a | %a %a", iterat, lg.numer, numer, lg.denom, denom);
        CHECK(same_bits(lg.norm, norm_out) && same_bits(lg.shift, en_shift), "fciqmc_fp it %u norm/shift %a %a", iterat, lg.norm, norm_out);
        CHECK(lg.n_nonz == n_nonz && lg.n_ini == n_ini && lg.curr_size == sol_vec.curr_size() && lg.n_spawn == n_spawn, "fciqmc_fp it %u counts nnz %d/%d ini %u/%zu size %zu/%zu spawn %zu/%zu",
              iterat, lg.n_nonz, n_nonz, lg.n_ini, n_ini, lg.curr_size, (size_t)sol_vec.curr_size(), lg.n_spawn, n_spawn);
        size_t bad = 0, nmin = std::min(lg.curr_size, (size_t)sol_vec.curr_size());
        for (size_t i = 0; i < nmin; i++) {
            double rv = sol_vec.values()[i];
            if (!same_bits(rv, fq.sol.vals[0][i])) bad++;
            if (rv != 0 && to_u64(sol_vec.indices()[i], det_size) != fq.sol.dets[i]) bad++;
        }
        CHECK(bad == 0, "fciqmc_fp it %u vector mismatch in %zu slots", iterat, bad);
    }
    fclose(f);
    if (proc_rank == 0) printf("FCIQMC_FP ranks=%d hf_proc=%u iters=%u checks=%d fails=%d final n_nonz=%d\n", n_procs, hf_proc, n_iter, n_chk, n_fail, sol_vec.n_nonz());
    return n_fail != 0;
}

int main(int argc, char **argv) {
    MPI_Init(NULL, NULL);
    int rc = 2;
    if (argc >= 2 && !strcmp(argv[1], "unit")) rc = run_unit();
    else if (argc >= 3 && !strcmp(argv[1], "hbpp_all")) rc = run_hbpp_all(argv[2]);
    else if (argc >= 3 && !strcmp(argv[1], "piv")) rc = run_piv(argv[2]);
    else if (argc >= 2 && !strcmp(argv[1], "hbpiv")) rc = run_hbpiv(argc, argv);
    else if (argc >= 2 && !strcmp(argv[1], "frimulti")) rc = run_frimulti(argc, argv);
    else if (argc >= 2 && !strcmp(argv[1], "fciqmc_fp")) rc = run_fciqmc_fp(argc, argv);
    else if (argc >= 2 && !strcmp(argv[1], "frifull")) rc = run_frifull(argc, argv);
    else if (argc >= 2 && !strcmp(argv[1], "frisys")) rc = run_frisys(argc, argv, false);
    else if (argc >= 2 && !strcmp(argv[1], "time")) rc = run_frisys(argc, argv, true);
    else if (argc >= 2 && !strcmp(argv[1], "frisys_mpi")) rc = run_frisys_mpi(argc, argv);
    else if (argc >= 2 && !strcmp(argv[1], "frifull_mpi")) rc = run_frifull_mpi(argc, argv);
    else if (argc >= 2 && !strcmp(argv[1], "restart")) rc = run_restart(argc, argv);
    else if (argc >= 2 && !strcmp(argv[1], "pin")) rc = run_pin(argc, argv);
    else if (argc >= 2 && !strcmp(argv[1], "reload")) rc = run_reload(argc, argv);
    else if (argc >= 2 && !strcmp(argv[1], "tr")) rc = run_tr(argc, argv);
    else if (argc >= 2 && !strcmp(argv[1], "hh")) rc = run_hh(argc, argv);
    else if (argc >= 2 && !strcmp(argv[1], "fciqmc")) rc = run_fciqmc(argc, argv);
    else if (argc >= 5 && !strcmp(argv[1], "dump_ints")) rc = run_dump_ints(argv[2], argv[3], argv[4]);
    else if (argc >= 4 && !strcmp(argv[1], "hfdir")) rc = run_hfdir(argv[2], argv[3]);
    else fprintf(stderr, "unknown command\n");
    MPI_Finalize();
    return rc;
}
