// CPU test of the HOST logic in include/FRIES (no device call is made): bit strings, fermionic signs, excitation lists in the reference's
// order, the rank / vector hash, the host DistVec + Adder (hash of determinant -> position, LIFO stack of freed positions, initiator rule,
// the order of additions), Matrix / SymmERIs, adjust_shift and the command-line parser -- against the oracle's restatement (oracle/, pinned
// by the reference) on random inputs.  Built and run by tests/test_cpu_oracle.py::test_fries_headers_host_logic.
#include <FRIES/Hamiltonians/molecule.hpp>
#include <FRIES/Ext_Libs/argparse.hpp>
#include <FRIES/compress_utils.hpp>
#include "fries_oracle.hpp"
#include <cstdio>
#include <random>

static int n_fail = 0, n_chk = 0;
#define CHECK(c, ...) do { n_chk++; if (!(c)) { if (n_fail < 20) { printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } n_fail++; } } while (0)

static fo::det_t rand_det(std::mt19937 &mt, unsigned n_orb, unsigned n_elec) {
    fo::det_t d = 0;
    for (int spin = 0; spin < 2; spin++) {
        unsigned placed = 0;
        while (placed < n_elec / 2) { unsigned o = mt() % n_orb + spin * n_orb; if (!((d >> o) & 1)) { d |= (fo::det_t)1 << o; placed++; } }
    }
    return d;
}

struct MyArgs : public argparse::Args {
    std::string &path = kwarg("fcidump_path", "file");
    double &target = kwarg("target", "norm").set_default(0);
    uint32_t &iters = kwarg("max_iter", "iterations").set_default(1000000);
    std::string &dir = kwarg("result_dir", "dir").set_default<std::string>("./");
    std::shared_ptr<std::string> &load = kwarg("load_dir", "checkpoint");
    std::shared_ptr<double> &shift = kwarg("ham_shift", "shift");
    double &eps = kwarg("epsilon", "step");
};

int main() {
    std::mt19937 mt(12345);
    // ---- bit strings, signs, excited determinants
    for (int trial = 0; trial < 400; trial++) {
        const unsigned n_orb = 4 + mt() % 28, n_elec = 2 * (1 + mt() % (n_orb / 2 > 7 ? 7 : n_orb / 2));
        const unsigned nb = (2 * n_orb + 7) / 8;
        fo::det_t d = rand_det(mt, n_orb, n_elec);
        uint8_t bytes[8]; memcpy(bytes, &d, 8);
        uint8_t occ_a[64], occ_b[64];
        const unsigned na = find_bits(bytes, occ_a, (uint8_t)nb), nbo = (unsigned)fo::occ_list(d, occ_b);
        CHECK(na == nbo && !memcmp(occ_a, occ_b, na), "find_bits");
        uint8_t hf[8] = {0}; gen_hf_bitstring(n_orb, n_elec, hf);
        fo::det_t hfw = 0; memcpy(&hfw, hf, nb);
        CHECK(hfw == fo::gen_hf_det(n_orb, n_elec), "gen_hf_bitstring");
        // a random single and a random double excitation (same spin for the single; any allowed spin pattern for the double)
        uint8_t so[2], dob[4];
        { unsigned i = mt() % n_elec; so[0] = occ_a[i]; unsigned sp = so[0] / n_orb; do { so[1] = (uint8_t)(mt() % n_orb + sp * n_orb); } while ((d >> so[1]) & 1); }
        {
            unsigned i = mt() % n_elec, j; do { j = mt() % n_elec; } while (j == i);
            if (i > j) std::swap(i, j);
            dob[0] = occ_a[i]; dob[1] = occ_a[j];
            unsigned s0 = dob[0] / n_orb, s1 = dob[1] / n_orb;
            do { dob[2] = (uint8_t)(mt() % n_orb + s0 * n_orb); } while ((d >> dob[2]) & 1);
            do { dob[3] = (uint8_t)(mt() % n_orb + s1 * n_orb); } while (((d >> dob[3]) & 1) || dob[3] == dob[2]);
            if (dob[2] > dob[3]) std::swap(dob[2], dob[3]);
        }
        CHECK(sing_parity(bytes, so) == fo::sing_parity(d, so), "sing_parity");
        CHECK(doub_parity(bytes, dob) == fo::doub_parity(d, dob), "doub_parity");
        { uint8_t b2[8]; memcpy(b2, bytes, 8); fo::det_t d2 = d; int s1 = sing_det_parity(b2, so), s2 = fo::sing_det_parity(&d2, so); fo::det_t w = 0; memcpy(&w, b2, 8); CHECK(s1 == s2 && w == d2, "sing_det_parity"); }
        { uint8_t b2[8]; memcpy(b2, bytes, 8); fo::det_t d2 = d; int s1 = doub_det_parity(b2, dob), s2 = fo::doub_det_parity(&d2, dob); fo::det_t w = 0; memcpy(&w, b2, 8); CHECK(s1 == s2 && w == d2, "doub_det_parity"); }
        { uint8_t b2[8]; memcpy(b2, bytes, 8); sing_det(b2, so); fo::det_t w = 0; memcpy(&w, b2, 8); CHECK(w == fo::sing_det(d, so), "sing_det"); }
        { uint8_t b2[8]; memcpy(b2, bytes, 8); doub_det(b2, dob); fo::det_t w = 0; memcpy(&w, b2, 8); CHECK(w == fo::doub_det(d, dob), "doub_det"); }
        { unsigned a = mt() % (2 * n_orb), b = mt() % (2 * n_orb); CHECK(bits_between(bytes, (uint8_t)a, (uint8_t)b) == fo::bits_between(d, a, b), "bits_between"); }
        for (int spin = 0; spin < 2; spin++) {
            const unsigned n_virt = n_orb - n_elec / 2, k = mt() % n_virt;
            CHECK(find_nth_virt(occ_a, spin, (uint8_t)n_elec, (uint8_t)n_orb, (uint8_t)k) == fo::find_nth_virt(occ_a, spin, n_elec, n_orb, k), "find_nth_virt");
        }
        // ---- excitation lists in the reference's order, SymmInfo
        std::vector<uint8_t> irr(n_orb);
        for (auto &x : irr) x = (uint8_t)(mt() % (trial % 3 == 0 ? 1 : 8));
        fo::Symm sy; sy.init(irr.data(), n_orb);
        SymmInfo si(irr.data(), n_orb);
        CHECK(si.max_n_symm == sy.max_n_symm, "max_n_symm");
        for (unsigned s = 0; s < 8; s++) for (unsigned c = 0; c <= si.symm_lookup(s, 0); c++) CHECK(si.symm_lookup(s, c) == sy.lk(s, c), "symm_lookup");
        std::vector<uint8_t> ex_o;
        std::vector<uint8_t> buf(4 * (size_t)n_orb * n_orb * n_elec * n_elec + 16);
        size_t n1 = sing_ex_symm(bytes, occ_a, n_elec, n_orb, (uint8_t (*)[2])buf.data(), irr.data());
        size_t n2 = fo::sing_ex_symm(d, occ_a, n_elec, n_orb, ex_o, irr.data());
        CHECK(n1 == n2 && !memcmp(buf.data(), ex_o.data(), 2 * n1), "sing_ex_symm %zu %zu", n1, n2);
        CHECK(count_singex(bytes, occ_a, n_elec, &si) == fo::count_singex(d, occ_a, n_elec, sy) && n1 == count_singex(bytes, occ_a, n_elec, &si), "count_singex");
        n1 = doub_ex_symm(bytes, occ_a, n_elec, n_orb, (uint8_t (*)[4])buf.data(), irr.data());
        n2 = fo::doub_ex_symm(d, occ_a, n_elec, n_orb, ex_o, irr.data());
        CHECK(n1 == n2 && !memcmp(buf.data(), ex_o.data(), 4 * n1), "doub_ex_symm %zu %zu", n1, n2);
        // ---- hashes
        std::vector<uint32_t> scr(2 * n_orb);
        for (auto &x : scr) x = (uint32_t)mt();
        HashTable<ssize_t> ht(0, scr);
        CHECK(ht.hash_fxn(occ_a, (uint8_t)n_elec, NULL, 0) == fo::hash_fxn(occ_a, n_elec, scr.data()), "hash_fxn");
    }
    // ---- host DistVec + Adder against the oracle's vector: random adds (initiator and not), deletes, re-use of freed positions
    for (int trial = 0; trial < 30; trial++) {
        const unsigned n_orb = 6 + mt() % 10, n_elec = 2 * (1 + mt() % 3);
        std::vector<uint32_t> ps(2 * n_orb), vs(2 * n_orb);
        for (auto &x : ps) x = (uint32_t)mt();
        for (auto &x : vs) x = (uint32_t)mt();
        const size_t cap = 400, add_cap = 64;
        DistVec<double> v(cap, add_cap, (uint8_t)(2 * n_orb), n_elec, 1, nullptr, 2, ps, vs);
        fo::Vec o; o.init(cap, add_cap, n_elec, 2, fo::Comm::self(), ps.data());
        std::vector<fo::det_t> pool(60);
        for (auto &x : pool) x = rand_det(mt, n_orb, n_elec);
        for (int round = 0; round < 25; round++) {
            const unsigned col = mt() % 2;
            v.set_curr_vec_idx((uint8_t)col); o.cur = col;
            const unsigned n_add = 1 + mt() % 50;
            for (unsigned k = 0; k < n_add; k++) {
                fo::det_t d = pool[mt() % pool.size()];
                double val = (mt() % 7 == 0) ? 0.0 : ((int)(mt() % 2001) - 1000) / 64.0;
                uint8_t ini = (uint8_t)(mt() % 2);
                uint8_t b[8]; memcpy(b, &d, 8);
                bool r1 = v.add(b, val, ini), r2 = o.add(d, val, ini);
                CHECK(r1 == r2, "add return");
            }
            v.perform_add(0); o.perform_add(0);
            // delete what became zero in both columns, like the drivers do after a compression
            for (size_t i = 0; i < v.curr_size(); i++) if ((mt() % 4) == 0) { *v(0, i) = 0; *v(1, i) = 0; o.vals[0][i] = 0; o.vals[1][i] = 0; v.del_at_pos(i); o.del_at_pos(i); }
            CHECK(v.curr_size() == o.curr_size && v.n_nonz() == o.n_nonz, "sizes %zu %zu %d %d", v.curr_size(), o.curr_size, v.n_nonz(), o.n_nonz);
            for (size_t i = 0; i < o.curr_size && i < v.curr_size(); i++) {
                fo::det_t w = 0; memcpy(&w, v.indices()[i], (2 * n_orb + 7) / 8);
                CHECK(*v(0, i) == o.vals[0][i] && *v(1, i) == o.vals[1][i], "value at %zu", i);
                if (o.vals[0][i] != 0 || o.vals[1][i] != 0) CHECK(w == o.dets[i], "determinant at %zu", i);
            }
            v.set_curr_vec_idx(0); o.cur = 0;
            CHECK(v.local_norm() == o.local_norm(), "local_norm");
            std::vector<double> w2(pool.size());
            Matrix<uint8_t> idx2(pool.size(), (2 * n_orb + 7) / 8);
            for (size_t k = 0; k < pool.size(); k++) { w2[k] = ((int)(mt() % 201) - 100) / 8.0; memcpy(idx2[k], &pool[k], (2 * n_orb + 7) / 8); }
            CHECK(v.dot(idx2, w2.data(), pool.size()) == o.dot(pool, w2), "dot");
        }
    }
    // ---- Matrix, Matrix<bool>, SymmERIs
    {
        Matrix<double> m(3, 4);
        for (size_t r = 0; r < 3; r++) for (size_t c = 0; c < 4; c++) m(r, c) = 10.0 * r + c;
        CHECK(m[2][3] == 23.0 && m.rows() == 3 && m.cols() == 4, "Matrix index");
        m.enlarge_cols(6, 4);
        CHECK(m.cols() == 6 && m(2, 3) == 23.0 && m(1, 0) == 10.0, "enlarge_cols keeps the rows");
        Matrix<bool> b(5, 19);
        b(3, 17) = true; b(0, 0) = true;
        CHECK(b(3, 17) && b(0, 0) && !b(3, 16) && b[3][17], "Matrix<bool>");
        const unsigned n = 6;
        SymmERIs e(n);
        fo::Integrals oi; oi.n_orb = n; oi.eri.assign(fo::Integrals::packed_len(n), 0.0);
        for (unsigned i = 0; i < n; i++) for (unsigned j = i; j < n; j++) for (unsigned k = 0; k < n; k++) for (unsigned l = k; l < n; l++)
            if (j * (j + 1) / 2 + i <= l * (l + 1) / 2 + k) e.chemist_ordered(i, j, k, l) = 1.0 + i + 10 * j + 100 * k + 1000 * l;
        memcpy(oi.eri.data(), e.packed(), 8 * oi.eri.size());
        for (int t = 0; t < 500; t++) { unsigned i = mt() % n, j = mt() % n, k = mt() % n, l = mt() % n; CHECK(e.chemist(i, j, k, l) == oi.chem(i, j, k, l) && e.physicist(i, j, k, l) == oi.phys(i, j, k, l) && e.chemist(i, j, k, l) == e.chemist(l, k, j, i), "SymmERIs"); }
    }
    // ---- adjust_shift (compress_utils.cpp:684-693)
    {
        double shift = 0.25, last = 0;
        adjust_shift(&shift, 90.0, &last, 100.0, 0.5);
        CHECK(shift == 0.25 && last == 0, "below the target: nothing moves");
        adjust_shift(&shift, 120.0, &last, 100.0, 0.5);
        CHECK(shift == 0.25 && last == 120.0, "first norm above the target is remembered");
        adjust_shift(&shift, 150.0, &last, 100.0, 0.5);
        CHECK(shift == 0.25 - 0.5 * log(150.0 / 120.0) && last == 150.0, "shift update");
    }
    // ---- the command-line parser
    {
        const char *av[] = {"prog", "--fcidump_path", "/x/FCIDUMP", "--epsilon", "0.01", "--max_iter=7", "--ham_shift", "-44.5"};
        MyArgs a = argparse::parse<MyArgs>(8, (char **)av);
        CHECK(a.path == "/x/FCIDUMP" && a.eps == 0.01 && a.iters == 7 && a.target == 0 && a.dir == "./" && a.load == nullptr && a.shift && *a.shift == -44.5, "argparse");
    }
    printf("HEADERS checks=%d fails=%d\n", n_chk, n_fail);
    return n_fail != 0;
}
