// Host-side test of include/fries_facade.hpp, written the way the reference's own code is:
//   (1) "Test adding elements to vector" (tests/test_vector.cpp:192-224) against the device vector;
//   (2) the frisys_mol iteration (FRIES_bin/frisys_mol.cpp:405-552) assembled from the operator-level calls -- apply_HBPP_sys,
//       the two-pass initiator add, death/cloning, find_preserve, adjust_shift, the dots, sys_comp -- with the spawns built on
//       the host exactly as the reference's loop builds them, compared iteration by iteration with the fused
//       fries_frisys_iterate of a second engine started from the same seed;
//   (3) apply_HBPP_piv through the facade with the caller's std::mt19937.
// usage: test_facade <FCIDUMP> <point group> <n_iter>
#include "../../include/fries_facade.hpp"
#include "../../fries_amd/drivers/driver_common.hpp"

using namespace fries_hip;

static int n_chk = 0, n_fail = 0;
#define REQUIRE(cond) do { n_chk++; if (!(cond)) { n_fail++; if (n_fail < 20) fprintf(stderr, "FAILED line %d: %s\n", __LINE__, #cond); } } while (0)

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: test_facade <FCIDUMP> <point group> <n_iter>\n"); return 2; }
    try {
        Fcidump in = parse_fcidump(argv[1], argv[2]);
        const unsigned n_iter = (unsigned)atoi(argv[3]);
        const unsigned n_orb = in.n_orb, n_elec = in.n_elec;
        const size_t det_size = (2 * n_orb + 7) / 8;
        fries_frisys_params par{0.01, 2000.0, 1.0, 4000, 4000, 60000, 20250215u, 1};
        const double eps = par.epsilon;

        Engine eng_ops, eng_fused;
        for (Engine *e : {&eng_ops, &eng_fused}) { e->set_molecule(n_orb, n_elec, in.symm.data(), in.hcore.data(), in.eris.data()); e->setup(par); }
        unsigned spawn_length = par.mat_nonz * 4;
        size_t adder_size = spawn_length > 1000000 ? 1000000 : spawn_length;
        DistVec<double> sol_vec(eng_ops, adder_size);

        // ---- "Test adding elements to vector": the vector starts as 100 x HF at position 0 (frisys_mol.cpp:277-281)
        {
            std::vector<uint8_t> bit_str1(det_size, 255);
            gen_hf_bitstring(n_orb, n_elec, bit_str1.data());
            REQUIRE(*sol_vec[0] == 100);
            REQUIRE(!memcmp(bit_str1.data(), sol_vec.indices()[0], det_size));
            sol_vec.add(bit_str1.data(), 1, 1);
            sol_vec.perform_add(0);
            REQUIRE(*sol_vec[0] == 101);
            sol_vec.add(bit_str1.data(), -1, 1);
            sol_vec.perform_add(0);
            REQUIRE(*sol_vec[0] == 100 && sol_vec.curr_size() == 1 && sol_vec.n_nonz() == 1);
        }

        // ---- the iteration, operator by operator
        std::mt19937 mt_obj(par.seed);
        for (unsigned k = 0; k < 4 * n_orb; k++) mt_obj();            // the proc and vec scramblers come first (frisys_mol.cpp:132-145)
        size_t n_states = n_elec > (n_orb - n_elec / 2) ? n_elec : n_orb - n_elec / 2;
        HBCompressSys comp_vecs(spawn_length, n_states);
        double en_shift = 0, last_one_norm = 0, glob_norm = 0;
        const double shift_damping = 0.05;
        const unsigned shift_interval = 10;
        for (unsigned iterat = 0; iterat < n_iter; iterat++) {
            apply_HBPP_sys(sol_vec, &comp_vecs, mt_obj, par.mat_nonz);
            size_t comp_len = comp_vecs.vec_len;
            sol_vec.set_curr_vec_idx(0);
            double *vals_before_mult = sol_vec.values();
            size_t vec_size = sol_vec.curr_size();
            sol_vec.set_curr_vec_idx(1);
            sol_vec.zero_vec();
            for (int add_ini = 0; add_ini < 2; add_ini++) {           // first the spawns of non-initiators (:429-471)
                int num_added = 1;
                size_t samp_idx = 0;
                while (num_added > 0) {
                    num_added = 0;
                    while (samp_idx < comp_len) {
                        size_t det_idx = comp_vecs.det_indices2[samp_idx];
                        double curr_val = vals_before_mult[det_idx];
                        uint8_t ini_flag = fabs(curr_val) >= par.initiator;
                        if (ini_flag != add_ini) { samp_idx++; continue; }
                        uint8_t *curr_det = sol_vec.indices()[det_idx];
                        uint8_t new_det[8];
                        double add_el = -eps * comp_vecs.vec1[samp_idx];
                        if (curr_val < 0) add_el *= -1;
                        std::copy(curr_det, curr_det + det_size, new_det);
                        uint8_t *orbs = comp_vecs.orb_indices1[samp_idx];
                        if (!(orbs[2] == 0 && orbs[3] == 0)) doub_det(new_det, orbs);
                        else sing_det(new_det, orbs);
                        num_added++;
                        samp_idx++;
                        if (!sol_vec.add(new_det, add_el, ini_flag)) break;
                    }
                    // the host mirror of column 0 stays valid across perform_add on column 1: spawning never moves column 0
                    sol_vec.perform_add(0);
                    sol_vec.set_curr_vec_idx(0);
                    vals_before_mult = sol_vec.values();
                    sol_vec.set_curr_vec_idx(1);
                    if (samp_idx >= comp_len) num_added = 0;          // one rank: nothing left to add
                }
            }
            sol_vec.set_curr_vec_idx(0);
            death_clone_and_add(sol_vec, eps, en_shift, vec_size);    // :487-499
            unsigned int n_samp = par.vec_nonz;
            find_preserve(sol_vec, &n_samp, &glob_norm);              // :503
            unsigned nkept = par.vec_nonz - n_samp;
            if ((iterat + 1) % shift_interval == 0) adjust_shift(&en_shift, glob_norm, &last_one_norm, par.target_norm, shift_damping / shift_interval / eps);
            double numer, denom;
            proj_dots(sol_vec, &numer, &denom);                       // :511-517
            double rn_sys = mt_obj() / (1. + UINT32_MAX);
            sys_comp(sol_vec, n_samp, rn_sys);                        // :533-539

            fries_iter_log lg = eng_fused.iterate();
            REQUIRE(lg.numer == numer && lg.denom == denom);
            REQUIRE(lg.norm == glob_norm && lg.shift == en_shift);
            REQUIRE(lg.nkept == nkept && lg.n_nonz == sol_vec.n_nonz() && lg.curr_size == sol_vec.curr_size() && lg.num_success == comp_len);
        }
        // the two vectors, slot by slot
        DistVec<double> other(eng_fused, adder_size);
        REQUIRE(other.curr_size() == sol_vec.curr_size());
        size_t n = sol_vec.curr_size(), bad = 0;
        for (size_t i = 0; i < n && i < other.curr_size(); i++)
            if (memcmp(sol_vec[i], other[i], 8) || (*sol_vec[i] != 0 && memcmp(sol_vec.indices()[i], other.indices()[i], det_size))) bad++;
        REQUIRE(bad == 0);
        // ---- (3) apply_HBPP_piv with the caller's generator (as subsp_mol.cpp:537 calls it): the generator travels to the context and
        //      back; the second engine holds the same vector and draws from its own generator seeded alike
        {
            const uint32_t n_piv = 1500;
            HBCompressPiv piv_vecs(n_piv + 4096, n_states);
            std::mt19937 pm(77);
            apply_HBPP_piv(sol_vec, &piv_vecs, pm, n_piv);
            REQUIRE(piv_vecs.vec_len > 0 && piv_vecs.vec_len <= n_piv && piv_vecs.stage_len[0] <= n_piv);
            fries_hip::ck(fries_frisys_restart(eng_fused.ctx(), 77, 0.0, 0.0, 0));
            std::vector<uint32_t> pos(n_piv + 4096); std::vector<uint8_t> orbs(4 * (n_piv + 4096)); std::vector<double> vals(n_piv + 4096);
            size_t m = 0; uint32_t sl[5];
            fries_hip::ck(fries_apply_hbpp_piv(eng_fused.ctx(), n_piv, 0, pos.data(), orbs.data(), vals.data(), pos.size(), &m, sl));
            REQUIRE(m == piv_vecs.vec_len);
            size_t badp = 0;
            for (size_t i = 0; i < m && i < piv_vecs.vec_len; i++)
                if (pos[i] != piv_vecs.det_indices2[i] || memcmp(&orbs[4 * i], piv_vecs.orb_indices1[i], 4) || memcmp(&vals[i], &piv_vecs.vec1[i], 8)) badp++;
            REQUIRE(badp == 0);
            REQUIRE(pm() == fries_next_draw(eng_fused.ctx()));        // the caller's generator came back advanced by exactly the draws made
        }
        printf("FACADE checks=%d fails=%d iterations=%u final n_nonz=%d\n", n_chk, n_fail, n_iter, sol_vec.n_nonz());
    } catch (std::exception &ex) { fprintf(stderr, "Exception: %s\n", ex.what()); return 3; }
    return n_fail != 0;
}
