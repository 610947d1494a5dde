// frisys_mol written against the REFERENCE'S OWN C++ interface -- DistVec, Adder, HBCompressSys, apply_HBPP_sys, find_preserve,
// sys_comp, adjust_shift, h_op_offdiag, sum_mpi, parse_fcidump -- as this build ships it under include/FRIES/ (see
// include/FRIES/backend.hpp for what runs where).  The iteration below makes the calls of FRIES_bin/frisys_mol.cpp:405-552 in the
// same order with the same arguments; what differs from that program is the --seed option (the reference seeds from the clock,
// :104-106), 17-digit output, and that the semi-stochastic dense subspace (--det_space) is not offered.
//
// The purpose is the boundary: a program that drives FRI through the reference's classes links against libfries_hip.so with only
// the include path changed, and its trajectory is the reference's (tests/test_gpu_refapi.py).  The fused loop (frisys_mol_hip,
// fries_frisys_iterate) is ~10x faster because it never mirrors the vector to the host; use that one for production runs.
//
// build: g++ -std=c++17 -O2 -ffp-contract=off -I include -I include/FRIES/compat frisys_mol_ref_api.cpp -L fries_amd -lfries_hip
#include <FRIES/Hamiltonians/near_uniform.hpp>
#include <FRIES/io_utils.hpp>
#include <FRIES/compress_utils.hpp>
#include <FRIES/Ext_Libs/argparse.hpp>
#include <FRIES/Hamiltonians/heat_bathPP.hpp>
#include <FRIES/Hamiltonians/molecule.hpp>
#include <chrono>
#include <iomanip>
#include <sstream>
#include <stdexcept>

struct MyArgs : public argparse::Args {
    std::string &fcidump_path = kwarg("fcidump_path", "FCIDUMP file with the integrals");
    double &target_norm = kwarg("target", "target one-norm of the solution vector").set_default(0);
    std::string &dist_str = kwarg("distribution", "HB or HB_unnorm");
    uint32_t &max_iter = kwarg("max_iter", "number of iterations").set_default(1000000);
    uint32_t &target_nonz = kwarg("vec_nonz", "non-zero vector elements kept per iteration");
    uint32_t &matr_samp = kwarg("mat_nonz", "non-zero matrix elements kept per factor");
    std::string &result_dir = kwarg("result_dir", "output directory").set_default<std::string>("./");
    uint32_t &max_n_dets = kwarg("max_dets", "capacity of the vector");
    double &init_thresh = kwarg("initiator", "initiator threshold").set_default(0);
    std::shared_ptr<std::string> &load_dir = kwarg("load_dir", "checkpoint directory of an earlier run");
    std::shared_ptr<std::string> &ini_path = kwarg("ini_vec", "prefix of <ini_vec>dets / <ini_vec>vals");
    std::shared_ptr<std::string> &trial_path = kwarg("trial_vec", "prefix of <trial_vec>dets / <trial_vec>vals");
    double &epsilon = kwarg("epsilon", "imaginary time step");
    std::string &point_group = kwarg("point_group", "point group of the FCIDUMP irrep labels").set_default<std::string>("C1");
    std::shared_ptr<double> &ham_shift = kwarg("ham_shift", "energy subtracted from the diagonal");
    std::shared_ptr<uint32_t> &seed = kwarg("seed", "mt19937 seed (default: the clock, as in the reference)");
    std::shared_ptr<std::string> &piv_after = kwarg("piv_after", "after the last iteration: apply_HBPP_piv on the vector for every seed:n_samp of this comma-separated list, results to <result_dir>piv<k>.bin");
};

int main(int argc, char *argv[]) {
    MyArgs args = argparse::parse<MyArgs>(argc, argv);
    if (args.dist_str != "HB" && args.dist_str != "HB_unnorm") { std::cerr << "\nError parsing command line: \"distribution\" must be HB or HB_unnorm\n\n"; return 1; }
    const bool new_hb = args.dist_str == "HB_unnorm";
    const double target_norm = args.target_norm;
    try {
        int n_procs = 1, proc_rank = 0;
        MPI_Init(NULL, NULL);
        MPI_Comm_size(MPI_COMM_WORLD, &n_procs);
        MPI_Comm_rank(MPI_COMM_WORLD, &proc_rank);
        size_t max_n_dets = args.max_n_dets;
        const uint32_t matr_samp = args.matr_samp;
        const double shift_damping = 0.05, eps = args.epsilon;
        const unsigned int shift_interval = 10, save_interval = 100;
        double en_shift = 0;

        fcidump_input *in_data = parse_fcidump(args.fcidump_path, args.point_group);
        const unsigned int n_elec = in_data->n_elec, n_orb = in_data->n_orb_, n_frz = 0;
        const size_t det_size = CEILING(2 * n_orb, 8);
        uint8_t *symm = in_data->symm;
        Matrix<double> *h_core = in_data->hcore;
        SymmERIs *eris = &(in_data->eris);

        std::vector<uint8_t> tmp_orbs(n_elec), hf_det(det_size);
        gen_hf_bitstring(n_orb, n_elec, hf_det.data());
        find_bits(hf_det.data(), tmp_orbs.data(), (uint8_t)det_size);
        const double hf_en = args.ham_shift ? *args.ham_shift - in_data->core_en : diag_matrel(tmp_orbs.data(), n_orb, *eris, *h_core, n_frz, n_elec);

        const unsigned int seed = args.seed ? *args.seed : (unsigned int)std::chrono::high_resolution_clock::now().time_since_epoch().count();
        std::cout << "seed on process " << proc_rank << " is " << seed << std::endl;
        std::mt19937 mt_obj(seed);

        const unsigned int spawn_length = matr_samp * 4 / n_procs;
        const size_t adder_size = spawn_length > 1000000 ? 1000000 : spawn_length;
        std::function<double(const uint8_t *)> diag_shortcut = [=](const uint8_t *occ) { return diag_matrel(occ, n_orb, *eris, *h_core, n_frz, n_elec) - hf_en; };
        std::function<double(uint8_t *, uint8_t *)> sing_shortcut = [=](uint8_t *ex, uint8_t *occ) { return sing_matr_el_nosgn(ex, occ, n_orb, *eris, *h_core, n_frz, n_elec); };
        std::function<double(uint8_t *)> doub_shortcut = [=](uint8_t *ex) { return doub_matr_el_nosgn(ex, n_orb, *eris, n_frz); };
        SymmInfo basis_symm(in_data->symm, n_orb);

        std::vector<uint32_t> proc_scrambler(2 * n_orb), vec_scrambler(2 * n_orb);
        if (args.load_dir) load_proc_hash(*args.load_dir, proc_scrambler.data());
        else { for (auto &x : proc_scrambler) x = (uint32_t)mt_obj(); save_proc_hash(args.result_dir, proc_scrambler.data(), 2 * n_orb); }
        for (auto &x : vec_scrambler) x = (uint32_t)mt_obj();

        DistVec<double> sol_vec(max_n_dets, adder_size, n_orb * 2, n_elec, n_procs, diag_shortcut, 2, proc_scrambler, vec_scrambler);
        const size_t n_states = n_elec > (n_orb - n_elec / 2) ? n_elec : n_orb - n_elec / 2;
        HBCompressSys comp_vecs(spawn_length, n_states);
        const unsigned int hf_proc = sol_vec.idx_to_proc(hf_det.data());

        // trial vector and H * trial (host vectors, :147-214)
        size_t n_trial = 1;
        const size_t n_ex = (size_t)n_orb * n_orb * n_elec * n_elec;
        Matrix<uint8_t> &load_dets = sol_vec.indices();
        double *load_vals = sol_vec.values();
        if (args.trial_path) n_trial = load_vec_txt(*args.trial_path, load_dets, load_vals);
        unsigned int tot_trial = sum_mpi((int)n_trial, proc_rank, n_procs);
        tot_trial = CEILING(tot_trial * 2, n_procs);
        DistVec<double> trial_vec(tot_trial, tot_trial, n_orb * 2, n_elec, n_procs, proc_scrambler, vec_scrambler);
        DistVec<double> htrial_vec(tot_trial * n_ex / n_procs, tot_trial * n_ex / n_procs, n_orb * 2, n_elec, n_procs, diag_shortcut, 2, proc_scrambler, vec_scrambler);
        if (args.trial_path) {
            for (size_t i = 0; i < n_trial; i++)
                if (!trial_vec.add(load_dets[i], load_vals[i], 1) || !htrial_vec.add(load_dets[i], load_vals[i], 1)) throw std::runtime_error("Insufficient memory allocated in adder");
        }
        else if (hf_proc == (unsigned)proc_rank) { trial_vec.add(hf_det.data(), 1, 1); htrial_vec.add(hf_det.data(), 1, 1); }
        trial_vec.perform_add(0);
        htrial_vec.perform_add(0);
        trial_vec.collect_procs();
        std::vector<uintmax_t> trial_hashes(trial_vec.curr_size());
        for (size_t i = 0; i < trial_vec.curr_size(); i++) trial_hashes[i] = sol_vec.idx_to_hash(trial_vec.indices()[i], tmp_orbs.data());

        std::vector<uint8_t> scratch(4 * (n_ex > spawn_length ? n_ex : spawn_length));
        h_op_offdiag(htrial_vec, symm, n_orb, *eris, *h_core, scratch.data(), scratch.size(), n_frz, n_elec, 1, 1, 0);
        htrial_vec.set_curr_vec_idx(0);
        h_op_diag(htrial_vec, 0, 0, 1);
        htrial_vec.add_vecs(0, 1);
        htrial_vec.collect_procs();
        std::vector<uintmax_t> htrial_hashes(htrial_vec.curr_size());
        for (size_t i = 0; i < htrial_vec.curr_size(); i++) htrial_hashes[i] = sol_vec.idx_to_hash(htrial_vec.indices()[i], tmp_orbs.data());

        sol_vec.gen_orb_list(hf_det.data(), tmp_orbs.data());
        const size_t n_hf_doub = doub_ex_symm(hf_det.data(), tmp_orbs.data(), n_elec, n_orb, (uint8_t (*)[4])scratch.data(), symm);
        const size_t n_hf_sing = count_singex(hf_det.data(), tmp_orbs.data(), n_elec, &basis_symm);
        const double p_doub = (double)n_hf_doub / (n_hf_sing + n_hf_doub);

        // starting vector (:261-288)
        double loc_norm, glob_norm, last_norm = 0;
        if (args.load_dir) { sol_vec.load(*args.load_dir); load_last_line(*args.load_dir + "S.txt", &en_shift); }
        else if (args.ini_path) {
            Matrix<uint8_t> ini_dets(max_n_dets, det_size);
            double *ini_vals = sol_vec.values();
            size_t n_dets = load_vec_txt(*args.ini_path, ini_dets, ini_vals);
            for (size_t i = 0; i < n_dets; i++) sol_vec.add(ini_dets[i], ini_vals[i], 1);
            std::fill(ini_vals, ini_vals + n_dets + 1, 0.0);
        }
        else if (hf_proc == (unsigned)proc_rank) sol_vec.add(hf_det.data(), 100, 1);
        sol_vec.perform_add(0);
        loc_norm = sol_vec.local_norm();
        glob_norm = sum_mpi(loc_norm, proc_rank, n_procs);
        if (args.load_dir) last_norm = glob_norm;
        (void)last_norm;

        std::ofstream norm_file, num_file, den_file, shift_file, nkept_file, ini_file;
        auto open_out = [&](std::ofstream &f, const char *name) {
            f.open(args.result_dir + name, std::ofstream::app);
            if (!f.is_open()) throw std::runtime_error("Could not open file for writing in directory " + args.result_dir);
            f << std::setprecision(17);
        };
        open_out(num_file, "projnum.txt"); open_out(den_file, "projden.txt"); open_out(shift_file, "S.txt");
        open_out(norm_file, "norm.txt"); open_out(nkept_file, "nkept.txt"); open_out(ini_file, "nini.txt");
        if (!args.load_dir) { std::ofstream dense_f(args.result_dir + "dense.txt"); dense_f << 0 << ", \n"; }

        hb_info *hb_probs = set_up(n_orb, n_orb, *eris);
        double last_one_norm = 0, rn_sys = 0;
        int glob_n_nonz;
        std::vector<double> loc_norms(n_procs);
        max_n_dets = sol_vec.max_size();
        std::vector<size_t> srt_arr(max_n_dets);
        std::vector<bool> keep_exact(max_n_dets, false);
        const size_t n_determ = 0;
        const unsigned int tot_dense_h = 0;

        for (unsigned int iterat = 0; iterat < args.max_iter; iterat++) {
            size_t n_ini = 0;
            glob_n_nonz = sum_mpi(sol_vec.n_nonz(), proc_rank, n_procs);
            if ((uint32_t)glob_n_nonz > args.matr_samp) std::cerr << "Warning: target number of matrix samples " << args.matr_samp << " is less than number of nonzero vector elements (" << glob_n_nonz << ")\n";

            // systematic matrix compression (:414-422)
            std::copy(sol_vec.values() + n_determ, sol_vec.values() + sol_vec.curr_size(), comp_vecs.vec1.begin());
            for (size_t det_idx = n_determ; det_idx < sol_vec.curr_size(); det_idx++) comp_vecs.det_indices1[det_idx - n_determ] = det_idx;
            size_t comp_len = sol_vec.curr_size() - n_determ;
            comp_vecs.vec_len = comp_len;
            apply_HBPP_sys(sol_vec.occ_orbs(), sol_vec.indices(), &comp_vecs, hb_probs, &basis_symm, p_doub, new_hb, mt_obj, matr_samp - tot_dense_h, sing_shortcut, doub_shortcut);
            comp_len = comp_vecs.vec_len;

            // spawning: elements from non-initiators first, then from initiators, in adder-sized pieces (:424-471)
            double *vals_before_mult = sol_vec.values();
            sol_vec.set_curr_vec_idx(1);
            sol_vec.zero_vec();
            const size_t vec_size = sol_vec.curr_size();
            for (int add_ini = 0; add_ini < 2; add_ini++) {
                int num_added = 1;
                size_t samp_idx = 0;
                while (num_added > 0) {
                    num_added = 0;
                    Matrix<uint8_t> &all_dets = sol_vec.indices();
                    while (samp_idx < comp_len) {
                        const size_t det_idx = comp_vecs.det_indices2[samp_idx];
                        const double curr_val = vals_before_mult[det_idx];
                        const uint8_t ini_flag = fabs(curr_val) >= args.init_thresh;
                        if (ini_flag != add_ini) { samp_idx++; continue; }
                        uint8_t new_det[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                        double add_el = -eps * comp_vecs.vec1[samp_idx];
                        if (curr_val < 0) add_el *= -1;
                        std::copy(all_dets[det_idx], all_dets[det_idx] + det_size, new_det);
                        uint8_t *ex_orbs = comp_vecs.orb_indices1[samp_idx];
                        if (!(ex_orbs[2] == 0 && ex_orbs[3] == 0)) doub_det(new_det, ex_orbs);
                        else sing_det(new_det, ex_orbs);
                        num_added++;
                        samp_idx++;
                        if (!sol_vec.add(new_det, add_el, ini_flag)) break;
                    }
                    sol_vec.perform_add(0);
                    sol_vec.set_curr_vec_idx(0);
                    vals_before_mult = sol_vec.values();
                    sol_vec.set_curr_vec_idx(1);
                    num_added = sum_mpi(num_added, proc_rank, n_procs);
                }
            }

            // death / cloning (:487-499)
            sol_vec.set_curr_vec_idx(0);
            for (size_t det_idx = 0; det_idx < vec_size; det_idx++) {
                double *curr_val = sol_vec[det_idx];
                if (*curr_val != 0) {
                    const double diag_el = sol_vec.matr_el_at_pos(det_idx);
                    *curr_val *= 1 - eps * (diag_el - en_shift);
                }
            }
            sol_vec.add_vecs(0, 1);
            sol_vec.set_curr_vec_idx(1);
            sol_vec.zero_vec();
            sol_vec.set_curr_vec_idx(0);

            // vector compression (:501-539)
            unsigned int n_samp = args.target_nonz;
            loc_norms[proc_rank] = find_preserve(&(sol_vec.values()[n_determ]), srt_arr, keep_exact, sol_vec.curr_size() - n_determ, &n_samp, &glob_norm);
            glob_norm += sol_vec.dense_norm();
            nkept_file << args.target_nonz - n_samp << '\n';
            if ((iterat + 1) % shift_interval == 0) {
                adjust_shift(&en_shift, glob_norm, &last_one_norm, target_norm, shift_damping / shift_interval / eps);
                shift_file << en_shift << "\n";
                norm_file << glob_norm << "\n";
            }
            double numer = sol_vec.dot(htrial_vec.indices(), htrial_vec.values(), htrial_vec.curr_size(), htrial_hashes);
            double denom = sol_vec.dot(trial_vec.indices(), trial_vec.values(), trial_vec.curr_size(), trial_hashes);
            numer = sum_mpi(numer, proc_rank, n_procs);
            denom = sum_mpi(denom, proc_rank, n_procs);
            num_file << numer << '\n';
            den_file << denom << '\n';
            std::cout << iterat << ", en est: " << numer / denom << ", shift: " << en_shift << ", norm: " << glob_norm << '\n';
            ini_file << n_ini << '\n';

            rn_sys = mt_obj() / (1. + UINT32_MAX);
            MPI_Allgather(MPI_IN_PLACE, 0, MPI_DOUBLE, loc_norms.data(), 1, MPI_DOUBLE, MPI_COMM_WORLD);
            sys_comp(&(sol_vec.values()[n_determ]), sol_vec.curr_size() - n_determ, loc_norms.data(), n_samp, keep_exact, rn_sys);
            for (size_t det_idx = 0; det_idx < sol_vec.curr_size() - n_determ; det_idx++)
                if (keep_exact[det_idx]) { sol_vec.del_at_pos(det_idx + n_determ); keep_exact[det_idx] = 0; }

            if ((iterat + 1) % save_interval == 0) {
                sol_vec.save(args.result_dir);
                num_file.flush(); den_file.flush(); shift_file.flush(); nkept_file.flush();
            }
        }
        sol_vec.save(args.result_dir);
        if (args.piv_after) {
            // the pivotal variant of the H compression through the same surface (heat_bathPP.hpp:353-357), each case with a generator of its own
            HBCompressPiv piv_vecs(spawn_length, n_states);
            std::stringstream list(*args.piv_after);
            std::string item;
            for (int k = 0; std::getline(list, item, ','); k++) {
                const size_t colon = item.find(':');
                std::mt19937 mt_piv((unsigned int)std::stoul(item.substr(0, colon)));
                const uint32_t n_piv = (uint32_t)std::stoul(item.substr(colon + 1));
                piv_vecs.vec_len = sol_vec.curr_size();
                apply_HBPP_piv(sol_vec.occ_orbs(), sol_vec.indices(), &piv_vecs, hb_probs, &basis_symm, p_doub, new_hb, mt_piv, n_piv, sing_shortcut, doub_shortcut, 0);
                std::ofstream f(args.result_dir + "piv" + std::to_string(k) + ".bin", std::ios::binary);
                const uint64_t n_out = piv_vecs.vec_len, next_draw = mt_piv();
                f.write((const char *)&n_out, 8); f.write((const char *)&next_draw, 8);
                for (size_t i = 0; i < n_out; i++) { const uint64_t d = piv_vecs.det_indices2[i]; f.write((const char *)&d, 8); }
                f.write((const char *)piv_vecs.orb_indices1, 4 * n_out);
                f.write((const char *)piv_vecs.vec1.data(), 8 * n_out);
            }
        }
        MPI_Finalize();
    } catch (std::exception &ex) {
        std::cerr << "\nException : " << ex.what() << "\n";
        return 1;
    }
    return 0;
}
