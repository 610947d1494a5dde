// fciqmc_mol and fciqmc_fp_mol on the MI355X engine (FRIES_bin/fciqmc_mol.cpp, FRIES_bin/fciqmc_fp_mol.cpp, --distribution NU or HB) over the C ABI.
//
//   fciqmc_mol_hip --fcidump_path F --point_group D2h --distribution NU|HB --target W --max_dets N --epsilon E
//                  [--initiator I] [--max_iter K] [--result_dir DIR/] [--ini_vec PREFIX] [--trial_vec PREFIX] [--seed S] [--device D]
//                  [--fp 1]       (fciqmc_fp_mol: real-valued walkers; without --ini_vec / --trial_vec)
//
// Output files as the reference's (fciqmc_mol.cpp:262-300, 415-445): projnum.txt, projden.txt, nini.txt every iteration; S.txt,
// N.txt (walkers), nnonz.txt every 10 iterations; params.txt.
#include "driver_common.hpp"

int main(int argc, char **argv) {
    std::map<std::string, std::string> kv;
    try {
        kv = parse_kv(argc, argv);
        for (const char *k : {"fcidump_path", "max_dets", "epsilon"}) if (!kv.count(k)) throw std::runtime_error(std::string("missing required option --") + k);
        if (!kv.count("distribution") || (kv["distribution"] != "NU" && kv["distribution"] != "HB")) throw std::runtime_error("\"dist_str\" argument must be either \"NU\" or \"HB\"");
    } catch (std::exception &ex) { std::cerr << "\nError parsing command line: " << ex.what() << "\n\n"; return 1; }
    try {
        const std::string pg = kv.count("point_group") ? kv["point_group"] : "C1";
        const std::string rd = kv.count("result_dir") ? kv["result_dir"] : "./";
        Fcidump in = parse_fcidump(kv["fcidump_path"], pg);
        fries_ctx *ctx = nullptr;
        ck(fries_ctx_create(&ctx, kv.count("device") ? std::stoi(kv["device"]) : 0));
        ck(fries_set_molecule(ctx, in.n_orb, in.n_elec, in.symm.data(), in.hcore.data(), in.eris.data()));
        uint32_t seed = kv.count("seed") ? (uint32_t)std::stoul(kv["seed"]) : wall_clock_seed();
        std::cout << "seed on process 0 is " << seed << std::endl;
        fries_fciqmc_params p{std::stod(kv["epsilon"]), kv.count("target") ? (uint32_t)std::stoul(kv["target"]) : 0u,
                              kv.count("initiator") ? (uint32_t)std::stoul(kv["initiator"]) : 0u, (uint32_t)std::stoul(kv["max_dets"]), seed,
                              kv["distribution"] == "HB" ? 1 : 0, (kv.count("fp") && std::stoul(kv["fp"]) != 0) ? 1 : 0};
        std::vector<uint64_t> tdets; std::vector<double> tvals;
        if (kv.count("trial_vec")) { load_vec_txt(kv["trial_vec"], tdets, tvals); ck(fries_set_trial_vector(ctx, tdets.data(), tvals.data(), tvals.size())); }      // fciqmc_mol.cpp:150-177
        if (kv.count("ini_vec")) { load_vec_txt(kv["ini_vec"], tdets, tvals); ck(fries_set_initial_vector(ctx, tdets.data(), tvals.data(), tvals.size())); }          // :226-237
        ck(fries_fciqmc_setup(ctx, &p));
        const uint32_t max_iter = kv.count("max_iter") ? (uint32_t)std::stoul(kv["max_iter"]) : 1000000u;
        std::ofstream num_file(rd + "projnum.txt", std::ofstream::app), den_file(rd + "projden.txt", std::ofstream::app), shift_file(rd + "S.txt", std::ofstream::app),
            walk_file(rd + "N.txt", std::ofstream::app), nonz_file(rd + "nnonz.txt", std::ofstream::app), ini_file(rd + "nini.txt", std::ofstream::app);
        if (!num_file.is_open()) throw std::runtime_error("Could not open file for writing in directory " + rd);
        num_file.precision(17); den_file.precision(17); shift_file.precision(17);
        {
            std::ofstream param_f(rd + "params.txt");
            param_f << (p.real_walkers ? "Non-integer FCIQMC calculation\nFCIDUMP path: " : "FCIQMC calculation\nFCIDUMP path: ") << kv["fcidump_path"] << "\nepsilon (imaginary time step): " << p.epsilon << "\nTarget number of walkers "
                    << p.target_walkers << "\nInitiator threshold: " << p.initiator << "\nInitializing calculation from HF unit vector\n";
        }
        for (uint32_t it = 0; it < max_iter; it++) {
            fries_fciqmc_log lg;
            ck(fries_fciqmc_iterate(ctx, 1, &lg));
            if ((it + 1) % 10 == 0) { walk_file << (uint32_t)lg.norm << "\n"; shift_file << lg.shift << "\n"; nonz_file << lg.n_nonz << "\n"; }
            num_file << lg.numer << '\n'; den_file << lg.denom << '\n'; ini_file << lg.n_ini << '\n';
            std::cout << it << ", n walk: " << (uint32_t)lg.norm << ", en est: " << lg.numer / lg.denom << ", shift: " << lg.shift << '\n';
        }
        fries_ctx_destroy(ctx);
    } catch (std::exception &ex) { std::cerr << "\nException : " << ex.what() << "\n"; }
    return 0;
}
