// frimulti_mol on the MI355X engine (FRIES_bin/frimulti_mol.cpp) over the C ABI.
//
//   frimulti_mol_hip --fcidump_path F --point_group D2h --distribution HB --vec_nonz N --mat_nonz M --max_dets N --epsilon E
//                    [--target T] [--initiator I] [--max_iter K] [--result_dir DIR/] [--ini_vec PREFIX] [--trial_vec PREFIX] [--seed S] [--device D]
//
// The reference reads its integrals from the legacy --hf_path directory (the time step comes from there); here they come from an
// FCIDUMP file as in frisys_mol, and --epsilon is a flag.  --distribution: the reference's check accepts only "HB"
// (frimulti_mol.cpp:38-46).  --ini_vec: the reference's text vectors (<prefix>dets, <prefix>vals; frimulti_mol.cpp:205-215).  --trial_vec ends as in the
// reference: its trial vector's Adder holds n_trial entries and the driver throws on the add() that fills it (frimulti_mol.cpp:149-157), so every trial file
// is refused with "Insufficient memory allocated in adder".  Not provided: --load_dir, --det_space, --unbias.
// Output files (frimulti_mol.cpp:235-267, 387-411): projnum.txt, projden.txt, nini.txt every iteration; S.txt and norm.txt every
// 10 iterations; params.txt.
#include "driver_common.hpp"

int main(int argc, char **argv) {
    std::map<std::string, std::string> kv;
    try {
        kv = parse_kv(argc, argv);
        for (const char *k : {"fcidump_path", "max_dets", "epsilon", "vec_nonz", "mat_nonz"}) if (!kv.count(k)) throw std::runtime_error(std::string("missing required option --") + k);
        if (!kv.count("distribution") || kv["distribution"] != "HB") throw std::runtime_error("\"dist_str\" argument must be either \"NU\" or \"HB\"");
        for (const char *k : {"load_dir", "det_space", "unbias"}) if (kv.count(k)) throw std::runtime_error(std::string("option --") + k + " is not provided by this driver");
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
        fries_frimulti_params p{std::stod(kv["epsilon"]), kv.count("target") ? std::stod(kv["target"]) : 0.0, kv.count("initiator") ? std::stod(kv["initiator"]) : 0.0,
                                (uint32_t)std::stoul(kv["vec_nonz"]), (uint32_t)std::stoul(kv["mat_nonz"]), (uint32_t)std::stoul(kv["max_dets"]), seed};
        std::vector<uint64_t> tdets; std::vector<double> tvals;
        if (kv.count("trial_vec")) { load_vec_txt(kv["trial_vec"], tdets, tvals); ck(fries_set_trial_vector(ctx, tdets.data(), tvals.data(), tvals.size())); }      // frimulti_mol.cpp:139-163
        if (kv.count("ini_vec")) { load_vec_txt(kv["ini_vec"], tdets, tvals); ck(fries_set_initial_vector(ctx, tdets.data(), tvals.data(), tvals.size())); }          // :205-215
        ck(fries_frimulti_setup(ctx, &p));
        const uint32_t max_iter = kv.count("max_iter") ? (uint32_t)std::stoul(kv["max_iter"]) : 1000000u;
        std::ofstream num_file(rd + "projnum.txt", std::ofstream::app), den_file(rd + "projden.txt", std::ofstream::app), shift_file(rd + "S.txt", std::ofstream::app),
            norm_file(rd + "norm.txt", std::ofstream::app), ini_file(rd + "nini.txt", std::ofstream::app);
        if (!num_file.is_open()) throw std::runtime_error("Could not open file for writing in directory " + rd);
        num_file.precision(out_precision(kv)); den_file.precision(out_precision(kv)); shift_file.precision(out_precision(kv)); norm_file.precision(out_precision(kv));
        {
            std::ofstream param_f(rd + "params.txt");
            param_f << "FRI calculation\nFCIDUMP path: " << kv["fcidump_path"] << "\nepsilon (imaginary time step): " << p.epsilon << "\nTarget norm " << p.target_norm
                    << "\nInitiator threshold: " << p.initiator << "\nMatrix nonzero: " << p.mat_nonz << "\nVector nonzero: " << p.vec_nonz
                    << (kv.count("ini_vec") ? "\nInitializing calculation from vector files with prefix " + kv["ini_vec"] + "\n" : std::string("\nInitializing calculation from HF unit vector\n"));
        }
        for (uint32_t it = 0; it < max_iter; it++) {
            fries_fciqmc_log lg;
            ck(fries_frimulti_iterate(ctx, 1, &lg));
            if ((it + 1) % 10 == 0) { shift_file << lg.shift << "\n"; norm_file << lg.norm << "\n"; }
            num_file << lg.numer << '\n'; den_file << lg.denom << '\n'; ini_file << lg.n_ini << '\n';
            std::cout << it << ", en est: " << lg.numer / lg.denom << ", shift: " << lg.shift << ", norm: " << lg.norm << '\n';
        }
        fries_ctx_destroy(ctx);
    } catch (std::exception &ex) { std::cerr << "\nException : " << ex.what() << "\n"; }
    return 0;
}
