// frifull_mol on the MI355X engine (FRIES_bin/frifull_mol.cpp) over the C ABI: FRI with systematic vector compression and
// the Hamiltonian applied in full every iteration.
//
//   frifull_mol_hip (--fcidump_path F --point_group D2h --epsilon E | --hf_path DIR/) --vec_nonz M --max_dets N
//                   [--target T] [--max_iter K] [--result_dir DIR/] [--seed S] [--device D]
//
// Inputs: either the FCIDUMP file the newer drivers take, or -- as the reference's frifull_mol does (frifull_mol.cpp:15, 48) -- the
// legacy Hartree-Fock directory --hf_path (eris.txt, hcore.txt, symm.txt, sys_params.txt; parse_hf_dir in driver_common.hpp), whose
// sys_params.txt also carries epsilon (--epsilon overrides it).  Directories with frozen orbitals are refused.  Output files as the reference's
// (frifull_mol.cpp:206-243, 266-300): projnum.txt, projden.txt, nkept.txt every iteration; S.txt, norm.txt every 10; params.txt.
// --load_dir / --ini_vec / --trial_vec are not implemented (HF start, HF trial vector).
#include "driver_common.hpp"

int main(int argc, char **argv) {
    std::map<std::string, std::string> kv;
    try {
        kv = parse_kv(argc, argv);
        for (const char *k : {"load_dir", "ini_vec", "trial_vec"}) if (kv.count(k)) throw std::runtime_error(std::string("--") + k + " is not implemented in frifull_mol_hip");
        for (const char *k : {"max_dets", "vec_nonz"}) if (!kv.count(k)) throw std::runtime_error(std::string("missing required option --") + k);
        if (!kv.count("hf_path")) for (const char *k : {"fcidump_path", "epsilon"}) if (!kv.count(k)) throw std::runtime_error(std::string("missing required option --") + k + " (or pass --hf_path)");
    } catch (std::exception &ex) { std::cerr << "\nError parsing command line: " << ex.what() << "\n\n"; return 1; }
    try {
        const std::string pg = kv.count("point_group") ? kv["point_group"] : "C1";
        const std::string rd = kv.count("result_dir") ? kv["result_dir"] : "./";
        Fcidump in;
        double eps_in = 0;
        if (kv.count("hf_path")) { HfDir h = parse_hf_dir(kv["hf_path"]); in = h.mol; eps_in = h.eps; }
        else in = parse_fcidump(kv["fcidump_path"], pg);
        if (kv.count("epsilon")) eps_in = std::stod(kv["epsilon"]);
        fries_ctx *ctx = nullptr;
        ck(fries_ctx_create(&ctx, kv.count("device") ? std::stoi(kv["device"]) : 0));
        ck(fries_set_molecule(ctx, in.n_orb, in.n_elec, in.symm.data(), in.hcore.data(), in.eris.data()));
        uint32_t seed = kv.count("seed") ? (uint32_t)std::stoul(kv["seed"]) : wall_clock_seed();
        std::cout << "seed on process 0 is " << seed << std::endl;
        fries_frifull_params p{eps_in, kv.count("target") ? std::stod(kv["target"]) : 0.0, (uint32_t)std::stoul(kv["vec_nonz"]),
                               (uint32_t)std::stoul(kv["max_dets"]), seed, kv.count("spawn_cap") ? (uint32_t)std::stoul(kv["spawn_cap"]) : 0u};
        ck(fries_frifull_setup(ctx, &p));
        const uint32_t max_iter = kv.count("max_iter") ? (uint32_t)std::stoul(kv["max_iter"]) : 1000000u;
        std::ofstream num_file(rd + "projnum.txt", std::ofstream::app), den_file(rd + "projden.txt", std::ofstream::app), shift_file(rd + "S.txt", std::ofstream::app),
            norm_file(rd + "norm.txt", std::ofstream::app), nkept_file(rd + "nkept.txt", std::ofstream::app);
        if (!num_file.is_open()) throw std::runtime_error("Could not open file for writing in directory " + rd);
        num_file.precision(17); den_file.precision(17); shift_file.precision(17); norm_file.precision(17);
        {
            std::ofstream param_f(rd + "params.txt");
            param_f << "FRI calculation\n" << (kv.count("hf_path") ? "HF path: " + kv["hf_path"] : "FCIDUMP path: " + kv["fcidump_path"]) << "\nepsilon (imaginary time step): " << p.epsilon << "\nTarget norm " << p.target_norm
                    << "\nVector nonzero: " << p.vec_nonz << "\nInitializing calculation from HF unit vector\n";
        }
        for (uint32_t it = 0; it < max_iter; it++) {
            fries_iter_log lg;
            ck(fries_frifull_iterate(ctx, 1, &lg));
            nkept_file << lg.nkept << '\n';
            if ((it + 1) % 10 == 0) { shift_file << lg.shift << "\n"; norm_file << lg.norm << "\n"; }
            num_file << lg.numer << '\n'; den_file << lg.denom << "\n";
            std::cout << it << ", en est: " << lg.numer / lg.denom << ", shift: " << lg.shift << ", norm: " << lg.norm << '\n';
        }
        fries_ctx_destroy(ctx);
    } catch (std::exception &ex) { std::cerr << "\nException : " << ex.what() << "\n"; }
    return 0;
}
