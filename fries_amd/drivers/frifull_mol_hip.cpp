// frifull_mol on the MI355X engine (FRIES_bin/frifull_mol.cpp) over the C ABI: FRI with systematic vector compression and
// the Hamiltonian applied in full every iteration.
//
//   frifull_mol_hip (--fcidump_path F --point_group D2h --epsilon E | --hf_path DIR/) --vec_nonz M --max_dets N
//                   [--target T] [--max_iter K] [--result_dir DIR/] [--seed S] [--device D] [--ranks P [--spawn_cap N]]
//
// Inputs: either the FCIDUMP file the newer drivers take, or -- as the reference's frifull_mol does (frifull_mol.cpp:15, 48) -- the
// legacy Hartree-Fock directory --hf_path (eris.txt, hcore.txt, symm.txt, sys_params.txt; parse_hf_dir in driver_common.hpp), whose
// sys_params.txt also carries epsilon (--epsilon overrides it).  Directories with frozen orbitals are refused.  Output files as the reference's
// (frifull_mol.cpp:206-243, 266-300): projnum.txt, projden.txt, nkept.txt every iteration; S.txt, norm.txt every 10; params.txt.
// --load_dir / --ini_vec / --trial_vec are not implemented (HF start, HF trial vector).
#include "driver_common.hpp"
#include <thread>

// one rank of the run; `tr` == nullptr is the one-rank run.  The rank that owns the HF determinant writes the text outputs (frifull_mol.cpp:206-243).
static void run_rank(std::map<std::string, std::string> kv, const Fcidump &in, double eps_in, uint32_t seed, int rank, int device, fries_transport *tr) {
    const std::string rd = kv.count("result_dir") ? kv["result_dir"] : "./";
    fries_ctx *ctx = nullptr;
    ck(fries_ctx_create(&ctx, device));
    ck(fries_set_molecule(ctx, in.n_orb, in.n_elec, in.symm.data(), in.hcore.data(), in.eris.data()));
    if (tr) {
        fries_comm cm;
        ck(fries_transport_comm(tr, &cm));
        ck(fries_set_comm(ctx, &cm));
    }
    fries_frifull_params p{eps_in, kv.count("target") ? std::stod(kv["target"]) : 0.0, (uint32_t)std::stoul(kv["vec_nonz"]),
                           (uint32_t)std::stoul(kv["max_dets"]), seed, kv.count("spawn_cap") ? (uint32_t)std::stoul(kv["spawn_cap"]) : 0u};
    ck(fries_frifull_setup(ctx, &p));
    int32_t hf_proc = 0;
    {
        uint64_t hf = 0;
        for (unsigned k = 0; k < in.n_elec / 2; k++) hf |= (1ull << k) | (1ull << (k + in.n_orb));
        if (tr) ck(fries_idx_to_proc(ctx, &hf, 1, &hf_proc));
    }
    const bool writer = rank == hf_proc;
    const uint32_t max_iter = kv.count("max_iter") ? (uint32_t)std::stoul(kv["max_iter"]) : 1000000u;
    std::ofstream num_file, den_file, shift_file, norm_file, nkept_file;
    if (writer) {
        num_file.open(rd + "projnum.txt", std::ofstream::app); den_file.open(rd + "projden.txt", std::ofstream::app); shift_file.open(rd + "S.txt", std::ofstream::app);
        norm_file.open(rd + "norm.txt", std::ofstream::app); nkept_file.open(rd + "nkept.txt", std::ofstream::app);
        if (!num_file.is_open()) throw std::runtime_error("Could not open file for writing in directory " + rd);
        num_file.precision(out_precision(kv)); den_file.precision(out_precision(kv)); shift_file.precision(out_precision(kv)); norm_file.precision(out_precision(kv));
        std::ofstream param_f(rd + "params.txt");
        param_f << "FRI calculation\n" << (kv.count("hf_path") ? "HF path: " + kv["hf_path"] : "FCIDUMP path: " + kv["fcidump_path"]) << "\nepsilon (imaginary time step): " << p.epsilon << "\nTarget norm " << p.target_norm
                << "\nVector nonzero: " << p.vec_nonz << "\nInitializing calculation from HF unit vector\n";
    }
    for (uint32_t it = 0; it < max_iter; it++) {
        fries_iter_log lg;
        ck(fries_frifull_iterate(ctx, 1, &lg));
        if (!writer) continue;
        nkept_file << lg.nkept << '\n';
        if ((it + 1) % 10 == 0) { shift_file << lg.shift << "\n"; norm_file << lg.norm << "\n"; }
        num_file << lg.numer << '\n'; den_file << lg.denom << "\n";
        std::cout << it << ", en est: " << lg.numer / lg.denom << ", shift: " << lg.shift << ", norm: " << lg.norm << '\n';
    }
    fries_ctx_destroy(ctx);
}

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
        Fcidump in;
        double eps_in = 0;
        if (kv.count("hf_path")) { HfDir h = parse_hf_dir(kv["hf_path"]); in = h.mol; eps_in = h.eps; }
        else in = parse_fcidump(kv["fcidump_path"], pg);
        if (kv.count("epsilon")) eps_in = std::stod(kv["epsilon"]);
        const int device = kv.count("device") ? std::stoi(kv["device"]) : 0;
        const int P = kv.count("ranks") ? std::stoi(kv["ranks"]) : 1;
        if (P > 1 && !kv.count("seed")) throw std::runtime_error("several ranks need the same --seed on every rank (the reference broadcasts rank 0's draws)");
        uint32_t seed = kv.count("seed") ? (uint32_t)std::stoul(kv["seed"]) : wall_clock_seed();
        std::cout << "seed on process 0 is " << seed << std::endl;
        if (P > 1) {
            // --ranks P: P rank threads of this process over the native local transport (frifull_mol under mpiexec -n P: the singles and the doubles
            // of a rank's determinants travel as two passes of adds); the spawn buffer of a rank holds one pass
            const uint64_t cap = kv.count("spawn_cap") ? std::stoull(kv["spawn_cap"]) : 8000000ull;
            fries_local_group *grp = nullptr;
            ck(fries_local_group_create(&grp, P, 16ull * (cap + 4096)));
            std::vector<fries_transport *> tr(P, nullptr);
            for (int r = 0; r < P; r++) ck(fries_local_create(&tr[r], grp, r, device));
            std::vector<std::string> errs(P);
            std::vector<std::thread> th;
            for (int r = 0; r < P; r++)
                th.emplace_back([&, r] {
                    try { run_rank(kv, in, eps_in, seed, r, device, tr[r]); }
                    catch (std::exception &ex) { errs[r] = ex.what(); }
                });
            for (auto &t : th) t.join();
            for (int r = 0; r < P; r++) { if (!errs[r].empty()) std::cerr << "\nException on rank " << r << " : " << errs[r] << "\n"; fries_transport_destroy(tr[r]); }
            fries_local_group_destroy(grp);
        }
        else run_rank(kv, in, eps_in, seed, 0, device, nullptr);
    } catch (std::exception &ex) { std::cerr << "\nException : " << ex.what() << "\n"; }
    return 0;
}
