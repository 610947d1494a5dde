// fciqmc_mol and fciqmc_fp_mol on the MI355X engine (FRIES_bin/fciqmc_mol.cpp, FRIES_bin/fciqmc_fp_mol.cpp, --distribution NU or HB) over the C ABI.
//
//   fciqmc_mol_hip --fcidump_path F --point_group D2h --distribution NU|HB --target W --max_dets N --epsilon E
//                  [--initiator I] [--max_iter K] [--result_dir DIR/] [--ini_vec PREFIX] [--trial_vec PREFIX] [--seed S] [--device D]
//                  [--fp 1]       (fciqmc_fp_mol: real-valued walkers; --ini_vec then holds reals, fciqmc_fp_mol.cpp:233-246)
//                  [--load_dir DIR/]   (fciqmc_mol.cpp:122-124, 214-222, 250-252: proc scrambler from hash.dat, DistVec<int>::load of
//                                       dets0.dat / vals0.dat, shift from S.txt's last line, last walker number = the loaded one)
//
// Checkpoint (every 1000 iterations and at the end, :447-459): dets0.dat (index bytes), vals0.dat (one int32 per position; doubles with
// --fp 1), dense.txt, hash.dat -- DistVec<int>::save / save_proc_hash.
//
// Output files as the reference's (fciqmc_mol.cpp:262-300, 415-445): projnum.txt, projden.txt, nini.txt every iteration; S.txt,
// N.txt (walkers), nnonz.txt every 10 iterations; params.txt.
#include "driver_common.hpp"

static void save_walkers(fries_ctx *ctx, const std::string &dir, unsigned n_orb, bool real_walkers) {
    uint32_t n = 0; int32_t nz; uint32_t nf;
    ck(fries_vec_info(ctx, &n, &nz, &nf));
    std::vector<uint64_t> dets(n ? n : 1); std::vector<double> vals(n ? n : 1);
    size_t m = 0;
    ck(fries_vec_download(ctx, dets.data(), vals.data(), dets.size(), &m));
    const size_t n_bytes = (2 * n_orb + 7) / 8;
    std::ofstream fd(dir + "dets0.dat", std::ios::binary);
    for (size_t i = 0; i < m; i++) fd.write((const char *)&dets[i], (std::streamsize)n_bytes);
    std::ofstream fv(dir + "vals0.dat", std::ios::binary);
    if (real_walkers) fv.write((const char *)vals.data(), (std::streamsize)(8 * m));
    else { std::vector<int32_t> iv(m); for (size_t i = 0; i < m; i++) iv[i] = (int32_t)vals[i]; fv.write((const char *)iv.data(), (std::streamsize)(4 * m)); }
    std::ofstream fx(dir + "dense.txt");
    fx << 0 << '\n';
}

// DistVec<int>::load (vec_utils.hpp:761-844): entries with a non-zero walker number, in file order, into positions 0..; -> walkers loaded
static double load_walkers(fries_ctx *ctx, const std::string &dir, unsigned n_orb, bool real_walkers) {
    const size_t n_bytes = (2 * n_orb + 7) / 8;
    std::ifstream fd(dir + "dets0.dat", std::ios::binary | std::ios::ate);
    if (!fd.is_open()) throw std::runtime_error("Could not open saved binary vector file at path " + dir + "dets0.dat");
    size_t n = (size_t)fd.tellg() / n_bytes;
    fd.seekg(0);
    std::vector<uint64_t> dets(n, 0); std::vector<double> vals(n);
    for (size_t i = 0; i < n; i++) fd.read((char *)&dets[i], (std::streamsize)n_bytes);
    std::ifstream fv(dir + "vals0.dat", std::ios::binary);
    if (!fv.is_open()) throw std::runtime_error("Could not open saved binary vector file at path " + dir + "vals0.dat");
    if (real_walkers) fv.read((char *)vals.data(), (std::streamsize)(8 * n));
    else { std::vector<int32_t> iv(n); fv.read((char *)iv.data(), (std::streamsize)(4 * n)); for (size_t i = 0; i < n; i++) vals[i] = iv[i]; }
    std::vector<uint64_t> d2; std::vector<double> v2;
    double walkers = 0;
    for (size_t i = 0; i < n; i++) if (fabs(vals[i]) > 1e-9) { d2.push_back(dets[i]); v2.push_back(vals[i]); walkers += fabs(vals[i]); }
    ck(fries_vec_load(ctx, d2.data(), v2.data(), d2.size()));
    return walkers;
}

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
        const std::string load_dir = kv.count("load_dir") ? kv["load_dir"] : "";
        if (!load_dir.empty()) {                            // :122-124 load_proc_hash
            std::ifstream fh(load_dir + "hash.dat", std::ios::binary);
            if (!fh.is_open()) throw std::runtime_error("Error: could not open saved hash scrambler at " + load_dir + "hash.dat");
            std::vector<uint32_t> scr(2 * in.n_orb);
            fh.read((char *)scr.data(), (std::streamsize)(4 * scr.size()));
            ck(fries_set_proc_scrambler(ctx, scr.data(), scr.size()));
        }
        ck(fries_fciqmc_setup(ctx, &p));
        if (!load_dir.empty()) {                            // :214-222, :250-252
            const double walkers = load_walkers(ctx, load_dir, in.n_orb, p.real_walkers != 0);
            double en_shift = 0, v; bool any = false;
            { std::ifstream f(load_dir + "S.txt"); while (f >> v) { en_shift = v; any = true; } }
            if (!any) throw std::runtime_error("Error reading energy shift from last line of S.txt");
            ck(fries_frisys_restart(ctx, seed, en_shift, walkers, 0));       // last_norm = the loaded walker number (:251)
            std::cout << "loaded " << (long long)walkers << " walkers, shift " << en_shift << std::endl;
        }
        const uint32_t max_iter = kv.count("max_iter") ? (uint32_t)std::stoul(kv["max_iter"]) : 1000000u;
        std::ofstream num_file(rd + "projnum.txt", std::ofstream::app), den_file(rd + "projden.txt", std::ofstream::app), shift_file(rd + "S.txt", std::ofstream::app),
            walk_file(rd + "N.txt", std::ofstream::app), nonz_file(rd + "nnonz.txt", std::ofstream::app), ini_file(rd + "nini.txt", std::ofstream::app);
        if (!num_file.is_open()) throw std::runtime_error("Could not open file for writing in directory " + rd);
        num_file.precision(out_precision(kv)); den_file.precision(out_precision(kv)); shift_file.precision(out_precision(kv));
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
            if ((it + 1) % 1000 == 0) save_walkers(ctx, rd, in.n_orb, p.real_walkers != 0);       // :447-449
        }
        save_walkers(ctx, rd, in.n_orb, p.real_walkers != 0);
        {   // hash.dat (save_proc_hash, :130)
            std::vector<uint32_t> scr(2 * in.n_orb);
            ck(fries_get_scramblers(ctx, scr.data(), nullptr, scr.size()));
            std::ofstream fh(rd + "hash.dat", std::ios::binary);
            fh.write((const char *)scr.data(), (std::streamsize)(4 * scr.size()));
        }
        fries_ctx_destroy(ctx);
    } catch (std::exception &ex) { std::cerr << "\nException : " << ex.what() << "\n"; }
    return 0;
}
