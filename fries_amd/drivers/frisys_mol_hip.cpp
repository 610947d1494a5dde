// frisys_mol on the MI355X engine: the reference driver's command line, inputs and output files
// (FRIES_bin/frisys_mol.cpp) over the C ABI of libfries_hip.so (include/fries_hip.h).
//
//   frisys_mol_hip --fcidump_path F --point_group D2h --distribution HB_unnorm --vec_nonz N --mat_nonz N --max_dets N
//                  [--target T] [--initiator I] [--epsilon E] [--max_iter K] [--result_dir DIR/] [--load_dir DIR/]
//                  [--ini_vec PREFIX] [--trial_vec PREFIX] [--ham_shift E] [--seed S] [--device D]
//
// Host side only: option parsing (argparse there, a loop here), the FCIDUMP reader (parse_fcidump / convert_symm,
// FRIES/io_utils.cpp:189-318), the text outputs projnum.txt / projden.txt / S.txt / norm.txt / nkept.txt / params.txt
// (frisys_mol.cpp:288-345, 505-531) and the binary checkpoint dets0.dat / vals0.dat / dense.txt / hash.dat
// (DistVec::save / load, FRIES/vec_utils.hpp:703-844; save_proc_hash, io_utils.cpp:589-606).  Everything numeric runs on
// the GPU.  --ini_vec / --trial_vec read the reference's text vectors (<prefix>dets, <prefix>vals; load_vec_txt, io_utils.cpp:447-482).
// --det_space (the semi-stochastic space) is not implemented.  One rank; ranks are driven through fries_set_comm.
#include "driver_common.hpp"
#include <sstream>
#include <random>

struct Args {
    std::string fcidump_path, point_group = "C1", dist = "HB_unnorm", result_dir = "./", load_dir, ini_vec, trial_vec;
    bool have_ham_shift = false; double ham_shift = 0;
    double target = 0, initiator = 0, epsilon = 0.01;
    uint32_t max_iter = 1000000, vec_nonz = 0, mat_nonz = 0, max_dets = 0, seed = 0, device = 0;
    bool have_seed = false;
};

static Args parse_args(int argc, char **argv) {
    std::map<std::string, std::string> kv = parse_kv(argc, argv);
    Args r;
    auto need = [&](const char *k) { if (!kv.count(k)) throw std::runtime_error(std::string("missing required option --") + k); return kv[k]; };
    r.fcidump_path = need("fcidump_path"); r.vec_nonz = (uint32_t)std::stoul(need("vec_nonz")); r.mat_nonz = (uint32_t)std::stoul(need("mat_nonz"));
    r.max_dets = (uint32_t)std::stoul(need("max_dets"));
    if (kv.count("point_group")) r.point_group = kv["point_group"];
    if (kv.count("distribution")) r.dist = kv["distribution"];
    if (kv.count("result_dir")) r.result_dir = kv["result_dir"];
    if (kv.count("load_dir")) r.load_dir = kv["load_dir"];
    if (kv.count("ini_vec")) r.ini_vec = kv["ini_vec"];
    if (kv.count("trial_vec")) r.trial_vec = kv["trial_vec"];
    if (kv.count("ham_shift")) { r.ham_shift = std::stod(kv["ham_shift"]); r.have_ham_shift = true; }
    if (kv.count("det_space")) throw std::runtime_error("--det_space (semi-stochastic space) is not implemented in frisys_mol_hip");
    if (kv.count("target")) r.target = std::stod(kv["target"]);
    if (kv.count("initiator")) r.initiator = std::stod(kv["initiator"]);
    if (kv.count("epsilon")) r.epsilon = std::stod(kv["epsilon"]);
    if (kv.count("max_iter")) r.max_iter = (uint32_t)std::stoul(kv["max_iter"]);
    if (kv.count("seed")) { r.seed = (uint32_t)std::stoul(kv["seed"]); r.have_seed = true; }
    if (kv.count("device")) r.device = (uint32_t)std::stoul(kv["device"]);
    return r;
}

// DistVec::save (vec_utils.hpp:703-737): raw index bytes, then the value columns; dense.txt; hash.dat is written by the caller
static void save_vector(fries_ctx *ctx, const std::string &dir, unsigned n_orb) {
    uint32_t n = 0; int32_t nz; uint32_t nf;
    ck(fries_vec_info(ctx, &n, &nz, &nf));
    std::vector<uint64_t> dets(n ? n : 1); std::vector<double> vals(n ? n : 1);
    size_t m = 0;
    ck(fries_vec_download(ctx, dets.data(), vals.data(), dets.size(), &m));
    const size_t n_bytes = (2 * n_orb + 7) / 8;
    std::ofstream fd(dir + "dets0.dat", std::ios::binary);
    for (size_t i = 0; i < m; i++) fd.write((const char *)&dets[i], (std::streamsize)n_bytes);     // little-endian byte string, det_store.h:23-26
    std::ofstream fv(dir + "vals0.dat", std::ios::binary);
    fv.write((const char *)vals.data(), (std::streamsize)(8 * m));
    std::vector<double> zeros(m, 0.0);                                  // column 1 is zero between iterations (frisys_mol.cpp:498)
    fv.write((const char *)zeros.data(), (std::streamsize)(8 * m));
    std::ofstream fx(dir + "dense.txt");
    fx << 0 << '\n';
}

// DistVec::load (:739-844): positions 0.. hold the stored elements with |value| > 1e-9, in file order
static size_t load_vector(fries_ctx *ctx, const std::string &dir, unsigned n_orb) {
    const size_t n_bytes = (2 * n_orb + 7) / 8;
    std::ifstream fd(dir + "dets0.dat", std::ios::binary | std::ios::ate);
    if (!fd.is_open()) throw std::runtime_error("Could not open saved binary vector file at path " + dir + "dets0.dat");
    size_t n = (size_t)fd.tellg() / n_bytes;
    fd.seekg(0);
    std::vector<uint64_t> dets(n, 0); std::vector<double> vals(n);
    for (size_t i = 0; i < n; i++) fd.read((char *)&dets[i], (std::streamsize)n_bytes);
    std::ifstream fv(dir + "vals0.dat", std::ios::binary);
    if (!fv.is_open()) throw std::runtime_error("Could not open saved binary vector file at path " + dir + "vals0.dat");
    fv.read((char *)vals.data(), (std::streamsize)(8 * n));
    std::vector<uint64_t> d2; std::vector<double> v2;
    for (size_t i = 0; i < n; i++) if (fabs(vals[i]) > 1e-9) { d2.push_back(dets[i]); v2.push_back(vals[i]); }
    ck(fries_vec_load(ctx, d2.data(), v2.data(), d2.size()));
    return d2.size();
}

static bool load_last_line(const std::string &path, double *out) {
    std::ifstream f(path);
    double v; bool any = false;
    while (f >> v) { *out = v; any = true; }
    return any;
}

int main(int argc, char **argv) {
    Args args;
    try { args = parse_args(argc, argv); if (args.dist != "HB" && args.dist != "HB_unnorm") throw std::runtime_error("\"dist_str\" argument must be either \"HB\" or \"HB_unnorm\""); }
    catch (std::exception &ex) { std::cerr << "\nError parsing command line: " << ex.what() << "\n\n"; return 1; }
    try {
        Fcidump in = parse_fcidump(args.fcidump_path, args.point_group);
        fries_ctx *ctx = nullptr;
        ck(fries_ctx_create(&ctx, (int)args.device));
        ck(fries_set_molecule(ctx, in.n_orb, in.n_elec, in.symm.data(), in.hcore.data(), in.eris.data()));
        uint32_t seed = args.seed;
        if (!args.have_seed) seed = wall_clock_seed();      // frisys_mol.cpp:104-106
        std::cout << "seed on process 0 is " << seed << std::endl;
        fries_frisys_params p{args.epsilon, args.target, args.initiator, args.vec_nonz, args.mat_nonz, args.max_dets, seed, args.dist == "HB_unnorm" ? 1 : 0};
        std::vector<uint64_t> tdets; std::vector<double> tvals;
        if (!args.trial_vec.empty()) { load_vec_txt(args.trial_vec, tdets, tvals); ck(fries_set_trial_vector(ctx, tdets.data(), tvals.data(), tvals.size())); }      // :157-181
        if (args.load_dir.empty() && !args.ini_vec.empty()) { load_vec_txt(args.ini_vec, tdets, tvals); ck(fries_set_initial_vector(ctx, tdets.data(), tvals.data(), tvals.size())); }   // :264-274
        if (args.have_ham_shift) ck(fries_set_ham_shift(ctx, args.ham_shift - in.core_en));      // :95-98
        if (!args.load_dir.empty()) {                       // :128-130 load_proc_hash: the shards of the run that wrote the checkpoint
            std::ifstream fh(args.load_dir + "hash.dat", std::ios::binary);
            if (!fh.is_open()) throw std::runtime_error("Error: could not open saved hash scrambler at " + args.load_dir + "hash.dat");
            std::vector<uint32_t> scr(2 * in.n_orb);
            fh.read((char *)scr.data(), (std::streamsize)(4 * scr.size()));
            ck(fries_set_proc_scrambler(ctx, scr.data(), scr.size()));
        }
        ck(fries_frisys_setup(ctx, &p));
        if (!args.load_dir.empty()) {                       // :257-263
            double en_shift = 0;
            load_vector(ctx, args.load_dir, in.n_orb);
            load_last_line(args.load_dir + "S.txt", &en_shift);
            // The loaded one-norm goes into `last_norm` (:284-286), which nothing reads: the shift control starts from
            // last_one_norm = 0 (:337), i.e. the shift stays put until a shift iteration sees the norm above the target.
            ck(fries_frisys_restart(ctx, seed, en_shift, 0.0, 0));
            // Generator: seeded, then the 2 n_orb draws of the vec scrambler (:141-144) -- with --load_dir the proc scrambler
            // takes none -- and the iterations continue that stream.
            std::mt19937 mt(seed);
            mt.discard(2 * in.n_orb);
            std::ostringstream os;
            os << mt;
            ck(fries_rng_set_state(ctx, os.str().c_str()));
        }
        const std::string &rd = args.result_dir;
        std::ofstream num_file(rd + "projnum.txt", std::ofstream::app), den_file(rd + "projden.txt", std::ofstream::app),
            shift_file(rd + "S.txt", std::ofstream::app), norm_file(rd + "norm.txt", std::ofstream::app), nkept_file(rd + "nkept.txt", std::ofstream::app);
        if (!num_file.is_open()) throw std::runtime_error("Could not open file for writing in directory " + rd);
        num_file.precision(17); den_file.precision(17); shift_file.precision(17); norm_file.precision(17);
        {
            std::ofstream param_f(rd + "params.txt");
            param_f << "FRI calculation\nFCIDUMP path: " << args.fcidump_path << "\nepsilon (imaginary time step): " << args.epsilon << "\nTarget norm " << args.target
                    << "\nInitiator threshold: " << args.initiator << "\nMatrix nonzero: " << args.mat_nonz << "\nVector nonzero: " << args.vec_nonz << "\n";
            if (!args.load_dir.empty()) param_f << "Restarting calculation from " << args.load_dir << "\n";
            else if (!args.ini_vec.empty()) param_f << "Initializing calculation from vector files with prefix " << args.ini_vec << '\n';
            else param_f << "Initializing calculation from HF unit vector\n";
        }
        {   // hash.dat: the proc scrambler (io_utils.cpp:589-606); one rank never uses it, but a restart on several ranks would
            std::vector<uint32_t> scr(2 * in.n_orb);
            ck(fries_get_scramblers(ctx, scr.data(), nullptr, scr.size()));
            std::ofstream fh(rd + "hash.dat", std::ios::binary);
            fh.write((const char *)scr.data(), (std::streamsize)(4 * scr.size()));
        }
        const unsigned shift_interval = 10, save_interval = 100;
        for (uint32_t it = 0; it < args.max_iter; it++) {
            fries_iter_log lg;
            ck(fries_frisys_iterate(ctx, 1, &lg));
            num_file << lg.numer << '\n'; den_file << lg.denom << '\n'; nkept_file << lg.nkept << '\n';
            if ((it + 1) % shift_interval == 0) { shift_file << lg.shift << '\n'; norm_file << lg.norm << '\n'; }
            std::cout << it << ", en est: " << lg.numer / lg.denom << ", shift: " << lg.shift << ", norm: " << lg.norm << '\n';      // :518-520
            if ((it + 1) % save_interval == 0) {
                save_vector(ctx, rd, in.n_orb);
                num_file.flush(); den_file.flush(); shift_file.flush(); nkept_file.flush();
            }
        }
        save_vector(ctx, rd, in.n_orb);
        fries_ctx_destroy(ctx);
    } catch (std::exception &ex) {
        std::cerr << "\nException : " << ex.what() << "\n";       // the reference prints and exits 0 (frisys_mol.cpp:562-566)
    }
    return 0;
}
