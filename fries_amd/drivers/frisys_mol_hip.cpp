// frisys_mol on the MI355X engine: the reference driver's command line, inputs and output files
// (FRIES_bin/frisys_mol.cpp) over the C ABI of libfries_hip.so (include/fries_hip.h).
//
//   frisys_mol_hip --fcidump_path F --point_group D2h --distribution HB_unnorm --vec_nonz N --mat_nonz N --max_dets N
//                  [--target T] [--initiator I] [--epsilon E] [--max_iter K] [--result_dir DIR/] [--load_dir DIR/]
//                  [--ini_vec PREFIX] [--trial_vec PREFIX] [--det_space FILE] [--ham_shift E] [--seed S] [--device D]
//
// Host side only: option parsing (argparse there, a loop here), the FCIDUMP reader (parse_fcidump / convert_symm,
// FRIES/io_utils.cpp:189-318), the text outputs projnum.txt / projden.txt / S.txt / norm.txt / nkept.txt / params.txt
// (frisys_mol.cpp:288-345, 505-531) and the binary checkpoint dets0.dat / vals0.dat / dense.txt / hash.dat
// (DistVec::save / load, FRIES/vec_utils.hpp:703-844; save_proc_hash, io_utils.cpp:589-606).  Everything numeric runs on
// the GPU.  --ini_vec / --trial_vec read the reference's text vectors (<prefix>dets, <prefix>vals; load_vec_txt, io_utils.cpp:447-482).
// --det_space FILE: the semi-stochastic dense space (every rank reads the file and keeps the determinants it owns; with --load_dir the
// dense space is the one the checkpoint's dense.txt records, as in the reference).
//
// Ranks (the reference under mpiexec -n P: hash-sharded vector, MPI_Alltoallv of the adds, rank-ordered sums):
//   * one process per MI355X over librccl: start P copies with RANK / WORLD_SIZE / LOCAL_RANK in the environment (torchrun's names;
//     OMPI_COMM_WORLD_RANK / _SIZE / _LOCAL_RANK and PMI_RANK / PMI_SIZE are understood too) and the same --result_dir: rank 0
//     writes the RCCL id to <result_dir>.rccl_id[.<launch nonce>], the others wait for it; rank r uses GPU LOCAL_RANK (or --device);
//   * --ranks P: P host threads of this one process (fries_local_*), each with its own context, on --device (several ranks may
//     share a GPU) or on --devices 0,1,...: for machines with fewer GPUs than ranks, and for tests.
// Every rank writes dets<rank>.dat / vals<rank>.dat; the rank that owns the HF determinant writes the text outputs, rank 0
// dense.txt and hash.dat (frisys_mol.cpp:288-345, vec_utils.hpp:713-746).  All ranks must be given the same --seed.
#include "driver_common.hpp"
#include <sstream>
#include <random>
#include <thread>
#include <unistd.h>

struct Args {
    int precision = 17;      // digits in the output files (--precision; 6 = the reference's stream format)
    std::string fcidump_path, point_group = "C1", dist = "HB_unnorm", result_dir = "./", load_dir, ini_vec, trial_vec, det_space;
    bool have_ham_shift = false; double ham_shift = 0;
    double target = 0, initiator = 0, epsilon = 0.01;
    uint32_t max_iter = 1000000, vec_nonz = 0, mat_nonz = 0, max_dets = 0, seed = 0, device = 0;
    bool have_seed = false, have_device = false;
    uint32_t ranks = 0;                 // --ranks: thread ranks inside this process
    std::vector<int> devices;           // --devices: GPU of every thread rank
};

static Args parse_args(int argc, char **argv) {
    std::map<std::string, std::string> kv = parse_kv(argc, argv);
    Args r;
    auto need = [&](const char *k) { if (!kv.count(k)) throw std::runtime_error(std::string("missing required option --") + k); return kv[k]; };
    r.fcidump_path = need("fcidump_path"); r.vec_nonz = (uint32_t)std::stoul(need("vec_nonz")); r.mat_nonz = (uint32_t)std::stoul(need("mat_nonz"));
    r.max_dets = (uint32_t)std::stoul(need("max_dets"));
    if (kv.count("point_group")) r.point_group = kv["point_group"];
    r.precision = out_precision(kv);
    if (kv.count("distribution")) r.dist = kv["distribution"];
    if (kv.count("result_dir")) r.result_dir = kv["result_dir"];
    if (kv.count("load_dir")) r.load_dir = kv["load_dir"];
    if (kv.count("ini_vec")) r.ini_vec = kv["ini_vec"];
    if (kv.count("trial_vec")) r.trial_vec = kv["trial_vec"];
    if (kv.count("det_space")) r.det_space = kv["det_space"];
    if (kv.count("ham_shift")) { r.ham_shift = std::stod(kv["ham_shift"]); r.have_ham_shift = true; }
    if (kv.count("target")) r.target = std::stod(kv["target"]);
    if (kv.count("initiator")) r.initiator = std::stod(kv["initiator"]);
    if (kv.count("epsilon")) r.epsilon = std::stod(kv["epsilon"]);
    if (kv.count("max_iter")) r.max_iter = (uint32_t)std::stoul(kv["max_iter"]);
    if (kv.count("seed")) { r.seed = (uint32_t)std::stoul(kv["seed"]); r.have_seed = true; }
    if (kv.count("device")) { r.device = (uint32_t)std::stoul(kv["device"]); r.have_device = true; }
    if (kv.count("ranks")) r.ranks = (uint32_t)std::stoul(kv["ranks"]);
    if (kv.count("devices")) { std::stringstream ss(kv["devices"]); std::string t; while (std::getline(ss, t, ',')) r.devices.push_back(std::stoi(t)); }
    return r;
}

// DistVec::save (vec_utils.hpp:703-737): raw index bytes, then the value columns; dense.txt; hash.dat is written by the caller
static void save_vector(fries_ctx *ctx, const std::string &dir, unsigned n_orb, int rank = 0, int n_ranks = 1) {
    uint32_t n = 0; int32_t nz; uint32_t nf;
    ck(fries_vec_info(ctx, &n, &nz, &nf));
    std::vector<uint64_t> dets(n ? n : 1); std::vector<double> vals(n ? n : 1);
    size_t m = 0;
    ck(fries_vec_download(ctx, dets.data(), vals.data(), dets.size(), &m));
    const size_t n_bytes = (2 * n_orb + 7) / 8;
    const std::string rk = std::to_string(rank);
    std::ofstream fd(dir + "dets" + rk + ".dat", std::ios::binary);
    for (size_t i = 0; i < m; i++) fd.write((const char *)&dets[i], (std::streamsize)n_bytes);     // little-endian byte string, det_store.h:23-26
    std::ofstream fv(dir + "vals" + rk + ".dat", std::ios::binary);
    fv.write((const char *)vals.data(), (std::streamsize)(8 * m));
    std::vector<double> zeros(m, 0.0);                                  // column 1 is zero between iterations (frisys_mol.cpp:498)
    fv.write((const char *)zeros.data(), (std::streamsize)(8 * m));
    if (rank == 0) {        // one dense-space size per rank (vec_utils.hpp:736-745)
        std::vector<uint32_t> ds((size_t)n_ranks, 0u);
        size_t nr = 0;
        ck(fries_dense_sizes(ctx, ds.data(), ds.size(), &nr));
        std::ofstream fx(dir + "dense.txt");
        for (int p = 0; p < n_ranks - 1; p++) fx << ds[p] << ",";
        fx << ds[n_ranks - 1] << '\n';
    }
}

// DistVec::load (:761-844): dense.txt gives every rank its n_dense; positions 0.. hold the first n_dense stored elements whatever their value,
// then those with |value| > 1e-9, in file order; the dense space is declared again on the device (fries_vec_set_dense: H inside it, the budget)
static size_t load_vector(fries_ctx *ctx, const std::string &dir, unsigned n_orb, int rank = 0) {
    const size_t n_bytes = (2 * n_orb + 7) / 8;
    const std::string rk = std::to_string(rank);
    size_t n_dense = 0;
    {
        std::ifstream fx(dir + "dense.txt");       // read_csv (io_utils.cpp:75-113): comma-separated integers, one per rank
        std::string tok; int col = 0;
        while (fx.is_open() && std::getline(fx, tok, ',')) {
            if (tok.find_first_of("0123456789") == std::string::npos) continue;
            if (col == rank) n_dense = (size_t)std::stoul(tok);
            col++;
        }
    }
    std::ifstream fd(dir + "dets" + rk + ".dat", std::ios::binary | std::ios::ate);
    if (!fd.is_open()) throw std::runtime_error("Could not open saved binary vector file at path " + dir + "dets" + rk + ".dat");
    size_t n = (size_t)fd.tellg() / n_bytes;
    fd.seekg(0);
    std::vector<uint64_t> dets(n, 0); std::vector<double> vals(n);
    for (size_t i = 0; i < n; i++) fd.read((char *)&dets[i], (std::streamsize)n_bytes);
    std::ifstream fv(dir + "vals" + rk + ".dat", std::ios::binary);
    if (!fv.is_open()) throw std::runtime_error("Could not open saved binary vector file at path " + dir + "vals" + rk + ".dat");
    fv.read((char *)vals.data(), (std::streamsize)(8 * n));
    if (n_dense > n) throw std::runtime_error("dense.txt names more dense determinants than " + dir + "dets" + rk + ".dat holds");
    std::vector<uint64_t> d2; std::vector<double> v2;
    for (size_t i = 0; i < n; i++) if (i < n_dense || fabs(vals[i]) > 1e-9) { d2.push_back(dets[i]); v2.push_back(vals[i]); }
    ck(fries_vec_load(ctx, d2.data(), v2.data(), d2.size()));
    ck(fries_vec_set_dense(ctx, (uint32_t)n_dense));        // every rank calls it (collective), also with n_dense == 0
    return d2.size();
}

static bool load_last_line(const std::string &path, double *out) {
    std::ifstream f(path);
    double v; bool any = false;
    while (f >> v) { *out = v; any = true; }
    return any;
}

// one rank of the run (the body of the reference's main after MPI_Init): `tr` == nullptr is the one-rank run
static void run_rank(const Args &args, const Fcidump &in, uint32_t seed, int rank, int n_ranks, int device, fries_transport *tr) {
    fries_ctx *ctx = nullptr;
    ck(fries_ctx_create(&ctx, device));
    ck(fries_set_molecule(ctx, in.n_orb, in.n_elec, in.symm.data(), in.hcore.data(), in.eris.data()));
    if (tr) {
        fries_comm cm;
        ck(fries_transport_comm(tr, &cm));
        ck(fries_set_comm(ctx, &cm));
    }
    fries_frisys_params p{args.epsilon, args.target, args.initiator, args.vec_nonz, args.mat_nonz, args.max_dets, seed, args.dist == "HB_unnorm" ? 1 : 0};
    std::vector<uint64_t> tdets; std::vector<double> tvals;
    if (!args.trial_vec.empty()) { load_vec_txt(args.trial_vec, tdets, tvals); ck(fries_set_trial_vector(ctx, tdets.data(), tvals.data(), tvals.size())); }      // :157-181
    if (args.load_dir.empty() && !args.ini_vec.empty()) { load_vec_txt(args.ini_vec, tdets, tvals); ck(fries_set_initial_vector(ctx, tdets.data(), tvals.data(), tvals.size())); }   // :264-274
    if (args.have_ham_shift) ck(fries_set_ham_shift(ctx, args.ham_shift - in.core_en));      // :95-98
    if (args.load_dir.empty() && !args.det_space.empty()) {     // --det_space (:236-239): the integers read_dets reads (io_utils.cpp:565-586)
        std::ifstream f(args.det_space);
        if (!f.is_open()) throw std::runtime_error("Could not open file: " + args.det_space);
        std::vector<uint64_t> space;
        long long d;
        while (f >> d) space.push_back((uint64_t)d);
        ck(fries_set_det_space(ctx, space.data(), space.size()));
    }
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
        load_vector(ctx, args.load_dir, in.n_orb, rank);
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
    // the rank that owns the HF determinant keeps the text outputs (:288-333)
    int32_t hf_proc = 0;
    {
        uint64_t hf = 0;
        for (unsigned k = 0; k < in.n_elec / 2; k++) hf |= (1ull << k) | (1ull << (k + in.n_orb));
        ck(fries_idx_to_proc(ctx, &hf, 1, &hf_proc));
    }
    const bool writer = rank == hf_proc;
    const std::string &rd = args.result_dir;
    std::ofstream num_file, den_file, shift_file, norm_file, nkept_file;
    if (writer) {
        num_file.open(rd + "projnum.txt", std::ofstream::app); den_file.open(rd + "projden.txt", std::ofstream::app);
        shift_file.open(rd + "S.txt", std::ofstream::app); norm_file.open(rd + "norm.txt", std::ofstream::app); nkept_file.open(rd + "nkept.txt", std::ofstream::app);
        if (!num_file.is_open()) throw std::runtime_error("Could not open file for writing in directory " + rd);
        num_file.precision(args.precision); den_file.precision(args.precision); shift_file.precision(args.precision); norm_file.precision(args.precision);
        std::ofstream param_f(rd + "params.txt");
        param_f << "FRI calculation\nFCIDUMP path: " << args.fcidump_path << "\nepsilon (imaginary time step): " << args.epsilon << "\nTarget norm " << args.target
                << "\nInitiator threshold: " << args.initiator << "\nMatrix nonzero: " << args.mat_nonz << "\nVector nonzero: " << args.vec_nonz << "\n";
        if (!args.load_dir.empty()) param_f << "Restarting calculation from " << args.load_dir << "\n";
        else if (!args.ini_vec.empty()) param_f << "Initializing calculation from vector files with prefix " << args.ini_vec << '\n';
        else param_f << "Initializing calculation from HF unit vector\n";
    }
    if (rank == 0) {   // hash.dat: the proc scrambler (io_utils.cpp:589-606), what a restart shards by
        std::vector<uint32_t> scr(2 * in.n_orb);
        ck(fries_get_scramblers(ctx, scr.data(), nullptr, scr.size()));
        std::ofstream fh(rd + "hash.dat", std::ios::binary);
        fh.write((const char *)scr.data(), (std::streamsize)(4 * scr.size()));
    }
    const unsigned shift_interval = 10, save_interval = 100;
    for (uint32_t it = 0; it < args.max_iter; it++) {
        fries_iter_log lg;
        ck(fries_frisys_iterate(ctx, 1, &lg));
        if (writer) {
            num_file << lg.numer << '\n'; den_file << lg.denom << '\n'; nkept_file << lg.nkept << '\n';
            if ((it + 1) % shift_interval == 0) { shift_file << lg.shift << '\n'; norm_file << lg.norm << '\n'; }
            std::cout << it << ", en est: " << lg.numer / lg.denom << ", shift: " << lg.shift << ", norm: " << lg.norm << '\n';      // :518-520
        }
        if ((it + 1) % save_interval == 0) {
            save_vector(ctx, rd, in.n_orb, rank, n_ranks);
            if (writer) { num_file.flush(); den_file.flush(); shift_file.flush(); nkept_file.flush(); }
        }
    }
    save_vector(ctx, rd, in.n_orb, rank, n_ranks);
    fries_ctx_destroy(ctx);
}

// rank / size / local rank from the launcher's environment (torchrun, Open MPI, PMI); -1: not launched as ranks
static int env_int(const char *a, const char *b, const char *c) {
    for (const char *k : {a, b, c}) if (k && getenv(k)) return atoi(getenv(k));
    return -1;
}

int main(int argc, char **argv) {
    Args args;
    try { args = parse_args(argc, argv); if (args.dist != "HB" && args.dist != "HB_unnorm") throw std::runtime_error("\"dist_str\" argument must be either \"HB\" or \"HB_unnorm\""); }
    catch (std::exception &ex) { std::cerr << "\nError parsing command line: " << ex.what() << "\n\n"; return 1; }
    try {
        Fcidump in = parse_fcidump(args.fcidump_path, args.point_group);
        uint32_t seed = args.seed;
        const int env_size = env_int("WORLD_SIZE", "OMPI_COMM_WORLD_SIZE", "PMI_SIZE");
        if ((args.ranks > 1 || env_size > 1) && !args.have_seed) throw std::runtime_error("several ranks need the same --seed on every rank (the reference broadcasts rank 0's draws)");
        if (!args.have_seed) seed = wall_clock_seed();      // frisys_mol.cpp:104-106
        const uint64_t big_bytes = 16ull * ((uint64_t)args.mat_nonz + 4096);
        if (args.ranks > 1) {
            // P ranks = P threads of this process (fries_local_*)
            const int P = (int)args.ranks;
            std::cout << "seed on process 0 is " << seed << std::endl;
            fries_local_group *grp = nullptr;
            ck(fries_local_group_create(&grp, P, big_bytes));
            std::vector<fries_transport *> tr(P, nullptr);
            for (int r = 0; r < P; r++) ck(fries_local_create(&tr[r], grp, r, args.devices.empty() ? (int)args.device : args.devices[r % args.devices.size()]));
            std::vector<std::string> errs(P);
            std::vector<std::thread> th;
            for (int r = 0; r < P; r++)
                th.emplace_back([&, r] {
                    try { run_rank(args, in, seed, r, P, args.devices.empty() ? (int)args.device : args.devices[r % args.devices.size()], tr[r]); }
                    catch (std::exception &ex) { errs[r] = ex.what(); }
                });
            for (auto &t : th) t.join();
            for (int r = 0; r < P; r++) { if (!errs[r].empty()) std::cerr << "\nException on rank " << r << " : " << errs[r] << "\n"; fries_transport_destroy(tr[r]); }
            fries_local_group_destroy(grp);
        }
        else if (env_size >= 1) {
            // one process per GPU, librccl; the 128-byte communicator id travels through a file in the (shared) result directory
            const int rank = env_int("RANK", "OMPI_COMM_WORLD_RANK", "PMI_RANK"), lrank = env_int("LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", nullptr);
            if (rank < 0 || rank >= env_size) throw std::runtime_error("WORLD_SIZE is set but RANK is not a rank of it");
            const int device = args.have_device ? (int)args.device : (lrank >= 0 ? lrank : rank);
            if (rank == 0) std::cout << "seed on process 0 is " << seed << std::endl;
            // The file's name carries a per-launch nonce (the launcher's rendezvous port or job id) so that the id a run left behind when it died
            // between publishing and removing it is never taken for this run's; rank 0 also removes any file of that name before it publishes.
            std::string nonce;
            for (const char *k : {"MASTER_PORT", "TORCHELASTIC_RUN_ID", "SLURM_JOB_ID", "PMI_ID", "OMPI_MCA_ess_base_jobid"}) if (const char *v = getenv(k)) { nonce = std::string(".") + v; break; }
            const std::string idf = args.result_dir + ".rccl_id" + nonce, tmpf = idf + ".tmp";
            uint8_t id[128];
            if (rank == 0) {
                remove(idf.c_str());
                ck(fries_rccl_unique_id(id));
                { std::ofstream f(tmpf, std::ios::binary); f.write((const char *)id, 128); }
                if (rename(tmpf.c_str(), idf.c_str())) throw std::runtime_error("cannot publish the RCCL id at " + idf);
            }
            else {
                bool got = false;
                for (int tries = 0; tries < 6000 && !got; tries++) {
                    std::ifstream f(idf, std::ios::binary);
                    if (f.is_open() && f.read((char *)id, 128) && f.gcount() == 128) got = true;
                    else usleep(10000);
                }
                if (!got) throw std::runtime_error("rank 0 never published the RCCL id at " + idf);
            }
            fries_transport *tr = nullptr;
            ck(fries_rccl_create(&tr, id, rank, env_size, device, big_bytes));       // collective: returns once every rank has joined
            if (rank == 0) remove(idf.c_str());
            try { run_rank(args, in, seed, rank, env_size, device, tr); }
            catch (std::exception &ex) {
                // The other ranks sit in their next collective, which has no timeout: leave at once with a failure code (no destructor, no
                // communicator teardown that would itself wait for them) so that the launcher tears the job down.
                std::cerr << "\nException on rank " << rank << " : " << ex.what() << "\n" << std::flush;
                _exit(3);
            }
            fries_transport_destroy(tr);
        }
        else {
            std::cout << "seed on process 0 is " << seed << std::endl;
            run_rank(args, in, seed, 0, 1, (int)args.device, nullptr);
        }
    } catch (std::exception &ex) {
        std::cerr << "\nException : " << ex.what() << "\n";       // the reference prints and exits 0 (frisys_mol.cpp:562-566)
    }
    return 0;
}
