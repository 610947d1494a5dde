// frisys_hh and frifull_hh on the MI355X engine (FRIES_bin/frisys_hh.cpp, FRIES_bin/frifull_hh.cpp) over the C ABI.
//
//   frisys_hh_hip --params_path P --vec_nonz N --max_dets N [--target T] [--initiator I] [--max_iter K] [--result_dir DIR/] [--seed S] [--device D]
//                 [--full 1]      (frifull_hh: the Hamiltonian applied in full instead of compressed)
//
// P is the reference's parameter file (parse_hh_input, FRIES/io_utils.cpp:320-405): the keywords n_elec, lat_len, n_dim, eps, U,
// omega, g, gs_energy, each on its own line followed by its value.  Outputs (frisys_hh.cpp:127-160, 330-347): projnum.txt,
// projden.txt every iteration, S.txt and norm.txt every 10 iterations, params.txt.
#include "driver_common.hpp"

int main(int argc, char **argv) {
    std::map<std::string, std::string> kv;
    try {
        kv = parse_kv(argc, argv);
        for (const char *k : {"params_path", "vec_nonz", "max_dets"}) if (!kv.count(k)) throw std::runtime_error(std::string("missing required option --") + k);
    } catch (std::exception &ex) { std::cerr << "\nError parsing command line: " << ex.what() << "\n\n"; return 1; }
    try {
        std::ifstream f(kv["params_path"]);
        if (!f.is_open()) throw std::runtime_error("Could not open file containing Hubbard-Holstein parameters");
        std::map<std::string, double> val;
        const char *keys[] = {"n_elec", "lat_len", "n_dim", "eps", "U", "omega", "g", "gs_energy"};
        for (const char *k : keys) {            // the reference insists on this order
            std::string word; double x;
            if (!(f >> word) || word != k || !(f >> x)) throw std::runtime_error(std::string("Could not find ") + k + " parameter in file containing Hubbard-Holstein parameters");
            val[k] = x;
        }
        if ((int)val["n_dim"] != 1) { fprintf(stderr, "Error: only 1-D Hubbard calculations supported right now.\n"); return 0; }
        const std::string rd = kv.count("result_dir") ? kv["result_dir"] : "./";
        fries_ctx *ctx = nullptr;
        ck(fries_ctx_create(&ctx, kv.count("device") ? std::stoi(kv["device"]) : 0));
        uint32_t seed = kv.count("seed") ? (uint32_t)std::stoul(kv["seed"]) : wall_clock_seed();
        std::cout << "seed on process 0 is " << seed << std::endl;
        fries_hh_params p{(uint32_t)val["n_elec"], (uint32_t)val["lat_len"], val["eps"], val["U"], val["omega"], val["g"], val["gs_energy"],
                          kv.count("target") ? std::stod(kv["target"]) : 0.0, kv.count("initiator") ? std::stod(kv["initiator"]) : 0.0,
                          (uint32_t)std::stoul(kv["vec_nonz"]), (uint32_t)std::stoul(kv["max_dets"]), seed,
                          (uint32_t)(kv.count("full") ? std::stoul(kv["full"]) != 0 : 0)};
        ck(fries_hh_setup(ctx, &p));
        const uint32_t max_iter = kv.count("max_iter") ? (uint32_t)std::stoul(kv["max_iter"]) : 1000000u;
        std::ofstream num_file(rd + "projnum.txt", std::ofstream::app), den_file(rd + "projden.txt", std::ofstream::app), shift_file(rd + "S.txt", std::ofstream::app),
            norm_file(rd + "norm.txt", std::ofstream::app);
        if (!num_file.is_open()) throw std::runtime_error("Could not open file for writing in directory " + rd);
        num_file.precision(out_precision(kv)); den_file.precision(out_precision(kv)); shift_file.precision(out_precision(kv)); norm_file.precision(out_precision(kv));
        {
            std::ofstream param_f(rd + "params.txt");
            param_f << "FRI calculation\nHubbard-Holstein parameters path: " << kv["params_path"] << "\nepsilon (imaginary time step): " << p.eps << "\nTarget norm " << p.target_norm
                    << "\nInitiator threshold: " << p.initiator << "\nVector nonzero: " << p.vec_nonz << "\nInitializing calculation from Neel unit vector\n";
        }
        for (uint32_t it = 0; it < max_iter; it++) {
            fries_iter_log lg;
            ck(fries_hh_iterate(ctx, 1, &lg));
            if ((it + 1) % 10 == 0) { shift_file << lg.shift << '\n'; norm_file << lg.norm << '\n'; }
            num_file << lg.numer << '\n'; den_file << lg.denom << '\n';
            std::cout << it << ", norm: " << lg.norm << ", en est: " << lg.numer / lg.denom << ", shift: " << lg.shift << ", n_neel: " << lg.denom << '\n';
        }
        fries_ctx_destroy(ctx);
    } catch (std::exception &ex) { std::cerr << "\nException : " << ex.what() << "\n"; }
    return 0;
}
