// Host-side pieces shared by the command-line drivers: status check, the FCIDUMP reader (parse_fcidump / convert_symm,
// FRIES/io_utils.cpp:189-318) and "--option value" parsing (argparse in the reference).
#pragma once
#include "../../include/fries_hip.h"
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

static void ck(int rc) { if (rc) throw std::runtime_error(fries_last_error()); }

struct Fcidump { uint32_t n_orb = 0, n_elec = 0; double core_en = 0; std::vector<uint8_t> symm; std::vector<double> hcore, eris; };

// io_utils.cpp:189-239
static void convert_symm(std::vector<uint8_t> &irreps, const std::string &pg_in) {
    std::string pg = pg_in;
    for (auto &ch : pg) ch = (char)tolower(ch);
    std::vector<uint8_t> map;
    unsigned max_label;
    if (pg == "d2h") { map = {0, 7, 6, 1, 5, 2, 3, 4}; max_label = 8; }
    else if (pg == "c2v" || pg == "c2h") { map = {0, 2, 3, 1}; max_label = 4; }
    else if (pg == "d2") { map = {0, 3, 2, 1}; max_label = 4; }
    else if (pg == "cs" || pg == "c2" || pg == "ci" || pg == "c1") { map = {0, 1}; max_label = 2; }
    else throw std::runtime_error("Point group " + pg_in + " not recognized");
    for (auto &ir : irreps) {
        if (ir > max_label || ir == 0) {
            std::stringstream msg;
            msg << "irrep index " << (unsigned)ir << " read from the FCIDUMP file exceeds the maximum allowed irrep index (" << max_label << ") for point group " << pg_in;
            throw std::runtime_error(msg.str());
        }
        ir = map[ir - 1];
    }
}

static size_t tri(size_t i, size_t j) { return i <= j ? j * (j + 1) / 2 + i : i * (i + 1) / 2 + j; }

// io_utils.cpp:241-318: line 1 NORB / NELEC / MS2, line 2 ORBSYM, two more header lines, then "value i j k l" records
static Fcidump parse_fcidump(const std::string &path, const std::string &point_group) {
    std::ifstream in(path);
    if (!in.is_open()) throw std::runtime_error("Could not open FCIDUMP file " + path);
    std::string line;
    std::getline(in, line);
    auto field = [&](const char *key) {
        size_t p = line.find(key);
        if (p == std::string::npos) throw std::runtime_error(std::string("FCIDUMP header lacks ") + key);
        size_t e = line.find(",", p);
        return std::stoi(line.substr(p + strlen(key), e - (p + strlen(key))));
    };
    Fcidump f;
    f.n_orb = (uint32_t)field("NORB="); f.n_elec = (uint32_t)field("NELEC=");
    if (field("MS2=") != 0) throw std::runtime_error("MS2 is not zero in FCIDUMP file.");
    std::getline(in, line);
    size_t op = line.find("ORBSYM=");
    if (op == std::string::npos) throw std::runtime_error("ORBSYM missing on line 2 of the FCIDUMP file");
    std::stringstream ss(line.substr(op + 7));
    std::string tok;
    while (std::getline(ss, tok, ',')) { try { if (!tok.empty()) f.symm.push_back((uint8_t)std::stoi(tok)); } catch (std::invalid_argument &) {} }
    if (f.symm.size() != f.n_orb) throw std::runtime_error("Number of irrep labels read in after ORBSYM in FCIDUMP file does not equal number of orbitals");
    convert_symm(f.symm, point_group);
    std::getline(in, line);     // ISYM
    std::getline(in, line);     // &END
    const size_t n = f.n_orb, np = n * (n + 1) / 2;
    f.hcore.assign(n * n, 0.0); f.eris.assign(np * (np + 1) / 2, 0.0);
    double v; unsigned o[4];
    while (in >> v >> o[0] >> o[1] >> o[2] >> o[3]) {
        if (!o[0] && !o[1] && !o[2] && !o[3]) f.core_en = v;
        else if (!o[1] && !o[2] && !o[3]) continue;                  // orbital energy
        else if (!o[2] && !o[3]) f.hcore[(o[0] - 1) * n + (o[1] - 1)] = f.hcore[(o[1] - 1) * n + (o[0] - 1)] = v;
        else { size_t p1 = tri(o[0] - 1, o[1] - 1), p2 = tri(o[2] - 1, o[3] - 1); f.eris[tri(p1, p2)] = v; }     // 8-fold packed, ndarr.hpp:206-244
    }
    return f;
}

// The legacy HF-output directory (parse_hf_input, FRIES/io_utils.cpp:98-187; still what frifull_mol / frimulti_mol take as --hf_path):
// sys_params.txt (n_elec, n_frozen, n_orb, eps, hf_energy as keyword / value line pairs), symm.txt (irreps, already in the library's
// labels, one per orbital incl. the frozen ones), hcore.txt (tot_orb^2 values), eris.txt (tot_orb^4 values, eris[i][j][k][l] =
// <ij|kl> in physicists' order, FRIES/ndarr.hpp:150-195); values separated by commas and / or line breaks.  The engine works
// without frozen orbitals (the FCIDUMP drivers hard-code n_frz = 0, frisys_mol.cpp:79): a directory with n_frozen > 0 is refused
// rather than folded, because the reference sums the frozen orbitals inside every matrix element in its own order.
static std::vector<double> read_csv_doubles(const std::string &path) {
    std::ifstream f(path);
    if (!f.is_open()) throw std::runtime_error("Could not open file " + path);
    std::vector<double> out;
    std::string line, tok;
    while (std::getline(f, line)) {
        std::stringstream ss(line);
        while (std::getline(ss, tok, ',')) { std::stringstream num(tok); double v; if (num >> v) out.push_back(v); }
    }
    return out;
}
struct HfDir { Fcidump mol; double eps = 0, hf_en = 0; unsigned n_frz = 0; };
static HfDir parse_hf_dir(const std::string &dir) {
    HfDir r;
    std::ifstream in(dir + "sys_params.txt");
    if (!in.is_open()) throw std::runtime_error("Could not open file sys_params.txt");
    auto kw = [&](const char *name, double *out) {
        std::string k;
        if (!std::getline(in, k) || k != name) throw std::runtime_error(std::string("Could not find ") + name + " parameter in sys_params.txt");
        std::string v;
        std::getline(in, v);
        *out = std::stod(v);
    };
    double ne, nf, no;
    kw("n_elec", &ne); kw("n_frozen", &nf); kw("n_orb", &no); kw("eps", &r.eps); kw("hf_energy", &r.hf_en);
    r.n_frz = (unsigned)nf;
    if (r.n_frz != 0) throw std::runtime_error("legacy HF directory with frozen orbitals (n_frozen > 0) is not supported: freeze them when writing the integrals");
    Fcidump &f = r.mol;
    f.n_orb = (uint32_t)no; f.n_elec = (uint32_t)ne; f.core_en = 0;
    const size_t n = f.n_orb, np = n * (n + 1) / 2;
    std::vector<double> sy = read_csv_doubles(dir + "symm.txt");
    if (sy.size() < n) throw std::runtime_error("Could not read the irreps of all orbitals from symm.txt");
    f.symm.resize(n);
    for (size_t i = 0; i < n; i++) f.symm[i] = (uint8_t)sy[i];
    f.hcore = read_csv_doubles(dir + "hcore.txt");
    if (f.hcore.size() < n * n) { std::stringstream m; m << "Could not read " << n * n << " elements from " << dir << "hcore.txt"; throw std::runtime_error(m.str()); }
    f.hcore.resize(n * n);
    std::vector<double> e4 = read_csv_doubles(dir + "eris.txt");
    if (e4.size() < n * n * n * n) { std::stringstream m; m << "Could not read " << n * n * n * n << " elements from " << dir << "eris.txt"; throw std::runtime_error(m.str()); }
    f.eris.assign(np * (np + 1) / 2, 0.0);
    // (pq|rs) = <pr|qs>; the packed array keeps one representative of the 8 equivalent index orders (SymmERIs, ndarr.hpp:206-244)
    for (size_t q = 0; q < n; q++) for (size_t p_ = 0; p_ <= q; p_++) for (size_t s_ = 0; s_ < n; s_++) for (size_t r_ = 0; r_ <= s_; r_++) {
        const size_t p1 = tri(p_, q), p2 = tri(r_, s_);
        if (p1 <= p2) f.eris[tri(p1, p2)] = e4[((p_ * n + r_) * n + q) * n + s_];
    }
    return r;
}

static std::map<std::string, std::string> parse_kv(int argc, char **argv) {
    std::map<std::string, std::string> kv;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a.rfind("--", 0) != 0 || i + 1 >= argc) throw std::runtime_error("expected --option value pairs, got " + a);
        kv[a.substr(2)] = argv[++i];
    }
    return kv;
}
// digits of the doubles in the output files: 17 (round-trip exact; what the tests compare) unless --precision N is given; --precision 6 is the reference's
// own format (its drivers stream doubles with the default precision), for byte-wise comparison of result files
static inline int out_precision(std::map<std::string, std::string> &kv) { return kv.count("precision") ? std::stoi(kv["precision"]) : 17; }
static uint32_t wall_clock_seed() { return (uint32_t)std::chrono::high_resolution_clock::now().time_since_epoch().count(); }

// read_dets + load_vec_txt (FRIES/io_utils.cpp:447-482, 565-586): <prefix>dets holds one determinant per token as a signed 64-bit
// integer (byte k of the bit string = bits 8k..8k+7), <prefix>vals one value per token; the shorter file decides the length
static size_t load_vec_txt(const std::string &prefix, std::vector<uint64_t> &dets, std::vector<double> &vals) {
    std::ifstream file_d(prefix + "dets");
    if (!file_d.is_open()) throw std::runtime_error("Could not open file: " + prefix + "dets");
    dets.clear(); vals.clear();
    long long in_det;
    while (file_d >> in_det) dets.push_back((uint64_t)in_det);
    std::ifstream file_v(prefix + "vals");
    if (!file_v.is_open()) throw std::runtime_error("Could not open file: " + prefix + "vals");
    double v;
    while (file_v >> v) vals.push_back(v);
    if (vals.size() > dets.size()) { std::cerr << "Warning: fewer determinants (" << dets.size() << ") than values (" << vals.size() << ") read in\n"; vals.resize(dets.size()); }
    else if (vals.size() < dets.size()) { std::cerr << "Warning: fewer values (" << vals.size() << ") than determinants (" << dets.size() << ") read in\n"; dets.resize(vals.size()); }
    return vals.size();
}
