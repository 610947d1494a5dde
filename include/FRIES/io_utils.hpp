/*! \file  FRIES/io_utils.hpp for the MI355X build: the readers and writers the drivers call, same names and file formats as the
 * reference (FRIES/io_utils.cpp): parse_fcidump + convert_symm (:189-318), load_vec_txt / read_dets (:447-482, 565-586),
 * save_proc_hash / load_proc_hash (:589-619), load_last_line (:636-663), read_csv (:11-96).  parse_fcidump also remembers the molecule
 * so that the solution vector can hand it to the device when it is bound (FRIES/backend.hpp). */
#ifndef io_utils_h
#define io_utils_h
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <mpi.h>
#include <FRIES/ndarr.hpp>
#include <FRIES/backend.hpp>

/* the Hubbard-Holstein parameter file (io_utils.hpp:44-53, io_utils.cpp:320-405): keyword lines n_elec, lat_len, n_dim, eps, U, omega, g,
 * gs_energy, each followed by a line with its value, in that order */
struct hh_input {
    unsigned int n_elec;    ///< Total number of electrons in the system
    unsigned int lat_len;   ///< Number of sites along one dimension of the lattice
    unsigned int n_dim;     ///< Dimensionality of the lattice
    double elec_int;        ///< On-site repulsion term
    double eps;             ///< Suggested imaginary time step
    double hf_en;           ///< HF electronic energy
    double elec_ph;         ///< Electron-phonon coupling
    double ph_freq;         ///< Phonon energy
};

struct fcidump_input {
    uint32_t n_elec;            ///< Total number of electrons in the system
    uint32_t n_orb_;            ///< Number of spatial orbitals in the HF basis
    double core_en;             ///< Core energy to add to diagonal elements of H
    uint8_t *symm;              ///< Irreps of orbitals in the HF basis
    Matrix<double> *hcore;      ///< 1-electron integrals
    SymmERIs eris;              ///< 2-electron integrals
    fcidump_input(uint32_t n_orb) : n_orb_(n_orb), eris(n_orb) { symm = (uint8_t *)malloc(sizeof(uint8_t) * n_orb); hcore = nullptr; core_en = 0; n_elec = 0; }
    ~fcidump_input() { free(symm); }
};

template <class T> inline size_t fries_read_csv_line(std::ifstream &file, T *data) {
    size_t n_read = 0;
    std::string line;
    if (std::getline(file, line)) {
        std::stringstream ss_line(line);
        while (ss_line.good()) {
            std::string substr;
            std::getline(ss_line, substr, ',');
            std::stringstream number(substr);
            if (sizeof(T) == 1) { uint16_t inp = 0; number >> inp; data[n_read] = (T)inp; }
            else number >> data[n_read];
            n_read++;
        }
    }
    return n_read;
}
template <class T> inline size_t fries_read_csv(T *data, const std::string &fname) {
    std::ifstream in_f(fname);
    size_t n_read = 0, line_n_read = 1;
    while (line_n_read) { line_n_read = fries_read_csv_line(in_f, &data[n_read]); n_read += line_n_read; }
    return n_read;
}
inline size_t read_csv(double *data, const std::string &fname) { return fries_read_csv(data, fname); }
inline size_t read_csv(uint8_t *data, const std::string &fname) { return fries_read_csv(data, fname); }
inline size_t read_csv(int *data, const std::string &fname) { return fries_read_csv(data, fname); }

inline void convert_symm(uint8_t *irreps, size_t n_irreps, const std::string &point_group) {
    std::string pg = point_group;
    for (auto &ch : pg) ch = (char)tolower(ch);
    std::vector<uint8_t> map;
    unsigned max_label;
    if (pg == "d2h") { map = {0, 7, 6, 1, 5, 2, 3, 4}; max_label = 8; }
    else if (pg == "c2v" || pg == "c2h") { map = {0, 2, 3, 1}; max_label = 4; }
    else if (pg == "d2") { map = {0, 3, 2, 1}; max_label = 4; }
    else if (pg == "cs" || pg == "c2" || pg == "ci" || pg == "c1") { map = {0, 1}; max_label = 2; }
    else throw std::runtime_error("Point group " + point_group + " not recognized");
    for (size_t i = 0; i < n_irreps; i++) {
        if (irreps[i] > max_label || irreps[i] == 0) {
            std::stringstream msg;
            msg << "irrep index " << (unsigned)irreps[i] << " read from the FCIDUMP file exceeds the maximum allowed irrep index (" << max_label << ") for point group " << point_group;
            throw std::runtime_error(msg.str());
        }
        irreps[i] = map[irreps[i] - 1];
    }
}

inline fcidump_input *parse_fcidump(const std::string &fcidump_path, const std::string &point_group) {
    std::ifstream in(fcidump_path);
    if (!in.is_open()) throw std::runtime_error("Could not open FCIDUMP file " + fcidump_path);
    std::string line;
    std::getline(in, line);
    auto field = [&](const char *key) {
        size_t p = line.find(key);
        if (p == std::string::npos) throw std::runtime_error(std::string("FCIDUMP header lacks ") + key);
        size_t e = line.find(",", p);
        return std::stoi(line.substr(p + strlen(key), e - (p + strlen(key))));
    };
    const uint32_t n_orb = (uint32_t)field("NORB=");
    fcidump_input *f = new fcidump_input(n_orb);
    f->n_elec = (uint32_t)field("NELEC=");
    if (field("MS2=") != 0) throw std::runtime_error("MS2 is not zero in FCIDUMP file.");
    std::getline(in, line);
    size_t op = line.find("ORBSYM=");
    if (op == std::string::npos) throw std::runtime_error("ORBSYM missing on line 2 of the FCIDUMP file");
    std::stringstream ss(line.substr(op + 7));
    std::string tok;
    size_t ns = 0;
    while (std::getline(ss, tok, ',')) { try { if (!tok.empty() && ns < n_orb) f->symm[ns++] = (uint8_t)std::stoi(tok); } catch (std::invalid_argument &) {} }
    if (ns != n_orb) throw std::runtime_error("Number of irrep labels read in after ORBSYM in FCIDUMP file does not equal number of orbitals");
    convert_symm(f->symm, n_orb, point_group);
    std::getline(in, line);     // ISYM
    std::getline(in, line);     // &END
    f->hcore = new Matrix<double>(n_orb, n_orb);
    f->hcore->zero();
    double v; unsigned o[4];
    while (in >> v >> o[0] >> o[1] >> o[2] >> o[3]) {
        if (!o[0] && !o[1] && !o[2] && !o[3]) f->core_en = v;
        else if (!o[1] && !o[2] && !o[3]) continue;                  // orbital energy
        else if (!o[2] && !o[3]) { (*f->hcore)(o[0] - 1, o[1] - 1) = v; (*f->hcore)(o[1] - 1, o[0] - 1) = v; }
        else {
            size_t a = o[0] - 1, b = o[1] - 1, c = o[2] - 1, d = o[3] - 1;
            size_t p1 = a <= b ? I_J_TO_TRI_WDIAG(a, b) : I_J_TO_TRI_WDIAG(b, a), p2 = c <= d ? I_J_TO_TRI_WDIAG(c, d) : I_J_TO_TRI_WDIAG(d, c);
            ((double *)f->eris.packed())[p1 <= p2 ? I_J_TO_TRI_WDIAG(p1, p2) : I_J_TO_TRI_WDIAG(p2, p1)] = v;
        }
    }
    // MI355X build: the device gets these integrals when the solution vector is bound to it
    fries_hip::Backend &B = fries_hip::Backend::get();
    B.n_orb = n_orb; B.n_elec = f->n_elec;
    B.symm.assign(f->symm, f->symm + n_orb);
    B.hcore.assign(f->hcore->data(), f->hcore->data() + (size_t)n_orb * n_orb);
    const size_t np = (size_t)n_orb * (n_orb + 1) / 2;
    B.eris.assign(f->eris.packed(), f->eris.packed() + np * (np + 1) / 2);
    B.eris_obj = &f->eris; B.hcore_obj = f->hcore;
    B.have_mol = true;
    return f;
}

inline size_t read_dets(const std::string &path, Matrix<uint8_t> &dets) {
    std::ifstream file_d(path);
    if (!file_d.is_open()) throw std::runtime_error("Could not open file: " + path);
    size_t n_dets = 0;
    long long in_det;
    size_t max_size = dets.cols();
    while (file_d >> in_det) {
        for (size_t byte_idx = 0; byte_idx < 8 && byte_idx < max_size; byte_idx++) { dets(n_dets, byte_idx) = in_det & 255; in_det >>= 8; }
        n_dets++;
    }
    return n_dets;
}
template <class T> inline size_t fries_load_vec_txt(const std::string &prefix, Matrix<uint8_t> &dets, T *vals) {
    int my_rank = 0;
    MPI_Comm_rank(MPI_COMM_WORLD, &my_rank);
    if (my_rank != 0) return 0;
    size_t n_dets = read_dets(prefix + "dets", dets);
    std::ifstream file_v(prefix + "vals");
    if (!file_v.is_open()) throw std::runtime_error("Could not open file: " + prefix + "vals");
    size_t n_vals = 0;
    double in_val;
    while (file_v >> in_val) vals[n_vals++] = (T)in_val;
    if (n_vals > n_dets) { std::cerr << "Warning: fewer determinants than values read in\n"; return n_dets; }
    else if (n_vals < n_dets) { std::cerr << "Warning: fewer values than determinants read in\n"; return n_vals; }
    return n_vals;
}
inline size_t load_vec_txt(const std::string &prefix, Matrix<uint8_t> &dets, int *vals) { return fries_load_vec_txt(prefix, dets, vals); }
inline size_t load_vec_txt(const std::string &prefix, Matrix<uint8_t> &dets, double *vals) { return fries_load_vec_txt(prefix, dets, vals); }

inline void save_proc_hash(const std::string &path, unsigned int *proc_hash, size_t n_hash) {
    std::ofstream f(path + "hash.dat", std::ios::binary);
    if (!f.is_open()) throw std::runtime_error("Error: could not open file for saving hash scrambler at " + path + "hash.dat");
    f.write((const char *)proc_hash, (std::streamsize)(sizeof(unsigned int) * n_hash));
}
inline void load_proc_hash(const std::string &path, unsigned int *proc_hash) {
    std::ifstream f(path + "hash.dat", std::ios::binary);
    if (!f.is_open()) throw std::runtime_error("Error: could not open saved hash scrambler at " + path + "hash.dat");
    f.read((char *)proc_hash, 1000);       // the reference reads a fixed 1000 bytes (io_utils.cpp:617); what the file holds is what counts
}
inline size_t load_last_line(const std::string &path, double *vals) {
    std::ifstream in_f(path);
    size_t n_read = 0;
    if (in_f.is_open()) {
        std::string line, last;
        while (std::getline(in_f, line)) if (!line.empty()) last = line;
        std::stringstream ss(last);
        std::string tok;
        while (std::getline(ss, tok, ',')) { std::stringstream num(tok); if (num >> vals[n_read]) n_read++; }
    }
    return n_read;
}
inline void parse_hh_input(const std::string &hh_path, hh_input *in_struct) {
    std::ifstream in(hh_path);
    if (!in.is_open()) throw std::runtime_error("Could not open file containing Hubbard-Holstein parameters");
    std::string line;
    auto keyword = [&](const char *key, const char *what) {
        if (!std::getline(in, line) || line != key) throw std::runtime_error(std::string("Could not find ") + what + " in file containing Hubbard-Holstein parameters");
    };
    auto value_of = [&](const char *key, const char *what) { keyword(key, what); double v = 0; in >> v; std::getline(in, line); return v; };
    in_struct->n_elec = (unsigned int)value_of("n_elec", "n_elec parameter");
    in_struct->lat_len = (unsigned int)value_of("lat_len", "lat_len parameter");
    in_struct->n_dim = (unsigned int)value_of("n_dim", "n_dim parameter");
    in_struct->eps = value_of("eps", "eps parameter");
    in_struct->elec_int = value_of("U", "electron interaction parameter (U)");
    in_struct->ph_freq = value_of("omega", "phonon frequency parameter (omega)");
    in_struct->elec_ph = value_of("g", "electron-phonon interaction parameter (g)");
    in_struct->hf_en = value_of("gs_energy", "gs_energy parameter");
    // MI355X build: the device context of the Hubbard-Holstein vector is set up from these (HubHolVec::device_setup)
    fries_hip::HHParams &P = fries_hip::hh_params();
    P.n_elec = in_struct->n_elec; P.lat_len = in_struct->lat_len; P.eps = in_struct->eps; P.U = in_struct->elec_int; P.omega = in_struct->ph_freq;
    P.g = in_struct->elec_ph; P.gs_energy = in_struct->hf_en; P.set = true;
    fries_hip::Backend::get().hh_mode = true;
}
#endif /* io_utils_h */
