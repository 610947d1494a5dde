// TEST INFRASTRUCTURE ONLY -- CPU oracle for the FRI hot path.
//
// This is a sequential CPU restatement of the reference algorithm
// (sgreene8/FRIES @ 2025-02-15).  It exists so that tests/, bench.py's
// cpu_baseline leg and __graft_entry__.smoke() can check the HIP path.  Nothing
// under fries_amd/ (the product) may include, link or call it.
//
// Conventions that differ from the reference's *representation* (never from
// its arithmetic):
//   * a determinant is one uint64_t; bit i == spin orbital i (alpha = 0..n_orb-1,
//     beta = n_orb..2n_orb-1).  This equals the reference's little-endian byte
//     string (FRIES/det_store.h:23-26) for 2*n_orb <= 64.
//   * n_frz == 0 everywhere (FRIES_bin/frisys_mol.cpp:79 hard-codes it).
//   * one MPI rank (n_procs == 1) unless a function says otherwise; sum_mpi of
//     one rank is the identity (FRIES/compress_utils.hpp:179-231).
//
// Parity status: PINNED.  oracle/ref_harness.cpp links the real reference
// (built from /root/reference by oracle/Makefile into oracle/_ref/) and checks
// these functions bit-for-bit; tests/golden/ holds the vectors it emitted.
#pragma once
#include <cstdint>
#include <cstddef>
#include <vector>
#include <stack>
#include <unordered_map>
#include <random>
#include <string>
#include <functional>

namespace fo {

typedef uint64_t det_t;

// ---------------------------------------------------------------- ranks
// The reference's only parallelism is hash-sharding over MPI ranks with blocking collectives on
// MPI_COMM_WORLD.  Comm emulates exactly those collectives between in-process ranks
// (one std::thread per rank, see run_ranks); size == 1 needs no threads.
struct CommShared;
struct Comm {
    int rank = 0, size = 1;
    CommShared *sh = nullptr;
    void barrier() const;
    void allgather(const void *in, void *out, size_t bytes) const;              // MPI_Allgather, rank order
    // MPI_Alltoallv of byte strings: send[d] goes to rank d; recv[s] is what rank s sent here
    void alltoallv(const std::vector<std::vector<uint8_t>> &send, std::vector<std::vector<uint8_t>> &recv) const;
    double sum(double x) const;          // sum_mpi: all-gather, then add in rank order (compress_utils.hpp:177-187)
    int sum(int x) const;
    static const Comm &self();
};
CommShared *comm_create(int size);
void comm_destroy(CommShared *);

// ---------------------------------------------------------------- bit strings
// FRIES/math_utils.c:62-98 (find_bits): ascending list of set bits.
int occ_list(det_t det, uint8_t *occ);
// FRIES/math_utils.c:9-58 (bits_between): set bits strictly between a and b.
unsigned bits_between(det_t det, unsigned a, unsigned b);
// FRIES/fci_utils.c:130-136
int excite_sign(unsigned cre, unsigned des, det_t det);
// FRIES/fci_utils.c:46-58 / 60-64
int sing_det_parity(det_t *det, const uint8_t *orbs);
int sing_parity(det_t det, const uint8_t *orbs);
// FRIES/fci_utils.c:66-96
int doub_det_parity(det_t *det, const uint8_t *orbs);
int doub_parity(det_t det, const uint8_t *orbs);
det_t sing_det(det_t det, const uint8_t *orbs);
det_t doub_det(det_t det, const uint8_t *orbs);
// FRIES/fci_utils.c:9-43
det_t gen_hf_det(unsigned n_orb, unsigned n_elec);
// FRIES/fci_utils.c:138-148
uint8_t find_nth_virt(const uint8_t *occ, int spin, unsigned n_elec, unsigned n_orb, unsigned n);

// ---------------------------------------------------------------- integrals
struct Integrals {
    unsigned n_orb = 0;
    std::vector<double> h;    // n_orb x n_orb, row major (FRIES/io_utils.cpp:307)
    std::vector<double> eri;  // 8-fold packed, FRIES/ndarr.hpp:206-244
    double chem(unsigned i, unsigned j, unsigned k, unsigned l) const;
    double phys(unsigned i, unsigned j, unsigned k, unsigned l) const { return chem(i, k, j, l); }
    static size_t packed_len(unsigned n) { size_t p = (size_t)n * (n + 1) / 2; return p * (p + 1) / 2; }
};

static const unsigned N_IRREPS = 8;  // FRIES/Hamiltonians/near_uniform.hpp (n_irreps)

// FRIES/Hamiltonians/molecule.hpp:265-280, molecule.cpp:1050-1065
struct Symm {
    unsigned n_orb = 0;
    std::vector<uint8_t> irrep;              // per spatial orbital
    std::vector<uint8_t> lookup;             // N_IRREPS x (n_orb + 1); col 0 = count
    unsigned max_n_symm = 0;
    void init(const uint8_t *irreps, unsigned n);
    uint8_t lk(unsigned ir, unsigned col) const { return lookup[ir * (n_orb + 1) + col]; }
};

// FRIES/Hamiltonians/molecule.cpp:983-1029 / 76-105 / 26-42
double diag_matrel(const uint8_t *occ, const Integrals &in, unsigned n_elec);
double sing_matrel_nosgn(const uint8_t *ex, const uint8_t *occ, const Integrals &in, unsigned n_elec);
double doub_matrel_nosgn(const uint8_t *ex, const Integrals &in);
// FRIES/Hamiltonians/molecule.cpp:178-203 / 108-175 / 914-933
size_t sing_ex_symm(det_t det, const uint8_t *occ, unsigned n_elec, unsigned n_orb, std::vector<uint8_t> &out, const uint8_t *irrep);
size_t doub_ex_symm(det_t det, const uint8_t *occ, unsigned n_elec, unsigned n_orb, std::vector<uint8_t> &out, const uint8_t *irrep);
size_t count_singex(det_t det, const uint8_t *occ, unsigned n_elec, const Symm &s);

// FRIES/Hamiltonians/near_uniform.cpp:14-28 / 316-327 / 330-347 / 419-433
void count_symm_virt(unsigned counts[][2], const uint8_t *occ, unsigned n_elec, const Symm &s);
unsigned count_sing_allowed(const uint8_t *occ, unsigned n_elec, const Symm &s, unsigned counts[][2]);
unsigned count_sing_virt(const uint8_t *occ, unsigned n_elec, const Symm &s, unsigned counts[][2], uint8_t *occ_choice);
uint8_t virt_from_idx(det_t det, const Symm &s, unsigned irrep, unsigned spin_shift, unsigned index);

// ---------------------------------------------------------------- HB-PP
// FRIES/Hamiltonians/heat_bathPP.hpp:25-34, heat_bathPP.cpp:99-179
struct HBInfo {
    unsigned n_orb = 0;
    std::vector<double> s_tens, d_same, d_diff, exch_sqrt, diag_sqrt, exch_norms;
    double s_norm = 0;
    void set_up(const Integrals &in);
};
double calc_o1_probs(const HBInfo &t, double *p, unsigned n_elec, const uint8_t *occ, int exclude_first);
double calc_o2_probs(const HBInfo &t, double *p, unsigned n_elec, const uint8_t *occ, unsigned o1_idx);
double calc_o2_probs_half(const HBInfo &t, double *p, unsigned n_elec, const uint8_t *occ, unsigned o1_idx);
double calc_u1_probs(const HBInfo &t, double *p, unsigned o1_orb, const uint8_t *occ, unsigned n_elec, int exclude_first);
double calc_u2_probs(const HBInfo &t, double *p, unsigned o1, unsigned o2, unsigned u1, const Symm &s, uint16_t *len);
double calc_u2_probs_half(const HBInfo &t, double *p, unsigned o1, unsigned o2, unsigned u1, det_t det, const Symm &s, uint16_t *len);
double calc_unnorm_wt(const HBInfo &t, const uint8_t *orbs);
double calc_norm_wt(const HBInfo &t, const uint8_t *orbs, const uint8_t *occ, unsigned n_elec, det_t det, const Symm &s);

// ---------------------------------------------------------------- compression
// FRIES/compress_utils.cpp:29-105
double find_preserve(const double *values, std::vector<size_t> &srt, std::vector<uint8_t> &keep,
                     size_t count, unsigned *n_samp, double *global_norm, const Comm &cm = Comm::self());
// FRIES/compress_utils.cpp:107-127
double seed_sys(const double *norms, double *rn, unsigned n_samp, const Comm &cm = Comm::self());
// FRIES/compress_utils.cpp:283-327; loc_norms[size]: in = every rank's remaining norm, out = every rank's new norm
void sys_comp(double *vals, size_t len, double *loc_norms, unsigned n_samp, std::vector<uint8_t> &keep, double rn, const Comm &cm = Comm::self());

// Sub-weight matrix + keep bits for comp_sub.  keep bit (row, col) lives in a
// uint32 mask per row; the reference's byte-granular thresholds
// (compress_utils.cpp:213 vs :233) are reproduced from the column index.
struct SubWts {
    size_t cols = 0;
    std::vector<double> w;        // rows x cols
    std::vector<uint32_t> keep;   // rows
    void reshape(size_t rows, size_t c) { cols = c; if (w.size() < rows * c) w.resize(rows * c); if (keep.size() < rows) keep.resize(rows, 0); }
    double *row(size_t r) { return &w[r * cols]; }
    const double *row(size_t r) const { return &w[r * cols]; }
};
// FRIES/compress_utils.cpp:130-276
double find_keep_sub(const double *values, const uint32_t *n_div, SubWts &sw, const uint16_t *sub_sizes,
                     size_t count, unsigned *n_samp, double *wt_remain, const Comm &cm = Comm::self());
// FRIES/compress_utils.cpp:702-794
size_t sys_sub(const double *values, const uint32_t *n_div, SubWts &sw, const uint16_t *sub_sizes,
               size_t count, unsigned n_samp, const double *wt_remain, double *loc_norms, double rn,
               double *new_vals, size_t (*new_idx)[2], const Comm &cm = Comm::self());
// FRIES/compress_utils.cpp:797-820
size_t comp_sub(const double *values, size_t count, const uint32_t *n_div, SubWts &sw, const uint16_t *sub_sizes,
                unsigned n_samp, double *wt_remain, double rn, double *new_vals, size_t (*new_idx)[2], const Comm &cm = Comm::self());
// ---- pivotal compression (FRIES/compress_utils.cpp:354-681)
// :389-518.  flag[] in: elements preserved exactly; out: elements that ended up zero (to be deleted).
void piv_samp_serial(double *vals, size_t len, double seg_norm, uint32_t n_samp, std::vector<uint8_t> &flag, std::mt19937 &mt);
// :552-604.  Rank 0 apportions the samples among the ranks (it alone draws from its generator).
uint32_t piv_budget(const double *loc_norms, uint32_t n_samp, std::mt19937 &mt, const Comm &cm = Comm::self());
// :606-681
double adjust_probs(double *vals, size_t len, uint32_t *n_samp_loc, double exp_nsamp_loc, uint32_t n_samp_tot, double tot_norm, std::vector<uint8_t> &flag);
// :354-386
void piv_comp_parallel(double *vals, size_t len, uint32_t compress_size, std::vector<size_t> &srt, std::vector<uint8_t> &flag, std::mt19937 &mt,
                       const Comm &cm = Comm::self());
// FRIES/compress_utils.cpp:684-693
void adjust_shift(double *shift, double one_norm, double *last_norm, double target_norm, double damp);

// ---------------------------------------------------------------- sparse vector
// FRIES/det_hash.hpp:160-170 (hash_fxn, no phonons)
uint64_t hash_fxn(const uint8_t *occ, unsigned n_elec, const uint32_t *scrambler);

// Restatement of DistVec<double> + Adder<double> for one rank
// (FRIES/vec_utils.hpp:121-953, 957-1019).
struct Vec {
    unsigned n_elec = 0, n_vecs = 0;
    size_t max_size = 0, curr_size = 0, adder_cap = 0;
    int n_nonz = 0;
    unsigned cur = 0;
    uint64_t nonini_occ_add = 0;
    std::vector<det_t> dets;
    std::vector<std::vector<double>> vals;   // n_vecs columns
    std::vector<uint8_t> occ;                // max_size x n_elec
    std::vector<double> diag;                // NaN = not cached
    std::vector<uint8_t> active;
    std::vector<size_t> free_stack;          // back() == top
    std::unordered_map<det_t, ptrdiff_t> table;
    // pending adds, one buffer per destination rank (Adder, vec_utils.hpp:98-106)
    std::vector<std::vector<det_t>> add_det; std::vector<std::vector<double>> add_val; std::vector<std::vector<uint8_t>> add_ini;
    Comm cm;                                 // ranks this vector is sharded over
    const uint32_t *proc_scr = nullptr;      // proc_hash_ scrambler (vec_utils.hpp:137)
    // Hubbard-Holstein indices (HubHolVec, hh_vec.hpp:10-262): electrons live in the low 2 * hh_sites bits, hh_ph_bits per phonon above
    unsigned hh_sites = 0, hh_ph_bits = 0;
    const uint32_t *vec_scr = nullptr;       // vec_hash_ scrambler, only needed for key_of() below
    size_t n_buckets = 0;
    det_t elec_mask() const { return hh_sites ? ((det_t)1 << (2 * hh_sites)) - 1 : ~(det_t)0; }
    // What the reference's HashTable can tell apart (det_hash.hpp:47, :60-94): entries live in bucket hash % n_buckets and are
    // compared over ceil(scrambler.size() / 8) bytes.  For molecules that is the whole determinant.  For Hubbard-Holstein the
    // scrambler has 2 * n_sites entries, so only the first ceil(2 n_sites / 8) bytes are compared: two states with the same
    // electrons, different phonons and the same bucket ARE one entry there (the later one adds into the earlier one's slot).
    det_t key_of(det_t det) const;

    void init(size_t size, size_t add_size, unsigned n_el, unsigned nv, const Comm &c = Comm::self(), const uint32_t *pscr = nullptr);
    int idx_to_proc(det_t det) const;         // vec_utils.hpp:360-379
    void expand();
    bool add(det_t det, double val, uint8_t ini);     // vec_utils.hpp:418-423, 957-971
    void perform_add(size_t origin);                   // vec_utils.hpp:991-1019 -> add_elements :606-641
    void del_at_pos(size_t pos);                       // vec_utils.hpp:458-476
    double local_norm() const;                         // vec_utils.hpp:683-689
    const uint8_t *orbs_at(size_t pos) const { return &occ[pos * n_elec]; }
    double dot(const std::vector<det_t> &d2, const std::vector<double> &v2) const;  // vec_utils.hpp:228-238
    void zero_cur() { std::fill(vals[cur].begin(), vals[cur].end(), 0.0); }
    void add_vecs(unsigned i1, unsigned i2) { for (size_t i = 0; i < curr_size; i++) vals[i1][i] += vals[i2][i] * 1.0; }
};

// ---------------------------------------------------------------- HB-PP compress-multiply
// FRIES/Hamiltonians/heat_bathPP.hpp:250-297
struct HBScratch {
    size_t len = 0, vec_len = 0;
    std::vector<double> vec1, vec2, wt_remain;
    std::vector<size_t> det_idx1, det_idx2;
    std::vector<uint8_t> orb1, orb2;         // len x 4
    std::vector<uint16_t> nsub;
    std::vector<uint32_t> ndiv;
    std::vector<size_t> comp_idx;            // len x 2
    SubWts sw;
    size_t stage_len[5] = {0, 0, 0, 0, 0};    // elements after each of the five comp_sub calls of the last apply_HBPP_sys
    void init(size_t length, size_t n_subwt);
};
struct MolSys {
    unsigned n_orb = 0, n_elec = 0;
    Integrals ints;
    Symm symm;
    HBInfo hb;
    double hf_en = 0;
};
// FRIES/Hamiltonians/heat_bathPP.cpp:686-992.  rn[5] are the five uniforms the
// reference draws at :729,:765,:811,:859,:910.  unit_matrel selects the
// |value| == 1 lambdas of tests/test_hamiltonian.cpp:493-500.
void apply_HBPP_sys(const Vec &v, HBScratch &sc, const MolSys &sys, double p_doub, bool new_hb,
                    const double rn[5], uint32_t n_samp, bool unit_matrel, const Comm &cm = Comm::self());

// FRIES/Hamiltonians/heat_bathPP.hpp:303-311 (HBCompressPiv) and heat_bathPP.cpp:994-1419 (collapse_long_, apply_HBPP_piv) for
// spin_parity == 0: every factor of the HB-PP factorisation is multiplied out into long_vec, compressed by
// piv_comp_parallel (find_preserve + pivotal sampling, draws from mt) and collapsed back.  stage_len[k] = short length after stage k.
struct HBPivScratch {
    size_t len = 0, vec_len = 0;
    std::vector<double> vec1, long_vec;
    std::vector<size_t> det_idx1, det_idx2, srt;
    std::vector<uint8_t> orb1, orb2, flag;
    std::vector<uint16_t> group;
    size_t stage_len[5] = {0, 0, 0, 0, 0};
    void init(size_t length, size_t n_subwt);
};
void apply_HBPP_piv(const Vec &v, HBPivScratch &sc, const MolSys &sys, double p_doub, bool new_hb,
                    std::mt19937 &mt, uint32_t n_samp, bool unit_matrel, const Comm &cm = Comm::self(), int spin_parity = 0);

// ---------------------------------------------------------------- driver loop
struct FrisysParams {
    double eps = 0.01, target_norm = 0, init_thresh = 0;
    uint32_t vec_nonz = 0, mat_nonz = 0;
    size_t max_dets = 0;
    bool new_hb = true;
    uint32_t seed = 0;
};
struct IterLog {
    double numer, denom, shift, norm;
    uint32_t nkept;
    int n_nonz;
    size_t curr_size, num_success, comp_len[5];
};
// FRIES_bin/frisys_mol.cpp:35-566 with n_procs == 1, HF trial vector, HF start,
// no dense space, seed injected instead of the wall clock (:104-106).
struct Frisys {
    MolSys sys;
    FrisysParams par;
    std::mt19937 mt;
    std::vector<uint32_t> proc_scr, vec_scr;
    Vec sol;
    HBScratch sc;
    std::vector<det_t> trial_det, htrial_det;
    std::vector<double> trial_val, htrial_val;
    // optional inputs of the driver: --trial_vec, --ini_vec (text vectors, rank 0 reads and adds them: frisys_mol.cpp:157-181,
    // 264-274) and --ham_shift (:95-98: the diagonal offset replaces the HF energy; core energy already subtracted by the caller)
    std::vector<det_t> trial_in_det, ini_det;
    std::vector<double> trial_in_val, ini_val;
    bool has_ham_shift = false; double ham_shift = 0;
    // --det_space (semi-stochastic): the determinants of the file take positions 0 .. n_determ - 1 (DistVec::init_dense,
    // vec_utils.hpp:858-897), are never compressed or deleted, and H restricted to them is applied exactly every iteration
    // (frisys_mol.cpp:236-239, 347-401, 414-421, 480-485, 502-539)
    std::vector<det_t> det_space;
    size_t n_determ = 0;
    uint32_t tot_dense_h = 0;
    std::vector<size_t> determ_from; std::vector<det_t> determ_to; std::vector<double> determ_el;
    double p_doub = 0, en_shift = 0, last_one_norm = 0;
    det_t hf_det = 0;
    unsigned iterat = 0;
    std::vector<size_t> srt; std::vector<uint8_t> keep;
    std::vector<IterLog> log;
    Comm cm;
    int hf_proc = 0;
    void setup();
    void iterate(unsigned n);
};
// ---------------------------------------------------------------- deterministic H application + frifull_mol
// FRIES/Hamiltonians/molecule.cpp:448-665 without time-reversal symmetry (spin_parity 0): every symmetry-allowed single
// excitation of every stored determinant, then every double, each added to column dest as value * h_fac * <j|H|i>.
// Returns the number of add() calls.
size_t h_op_offdiag(Vec &v, size_t vec_size, const MolSys &sys, unsigned dest, double h_fac, int spin_parity = 0);
// Time-reversal symmetry (spin_parity = +-1: the vector holds one representative of every pair {determinant, its spin-flipped image}).
// FRIES/fci_utils.c:158-204 (flip_spins: alpha and beta strings trade places), :310-359 (tr_doub_connect), and the adjust_tr lambda of
// h_op_offdiag (FRIES/Hamiltonians/molecule.cpp:298-369, 472-552): a matrix element <new|H|cur> becomes the element between the
// symmetrised functions -- the image's contribution added with the parity, norms of self-paired functions, and the representative (the
// byte-wise smaller of new and its image) as the target.  Returns 0: no contribution; 1: add to *target.
det_t flip_spins(det_t det, unsigned n_orb);
int det_memcmp(det_t a, det_t b);            // memcmp over the little-endian byte string
int tr_doub_connect(const uint8_t *occ, unsigned n_orb, unsigned n_elec, uint8_t *diff_idx);
// weight_fix (optional): called with (kind 1 single / 2 double, reordered diff orbitals) when the image contributes (apply_HBPP_piv adds
// the image's selection probability to the weight there, heat_bathPP.cpp:1363, 1395)
int adjust_tr(const MolSys &sys, det_t cur, det_t nd, const uint8_t *occ, double *matr_el, int spin_parity, det_t *target, bool unit_matrel = false,
              const std::function<void(int, const uint8_t *)> &weight_fix = nullptr);
// molecule.cpp:205-219
void h_op_diag(Vec &v, unsigned dest, double id_fac, double h_fac, const MolSys &sys);
struct FrifullParams { double eps = 0.01, target_norm = 0; uint32_t vec_nonz = 0; size_t max_dets = 0; uint32_t seed = 0; };
// FRIES_bin/frifull_mol.cpp:27-336 with one rank, HF trial vector, HF start, seed injected instead of the wall clock (:63-65)
struct Frifull {
    MolSys sys;
    FrifullParams par;
    std::mt19937 mt;
    std::vector<uint32_t> proc_scr, vec_scr;
    Vec sol;
    std::vector<det_t> trial_det;
    std::vector<double> trial_val;
    double en_shift = 0, last_one_norm = 0;
    det_t hf_det = 0;
    unsigned iterat = 0, vec_idx = 0;
    std::vector<size_t> srt; std::vector<uint8_t> keep;
    std::vector<IterLog> log;       // num_success = add() calls of the off-diagonal application
    void setup();
    void iterate(unsigned n);
};

// ---------------------------------------------------------------- Hubbard-Holstein (frisys_hh)
// bit string = [alpha sites | beta sites | ph_bits per site] (hh_vec.hpp:22, hub_holstein.cpp:139-171)
struct HHParams {
    unsigned n_elec = 0, n_sites = 0, ph_bits = 3;      // ph_bits: frisys_hh.cpp:96
    double eps = 0, U = 0, omega = 0, g = 0, hf_en = 0; // parse_hh_input, io_utils.cpp:320-405
    double target_norm = 0, init_thresh = 0;
    uint32_t vec_nonz = 0; size_t max_dets = 0;
    uint32_t seed = 0;
};
// FRIES/Hamiltonians/hub_holstein.cpp:101-136 / 139-171; FRIES/hh_vec.hpp:139-175 / 185-197 / 207-233
unsigned hub_diag(det_t det, unsigned n_sites);
det_t gen_neel_det_1D(unsigned n_sites, unsigned n_elec);
void find_neighbors_1D(det_t det, unsigned n_sites, unsigned n_elec, uint8_t *neighbors /* 2 (n_elec + 1) */);
void decode_phonons(det_t det, unsigned n_sites, unsigned ph_bits, uint8_t *numbers);
bool det_from_ph(det_t det, det_t *out, unsigned n_sites, unsigned ph_bits, unsigned site, int change);
// FRIES/Hamiltonians/hub_holstein.hpp:93-186
double calc_ref_ovlp(const det_t *dets, const double *vals, size_t n, det_t ref, unsigned n_elec, unsigned n_sites, unsigned ph_bits, double g_over_t);
uint64_t hash_fxn_hh(const uint8_t *occ, unsigned n_elec, const uint8_t *ph, unsigned n_sites, const uint32_t *scr);   // det_hash.hpp:160-170

struct HHLog { double numer, denom, shift, norm; uint32_t nkept; int n_nonz; size_t curr_size, num_success; };
// FRIES_bin/frisys_hh.cpp:27-380, seed injected instead of the wall clock (:66-68)
struct FrisysHH {
    HHParams par;
    bool full = false;          // frifull_hh (FRIES_bin/frifull_hh.cpp): H applied in full instead of the two compressions
    std::mt19937 mt;
    std::vector<uint32_t> proc_scr, vec_scr;
    Vec sol;
    det_t neel = 0;
    int ref_proc = 0;
    double en_shift = 0, last_one_norm = 0;
    unsigned iterat = 0;
    std::vector<double> comp1, comp2, wt_remain;
    std::vector<uint32_t> ndiv;
    std::vector<size_t> comp_idx, det_indices;
    std::vector<uint8_t> ph_ex;
    SubWts sw;
    std::vector<size_t> srt; std::vector<uint8_t> keep;
    std::vector<HHLog> log;
    Comm cm;
    void setup();
    void iterate(unsigned n);
};

// ---------------------------------------------------------------- FCIQMC (fciqmc_mol, near-uniform excitation generator)
// The reference draws every uniform from ONE sequential mt19937 stream, and how many draws a determinant consumes depends
// on the values drawn (rejection loops): a parallel sampler cannot replay that stream.  Rng therefore has two modes.
//   mt mode      -- the reference's stream: used to pin every function and the whole loop against the reference.
//   counter mode -- uniform = hash(seed, iteration, determinant, attempt, purpose, n-th draw of that attempt): the draws
//                   of different determinants / attempts are independent of each other, so the GPU can reproduce the
//                   oracle bit for bit.  Same distribution (i.i.d. uniforms), different stream.
struct Rng {
    std::mt19937 *mt = nullptr;
    uint64_t seed = 0, key = 0;
    uint32_t ctr = 0;
    void begin(uint64_t iter, det_t det, uint32_t attempt, uint32_t purpose);
    double uni();
    static uint64_t mix(uint64_t x);
};
enum { RNG_BIN = 0, RNG_DOUB = 1, RNG_SING = 2, RNG_ROUND_D = 3, RNG_ROUND_S = 4, RNG_DEATH = 5, RNG_HB_O1 = 6, RNG_HB_O2 = 7, RNG_HB_U1 = 8, RNG_HB_U2 = 9, RNG_NWALK = 10, RNG_COMP = 11 };
// FRIES/compress_utils.cpp:823-856 / 858-877 (one sample)
void setup_alias(const double *probs, unsigned *aliases, double *alias_probs, size_t n_states);
unsigned sample_alias_one(const unsigned *aliases, const double *alias_probs, size_t n_states, Rng &rng);
// FRIES/Hamiltonians/heat_bathPP.cpp:601-683: num_sampl heat-bath double excitations of one determinant.  Samples come out grouped
// by their first occupied orbital.  att[k] = (electron index of o1) << 20 | index inside that group: what counter mode keys the
// later draws of sample k by.  iter: iteration number for the counter keys.
unsigned hb_doub_multi(det_t det, const uint8_t *occ, unsigned n_elec, const Symm &s, const HBInfo &hb, unsigned num_sampl, Rng &rng, uint64_t iter,
                       uint8_t *orbs /* 4 per sample */, double *prob, uint32_t *att);
// FRIES/Hamiltonians/near_uniform.cpp:31-39, compress_utils.cpp:19-27
unsigned bin_sample(unsigned n, double p, Rng &rng);
int round_binomially(double p, unsigned n, Rng &rng);
// one sample of doub_multin / sing_multin (near_uniform.cpp:193-245, 277-313); false = null excitation
bool nu_doub_sample(det_t det, const uint8_t *occ, unsigned n_elec, const Symm &s, unsigned counts[][2], Rng &rng, uint8_t orbs[4], double *prob);
void nu_sing_setup(const uint8_t *occ, unsigned n_elec, const Symm &s, unsigned counts[][2], unsigned *m_allow, unsigned *delta_s);
void nu_sing_sample(det_t det, const uint8_t *occ, unsigned n_elec, const Symm &s, const unsigned *m_allow, unsigned delta_s, Rng &rng, uint8_t orbs[2], double *prob);

struct FciqmcParams {
    double eps = 0.001;
    uint32_t target_walkers = 0, init_thresh = 0;
    size_t max_dets = 0;
    uint32_t seed = 0;
    bool counter_rng = false;
    bool heat_bath = false;         // --distribution HB (hb_doub_multi for the doubles) instead of NU
    // frimulti_mol (FRIES_bin/frimulti_mol.cpp): FRI with multinomial matrix compression -- real-valued vector, the number of samples
    // per column from one systematic comb over |v|, the vector compressed to vec_nonz by find_preserve + sys_comp
    // fciqmc_fp_mol (FRIES_bin/fciqmc_fp_mol.cpp): real-valued walkers -- |v| rounded stochastically to the number of spawning attempts,
    // spawns below 0.01 rounded to integers and kept real otherwise, death in place, then every |v| < 1 rounded to -1 / 0 / 1
    bool fp = false;
    bool multi = false;
    uint32_t vec_nonz = 0, mat_nonz = 0;
    double target_norm = 0, init_thresh_f = 0;
};
struct FciqmcLog { double numer, denom, shift, norm; int n_nonz; uint32_t n_ini; size_t curr_size, n_spawn; };
// FRIES_bin/fciqmc_mol.cpp:35-480, HF trial vector, HF start; numer / denom are the rank-ordered sums every rank would see on
// the rank that owns HF (:433-441), n_nonz / n_ini are this rank's (:329-340), norm the global walker number (:417)
struct Fciqmc {
    MolSys sys;
    FciqmcParams par;
    std::mt19937 mt;
    Rng rng;
    std::vector<uint32_t> proc_scr, vec_scr;
    Vec sol;
    std::vector<det_t> trial_det, htrial_det;
    std::vector<double> trial_val, htrial_val;
    double p_doub = 0, en_shift = 0, last_norm = 0;
    det_t hf_det = 0;
    unsigned iterat = 0;
    std::vector<FciqmcLog> log;
    Comm cm;                    // ranks: every process seeds its own generator (par.seed + rank), rank 0's scrambler is broadcast
    int hf_proc = 0;
    // --trial_vec / --ini_vec (fciqmc_mol.cpp:150-177, 226-241): text vectors, added with `while (!add) perform_add` loops
    std::vector<det_t> trial_in_det, ini_det;
    std::vector<double> trial_in_val, ini_val;       // ini_val: integers for fciqmc_mol (its reader fills an int array), reals for fciqmc_fp_mol / frimulti_mol
    // frimulti_mol state: every rank's norm after the last compression, the global norm before it (frimulti_mol.cpp:227-233, 393, 414)
    std::vector<double> loc_norms; double glob_norm = 0;
    std::vector<size_t> srt; std::vector<uint8_t> keep;
    uint32_t nkept = 0;
    void setup();
    void iterate(unsigned n);
    void iterate_multi(unsigned n);      // frimulti_mol.cpp:296-425
};

// runs fn(rank) on `size` in-process ranks that share one communicator (fn receives its Comm)
void run_ranks(int size, const std::function<void(const Comm &)> &fn);

}  // namespace fo
