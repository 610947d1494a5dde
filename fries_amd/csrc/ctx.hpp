// Host-side context of the MI355X FRI engine.  Everything numeric lives in HBM; the host keeps
// pointers, launch geometry and the handful of scalars the drivers print.
#pragma once
#include "fries_dev.hpp"
#include "comp_kernels.hpp"
#include "fks2.hpp"
#include "../../include/fries_hip.h"
#include <string>
#include <vector>
#include <random>
#include <sstream>
#include <stdexcept>

struct FriesError : std::runtime_error { using std::runtime_error::runtime_error; };

void fr_set_error(const std::string &msg);

#define FR_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { throw FriesError(std::string(#call) + ": " + hipGetErrorString(e_)); } } while (0)

template <class T> static inline T *fr_alloc(size_t n) {
    T *p = nullptr;
    FR_HIP(hipMalloc((void **)&p, (n ? n : 1) * sizeof(T)));
    return p;
}
static inline unsigned fr_blocks(size_t n, size_t per) { return (unsigned)((n + per - 1) / per); }

// Optional per-kernel timing with HIP events on the engine's own stream (bench.py's roofline).
struct ProfSpan { const char *name; hipEvent_t a, b; };
struct ProfAgg { std::string name; double ms; uint64_t calls; };
struct FriesCtx;
void fr_prof_begin(FriesCtx *c, const char *name);
void fr_prof_end(FriesCtx *c);
#define FR_LAUNCH(c, name, kern, grid, block, ...) do { \
    if ((c)->prof_on) fr_prof_begin((c), name); \
    hipLaunchKernelGGL(kern, grid, block, 0, (c)->stream, __VA_ARGS__); \
    if ((c)->prof_on) fr_prof_end((c)); \
    (c)->n_kernel_launch++; } while (0)

// merge / spawn scratch (vec.hip)
struct SpawnBuf {
    uint32_t cap;
    det_t *det; double *val; uint8_t *ini;     // spawn list in arrival order
    uint32_t *slot;                            // hash slot or FR_NOPOS (dropped)
    uint32_t *flag;                            // first-arrival flags / scan scratch
    uint32_t *key[2], *pay[2];                 // radix sort ping-pong
    uint32_t *hist;                            // digit histograms
    uint32_t *pcnt;                            // block partial counts
    uint32_t *n_spawn;                         // device scalar: list length
    // exchange between ranks (n_ranks > 1)
    uint8_t *xkey; uint32_t *xcnt, *xoff, *xbucket;
    // several perform_add rounds per pass (the Adder filled up, vec.hip fr_xch_rounds): the local list kept across the rounds, and where the rounds end
    det_t *bdet = nullptr; double *bval = nullptr; uint8_t *bini = nullptr; uint32_t *bn = nullptr; uint32_t *xbounds = nullptr;
};

// vector-compression scratch (compress.hip)
struct VcompBuf {
    uint8_t *keep;          // preserved exactly
    uint8_t *del;           // zeroed by sys_comp -> delete
    double *S;              // lbound prefix
    double *psum[2]; uint32_t *pcnt[2];
    CompState *state;       // [FR_MAX_ROUNDS + 2]
    Teeth *teeth;
    double *dots;           // [2] numerator, denominator
    uint32_t *fix_list;
    SeqWork seq;            // exact in-order sums of |v|
    double *gnorm;          // [1] exact in-order sum of all |v| (find_preserve's *global_norm)
};

// addend sequences of the vector's in-order sums
struct AccAbs {             // compress_utils.cpp:34-35 (the dense space in front is not part of the array find_preserve is given)
    const double *v; const VecState *st; uint32_t n_dense = 0;
    __device__ unsigned count() const { return st->curr_size; }
    __device__ double get(size_t i) const { return i < n_dense ? 0.0 : fabs(v[i]); }
};
struct AccUnkept {          // compress_utils.cpp:98-101 and the lbound of sys_comp (:313-314)
    const double *v; const uint8_t *keep; const VecState *st;
    __device__ unsigned count() const { return st->curr_size; }
    __device__ double get(size_t i) const { return keep[i] ? 0.0 : fabs(v[i]); }
};

// pivotal compression scratch (pivotal.hip)
struct PivUnit {
    uint32_t H;             // 0: the residual piece was drawn; k >= 1: the k-th unpreserved element of the unit
    uint32_t new_resid;     // the element this unit hands on (unless it hands on the one it received)
    uint8_t pass;           // 1: border element sampled, the drawn candidate handed on
    uint8_t pad[3];
    uint32_t n_inner;       // elements of the unit before its border element
};
struct PivScal { uint32_t n_units, end_pos, too_big, n_loc, uncertain, pad; };

struct PivBuf {
    uint32_t cap = 0;
    uint32_t *start = nullptr; double *carry = nullptr;      // per unit: first element, overshoot carried in
    double *U = nullptr;                                     // two uniforms per unit
    PivUnit *unit = nullptr;
    PivScal *scal = nullptr;
    void *tile_dd = nullptr;                                 // double-double tile sums / offsets of the parallel cut-point search
    uint32_t *tile_nz = nullptr, *nz_start = nullptr;        // non-zero unpreserved elements per tile (then: before each tile) / before each unit: adds that can round
    uint64_t n_certified = 0, n_fallback = 0; uint32_t last_reason = 0;                // calls settled by the parallel search / by the sequential chain
};
// a plain device array compressed by the vector's pivotal kernels (apply_HBPP_piv's long_vec; pivotal.hip: fr_piv_comp_flat)
struct FlatPiv {
    uint32_t cap = 0;
    double *vals = nullptr;         // the values, compressed in place
    uint32_t *parent = nullptr;     // element of the short vector each value was expanded from
    VecState *st = nullptr;
    uint32_t *total = nullptr;      // [2] device scalars of the expansion / collapse
    VcompBuf vc{};
    PivBuf piv;
};
// ranks (include/fries_hip.h: fries_comm).  size == 1: no callbacks, the "gathered" block is the send block.
struct fries_comm_ops {
    void *user; int32_t rank, size;
    void *small_send, *small_recv, *big_send, *big_recv; uint64_t big_bytes;
    int (*allgather)(void *, uint64_t, void *);
    int (*alltoallv)(void *, const uint64_t *, const uint64_t *, void *);
};
struct FriesCtx;
// gathers c->comm.small_send[0, bytes) of every rank; returns the device address of the rank-ordered blocks
const void *fr_allgather(FriesCtx *c, size_t bytes);

// FCIQMC work arrays (fciqmc.hip): per stored determinant and per spawning attempt
struct FqWork {
    uint32_t cap_d, cap_a;
    uint32_t *n_doub, *n_att, *att_off; double *new_val;
    uint32_t *blk_att, *blk_nz, *blk_ini, *blk_sp, *totals, *heavy;      // totals: attempts, non-zero, initiators, heavy determinants
    double *sp_val; det_t *sp_det; uint8_t *sp_ini;
    double *norm;
    uint32_t *o1cnt;            // heat-bath generator: samples per first occupied electron, [determinant][electron]
    // frimulti_mol (multinomial matrix compression): samples per column from the systematic comb, real-valued spawns
    int multi; uint32_t *n_walk; double samp_unit, init_f;
};

struct FriesCtx {
    int device = 0;
    hipStream_t stream = nullptr;
    fries_comm_ops comm{};
    int rank = 0, n_ranks = 1, hf_proc = 0;
    bool use_comm = false;
    uint32_t *d_proc_scr = nullptr;          // proc_hash_ scrambler on the device
    void *own_small = nullptr;               // size == 1: engine-owned small_send
    double *d_norms_keep = nullptr, *d_seq_scratch = nullptr;
    // FCIQMC driver (fciqmc.hip)
    bool fq_mode = false;
    fries_fciqmc_params fq{};
    fries_frimulti_params fm{};
    bool dots_slot0_from_hf = false;         // fciqmc_fp_mol's gather quirk (compress.hip: fr_dots)
    FqWork fqw{};
    // Hubbard-Holstein driver (hh.hip)
    bool hh_mode = false, hh_keep0 = false;
    int spin_parity = 0;                     // fries_set_spin_parity: +-1 = time-reversal symmetrised vectors (k_enum, k_final_eval_piv)
    fries_hh_params hh{};
    uint32_t *d_vec_scr = nullptr;
    det_t *hh_fdet = nullptr; double *hh_ovlp = nullptr;
    FlatPiv flat;                   // apply_HBPP_piv
    uint32_t *pv_goff = nullptr;    // first long index of every short element's group (W.cap)
    unsigned long long *hhf_cnt = nullptr;      // frifull_hh: {adds tried, adds written} of the iteration
    uint4 *stg_mem = nullptr, *spill_mem = nullptr; uint32_t *spill_cnt_mem = nullptr; uint8_t *tile_dirty_mem = nullptr;      // CompWork::stg etc. (set per stage: not with the propagation repair)
    uint32_t adder_cap = 0;                  // the reference's Adder capacity per destination (frisys_mol.cpp:109-110)
    uint64_t n_collectives = 0;
    uint64_t n_adder_rounds = 0;             // perform_add rounds beyond the usual one per pass (the Adder filled up)
    // system
    uint32_t n_orb = 0, n_elec = 0;
    double *d_h = nullptr, *d_eris = nullptr;
    HbTables *d_hb = nullptr;
    HbTables h_hb;
    double hf_en = 0, p_doub = 0;
    det_t hf_det = 0;
    // solution vector
    VecDev vec{};
    VecState h_vst{};
    // HB-PP work arrays
    CompWork W{};
    Fks2Work F2{};
    uint32_t *fks_sxk8 = nullptr; double *fks_sxg8 = nullptr;
    FksSaved *fks_saved = nullptr; uint32_t *fks_wk = nullptr, *fks_wkx = nullptr; double *fks_wg = nullptr, *fks_wgx = nullptr;    // per-stage warm-start records
    struct FksSeq *fks_seq = nullptr;        // sequential find_keep_sub (fks_seq.hpp)
    FksSq fsq{};                             // ... its parallel form: work arrays, knobs, statistics
    bool fks_no_merged_norm = false;         // FRIES_FKS_NO_MERGED_NORM=1 (ranks): the remaining norms in a message of their own instead of riding with the closing pass's flag
    double *d_norms_all = nullptr;           // ... where k_fks_close_flag2 leaves them
    bool fsq_walk_only = false;              // FRIES_FKS_SEQ_WALK=1: the one-wave walk for every sweep
    int fsq_guess_rounds = 24, fsq_exact_rounds = 6;
    bool fsq_use_maps = true, fsq_check = false;   // the chain as integer arithmetic inside a binade (FRIES_FSQ_MAPS=0: element by element); FRIES_FSQ_CHECK=1: both, compared
    int fsq_sparse_max = 32;                 // k_fsq_chain: a tile with at most this many touched element pairs (of 128) is walked pair by pair
    uint64_t n_fsq_guess = 0, n_fsq_exact = 0, n_fsq_chain_tiles = 0, n_fsq_walk = 0, n_fsq_walk_tiles = 0;
    FksHost *h_fks = nullptr;                // host side of Fks2Work::hm
    // small device-to-host readbacks (state structs, counters, norms): a one-wave kernel copies them into a pinned, host-coherent block
    // mapped into the device's address space; the host reads it after the stream synchronisation it needs anyway.  (hipMemcpy of a few
    // bytes into pageable memory costs a blit kernel, a staging buffer and a host copy per call.)
    uint8_t *h_rb = nullptr, *d_rb = nullptr; size_t rb_used = 0;
    static constexpr size_t RB_BYTES = 8192, RB_HELD_BYTES = 2048;     // the ring, and one slot for a readback that is held across other readbacks (fr_dots_enqueue)
    // behind them: MISC_BYTES that kernels write for the host -- word 0 the ticket of fr_stream_wait, words 16.. the five stages' emission counts
    static constexpr size_t MISC_BYTES = 256;
    uint32_t ticket = 0;                     // last ticket handed to k_ticket
    bool wait_by_sync = false;               // FRIES_WAIT_SYNC=1: hipStreamSynchronize instead of polling the ticket word
    volatile uint32_t *h_misc() const { return (volatile uint32_t *)(h_rb + RB_BYTES + RB_HELD_BYTES); }
    uint32_t *d_misc() const { return (uint32_t *)(d_rb + RB_BYTES + RB_HELD_BYTES); }
    uint32_t prop_tag = 0;                   // last tag handed to a comb-repair round (k_sys_walk)
    int fks_rec_at = -1;                     // FRIES_FKS_REC_AT=k: the replay that records the tiles' margins (default: rounds hint - 2)
    bool fks_group_warm_all = true;          // FRIES_GROUP_WARM_ALL=0: only stage 1 starts its first replay from the previous iteration's per-group prefixes
    bool fks_no_speculation = false;         // FRIES_FKS_NO_SPECULATION=1: nothing is enqueued behind the closing pass before the host has seen its flag
    bool fks_light_full_grid = true;         // FRIES_FKS_LIGHT_FULL_GRID=1: light replays launch one workgroup per tile
    bool fks_fuse_totals = false;            // FRIES_FKS_FUSE_TOTALS=1: the last workgroup of k_fks_scan does k_fks_totals' work
    bool fks_no_ext = false;                 // FRIES_FKS_NO_EXT=1: a wave re-decides whenever the stage runs another number of sweeps than it last ran
    bool fks_no_group_warm = false;          // FRIES_NO_GROUP_WARM=1
    bool fks_no_light = false;               // FRIES_FKS_NO_LIGHT=1: every replay evaluates every tile (tests, comparisons)
    bool fks_no_closing = false;             // FRIES_FKS_NO_CLOSING=1: a confirming replay and a final pass of its own instead of the closing pass (k_fks_sweep MODE 4)
    bool fks_no_collapse_walk = false;       // FRIES_FKS_COLLAPSE_WALK=0: keep the parallel replay's result in collapsing stages (fast, not bit-identical to the reference there)
    bool fks_force_seq = false;              // FRIES_FKS_SEQ=1: every stage through the in-order walk (tests)
    uint64_t n_fks_sequential = 0;
    unsigned fks_grid = 1280, fks_grid0 = 1280;
    bool warm_start = true;
    uint32_t *c_pos = nullptr, *c_orbs = nullptr; double *c_val = nullptr;   // compacted apply_HBPP_sys output
    uint32_t *d_nsucc = nullptr;
    SpawnBuf sp{};
    VcompBuf vc{};
    PivBuf piv{};
    // frifull_mol: per-determinant excitation counts / offsets of the deterministic H application (system.hip)
    bool full_mode = false;
    uint32_t full_cap = 0, *full_cnt = nullptr, *full_nz = nullptr, *full_off = nullptr, *full_list = nullptr;
    uint32_t *d_err = nullptr;
    uint32_t *d_tie = nullptr;               // [2] tie statistics when enabled (fries_tie_margins): float bits of the smallest relative margin in find_keep_sub / find_preserve
    // optional driver inputs, set before fries_frisys_setup: --trial_vec, --ini_vec, --ham_shift (frisys_mol.cpp:95-98, 157-181, 264-274)
    std::vector<det_t> in_trial_det, in_ini_det; std::vector<double> in_trial_val, in_ini_val;
    bool fq_ini_real = false;                // the --ini_vec values are reals (frimulti_mol; fciqmc_fp_mol says so through its parameters)
    // --det_space (semi-stochastic, one rank): the dense determinants (positions 0 .. n-1 of the vector) and H inside that space
    // times -eps, per determinant its singles then its doubles (frisys_mol.cpp:236-239, 347-401)
    std::vector<det_t> in_det_space;
    uint32_t n_dense_h = 0, n_dense_h_nz = 0;       // symmetry-allowed excitations of this rank's dense determinants / those with a non-zero element
    uint32_t n_dense_h_glob = 0;                    // the former summed over the ranks: what the matrix sample budget is reduced by
    uint32_t *d_dh_from = nullptr; det_t *d_dh_to = nullptr; double *d_dh_el = nullptr, *d_dense_norm = nullptr;
    std::vector<uint32_t> dense_sizes;      // every rank's n_dense (dense.txt of a checkpoint)
    bool ham_shift_set = false; double ham_shift_hf_en = 0;
    // trial vectors (replicated, small)
    uint32_t n_trial = 0, n_htrial = 0;
    det_t *tr_det = nullptr, *htr_det = nullptr;
    double *tr_val = nullptr, *htr_val = nullptr;
    // driver state (FRIES_bin/frisys_mol.cpp)
    std::mt19937 mt;
    std::vector<uint32_t> proc_scr, vec_scr;
    std::vector<uint32_t> in_proc_scr;       // --load_dir: the proc scrambler of hash.dat instead of fresh draws (fries_set_proc_scrambler)
    std::vector<uint32_t> in_vec_scr;        // fries_set_vec_scrambler: the vector hash's scrambler (frisys_hh keys its table on it)
    double eps = 0, target_norm = 0, init_thresh = 0, en_shift = 0, last_one_norm = 0;
    uint32_t vec_nonz = 0, mat_nonz = 0;
    bool new_hb = true;
    unsigned iterat = 0;
    int rounds_hint[8] = {3, 3, 3, 3, 3, 3, 3, 3};
    int fks_iters[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // last iteration's observables
    double numer = 0, denom = 0, glob_norm = 0;
    uint32_t nkept = 0, num_success = 0, comp_len[5] = {0, 0, 0, 0, 0};
    uint64_t n_kernel_launch = 0, tot_spawns = 0, tot_iters = 0, tot_fks_iters = 0, tot_stage_elems = 0;
    // profiling
    bool prof_on = false;
    int dbg = 0;
    std::vector<ProfSpan> prof_spans;
    std::vector<hipEvent_t> prof_pool;
    std::vector<ProfAgg> prof_agg;
};

// vec.hip
void fr_vec_alloc(FriesCtx *c, VecDev *v, uint32_t cap);
void fr_vec_sync_state(FriesCtx *c, VecDev *v, VecState *out);
void fr_vec_merge(FriesCtx *c, VecDev *v, uint32_t n_spawn_bound, bool same_column, bool arrival_order = false);
void fr_vec_delete_flagged(FriesCtx *c, VecDev *v, const uint8_t *d_flags, uint32_t n);
void fr_vec_maybe_rebuild(FriesCtx *c, VecDev *v);
void fr_vec_reserve_hash(FriesCtx *c, VecDev *v, uint32_t n_new);
void fr_spawn_alloc(FriesCtx *c, uint32_t cap);
void fr_xch_alloc(FriesCtx *c, uint32_t cap);
uint32_t fr_spawn_exchange(FriesCtx *c, uint32_t n_local, int one_pass = 0, bool *merged = nullptr);      // 0: two passes (frisys_mol); 1: one pass, flag inside an integer value; 2: one pass, flag in bit 63 of the index
// hbpp.hip
void fr_hbpp_alloc(FriesCtx *c, uint32_t cap);
void fr_hbpp_apply(FriesCtx *c, uint32_t n_samp, const double rn[5]);
void fr_spawn_from_comp(FriesCtx *c);
// compress.hip
void fr_vcomp_alloc(FriesCtx *c, uint32_t cap);
void fr_death_clone(FriesCtx *c, uint32_t vec_size_before);
void fr_abs_sums(FriesCtx *c);
void fr_find_preserve(FriesCtx *c, uint32_t *n_samp_io, double *glob_norm);
void fr_sys_comp(FriesCtx *c, uint32_t n_samp, double rn);
void fr_dots(FriesCtx *c, double *numer, double *denom);
const void *fr_dots_enqueue(FriesCtx *c);
void fr_dots_collect(FriesCtx *c, const void *h_d, double *numer, double *denom);
void fr_unkept_norm(FriesCtx *c, uint32_t bound);
// pivotal.hip
void fr_piv_flat_reserve(FriesCtx *c, uint32_t cap);
void fr_piv_flat_free(FriesCtx *c);
void fr_piv_comp_flat(FriesCtx *c, uint32_t n, uint32_t compress_size);
void fr_hbpp_piv_apply(FriesCtx *c, uint32_t n_samp, int unit_matrel, uint32_t stage_len[5]);
void fr_piv_comp(FriesCtx *c, uint32_t compress_size, uint32_t *n_kept, double *glob_norm);
void fr_test_piv_adjust(FriesCtx *c, uint32_t *n_loc_io, double exp_loc, uint32_t n_tot, double tot_norm, double *new_norm, uint8_t *flags_out);
// fciqmc.hip
void fr_fq_setup(FriesCtx *c, const fries_fciqmc_params *p);
void fr_fq_iterate(FriesCtx *c, fries_fciqmc_log *lg);
// hh.hip
void fr_multi_setup(FriesCtx *c, const fries_frimulti_params *p);
void fr_multi_iterate(FriesCtx *c, fries_fciqmc_log *lg);
double fr_abs_norm(FriesCtx *c);
void fr_multi_walks(FriesCtx *c, double rn, double prev_glob_norm, uint32_t n_teeth, uint32_t *n_walk, double *unit_out);
void fr_hh_setup(FriesCtx *c, const fries_hh_params *p);
void fr_hh_iterate(FriesCtx *c, fries_iter_log *lg);
void fr_hh_apply(FriesCtx *c, uint32_t n_samp, const double rn[2]);
void fr_hh_clear_pos0(FriesCtx *c);
void fr_hh_stage(FriesCtx *c, int stage, uint32_t n_samp, double rn);
void fr_hh_ref_ovlp(FriesCtx *c, double out[3]);
int fr_host_idx_to_proc(const FriesCtx *c, det_t d);
// system.hip
void fr_dense_h_setup(FriesCtx *c);      // system.hip
// enqueue a copy of `bytes` (a multiple of 4, <= 2 KB) at device address src into the readback block; -> host address to read AFTER the next
// synchronisation of the stream.  The block is a ring: a slot stays valid until ~RB_BYTES more have been asked for.  (vec.hip)
const void *fr_readback(FriesCtx *c, const void *src, size_t bytes, bool held = false, uint32_t *ticket = nullptr);      // ticket != nullptr: the copy raises a ticket itself (fr_stream_wait_ticket)
uint32_t fr_ticket_reserve(FriesCtx *c);
// Waits until everything enqueued on the context's stream so far has run: a one-thread kernel stores a ticket into host-coherent pinned
// memory and the host polls that word.  (hipStreamSynchronize returns ~20 us after the stream has drained; the iteration has about a
// dozen such waits, during each of which the GPU idles.)
void fr_stream_wait(FriesCtx *c);
uint32_t fr_stream_ticket(FriesCtx *c);
void fr_stream_wait_ticket(FriesCtx *c, uint32_t t);
void fr_rb_init(FriesCtx *c);
void fr_system_upload(FriesCtx *c, uint32_t n_orb, uint32_t n_elec, const uint8_t *irreps, const double *h_core, const double *eris);
void fr_h_trial_setup(FriesCtx *c);
void fr_h_diag_vec(FriesCtx *c, double id_fac, double h_fac);
uint64_t fr_h_offdiag_vec(FriesCtx *c, double h_fac);
