/* C ABI of the MI355X-native FRI engine (libfries_hip.so).
 *
 * The reference (sgreene8/FRIES) has no FFI: its hot path is C++ called in-process through the
 * headers it installs under include/FRIES/.  Each entry point below names the reference
 * interface it stands in for; the binding a FRIES maintainer would add is shown in
 * INTEGRATION.md.  Conventions: plain pointers and sizes only, host buffers unless a name says
 * "device"; every function returns 0 on success or a negative code, and fries_last_error()
 * then holds the message (the reference throws std::runtime_error and prints it,
 * FRIES_bin/frisys_mol.cpp:562-565).  A determinant is a uint64_t whose bit i is spin orbital i
 * (alpha 0..n_orb-1, beta n_orb..2n_orb-1) -- the reference's little-endian byte string
 * (FRIES/det_store.h:23-26) for 2*n_orb <= 64.
 */
#ifndef FRIES_HIP_H
#define FRIES_HIP_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct fries_ctx fries_ctx;

const char *fries_last_error(void);
/* number of HIP devices visible; 0 means the engine cannot run (there is no CPU fallback) */
int fries_device_count(void);

int fries_ctx_create(fries_ctx **out, int device);
void fries_ctx_destroy(fries_ctx *ctx);

/* parse_fcidump's result handed to the device: FRIES/io_utils.cpp:241-318 (fcidump_input),
 * SymmERIs packing FRIES/ndarr.hpp:206-244, SymmInfo FRIES/Hamiltonians/molecule.hpp:265-280,
 * plus set_up(tot_orb, n_orb, eris) FRIES/Hamiltonians/heat_bathPP.cpp:99-179. */
int fries_set_molecule(fries_ctx *ctx, uint32_t n_orb, uint32_t n_elec, const uint8_t *irreps,
                       const double *h_core, const double *eris_packed);

/* hb_info accessors (FRIES/Hamiltonians/heat_bathPP.hpp:25-34).  which: 0 s_tens[n], 1 d_same[n(n-1)/2],
 * 2 d_diff[n*n], 3 exch_sqrt[n(n-1)/2], 4 diag_sqrt[n], 5 exch_norms[n], 6 s_norm[1]. */
int fries_get_hb_tensor(fries_ctx *ctx, int which, double *out, size_t cap, size_t *len);
int fries_set_hb_tensor(fries_ctx *ctx, int which, const double *in, size_t len);
double fries_hf_energy(fries_ctx *ctx);     /* diag_matrel of the HF determinant, frisys_mol.cpp:100 */

/* Batched Slater-Condon elements (FRIES/Hamiltonians/molecule.cpp:983-1029, 76-105, 26-42) and
 * fermionic signs (FRIES/fci_utils.c:46-96).  orbs: n x 4 bytes, (o1,o2,u1,u2) or (o,u,0,0).
 * kind: 0 diag_matrel(det), 1 sing_matr_el_nosgn, 2 doub_matr_el_nosgn; sign receives
 * sing_parity / doub_parity for kinds 1, 2. */
int fries_matrel_batch(fries_ctx *ctx, int kind, const uint64_t *dets, const uint8_t *orbs, size_t n,
                       double *out, int32_t *sign);

/* ---- ranks ------------------------------------------------------------------------------------
 * The reference shards the solution vector over MPI ranks by determinant hash (DistVec::idx_to_proc,
 * FRIES/vec_utils.hpp:360-379) and synchronises with blocking collectives on MPI_COMM_WORLD:
 * sum_mpi = MPI_Allgather + a sum in rank order (compress_utils.hpp:170-231), MPI_Alltoallv of the
 * pending adds (Adder::perform_add, vec_utils.hpp:991-1019), MPI_Allgather of the local norms
 * (compress_utils.cpp:326, :818), MPI_Bcast of rank 0's uniform (:291, :806).
 * The engine does the same with ONE process per GPU and two caller-supplied collectives that run
 * on caller-owned DEVICE staging buffers, stream-ordered like an RCCL call: when the callback
 * returns, the collective must be ordered after everything already enqueued on `stream`, and work
 * enqueued on `stream` afterwards must see its result.  fries_amd/comm.py implements them with
 * torch.distributed (backend "nccl" = RCCL over xGMI; "gloo" for tests); INTEGRATION.md shows the
 * GPU-aware-MPI version a FRIES maintainer would write. */
#define FRIES_COMM_SMALL_BYTES 2048
#define FRIES_COMM_MAX_RANKS 64
typedef struct {
    void *user;
    int32_t rank, size;
    void *small_send;       /* device, FRIES_COMM_SMALL_BYTES */
    void *small_recv;       /* device, size * FRIES_COMM_SMALL_BYTES */
    void *big_send;         /* device, big_bytes: all-to-all segments, contiguous in destination order */
    void *big_recv;         /* device, big_bytes: received segments, contiguous in source order */
    uint64_t big_bytes;
    /* every rank contributes small_send[0, bytes); rank p's block lands at small_recv + p * bytes */
    int (*allgather)(void *user, uint64_t bytes, void *stream);
    /* send_bytes[d] bytes of big_send go to rank d, recv_bytes[s] bytes arrive from rank s (host arrays of `size`) */
    int (*alltoallv)(void *user, const uint64_t *send_bytes, const uint64_t *recv_bytes, void *stream);
} fries_comm;
/* Native transports (csrc/comm_native.hip) that fill a fries_comm -- the C++ counterpart of MPI_COMM_WORLD in the reference:
 *   RCCL: one process per MI355X.  Rank 0 makes the 128-byte id (fries_rccl_unique_id) and the launcher hands it to every rank
 *   (a file, MPI_Bcast, a TCP store: the drivers use a rendezvous file); fries_rccl_create joins the communicator on `device`.
 *   The all-gather is ncclAllGather and the all-to-all ncclAllToAllv, both on the engine's stream.
 *   local: `size` ranks = `size` host threads of one process, each driving its own context, on any devices; several ranks
 *   may share one GPU (RCCL refuses that), so multi-rank runs can be checked on a one-GPU machine.  Every rank's thread must
 *   enter each collective (the engine does); create the group once, then one transport per rank.
 * big_bytes >= 16 x (mat_nonz + 4096), the same on every rank. */
typedef struct fries_transport fries_transport;
typedef struct fries_local_group fries_local_group;
int fries_rccl_unique_id(uint8_t id[128]);
int fries_rccl_create(fries_transport **out, const uint8_t id[128], int rank, int size, int device, uint64_t big_bytes);
int fries_local_group_create(fries_local_group **out, int size, uint64_t big_bytes);
void fries_local_group_destroy(fries_local_group *g);
int fries_local_create(fries_transport **out, fries_local_group *g, int rank, int device);
/*   host: the caller's own collectives on HOST buffers -- for a program that already lives in an MPI communicator (the reference's
 *   drivers call MPI directly: MPI_Allgather inside sum_mpi, MPI_Alltoallv inside Adder::perform_add).  The transport moves the staging
 *   blocks to pinned host memory, calls back, and moves the result to the device; no GPU-aware MPI is needed.  allgather: every rank
 *   contributes `bytes` bytes, rank p's block lands at recv + p * bytes.  alltoallv: send holds the segments for ranks 0 .. size-1 back to
 *   back (send_bytes[d] each), recv receives the segments from ranks 0 .. size-1 back to back (recv_bytes[s] each).  Return 0 on success.
 *   include/FRIES/backend.hpp builds these two from MPI_Allgather / MPI_Alltoallv when the program runs on more than one rank. */
typedef struct {
    void *user;
    int (*allgather)(void *user, const void *send, void *recv, uint64_t bytes);
    int (*alltoallv)(void *user, const void *send, const uint64_t *send_bytes, void *recv, const uint64_t *recv_bytes);
} fries_host_collectives;
int fries_hostcomm_create(fries_transport **out, const fries_host_collectives *cb, int rank, int size, int device, uint64_t big_bytes);
/* fills *comm for fries_set_comm; the transport must outlive the context */
int fries_transport_comm(fries_transport *t, fries_comm *comm);
int fries_transport_counts(fries_transport *t, uint64_t *n_allgather, uint64_t *n_alltoallv);
void fries_transport_destroy(fries_transport *t);

/* call between fries_set_molecule and fries_frisys_setup; without it the context is one rank */
int fries_set_comm(fries_ctx *ctx, const fries_comm *comm);
/* the HIP stream (hipStream_t) every kernel of this context is launched on */
void *fries_stream(fries_ctx *ctx);
/* DistVec::idx_to_proc (vec_utils.hpp:360-379) for n determinants; valid after fries_frisys_setup */
int fries_idx_to_proc(fries_ctx *ctx, const uint64_t *dets, size_t n, int32_t *proc);

/* frisys_mol's run parameters (FRIES_bin/frisys_mol.cpp:16-33) plus the seed the reference takes
 * from the wall clock (:104-106). */
typedef struct {
    double epsilon, target_norm, initiator;
    uint32_t vec_nonz, mat_nonz, max_dets;
    uint32_t seed;
    int32_t hb_unnorm;      /* 1: --distribution HB_unnorm, 0: HB */
} fries_frisys_params;

typedef struct {
    double numer, denom;    /* projnum.txt / projden.txt */
    double shift, norm;     /* S.txt / norm.txt */
    uint32_t nkept;         /* nkept.txt */
    int32_t n_nonz;
    uint32_t curr_size;
    uint32_t num_success;   /* comp_vecs.vec_len after apply_HBPP_sys */
    uint32_t comp_len[5];
    uint32_t err;
} fries_iter_log;

/* Optional inputs of the frisys_mol and fciqmc_mol drivers, each before fries_frisys_setup / fries_fciqmc_setup (for fciqmc_mol the
 * initial values are walker numbers, and the last entry of the trial file counts twice in the trial vector, as in the
 * reference's `while (!add) perform_add` loop, fciqmc_mol.cpp:163-170; fciqmc_fp_mol, real_walkers = 1, keeps the initial values real,
 * fciqmc_fp_mol.cpp:157-185, 233-246; fries_frimulti_setup takes --ini_vec with real values, frimulti_mol.cpp:205-215, and answers a trial vector with the
 * reference's own error "Insufficient memory allocated in adder", frimulti_mol.cpp:149-157, which that driver raises for every trial file):
 *   --trial_vec (frisys_mol.cpp:157-181): the vector the energy is projected on, entries add()ed in the given order;
 *   --ini_vec   (:264-274): the starting vector instead of 100 x HF, entries add()ed in the given order (from rank 0);
 *   --ham_shift (:95-98): the offset subtracted from every diagonal element instead of the HF energy (pass ham_shift - core_en). */
int fries_set_trial_vector(fries_ctx *ctx, const uint64_t *dets, const double *vals, size_t n);
int fries_set_initial_vector(fries_ctx *ctx, const uint64_t *dets, const double *vals, size_t n);
int fries_set_ham_shift(fries_ctx *ctx, double hf_en);
/* --det_space (frisys_mol.cpp:236-239, 347-401, 480-485; DistVec::init_dense, vec_utils.hpp:858-897): the determinants of the semi-stochastic
 * dense space.  They take positions 0 .. n - 1 of the vector, are never compressed or deleted, H restricted to them is applied exactly every
 * iteration, and the matrix compression gets mat_nonz minus the number of (symmetry-allowed) matrix elements inside the space.  Before
 * fries_frisys_setup.  With ranks: pass the whole list on every rank; each keeps the determinants it owns, in list order. */
int fries_set_det_space(fries_ctx *ctx, const uint64_t *dets, size_t n);
/* frisys_mol.cpp:76-346: scramblers, solution vector, HF trial vector and H*trial, p_doub, HF start.
 * With ranks: vec_nonz / mat_nonz / target_norm are the GLOBAL budgets, max_dets is per rank, and every rank
 * must pass the same seed (the reference broadcasts rank 0's scramblers and uniforms). */
int fries_frisys_setup(fries_ctx *ctx, const fries_frisys_params *p);
/* frisys_mol.cpp:405-552, n_iter times; logs may be NULL or hold n_iter entries */
int fries_frisys_iterate(fries_ctx *ctx, uint32_t n_iter, fries_iter_log *logs);
double fries_p_doub(fries_ctx *ctx);
/* the proc / vec hash scramblers drawn at setup (frisys_mol.cpp:132-145; hash.dat holds the first, io_utils.cpp:589-606).
 * Either pointer may be NULL; n = 2 * n_orb entries each. */
int fries_get_scramblers(fries_ctx *ctx, uint32_t *proc_scrambler, uint32_t *vec_scrambler, size_t n);
/* --load_dir: the proc scrambler read from hash.dat (load_proc_hash, io_utils.cpp:608-619; frisys_mol.cpp:128-130) instead of 2 n_orb
 * fresh draws -- the shards of a restarted run must be the shards of the run that wrote the checkpoint.  Before fries_frisys_setup;
 * the setup then draws only the vec scrambler, as the reference does. */
int fries_set_proc_scrambler(fries_ctx *ctx, const uint32_t *proc_scrambler, size_t n);
uint64_t fries_kernel_launches(fries_ctx *ctx);

/* ---- frifull_mol: FRI with the Hamiltonian applied in full (FRIES_bin/frifull_mol.cpp:258-304): systematic compression of
 * the vector to vec_nonz non-zeros, then every symmetry-allowed single and double excitation of every remaining determinant
 * (h_op_diag / h_op_offdiag, FRIES/Hamiltonians/molecule.cpp:205-219, 448-665) merged into the other value column.  HF trial
 * vector, start from 100 x HF, one rank.  Logs: numer / denom as written to projnum.txt / projden.txt (:296-300), norm = the
 * one-norm before compression, num_success = non-zero excitations added; n_nonz / curr_size count stored determinants (an entry
 * is released only when it is zero in both value columns, vec_utils.hpp:458-476, so the table keeps growing as in the
 * reference).  spawn_cap: excitations merged per batch (0 = 8e6). */
typedef struct {
    double epsilon, target_norm;
    uint32_t vec_nonz, max_dets, seed, spawn_cap;
} fries_frifull_params;
int fries_frifull_setup(fries_ctx *ctx, const fries_frifull_params *p);
int fries_frifull_iterate(fries_ctx *ctx, uint32_t n_iter, fries_iter_log *logs);

/* ---- frisys_hh: FRI with systematic matrix compression for the 1-D Hubbard-Holstein model (FRIES_bin/frisys_hh.cpp),
 * open boundaries, hopping t = 1, 3 bits per phonon (:96).  Parameters are those of the reference's params file
 * (parse_hh_input, FRIES/io_utils.cpp:320-405: n_elec, lat_len, eps, U, omega, g, gs_energy) and command line (:15-23),
 * plus the seed.  Index = [alpha sites | beta sites | 3 bits per site] in one uint64_t (FRIES/hh_vec.hpp), n_sites <= 12.
 * Start: 100 x the Neel state (:113-117).  Logs: numer / denom are what projnum.txt / projden.txt receive on the rank that
 * owns the Neel state (:338-347) and 0 elsewhere; num_success = samples after the second compression.
 * Shares the vector accessors, fries_set_comm and fries_frisys_restart with frisys_mol. */
typedef struct {
    uint32_t n_elec, n_sites;
    double eps, U, omega, g, gs_energy, target_norm, initiator;
    uint32_t vec_nonz, max_dets, seed;
    uint32_t full;      /* 0: frisys_hh.  1: frifull_hh (FRIES_bin/frifull_hh.cpp) -- every hop and phonon move of every stored state
                         * instead of the two matrix compressions; num_success = number of adds; over ranks as long as a shard's adds of one
                         * iteration fit one Adder (200 000), as the reference ships them in one round then */
} fries_hh_params;
int fries_hh_setup(fries_ctx *ctx, const fries_hh_params *p);
int fries_hh_iterate(fries_ctx *ctx, uint32_t n_iter, fries_iter_log *logs);
/* The same driver's operators one by one, for a host loop that keeps the reference's shape (FRIES_bin/frisys_hh.cpp behind include/FRIES):
 *   fries_set_vec_scrambler   the vector hash's scrambler (the reference's table compares a truncated index inside a bucket chosen by that hash,
 *                             det_hash.hpp:47, so which states share an entry depends on it); with fries_set_proc_scrambler, before fries_hh_setup;
 *   fries_hh_comp_sub         one of the two comp_sub calls (:187-224): stage 1 on the magnitudes of the stored vector with the rows (t, g),
 *                             stage 2 on stage 1's emissions with their uniform subdivisions; outputs as comp_idx[k][0..1] and the value array;
 *   fries_hh_ref_ovlp         calc_ref_ovlp of this rank's shard against the Neel state (hub_holstein.hpp:93-186);
 * the vector itself goes through fries_vec_load / fries_vec_add_to / fries_vec_add_vecs / fries_find_preserve / fries_sys_comp. */
int fries_set_vec_scrambler(fries_ctx *ctx, const uint32_t *vec_scrambler, size_t n);
int fries_hh_comp_sub(fries_ctx *ctx, int stage, uint32_t n_samp, double rn, uint32_t *idx0, uint32_t *idx1, double *vals, size_t cap, size_t *n_out);
int fries_hh_ref_ovlp(fries_ctx *ctx, double *ovlp);

/* ---- fciqmc_mol: FCIQMC with the near-uniform or the heat-bath excitation generator (FRIES_bin/fciqmc_mol.cpp), HF trial
 * vector, start from 100 walkers on HF; one rank or hash-sharded ranks (fries_set_comm: one all-to-all of the spawns per
 * iteration; n_nonz / n_ini in the log are this rank's, norm and the projections global).  The reference's sequential mt19937
 * stream cannot be replayed in parallel;
 * the engine draws from a counter-based stream keyed by (seed, iteration, determinant, attempt, purpose) -- the same uniforms
 * in distribution -- which the CPU oracle shares, and the oracle's functions and loop are pinned against the reference on the
 * reference's own stream (oracle/ref_harness.cpp: fciqmc).  Walker numbers are exact integers in the vector's doubles. */
typedef struct {
    double epsilon;
    uint32_t target_walkers, initiator, max_dets, seed;
    int32_t heat_bath;          /* 0: --distribution NU (near_uniform.cpp), 1: --distribution HB (hb_doub_multi, heat_bathPP.cpp:601-683) */
    int32_t real_walkers;       /* 0: fciqmc_mol.  1: fciqmc_fp_mol (FRIES_bin/fciqmc_fp_mol.cpp): real-valued walkers -- |v| rounded stochastically to
                                 * the number of attempts (:342), spawns below 0.01 rounded and kept real otherwise (:385-390), death in place (:423-424),
                                 * then every |v| < 1 rounded to -1 / 0 / 1 and zeros deleted (:428-441); ranks as for fciqmc_mol (incl. the
                                 * gather at :461-462 that overwrites rank 0's projection terms with the HF owner's); no text vectors */
} fries_fciqmc_params;
typedef struct {
    double numer, denom;        /* projnum.txt / projden.txt */
    double shift, norm;         /* S.txt; walkers at the last shift update (N.txt), 0 on other iterations */
    int32_t n_nonz;             /* nnonz.txt */
    uint32_t n_ini;             /* nini.txt */
    uint32_t curr_size, n_spawn, n_attempts, err;
} fries_fciqmc_log;
int fries_fciqmc_setup(fries_ctx *ctx, const fries_fciqmc_params *p);
int fries_fciqmc_iterate(fries_ctx *ctx, uint32_t n_iter, fries_fciqmc_log *logs);

/* ---- frimulti_mol: FRI with multinomial matrix compression (FRIES_bin/frimulti_mol.cpp), --distribution HB (the only one the
 * reference's argument check lets through, :38-46): every iteration one systematic comb over |v| gives each column its number of
 * samples (:301-322), the samples are drawn with hb_doub_multi / sing_multin and spawn real-valued weights (:343-376), the
 * diagonal acts in place (:379-380), and the vector is compressed to vec_nonz by find_preserve + sys_comp (:385-421).  The first ten
 * iterations use mat_nonz / 10 samples (:306).  Random numbers as for fciqmc_mol: the two comb offsets per iteration come from the
 * mt19937, the sampling from the counter-based stream shared with the CPU oracle, whose loop is pinned against the reference on the
 * reference's stream (oracle/ref_harness.cpp: frimulti).  One rank or hash-sharded ranks (fries_set_comm).  Log: norm = one-norm before the compression (norm.txt),
 * n_nonz / curr_size after it, n_spawn = adds, n_attempts = samples. */
typedef struct {
    double epsilon, target_norm, initiator;
    uint32_t vec_nonz, mat_nonz, max_dets, seed;
} fries_frimulti_params;
int fries_frimulti_setup(fries_ctx *ctx, const fries_frimulti_params *p);
int fries_frimulti_iterate(fries_ctx *ctx, uint32_t n_iter, fries_fciqmc_log *logs);

/* DistVec accessors (FRIES/vec_utils.hpp:506-535): positions [0, curr_size) incl. holes (value 0) */
int fries_vec_info(fries_ctx *ctx, uint32_t *curr_size, int32_t *n_nonz, uint32_t *n_free);
int fries_vec_download(fries_ctx *ctx, uint64_t *dets, double *vals, size_t cap, size_t *n);
/* DistVec::add + perform_add(0) into column 0 with one rank (vec_utils.hpp:418-440, 606-641) */
int fries_vec_add(fries_ctx *ctx, const uint64_t *dets, const double *vals, const uint8_t *ini, size_t n);
/* the same with curr_vec_idx = column (0 or 1): column 1 collects the spawns under the initiator rule (frisys_mol.cpp:424-471).
 * With ranks the list is what THIS rank received (DistVec::add_elements, vec_utils.hpp:606-641): the caller has routed the adds
 * (MPI_Alltoallv inside Adder::perform_add) and passes them in arrival order; nothing is exchanged here. */
int fries_vec_add_to(fries_ctx *ctx, int column, const uint64_t *dets, const double *vals, const uint8_t *ini, size_t n);
/* frisys_mol.cpp:487-499: death / cloning of the first vec_size positions, add_vecs(0, 1), zero_vec on column 1 */
int fries_death_clone(fries_ctx *ctx, double eps, double shift, uint32_t vec_size);
/* DistVec::dot with H * trial (numer) and the trial vector (denom) (frisys_mol.cpp:511-517) */
int fries_dots(fries_ctx *ctx, double *numer, double *denom);
/* find_preserve (compress_utils.cpp:29-105) and sys_comp (:283-327) as two calls, as in the drivers' loops; the preserved
 * set lives on the device in between; fries_sys_comp ends with the deletes of frisys_mol.cpp:534-539 */
int fries_find_preserve(fries_ctx *ctx, uint32_t *n_samp, double *glob_norm);
int fries_sys_comp(fries_ctx *ctx, uint32_t n_samp, double rn);
/* replaces the stored vector: determinants land in positions 0..n-1 (DistVec::load, vec_utils.hpp:761-844) */
int fries_vec_load(fries_ctx *ctx, const uint64_t *dets, const double *vals, size_t n);
/* A checkpoint that holds a dense (semi-stochastic) space (DistVec::load returns n_dense_, vec_utils.hpp:766-779, 844; frisys_mol.cpp:257-263,
 * 347-401): after fries_vec_load, the first n_dense positions become this rank's dense space -- kept whatever their value, H inside the
 * space tabulated again, the matrix sample budget reduced by the number of its elements over all ranks.  Collective with ranks.
 * fries_dense_sizes: every rank's n_dense, what DistVec::save writes to dense.txt (:736-745). */
int fries_vec_set_dense(fries_ctx *ctx, uint32_t n_dense);
int fries_dense_sizes(fries_ctx *ctx, uint32_t *sizes, size_t cap, size_t *n_ranks);
/* Column mirrors for hosts that keep the reference's pointer semantics (DistVec::values(), operator[], zero_vec, matr_el_at_pos,
 * dot: vec_utils.hpp:506-508, 647-649, 577-579, 672-677, 228-238) -- what include/FRIES/vec_utils.hpp is built on.
 * fries_vec_dot_list sums in list order, bit-identical to the reference's loop. */
int fries_vec_column_download(fries_ctx *ctx, int column, double *out, size_t cap, size_t *n);
int fries_vec_column_upload(fries_ctx *ctx, int column, const double *in, size_t n);
int fries_vec_column_zero(fries_ctx *ctx, int column);
/* DistVec::add_vecs(idx1, idx2, c) (vec_utils.hpp:553-557) */
int fries_vec_add_vecs(fries_ctx *ctx, int idx1, int idx2, double c);
int fries_vec_diag_download(fries_ctx *ctx, double *out, size_t cap, size_t *n);
int fries_vec_dot_list(fries_ctx *ctx, int column, const uint64_t *dets, const double *vals, size_t n, double *out);
int fries_htrial_download(fries_ctx *ctx, uint64_t *dets, double *vals, size_t cap, size_t *n);
/* Time-reversal symmetry: vectors that hold one representative of every pair {determinant, its spin-flipped image} (the spin_parity = +-1
 * argument of the reference's h_op_offdiag and apply_HBPP_piv, FRIES/Hamiltonians/molecule.cpp:298-369, 472-552, heat_bathPP.cpp:1326-1407;
 * flip_spins, FRIES/fci_utils.c:158-204).  fries_set_spin_parity(ctx, +-1) makes every full enumeration of H on this context (H * trial at
 * setup, fries_frifull_iterate, the dense block of --det_space, fries_h_offdiag_list) and fries_apply_hbpp_piv (unnormalised heat bath
 * only, as in the reference) form the elements between the symmetrised functions; 0 switches it off.
 * fries_h_offdiag_list: h_op_offdiag(vec, dest_idx 1, h_fac 1, spin_parity) on a fresh two-column vector holding the list in column 0
 * (molecule.cpp:448-665): the stored determinants in position order with column 1. */
int fries_set_spin_parity(fries_ctx *ctx, int spin_parity);
int fries_h_offdiag_list(fries_ctx *ctx, const uint64_t *dets, const double *vals, size_t n, uint64_t *out_dets, double *out_vals, size_t cap, size_t *n_out);

/* The hot-path operators one by one, on the context's solution vector. */
/* apply_HBPP_sys (heat_bathPP.cpp:686-992) with the five uniforms it would draw; outputs as
 * comp_vecs.{det_indices2, orb_indices1, vec1}.  unit_matrel != 0 uses the |value| = 1 lambdas of
 * tests/test_hamiltonian.cpp:493-500. */
int fries_apply_hbpp_sys(fries_ctx *ctx, uint32_t n_samp, const double rn[5], int unit_matrel,
                         uint32_t *det_pos, uint8_t *orbs, double *vals, size_t cap, size_t *n_out, uint32_t comp_len[5]);
/* apply_HBPP_piv (heat_bathPP.cpp:1014-1419; spin_parity = what fries_set_spin_parity set): every factor multiplied out into the long vector, compressed by
 * piv_comp_parallel to n_samp elements and collapsed; outputs as comp_scratch->{det_indices2, orb_indices1, vec1}.  The uniforms
 * come from the context's generator in the reference's order (seed it with fries_frisys_restart), because their number
 * depends on the data.  stage_len[k] = elements after the k-th compression.  One factor may expand to at most 134e6 values. */
int fries_apply_hbpp_piv(fries_ctx *ctx, uint32_t n_samp, int unit_matrel,
                         uint32_t *det_pos, uint8_t *orbs, double *vals, size_t cap, size_t *n_out, uint32_t stage_len[5]);
/* find_preserve + sys_comp on column 0 (compress_utils.cpp:29-105, 283-327) followed by the deletes
 * of frisys_mol.cpp:534-539 */
int fries_compress_vec(fries_ctx *ctx, uint32_t n_samp, double rn, uint32_t *n_kept, double *glob_norm);

/* compress_vecs with one vector (FRIES/vec_utils.cpp:9-32): piv_comp_parallel on column 0 -- find_preserve,
 * piv_budget, adjust_probs, piv_samp_serial (compress_utils.cpp:354-681) -- then the deletes.  The uniforms are the
 * context's mt19937 stream (the generator the reference's caller passes in), two per sampling unit.  With ranks: a collective
 * call (two all-gathers); rank 0's generator also serves piv_budget. */
int fries_compress_vec_piv(fries_ctx *ctx, uint32_t n_samp, uint32_t *n_kept, double *glob_norm);
/* how many fries_compress_vec_piv calls were settled by the parallel, certified cut-point search and how many fell back to the
 * sequential one (csrc/pivotal.hip); FRIES_PIV_CHAIN=1 in the environment forces the sequential search */
int fries_piv_stats(fries_ctx *ctx, uint64_t *n_certified, uint64_t *n_fallback);
/* the next raw draw of the context's mt19937 (advances it): lets a caller interleave its own draws as the reference's
 * drivers do, and tests check the generator's position */
uint32_t fries_next_draw(fries_ctx *ctx);
/* the generator in std::mt19937's text form (operator<< / operator>>), so that a caller owning a std::mt19937 as the reference's
 * drivers do (frisys_mol.cpp:104-106) can lend it to fries_apply_hbpp_piv / fries_compress_vec_piv and take it back.
 * get: buf may be NULL to query the size (*need, with the terminator; about 6.9 kB) */
int fries_rng_set_state(fries_ctx *ctx, const char *text);
int fries_rng_get_state(fries_ctx *ctx, char *buf, size_t cap, size_t *need);

/* Restart: re-seed the driver RNG, restore the energy shift / last norm / iteration count -- the part of
 * --load_dir that is not the vector (frisys_mol.cpp:257-263, 284-286); pair with fries_vec_load. */
int fries_frisys_restart(fries_ctx *ctx, uint32_t seed, double en_shift, double last_one_norm, uint32_t iterat);

/* running totals since setup: iterations, successful spawns (num_success), kernel launches,
 * find_keep_sub replays, elements emitted by the five HB-PP stages */
int fries_counters(fries_ctx *ctx, uint64_t *iters, uint64_t *spawns, uint64_t *launches, uint64_t *fks_replays, uint64_t *stage_elems);

/* Tie statistics: how close the run came to a comparison whose outcome floating-point reassociation could change.  While enabled,
 * every find_keep_sub stage (its threshold tests v * w * budget >= norm, compress_utils.cpp:172-241) and every find_preserve round
 * (|v| >= norm / budget, :57-63) records the smallest |a - b| / max(a, b) over all its comparisons; the call returns the minima since
 * the last call (infinity when nothing was recorded) and restarts them.  The device forms those norms as prefix sums, the reference as
 * running sums: a margin far above ~1e-13 means the kept sets are provably the same.  enable: 1 start / keep recording, 0 stop. */
int fries_tie_margins(fries_ctx *ctx, int enable, double *fks_min_rel, double *fp_min_rel);

/* device-to-device copy bandwidth of the context's GPU in GB/s (read + write), `reps` copies of `bytes` on the engine's stream */
int fries_measure_copy_bandwidth(fries_ctx *ctx, size_t bytes, int reps, double *gb_per_s);
/* Per-kernel timing with HIP events recorded on the engine's own stream (off by default). */
int fries_prof_enable(fries_ctx *ctx, int on);
int fries_prof_count(fries_ctx *ctx);
int fries_prof_get(fries_ctx *ctx, int i, char *name, size_t name_cap, double *total_ms, uint64_t *calls);

/* test hook: adjust_probs (compress_utils.cpp:606-681) on column 0 with nothing preserved; flags_out[curr_size] = the
 * elements it pinned to one sampling unit */
int fries_test_piv_adjust(fries_ctx *ctx, uint32_t *n_samp_loc, double exp_nsamp_loc, uint32_t n_samp_tot, double tot_norm,
                          double *new_norm, uint8_t *flags_out);

/* test hook: positions of the first n comb teeth built from (r0, unit) -- see csrc/teeth.hpp */
int fries_test_teeth(fries_ctx *ctx, double r0, double unit, uint32_t n, double *out_pos, const double *query, uint32_t nq, uint32_t *out_below);

/* test hook: bit-exact left-to-right running sums S_i = fl(S_{i-1} + vals[i]) of non-negative
 * doubles, evaluated in parallel -- see csrc/seqsum.hpp */
int fries_test_seqsum(fries_ctx *ctx, const double *vals, uint32_t n, double start, double *out_prefix, double *out_total,
                      uint32_t *n_dirty_tiles, uint32_t *n_dirty_subs);

#ifdef __cplusplus
}
#endif
#endif
