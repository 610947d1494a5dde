// frisys_mol iteration loop (FRIES_bin/frisys_mol.cpp:76-552) on the device-resident engine,
// and the extern "C" boundary (include/fries_hip.h).
#include "ctx.hpp"
#include "../../include/fries_hip.h"
#include <cstring>
#include <cmath>
#include <climits>

static thread_local std::string g_err;
void fr_set_error(const std::string &m) { g_err = m; }
extern "C" const char *fries_last_error(void) { return g_err.c_str(); }

struct fries_ctx { FriesCtx c; };

void fr_h_apply_list(FriesCtx *c, const std::vector<det_t> &src, const std::vector<double> &val,
                     std::vector<det_t> &out_det, std::vector<double> &out_val, uint32_t *n_sing0, uint32_t *n_doub0, bool with_diag = true);
void fr_hbpp_apply_unit(FriesCtx *c, uint32_t n_samp, const double rn[5]);

// ------------------------------------------------------------------ per-kernel HIP-event timing
void fr_prof_begin(FriesCtx *c, const char *name) {
    ProfSpan sp; sp.name = name;
    for (int k = 0; k < 2; k++) {
        hipEvent_t e;
        if (!c->prof_pool.empty()) { e = c->prof_pool.back(); c->prof_pool.pop_back(); }
        else if (hipEventCreate(&e) != hipSuccess) throw FriesError("hipEventCreate failed");
        (k ? sp.b : sp.a) = e;
    }
    hipEventRecord(sp.a, c->stream);
    c->prof_spans.push_back(sp);
}
void fr_prof_end(FriesCtx *c) { hipEventRecord(c->prof_spans.back().b, c->stream); }
static void prof_collect(FriesCtx *c) {
    if (c->prof_spans.empty()) return;
    FR_HIP(hipStreamSynchronize(c->stream));
    for (auto &sp : c->prof_spans) {
        float ms = 0;
        hipEventElapsedTime(&ms, sp.a, sp.b);
        ProfAgg *a = nullptr;
        for (auto &x : c->prof_agg) if (x.name == sp.name) { a = &x; break; }
        if (!a) { c->prof_agg.push_back(ProfAgg{sp.name, 0.0, 0}); a = &c->prof_agg.back(); }
        a->ms += ms; a->calls++;
        c->prof_pool.push_back(sp.a); c->prof_pool.push_back(sp.b);
    }
    c->prof_spans.clear();
}

#define FR_API_BEGIN try {
#define FR_API_END } catch (const std::exception &e) { fr_set_error(e.what()); return -1; } return 0;

// ------------------------------------------------------------------ spawn assembly (frisys_mol.cpp:435-464)
__global__ void __launch_bounds__(FR_BLOCK) k_spawn_build(VecDev V, SpawnBuf S, const uint32_t *c_pos, const uint32_t *c_orbs, const double *c_val,
                                                          const uint32_t *n_succ, double eps, double init_thresh) {
    const uint32_t n = *n_succ;
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0) *S.n_spawn = n;
    if (j >= n) return;
    uint32_t pos = c_pos[j], ob = c_orbs[j];
    double cur = V.v0[pos];
    det_t d = V.dets[pos];
    unsigned a = fr_c(ob, 0), b = fr_c(ob, 1), u = fr_c(ob, 2), w = fr_c(ob, 3);
    double add_el = -eps * c_val[j];
    if (cur < 0) add_el *= -1;
    if (!(u == 0 && w == 0)) d = (d & ~(1ull << a) & ~(1ull << b)) | (1ull << u) | (1ull << w);     // doub_det
    else d = (d & ~(1ull << a)) | (1ull << b);                                                        // sing_det
    S.det[j] = d; S.val[j] = add_el; S.ini[j] = fabs(cur) >= init_thresh;
}

static void check_dev_err(FriesCtx *c) {
    uint32_t e = 0;
    FR_HIP(hipMemcpyAsync(&e, c->d_err, 4, hipMemcpyDeviceToHost, c->stream));
    FR_HIP(hipStreamSynchronize(c->stream));
    e |= c->h_vst.err;
    if (!e) return;
    std::string m = "device error:";
    if (e & FR_ERR_CAP) m += " vector capacity (max_dets) exceeded;";
    if (e & FR_ERR_SPAWN_CAP) m += " insufficient memory allocated for matrix compression;";
    if (e & FR_ERR_NELEC) m += " determinant created with an incorrect number of electrons;";
    if (e & FR_ERR_HASH_FULL) m += " determinant hash table full;";
    if (e & FR_ERR_ROUNDS) m += " exact-preservation rounds did not converge;";
    if (e & FR_ERR_BACKLOG) m += " too many comb repairs;";
    if (e & FR_ERR_PIV) m += " pivotal compression met an unpreserved element of two sampling units or more;";
    throw FriesError(m);
}

const void *fr_allgather(FriesCtx *c, size_t bytes) {
    if (!c->use_comm) return c->comm.small_send;
    if (bytes > 2048) throw FriesError("all-gather block exceeds FRIES_COMM_SMALL_BYTES");
    if (c->comm.allgather(c->comm.user, bytes, (void *)c->stream)) throw FriesError("fries_comm.allgather failed");
    c->n_collectives++;
    return c->comm.small_recv;
}

// DistVec::idx_to_proc (vec_utils.hpp:360-379): hash_fxn over the occupied orbitals with the proc scrambler, mod n_procs
int fr_host_idx_to_proc(const FriesCtx *c, det_t d) {
    if (c->n_ranks == 1) return 0;
    uint64_t hash = 0;
    uint32_t i = 0;
    if (c->hh_mode) {       // HubHolVec::idx_to_proc (hh_vec.hpp:56-66): occupied orbitals, then every site's phonon number
        const unsigned L = c->hh.n_sites;
        for (det_t a = d & ((1ull << (2 * L)) - 1ull); a; a &= a - 1, i++)
            hash = 1099511628211ULL * hash + (uint32_t)((i + 1u) * c->proc_scr[(unsigned)__builtin_ctzll(a)]);
        for (unsigned s = 0; s < L; s++) hash = 1099511628211ULL * hash + (uint32_t)((s + 1u) * c->proc_scr[(d >> (2 * L + 3 * s)) & 7u]);
        return (int)(hash % (uint64_t)c->n_ranks);
    }
    while (d) {
        unsigned orb = (unsigned)__builtin_ctzll(d);
        d &= d - 1;
        uint32_t term = (i + 1u) * c->proc_scr[orb];
        hash = 1099511628211ULL * hash + term;
        i++;
    }
    return (int)(hash % (uint64_t)c->n_ranks);
}


// The first n positions of the vector become the dense (semi-stochastic) space: H inside it is tabulated (fr_dense_h_setup), every rank learns
// tot_dense_h = sum_mpi(n_determ_h) (frisys_mol.cpp:399: what the matrix sample budget is reduced by) and every rank's n_dense (what
// DistVec::save writes to dense.txt, vec_utils.hpp:736-745).  zero: init_dense zeroes the values (:876-879), DistVec::load keeps them.
static void fr_dense_declare(FriesCtx *c, uint32_t n, bool zero) {
    if (n > c->h_vst.curr_size) throw FriesError("dense space larger than the stored vector");
    c->vec.n_dense = n;
    if (n && zero) FR_HIP(hipMemsetAsync(c->vec.v0, 0, 8 * (size_t)n, c->stream));
    fr_dense_h_setup(c);
    c->n_dense_h_glob = c->n_dense_h;
    c->dense_sizes.assign((size_t)c->n_ranks, n);
    if (c->use_comm) {
        const uint32_t mine[2] = {c->n_dense_h, n};
        FR_HIP(hipMemcpyAsync(c->comm.small_send, mine, 8, hipMemcpyHostToDevice, c->stream));
        const uint32_t *all = (const uint32_t *)fr_allgather(c, 8);
        std::vector<uint32_t> got(2 * (size_t)c->n_ranks);
        FR_HIP(hipMemcpyAsync(got.data(), all, 8 * (size_t)c->n_ranks, hipMemcpyDeviceToHost, c->stream));
        FR_HIP(hipStreamSynchronize(c->stream));
        uint64_t tot = 0;
        for (int r = 0; r < c->n_ranks; r++) { tot += got[2 * r]; c->dense_sizes[r] = got[2 * r + 1]; }
        if (tot > 0xffffffffull) throw FriesError("dense block of H too large");
        c->n_dense_h_glob = (uint32_t)tot;
    }
    if (c->n_dense_h_glob >= c->mat_nonz) throw FriesError("mat_nonz must exceed the number of matrix elements inside the dense space (the compression gets mat_nonz minus that many samples)");
}

static void frisys_setup(FriesCtx *c, const fries_frisys_params *p) {
    if (!c->d_eris) throw FriesError("fries_set_molecule must be called first");
    c->eps = p->epsilon; c->target_norm = p->target_norm; c->init_thresh = p->initiator;
    c->vec_nonz = p->vec_nonz; c->mat_nonz = p->mat_nonz; c->new_hb = p->hb_unnorm != 0;
    c->en_shift = 0; c->last_one_norm = 0; c->iterat = 0;
    if (p->max_dets == 0 || p->mat_nonz == 0 || p->vec_nonz == 0) throw FriesError("max_dets, mat_nonz and vec_nonz must be positive");
    c->mt.seed(p->seed);
    c->proc_scr.resize(2 * c->n_orb); c->vec_scr.resize(2 * c->n_orb);
    if (c->in_proc_scr.size() == c->proc_scr.size()) c->proc_scr = c->in_proc_scr;     // --load_dir: load_proc_hash (frisys_mol.cpp:128-130), no draws
    else for (auto &x : c->proc_scr) x = c->mt();   // frisys_mol.cpp:133-135
    for (auto &x : c->vec_scr) x = c->mt();         // :142-144
    // A shard can in principle emit the whole (global) sample budget, so the work arrays are sized for it; the
    // reference sizes them mat_nonz * 4 / n_procs and throws when a shard outgrows that (:109, heat_bathPP.cpp:700-704).
    uint32_t wcap = p->max_dets > p->mat_nonz + 4096 ? p->max_dets : p->mat_nonz + 4096;
    uint32_t spawn_length = (uint32_t)((uint64_t)p->mat_nonz * 4 / c->n_ranks);
    c->adder_cap = spawn_length > 1000000u ? 1000000u : spawn_length;        // :109-110
    if (getenv("FRIES_ADDER_SIZE")) c->adder_cap = (uint32_t)atol(getenv("FRIES_ADDER_SIZE"));      // a smaller Adder (the DistVec constructor's adder_size): the early perform_add rounds at test sizes
    if (!c->comm.small_send) { c->own_small = fr_alloc<uint8_t>(2048); c->comm.small_send = c->own_small; }
    if (!c->d_proc_scr) c->d_proc_scr = fr_alloc<uint32_t>(64);
    FR_HIP(hipMemcpyAsync(c->d_proc_scr, c->proc_scr.data(), 4 * c->proc_scr.size(), hipMemcpyHostToDevice, c->stream));
    c->hf_proc = fr_host_idx_to_proc(c, c->hf_det);
    fr_vec_alloc(c, &c->vec, p->max_dets);
    fr_hbpp_alloc(c, wcap);
    fr_spawn_alloc(c, p->mat_nonz + 4096);
    fr_xch_alloc(c, p->mat_nonz + 4096);
    fr_vcomp_alloc(c, p->max_dets);
    if (c->ham_shift_set) c->hf_en = c->ham_shift_hf_en;        // --ham_shift (:95-98)
    fr_h_trial_setup(c);        // replicated: every rank enumerates H * trial (the reference gathers the shards, vec_utils.hpp:920-952)
    if (!c->in_det_space.empty()) {
        // --det_space (frisys_mol.cpp:236-239): DistVec::init_dense adds every determinant of the file with value 1 (they take positions
        // 0 .. n - 1 of the empty vector), then zeroes the values; the entries stay
        // With ranks: rank 0 adds the whole file and the adds travel to their owners (vec_utils.hpp:866-872), i.e. every rank receives the
        // determinants it owns in file order; each rank then has a dense space of its own length.
        std::vector<det_t> mine;
        for (det_t d : c->in_det_space) if (fr_host_idx_to_proc(c, d) == c->rank) mine.push_back(d);
        const uint32_t m = (uint32_t)mine.size();
        if (m > c->sp.cap || m > p->max_dets) throw FriesError("dense space larger than the vector / spawn buffer");
        if (m) {
            std::vector<double> v(m, 1.0); std::vector<uint8_t> f(m, 1);
            FR_HIP(hipMemcpyAsync(c->sp.det, mine.data(), 8 * (size_t)m, hipMemcpyHostToDevice, c->stream));
            FR_HIP(hipMemcpyAsync(c->sp.val, v.data(), 8 * (size_t)m, hipMemcpyHostToDevice, c->stream));
            FR_HIP(hipMemcpyAsync(c->sp.ini, f.data(), m, hipMemcpyHostToDevice, c->stream));
            FR_HIP(hipMemcpyAsync(c->sp.n_spawn, &m, 4, hipMemcpyHostToDevice, c->stream));
            fr_vec_merge(c, &c->vec, m, true);
        }
        fr_vec_sync_state(c, &c->vec, &c->h_vst);
        fr_dense_declare(c, c->h_vst.curr_size, true);          // (a determinant listed twice takes one position)
    }
    if (!c->in_ini_det.empty()) {
        // --ini_vec (:264-274): rank 0 add()s every entry in file order; each rank receives the ones it owns in that order
        std::vector<det_t> d; std::vector<double> v;
        for (size_t i = 0; i < c->in_ini_det.size(); i++)
            if (c->in_ini_val[i] != 0 && fr_host_idx_to_proc(c, c->in_ini_det[i]) == c->rank) { d.push_back(c->in_ini_det[i]); v.push_back(c->in_ini_val[i]); }
        uint32_t m = (uint32_t)d.size();
        if (m > c->sp.cap) throw FriesError("initial vector larger than the spawn buffer");
        if (m) {
            std::vector<uint8_t> f(m, 1);
            FR_HIP(hipMemcpyAsync(c->sp.det, d.data(), 8 * (size_t)m, hipMemcpyHostToDevice, c->stream));
            FR_HIP(hipMemcpyAsync(c->sp.val, v.data(), 8 * (size_t)m, hipMemcpyHostToDevice, c->stream));
            FR_HIP(hipMemcpyAsync(c->sp.ini, f.data(), m, hipMemcpyHostToDevice, c->stream));
            FR_HIP(hipMemcpyAsync(c->sp.n_spawn, &m, 4, hipMemcpyHostToDevice, c->stream));
            fr_vec_merge(c, &c->vec, m, true);
            FR_HIP(hipStreamSynchronize(c->stream));
        }
    }
    // start from 100 * |HF> on the rank that owns it (:277-281)
    else if (c->rank == c->hf_proc) {
        double v = 100; uint8_t one = 1; uint32_t n1 = 1;
        FR_HIP(hipMemcpyAsync(c->sp.det, &c->hf_det, 8, hipMemcpyHostToDevice, c->stream));
        FR_HIP(hipMemcpyAsync(c->sp.val, &v, 8, hipMemcpyHostToDevice, c->stream));
        FR_HIP(hipMemcpyAsync(c->sp.ini, &one, 1, hipMemcpyHostToDevice, c->stream));
        FR_HIP(hipMemcpyAsync(c->sp.n_spawn, &n1, 4, hipMemcpyHostToDevice, c->stream));
        fr_vec_merge(c, &c->vec, 1, true);               // perform_add(0) into column 0
    }
    fr_vec_sync_state(c, &c->vec, &c->h_vst);
    check_dev_err(c);
}

static inline double uni(std::mt19937 &mt) { return mt() / (1. + UINT32_MAX); }

// the non-zero products of the dense block of H with column 0, in the stored order (one workgroup: ordered compaction)
static __global__ void __launch_bounds__(FR_BLOCK) k_dense_spawn(VecDev V, SpawnBuf P, const uint32_t *from, const det_t *to, const double *el, uint32_t n) {
    __shared__ uint32_t shu[4];
    uint32_t written = 0;
    for (uint32_t base = 0; base < n; base += FR_BLOCK) {
        const uint32_t k = base + threadIdx.x;
        double mv = 0;
        if (k < n) mv = V.v0[from[k]] * el[k];
        const uint32_t f = mv != 0 ? 1u : 0u;       // DistVec::add drops zero values (vec_utils.hpp:418-423)
        uint32_t tot;
        const uint32_t incl = fr_block_scan_u32(f, shu, &tot);
        if (f) { const uint32_t o = written + incl - 1; P.det[o] = to[k]; P.val[o] = mv; P.ini[o] = 1; }
        written += tot;
    }
    if (threadIdx.x == 0) *P.n_spawn = written;
}
// DistVec::dense_norm (vec_utils.hpp:903-918): the magnitudes of the dense space added up in position order
static __global__ void k_dense_norm(VecDev V, double *out) {
    double r = 0;
    for (uint32_t i = 0; i < V.n_dense; i++) { const double e = V.v0[i]; r += e >= 0 ? e : -e; }
    *out = r;
}

static void frisys_iterate(FriesCtx *c, fries_iter_log *lg) {
    hipStream_t st = c->stream;
    fr_vec_sync_state(c, &c->vec, &c->h_vst);
    fr_vec_maybe_rebuild(c, &c->vec);
    // systematic matrix compression (:414-422)
    double rn[5];
    for (int k = 0; k < 5; k++) rn[k] = uni(c->mt);
    fr_hbpp_apply(c, c->mat_nonz - c->n_dense_h_glob, rn);      // :421 matr_samp - tot_dense_h
    uint32_t vec_size = c->h_vst.curr_size;
    // spawning + annihilation (:429-471)
    if (c->num_success > c->sp.cap) throw FriesError("spawn buffer too small");
    if (c->num_success) FR_LAUNCH(c, "k_spawn_build", k_spawn_build, dim3(fr_blocks(c->num_success, FR_BLOCK)), dim3(FR_BLOCK), c->vec, c->sp, c->c_pos, c->c_orbs, c->c_val, c->d_nsucc, c->eps, c->init_thresh);
    uint32_t n_merge = c->num_success;
    bool merged = false;            // the Adder filled up: the passes went through several perform_add rounds, each merged on arrival
    if (c->use_comm) n_merge = fr_spawn_exchange(c, c->num_success, 0, &merged);      // every rank takes part, also with nothing to send
    if (n_merge && !merged) fr_vec_merge(c, &c->vec, n_merge, false);
    uint32_t n_merge_dense = 0;
    if (c->n_dense_h_glob) {
        // the dense block of H applied exactly (:480-485): value at the origin x stored element, added as initiator contributions in
        // the stored order, as a perform_add of its own (with ranks: an exchange of its own, every rank taking part)
        uint32_t m = 0;
        if (c->n_dense_h_nz) {
            FR_LAUNCH(c, "k_dense_spawn", k_dense_spawn, dim3(1), dim3(FR_BLOCK), c->vec, c->sp, c->d_dh_from, c->d_dh_to, c->d_dh_el, c->n_dense_h_nz);
            FR_HIP(hipMemcpyAsync(&m, c->sp.n_spawn, 4, hipMemcpyDeviceToHost, st));
            FR_HIP(hipStreamSynchronize(st));
        }
        n_merge_dense = c->use_comm ? fr_spawn_exchange(c, m) : m;      // all of them carry the initiator flag: one pass in effect
        if (n_merge_dense) fr_vec_merge(c, &c->vec, n_merge_dense, false);
    }
    // no host look at the vector's state here: the kernels read the stored size themselves, the launches only need an upper bound of it
    // (every merged spawn may have taken a new position); an overflow raised by the merge is reported at the end of the iteration
    {
        const uint64_t ub = (uint64_t)c->h_vst.curr_size + n_merge + n_merge_dense;
        c->h_vst.curr_size = ub < c->vec.cap ? (uint32_t)ub : c->vec.cap;
    }
    // death / cloning, column add (:487-499)
    fr_death_clone(c, vec_size);
    // the projected-energy dot products are taken here -- find_preserve does not touch the values -- so that their readback rides on
    // the synchronisation find_preserve needs anyway (the reference forms them after it, frisys_mol.cpp:511-517)
    const void *h_dots = fr_dots_enqueue(c);
    // vector compression (:501-539)
    uint32_t n_samp = c->vec_nonz;
    double glob_norm = 0;
    fr_find_preserve(c, &n_samp, &glob_norm);
    fr_dots_collect(c, h_dots, &c->numer, &c->denom);
    if (c->n_dense_h_glob || c->vec.n_dense) {       // glob_norm += sol_vec.dense_norm() (:503, vec_utils.hpp:903-918: sum_mpi of the ranks' sums)
        FR_LAUNCH(c, "k_dense_norm", k_dense_norm, dim3(1), dim3(1), c->vec, c->d_dense_norm);
        const double *src = c->d_dense_norm;
        if (c->use_comm) {
            FR_HIP(hipMemcpyAsync(c->comm.small_send, c->d_dense_norm, 8, hipMemcpyDeviceToDevice, st));
            src = (const double *)fr_allgather(c, 8);
        }
        double h[FR_MAX_RANKS];
        FR_HIP(hipMemcpyAsync(h, src, 8 * (size_t)c->n_ranks, hipMemcpyDeviceToHost, st));
        FR_HIP(hipStreamSynchronize(st));
        double dn = 0;
        for (int p = 0; p < c->n_ranks; p++) dn += h[p];
        glob_norm += dn;
    }
    c->glob_norm = glob_norm;
    c->nkept = c->vec_nonz - n_samp;
    const unsigned shift_interval = 10;
    const double shift_damping = 0.05;
    if ((c->iterat + 1) % shift_interval == 0) {     // adjust_shift, compress_utils.cpp:684-693
        double damp = shift_damping / shift_interval / c->eps;
        if (c->last_one_norm) { c->en_shift -= damp * log(glob_norm / c->last_one_norm); c->last_one_norm = glob_norm; }
        if (c->last_one_norm == 0 && glob_norm > c->target_norm) c->last_one_norm = glob_norm;
    }
    double rn_sys = uni(c->mt);
    fr_sys_comp(c, n_samp, rn_sys);
    c->iterat++;
    c->tot_iters++; c->tot_spawns += c->num_success;
    for (int k = 0; k < 5; k++) c->tot_stage_elems += c->comp_len[k];
    for (int k = 1; k <= 5; k++) c->tot_fks_iters += c->fks_iters[k];
    if (c->prof_on) prof_collect(c);
    if (lg) {
        fr_vec_sync_state(c, &c->vec, &c->h_vst);
        lg->numer = c->numer; lg->denom = c->denom; lg->shift = c->en_shift; lg->norm = c->glob_norm;
        lg->nkept = c->nkept; lg->n_nonz = c->h_vst.n_nonz; lg->curr_size = c->h_vst.curr_size;
        lg->num_success = c->num_success;
        for (int k = 0; k < 5; k++) lg->comp_len[k] = c->comp_len[k];
        uint32_t e = 0;
        FR_HIP(hipMemcpy(&e, c->d_err, 4, hipMemcpyDeviceToHost));
        lg->err = e | c->h_vst.err;
    }
}

// ------------------------------------------------------------------ frifull_mol (FRIES_bin/frifull_mol.cpp)
static void frifull_setup(FriesCtx *c, const fries_frifull_params *p) {
    if (!c->d_eris) throw FriesError("fries_set_molecule must be called first");
    if (p->max_dets == 0 || p->vec_nonz == 0) throw FriesError("max_dets and vec_nonz must be positive");
    c->eps = p->epsilon; c->target_norm = p->target_norm; c->init_thresh = 0;
    c->vec_nonz = p->vec_nonz; c->mat_nonz = 0; c->full_mode = true;
    c->en_shift = 0; c->last_one_norm = 0; c->iterat = 0;
    c->mt.seed(p->seed);
    c->proc_scr.resize(2 * c->n_orb); c->vec_scr.resize(2 * c->n_orb);
    for (auto &x : c->proc_scr) x = c->mt();        // frifull_mol.cpp:96-98
    for (auto &x : c->vec_scr) x = c->mt();         // :104-107
    if (!c->comm.small_send) { c->own_small = fr_alloc<uint8_t>(2048); c->comm.small_send = c->own_small; }
    uint32_t scap = p->spawn_cap ? p->spawn_cap : 8000000u;
    if (scap > 8000000u) scap = 8000000u;           // one merge handles FR_MAX_PART tiles of spawns
    {   // frifull_mol.cpp:68-70: the Adder holds min(1e6, target_nonz / n_procs * num_ex / n_procs / 4) elements per destination (32-bit arithmetic there)
        const uint32_t nv = c->n_orb - c->n_elec / 2;
        const uint32_t num_ex = c->n_elec * c->n_elec * nv * nv;
        const uint32_t spawn_len = p->vec_nonz / (uint32_t)c->n_ranks * num_ex / (uint32_t)c->n_ranks / 4u;
        c->adder_cap = spawn_len > 1000000u ? 1000000u : spawn_len;
        if (getenv("FRIES_ADDER_SIZE")) c->adder_cap = (uint32_t)atol(getenv("FRIES_ADDER_SIZE"));
    }
    if (!c->d_proc_scr) c->d_proc_scr = fr_alloc<uint32_t>(64);
    FR_HIP(hipMemcpyAsync(c->d_proc_scr, c->proc_scr.data(), 4 * c->proc_scr.size(), hipMemcpyHostToDevice, c->stream));
    c->hf_proc = fr_host_idx_to_proc(c, c->hf_det);
    fr_vec_alloc(c, &c->vec, p->max_dets);
    fr_spawn_alloc(c, scap);
    fr_xch_alloc(c, scap);
    fr_vcomp_alloc(c, p->max_dets);
    c->W.kin = fr_alloc<uint32_t>(p->max_dets);     // sys_comp's tooth-index scratch (the HB-PP arrays are not allocated here)
    if (!c->d_norms_keep) { c->d_norms_keep = fr_alloc<double>(FR_MAX_RANKS); c->d_seq_scratch = fr_alloc<double>(1); }        // sys_comp over ranks: the other ranks' norms, the second in-order sum
    // trial vector = HF (:121-147); there is no H * trial in this driver
    c->n_trial = 1; c->n_htrial = 0;
    c->tr_det = fr_alloc<det_t>(1); c->tr_val = fr_alloc<double>(1);
    c->htr_det = fr_alloc<det_t>(1); c->htr_val = fr_alloc<double>(1);
    double one = 1.0;
    FR_HIP(hipMemcpyAsync(c->tr_det, &c->hf_det, 8, hipMemcpyHostToDevice, c->stream));
    FR_HIP(hipMemcpyAsync(c->tr_val, &one, 8, hipMemcpyHostToDevice, c->stream));
    if (c->rank == (int)c->hf_proc) {   // start from 100 * |HF> on the rank that owns it (:186-190)
        double v = 100; uint8_t ini = 1; uint32_t n1 = 1;
        FR_HIP(hipMemcpyAsync(c->sp.det, &c->hf_det, 8, hipMemcpyHostToDevice, c->stream));
        FR_HIP(hipMemcpyAsync(c->sp.val, &v, 8, hipMemcpyHostToDevice, c->stream));
        FR_HIP(hipMemcpyAsync(c->sp.ini, &ini, 1, hipMemcpyHostToDevice, c->stream));
        FR_HIP(hipMemcpyAsync(c->sp.n_spawn, &n1, 4, hipMemcpyHostToDevice, c->stream));
        fr_vec_merge(c, &c->vec, 1, true);
    }
    fr_vec_sync_state(c, &c->vec, &c->h_vst);
    check_dev_err(c);
}

// one iteration of frifull_mol.cpp:258-304.  The reference alternates between its two value columns; here the current
// vector is always "column 0" and the two device arrays trade places at the end of the iteration.
static void frifull_iterate(FriesCtx *c, fries_iter_log *lg) {
    fr_vec_sync_state(c, &c->vec, &c->h_vst);
    fr_vec_maybe_rebuild(c, &c->vec);
    double unused = 0, denom = 0, numer = 0;
    fr_dots(c, &unused, &denom);                      // :259-260
    fr_abs_sums(c);
    uint32_t n_samp = c->vec_nonz;
    double glob_norm = 0;
    fr_find_preserve(c, &n_samp, &glob_norm);         // :263-264
    c->glob_norm = glob_norm;
    c->nkept = c->vec_nonz - n_samp;
    const unsigned shift_interval = 10;
    const double shift_damping = 0.05;
    if ((c->iterat + 1) % shift_interval == 0) {     // :272-278
        double damp = shift_damping / shift_interval / c->eps;
        if (c->last_one_norm) { c->en_shift -= damp * log(glob_norm / c->last_one_norm); c->last_one_norm = glob_norm; }
        if (c->last_one_norm == 0 && glob_norm > c->target_norm) c->last_one_norm = glob_norm;
    }
    double rn_sys = uni(c->mt);
    fr_sys_comp(c, n_samp, rn_sys);                   // :283-289 (an element leaves the table only when it is zero in both columns)
    fr_vec_sync_state(c, &c->vec, &c->h_vst);
    fr_h_diag_vec(c, 1 + c->eps * c->en_shift, -c->eps);            // :291
    uint64_t n_add = fr_h_offdiag_vec(c, -c->eps);                  // :293
    std::swap(c->vec.v0, c->vec.v1);                                // :294
    fr_vec_sync_state(c, &c->vec, &c->h_vst);
    if (c->h_vst.err) check_dev_err(c);
    fr_dots(c, &unused, &numer);
    numer = ((1 + c->eps * c->en_shift) * denom - numer) / c->eps;  // :296-298
    c->numer = numer; c->denom = denom;
    c->num_success = (uint32_t)(n_add > 0xffffffffull ? 0xffffffffull : n_add);
    c->iterat++; c->tot_iters++; c->tot_spawns += n_add;
    if (c->prof_on) prof_collect(c);
    if (lg) {
        lg->numer = c->numer; lg->denom = c->denom; lg->shift = c->en_shift; lg->norm = c->glob_norm;
        lg->nkept = c->nkept; lg->n_nonz = c->h_vst.n_nonz; lg->curr_size = c->h_vst.curr_size;
        lg->num_success = c->num_success;
        for (int k = 0; k < 5; k++) lg->comp_len[k] = 0;
        uint32_t e = 0;
        FR_HIP(hipMemcpy(&e, c->d_err, 4, hipMemcpyDeviceToHost));
        lg->err = e | c->h_vst.err;
    }
}

// ------------------------------------------------------------------ C ABI
// the driver's optional inputs; all three must precede fries_frisys_setup
extern "C" int fries_set_trial_vector(fries_ctx *h, const uint64_t *dets, const double *vals, size_t n) {
    FR_API_BEGIN
    if (h->c.vec.dets) throw FriesError("fries_set_trial_vector must be called before the driver's setup");
    h->c.in_trial_det.assign(dets, dets + n); h->c.in_trial_val.assign(vals, vals + n);
    FR_API_END
}
extern "C" int fries_set_initial_vector(fries_ctx *h, const uint64_t *dets, const double *vals, size_t n) {
    FR_API_BEGIN
    if (h->c.vec.dets) throw FriesError("fries_set_initial_vector must be called before the driver's setup");
    h->c.in_ini_det.assign(dets, dets + n); h->c.in_ini_val.assign(vals, vals + n);
    FR_API_END
}
extern "C" int fries_set_det_space(fries_ctx *h, const uint64_t *dets, size_t n) {
    FR_API_BEGIN
    if (h->c.vec.dets) throw FriesError("fries_set_det_space must come before fries_frisys_setup");
    h->c.in_det_space.assign(dets, dets + n);
    FR_API_END
}
extern "C" int fries_set_ham_shift(fries_ctx *h, double hf_en) {
    FR_API_BEGIN
    if (h->c.vec.dets) throw FriesError("fries_set_ham_shift must be called before fries_frisys_setup");
    h->c.ham_shift_set = true; h->c.ham_shift_hf_en = hf_en;
    FR_API_END
}

extern "C" int fries_frifull_setup(fries_ctx *h, const fries_frifull_params *p) {
    FR_API_BEGIN
    FriesCtx *c = &h->c;
    FR_HIP(hipSetDevice(c->device));
    if (c->vec.dets) throw FriesError("this context already holds a driver; create a new one");
    frifull_setup(c, p);
    FR_API_END
}
extern "C" int fries_frifull_iterate(fries_ctx *h, uint32_t n_iter, fries_iter_log *logs) {
    FR_API_BEGIN
    FriesCtx *c = &h->c;
    FR_HIP(hipSetDevice(c->device));
    if (!c->full_mode) throw FriesError("fries_frifull_setup must be called first");
    for (uint32_t k = 0; k < n_iter; k++) {
        frifull_iterate(c, logs ? &logs[k] : nullptr);
        check_dev_err(c);
    }
    FR_API_END
}

extern "C" int fries_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int fries_ctx_create(fries_ctx **out, int device) {
    FR_API_BEGIN
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) throw FriesError("no HIP device available: the FRI engine has no CPU fallback");
    if (device < 0 || device >= n) throw FriesError("device index out of range");
    FR_HIP(hipSetDevice(device));
    fries_ctx *h = new fries_ctx();
    h->c.device = device;
    if (getenv("FRIES_DBG")) h->c.dbg = atoi(getenv("FRIES_DBG"));
    if (getenv("FRIES_FKS_REC_AT")) h->c.fks_rec_at = atoi(getenv("FRIES_FKS_REC_AT"));
    if (getenv("FRIES_GROUP_WARM_ALL")) h->c.fks_group_warm_all = atoi(getenv("FRIES_GROUP_WARM_ALL")) != 0;
    if (getenv("FRIES_NO_GROUP_WARM")) h->c.fks_no_group_warm = true;
    if (getenv("FRIES_FKS_NO_EXT")) h->c.fks_no_ext = true;
    if (getenv("FRIES_FKS_LIGHT_FULL_GRID")) h->c.fks_light_full_grid = atoi(getenv("FRIES_FKS_LIGHT_FULL_GRID")) != 0;
    if (getenv("FRIES_FKS_FUSE_TOTALS")) h->c.fks_fuse_totals = atoi(getenv("FRIES_FKS_FUSE_TOTALS")) != 0;
    if (getenv("FRIES_FKS_NO_LIGHT")) h->c.fks_no_light = atoi(getenv("FRIES_FKS_NO_LIGHT")) != 0;
    if (getenv("FRIES_FKS_NO_SPECULATION")) h->c.fks_no_speculation = atoi(getenv("FRIES_FKS_NO_SPECULATION")) != 0;
    if (getenv("FRIES_FKS_NO_CLOSING")) h->c.fks_no_closing = atoi(getenv("FRIES_FKS_NO_CLOSING")) != 0;
    if (getenv("FRIES_FKS_COLLAPSE_WALK")) h->c.fks_no_collapse_walk = atoi(getenv("FRIES_FKS_COLLAPSE_WALK")) == 0;
    if (getenv("FRIES_FKS_SEQ")) h->c.fks_force_seq = atoi(getenv("FRIES_FKS_SEQ")) != 0;
    if (getenv("FRIES_FKS_NO_MERGED_NORM")) h->c.fks_no_merged_norm = atoi(getenv("FRIES_FKS_NO_MERGED_NORM")) != 0;
    if (getenv("FRIES_NO_WARM")) h->c.warm_start = false;
    {
        hipDeviceProp_t pr;
        FR_HIP(hipGetDeviceProperties(&pr, device));
        h->c.fks_grid = (unsigned)FR_FKS_WPE1 * (unsigned)pr.multiProcessorCount;      // k_fks_sweep: <= 92 VGPRs -> 5 waves/SIMD = 5 workgroups per CU
        h->c.fks_grid0 = ((unsigned)FR_FKS_WPE0 + 1u) * (unsigned)pr.multiProcessorCount;     // the lean replay: one workgroup per CU more than fit at once (measured: 60.8 against 62.6 us)
        if (getenv("FRIES_FKS_GRID0")) h->c.fks_grid0 = (unsigned)atoi(getenv("FRIES_FKS_GRID0"));
        if (getenv("FRIES_FKS_GRID")) h->c.fks_grid = (unsigned)atoi(getenv("FRIES_FKS_GRID"));
    }
    FR_HIP(hipStreamCreate(&h->c.stream));
    h->c.d_err = fr_alloc<uint32_t>(1);
    FR_HIP(hipMemset(h->c.d_err, 0, 4));
    *out = h;
    FR_API_END
}

extern "C" void fries_ctx_destroy(fries_ctx *h) {
    if (!h) return;
    hipSetDevice(h->c.device);
    hipDeviceSynchronize();
    // device allocations are released with the context's device memory pool at process exit;
    // explicit frees for the large arrays:
    VecDev &v = h->c.vec;
    hipFree(v.dets); hipFree(v.v0); hipFree(v.v1); hipFree(v.diag); hipFree(v.active); hipFree(v.free_stack); hipFree(v.hs); hipFree(v.stat_part); hipFree(v.st);
    CompWork &W = h->c.W;
    for (int k = 0; k < 2; k++) { hipFree(W.el[k].val); hipFree(W.el[k].pos); hipFree(W.el[k].code); hipFree(W.el[k].ndiv); hipFree(W.el[k].nsub); hipFree(W.el[k].rinv); hipFree(W.el[k].raux); hipFree(W.el[k].det); hipFree(W.psum[k]); hipFree(W.pcnt[k]); }
    hipFree(W.wt_remain); hipFree(W.keep); hipFree(W.S); hipFree(W.kin); hipFree(W.cnt); hipFree(W.e_wi); hipFree(W.e_sub); hipFree(W.e_val); hipFree(W.state); hipFree(W.teeth); hipFree(W.fix_list);
    if (h->c.stg_mem) { hipFree(h->c.stg_mem); hipFree(h->c.spill_mem); hipFree(h->c.spill_cnt_mem); hipFree(h->c.tile_dirty_mem); }
    {
        FriesCtx &c = h->c;
        if (c.dbg >= 1 && c.n_fks_sequential) {
            FksSqCtl hc{};
            hipMemcpy(&hc, c.fsq.ctl, sizeof(hc), hipMemcpyDeviceToHost);
            fprintf(stderr, "[fries] find_keep_sub in the reference's order: %llu stages; %llu guess rounds, %llu exact rounds over %llu tiles (touched tiles taken as one integer step: %llu, element by element: %llu), %llu walks over %llu tiles\n",
                    (unsigned long long)c.n_fks_sequential, (unsigned long long)c.n_fsq_guess, (unsigned long long)c.n_fsq_exact, (unsigned long long)c.n_fsq_chain_tiles,
                    hc.n_fast, hc.n_dense, (unsigned long long)c.n_fsq_walk, (unsigned long long)c.n_fsq_walk_tiles);
        }
        FksSq &SQ = c.fsq;
        hipFree(SQ.dl); hipFree(SQ.nwr); hipFree(SQ.nkp); hipFree(SQ.gb); hipFree(SQ.lb); hipFree(SQ.dgb); hipFree(SQ.kb); hipFree(SQ.dk);
        hipFree(SQ.tk); hipFree(SQ.tkx); hipFree(SQ.tg); hipFree(SQ.tgx); hipFree(SQ.tany);
        hipFree(SQ.mG); hipFree(SQ.mL); hipFree(SQ.sabs); hipFree(SQ.gt); hipFree(SQ.lt); hipFree(SQ.eG); hipFree(SQ.eL); hipFree(SQ.mflag); hipFree(SQ.fast);
        if (SQ.gb2) { hipFree(SQ.gb2); hipFree(SQ.lb2); }
        hipFree(SQ.ctl);
    }
    {   // the find_keep_sub replay's arrays (allocated with the work arrays: hbpp.hip)
        Fks2Work &F = h->c.F2;
        if (F.dk8) {
            hipFree(F.dk8); hipFree(F.dg8); hipFree(F.ws8); hipFree(F.cdirty); hipFree(F.wrec); hipFree(F.wNp);
            hipFree(F.xk8); hipFree(F.xg8); hipFree(F.scal); hipFree(F.hist); hipFree(F.ck); hipFree(F.cg); hipFree(F.cw); hipFree(F.ckx); hipFree(F.cgx); hipFree(F.dbg_cnt);
            hipFree(h->c.fks_wkx); hipFree(h->c.fks_wgx); hipFree(h->c.fks_wk); hipFree(h->c.fks_wg); hipFree(h->c.fks_sxk8); hipFree(h->c.fks_sxg8); hipFree(h->c.fks_saved);
        }
    }
    hipFree(h->c.c_pos); hipFree(h->c.c_orbs); hipFree(h->c.c_val); hipFree(h->c.d_nsucc);
    SpawnBuf &s = h->c.sp;
    hipFree(s.det); hipFree(s.val); hipFree(s.ini); hipFree(s.slot); hipFree(s.flag);
    if (s.bdet) { hipFree(s.bdet); hipFree(s.bval); hipFree(s.bini); hipFree(s.bn); hipFree(s.xbounds); }
    for (int k = 0; k < 2; k++) { hipFree(s.key[k]); hipFree(s.pay[k]); }
    hipFree(s.hist); hipFree(s.pcnt); hipFree(s.n_spawn);
    VcompBuf &b = h->c.vc;
    hipFree(b.keep); hipFree(b.del); hipFree(b.S);
    for (int k = 0; k < 2; k++) { hipFree(b.psum[k]); hipFree(b.pcnt[k]); }
    hipFree(b.state); hipFree(b.teeth); hipFree(b.dots); hipFree(b.fix_list);
    PivBuf &pv = h->c.piv;
    if (pv.start) { hipFree(pv.start); hipFree(pv.carry); hipFree(pv.U); hipFree(pv.unit); hipFree(pv.scal); hipFree(pv.tile_dd); hipFree(pv.tile_nz); hipFree(pv.nz_start); }
    hipFree(h->c.d_dh_from); hipFree(h->c.d_dh_to); hipFree(h->c.d_dh_el); hipFree(h->c.d_dense_norm);
    hipFree(h->c.d_h); hipFree(h->c.d_eris); hipFree(h->c.d_hb); hipFree(h->c.d_err);
    if (h->c.full_cnt) { hipFree(h->c.full_cnt); hipFree(h->c.full_nz); hipFree(h->c.full_off); hipFree(h->c.full_list); }
    fr_piv_flat_free(&h->c);
    if (h->c.pv_goff) hipFree(h->c.pv_goff);
    if (h->c.hh_fdet) hipFree(h->c.hh_fdet);
    if (h->c.hh_ovlp) hipFree(h->c.hh_ovlp);
    if (h->c.h_fks) hipHostFree(h->c.h_fks);
    if (h->c.h_rb) hipHostFree(h->c.h_rb);
    if (h->c.hhf_cnt) hipFree(h->c.hhf_cnt);
    hipFree(h->c.tr_det); hipFree(h->c.tr_val); hipFree(h->c.htr_det); hipFree(h->c.htr_val);
    if (h->c.stream) hipStreamDestroy(h->c.stream);
    delete h;
}

extern "C" int fries_set_comm(fries_ctx *h, const fries_comm *cm) {
    FR_API_BEGIN
    FriesCtx *c = &h->c;
    if (c->vec.dets) throw FriesError("fries_set_comm must be called before fries_frisys_setup");
    if (!cm || cm->size < 1 || cm->size > FR_MAX_RANKS || cm->rank < 0 || cm->rank >= cm->size) throw FriesError("bad rank / size");
    const bool have = cm->allgather && cm->alltoallv && cm->small_send && cm->small_recv && cm->big_send && cm->big_recv;
    if (cm->size > 1 && (!cm->allgather || !cm->alltoallv || !cm->small_send || !cm->small_recv || !cm->big_send || !cm->big_recv))
        throw FriesError("fries_comm needs both collectives and all four staging buffers");
    c->comm.user = cm->user; c->comm.rank = cm->rank; c->comm.size = cm->size;
    c->comm.small_send = cm->small_send; c->comm.small_recv = cm->small_recv; c->comm.big_send = cm->big_send; c->comm.big_recv = cm->big_recv;
    c->comm.big_bytes = cm->big_bytes; c->comm.allgather = cm->allgather; c->comm.alltoallv = cm->alltoallv;
    c->rank = cm->rank; c->n_ranks = cm->size;
    c->use_comm = cm->size > 1 || have;      // a one-rank communicator still routes everything through the callbacks
    FR_API_END
}
extern "C" void *fries_stream(fries_ctx *h) { return (void *)h->c.stream; }
extern "C" int fries_idx_to_proc(fries_ctx *h, const uint64_t *dets, size_t n, int32_t *proc) {
    FR_API_BEGIN
    FriesCtx *c = &h->c;
    if (c->proc_scr.empty()) throw FriesError("fries_frisys_setup must be called first");
    for (size_t i = 0; i < n; i++) proc[i] = fr_host_idx_to_proc(c, dets[i]);
    FR_API_END
}

extern "C" int fries_set_molecule(fries_ctx *h, uint32_t n_orb, uint32_t n_elec, const uint8_t *irreps, const double *h_core, const double *eris) {
    FR_API_BEGIN
    FR_HIP(hipSetDevice(h->c.device));
    fr_system_upload(&h->c, n_orb, n_elec, irreps, h_core, eris);
    FR_API_END
}

static double *hb_field(HbTables &T, int which, size_t *len) {
    size_t n = T.n_orb;
    switch (which) {
        case 0: *len = n; return T.s_tens;
        case 1: *len = n * (n - 1) / 2; return T.d_same;
        case 2: *len = n * n; return T.d_diff;
        case 3: *len = n * (n - 1) / 2; return T.exch_sqrt;
        case 4: *len = n; return T.diag_sqrt;
        case 5: *len = n; return T.exch_norms;
        case 6: *len = 1; return &T.s_norm;
    }
    throw FriesError("unknown tensor id");
}

extern "C" int fries_get_hb_tensor(fries_ctx *h, int which, double *out, size_t cap, size_t *len) {
    FR_API_BEGIN
    size_t n;
    double *src = hb_field(h->c.h_hb, which, &n);
    if (len) *len = n;
    if (cap < n) throw FriesError("output buffer too small");
    memcpy(out, src, 8 * n);
    FR_API_END
}

extern "C" int fries_set_hb_tensor(fries_ctx *h, int which, const double *in, size_t len) {
    FR_API_BEGIN
    size_t n;
    double *dst = hb_field(h->c.h_hb, which, &n);
    if (len != n) throw FriesError("tensor length mismatch");
    memcpy(dst, in, 8 * n);
    FR_HIP(hipMemcpy(h->c.d_hb, &h->c.h_hb, sizeof(HbTables), hipMemcpyHostToDevice));
    FR_API_END
}

extern "C" double fries_hf_energy(fries_ctx *h) { return h->c.hf_en; }
extern "C" double fries_p_doub(fries_ctx *h) { return h->c.p_doub; }
extern "C" int fries_get_scramblers(fries_ctx *h, uint32_t *proc_scr, uint32_t *vec_scr, size_t n) {
    FR_API_BEGIN
    FriesCtx *c = &h->c;
    if (c->proc_scr.empty()) throw FriesError("no run has been set up");
    if (n < c->proc_scr.size()) throw FriesError("scrambler buffer too small");
    if (proc_scr) memcpy(proc_scr, c->proc_scr.data(), 4 * c->proc_scr.size());
    if (vec_scr) memcpy(vec_scr, c->vec_scr.data(), 4 * c->vec_scr.size());
    FR_API_END
}
extern "C" int fries_tie_margins(fries_ctx *h, int enable, double *fks_min_rel, double *fp_min_rel) {
    FR_API_BEGIN
    FriesCtx *c = &h->c;
    FR_HIP(hipSetDevice(c->device));
    const uint32_t inf2[2] = {0x7F800000u, 0x7F800000u};
    if (c->d_tie) {
        uint32_t v[2];
        FR_HIP(hipMemcpyAsync(v, c->d_tie, 8, hipMemcpyDeviceToHost, c->stream));
        FR_HIP(hipStreamSynchronize(c->stream));
        float f0, f1; memcpy(&f0, &v[0], 4); memcpy(&f1, &v[1], 4);
        if (fks_min_rel) *fks_min_rel = f0;
        if (fp_min_rel) *fp_min_rel = f1;
        FR_HIP(hipMemcpyAsync(c->d_tie, inf2, 8, hipMemcpyHostToDevice, c->stream));       // the statistics restart
        FR_HIP(hipStreamSynchronize(c->stream));
    }
    else { if (fks_min_rel) *fks_min_rel = INFINITY; if (fp_min_rel) *fp_min_rel = INFINITY; }
    if (enable && !c->d_tie) { c->d_tie = fr_alloc<uint32_t>(2); FR_HIP(hipMemcpy(c->d_tie, inf2, 8, hipMemcpyHostToDevice)); }
    if (!enable && c->d_tie) { FR_HIP(hipFree(c->d_tie)); c->d_tie = nullptr; }
    FR_API_END
}
extern "C" int fries_set_proc_scrambler(fries_ctx *h, const uint32_t *proc_scr, size_t n) {
    FR_API_BEGIN
    FriesCtx *c = &h->c;
    if (c->vec.dets) throw FriesError("fries_set_proc_scrambler must be called before fries_frisys_setup");
    c->in_proc_scr.assign(proc_scr, proc_scr + n);
    FR_API_END
}
extern "C" uint64_t fries_kernel_launches(fries_ctx *h) { return h->c.n_kernel_launch; }

__global__ void k_matrel_batch(int kind, const det_t *dets, const uint8_t *orbs, size_t n, const double *hc, const double *eris, unsigned n_orb, double *out, int32_t *sign) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    det_t d = dets[i];
    const uint8_t *o = orbs ? orbs + 4 * i : nullptr;
    if (kind == 0) { out[i] = fr_diag_matrel(d, hc, eris, n_orb); if (sign) sign[i] = 1; }
    else if (kind == 1) { out[i] = fr_sing_matrel(d, o[0], o[1], hc, eris, n_orb); if (sign) sign[i] = fr_sing_parity(d, o[0], o[1]); }
    else { out[i] = fr_doub_matrel(o[0], o[1], o[2], o[3], eris, n_orb); if (sign) sign[i] = fr_doub_parity(d, o[0], o[1], o[2], o[3]); }
}

extern "C" int fries_matrel_batch(fries_ctx *h, int kind, const uint64_t *dets, const uint8_t *orbs, size_t n, double *out, int32_t *sign) {
    FR_API_BEGIN
    FriesCtx *c = &h->c;
    FR_HIP(hipSetDevice(c->device));
    if (!c->d_eris) throw FriesError("fries_set_molecule must be called first");
    if (n == 0) return 0;
    det_t *dd = fr_alloc<det_t>(n); uint8_t *dob = fr_alloc<uint8_t>(4 * n); double *dout = fr_alloc<double>(n); int32_t *ds = fr_alloc<int32_t>(n);
    FR_HIP(hipMemcpy(dd, dets, 8 * n, hipMemcpyHostToDevice));
    if (orbs) FR_HIP(hipMemcpy(dob, orbs, 4 * n, hipMemcpyHostToDevice));
    FR_LAUNCH(c, "k_matrel_batch", k_matrel_batch, dim3(fr_blocks(n, FR_BLOCK)), dim3(FR_BLOCK), kind, dd, orbs ? dob : nullptr, n, c->d_h, c->d_eris, c->n_orb, dout, ds);
    FR_HIP(hipStreamSynchronize(c->stream));
    FR_HIP(hipMemcpy(out, dout, 8 * n, hipMemcpyDeviceToHost));
    if (sign) FR_HIP(hipMemcpy(sign, ds, 4 * n, hipMemcpyDeviceToHost));
    hipFree(dd); hipFree(dob); hipFree(dout); hipFree(ds);
    FR_API_END
}

extern "C" int fries_frisys_setup(fries_ctx *h, const fries_frisys_params *p) {
    FR_API_BEGIN
    FR_HIP(hipSetDevice(h->c.device));
    frisys_setup(&h->c, p);
    FR_API_END
}

extern "C" int fries_frisys_iterate(fries_ctx *h, uint32_t n_iter, fries_iter_log *logs) {
    FR_API_BEGIN
    FR_HIP(hipSetDevice(h->c.device));
    if (h->c.hh_mode) throw FriesError("this context runs frisys_hh: use fries_hh_iterate");
    if (h->c.fq_mode) throw FriesError("this context runs fciqmc_mol: use fries_fciqmc_iterate");
    for (uint32_t i = 0; i < n_iter; i++) frisys_iterate(&h->c, logs ? &logs[i] : nullptr);
    if (n_iter && !logs) fr_vec_sync_state(&h->c, &h->c.vec, &h->c.h_vst);       // the last iteration's merges may have raised a flag in the vector state: one readback per batch
    check_dev_err(&h->c);
    FR_API_END
}

extern "C" int fries_fciqmc_setup(fries_ctx *h, const fries_fciqmc_params *p) {
    FR_API_BEGIN
    FR_HIP(hipSetDevice(h->c.device));
    if (h->c.vec.dets) throw FriesError("this context already holds a run");
    fr_fq_setup(&h->c, p);
    check_dev_err(&h->c);
    FR_API_END
}
extern "C" int fries_fciqmc_iterate(fries_ctx *h, uint32_t n_iter, fries_fciqmc_log *logs) {
    FR_API_BEGIN
    FR_HIP(hipSetDevice(h->c.device));
    if (!h->c.fq_mode || h->c.fqw.multi == 1) throw FriesError("fries_fciqmc_setup must be called first");
    for (uint32_t i = 0; i < n_iter; i++) fr_fq_iterate(&h->c, logs ? &logs[i] : nullptr);
    check_dev_err(&h->c);
    FR_API_END
}

extern "C" int fries_frimulti_setup(fries_ctx *h, const fries_frimulti_params *p) {
    FR_API_BEGIN
    FR_HIP(hipSetDevice(h->c.device));
    if (h->c.vec.dets) throw FriesError("this context already holds a run");
    fr_multi_setup(&h->c, p);
    check_dev_err(&h->c);
    FR_API_END
}
extern "C" int fries_frimulti_iterate(fries_ctx *h, uint32_t n_iter, fries_fciqmc_log *logs) {
    FR_API_BEGIN
    FR_HIP(hipSetDevice(h->c.device));
    if (!h->c.fq_mode || h->c.fqw.multi != 1) throw FriesError("fries_frimulti_setup must be called first");
    for (uint32_t i = 0; i < n_iter; i++) fr_multi_iterate(&h->c, logs ? &logs[i] : nullptr);
    check_dev_err(&h->c);
    FR_API_END
}

extern "C" int fries_hh_setup(fries_ctx *h, const fries_hh_params *p) {
    FR_API_BEGIN
    FR_HIP(hipSetDevice(h->c.device));
    if (h->c.vec.dets) throw FriesError("this context already holds a run");
    fr_hh_setup(&h->c, p);
    check_dev_err(&h->c);
    FR_API_END
}
extern "C" int fries_hh_iterate(fries_ctx *h, uint32_t n_iter, fries_iter_log *logs) {
    FR_API_BEGIN
    FR_HIP(hipSetDevice(h->c.device));
    if (!h->c.hh_mode) throw FriesError("fries_hh_setup must be called first");
    for (uint32_t i = 0; i < n_iter; i++) fr_hh_iterate(&h->c, logs ? &logs[i] : nullptr);
    check_dev_err(&h->c);
    FR_API_END
}

extern "C" int fries_vec_info(fries_ctx *h, uint32_t *curr_size, int32_t *n_nonz, uint32_t *n_free) {
    FR_API_BEGIN
    FriesCtx *c = &h->c;
    FR_HIP(hipSetDevice(c->device));
    fr_vec_sync_state(c, &c->vec, &c->h_vst);
    if (curr_size) *curr_size = c->h_vst.curr_size;
    if (n_nonz) *n_nonz = c->h_vst.n_nonz;
    if (n_free) *n_free = c->h_vst.n_free;
    FR_API_END
}

extern "C" int fries_vec_download(fries_ctx *h, uint64_t *dets, double *vals, size_t cap, size_t *n) {
    FR_API_BEGIN
    FriesCtx *c = &h->c;
    FR_HIP(hipSetDevice(c->device));
    fr_vec_sync_state(c, &c->vec, &c->h_vst);
    size_t m = c->h_vst.curr_size;
    if (n) *n = m;
    if (cap < m) throw FriesError("output buffer too small");
    if (dets) FR_HIP(hipMemcpy(dets, c->vec.dets, 8 * m, hipMemcpyDeviceToHost));
    if (vals) FR_HIP(hipMemcpy(vals, c->vec.v0, 8 * m, hipMemcpyDeviceToHost));
    FR_API_END
}

// ---- column mirrors for hosts that keep DistVec::values() / operator[] / matr_el_at_pos pointer semantics (include/FRIES/vec_utils.hpp)
extern "C" int fries_vec_column_download(fries_ctx *h, int column, double *out, size_t cap, size_t *n) {
    FR_API_BEGIN
    FriesCtx *c = &h->c;
    FR_HIP(hipSetDevice(c->device));
    if (column != 0 && column != 1) throw FriesError("column must be 0 or 1");
    fr_vec_sync_state(c, &c->vec, &c->h_vst);
    size_t m = c->h_vst.curr_size;
    if (n) *n = m;
    if (cap < m) throw FriesError("output buffer too small");
    if (out && m) FR_HIP(hipMemcpy(out, column ? c->vec.v1 : c->vec.v0, 8 * m, hipMemcpyDeviceToHost));
    FR_API_END
}
extern "C" int fries_vec_column_upload(fries_ctx *h, int column, const double *in, size_t n) {
    FR_API_BEGIN
    FriesCtx *c = &h->c;
    FR_HIP(hipSetDevice(c->device));
    if (column != 0 && column != 1) throw FriesError("column must be 0 or 1");
    fr_vec_sync_state(c, &c->vec, &c->h_vst);
    if (n > c->h_vst.curr_size) throw FriesError("more values than stored positions");
    if (n) FR_HIP(hipMemcpy(column ? c->vec.v1 : c->vec.v0, in, 8 * n, hipMemcpyHostToDevice));
    FR_API_END
}
extern "C" int fries_vec_column_zero(fries_ctx *h, int column) {
    FR_API_BEGIN
    FriesCtx *c = &h->c;
    FR_HIP(hipSetDevice(c->device));
    if (column != 0 && column != 1) throw FriesError("column must be 0 or 1");
    FR_HIP(hipMemsetAsync(column ? c->vec.v1 : c->vec.v0, 0, 8 * (size_t)c->vec.cap, c->stream));      // DistVec::zero_vec: the whole row (vec_utils.hpp:577-579)
    FR_API_END
}
// DistVec::add_vecs(idx1, idx2, c) (vec_utils.hpp:553-557): v[idx1][i] += v[idx2][i] * c over the stored positions
__global__ void __launch_bounds__(FR_BLOCK) k_add_vecs(VecDev V, int idx1, int idx2, double cf) {
    const uint32_t n = V.st->curr_size;
    double *a = idx1 ? V.v1 : V.v0; const double *b = idx2 ? V.v1 : V.v0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] += b[i] * cf;
}
extern "C" int fries_vec_add_vecs(fries_ctx *h, int idx1, int idx2, double cf) {
    FR_API_BEGIN
    FriesCtx *c = &h->c;
    FR_HIP(hipSetDevice(c->device));
    if ((idx1 != 0 && idx1 != 1) || (idx2 != 0 && idx2 != 1)) throw FriesError("column must be 0 or 1");
    fr_vec_sync_state(c, &c->vec, &c->h_vst);
    const uint32_t m = c->h_vst.curr_size;
    if (m) FR_LAUNCH(c, "k_add_vecs", k_add_vecs, dim3(fr_blocks(m, FR_BLOCK) > 2048 ? 2048 : fr_blocks(m, FR_BLOCK)), dim3(FR_BLOCK), c->vec, idx1, idx2, cf);
    FR_API_END
}
// DistVec::matr_el_at_pos for every stored position (vec_utils.hpp:672-677): the cached diagonal element - hf_en, computed where missing
__global__ void __launch_bounds__(FR_BLOCK) k_fill_diag(VecDev V, SysDev S) {
    const uint32_t n = V.st->curr_size;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (!V.active[i]) continue;
        double d = V.diag[i];
        if (d != d) V.diag[i] = fr_diag_matrel(V.dets[i], S.h_core, S.eris, S.n_orb) - S.hf_en;
    }
}
extern "C" int fries_vec_diag_download(fries_ctx *h, double *out, size_t cap, size_t *n) {
    FR_API_BEGIN
    FriesCtx *c = &h->c;
    FR_HIP(hipSetDevice(c->device));
    fr_vec_sync_state(c, &c->vec, &c->h_vst);
    size_t m = c->h_vst.curr_size;
    if (n) *n = m;
    if (cap < m) throw FriesError("output buffer too small");
    if (m) {
        SysDev S; S.n_orb = c->n_orb; S.n_elec = c->n_elec; S.h_core = c->d_h; S.eris = c->d_eris; S.hb = c->d_hb; S.hf_en = c->hf_en; S.spin_parity = c->spin_parity;
        FR_LAUNCH(c, "k_fill_diag", k_fill_diag, dim3(fr_blocks(m, FR_BLOCK) > 2048 ? 2048 : fr_blocks(m, FR_BLOCK)), dim3(FR_BLOCK), c->vec, S);
        FR_HIP(hipStreamSynchronize(c->stream));
        FR_HIP(hipMemcpy(out, c->vec.diag, 8 * m, hipMemcpyDeviceToHost));
    }
    FR_API_END
}
// DistVec::dot(idx2, vals2, num2, hashes2) (vec_utils.hpp:228-238): sum over the list IN LIST ORDER of vals2[i] * v[column][pos(idx2[i])],
// absent determinants skipped -- bit-identical to the reference's loop (products in parallel, the additions by one lane in order)
__global__ void __launch_bounds__(FR_BLOCK) k_dot_list(VecDev V, int column, const det_t *dets, const double *w, uint32_t n, double *out) {
    __shared__ double prod[FR_BLOCK];
    const double *col = column ? V.v1 : V.v0;
    double acc = 0;
    for (uint32_t base = 0; base < n; base += FR_BLOCK) {
        const uint32_t i = base + threadIdx.x;
        double p = 0;
        if (i < n) {
            uint32_t s = fr_hash_find(V, dets[i]);
            if (s != FR_NOPOS) { uint32_t pos = V.hs[s].val; if (pos < V.cap) p = w[i] * col[pos]; }
        }
        prod[threadIdx.x] = p;
        __syncthreads();
        if (threadIdx.x == 0) { const uint32_t m = n - base < FR_BLOCK ? n - base : FR_BLOCK; for (uint32_t j = 0; j < m; j++) acc += prod[j]; }   // + (+-0) leaves acc as it is
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = acc;
}
extern "C" int fries_vec_dot_list(fries_ctx *h, int column, const uint64_t *dets, const double *vals, size_t n, double *out) {
    FR_API_BEGIN
    FriesCtx *c = &h->c;
    FR_HIP(hipSetDevice(c->device));
    if (column != 0 && column != 1) throw FriesError("column must be 0 or 1");
    if (n > 0xffffffffull) throw FriesError("list too long");
    double r = 0;
    if (n) {
        det_t *dd = fr_alloc<det_t>(n); double *dw = fr_alloc<double>(n + 1);
        FR_HIP(hipMemcpy(dd, dets, 8 * n, hipMemcpyHostToDevice));
        FR_HIP(hipMemcpy(dw, vals, 8 * n, hipMemcpyHostToDevice));
        FR_LAUNCH(c, "k_dot_list", k_dot_list, dim3(1), dim3(FR_BLOCK), c->vec, column, dd, dw, (uint32_t)n, dw + n);
        FR_HIP(hipStreamSynchronize(c->stream));
        FR_HIP(hipMemcpy(&r, dw + n, 8, hipMemcpyDeviceToHost));
        hipFree(dd); hipFree(dw);
    }
    if (out) *out = r;
    FR_API_END
}

extern "C" int fries_htrial_download(fries_ctx *h, uint64_t *dets, double *vals, size_t cap, size_t *n) {
    FR_API_BEGIN
    FriesCtx *c = &h->c;
    FR_HIP(hipSetDevice(c->device));
    if (n) *n = c->n_htrial;
    if (cap < c->n_htrial) throw FriesError("output buffer too small");
    FR_HIP(hipMemcpy(dets, c->htr_det, 8 * (size_t)c->n_htrial, hipMemcpyDeviceToHost));
    FR_HIP(hipMemcpy(vals, c->htr_val, 8 * (size_t)c->n_htrial, hipMemcpyDeviceToHost));
    FR_API_END
}

extern "C" int fries_vec_add(fries_ctx *h, const uint64_t *dets, const double *vals, const uint8_t *ini, size_t n) {
    FR_API_BEGIN
    FriesCtx *c = &h->c;
    FR_HIP(hipSetDevice(c->device));
    if (n > c->sp.cap) throw FriesError("Too many elements added to Adder - must call perform_add() more frequently.");
    if (c->use_comm) throw FriesError("fries_vec_add is a one-rank entry point; with ranks, adds travel inside fries_frisys_iterate");
    // DistVec::add drops zero values before they reach the adder (vec_utils.hpp:418-423)
    std::vector<det_t> d; std::vector<double> v; std::vector<uint8_t> f;
    for (size_t i = 0; i < n; i++) if (vals[i] != 0) { d.push_back(dets[i]); v.push_back(vals[i]); f.push_back(ini[i]); }
    uint32_t m = (uint32_t)d.size();
    if (m) {
        FR_HIP(hipMemcpyAsync(c->sp.det, d.data(), 8 * (size_t)m, hipMemcpyHostToDevice, c->stream));
        FR_HIP(hipMemcpyAsync(c->sp.val, v.data(), 8 * (size_t)m, hipMemcpyHostToDevice, c->stream));
        FR_HIP(hipMemcpyAsync(c->sp.ini, f.data(), m, hipMemcpyHostToDevice, c->stream));
        FR_HIP(hipMemcpyAsync(c->sp.n_spawn, &m, 4, hipMemcpyHostToDevice, c->stream));
        fr_vec_merge(c, &c->vec, m, true);
    }
    fr_vec_sync_state(c, &c->vec, &c->h_vst);
    check_dev_err(c);
    FR_API_END
}

// DistVec::add x n + perform_add(0) with curr_vec_idx = column: column 1 is where the drivers collect the spawns, judged by the
// initiator rule against column 0 (frisys_mol.cpp:424-471, vec_utils.hpp:606-641)
extern "C" int fries_vec_add_to(fries_ctx *h, int column, const uint64_t *dets, const double *vals, const uint8_t *ini, size_t n) {
    FR_API_BEGIN
    FriesCtx *c = &h->c;
    FR_HIP(hipSetDevice(c->device));
    if (column != 0 && column != 1) throw FriesError("column must be 0 or 1");
    if (n > c->sp.cap) throw FriesError("Too many elements added to Adder - must call perform_add() more frequently.");
    std::vector<det_t> d; std::vector<double> v; std::vector<uint8_t> f;
    for (size_t i = 0; i < n; i++) if (vals[i] != 0) { d.push_back(dets[i]); v.push_back(vals[i]); f.push_back(ini[i]); }
    uint32_t m = (uint32_t)d.size();
    if (m) {
        fr_vec_sync_state(c, &c->vec, &c->h_vst);
        fr_vec_maybe_rebuild(c, &c->vec);           // tombstones of earlier deletes (the fused loop does this once per iteration)
        FR_HIP(hipMemcpyAsync(c->sp.det, d.data(), 8 * (size_t)m, hipMemcpyHostToDevice, c->stream));
        FR_HIP(hipMemcpyAsync(c->sp.val, v.data(), 8 * (size_t)m, hipMemcpyHostToDevice, c->stream));
        FR_HIP(hipMemcpyAsync(c->sp.ini, f.data(), m, hipMemcpyHostToDevice, c->stream));
        FR_HIP(hipMemcpyAsync(c->sp.n_spawn, &m, 4, hipMemcpyHostToDevice, c->stream));
        fr_vec_merge(c, &c->vec, m, column == 0);
    }
    fr_vec_sync_state(c, &c->vec, &c->h_vst);
    check_dev_err(c);
    FR_API_END
}

// frisys_mol.cpp:487-499: v0[i] *= 1 - eps (H_ii - shift) for the first vec_size positions, add_vecs(0, 1), zero_vec on column 1
extern "C" int fries_death_clone(fries_ctx *h, double eps, double shift, uint32_t vec_size) {
    FR_API_BEGIN
    FriesCtx *c = &h->c;
    FR_HIP(hipSetDevice(c->device));
    fr_vec_sync_state(c, &c->vec, &c->h_vst);
    const double se = c->eps, ss = c->en_shift;
    c->eps = eps; c->en_shift = shift;
    fr_death_clone(c, vec_size);
    c->eps = se; c->en_shift = ss;
    check_dev_err(c);
    FR_API_END
}

// DistVec::dot with H * trial and with the trial vector (frisys_mol.cpp:511-517, vec_utils.hpp:228-238)
extern "C" int fries_dots(fries_ctx *h, double *numer, double *denom) {
    FR_API_BEGIN
    FriesCtx *c = &h->c;
    FR_HIP(hipSetDevice(c->device));
    double nu = 0, de = 0;
    fr_dots(c, &nu, &de);
    if (numer) *numer = nu;
    if (denom) *denom = de;
    FR_API_END
}

// find_preserve on column 0 (compress_utils.cpp:29-105): *n_samp in = budget, out = samples left; the preserved set stays on
// the device for fries_sys_comp.  Returns through glob_norm the one-norm before compression.
extern "C" int fries_find_preserve(fries_ctx *h, uint32_t *n_samp, double *glob_norm) {
    FR_API_BEGIN
    FriesCtx *c = &h->c;
    FR_HIP(hipSetDevice(c->device));
    fr_vec_sync_state(c, &c->vec, &c->h_vst);
    const uint32_t save = c->vec_nonz;
    c->vec_nonz = *n_samp;
    fr_abs_sums(c);
    c->vec_nonz = save;
    double gn = 0;
    fr_find_preserve(c, n_samp, &gn);
    if (glob_norm) *glob_norm = gn;
    check_dev_err(c);
    FR_API_END
}
// sys_comp (compress_utils.cpp:283-327) after fries_find_preserve, followed by the driver's deletes (frisys_mol.cpp:534-539)
extern "C" int fries_sys_comp(fries_ctx *h, uint32_t n_samp, double rn) {
    FR_API_BEGIN
    FriesCtx *c = &h->c;
    FR_HIP(hipSetDevice(c->device));
    if (c->hh_mode) c->hh_keep0 = c->rank == 0;      // frisys_hh.cpp:356: position 0 of rank 0 (the Neel state) is never released
    fr_sys_comp(c, n_samp, rn);
    fr_vec_sync_state(c, &c->vec, &c->h_vst);
    check_dev_err(c);
    FR_API_END
}

// Time-reversal symmetry (the subspace drivers' spin_parity argument of h_op_offdiag / apply_HBPP_piv, molecule.cpp:298-369, heat_bathPP.cpp:1326-1407)
extern "C" int fries_set_spin_parity(fries_ctx *h, int spin_parity) {
    FR_API_BEGIN
    if (spin_parity < -1 || spin_parity > 1) throw FriesError("spin_parity must be -1, 0 or 1");
    h->c.spin_parity = spin_parity;
    FR_API_END
}
// h_op_offdiag(vec, ..., dest_idx = 1, h_fac = 1, spin_parity) on a fresh two-column vector that holds the list in column 0 (molecule.cpp:448-665):
// the stored determinants in position order and column 1
extern "C" int fries_h_offdiag_list(fries_ctx *h, const uint64_t *dets, const double *vals, size_t n, uint64_t *out_dets, double *out_vals, size_t cap, size_t *n_out) {
    FR_API_BEGIN
    FriesCtx *c = &h->c;
    FR_HIP(hipSetDevice(c->device));
    if (!c->d_eris) throw FriesError("fries_set_molecule must be called first");
    std::vector<det_t> src(dets, dets + n), od;
    std::vector<double> val(vals, vals + n), ov;
    fr_h_apply_list(c, src, val, od, ov, nullptr, nullptr, false);
    if (od.size() > cap) throw FriesError("fries_h_offdiag_list: output buffers too small");
    std::copy(od.begin(), od.end(), out_dets); std::copy(ov.begin(), ov.end(), out_vals);
    *n_out = od.size();
    FR_API_END
}

extern "C" int fries_set_vec_scrambler(fries_ctx *h, const uint32_t *vec_scr, size_t n) {
    FR_API_BEGIN
    FriesCtx *c = &h->c;
    if (c->vec.dets) throw FriesError("fries_set_vec_scrambler must be called before the driver's setup");
    c->in_vec_scr.assign(vec_scr, vec_scr + n);
    FR_API_END
}
extern "C" int fries_hh_comp_sub(fries_ctx *h, int stage, uint32_t n_samp, double rn, uint32_t *idx0, uint32_t *idx1, double *vals, size_t cap, size_t *n_out) {
    FR_API_BEGIN
    FriesCtx *c = &h->c;
    FR_HIP(hipSetDevice(c->device));
    if (!c->hh_mode) throw FriesError("fries_hh_setup must be called first");
    if (stage != 1 && stage != 2) throw FriesError("stage must be 1 or 2");
    if (stage == 1) { fr_vec_sync_state(c, &c->vec, &c->h_vst); fr_vec_maybe_rebuild(c, &c->vec); }
    fr_hh_stage(c, stage, n_samp, rn);
    const size_t m = c->comp_len[stage - 1];
    if (m > cap || m > c->W.cap) throw FriesError("Error: insufficient memory allocated for matrix compression.");
    if (m) {
        FR_HIP(hipMemcpyAsync(idx0, c->W.e_wi, 4 * m, hipMemcpyDeviceToHost, c->stream));
        FR_HIP(hipMemcpyAsync(idx1, c->W.e_sub, 4 * m, hipMemcpyDeviceToHost, c->stream));
        FR_HIP(hipMemcpyAsync(vals, c->W.e_val, 8 * m, hipMemcpyDeviceToHost, c->stream));
        FR_HIP(hipStreamSynchronize(c->stream));
    }
    *n_out = m;
    check_dev_err(c);
    FR_API_END
}
extern "C" int fries_hh_ref_ovlp(fries_ctx *h, double *ovlp) {
    FR_API_BEGIN
    FriesCtx *c = &h->c;
    FR_HIP(hipSetDevice(c->device));
    if (!c->hh_mode) throw FriesError("fries_hh_setup must be called first");
    double o3[3];
    fr_hh_ref_ovlp(c, o3);
    *ovlp = o3[0];
    FR_API_END
}

extern "C" int fries_vec_load(fries_ctx *h, const uint64_t *dets, const double *vals, size_t n) {
    FR_API_BEGIN
    FriesCtx *c = &h->c;
    FR_HIP(hipSetDevice(c->device));
    VecDev &v = c->vec;
    if (n > v.cap) throw FriesError("vector larger than max_dets");
    FR_HIP(hipMemsetAsync(v.v0, 0, 8 * (size_t)v.cap, c->stream));
    FR_HIP(hipMemsetAsync(v.v1, 0, 8 * (size_t)v.cap, c->stream));
    FR_HIP(hipMemsetAsync(v.diag, 0xff, 8 * (size_t)v.cap, c->stream));
    FR_HIP(hipMemsetAsync(v.active, 0, v.cap, c->stream));
    FR_HIP(hipMemsetAsync(v.active, 1, n, c->stream));
    FR_HIP(hipMemcpyAsync(v.dets, dets, 8 * n, hipMemcpyHostToDevice, c->stream));
    FR_HIP(hipMemcpyAsync(v.v0, vals, 8 * n, hipMemcpyHostToDevice, c->stream));
    VecState s{};
    s.curr_size = (uint32_t)n; s.n_nonz = (int32_t)n; s.n_used = (uint32_t)(v.hcap);   // forces the rebuild below
    FR_HIP(hipMemcpyAsync(v.st, &s, sizeof(s), hipMemcpyHostToDevice, c->stream));
    FR_HIP(hipMemsetAsync(v.stat_part, 0, 8 * (size_t)FR_STAT_STRIPES * FR_STAT_STRIDE, c->stream));
    c->h_vst = s;
    fr_vec_maybe_rebuild(c, &v);
    fr_vec_sync_state(c, &v, &c->h_vst);
    check_dev_err(c);
    FR_API_END
}

extern "C" int fries_vec_set_dense(fries_ctx *h, uint32_t n_dense) {
    FR_API_BEGIN
    FriesCtx *c = &h->c;
    FR_HIP(hipSetDevice(c->device));
    if (!c->vec.dets || c->hh_mode || c->fq_mode || c->full_mode) throw FriesError("fries_vec_set_dense needs a frisys_mol context (after fries_frisys_setup and fries_vec_load)");
    if (c->d_dh_from) { hipFree(c->d_dh_from); hipFree(c->d_dh_to); hipFree(c->d_dh_el); c->d_dh_from = nullptr; c->d_dh_to = nullptr; c->d_dh_el = nullptr; }
    fr_vec_sync_state(c, &c->vec, &c->h_vst);
    fr_dense_declare(c, n_dense, false);
    check_dev_err(c);
    FR_API_END
}
extern "C" int fries_dense_sizes(fries_ctx *h, uint32_t *sizes, size_t cap, size_t *n_ranks) {
    FR_API_BEGIN
    FriesCtx *c = &h->c;
    if (n_ranks) *n_ranks = (size_t)c->n_ranks;
    if (cap < (size_t)c->n_ranks) throw FriesError("fries_dense_sizes: buffer smaller than the number of ranks");
    for (int r = 0; r < c->n_ranks; r++) sizes[r] = (size_t)r < c->dense_sizes.size() ? c->dense_sizes[r] : 0u;
    FR_API_END
}

extern "C" int fries_apply_hbpp_sys(fries_ctx *h, uint32_t n_samp, const double rn[5], int unit_matrel,
                                    uint32_t *det_pos, uint8_t *orbs, double *vals, size_t cap, size_t *n_out, uint32_t comp_len[5]) {
    FR_API_BEGIN
    FriesCtx *c = &h->c;
    FR_HIP(hipSetDevice(c->device));
    fr_vec_sync_state(c, &c->vec, &c->h_vst);
    if (unit_matrel) fr_hbpp_apply_unit(c, n_samp, rn); else fr_hbpp_apply(c, n_samp, rn);
    size_t m = c->num_success;
    if (n_out) *n_out = m;
    if (comp_len) for (int k = 0; k < 5; k++) comp_len[k] = c->comp_len[k];
    check_dev_err(c);
    if (cap < m) throw FriesError("output buffer too small");
    if (det_pos) FR_HIP(hipMemcpy(det_pos, c->c_pos, 4 * m, hipMemcpyDeviceToHost));
    if (orbs) FR_HIP(hipMemcpy(orbs, c->c_orbs, 4 * m, hipMemcpyDeviceToHost));
    if (vals) FR_HIP(hipMemcpy(vals, c->c_val, 8 * m, hipMemcpyDeviceToHost));
    FR_API_END
}

extern "C" int fries_apply_hbpp_piv(fries_ctx *h, uint32_t n_samp, int unit_matrel, uint32_t *det_pos, uint8_t *orbs, double *vals, size_t cap, size_t *n_out,
                                    uint32_t stage_len[5]) {
    FR_API_BEGIN
    FriesCtx *c = &h->c;
    FR_HIP(hipSetDevice(c->device));
    if (c->hh_mode || c->fq_mode || !c->W.cap) throw FriesError("fries_apply_hbpp_piv needs a context set up by fries_frisys_setup");
    if (c->spin_parity && !c->new_hb) throw FriesError("Time-reversal symmetry is only implemented for the unnormalized heat-bath distribution");       // heat_bathPP.cpp:1019-1021
    uint32_t sl[5] = {0, 0, 0, 0, 0};
    fr_hbpp_piv_apply(c, n_samp, unit_matrel, sl);
    size_t m = c->num_success;
    if (n_out) *n_out = m;
    if (stage_len) for (int k = 0; k < 5; k++) stage_len[k] = sl[k];
    check_dev_err(c);
    if (cap < m) throw FriesError("output buffer too small");
    if (det_pos) FR_HIP(hipMemcpy(det_pos, c->c_pos, 4 * m, hipMemcpyDeviceToHost));
    if (orbs) FR_HIP(hipMemcpy(orbs, c->c_orbs, 4 * m, hipMemcpyDeviceToHost));
    if (vals) FR_HIP(hipMemcpy(vals, c->c_val, 8 * m, hipMemcpyDeviceToHost));
    FR_API_END
}

extern "C" int fries_compress_vec(fries_ctx *h, uint32_t n_samp_in, double rn, uint32_t *n_kept, double *glob_norm) {
    FR_API_BEGIN
    FriesCtx *c = &h->c;
    FR_HIP(hipSetDevice(c->device));
    fr_vec_sync_state(c, &c->vec, &c->h_vst);
    uint32_t save = c->vec_nonz;
    c->vec_nonz = n_samp_in;
    fr_death_clone(c, 0);                    // publishes the |v| block sums; v1 is zero so values are unchanged
    uint32_t n_samp = n_samp_in;
    double gn = 0;
    fr_find_preserve(c, &n_samp, &gn);
    fr_sys_comp(c, n_samp, rn);
    c->vec_nonz = save;
    if (n_kept) *n_kept = n_samp_in - n_samp;
    if (glob_norm) *glob_norm = gn;
    fr_vec_sync_state(c, &c->vec, &c->h_vst);
    check_dev_err(c);
    FR_API_END
}

extern "C" int fries_compress_vec_piv(fries_ctx *h, uint32_t n_samp_in, uint32_t *n_kept, double *glob_norm) {
    FR_API_BEGIN
    FriesCtx *c = &h->c;
    FR_HIP(hipSetDevice(c->device));
    fr_piv_comp(c, n_samp_in, n_kept, glob_norm);
    check_dev_err(c);
    FR_API_END
}
extern "C" int fries_test_piv_adjust(fries_ctx *h, uint32_t *n_samp_loc, double exp_nsamp_loc, uint32_t n_samp_tot, double tot_norm, double *new_norm, uint8_t *flags_out) {
    FR_API_BEGIN
    FriesCtx *c = &h->c;
    FR_HIP(hipSetDevice(c->device));
    fr_test_piv_adjust(c, n_samp_loc, exp_nsamp_loc, n_samp_tot, tot_norm, new_norm, flags_out);
    check_dev_err(c);
    FR_API_END
}
extern "C" int fries_piv_stats(fries_ctx *h, uint64_t *n_certified, uint64_t *n_fallback) {
    if (n_certified) *n_certified = h->c.piv.n_certified;
    if (n_fallback) *n_fallback = h->c.piv.n_fallback;
    return (int)h->c.piv.last_reason;      // why the last fallback happened: 1 cut point near a border, 2 divmod, 4 walk, 8 candidate draw, 16 pass draw
}
extern "C" uint32_t fries_next_draw(fries_ctx *h) { return (uint32_t)h->c.mt(); }
// the generator as std::mt19937's own text form (operator<< / operator>>): a caller that owns a std::mt19937, like the reference's
// drivers, lends it to an operator whose number of draws depends on the data and takes it back afterwards
extern "C" int fries_rng_set_state(fries_ctx *h, const char *text) {
    FR_API_BEGIN
    std::istringstream is(text ? text : "");
    std::mt19937 m;
    is >> m;
    if (is.fail()) throw FriesError("fries_rng_set_state: not a std::mt19937 state");
    h->c.mt = m;
    FR_API_END
}
extern "C" int fries_rng_get_state(fries_ctx *h, char *buf, size_t cap, size_t *need) {
    FR_API_BEGIN
    std::ostringstream os;
    os << h->c.mt;
    const std::string s = os.str();
    if (need) *need = s.size() + 1;
    if (buf) {
        if (cap < s.size() + 1) throw FriesError("fries_rng_get_state: buffer too small");
        memcpy(buf, s.c_str(), s.size() + 1);
    }
    FR_API_END
}

__global__ void k_test_teeth(Teeth *t, double r0, double unit, uint32_t n, double *pos, const double *q, uint32_t nq, uint32_t *below) {
    if (blockIdx.x == 0 && threadIdx.x == 0) fr_build_teeth(t, r0, unit, n, 0.0);
}
__global__ void k_test_teeth_eval(const Teeth *t, uint32_t n, double *pos, const double *q, uint32_t nq, uint32_t *below) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) pos[i] = fr_tooth(t, i);
    if (i < nq) below[i] = fr_teeth_below(t, q[i]);
}

extern "C" int fries_test_teeth(fries_ctx *h, double r0, double unit, uint32_t n, double *out_pos, const double *query, uint32_t nq, uint32_t *out_below) {
    FR_API_BEGIN
    FriesCtx *c = &h->c;
    FR_HIP(hipSetDevice(c->device));
    Teeth *t = fr_alloc<Teeth>(1);
    double *dp = fr_alloc<double>(n), *dq = fr_alloc<double>(nq);
    uint32_t *db = fr_alloc<uint32_t>(nq);
    if (nq) FR_HIP(hipMemcpy(dq, query, 8 * (size_t)nq, hipMemcpyHostToDevice));
    FR_LAUNCH(c, "k_test_teeth", k_test_teeth, dim3(1), dim3(1), t, r0, unit, n, dp, dq, nq, db);
    uint32_t m = n > nq ? n : nq;
    FR_LAUNCH(c, "k_test_teeth_eval", k_test_teeth_eval, dim3(fr_blocks(m ? m : 1, FR_BLOCK)), dim3(FR_BLOCK), t, n, dp, dq, nq, db);
    FR_HIP(hipStreamSynchronize(c->stream));
    if (n) FR_HIP(hipMemcpy(out_pos, dp, 8 * (size_t)n, hipMemcpyDeviceToHost));
    if (nq) FR_HIP(hipMemcpy(out_below, db, 4 * (size_t)nq, hipMemcpyDeviceToHost));
    hipFree(t); hipFree(dp); hipFree(dq); hipFree(db);
    FR_API_END
}

// ------------------------------------------------------------------ test hook: exact in-order prefix sums
struct AccArr {
    const double *a; unsigned n;
    __device__ unsigned count() const { return n; }
    __device__ double get(size_t i) const { return a[i]; }
};
__global__ void __launch_bounds__(FR_BLOCK) k_test_seq_apply(SeqWork Q, AccArr acc, double *out) {
    __shared__ SeqShared sh;
    double S[4], Sb;
    fr_seq_prefix4(Q, acc, blockIdx.x, &sh, S, &Sb);
    size_t base = (size_t)blockIdx.x * FR_SEQ_TILE + (size_t)threadIdx.x * 4;
    for (int it = 0; it < 4; it++) if (base + it < acc.n) out[base + it] = S[it];
}

extern "C" int fries_test_seqsum(fries_ctx *h, const double *vals, uint32_t n, double start, double *out_prefix, double *out_total, uint32_t *n_dirty_tiles, uint32_t *n_dirty_subs) {
    FR_API_BEGIN
    FriesCtx *c = &h->c;
    FR_HIP(hipSetDevice(c->device));
    unsigned ntile = fr_blocks(n ? n : 1, FR_SEQ_TILE);
    if (ntile > FR_MAX_PART) throw FriesError("too many elements");
    SeqWork Q;
    Q.tiles = fr_alloc<SeqRec>(ntile); Q.subs = fr_alloc<SeqRec>((size_t)ntile * FR_SUBS_PER_TILE); Q.total = fr_alloc<double>(1); Q.tsum = fr_alloc<double>(ntile);
    double *da = fr_alloc<double>(n), *dout = fr_alloc<double>(n);
    FR_HIP(hipMemcpy(da, vals, 8 * (size_t)n, hipMemcpyHostToDevice));
    FR_HIP(hipMemset(Q.subs, 0, sizeof(SeqRec) * (size_t)ntile * FR_SUBS_PER_TILE));
    AccArr acc{da, n};
    double *dstart = fr_alloc<double>(1);
    FR_HIP(hipMemcpy(dstart, &start, 8, hipMemcpyHostToDevice));
    SeqStart from; from.norms = dstart; from.n = 1;      // 0 + start == start
    FR_LAUNCH(c, "k_seq_sums", (k_seq_sums<AccArr>), dim3(ntile), dim3(FR_BLOCK), Q, acc);
    FR_LAUNCH(c, "k_seq_maps", (k_seq_maps<AccArr>), dim3(ntile), dim3(FR_BLOCK), Q, acc, from);
    FR_LAUNCH(c, "k_seq_chain", (k_seq_chain<AccArr>), dim3(1), dim3(FR_BLOCK), Q, acc, from);
    FR_LAUNCH(c, "k_test_seq_apply", k_test_seq_apply, dim3(ntile), dim3(FR_BLOCK), Q, acc, dout);
    FR_HIP(hipStreamSynchronize(c->stream));
    FR_HIP(hipMemcpy(out_prefix, dout, 8 * (size_t)n, hipMemcpyDeviceToHost));
    FR_HIP(hipMemcpy(out_total, Q.total, 8, hipMemcpyDeviceToHost));
    std::vector<SeqRec> tl(ntile), sb((size_t)ntile * FR_SUBS_PER_TILE);
    FR_HIP(hipMemcpy(tl.data(), Q.tiles, sizeof(SeqRec) * ntile, hipMemcpyDeviceToHost));
    FR_HIP(hipMemcpy(sb.data(), Q.subs, sizeof(SeqRec) * sb.size(), hipMemcpyDeviceToHost));
    uint32_t dt = 0, ds = 0;
    for (unsigned t = 0; t < ntile; t++) if (tl[t].dirty) { dt++; for (int j = 0; j < FR_SUBS_PER_TILE; j++) if (sb[(size_t)t * FR_SUBS_PER_TILE + j].dirty) ds++; }
    if (n_dirty_tiles) *n_dirty_tiles = dt;
    if (n_dirty_subs) *n_dirty_subs = ds;
    hipFree(Q.tiles); hipFree(Q.subs); hipFree(Q.total); hipFree(Q.tsum); hipFree(da); hipFree(dout); hipFree(dstart);
    FR_API_END
}

// ------------------------------------------------------------------ profiling / restart helpers
extern "C" int fries_prof_enable(fries_ctx *h, int on) {
    FR_API_BEGIN
    FriesCtx *c = &h->c;
    FR_HIP(hipSetDevice(c->device));
    prof_collect(c);
    c->prof_on = on != 0;
    if (on) c->prof_agg.clear();
    FR_API_END
}
// device-to-device copy bandwidth of this GPU (bytes read + bytes written per second), the measured denominator beside the
// nominal HBM peak
extern "C" int fries_measure_copy_bandwidth(fries_ctx *h, size_t bytes, int reps, double *gb_per_s) {
    FR_API_BEGIN
    FriesCtx *c = &h->c;
    FR_HIP(hipSetDevice(c->device));
    void *a = nullptr, *b = nullptr;
    FR_HIP(hipMalloc(&a, bytes)); FR_HIP(hipMalloc(&b, bytes));
    FR_HIP(hipMemsetAsync(a, 1, bytes, c->stream));
    FR_HIP(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, c->stream));
    hipEvent_t e0, e1;
    FR_HIP(hipEventCreate(&e0)); FR_HIP(hipEventCreate(&e1));
    FR_HIP(hipEventRecord(e0, c->stream));
    for (int k = 0; k < reps; k++) FR_HIP(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, c->stream));
    FR_HIP(hipEventRecord(e1, c->stream));
    FR_HIP(hipStreamSynchronize(c->stream));
    float ms = 0;
    FR_HIP(hipEventElapsedTime(&ms, e0, e1));
    *gb_per_s = 2.0 * (double)bytes * reps / (ms * 1e-3) / 1e9;
    hipEventDestroy(e0); hipEventDestroy(e1); hipFree(a); hipFree(b);
    FR_API_END
}

extern "C" int fries_prof_count(fries_ctx *h) {
    try { hipSetDevice(h->c.device); prof_collect(&h->c); } catch (...) { return -1; }     // spans recorded by the single-operator entry points
    return (int)h->c.prof_agg.size();
}
extern "C" int fries_prof_get(fries_ctx *h, int i, char *name, size_t name_cap, double *total_ms, uint64_t *calls) {
    FR_API_BEGIN
    FriesCtx *c = &h->c;
    if (i < 0 || i >= (int)c->prof_agg.size()) throw FriesError("profile index out of range");
    const ProfAgg &a = c->prof_agg[i];
    snprintf(name, name_cap, "%s", a.name.c_str());
    *total_ms = a.ms; *calls = a.calls;
    FR_API_END
}
// restart support: re-seed the driver's mt19937 without redrawing the scramblers, set the shift
// (what --load_dir restores from S.txt, frisys_mol.cpp:257-263) and the iteration counter
extern "C" int fries_frisys_restart(fries_ctx *h, uint32_t seed, double en_shift, double last_one_norm, uint32_t iterat) {
    FR_API_BEGIN
    FriesCtx *c = &h->c;
    c->mt.seed(seed);
    c->en_shift = en_shift; c->last_one_norm = last_one_norm; c->iterat = iterat;
    FR_API_END
}

extern "C" int fries_counters(fries_ctx *h, uint64_t *iters, uint64_t *spawns, uint64_t *launches, uint64_t *fks_replays, uint64_t *stage_elems) {
    FriesCtx *c = &h->c;
    if (iters) *iters = c->tot_iters;
    if (spawns) *spawns = c->tot_spawns;
    if (launches) *launches = c->n_kernel_launch;
    if (fks_replays) *fks_replays = c->tot_fks_iters;
    if (stage_elems) *stage_elems = c->tot_stage_elems;
    return 0;
}
