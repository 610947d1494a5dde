// TEST INFRASTRUCTURE ONLY -- flat C entry points onto the CPU oracle for ctypes
// (tests/, bench.py's cpu_baseline leg, __graft_entry__.smoke()).
#include "fries_oracle.hpp"
#include <cstring>
#include <cstdio>
#include <stdexcept>
#include <cmath>
#include <memory>

using namespace fo;

struct OracleLog { double numer, denom, shift, norm; uint32_t nkept; int32_t n_nonz; uint32_t curr_size, num_success; uint32_t comp_len[5]; uint32_t err; };

static std::string g_last_error;
extern "C" {

// the per-iteration digest oracle/ref_harness.cpp logs: FNV-style over (determinant, value bits, position) of the non-zero entries
uint64_t fo_vec_digest(const uint64_t *dets, const double *vals, size_t n) {
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) {
        if (vals[i] == 0) continue;
        uint64_t vb; memcpy(&vb, &vals[i], 8);
        h = (h ^ dets[i]) * 1099511628211ull; h = (h ^ vb) * 1099511628211ull; h = (h ^ (uint64_t)i) * 1099511628211ull;
    }
    return h;
}

void *fo_frisys_create(uint32_t n_orb, uint32_t n_elec, const uint8_t *irreps, const double *h, const double *eris,
                       double eps, double target, double init, uint32_t vec_nonz, uint32_t mat_nonz, uint32_t max_dets, uint32_t seed, int hb_unnorm) {
    Frisys *f = new Frisys();
    f->sys.n_orb = n_orb; f->sys.n_elec = n_elec;
    f->sys.ints.n_orb = n_orb;
    f->sys.ints.h.assign(h, h + (size_t)n_orb * n_orb);
    f->sys.ints.eri.assign(eris, eris + Integrals::packed_len(n_orb));
    f->sys.symm.init(irreps, n_orb);
    f->par.eps = eps; f->par.target_norm = target; f->par.init_thresh = init;
    f->par.vec_nonz = vec_nonz; f->par.mat_nonz = mat_nonz; f->par.max_dets = max_dets;
    f->par.new_hb = hb_unnorm != 0; f->par.seed = seed;
    f->setup();
    return f;
}
// the same with the driver's optional inputs (--trial_vec, --ini_vec, --ham_shift); n == 0 / has_shift == 0 leave the defaults
void *fo_frisys_create_ex2(uint32_t n_orb, uint32_t n_elec, const uint8_t *irreps, const double *h, const double *eris,
                           double eps, double target, double init, uint32_t vec_nonz, uint32_t mat_nonz, uint32_t max_dets, uint32_t seed, int hb_unnorm,
                           const uint64_t *tr_det, const double *tr_val, size_t n_tr, const uint64_t *in_det, const double *in_val, size_t n_in,
                           int has_shift, double ham_shift_hf_en, const uint64_t *space, size_t n_space);
void *fo_frisys_create_ex(uint32_t n_orb, uint32_t n_elec, const uint8_t *irreps, const double *h, const double *eris,
                          double eps, double target, double init, uint32_t vec_nonz, uint32_t mat_nonz, uint32_t max_dets, uint32_t seed, int hb_unnorm,
                          const uint64_t *tr_det, const double *tr_val, size_t n_tr, const uint64_t *in_det, const double *in_val, size_t n_in,
                          int has_shift, double ham_shift_hf_en) {
    return fo_frisys_create_ex2(n_orb, n_elec, irreps, h, eris, eps, target, init, vec_nonz, mat_nonz, max_dets, seed, hb_unnorm, tr_det, tr_val, n_tr, in_det, in_val, n_in,
                                has_shift, ham_shift_hf_en, nullptr, 0);
}
/* ... and --det_space: the determinants of the dense (semi-stochastic) space */
void *fo_frisys_create_ex2(uint32_t n_orb, uint32_t n_elec, const uint8_t *irreps, const double *h, const double *eris,
                           double eps, double target, double init, uint32_t vec_nonz, uint32_t mat_nonz, uint32_t max_dets, uint32_t seed, int hb_unnorm,
                           const uint64_t *tr_det, const double *tr_val, size_t n_tr, const uint64_t *in_det, const double *in_val, size_t n_in,
                           int has_shift, double ham_shift_hf_en, const uint64_t *space, size_t n_space) {
    Frisys *f = new Frisys();
    if (n_space) f->det_space.assign(space, space + n_space);
    f->sys.n_orb = n_orb; f->sys.n_elec = n_elec;
    f->sys.ints.n_orb = n_orb;
    f->sys.ints.h.assign(h, h + (size_t)n_orb * n_orb);
    f->sys.ints.eri.assign(eris, eris + Integrals::packed_len(n_orb));
    f->sys.symm.init(irreps, n_orb);
    f->par.eps = eps; f->par.target_norm = target; f->par.init_thresh = init;
    f->par.vec_nonz = vec_nonz; f->par.mat_nonz = mat_nonz; f->par.max_dets = max_dets;
    f->par.new_hb = hb_unnorm != 0; f->par.seed = seed;
    if (n_tr) { f->trial_in_det.assign(tr_det, tr_det + n_tr); f->trial_in_val.assign(tr_val, tr_val + n_tr); }
    if (n_in) { f->ini_det.assign(in_det, in_det + n_in); f->ini_val.assign(in_val, in_val + n_in); }
    if (has_shift) { f->has_ham_shift = true; f->ham_shift = ham_shift_hf_en; }
    f->setup();
    return f;
}
void fo_frisys_destroy(void *h) { delete (Frisys *)h; }
void fo_frisys_iterate(void *h, uint32_t n, OracleLog *logs) {
    Frisys *f = (Frisys *)h;
    for (uint32_t i = 0; i < n; i++) {
        f->iterate(1);
        if (logs) {
            const IterLog &l = f->log.back();
            OracleLog &o = logs[i];
            o.numer = l.numer; o.denom = l.denom; o.shift = l.shift; o.norm = l.norm; o.nkept = l.nkept;
            o.n_nonz = l.n_nonz; o.curr_size = (uint32_t)l.curr_size; o.num_success = (uint32_t)l.num_success;
            for (int k = 0; k < 5; k++) o.comp_len[k] = (uint32_t)l.comp_len[k];
            o.err = 0;
        }
    }
}
double fo_frisys_p_doub(void *h) { return ((Frisys *)h)->p_doub; }
double fo_frisys_hf_en(void *h) { return ((Frisys *)h)->sys.hf_en; }
size_t fo_frisys_vec(void *h, uint64_t *dets, double *vals, size_t cap) {
    Frisys *f = (Frisys *)h;
    size_t n = f->sol.curr_size;
    if (dets && vals && cap >= n) for (size_t i = 0; i < n; i++) { dets[i] = f->sol.dets[i]; vals[i] = f->sol.vals[0][i]; }
    return n;
}
size_t fo_frisys_htrial(void *h, uint64_t *dets, double *vals, size_t cap) {
    Frisys *f = (Frisys *)h;
    size_t n = f->htrial_det.size();
    if (dets && vals && cap >= n) for (size_t i = 0; i < n; i++) { dets[i] = f->htrial_det[i]; vals[i] = f->htrial_val[i]; }
    return n;
}
// replace the stored vector (positions 0..n-1), as DistVec::load would
void fo_frisys_load(void *h, const uint64_t *dets, const double *vals, size_t n) {
    Frisys *f = (Frisys *)h;
    Vec &v = f->sol;
    size_t cap = v.max_size, ac = v.adder_cap;
    v.init(cap, ac, f->sys.n_elec, 2);
    uint8_t tmp[64];
    for (size_t i = 0; i < n; i++) {
        v.dets[i] = dets[i]; v.vals[0][i] = vals[i]; v.active[i] = 1; v.table[dets[i]] = (ptrdiff_t)i;
        occ_list(dets[i], tmp); memcpy(&v.occ[i * v.n_elec], tmp, v.n_elec);
    }
    v.curr_size = n; v.n_nonz = (int)n;
}
size_t fo_hb_tensor(void *h, int which, double *out, size_t cap) {
    Frisys *f = (Frisys *)h;
    const HBInfo &t = f->sys.hb;
    const std::vector<double> *src = nullptr;
    switch (which) {
        case 0: src = &t.s_tens; break; case 1: src = &t.d_same; break; case 2: src = &t.d_diff; break;
        case 3: src = &t.exch_sqrt; break; case 4: src = &t.diag_sqrt; break; case 5: src = &t.exch_norms; break;
        case 6: if (cap >= 1) out[0] = t.s_norm; return 1;
    }
    if (!src) return 0;
    if (cap >= src->size()) memcpy(out, src->data(), 8 * src->size());
    return src->size();
}
void fo_set_hb_tensor(void *h, int which, const double *in, size_t n) {
    Frisys *f = (Frisys *)h;
    HBInfo &t = f->sys.hb;
    switch (which) {
        case 0: t.s_tens.assign(in, in + n); break; case 1: t.d_same.assign(in, in + n); break; case 2: t.d_diff.assign(in, in + n); break;
        case 3: t.exch_sqrt.assign(in, in + n); break; case 4: t.diag_sqrt.assign(in, in + n); break; case 5: t.exch_norms.assign(in, in + n); break;
        case 6: t.s_norm = in[0]; break;
    }
}
void fo_matrel_batch(void *h, int kind, const uint64_t *dets, const uint8_t *orbs, size_t n, double *out, int32_t *sign) {
    Frisys *f = (Frisys *)h;
    uint8_t occ[64];
    for (size_t i = 0; i < n; i++) {
        occ_list(dets[i], occ);
        if (kind == 0) { out[i] = diag_matrel(occ, f->sys.ints, f->sys.n_elec); if (sign) sign[i] = 1; }
        else if (kind == 1) { out[i] = sing_matrel_nosgn(orbs + 4 * i, occ, f->sys.ints, f->sys.n_elec); if (sign) sign[i] = sing_parity(dets[i], orbs + 4 * i); }
        else { out[i] = doub_matrel_nosgn(orbs + 4 * i, f->sys.ints); if (sign) sign[i] = doub_parity(dets[i], orbs + 4 * i); }
    }
}
// apply_HBPP_sys on the handle's stored vector
size_t fo_apply_hbpp_sys(void *h, uint32_t n_samp, const double *rn, int unit_matrel, uint32_t *pos, uint8_t *orbs, double *vals, size_t cap) {
    Frisys *f = (Frisys *)h;
    HBScratch &sc = f->sc;
    size_t need = f->sol.curr_size > (size_t)n_samp * 4 ? f->sol.curr_size : (size_t)n_samp * 4;
    if (sc.len < need) { size_t ns = f->sys.n_elec > (f->sys.n_orb - f->sys.n_elec / 2) ? f->sys.n_elec : f->sys.n_orb - f->sys.n_elec / 2; sc.init(need, ns); }
    std::copy(f->sol.vals[0].begin(), f->sol.vals[0].begin() + f->sol.curr_size, sc.vec1.begin());
    for (size_t i = 0; i < f->sol.curr_size; i++) sc.det_idx1[i] = i;
    sc.vec_len = f->sol.curr_size;
    apply_HBPP_sys(f->sol, sc, f->sys, f->p_doub, f->par.new_hb, rn, n_samp, unit_matrel != 0);
    size_t n = sc.vec_len;
    if (cap >= n) for (size_t i = 0; i < n; i++) { pos[i] = (uint32_t)sc.det_idx2[i]; memcpy(orbs + 4 * i, &sc.orb1[4 * i], 4); vals[i] = sc.vec1[i]; }
    return n;
}
// time-reversal symmetry for the calls below (spin_parity of the reference's h_op_offdiag / apply_HBPP_piv)
static int g_spin_parity = 0;
void fo_set_spin_parity(int sp) { g_spin_parity = sp; }
// h_op_offdiag(vec, dest 1, h_fac 1, spin parity) on a fresh two-column vector holding (dets, vals): stored determinants and column 1
size_t fo_h_offdiag_list(void *h, const uint64_t *dets, const double *vals, size_t n, uint64_t *out_dets, double *out_vals, size_t cap) {
    Frisys *f = (Frisys *)h;
    const size_t room = n * ((size_t)f->sys.n_orb * f->sys.n_orb * f->sys.n_elec * f->sys.n_elec + 2) + 64;
    Vec v; v.init(room, room, f->sys.n_elec, 2);
    for (size_t i = 0; i < n; i++) v.add(dets[i], vals[i], 1);
    v.perform_add(0);
    h_op_offdiag(v, v.curr_size, f->sys, 1, 1.0, g_spin_parity);
    if (v.curr_size > cap) return (size_t)-1;
    for (size_t i = 0; i < v.curr_size; i++) { out_dets[i] = v.dets[i]; out_vals[i] = v.vals[1][i]; }
    return v.curr_size;
}
// apply_HBPP_piv on the handle's stored vector, drawing from the handle's generator (seed it with fo_frisys_restart)
size_t fo_apply_hbpp_piv(void *h, uint32_t n_samp, int unit_matrel, uint32_t *pos, uint8_t *orbs, double *vals, size_t cap, uint64_t *stage_len) {
    Frisys *f = (Frisys *)h;
    HBPivScratch ps;
    size_t n = f->sol.curr_size;
    size_t len = (n > (size_t)n_samp ? n : (size_t)n_samp) * 2 + 64;
    size_t ns = f->sys.n_elec > (f->sys.n_orb - f->sys.n_elec / 2) ? f->sys.n_elec : f->sys.n_orb - f->sys.n_elec / 2;
    ps.init(len, ns);
    std::copy(f->sol.vals[0].begin(), f->sol.vals[0].begin() + n, ps.vec1.begin());
    for (size_t i = 0; i < n; i++) ps.det_idx1[i] = i;
    ps.vec_len = n;
    apply_HBPP_piv(f->sol, ps, f->sys, f->p_doub, f->par.new_hb, f->mt, n_samp, unit_matrel != 0, Comm::self(), g_spin_parity);
    size_t m = ps.vec_len;
    if (cap >= m) for (size_t i = 0; i < m; i++) { pos[i] = (uint32_t)ps.det_idx2[i]; memcpy(orbs + 4 * i, &ps.orb1[4 * i], 4); vals[i] = ps.vec1[i]; }
    if (stage_len) for (int k = 0; k < 5; k++) stage_len[k] = ps.stage_len[k];
    return m;
}
void fo_set_p_doub(void *h, double p) { ((Frisys *)h)->p_doub = p; }
// find_preserve + sys_comp + deletes on the handle's stored vector
void fo_compress_vec(void *h, uint32_t n_samp, double rn, uint32_t *n_kept, double *glob_norm) {
    Frisys *f = (Frisys *)h;
    Vec &v = f->sol;
    if (f->srt.size() < v.max_size) { f->srt.resize(v.max_size); f->keep.resize(v.max_size, 0); }
    unsigned ns = n_samp;
    double gn;
    double ln = find_preserve(v.vals[0].data(), f->srt, f->keep, v.curr_size, &ns, &gn);
    sys_comp(v.vals[0].data(), v.curr_size, &ln, ns, f->keep, rn);
    for (size_t i = 0; i < v.curr_size; i++) if (f->keep[i]) { v.del_at_pos(i); f->keep[i] = 0; }
    if (n_kept) *n_kept = n_samp - ns;
    if (glob_norm) *glob_norm = gn;
}
// sum over every symmetry-allowed double excitation of `det` of calc_norm_wt (heat_bathPP.cpp:413-481): the probability
// hb_doub_multi assigns to it.  What is missing from 1 is the mass of draws it rejects (an occupied second virtual).
double fo_norm_wt_sum(void *h, uint64_t det, uint32_t *n_doub) {
    Frisys *f = (Frisys *)h;
    uint8_t occ[64];
    occ_list(det, occ);
    std::vector<uint8_t> ex;
    size_t nd = doub_ex_symm(det, occ, f->sys.n_elec, f->sys.n_orb, ex, f->sys.symm.irrep.data());
    double tot = 0;
    for (size_t e = 0; e < nd; e++) tot += calc_norm_wt(f->sys.hb, &ex[4 * e], occ, f->sys.n_elec, det, f->sys.symm);
    if (n_doub) *n_doub = (uint32_t)nd;
    return tot;
}
// n_draws heat-bath double excitations of `det` (hb_doub_multi on the reference's mt19937 stream, 32 per call): how often each
// symmetry-allowed double came out, the probability calc_norm_wt reported for it, and the probability listed for excitation k
// of the enumeration.  Returns the number of accepted draws.
uint64_t fo_hb_sample_hist(void *h, uint64_t det, uint64_t n_draws, uint32_t seed, uint64_t *counts, double *reported, double *listed, size_t cap) {
    Frisys *f = (Frisys *)h;
    uint8_t occ[64];
    occ_list(det, occ);
    std::vector<uint8_t> ex;
    size_t nd = doub_ex_symm(det, occ, f->sys.n_elec, f->sys.n_orb, ex, f->sys.symm.irrep.data());
    if (cap < nd) return 0;
    std::unordered_map<uint32_t, size_t> where;
    for (size_t e = 0; e < nd; e++) {
        uint32_t key; memcpy(&key, &ex[4 * e], 4);
        where[key] = e;
        counts[e] = 0; reported[e] = 0;
        listed[e] = calc_norm_wt(f->sys.hb, &ex[4 * e], occ, f->sys.n_elec, det, f->sys.symm);
    }
    std::mt19937 g(seed);
    Rng rng; rng.mt = &g;
    uint64_t accepted = 0;
    uint8_t orbs[32 * 4]; double prob[32]; uint32_t att[32];
    for (uint64_t done = 0; done < n_draws; done += 32) {
        unsigned got = hb_doub_multi(det, occ, f->sys.n_elec, f->sys.symm, f->sys.hb, 32, rng, 0, orbs, prob, att);
        for (unsigned k = 0; k < got; k++) {
            uint32_t key; memcpy(&key, &orbs[4 * k], 4);
            auto it = where.find(key);
            if (it == where.end()) return ~(uint64_t)0;        // a draw outside the enumeration
            counts[it->second]++; reported[it->second] = prob[k];
        }
        accepted += got;
    }
    return accepted;
}
// frifull_mol (fo::Frifull)
void *fo_frifull_create(uint32_t n_orb, uint32_t n_elec, const uint8_t *irreps, const double *h, const double *eris,
                        double eps, double target, uint32_t vec_nonz, uint32_t max_dets, uint32_t seed) {
    Frifull *f = new Frifull();
    f->sys.n_orb = n_orb; f->sys.n_elec = n_elec;
    f->sys.ints.n_orb = n_orb;
    f->sys.ints.h.assign(h, h + (size_t)n_orb * n_orb);
    f->sys.ints.eri.assign(eris, eris + Integrals::packed_len(n_orb));
    f->sys.symm.init(irreps, n_orb);
    f->par.eps = eps; f->par.target_norm = target; f->par.vec_nonz = vec_nonz; f->par.max_dets = max_dets; f->par.seed = seed;
    f->setup();
    return f;
}
void fo_frifull_destroy(void *h) { delete (Frifull *)h; }
void fo_frifull_iterate(void *h, uint32_t n, OracleLog *logs) {
    Frifull *f = (Frifull *)h;
    for (uint32_t i = 0; i < n; i++) {
        f->iterate(1);
        if (logs) {
            const IterLog &l = f->log.back();
            OracleLog &o = logs[i];
            o.numer = l.numer; o.denom = l.denom; o.shift = l.shift; o.norm = l.norm; o.nkept = l.nkept;
            o.n_nonz = l.n_nonz; o.curr_size = (uint32_t)l.curr_size; o.num_success = (uint32_t)l.num_success;
            for (int k = 0; k < 5; k++) o.comp_len[k] = 0;
            o.err = 0;
        }
    }
}
// the current value column over positions [0, curr_size)
size_t fo_frifull_vec(void *h, uint64_t *dets, double *vals, size_t cap) {
    Frifull *f = (Frifull *)h;
    size_t n = f->sol.curr_size;
    if (cap >= n) for (size_t i = 0; i < n; i++) { dets[i] = f->sol.dets[i]; vals[i] = f->sol.vals[f->vec_idx][i]; }
    return n;
}
// compress_vecs with one vector (vec_utils.cpp:9-32): piv_comp_parallel on column 0 with the handle's generator, then the deletes
void fo_compress_vec_piv(void *h, uint32_t n_samp) {
    Frisys *f = (Frisys *)h;
    Vec &v = f->sol;
    if (f->srt.size() < v.max_size) { f->srt.resize(v.max_size); f->keep.resize(v.max_size, 0); }
    piv_comp_parallel(v.vals[0].data(), v.curr_size, n_samp, f->srt, f->keep, f->mt);
    for (size_t i = 0; i < v.curr_size; i++) if (f->keep[i]) { v.del_at_pos(i); f->keep[i] = 0; }
}
uint32_t fo_next_draw(void *h) { return (uint32_t)((Frisys *)h)->mt(); }
// piv_comp_parallel on a bare array with a generator seeded here; flags_out: elements to delete; returns the generator's next draw
uint32_t fo_piv_comp(double *vals, size_t len, uint32_t compress_size, uint32_t seed, uint8_t *flags_out) {
    std::mt19937 g(seed);
    std::vector<size_t> srt(len);
    std::vector<uint8_t> flag(len, 0);
    piv_comp_parallel(vals, len, compress_size, srt, flag, g);
    memcpy(flags_out, flag.data(), len);
    return (uint32_t)g();
}
// adjust_probs on a bare array with nothing preserved; flags_out: the elements it pinned
double fo_adjust_probs(double *vals, size_t len, uint32_t *n_samp_loc, double exp_nsamp_loc, uint32_t n_samp_tot, double tot_norm, uint8_t *flags_out) {
    std::vector<uint8_t> flag(len, 0);
    double r = adjust_probs(vals, len, n_samp_loc, exp_nsamp_loc, n_samp_tot, tot_norm, flag);
    memcpy(flags_out, flag.data(), len);
    return r;
}
// DistVec::add x n + perform_add(0) on the handle's stored vector (column 0)
void fo_vec_add(void *h, const uint64_t *dets, const double *vals, const uint8_t *ini, size_t n) {
    Frisys *f = (Frisys *)h;
    Vec &v = f->sol;
    v.cur = 0;
    for (size_t i = 0; i < n; i++) v.add(dets[i], vals[i], ini[i]);
    v.perform_add(0);
}
int fo_vec_info(void *h, uint32_t *curr_size, int32_t *n_nonz, uint32_t *n_free) {
    Frisys *f = (Frisys *)h;
    *curr_size = (uint32_t)f->sol.curr_size; *n_nonz = f->sol.n_nonz; *n_free = (uint32_t)f->sol.free_stack.size();
    return 0;
}
void fo_frisys_restart(void *h, uint32_t seed, double en_shift, double last_one_norm, uint32_t iterat) {
    Frisys *f = (Frisys *)h;
    f->mt.seed(seed); f->en_shift = en_shift; f->last_one_norm = last_one_norm; f->iterat = iterat;
}
uint64_t fo_hash(const uint8_t *occ, uint32_t n_elec, const uint32_t *scr) { return hash_fxn(occ, n_elec, scr); }


// ---- P in-process ranks (std::thread per rank) sharing one communicator: the reference under mpiexec -n P
struct OracleRanks { std::vector<std::unique_ptr<Frisys>> fr; };

static void fill_log(const IterLog &l, OracleLog &o) {
    o.numer = l.numer; o.denom = l.denom; o.shift = l.shift; o.norm = l.norm; o.nkept = l.nkept;
    o.n_nonz = l.n_nonz; o.curr_size = (uint32_t)l.curr_size; o.num_success = (uint32_t)l.num_success;
    for (int k = 0; k < 5; k++) o.comp_len[k] = (uint32_t)l.comp_len[k];
    o.err = 0;
}

static std::vector<det_t> g_ranks_det_space;      // consumed by the next fo_ranks_create
void fo_ranks_set_det_space(const uint64_t *space, size_t n) { g_ranks_det_space.assign(space, space + n); }
void *fo_ranks_create(uint32_t n_ranks, uint32_t n_orb, uint32_t n_elec, const uint8_t *irreps, const double *h, const double *eris,
                      double eps, double target, double init, uint32_t vec_nonz, uint32_t mat_nonz, uint32_t max_dets, uint32_t seed, int hb_unnorm) {
    OracleRanks *R = new OracleRanks();
    std::vector<det_t> space; space.swap(g_ranks_det_space);
    for (uint32_t r = 0; r < n_ranks; r++) {
        Frisys *f = new Frisys();
        f->det_space = space;
        f->sys.n_orb = n_orb; f->sys.n_elec = n_elec;
        f->sys.ints.n_orb = n_orb;
        f->sys.ints.h.assign(h, h + (size_t)n_orb * n_orb);
        f->sys.ints.eri.assign(eris, eris + Integrals::packed_len(n_orb));
        f->sys.symm.init(irreps, n_orb);
        f->par.eps = eps; f->par.target_norm = target; f->par.init_thresh = init;
        f->par.vec_nonz = vec_nonz; f->par.mat_nonz = mat_nonz; f->par.max_dets = max_dets;
        f->par.new_hb = hb_unnorm != 0; f->par.seed = seed;
        R->fr.emplace_back(f);
    }
    try {
        run_ranks((int)n_ranks, [&](const Comm &c) { Frisys &f = *R->fr[c.rank]; f.cm = c; f.setup(); });
    } catch (std::exception &e) { fprintf(stderr, "fo_ranks_create: %s\n", e.what()); delete R; return nullptr; }
    return R;
}
void fo_ranks_destroy(void *h) { delete (OracleRanks *)h; }
// logs: [n_ranks][n] row-major
int fo_ranks_iterate(void *h, uint32_t n, OracleLog *logs) {
    OracleRanks *R = (OracleRanks *)h;
    int P = (int)R->fr.size();
    try {
        run_ranks(P, [&](const Comm &c) {
            Frisys &f = *R->fr[c.rank];
            f.cm = c; f.sol.cm = c;
            for (uint32_t i = 0; i < n; i++) { f.iterate(1); if (logs) fill_log(f.log.back(), logs[(size_t)c.rank * n + i]); }
        });
    } catch (std::exception &e) { fprintf(stderr, "fo_ranks_iterate: %s\n", e.what()); return 1; }
    return 0;
}
// compress_vecs with one vector on every rank (piv_comp_parallel over the communicator, then the deletes); each rank draws from its own generator
int fo_ranks_compress_piv(void *h, uint32_t n_samp) {
    OracleRanks *R = (OracleRanks *)h;
    int P = (int)R->fr.size();
    try {
        run_ranks(P, [&](const Comm &c) {
            Frisys &f = *R->fr[c.rank];
            f.cm = c; f.sol.cm = c;
            Vec &v = f.sol;
            if (f.srt.size() < v.max_size) { f.srt.resize(v.max_size); f.keep.resize(v.max_size, 0); }
            piv_comp_parallel(v.vals[0].data(), v.curr_size, n_samp, f.srt, f.keep, f.mt, c);
            for (size_t i = 0; i < v.curr_size; i++) if (f.keep[i]) { v.del_at_pos(i); f.keep[i] = 0; }
        });
    } catch (std::exception &e) { fprintf(stderr, "fo_ranks_compress_piv: %s\n", e.what()); return 1; }
    return 0;
}
// apply_HBPP_piv over the ranks (every piv_comp_parallel inside is collective); returns this rank's samples
size_t fo_ranks_apply_hbpp_piv(void *h, uint32_t n_samp, uint32_t rank, uint32_t *pos, uint8_t *orbs, double *vals, size_t cap, uint64_t *stage_len) {
    OracleRanks *R = (OracleRanks *)h;
    int P = (int)R->fr.size();
    std::vector<HBPivScratch> ps(P);
    try {
        run_ranks(P, [&](const Comm &c) {
            Frisys &f = *R->fr[c.rank];
            f.cm = c; f.sol.cm = c;
            size_t n = f.sol.curr_size;
            size_t len = (n > (size_t)n_samp ? n : (size_t)n_samp) * 2 + 64;
            size_t ns = f.sys.n_elec > (f.sys.n_orb - f.sys.n_elec / 2) ? f.sys.n_elec : f.sys.n_orb - f.sys.n_elec / 2;
            HBPivScratch &s = ps[c.rank];
            s.init(len, ns);
            std::copy(f.sol.vals[0].begin(), f.sol.vals[0].begin() + n, s.vec1.begin());
            for (size_t i = 0; i < n; i++) s.det_idx1[i] = i;
            s.vec_len = n;
            apply_HBPP_piv(f.sol, s, f.sys, f.p_doub, f.par.new_hb, f.mt, n_samp, false, c);
        });
    } catch (std::exception &e) { fprintf(stderr, "fo_ranks_apply_hbpp_piv: %s\n", e.what()); return (size_t)-1; }
    HBPivScratch &s = ps[rank];
    size_t m = s.vec_len;
    if (cap >= m) for (size_t i = 0; i < m; i++) { pos[i] = (uint32_t)s.det_idx2[i]; memcpy(orbs + 4 * i, &s.orb1[4 * i], 4); vals[i] = s.vec1[i]; }
    if (stage_len) for (int k = 0; k < 5; k++) stage_len[k] = s.stage_len[k];
    return m;
}
void *fo_ranks_get(void *h, uint32_t rank) { return ((OracleRanks *)h)->fr[rank].get(); }   // a Frisys* for fo_frisys_vec etc.
int fo_ranks_hf_proc(void *h) { return ((OracleRanks *)h)->fr[0]->hf_proc; }
int fo_idx_to_proc(void *h, uint64_t det) { return ((Frisys *)h)->sol.idx_to_proc(det); }


// ---- frisys_hh: P in-process ranks (P = 1: plain)
struct OracleHH { std::vector<std::unique_ptr<FrisysHH>> fr; };
void *fo_hh_create(uint32_t n_ranks, uint32_t n_elec, uint32_t n_sites, double eps, double U, double omega, double g, double gs_energy,
                   double target, double init, uint32_t vec_nonz, uint32_t max_dets, uint32_t seed, int full) {
    OracleHH *R = new OracleHH();
    for (uint32_t r = 0; r < n_ranks; r++) {
        FrisysHH *f = new FrisysHH();
        f->par.n_elec = n_elec; f->par.n_sites = n_sites; f->par.eps = eps; f->par.U = U; f->par.omega = omega; f->par.g = g; f->par.hf_en = gs_energy;
        f->par.target_norm = target; f->par.init_thresh = init; f->par.vec_nonz = vec_nonz; f->par.max_dets = max_dets; f->par.seed = seed;
        f->full = full != 0;       // frifull_hh instead of frisys_hh
        R->fr.emplace_back(f);
    }
    try {
        run_ranks((int)n_ranks, [&](const Comm &c) { FrisysHH &f = *R->fr[c.rank]; f.cm = c; f.setup(); });
    } catch (std::exception &e) { fprintf(stderr, "fo_hh_create: %s\n", e.what()); delete R; return nullptr; }
    return R;
}
void fo_hh_destroy(void *h) { delete (OracleHH *)h; }
int fo_hh_iterate(void *h, uint32_t n, OracleLog *logs) {
    OracleHH *R = (OracleHH *)h;
    int P = (int)R->fr.size();
    try {
        run_ranks(P, [&](const Comm &c) {
            FrisysHH &f = *R->fr[c.rank];
            f.cm = c; f.sol.cm = c;
            for (uint32_t i = 0; i < n; i++) {
                f.iterate(1);
                if (logs) {
                    const HHLog &l = f.log.back();
                    OracleLog &o = logs[(size_t)c.rank * n + i];
                    o.numer = l.numer; o.denom = l.denom; o.shift = l.shift; o.norm = l.norm; o.nkept = l.nkept; o.n_nonz = l.n_nonz;
                    o.curr_size = (uint32_t)l.curr_size; o.num_success = (uint32_t)l.num_success;
                    for (int k = 0; k < 5; k++) o.comp_len[k] = 0;
                    o.err = 0;
                }
            }
        });
    } catch (std::exception &e) { fprintf(stderr, "fo_hh_iterate: %s\n", e.what()); return 1; }
    return 0;
}
size_t fo_hh_vec(void *h, uint32_t rank, uint64_t *dets, double *vals, size_t cap) {
    FrisysHH *f = ((OracleHH *)h)->fr[rank].get();
    size_t n = f->sol.curr_size;
    if (dets && vals && cap >= n) for (size_t i = 0; i < n; i++) { dets[i] = f->sol.dets[i]; vals[i] = f->sol.vals[0][i]; }
    return n;
}
int fo_hh_ref_proc(void *h) { return ((OracleHH *)h)->fr[0]->ref_proc; }
uint64_t fo_hh_neel(void *h) { return ((OracleHH *)h)->fr[0]->neel; }


// ---- fciqmc_mol (near-uniform); counter_rng = 0: the reference's mt19937 stream, 1: the counter-based stream the GPU can replay
struct FqLog { double numer, denom, shift, norm; int32_t n_nonz; uint32_t n_ini, curr_size, n_spawn; };
void *fo_fciqmc_create(uint32_t n_orb, uint32_t n_elec, const uint8_t *irreps, const double *h, const double *eris,
                       double eps, uint32_t target_walkers, uint32_t init_thresh, uint32_t max_dets, uint32_t seed, int counter_rng) {
    Fciqmc *f = new Fciqmc();
    f->par.heat_bath = (counter_rng & 2) != 0; f->par.fp = (counter_rng & 4) != 0; counter_rng &= 1;        // bit 1 of the flag selects --distribution HB, bit 2 fciqmc_fp_mol
    f->sys.n_orb = n_orb; f->sys.n_elec = n_elec;
    f->sys.ints.n_orb = n_orb;
    f->sys.ints.h.assign(h, h + (size_t)n_orb * n_orb);
    f->sys.ints.eri.assign(eris, eris + Integrals::packed_len(n_orb));
    f->sys.symm.init(irreps, n_orb);
    f->par.eps = eps; f->par.target_walkers = target_walkers; f->par.init_thresh = init_thresh; f->par.max_dets = max_dets; f->par.seed = seed;
    f->par.counter_rng = counter_rng != 0;
    f->setup();
    return f;
}
// the same with --trial_vec / --ini_vec (n == 0: default)
void *fo_fciqmc_create_ex(uint32_t n_orb, uint32_t n_elec, const uint8_t *irreps, const double *h, const double *eris,
                          double eps, uint32_t target_walkers, uint32_t init_thresh, uint32_t max_dets, uint32_t seed, int flags,
                          const uint64_t *tr_det, const double *tr_val, size_t n_tr, const uint64_t *in_det, const int32_t *in_val, size_t n_in) {
    Fciqmc *f = new Fciqmc();
    f->par.heat_bath = (flags & 2) != 0; f->par.counter_rng = (flags & 1) != 0; f->par.fp = (flags & 4) != 0;
    f->sys.n_orb = n_orb; f->sys.n_elec = n_elec;
    f->sys.ints.n_orb = n_orb;
    f->sys.ints.h.assign(h, h + (size_t)n_orb * n_orb);
    f->sys.ints.eri.assign(eris, eris + Integrals::packed_len(n_orb));
    f->sys.symm.init(irreps, n_orb);
    f->par.eps = eps; f->par.target_walkers = target_walkers; f->par.init_thresh = init_thresh; f->par.max_dets = max_dets; f->par.seed = seed;
    if (n_tr) { f->trial_in_det.assign(tr_det, tr_det + n_tr); f->trial_in_val.assign(tr_val, tr_val + n_tr); }
    if (n_in) { f->ini_det.assign(in_det, in_det + n_in); f->ini_val.assign(in_val, in_val + n_in); }
    f->setup();
    return f;
}
// the general form: fciqmc_mol / fciqmc_fp_mol (flags bit 2) / frimulti_mol (flags bit 3: vec_nonz, mat_nonz, initiator_f, target_norm apply) with
// --trial_vec / --ini_vec, the initial vector's values as doubles (n == 0: default)
void *fo_fq_create_vecs(uint32_t n_orb, uint32_t n_elec, const uint8_t *irreps, const double *h, const double *eris,
                        double eps, uint32_t target_walkers, uint32_t init_thresh, uint32_t max_dets, uint32_t seed, int flags,
                        uint32_t vec_nonz, uint32_t mat_nonz, double initiator_f, double target_norm,
                        const uint64_t *tr_det, const double *tr_val, size_t n_tr, const uint64_t *in_det, const double *in_val, size_t n_in) {
    Fciqmc *f = new Fciqmc();
    f->par.heat_bath = (flags & 2) != 0; f->par.counter_rng = (flags & 1) != 0; f->par.fp = (flags & 4) != 0;
    f->sys.n_orb = n_orb; f->sys.n_elec = n_elec;
    f->sys.ints.n_orb = n_orb;
    f->sys.ints.h.assign(h, h + (size_t)n_orb * n_orb);
    f->sys.ints.eri.assign(eris, eris + Integrals::packed_len(n_orb));
    f->sys.symm.init(irreps, n_orb);
    f->par.eps = eps; f->par.target_walkers = target_walkers; f->par.init_thresh = init_thresh; f->par.max_dets = max_dets; f->par.seed = seed;
    if (flags & 8) { f->par.heat_bath = true; f->par.multi = true; f->par.vec_nonz = vec_nonz; f->par.mat_nonz = mat_nonz; f->par.init_thresh_f = initiator_f; f->par.target_norm = target_norm; }
    if (n_tr) { f->trial_in_det.assign(tr_det, tr_det + n_tr); f->trial_in_val.assign(tr_val, tr_val + n_tr); }
    if (n_in) { f->ini_det.assign(in_det, in_det + n_in); f->ini_val.assign(in_val, in_val + n_in); }
    try { f->setup(); } catch (std::exception &e) { g_last_error = e.what(); delete f; return nullptr; }       // (frimulti_mol refuses every --trial_vec on one rank)
    return f;
}
const char *fo_last_error() { return g_last_error.c_str(); }
// frimulti_mol: the same object in its multinomial mode (fo::Fciqmc::iterate_multi); flags bit 0 = counter-based uniforms
void *fo_frimulti_create(uint32_t n_orb, uint32_t n_elec, const uint8_t *irreps, const double *h, const double *eris,
                         double eps, uint32_t vec_nonz, uint32_t mat_nonz, uint32_t max_dets, uint32_t seed, double initiator, double target_norm, int flags) {
    Fciqmc *f = new Fciqmc();
    f->par.heat_bath = true; f->par.counter_rng = (flags & 1) != 0;
    f->sys.n_orb = n_orb; f->sys.n_elec = n_elec;
    f->sys.ints.n_orb = n_orb;
    f->sys.ints.h.assign(h, h + (size_t)n_orb * n_orb);
    f->sys.ints.eri.assign(eris, eris + Integrals::packed_len(n_orb));
    f->sys.symm.init(irreps, n_orb);
    f->par.eps = eps; f->par.max_dets = max_dets; f->par.seed = seed;
    f->par.multi = true; f->par.vec_nonz = vec_nonz; f->par.mat_nonz = mat_nonz; f->par.init_thresh_f = initiator; f->par.target_norm = target_norm;
    f->setup();
    return f;
}
uint32_t fo_frimulti_nkept(void *h) { return ((Fciqmc *)h)->nkept; }
// frimulti_mol over P in-process ranks: an OracleFqRanks whose members run iterate_multi (fo_fqranks_iterate dispatches on par.multi)
void *fo_multiranks_create(uint32_t n_ranks, uint32_t n_orb, uint32_t n_elec, const uint8_t *irreps, const double *h, const double *eris,
                           double eps, uint32_t vec_nonz, uint32_t mat_nonz, uint32_t max_dets, uint32_t seed, double initiator, double target_norm, int flags);
void fo_fciqmc_destroy(void *h) { delete (Fciqmc *)h; }
int fo_fciqmc_iterate(void *h, uint32_t n, FqLog *logs) {
    Fciqmc *f = (Fciqmc *)h;
    try {
        for (uint32_t i = 0; i < n; i++) {
            if (f->par.multi) f->iterate_multi(1); else f->iterate(1);
            if (logs) {
                const FciqmcLog &l = f->log.back();
                logs[i].numer = l.numer; logs[i].denom = l.denom; logs[i].shift = l.shift; logs[i].norm = l.norm; logs[i].n_nonz = l.n_nonz;
                logs[i].n_ini = l.n_ini; logs[i].curr_size = (uint32_t)l.curr_size; logs[i].n_spawn = (uint32_t)l.n_spawn;
            }
        }
    } catch (std::exception &e) { fprintf(stderr, "fo_fciqmc_iterate: %s\n", e.what()); return 1; }
    return 0;
}
size_t fo_fciqmc_vec(void *h, uint64_t *dets, double *vals, size_t cap) {
    Fciqmc *f = (Fciqmc *)h;
    size_t n = f->sol.curr_size;
    if (dets && vals && cap >= n) for (size_t i = 0; i < n; i++) { dets[i] = f->sol.dets[i]; vals[i] = f->sol.vals[0][i]; }
    return n;
}
double fo_fciqmc_p_doub(void *h) { return ((Fciqmc *)h)->p_doub; }
// replaces the stored walkers: determinants land in positions 0..n-1 (what fries_vec_load does on the device), then the part of a
// restart that is not the vector: shift, walker number at the last shift update, iteration count (the counter stream is keyed by it)
void fo_fciqmc_load(void *h, const uint64_t *dets, const double *vals, size_t n, double en_shift, double last_norm, uint32_t iterat) {
    Fciqmc *f = (Fciqmc *)h;
    Vec &v = f->sol;
    const size_t cap = v.max_size, ac = v.adder_cap;
    const unsigned nv = v.n_vecs;
    const Comm cm = v.cm; const uint32_t *ps = v.proc_scr;
    v.init(cap, ac, f->sys.n_elec, nv, cm, ps);
    uint8_t tmp[64];
    for (size_t i = 0; i < n; i++) {
        v.dets[i] = dets[i]; v.vals[0][i] = vals[i]; v.active[i] = 1; v.table[dets[i]] = (ptrdiff_t)i;
        occ_list(dets[i], tmp); memcpy(&v.occ[i * v.n_elec], tmp, v.n_elec);
    }
    v.curr_size = n; v.n_nonz = (int)n;
    f->en_shift = en_shift; f->last_norm = last_norm; f->iterat = iterat;
}

// ---- fciqmc_mol on P in-process ranks (the reference under mpiexec -n P: own generator per rank, one all-to-all per iteration)
struct OracleFqRanks { std::vector<std::unique_ptr<Fciqmc>> fr; };
void *fo_fqranks_create(uint32_t n_ranks, uint32_t n_orb, uint32_t n_elec, const uint8_t *irreps, const double *h, const double *eris,
                        double eps, uint32_t target_walkers, uint32_t init_thresh, uint32_t max_dets, uint32_t seed, int flags) {
    OracleFqRanks *R = new OracleFqRanks();
    for (uint32_t r = 0; r < n_ranks; r++) {
        Fciqmc *f = new Fciqmc();
        f->par.heat_bath = (flags & 2) != 0; f->par.counter_rng = (flags & 1) != 0; f->par.fp = (flags & 4) != 0;
        f->sys.n_orb = n_orb; f->sys.n_elec = n_elec;
        f->sys.ints.n_orb = n_orb;
        f->sys.ints.h.assign(h, h + (size_t)n_orb * n_orb);
        f->sys.ints.eri.assign(eris, eris + Integrals::packed_len(n_orb));
        f->sys.symm.init(irreps, n_orb);
        f->par.eps = eps; f->par.target_walkers = target_walkers; f->par.init_thresh = init_thresh; f->par.max_dets = max_dets; f->par.seed = seed;
        R->fr.emplace_back(f);
    }
    try {
        run_ranks((int)n_ranks, [&](const Comm &c) { Fciqmc &f = *R->fr[c.rank]; f.cm = c; f.setup(); });
    } catch (std::exception &e) { fprintf(stderr, "fo_fqranks_create: %s\n", e.what()); delete R; return nullptr; }
    return R;
}
void fo_fqranks_destroy(void *h) { delete (OracleFqRanks *)h; }
// logs: [n_ranks][n] row-major
int fo_fqranks_iterate(void *h, uint32_t n, FqLog *logs) {
    OracleFqRanks *R = (OracleFqRanks *)h;
    int P = (int)R->fr.size();
    try {
        run_ranks(P, [&](const Comm &c) {
            Fciqmc &f = *R->fr[c.rank];
            f.cm = c; f.sol.cm = c;
            for (uint32_t i = 0; i < n; i++) {
                if (f.par.multi) f.iterate_multi(1); else f.iterate(1);
                if (logs) {
                    const FciqmcLog &l = f.log.back();
                    FqLog &o = logs[(size_t)c.rank * n + i];
                    o.numer = l.numer; o.denom = l.denom; o.shift = l.shift; o.norm = l.norm; o.n_nonz = l.n_nonz;
                    o.n_ini = l.n_ini; o.curr_size = (uint32_t)l.curr_size; o.n_spawn = (uint32_t)l.n_spawn;
                }
            }
        });
    } catch (std::exception &e) { fprintf(stderr, "fo_fqranks_iterate: %s\n", e.what()); return 1; }
    return 0;
}
void *fo_multiranks_create(uint32_t n_ranks, uint32_t n_orb, uint32_t n_elec, const uint8_t *irreps, const double *h, const double *eris,
                           double eps, uint32_t vec_nonz, uint32_t mat_nonz, uint32_t max_dets, uint32_t seed, double initiator, double target_norm, int flags) {
    OracleFqRanks *R = new OracleFqRanks();
    for (uint32_t r = 0; r < n_ranks; r++) {
        Fciqmc *f = new Fciqmc();
        f->par.heat_bath = true; f->par.counter_rng = (flags & 1) != 0;
        f->sys.n_orb = n_orb; f->sys.n_elec = n_elec;
        f->sys.ints.n_orb = n_orb;
        f->sys.ints.h.assign(h, h + (size_t)n_orb * n_orb);
        f->sys.ints.eri.assign(eris, eris + Integrals::packed_len(n_orb));
        f->sys.symm.init(irreps, n_orb);
        f->par.eps = eps; f->par.max_dets = max_dets; f->par.seed = seed;
        f->par.multi = true; f->par.vec_nonz = vec_nonz; f->par.mat_nonz = mat_nonz; f->par.init_thresh_f = initiator; f->par.target_norm = target_norm;
        R->fr.emplace_back(f);
    }
    try {
        run_ranks((int)n_ranks, [&](const Comm &c) { Fciqmc &f = *R->fr[c.rank]; f.cm = c; f.setup(); });
    } catch (std::exception &e) { fprintf(stderr, "fo_multiranks_create: %s\n", e.what()); delete R; return nullptr; }
    return R;
}
void *fo_fqranks_get(void *h, uint32_t rank) { return ((OracleFqRanks *)h)->fr[rank].get(); }      // a Fciqmc* for fo_fciqmc_vec
int fo_fqranks_hf_proc(void *h) { return ((OracleFqRanks *)h)->fr[0]->hf_proc; }

}  // extern "C"
