// Pivotal compression of the stored vector: piv_comp_parallel = find_preserve + piv_budget + adjust_probs +
// piv_samp_serial (FRIES/compress_utils.cpp:354-681), the compression behind compress_vecs (FRIES/vec_utils.cpp:9-32).
//
// piv_samp_serial cuts the unpreserved elements, in storage order, into consecutive sampling units of weight
// seg_norm / n_samp and draws two uniforms per unit.  Where a unit ends depends on a running floating-point sum that
// carries the overshoot of one unit into the next, so the cut points are found by one wave that walks the vector in
// order (k_piv_chain: one add and one compare per element, elements fetched 64 at a time).  Everything else is
// parallel: a lane per sampling unit redoes the unit's arithmetic in the reference's order from the exact carried-in
// value, picks the candidate and decides who is sampled (k_piv_unit); the "residual" element handed from unit to unit
// is resolved by a short backward walk over the unit records (k_piv_resid); the tail and the deletes are elementwise.
// Results are the reference's bit for bit; the uniforms are the engine's mt19937 stream, two per unit, in unit order.
#include "ctx.hpp"

static void piv_alloc(FriesCtx *c, PivBuf &P, uint32_t cap) {
    if (P.cap >= cap) return;
    if (P.start) { FR_HIP(hipFree(P.start)); FR_HIP(hipFree(P.carry)); FR_HIP(hipFree(P.U)); FR_HIP(hipFree(P.unit)); FR_HIP(hipFree(P.scal)); }
    P.start = fr_alloc<uint32_t>(cap); P.carry = fr_alloc<double>(cap); P.U = fr_alloc<double>(2 * (size_t)cap); P.unit = fr_alloc<PivUnit>(cap);
    P.scal = fr_alloc<PivScal>(1);
    P.cap = cap;
}

__device__ __forceinline__ double fr_readlane_f64(double v, int lane) {
    long long b = __double_as_longlong(v);
    int lo = __builtin_amdgcn_readlane((int)b, lane), hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}

// adjust_probs, first loop (compress_utils.cpp:614-619): is any unpreserved element as large as the local sampling unit?
__global__ void __launch_bounds__(FR_BLOCK) k_piv_toobig(VecDev V, VcompBuf B, PivScal *S, double thr) {
    const uint32_t n = V.st->curr_size;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool big = i < n && !B.keep[i] && fabs(V.v0[i]) >= thr;
    if (__any(big) && fr_lane() == 0) atomicOr(&S->too_big, 1u);
}

// adjust_probs, second part (compress_utils.cpp:620-676): a sequential walk that stops as soon as the running expected
// count meets the local budget.  One wave, values fetched 64 at a time, every lane carrying the same scalars.
__global__ void __launch_bounds__(64) k_piv_adjust(VecDev V, VcompBuf B, PivScal *S, uint32_t n_loc_in, double exp_loc, double unit) {
    if (!S->too_big) { if (threadIdx.x == 0) S->n_loc = n_loc_in; return; }
    const uint32_t n = V.st->curr_size;
    const int lane = threadIdx.x;
    const double resid = exp_loc - (unsigned int)exp_loc;
    double counter = exp_loc;
    uint32_t n_loc = n_loc_in;
    const bool up = n_loc > exp_loc;
    bool stop = false;
    for (uint32_t base = 0; base < n && !stop; base += 64) {
        const uint32_t i = base + lane;
        double v = i < n ? V.v0[i] : 0.0;
        uint8_t kp = i < n ? B.keep[i] : 1;
        double nv = v; uint8_t nk = kp;
        for (int j = 0; j < 64 && base + j < n; j++) {
            const double vj = fr_readlane_f64(v, j);
            const int kj = __builtin_amdgcn_readlane((int)kp, j);
            if (kj) continue;
            const int sg = 2 * (vj > 0) - 1;
            const double pi = fabs(vj) / unit;
            double out;
            int knew = 0;
            if (up) {
                if (pi < resid) { counter += pi / resid - pi; out = vj / resid; }
                else { counter -= pi; out = sg * unit; knew = 1; n_loc--; }
                if (counter >= n_loc) { out += sg * unit * (n_loc - counter); stop = true; }
            }
            else {
                if (pi > resid) { double q = (pi - resid) / (1 - resid); counter += q - pi; out = sg * q * unit; }
                else { counter -= pi; out = 0; }
                if (counter <= n_loc) { out += sg * unit * (n_loc - counter); stop = true; }
            }
            if (lane == j) { nv = out; if (knew) nk = 1; }
            if (stop) break;
        }
        if (i < n) { V.v0[i] = nv; B.keep[i] = nk; }
    }
    if (lane == 0) S->n_loc = n_loc;
}

// n_samp == 0 (compress_utils.cpp:391-403)
__global__ void __launch_bounds__(FR_BLOCK) k_piv_none(VecDev V, VcompBuf B) {
    const uint32_t n = V.st->curr_size;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = V.v0[i];
    if (B.keep[i]) B.keep[i] = 0;
    else { v = 0; V.v0[i] = 0; }
    if (v == 0) B.del[i] = 1;
}

// Where the sampling units begin (compress_utils.cpp:409-424, :503-505): cum starts from the carried overshoot, grows by
// the unpreserved magnitudes in order, and the element that takes it to >= unit closes the unit.  The recurrence
//     t = cum + w_i;  cum = t >= unit ? t - unit : t
// is the whole dependence between units, so this kernel evaluates only that: one wave, 64 elements per step from
// registers (four batches of loads in flight), the step fully unrolled with constant-lane reads; every lane carries the
// same running value and lane j keeps the overshoot produced at element j.  Unit records are written once per batch.
#define FR_PIV_AHEAD 4
__global__ void __launch_bounds__(64) k_piv_chain(VecDev V, VcompBuf B, PivBuf P, double unit, uint32_t n_samp, uint32_t *err) {
    const uint32_t n = V.st->curr_size;
    const int lane = threadIdx.x;
    if (n == 0 || n_samp == 0) { if (lane == 0) { P.scal->n_units = 0; P.scal->end_pos = n; } return; }
    if (lane == 0) { P.start[0] = 0; P.carry[0] = 0; }
    uint32_t n_cross = 0;       // units closed so far; crossing number m closes unit m - 1 and (if it exists) opens unit m
    uint32_t n_units = 1, end_pos = n;
    double cum = 0;
    auto fetch = [&](uint32_t base) -> double {
        const uint32_t i = base + lane;
        return (i < n && !B.keep[i]) ? fabs(V.v0[i]) : 0.0;      // preserved elements are skipped; adding zero changes nothing
    };
    double q[FR_PIV_AHEAD];
#pragma unroll
    for (int a = 0; a < FR_PIV_AHEAD; a++) q[a] = fetch((uint32_t)a * 64);
    bool stop = false;
    for (uint32_t base = 0; base < n && !stop; base += 64) {
        const double w = q[0];
#pragma unroll
        for (int a = 0; a + 1 < FR_PIV_AHEAD; a++) q[a] = q[a + 1];
        q[FR_PIV_AHEAD - 1] = fetch(base + FR_PIV_AHEAD * 64);
        unsigned long long mask = 0;
        double mine = 0;
#pragma unroll
        for (int j = 0; j < 64; j++) {
            const double wj = fr_readlane_f64(w, j);
            const double t = cum + wj;
            const double ov = t - unit;
            const bool cr = t >= unit;
            cum = cr ? ov : t;
            if (lane == j) mine = ov;
            if (cr) mask |= 1ull << j;
        }
        if (mask) {
            const bool crossed = (mask >> lane) & 1ull;
            const uint32_t m = n_cross + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull)) + 1;      // my crossing's number
            const uint32_t nxt = base + lane + 1;
            if (crossed && mine >= unit) atomicOr(err, FR_ERR_PIV);       // an unpreserved element of two units or more: outside the contract
            if (crossed && m < n_samp && nxt < n) { P.start[m] = nxt; P.carry[m] = mine; }
            const unsigned long long opened = __ballot(crossed && m < n_samp && nxt < n);
            n_units += (uint32_t)__popcll(opened);
            const unsigned long long last = __ballot(crossed && m == n_samp);        // the budget's last unit closes here
            if (last) { end_pos = base + (uint32_t)__builtin_ctzll(last) + 1; if (end_pos > n) end_pos = n; stop = true; }
            n_cross += (uint32_t)__popcll(mask);
        }
    }
    if (lane == 0) { P.scal->n_units = n_units; P.scal->end_pos = end_pos; }
}

__device__ __forceinline__ double fr_sgn_unit(double unit, double v) { return unit * ((v > 0) - (v < 0)); }

// One sampling unit (compress_utils.cpp:409-502 without the residual bookkeeping, which k_piv_resid does)
__global__ void __launch_bounds__(FR_BLOCK) k_piv_unit(VecDev V, VcompBuf B, PivBuf P, double unit, uint32_t *err) {
    const uint32_t n = V.st->curr_size;
    const uint32_t n_units = P.scal->n_units;
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_units) return;
    const uint32_t pos = P.start[k];
    const double carry = P.carry[k];
    // the unit's elements: cum in the reference's order
    double cum = carry, last = 0;
    uint32_t used = 0, n_wt = 1;
    while (cum < unit && pos + used < n) {
        if (!B.keep[pos + used]) { last = fabs(V.v0[pos + used]); cum += last; n_wt++; }
        used++;
    }
    const bool at_end = pos + used == n;
    if (used == 0) { atomicOr(err, FR_ERR_PIV); return; }
    uint32_t n_inner = used - 1;
    if (at_end) n_inner++;
    const double over = cum - unit;
    if (!at_end) { n_wt--; cum -= last; }
    const double under = unit - cum;
    // candidate among the residual piece and the inner elements (:437-446)
    double r = P.U[2 * (size_t)k] * cum;
    double run = 0;
    uint32_t H = 0, h_idx = FR_NOPOS, e = pos;
    if (run < r && H < n_wt) { run += carry; H++; }
    while (run < r && H < n_wt) {
        while (B.keep[e]) e++;
        run += fabs(V.v0[e]); h_idx = e; e++; H++;
    }
    if (r > 0) H--;
    if (H == 0) h_idx = FR_NOPOS;
    double p_pass = under / (unit - over);
    if (at_end) p_pass = 0;
    const bool pass = P.U[2 * (size_t)k + 1] < p_pass;
    // the first unit's residual is element 0 itself, and the reference touches it before the unit's own elements (:478-480)
    if (k == 0 && !pass && H == 0) V.v0[0] = fr_sgn_unit(unit, V.v0[0]);
    uint32_t cnt = 1;
    for (uint32_t o = 0; o < n_inner; o++) {
        const uint32_t i = pos + o;
        if (!B.keep[i]) {
            if (cnt == H) { if (!pass) V.v0[i] = fr_sgn_unit(unit, V.v0[i]); }
            else { V.v0[i] = 0; B.del[i] = 1; }
            cnt++;
        }
        else B.keep[i] = 0;
    }
    if (pass) V.v0[pos + n_inner] = fr_sgn_unit(unit, V.v0[pos + n_inner]);
    PivUnit u;
    u.H = H; u.pass = pass ? 1 : 0; u.pad[0] = u.pad[1] = u.pad[2] = 0;
    u.new_resid = pass ? h_idx : pos + n_inner;      // pass with H == 0 hands on the residual it received (FR_NOPOS here)
    P.unit[k] = u;
}

// The residual element of unit k is whatever the nearest earlier unit handed on (:447-450, :478-480), and the last
// one is zeroed at the end (:515-518).  Thread k handles unit k; thread n_units the final residual.
__global__ void __launch_bounds__(FR_BLOCK) k_piv_resid(VecDev V, VcompBuf B, PivBuf P, double unit) {
    const uint32_t n = V.st->curr_size;
    const uint32_t n_units = P.scal->n_units;
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0 || k > n_units) return;
    bool zero, samp = false;
    if (k < n_units) {
        const PivUnit u = P.unit[k];
        zero = u.H != 0;
        samp = u.H == 0 && !u.pass;
        if (!zero && !samp) return;
    }
    else zero = true;
    uint32_t resid = 0;         // before any hand-over: element 0 (:408)
    for (uint32_t j = k; j-- > 0;) {
        const PivUnit u = P.unit[j];
        if (!(u.pass && u.H == 0)) { resid = u.new_resid; break; }
    }
    if (resid >= n) return;
    if (zero) { V.v0[resid] = 0; B.del[resid] = 1; }
    else if (samp) V.v0[resid] = fr_sgn_unit(unit, V.v0[resid]);
}

// elements after the last unit (:506-513)
__global__ void __launch_bounds__(FR_BLOCK) k_piv_tail(VecDev V, VcompBuf B, PivBuf P) {
    const uint32_t n = V.st->curr_size;
    const uint32_t i = P.scal->end_pos + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (!B.keep[i]) { V.v0[i] = 0; B.del[i] = 1; }
    else B.keep[i] = 0;
}

void fr_unkept_norm(FriesCtx *c, uint32_t bound);      // compress.hip

// compress_utils.cpp:552-604 with one rank (the scatter is the identity); draws from mt only if samples are left over
static uint32_t piv_budget_one_rank(double loc_norm, uint32_t n_samp) {
    double glob = 0;
    glob += loc_norm;
    uint32_t budget = loc_norm / glob * n_samp;
    uint32_t tot = budget;
    double frac = loc_norm - budget * glob / n_samp;
    if (frac < 1e-12) frac = 0;
    uint32_t n_frac = frac > 0 ? 1 : 0;
    if (n_frac == n_samp - tot) { if (frac > 0) budget++; tot = n_samp; }
    if (tot < n_samp) throw FriesError("pivotal budgeting left samples unassigned on a single rank");
    return budget;
}

void fr_piv_comp(FriesCtx *c, uint32_t compress_size, uint32_t *n_kept, double *glob_norm_out) {
    if (c->n_ranks > 1) throw FriesError("pivotal compression is built for one rank in this version");
    VcompBuf &B = c->vc;
    hipStream_t st = c->stream;
    fr_vec_sync_state(c, &c->vec, &c->h_vst);
    const uint32_t bound = c->h_vst.curr_size ? c->h_vst.curr_size : 1;
    const uint32_t save = c->vec_nonz;
    c->vec_nonz = compress_size;
    fr_death_clone(c, 0);                    // publishes the |v| block sums; column 1 is zero so values are unchanged
    c->vec_nonz = save;
    uint32_t n_samp = compress_size;
    double gn = 0;
    fr_find_preserve(c, &n_samp, &gn);
    if (n_kept) *n_kept = compress_size - n_samp;
    if (glob_norm_out) *glob_norm_out = gn;
    PivBuf &P = c->piv;       // created on first use
    piv_alloc(c, P, c->vec.cap);
    FR_HIP(hipMemsetAsync(P.scal, 0, sizeof(PivScal), st));
    double mine = 0;
    if (n_samp) {
        fr_unkept_norm(c, bound);
        FR_HIP(hipMemcpyAsync(&mine, B.seq.total, 8, hipMemcpyDeviceToHost, st));
        FR_HIP(hipStreamSynchronize(st));
    }
    double glob = 0;
    glob += mine;                                       // compress_utils.cpp:368-371
    uint32_t loc_samp = 0;
    double new_norm = 0;
    if (n_samp != 0) {
        loc_samp = piv_budget_one_rank(mine, n_samp);
        // adjust_probs (:606-681)
        const double exp_loc = n_samp * mine / glob;
        const double top = ceill(exp_loc);
        const double unit_t = glob / n_samp;
        const double loc_norm = exp_loc * unit_t;
        FR_LAUNCH(c, "k_piv_toobig", k_piv_toobig, dim3(fr_blocks(bound, FR_BLOCK)), dim3(FR_BLOCK), c->vec, B, P.scal, loc_norm / top);
        FR_LAUNCH(c, "k_piv_adjust", k_piv_adjust, dim3(1), dim3(64), c->vec, B, P.scal, loc_samp, exp_loc, unit_t);
        PivScal hs;
        FR_HIP(hipMemcpyAsync(&hs, P.scal, sizeof(hs), hipMemcpyDeviceToHost, st));
        FR_HIP(hipStreamSynchronize(st));
        if (hs.too_big) { loc_samp = hs.n_loc; new_norm = loc_samp * loc_norm / exp_loc; }
        else new_norm = loc_norm;
    }
    // piv_samp_serial(vals, len, new_norm, loc_samp, keep, mt) (:389-518)
    if (loc_samp == 0) FR_LAUNCH(c, "k_piv_none", k_piv_none, dim3(fr_blocks(bound, FR_BLOCK)), dim3(FR_BLOCK), c->vec, B);
    else {
        const double unit = new_norm / loc_samp;
        FR_LAUNCH(c, "k_piv_chain", k_piv_chain, dim3(1), dim3(64), c->vec, B, P, unit, loc_samp, c->d_err);
        PivScal hs;
        FR_HIP(hipMemcpyAsync(&hs, P.scal, sizeof(hs), hipMemcpyDeviceToHost, st));
        FR_HIP(hipStreamSynchronize(st));
        if (hs.n_units > P.cap) throw FriesError("pivotal compression: more sampling units than the work arrays hold");
        if (hs.n_units) {
            std::vector<double> u(2 * (size_t)hs.n_units);
            for (auto &x : u) x = c->mt() / (1. + UINT32_MAX);          // two per unit, in unit order (:437, :457)
            FR_HIP(hipMemcpyAsync(P.U, u.data(), u.size() * 8, hipMemcpyHostToDevice, st));
            FR_HIP(hipStreamSynchronize(st));
            FR_LAUNCH(c, "k_piv_unit", k_piv_unit, dim3(fr_blocks(hs.n_units, FR_BLOCK)), dim3(FR_BLOCK), c->vec, B, P, unit, c->d_err);
            FR_LAUNCH(c, "k_piv_resid", k_piv_resid, dim3(fr_blocks((size_t)hs.n_units + 1, FR_BLOCK)), dim3(FR_BLOCK), c->vec, B, P, unit);
        }
        if (hs.end_pos < bound) FR_LAUNCH(c, "k_piv_tail", k_piv_tail, dim3(fr_blocks(bound - hs.end_pos, FR_BLOCK)), dim3(FR_BLOCK), c->vec, B, P);
    }
    fr_vec_delete_flagged(c, &c->vec, B.del, bound);         // compress_vecs: vec_utils.cpp:25-30
    fr_vec_sync_state(c, &c->vec, &c->h_vst);
}

// test hook: adjust_probs alone on column 0 with nothing preserved (compress_utils.cpp:606-681)
void fr_test_piv_adjust(FriesCtx *c, uint32_t *n_loc_io, double exp_loc, uint32_t n_tot, double tot_norm, double *new_norm, uint8_t *flags_out) {
    VcompBuf &B = c->vc;
    hipStream_t st = c->stream;
    fr_vec_sync_state(c, &c->vec, &c->h_vst);
    const uint32_t bound = c->h_vst.curr_size ? c->h_vst.curr_size : 1;
    PivBuf &P = c->piv;
    piv_alloc(c, P, c->vec.cap);
    FR_HIP(hipMemsetAsync(P.scal, 0, sizeof(PivScal), st));
    const double top = ceill(exp_loc);
    const double unit_t = tot_norm / n_tot;
    const double loc_norm = exp_loc * unit_t;
    FR_LAUNCH(c, "k_piv_toobig", k_piv_toobig, dim3(fr_blocks(bound, FR_BLOCK)), dim3(FR_BLOCK), c->vec, B, P.scal, loc_norm / top);
    FR_LAUNCH(c, "k_piv_adjust", k_piv_adjust, dim3(1), dim3(64), c->vec, B, P.scal, *n_loc_io, exp_loc, unit_t);
    PivScal hs;
    FR_HIP(hipMemcpyAsync(&hs, P.scal, sizeof(hs), hipMemcpyDeviceToHost, st));
    FR_HIP(hipMemcpyAsync(flags_out, B.keep, c->h_vst.curr_size, hipMemcpyDeviceToHost, st));
    FR_HIP(hipStreamSynchronize(st));
    FR_HIP(hipMemsetAsync(B.keep, 0, c->h_vst.curr_size, st));
    if (hs.too_big) { *n_loc_io = hs.n_loc; *new_norm = hs.n_loc * loc_norm / exp_loc; }
    else *new_norm = loc_norm;
}
