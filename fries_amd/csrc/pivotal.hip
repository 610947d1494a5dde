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

struct DD_host { double hi, lo; };
static void piv_alloc(FriesCtx *c, PivBuf &P, uint32_t cap) {
    if (P.cap >= cap) return;
    if (P.start) { FR_HIP(hipFree(P.start)); FR_HIP(hipFree(P.carry)); FR_HIP(hipFree(P.U)); FR_HIP(hipFree(P.unit)); FR_HIP(hipFree(P.scal)); FR_HIP(hipFree(P.nz_start)); }
    P.start = fr_alloc<uint32_t>(cap); P.carry = fr_alloc<double>(cap); P.U = fr_alloc<double>(2 * (size_t)cap); P.unit = fr_alloc<PivUnit>(cap);
    P.nz_start = fr_alloc<uint32_t>(cap);
    P.scal = fr_alloc<PivScal>(1);
    if (!P.tile_dd) { P.tile_dd = fr_alloc<DD_host>(FR_MAX_PART); P.tile_nz = fr_alloc<uint32_t>(FR_MAX_PART); }
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

// ------------------------------------------------------------------ the cut points in parallel
// In exact arithmetic element i closes a unit iff floor(P_i / unit) > floor(P_{i-1} / unit), P = inclusive prefix sums of the
// unpreserved magnitudes, and the overshoot carried on is P_i mod unit.  The reference's running sum differs from that ideal
// by its accumulated rounding.  Adding an exact zero does not round.  A non-zero add that stays below unit lands below 2 eu,
// eu = the largest power of two <= unit, and rounds by at most h = eu * 2^-53; an add that crosses a border lands below 4 eu and
// rounds by at most 2 h; the subtraction of unit at a border is exact (Sterbenz).  So the drift at an element is at most
// h * (non-zero unpreserved elements so far + borders crossed so far) -- counted exactly (tile_nz, nz_start) instead of charging
// 2.5e-16 unit to every array position (round 1), which was 3-6 x looser on apply_HBPP_piv's long vectors where most positions hold
// zeros, and sent 1 compression in 8 to the sequential chain.  The prefix sums are formed in double-double (error ~1e-32
// relative, negligible), every element checks that its distance to the nearest unit border exceeds that drift bound, and
// k_piv_decide checks its own comparisons the same way.  If all clear, the decisions -- the only thing the outputs depend
// on -- are provably the reference's; otherwise the caller falls back to the sequential k_piv_chain.
struct DD { double hi, lo; };
__device__ __forceinline__ DD dd_make(double a) { return DD{a, 0.0}; }
__device__ __forceinline__ DD dd_add(DD a, DD b) {
    double s = a.hi + b.hi, bb = s - a.hi;
    double e = (a.hi - (s - bb)) + (b.hi - bb);       // two_sum
    e += a.lo + b.lo;
    double hi = s + e, lo = e - (hi - s);             // quick_two_sum
    return DD{hi, lo};
}
__device__ __forceinline__ DD dd_shfl_up(DD v, int off) { return DD{__shfl_up(v.hi, off), __shfl_up(v.lo, off)}; }
// inclusive scan over the workgroup; sh: 4 DD
__device__ __forceinline__ DD dd_block_scan(DD x, DD *sh, DD *total) {
    const int lane = fr_lane(), w = threadIdx.x >> 6;
    DD v = x;
    for (int off = 1; off < 64; off <<= 1) { DD t = dd_shfl_up(v, off); if (lane >= off) v = dd_add(t, v); }
    __syncthreads();
    if (lane == 63) sh[w] = v;
    __syncthreads();
    DD base = dd_make(0.0), tot = dd_make(0.0);
    for (int k = 0; k < 4; k++) { if (k < w) base = dd_add(base, sh[k]); tot = dd_add(tot, sh[k]); }
    if (w > 0) v = dd_add(base, v);
    *total = tot;
    return v;
}
__device__ __forceinline__ double piv_weight(const VecDev &V, const VcompBuf &B, uint32_t i, uint32_t n) { return (i < n && !B.keep[i]) ? fabs(V.v0[i]) : 0.0; }

__global__ void __launch_bounds__(FR_BLOCK) k_pivdd_tiles(VecDev V, VcompBuf B, DD *tile_sum, uint32_t *tile_nz) {
    __shared__ DD sh[4];
    __shared__ uint32_t shu[4];
    const uint32_t n = V.st->curr_size;
    const uint32_t base = blockIdx.x * FR_TILE + threadIdx.x * FR_ITEMS;
    DD s = dd_make(0.0);
    uint32_t nz = 0;
    for (int it = 0; it < FR_ITEMS; it++) { const double w = piv_weight(V, B, base + it, n); s = dd_add(s, dd_make(w)); nz += w != 0.0; }
    DD tot;
    dd_block_scan(s, sh, &tot);
    const uint32_t tnz = fr_block_sum_u32(nz, shu);
    if (threadIdx.x == 0) { tile_sum[blockIdx.x] = tot; tile_nz[blockIdx.x] = tnz; }
}
// exclusive prefix over the tiles (one workgroup; <= FR_MAX_PART tiles)
__global__ void __launch_bounds__(FR_BLOCK) k_pivdd_scan(DD *tile_sum, uint32_t *tile_nz, uint32_t n_tiles) {
    __shared__ DD sh[4];
    __shared__ uint32_t shu[4];
    const uint32_t per = (n_tiles + FR_BLOCK - 1) / FR_BLOCK;
    const uint32_t t0 = threadIdx.x * per;
    DD s = dd_make(0.0);
    uint32_t c = 0;
    for (uint32_t t = t0; t < t0 + per && t < n_tiles; t++) { s = dd_add(s, tile_sum[t]); c += tile_nz[t]; }
    DD tot;
    DD incl = dd_block_scan(s, sh, &tot);
    DD run = dd_add(incl, DD{-s.hi, -s.lo});
    uint32_t ctot;
    uint32_t crun = fr_block_scan_u32(c, shu, &ctot) - c;
    for (uint32_t t = t0; t < t0 + per && t < n_tiles; t++) { DD x = tile_sum[t]; tile_sum[t] = run; run = dd_add(run, x); const uint32_t y = tile_nz[t]; tile_nz[t] = crun; crun += y; }
}
__global__ void k_pivdd_init(PivBuf P, VecDev V) {
    P.scal->n_units = V.st->curr_size ? 1u : 0u; P.scal->end_pos = V.st->curr_size; P.scal->uncertain = 0;
    P.start[0] = 0; P.carry[0] = 0; P.nz_start[0] = 0;
}
// floor(P / unit) and the remainder, P >= 0
__device__ __forceinline__ void dd_divmod(DD Pv, double unit, double *q_out, double *r_out) {
    double q = floor(Pv.hi / unit);
    for (int guard = 0; guard < 4; guard++) {
        double p = q * unit, e = __fma_rn(q, unit, -p);        // q * unit exactly as p + e
        DD r = dd_add(Pv, DD{-p, -e});
        double rr = r.hi + r.lo;
        if (rr < 0) { q -= 1; continue; }
        if (rr >= unit) { q += 1; continue; }
        *q_out = q; *r_out = rr;
        return;
    }
    *q_out = q; *r_out = -1;      // not settled: the caller treats a negative remainder as uncertain
}
__global__ void __launch_bounds__(FR_BLOCK) k_pivdd_cuts(VecDev V, VcompBuf B, PivBuf P, const DD *tile_off, double unit, uint32_t n_samp, double tol_per_add, uint32_t *err, int dbg) {
    __shared__ DD sh[4];
    const uint32_t n = V.st->curr_size;
    const uint32_t base = blockIdx.x * FR_TILE + threadIdx.x * FR_ITEMS;
    double w[FR_ITEMS];
    DD s = dd_make(0.0);
#pragma unroll
    for (int it = 0; it < FR_ITEMS; it++) { w[it] = piv_weight(V, B, base + it, n); s = dd_add(s, dd_make(w[it])); }
    DD tot;
    DD incl = dd_block_scan(s, sh, &tot);
    DD run = dd_add(dd_add(tile_off[blockIdx.x], incl), DD{-s.hi, -s.lo});      // exclusive prefix at my first element
    uint32_t my_nz = 0;
#pragma unroll
    for (int it = 0; it < FR_ITEMS; it++) my_nz += w[it] != 0.0;
    __shared__ uint32_t shu[4];
    uint32_t nz_tot;
    uint32_t nz_run = P.tile_nz[blockIdx.x] + fr_block_scan_u32(my_nz, shu, &nz_tot) - my_nz;       // non-zero unpreserved elements before my first one
    uint32_t unsure_bits = 0; bool toobig = false;
#pragma unroll
    for (int it = 0; it < FR_ITEMS; it++) {
        const uint32_t i = base + it;
        if (i >= n || w[it] == 0.0) continue;
        if (w[it] >= unit) toobig = true;
        double q0, r0, q1, r1;
        dd_divmod(run, unit, &q0, &r0);
        run = dd_add(run, dd_make(w[it]));
        dd_divmod(run, unit, &q1, &r1);
        nz_run++;
        // rounding drift the reference's running sum can have accumulated by this add: adds that can round + borders crossed (see above)
        const double drift = tol_per_add * ((double)nz_run + q1 + 64.0);
        // whether the very last element closes its unit changes nothing: either way the unit ends with the vector (:425-428)
        const double tol = (i + 1 == n) ? -1.0 : drift;
        if (r0 < 0 || r1 < 0) { unsure_bits |= 2u; continue; }
        if (q1 > q0) {           // closes unit q1 - 1; the running sum lands r1 above the border
            if (q1 != q0 + 1) toobig = true;
            if (r1 <= tol || (unit - r0 <= drift)) { unsure_bits |= 1u; if (dbg) printf("[piv] crossing i=%u of %u q0=%.0f q1=%.0f r0/unit=%.3e r1/unit=%.3e tol/unit=%.3e w/unit=%.3e\n", i, n, q0, q1, r0 / unit, r1 / unit, tol / unit, w[it] / unit); }
            const double m = q1;
            const uint32_t nxt = i + 1;
            if (m < (double)n_samp && nxt < n) { const uint32_t mi = (uint32_t)m; P.start[mi] = nxt; P.carry[mi] = r1; P.nz_start[mi] = nz_run; atomicAdd(&P.scal->n_units, 1u); }
            if (m == (double)n_samp) P.scal->end_pos = nxt < n ? nxt : n;
        }
        else if (unit - r1 <= tol) { unsure_bits |= 1u; if (dbg) printf("[piv] inside i=%u of %u q=%.0f (unit-r1)/unit=%.3e tol/unit=%.3e\n", i, n, q1, (unit - r1) / unit, tol / unit); }
    }
    if (toobig) atomicOr(err, FR_ERR_PIV);
    if (unsure_bits) atomicOr(&P.scal->uncertain, unsure_bits);
}

__device__ __forceinline__ double fr_sgn_unit(double unit, double v) { return unit * ((v > 0) - (v < 0)); }

// One sampling unit, decision only (compress_utils.cpp:409-457 without side effects): which candidate H is drawn and
// whether the border element is sampled.  carry is exact after k_piv_chain; after the parallel cut-point search
// (k_pivdd_*) it is the ideal value, off from the reference's running sum by at most `tol` (the worst-case rounding
// drift of that sum), and then every comparison made here must clear that margin or the unit is reported uncertified.
__global__ void __launch_bounds__(FR_BLOCK) k_piv_decide(VecDev V, VcompBuf B, PivBuf P, double unit, double tol_per_add, uint32_t *err) {
    const uint32_t n = V.st->curr_size;
    const uint32_t n_units = P.scal->n_units;
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_units) return;
    const bool certify = tol_per_add > 0;
    const uint32_t pos = P.start[k];
    const double carry = P.carry[k];
    uint32_t unsure_bits = 0;
    // the unit's elements: cum in the reference's order
    double cum = carry, last = 0;
    uint32_t used = 0, n_wt = 1;
    while (cum < unit && pos + used < n) {
        if (!B.keep[pos + used]) { last = fabs(V.v0[pos + used]); cum += last; n_wt++; }
        used++;
    }
    const bool at_end = pos + used == n;
    if (used == 0) { atomicOr(err, FR_ERR_PIV); return; }
    const double tol = certify ? tol_per_add * ((double)P.nz_start[k] + (double)n_wt + (double)k + 64.0) * 2.0 : -1.0;      // carried drift + this unit's own adds; < 0: nothing to certify
    uint32_t n_inner = used - 1;
    if (at_end) n_inner++;
    if (certify) {      // the walk must end where the cut-point search put the border
        const uint32_t next_start = (k + 1 < n_units) ? P.start[k + 1] : (at_end ? n + 1 : P.scal->end_pos);
        if (!at_end && pos + used != next_start) unsure_bits |= 4u;
        if (at_end && k + 1 < n_units) unsure_bits |= 4u;
    }
    const double over = cum - unit;
    if (!at_end) { n_wt--; cum -= last; }
    const double under = unit - cum;
    // candidate among the residual piece and the inner elements (:437-446)
    const double r = P.U[2 * (size_t)k] * cum;
    double run = 0;
    uint32_t H = 0, h_idx = FR_NOPOS, e = pos;
    if (r != 0 && fabs(r) <= tol) unsure_bits |= 8u;
    if (run < r && H < n_wt) { run += carry; H++; if (fabs(run - r) <= tol) unsure_bits |= 8u; }
    while (run < r && H < n_wt) {
        while (B.keep[e]) e++;
        run += fabs(V.v0[e]); h_idx = e; e++; H++;
        if (fabs(run - r) <= tol) unsure_bits |= 8u;
    }
    if (r > 0) H--;
    if (H == 0) h_idx = FR_NOPOS;
    double p_pass = under / (unit - over);
    if (at_end) p_pass = 0;
    const double u2 = P.U[2 * (size_t)k + 1];
    const bool pass = u2 < p_pass;
    if (certify && !at_end) {
        const double den = unit - over;
        if (!(den > 4 * tol) || fabs(u2 - p_pass) <= 2 * tol * (1 + fabs(p_pass)) / (den - 2 * tol) + 1e-15) unsure_bits |= 16u;
    }
    if (certify && unsure_bits) atomicOr(&P.scal->uncertain, unsure_bits);
    PivUnit u;
    u.H = H; u.pass = pass ? 1 : 0; u.pad[0] = u.pad[1] = u.pad[2] = 0;
    u.new_resid = pass ? h_idx : pos + n_inner;      // pass with H == 0 hands on the residual it received (FR_NOPOS here)
    u.n_inner = n_inner;
    P.unit[k] = u;
}

// ... and what it does to its own elements (:458-501)
__global__ void __launch_bounds__(FR_BLOCK) k_piv_apply(VecDev V, VcompBuf B, PivBuf P, double unit) {
    const uint32_t n_units = P.scal->n_units;
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_units) return;
    const uint32_t pos = P.start[k];
    const PivUnit u = P.unit[k];
    const bool pass = u.pass != 0;
    // the first unit's residual is element 0 itself, and the reference touches it before the unit's own elements (:478-480)
    if (k == 0 && !pass && u.H == 0) V.v0[0] = fr_sgn_unit(unit, V.v0[0]);
    uint32_t cnt = 1;
    for (uint32_t o = 0; o < u.n_inner; o++) {
        const uint32_t i = pos + o;
        if (!B.keep[i]) {
            if (cnt == u.H) { if (!pass) V.v0[i] = fr_sgn_unit(unit, V.v0[i]); }
            else { V.v0[i] = 0; B.del[i] = 1; }
            cnt++;
        }
        else B.keep[i] = 0;
    }
    if (pass) V.v0[pos + u.n_inner] = fr_sgn_unit(unit, V.v0[pos + u.n_inner]);
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

// piv_samp_serial (compress_utils.cpp:389-518) on the host, for the handful of fractional rank weights piv_budget samples
// from (one per rank): the same sequence of operations on plain arrays.
static void piv_samp_host(std::vector<double> &vals, double seg_norm, uint32_t n_samp, std::vector<uint8_t> &flag, std::mt19937 &mt) {
    const size_t len = vals.size();
    if (n_samp == 0) {
        for (size_t i = 0; i < len; i++) { if (flag[i]) flag[i] = 0; else vals[i] = 0; if (vals[i] == 0) flag[i] = 1; }
        return;
    }
    const double unit = seg_norm / n_samp;
    std::vector<double> wt(len + 2);
    wt[0] = 0;
    size_t pos = 0, resid = 0;
    uint32_t n_done = 0;
    auto sgn_unit = [unit](double v) { return unit * ((v > 0) - (v < 0)); };
    while (pos < len && n_done < n_samp) {
        size_t n_wt = 1, used = 0;
        double cum = wt[0];
        for (; cum < unit && pos + used < len; used++)
            if (!flag[pos + used]) { wt[n_wt] = fabs(vals[pos + used]); cum += wt[n_wt]; n_wt++; }
        const bool at_end = pos + used == len;
        size_t n_inner = used > 0 ? used - 1 : 0;
        if (at_end) n_inner++;
        const double over = cum - unit;
        if (!at_end) { n_wt--; cum -= wt[n_wt]; }
        const double under = unit - cum;
        double r = mt() / (1. + UINT32_MAX) * cum;
        double run = 0;
        size_t H = 0;
        while (run < r && H < n_wt) { run += wt[H]; H++; }
        if (r > 0) H--;
        if (H != 0 && pos != 0) { vals[resid] = 0; flag[resid] = 1; }
        double p_pass = under / (unit - over);
        if (at_end) p_pass = 0;
        r = mt() / (1. + UINT32_MAX);
        size_t k = 1;
        if (r < p_pass) {
            for (size_t o = 0; o < n_inner; o++) {
                if (!flag[pos + o]) { if (k == H) resid = pos + o; else { vals[pos + o] = 0; flag[pos + o] = 1; } k++; }
                else flag[pos + o] = 0;
            }
            vals[pos + n_inner] = sgn_unit(vals[pos + n_inner]);
        }
        else {
            if (H == 0) vals[resid] = sgn_unit(vals[resid]);
            for (size_t o = 0; o < n_inner; o++) {
                if (!flag[pos + o]) { if (k != H) { vals[pos + o] = 0; flag[pos + o] = 1; } else vals[pos + o] = sgn_unit(vals[pos + o]); k++; }
                else flag[pos + o] = 0;
            }
            resid = pos + n_inner;
        }
        pos += n_inner + 1;
        wt[0] = over;
        n_done++;
    }
    for (; pos < len; pos++) { if (!flag[pos]) { vals[pos] = 0; flag[pos] = 1; } else flag[pos] = 0; }
    if (resid < len) { vals[resid] = 0; flag[resid] = 1; }
}

// piv_budget (compress_utils.cpp:552-604): rank 0 apportions the n_samp samples among the ranks from their remaining norms --
// the integer parts of the shares, then one more for the ranks a pivotal draw over the fractional parts selects (its generator
// alone advances) -- and scatters them; here the scatter is an all-gather of rank 0's table.
static uint32_t piv_budget(FriesCtx *c, const std::vector<double> &norms, uint32_t n_samp) {
    const int P = c->n_ranks;
    std::vector<uint32_t> budgets(P, 0);
    if (c->rank == 0) {
        double glob = 0;
        for (int p = 0; p < P; p++) glob += norms[p];
        uint32_t tot = 0, n_frac = 0;
        std::vector<double> frac(P);
        for (int p = 0; p < P; p++) {
            budgets[p] = norms[p] / glob * n_samp;
            tot += budgets[p];
            frac[p] = norms[p] - budgets[p] * glob / n_samp;
            if (frac[p] < 1e-12) frac[p] = 0;
            if (frac[p] > 0) n_frac++;
        }
        if (n_frac == n_samp - tot) { for (int p = 0; p < P; p++) if (frac[p] > 0) budgets[p]++; tot = n_samp; }
        if (tot < n_samp) {
            std::vector<uint8_t> none(P, 0);
            piv_samp_host(frac, glob * (n_samp - tot) / n_samp, n_samp - tot, none, c->mt);
            for (int p = 0; p < P; p++) if (frac[p] > 0) budgets[p]++;
        }
    }
    if (!c->use_comm) return budgets[0];
    FR_HIP(hipMemcpyAsync(c->comm.small_send, budgets.data(), 4 * (size_t)P, hipMemcpyHostToDevice, c->stream));
    const void *all = fr_allgather(c, 4 * (size_t)P);
    std::vector<uint32_t> got(P);
    FR_HIP(hipMemcpyAsync(got.data(), all, 4 * (size_t)P, hipMemcpyDeviceToHost, c->stream));      // rank 0's block
    FR_HIP(hipStreamSynchronize(c->stream));
    return got[c->rank];
}

// flat: c->vec / c->vc / c->piv stand for a plain array (fr_piv_comp_flat) -- no second column, no deletes afterwards
static void piv_comp_core(FriesCtx *c, uint32_t compress_size, uint32_t *n_kept, double *glob_norm_out, bool flat) {
    VcompBuf &B = c->vc;
    hipStream_t st = c->stream;
    fr_vec_sync_state(c, &c->vec, &c->h_vst);
    const uint32_t bound = c->h_vst.curr_size ? c->h_vst.curr_size : 1;
    const uint32_t save = c->vec_nonz;
    c->vec_nonz = compress_size;
    if (flat) fr_abs_sums(c);
    else fr_death_clone(c, 0);               // publishes the |v| block sums; column 1 is zero so values are unchanged
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
    // every rank's remaining norm, then their sum in rank order (compress_utils.cpp:366-371)
    std::vector<double> norms(c->n_ranks, mine);
    if (c->use_comm) {
        FR_HIP(hipMemcpyAsync(c->comm.small_send, &mine, 8, hipMemcpyHostToDevice, st));
        const void *all = fr_allgather(c, 8);
        FR_HIP(hipMemcpyAsync(norms.data(), all, 8 * (size_t)c->n_ranks, hipMemcpyDeviceToHost, st));
        FR_HIP(hipStreamSynchronize(st));
    }
    double glob = 0;
    for (int p = 0; p < c->n_ranks; p++) glob += norms[p];
    uint32_t loc_samp = 0;
    double new_norm = 0;
    if (n_samp != 0) {
        loc_samp = piv_budget(c, norms, n_samp);
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
        const double tol_per_add = ldexp(1.0, ilogb(unit) - 53) * 1.0001;     // h: what one non-crossing add of the reference's running sum can round by (see k_pivdd_*)
        bool chain = getenv("FRIES_PIV_CHAIN") != nullptr;  // force the sequential search (tests)
        const std::mt19937 mt_saved = c->mt;
        const unsigned n_tiles = fr_blocks(bound, FR_TILE);
        PivScal hs{};
        auto read_scal = [&]() { FR_HIP(hipMemcpyAsync(&hs, P.scal, sizeof(hs), hipMemcpyDeviceToHost, st)); FR_HIP(hipStreamSynchronize(st)); };
        auto draw = [&]() {
            if (hs.n_units > P.cap) throw FriesError("pivotal compression: more sampling units than the work arrays hold");
            std::vector<double> u(2 * (size_t)hs.n_units);
            for (size_t q = 0; q < 2 * (size_t)hs.n_units; q++) u[q] = c->mt() / (1. + UINT32_MAX);       // two per unit, in unit order (:437, :457)
            FR_HIP(hipMemcpyAsync(P.U, u.data(), 16 * (size_t)hs.n_units, hipMemcpyHostToDevice, st));
            FR_HIP(hipStreamSynchronize(st));
        };
        if (!chain) {
            DD *td = (DD *)P.tile_dd;
            FR_LAUNCH(c, "k_pivdd_init", k_pivdd_init, dim3(1), dim3(1), P, c->vec);
            FR_LAUNCH(c, "k_pivdd_tiles", k_pivdd_tiles, dim3(n_tiles), dim3(FR_BLOCK), c->vec, B, td, P.tile_nz);
            FR_LAUNCH(c, "k_pivdd_scan", k_pivdd_scan, dim3(1), dim3(FR_BLOCK), td, P.tile_nz, n_tiles);
            FR_LAUNCH(c, "k_pivdd_cuts", k_pivdd_cuts, dim3(n_tiles), dim3(FR_BLOCK), c->vec, B, P, td, unit, loc_samp, tol_per_add, c->d_err, c->dbg == 7 ? 1 : 0);
            read_scal();
            if (hs.uncertain) { chain = true; P.last_reason = hs.uncertain; }
            else if (hs.n_units) {
                draw();
                FR_LAUNCH(c, "k_piv_decide", k_piv_decide, dim3(fr_blocks(hs.n_units, FR_BLOCK)), dim3(FR_BLOCK), c->vec, B, P, unit, tol_per_add, c->d_err);
                read_scal();
                if (hs.uncertain) { chain = true; c->mt = mt_saved; P.last_reason = hs.uncertain; }
            }
            if (!chain) P.n_certified++;
        }
        if (chain) {            // the reference's own order of operations, one wave
            P.n_fallback++;
            FR_LAUNCH(c, "k_piv_chain", k_piv_chain, dim3(1), dim3(64), c->vec, B, P, unit, loc_samp, c->d_err);
            read_scal();
            if (hs.n_units) {
                draw();
                FR_LAUNCH(c, "k_piv_decide", k_piv_decide, dim3(fr_blocks(hs.n_units, FR_BLOCK)), dim3(FR_BLOCK), c->vec, B, P, unit, 0.0, c->d_err);
            }
        }
        if (hs.n_units) {
            FR_LAUNCH(c, "k_piv_apply", k_piv_apply, dim3(fr_blocks(hs.n_units, FR_BLOCK)), dim3(FR_BLOCK), c->vec, B, P, unit);
            FR_LAUNCH(c, "k_piv_resid", k_piv_resid, dim3(fr_blocks((size_t)hs.n_units + 1, FR_BLOCK)), dim3(FR_BLOCK), c->vec, B, P, unit);
        }
        if (hs.end_pos < bound) FR_LAUNCH(c, "k_piv_tail", k_piv_tail, dim3(fr_blocks(bound - hs.end_pos, FR_BLOCK)), dim3(FR_BLOCK), c->vec, B, P);
    }
    if (flat) return;
    fr_vec_delete_flagged(c, &c->vec, B.del, bound);         // compress_vecs: vec_utils.cpp:25-30
    fr_vec_sync_state(c, &c->vec, &c->h_vst);
}

void fr_piv_comp(FriesCtx *c, uint32_t compress_size, uint32_t *n_kept, double *glob_norm_out) { piv_comp_core(c, compress_size, n_kept, glob_norm_out, false); }

// piv_comp_parallel (compress_utils.cpp:354-386) of a plain device array of n values (apply_HBPP_piv's long_vec): afterwards
// F.vals holds the compressed values and F.vc.del[i] == 1 marks the elements that ended up zero.  The array rides through the same
// kernels as the solution vector: for the duration of the call it IS c->vec (a VecDev whose only live members are v0 and st).
void fr_piv_flat_reserve(FriesCtx *c, uint32_t cap) {
    FlatPiv &F = c->flat;
    if (F.cap >= cap) return;
    if (fr_blocks(cap, FR_TILE) > FR_MAX_PART) throw FriesError("pivotal matrix compression: a factor expands to more elements than the work arrays index (134e6)");
    if (F.cap) {
        FR_HIP(hipStreamSynchronize(c->stream));
        FR_HIP(hipFree(F.vals)); FR_HIP(hipFree(F.parent));
        VcompBuf &B = F.vc;
        FR_HIP(hipFree(B.keep)); FR_HIP(hipFree(B.del)); FR_HIP(hipFree(B.S));
        for (int h = 0; h < 2; h++) { FR_HIP(hipFree(B.psum[h])); FR_HIP(hipFree(B.pcnt[h])); }
        FR_HIP(hipFree(B.state)); FR_HIP(hipFree(B.teeth)); FR_HIP(hipFree(B.dots)); FR_HIP(hipFree(B.fix_list));
        FR_HIP(hipFree(B.seq.tiles)); FR_HIP(hipFree(B.seq.subs)); FR_HIP(hipFree(B.seq.total)); FR_HIP(hipFree(B.gnorm));
    }
    F.vals = fr_alloc<double>(cap); F.parent = fr_alloc<uint32_t>(cap);
    if (!F.st) { F.st = fr_alloc<VecState>(1); F.total = fr_alloc<uint32_t>(2); }
    std::swap(c->vc, F.vc);
    c->vc = VcompBuf{};
    fr_vcomp_alloc(c, cap);
    std::swap(c->vc, F.vc);
    F.cap = cap;
}
void fr_piv_flat_free(FriesCtx *c) {
    FlatPiv &F = c->flat;
    if (!F.cap) return;
    hipFree(F.vals); hipFree(F.parent); hipFree(F.st); hipFree(F.total);
    VcompBuf &B = F.vc;
    hipFree(B.keep); hipFree(B.del); hipFree(B.S);
    for (int h = 0; h < 2; h++) { hipFree(B.psum[h]); hipFree(B.pcnt[h]); }
    hipFree(B.state); hipFree(B.teeth); hipFree(B.dots); hipFree(B.fix_list);
    hipFree(B.seq.tiles); hipFree(B.seq.subs); hipFree(B.seq.total); hipFree(B.gnorm);
    PivBuf &P = F.piv;
    if (P.start) { hipFree(P.start); hipFree(P.carry); hipFree(P.U); hipFree(P.unit); hipFree(P.scal); }
    if (P.tile_dd) hipFree(P.tile_dd);
    F = FlatPiv{};
}
void fr_piv_comp_flat(FriesCtx *c, uint32_t n, uint32_t compress_size) {
    FlatPiv &F = c->flat;
    if (n > F.cap) throw FriesError("fr_piv_comp_flat: array longer than reserved");
    VecState hs{};
    hs.curr_size = n;
    FR_HIP(hipMemcpyAsync(F.st, &hs, sizeof(hs), hipMemcpyHostToDevice, c->stream));
    FR_HIP(hipStreamSynchronize(c->stream));        // hs is a host temporary
    VecDev fv{};
    fv.cap = F.cap; fv.v0 = F.vals; fv.st = F.st;
    struct Swap {       // the solution vector comes back whatever happens
        FriesCtx *c; FlatPiv &F; VecDev fv; VecState saved;
        Swap(FriesCtx *c_, FlatPiv &F_, VecDev v) : c(c_), F(F_), fv(v), saved(c_->h_vst) { std::swap(c->vec, fv); std::swap(c->vc, F.vc); std::swap(c->piv, F.piv); }
        ~Swap() {
            std::swap(c->vec, fv); std::swap(c->vc, F.vc); std::swap(c->piv, F.piv); c->h_vst = saved;
            // the statistics of fries_piv_stats count these calls, too
            c->piv.n_certified += F.piv.n_certified; c->piv.n_fallback += F.piv.n_fallback;
            if (F.piv.n_fallback) c->piv.last_reason = F.piv.last_reason;
            F.piv.n_certified = F.piv.n_fallback = 0;
        }
    } guard(c, F, fv);
    piv_comp_core(c, compress_size, nullptr, nullptr, true);
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
