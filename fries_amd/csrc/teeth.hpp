// Exact positions of the systematic-sampling comb.
//
// The reference advances its comb by repeated addition, rn_sys += norm / n_samp
// (FRIES/compress_utils.cpp:318, :761, :786), so tooth k is fl(fl(r0 + u) + u ...), not
// r0 + k*u.  At 1e6 samples the two differ by ~1e-13 relative, enough to move a sample
// to a neighbouring element.  Inside one binade [2^e, 2^(e+1)) every partial sum is a
// multiple of ulp = 2^(e-52), so each further addition of the constant u adds the same
// rounded increment; the sequence is therefore piecewise linear with at most a few
// segments per binade, and tooth k can be evaluated in O(log #segments) by any lane.
#pragma once
#include "fries_dev.hpp"
#include <cstddef>

#define FR_MAX_SEG 320

struct TeethSeg { double x0, inc; uint32_t k0, len; };   // teeth k0 .. k0+len-1 are x0 + j*inc
struct Teeth {
    uint32_t nseg, kmax;
    double unit, lbound0;
    TeethSeg seg[FR_MAX_SEG];
};

__device__ inline int fr_exponent(double x) { return (int)((__double_as_longlong(x) >> 52) & 0x7ff) - 1023; }

// One thread.  r0 = first tooth (after seed_sys), u = spacing, kmax = teeth to tabulate.
__device__ inline void fr_build_teeth(Teeth *t, double r0, double u, uint32_t kmax, double lbound0) {
    uint32_t ns = 0, k = 0;
    double x = r0;
    t->unit = u; t->lbound0 = lbound0; t->kmax = kmax;
    auto emit = [&](double x0, double inc, uint32_t k0, uint32_t len) {
        if (ns < FR_MAX_SEG) { t->seg[ns].x0 = x0; t->seg[ns].inc = inc; t->seg[ns].k0 = k0; t->seg[ns].len = len; ns++; }
    };
    if (!(u > 0) || !(r0 == r0) || isinf(r0)) { t->nseg = 0; t->kmax = 0; return; }
    while (k < kmax && ns < FR_MAX_SEG - 3) {
        double x1 = x + u;
        if (x <= 0 || x1 == x) {           // zero start, or stagnation (u below half an ulp)
            if (x1 == x) { emit(x, 0.0, k, kmax - k); k = kmax; break; }
            emit(x, x1 - x, k, 1); k += 1; x = x1; continue;
        }
        int e = fr_exponent(x);
        double top = ldexp(1.0, e + 1), ulp = ldexp(1.0, e - 52);
        if (x1 >= top) { emit(x, x1 - x, k, 1); k += 1; x = x1; continue; }
        double x2 = x1 + u;
        if (x2 >= top) { emit(x, x1 - x, k, 1); emit(x1, x2 - x1, k + 1, 1); k += 2; x = x2; continue; }
        double inc = x2 - x1;              // exact: both multiples of ulp in one binade
        emit(x, x1 - x, k, 1);
        const double iulp = ldexp(1.0, 52 - e);                                    // 1 / ulp: scaling by a power of two is exact
        long long A = (long long)((top - x1) * iulp), B = (long long)(inc * iulp);  // exact integers (< 2^53)
        // largest m with x1 + m*inc < top: (A - 1) / B.  The one thread that builds the comb would spend most of its time in the
        // software 64-bit division; the double quotient of two integers below 2^53 is off by less than one, and the two products
        // that settle it stay below 2^54.
        long long m = (long long)floor((double)(A - 1) / (double)B);
        while (m * B > A - 1) m--;
        while ((m + 1) * B <= A - 1) m++;
        emit(x1, inc, k + 1, (uint32_t)(m + 1));
        x = x1 + (double)m * inc;          // exact
        k = k + 1 + (uint32_t)m;
        // x is now tooth k; the loop top takes the (real) step that leaves the binade
    }
    // segments may run past kmax (lookups clamp); if the table filled up first, only the
    // teeth below k are tabulated
    t->nseg = ns;
    t->kmax = k < kmax ? k : kmax;
}

// Copies the table into LDS (all threads of the workgroup call it; ends with a barrier).  A lookup is a binary search
// over ~60 segments: from global memory that is ~7 dependent L2 round trips per call, three calls per element.
__device__ __forceinline__ void fr_stage_teeth(Teeth *dst, const Teeth *src) {
    const uint32_t nseg = src->nseg < FR_MAX_SEG ? src->nseg : FR_MAX_SEG;
    const uint32_t ndw = (uint32_t)((offsetof(Teeth, seg) + (size_t)nseg * sizeof(TeethSeg)) / 4);
    const uint32_t *s = (const uint32_t *)src;
    uint32_t *d = (uint32_t *)dst;
    for (uint32_t i = threadIdx.x; i < ndw; i += blockDim.x) d[i] = s[i];
    __syncthreads();
}

// position of tooth k (k < kmax)
__device__ inline double fr_tooth(const Teeth *t, uint32_t k) {
    if (k >= t->kmax) return INFINITY;
    int lo = 0, hi = (int)t->nseg - 1;
    while (lo < hi) {                      // last segment with k0 <= k
        int mid = (lo + hi + 1) >> 1;
        if (t->seg[mid].k0 <= k) lo = mid; else hi = mid - 1;
    }
    const TeethSeg &s = t->seg[lo];
    return s.x0 + (double)(k - s.k0) * s.inc;
}

// number of teeth strictly below S == index of the first tooth >= S
__device__ inline uint32_t fr_teeth_below(const Teeth *t, double S) {
    if (t->nseg == 0 || !(t->seg[0].x0 < S)) return 0;
    int lo = 0, hi = (int)t->nseg - 1;
    while (lo < hi) {                      // last segment whose first tooth is < S
        int mid = (lo + hi + 1) >> 1;
        if (t->seg[mid].x0 < S) lo = mid; else hi = mid - 1;
    }
    const TeethSeg &s = t->seg[lo];
    long long j;
    if (s.inc > 0) {
        j = (long long)((S - s.x0) / s.inc);
        if (j < 0) j = 0;
        if (j > (long long)s.len) j = s.len;
        while (j < (long long)s.len && s.x0 + (double)j * s.inc < S) j++;
        while (j > 0 && s.x0 + (double)(j - 1) * s.inc >= S) j--;
    }
    else j = s.len;                        // all equal and < S
    uint32_t k = s.k0 + (uint32_t)j;
    return k < t->kmax ? k : t->kmax;
}

// A lane's position on the comb while it walks forward: the tooth it is at and the segment that tooth lies in, in registers.  The kernels
// that replay sys_sub take teeth one after the other (compress_utils.cpp:766-790); looking each one up by binary search over ~60
// segments in LDS was most of their instruction count.  rn = position of tooth k (infinity beyond the last one), last = position of the
// tooth taken last.  The values are those of fr_tooth: x0 + (double)(k - k0) * inc of the last segment that starts at or before k.
struct ToothCur { uint32_t k, k0, kend, seg; double x0, inc, rn, last; };
// k lies in segment `seg` or a later one
__device__ inline void fr_cur_load(const Teeth *t, ToothCur &c, uint32_t seg, uint32_t k) {
    const uint32_t kmax = t->kmax, nseg = t->nseg;
    c.k = k; c.seg = seg;
    if (nseg == 0 || k >= kmax) { c.k0 = k; c.kend = k; c.x0 = 0; c.inc = 0; c.rn = INFINITY; return; }
    while (seg + 1 < nseg && t->seg[seg + 1].k0 <= k) seg++;
    const TeethSeg &s = t->seg[seg];
    const uint32_t nx = seg + 1 < nseg ? t->seg[seg + 1].k0 : 0xFFFFFFFFu;
    c.seg = seg; c.k0 = s.k0; c.kend = nx < kmax ? nx : kmax; c.x0 = s.x0; c.inc = s.inc;
    c.rn = s.x0 + (double)(k - s.k0) * s.inc;
}
__device__ __forceinline__ void fr_cur_next(const Teeth *t, ToothCur &c) {
    c.last = c.rn;
    c.k++;
    if (c.k < c.kend) c.rn = c.x0 + (double)(c.k - c.k0) * c.inc;
    else fr_cur_load(t, c, c.seg, c.k);
}
__device__ inline void fr_cur_seek_k(const Teeth *t, ToothCur &c, uint32_t k) {
    int lo = 0, hi = (int)t->nseg - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (t->seg[mid].k0 <= k) lo = mid; else hi = mid - 1;
    }
    c.last = -INFINITY;
    fr_cur_load(t, c, (uint32_t)lo, k);
}
// at the first tooth >= S (fr_teeth_below); the quotient is only a first guess that the two loops settle, so a reciprocal does
__device__ inline void fr_cur_seek_below(const Teeth *t, ToothCur &c, double S) {
    c.last = -INFINITY;
    if (t->nseg == 0 || !(t->seg[0].x0 < S)) { fr_cur_load(t, c, 0u, 0u); return; }
    int lo = 0, hi = (int)t->nseg - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (t->seg[mid].x0 < S) lo = mid; else hi = mid - 1;
    }
    const TeethSeg &s = t->seg[lo];
    long long j;
    if (s.inc > 0) {
        const double q = (S - s.x0) * __builtin_amdgcn_rcp(s.inc);
        j = q < 4.0e9 ? (long long)q : (long long)s.len;
        if (j < 0) j = 0;
        if (j > (long long)s.len) j = s.len;
        while (j < (long long)s.len && s.x0 + (double)j * s.inc < S) j++;
        while (j > 0 && s.x0 + (double)(j - 1) * s.inc >= S) j--;
    }
    else j = s.len;
    uint32_t k = s.k0 + (uint32_t)j;
    if (k > t->kmax) k = t->kmax;
    fr_cur_load(t, c, (uint32_t)lo, k);
}

// FRIES/compress_utils.cpp:107-127 (seed_sys) for a shard whose lower bound is the sum of
// the lower-ranked shards' norms.  Returns the first tooth; *unit = spacing.
__device__ inline double fr_seed_sys(double rn, double lbound, double global_norm, uint32_t n_samp, double *unit) {
    double u = global_norm / n_samp;
    rn *= u;
    rn += u * (int)(lbound * n_samp / global_norm);
    if (rn < lbound) rn += u;
    *unit = u;
    return rn;
}
