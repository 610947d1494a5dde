// Shared device-side definitions for the MI355X FRI engine (gfx950 only).
// Arithmetic contract: plain IEEE-754 double operations in the reference's literal
// order; everything here is compiled with -ffp-contract=off so hipcc never fuses a*b+c.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstddef>

#define FR_WAVE 64
#define FR_BLOCK 256            // 4 waves, one per SIMD
#define FR_MAX_ORB 32           // 2*n_orb <= 64 bits per determinant
#define FR_MAX_PART 131072      // max blocks whose partials one consumer block re-reduces (capacity: 134e6 elements per array; the re-reduction reads the LIVE tiles before its own, not the capacity)
#define FR_EMPTY_KEY 0ull
#define FR_TOMB_KEY (~0ull)
#define FR_NOPOS 0xFFFFFFFFu
#define FR_NEWBIT 0x80000000u
#define FR_POSBIT 0x80000000u      // SpawnBuf::slot: the rest is a vector position, not a hash slot

typedef uint64_t det_t;

// ------------------------------------------------------------------ system tables
// Everything the HB-PP row generators and weights need (reference hb_info +
// SymmInfo, FRIES/Hamiltonians/heat_bathPP.hpp:25-34, molecule.hpp:265-280), sized for
// n_orb <= 32 so one copy (~17.5 KB) is staged into LDS per workgroup.
struct HbTables {
    double d_diff[FR_MAX_ORB * FR_MAX_ORB];
    double d_same[FR_MAX_ORB * (FR_MAX_ORB - 1) / 2];
    double exch_sqrt[FR_MAX_ORB * (FR_MAX_ORB - 1) / 2];
    double s_tens[FR_MAX_ORB];
    double diag_sqrt[FR_MAX_ORB];
    double exch_norms[FR_MAX_ORB];
    double s_norm;
    uint8_t irrep[FR_MAX_ORB];
    uint8_t lookup[8][FR_MAX_ORB + 1];   // [irrep][0] = count, then orbitals ascending
    uint32_t n_orb, n_elec, max_n_symm, pad;
};

struct SysDev {
    uint32_t n_orb, n_elec;
    const double *h_core;   // n_orb x n_orb
    const double *eris;     // 8-fold packed (FRIES/ndarr.hpp:206-244)
    const HbTables *hb;     // device copy
    double hf_en;
    int spin_parity = 0;    // +-1: time-reversal symmetrised vectors (fr_adjust_tr, hbpp_rows.hpp)
};

// Device-resident sparse vector (reference DistVec<double>, FRIES/vec_utils.hpp:121-141).
struct VecState {           // lives in device memory, mirrored to the host on demand
    uint32_t curr_size;     // positions in use incl. holes
    uint32_t n_free;        // entries on the free stack
    int32_t n_nonz;
    uint32_t n_tomb;        // tombstones in the hash table
    uint32_t n_used;        // occupied hash slots incl. tombstones
    uint32_t err;           // sticky error bits
    unsigned long long nonini_occ_add;
};
// One slot of the vector's hash table: key and position side by side, 16 bytes, so that a probe that finds its key has the position in
// the same load (as two arrays every look-up was two random accesses into 32 + 16 MB at m = 1e6, and the merge made three of them).
struct alignas(16) HSlot {
    det_t key;              // FR_EMPTY_KEY / FR_TOMB_KEY / fr_vec_key of the stored index
    uint32_t val;           // position; FR_NOPOS while empty; FR_NEWBIT|j while being created
    uint32_t pad;
};
struct VecDev {
    uint32_t cap;           // max positions
    uint32_t hcap;          // hash slots in use, a power of two: sized to the live set (grown / shrunk at rebuilds), not to max_dets --
                            // random probes into a table 16 x the live set miss every cache (223 MB of traffic per lookup pass at m = 1e6)
    uint32_t hcap_max;      // slots allocated (host bookkeeping)
    uint32_t used_ub;       // host-side upper bound of the occupied slots (exact after fr_vec_sync_state, + spawns after every merge)
    det_t *dets;
    double *v0, *v1;        // the two value columns (n_vecs == 2)
    double *diag;           // cached diagonal element - hf_en; NaN = not yet computed
    uint8_t *active;
    uint32_t *free_stack;   // [n_free-1] is the top (LIFO like std::stack)
    HSlot *hs;              // open addressing, linear probing
    // Counters the merge kernels bump from every workgroup, striped over FR_STAT_STRIPES cache lines ([stripe][FR_STAT_STRIDE]: 0 = non-initiator
    // additions to occupied determinants, 1 = hash slots taken, 2 = tombstones re-used) and folded into *st when the host asks for the state
    // (fr_vec_sync_state): ~16 000 atomics on ONE address cost 70 us per counter and merge.
    unsigned long long *stat_part;
    VecState *st;
    // Hubbard-Holstein indices (hh_vec.hpp): electrons in the low 2 * hh_sites bits, 3 bits per phonon above them
    uint32_t hh_sites, hh_nelec, hh_buckets;
    const uint32_t *hh_scr;     // the reference's vec_hash_ scrambler (device)
    uint32_t n_dense;           // positions [0, n_dense) hold the semi-stochastic dense space: never compressed, never deleted (DistVec::init_dense)
};
#define FR_HH_PH_BITS 3
#define FR_STAT_STRIPES 64
#define FR_STAT_STRIDE 16

enum { FR_ERR_CAP = 1, FR_ERR_SPAWN_CAP = 2, FR_ERR_NELEC = 4, FR_ERR_HASH_FULL = 8, FR_ERR_ROUNDS = 16, FR_ERR_BACKLOG = 32, FR_ERR_PIV = 64 };

// ------------------------------------------------------------------ small helpers
__device__ __forceinline__ int fr_lane() { return threadIdx.x & 63; }
__device__ __forceinline__ bool fr_bit(det_t d, unsigned i) { return (d >> i) & 1ull; }
__device__ __forceinline__ unsigned fr_tri_nodiag(unsigned i, unsigned j) { return j * (j - 1) / 2 + i; }   // i < j
__device__ __forceinline__ size_t fr_tri_wdiag(size_t i, size_t j) { return j * (j + 1) / 2 + i; }         // i <= j

// k-th (0-based) set bit of d; caller guarantees it exists
__device__ __forceinline__ unsigned fr_nth_bit(det_t d, unsigned k) {
    for (unsigned i = 0; i < k; i++) d &= d - 1;
    return __ffsll((long long)d) - 1;
}

// FRIES/math_utils.c:9-58
__device__ __forceinline__ unsigned fr_bits_between(det_t det, unsigned a, unsigned b) {
    unsigned lo = a < b ? a : b, hi = a < b ? b : a;
    if (hi - lo < 2) return 0;
    det_t mask = ((1ull << hi) - 1ull) & ~((1ull << (lo + 1)) - 1ull);
    return __popcll(det & mask);
}
// FRIES/fci_utils.c:85-96 (doub_parity): sign evaluated on the doubly-annihilated string
__device__ __forceinline__ int fr_doub_parity(det_t det, unsigned o1, unsigned o2, unsigned u1, unsigned u2) {
    det &= ~(1ull << o1);
    det &= ~(1ull << o2);
    unsigned n = fr_bits_between(det, u1, o1) + fr_bits_between(det, u2, o2);
    return (n & 1) ? -1 : 1;
}
// FRIES/fci_utils.c:54-58 (sing_parity)
__device__ __forceinline__ int fr_sing_parity(det_t det, unsigned o, unsigned u) {
    return (fr_bits_between(det, o, u) & 1) ? -1 : 1;
}

__device__ __forceinline__ uint64_t fr_mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

// ------------------------------------------------------------------ integrals
__device__ __forceinline__ double fr_chem(const double *eris, unsigned i1, unsigned i2, unsigned i3, unsigned i4) {
    size_t mn1 = i1 < i2 ? i1 : i2, mx1 = i1 < i2 ? i2 : i1;
    size_t p1 = fr_tri_wdiag(mn1, mx1);
    size_t mn2 = i3 < i4 ? i3 : i4, mx2 = i3 < i4 ? i4 : i3;
    size_t p2 = fr_tri_wdiag(mn2, mx2);
    size_t mnp = p1 < p2 ? p1 : p2, mxp = p1 < p2 ? p2 : p1;
    return eris[fr_tri_wdiag(mnp, mxp)];
}
__device__ __forceinline__ double fr_phys(const double *eris, unsigned i, unsigned j, unsigned k, unsigned l) { return fr_chem(eris, i, k, j, l); }

// FRIES/Hamiltonians/molecule.cpp:983-1029 with n_frozen = 0: literal loop order.
__device__ inline double fr_diag_matrel(det_t det, const double *h, const double *eris, unsigned n_orb) {
    det_t lowmask = (n_orb >= 64) ? ~0ull : ((1ull << n_orb) - 1ull);
    det_t alpha = det & lowmask, beta = det >> n_orb;
    double sum = 0;
    for (det_t a = alpha; a; a &= a - 1) {
        unsigned e1 = __ffsll((long long)a) - 1;
        sum += h[e1 * n_orb + e1];
        for (det_t b = a & (a - 1); b; b &= b - 1) {
            unsigned e2 = __ffsll((long long)b) - 1;
            sum += fr_phys(eris, e1, e2, e1, e2);
            sum -= fr_phys(eris, e1, e2, e2, e1);
        }
        for (det_t b = beta; b; b &= b - 1) {
            unsigned e2 = __ffsll((long long)b) - 1;
            sum += fr_phys(eris, e1, e2, e1, e2);
        }
    }
    for (det_t a = beta; a; a &= a - 1) {
        unsigned e1 = __ffsll((long long)a) - 1;
        sum += h[e1 * n_orb + e1];
        for (det_t b = a & (a - 1); b; b &= b - 1) {
            unsigned e2 = __ffsll((long long)b) - 1;
            sum += fr_phys(eris, e1, e2, e1, e2);
            sum -= fr_phys(eris, e1, e2, e2, e1);
        }
    }
    return sum;
}

// FRIES/Hamiltonians/molecule.cpp:76-105
__device__ inline double fr_sing_matrel(det_t det, unsigned o_orb, unsigned u_orb, const double *h, const double *eris, unsigned n_orb) {
    unsigned o = o_orb % n_orb, u = u_orb % n_orb, spin = o_orb / n_orb;
    det_t lowmask = (1ull << n_orb) - 1ull;
    double el = h[o * n_orb + u];
    for (det_t a = det & lowmask; a; a &= a - 1) {
        unsigned j = __ffsll((long long)a) - 1;
        el += fr_phys(eris, o, j, u, j);
        if (spin == 0) el -= fr_phys(eris, o, j, j, u);
    }
    for (det_t b = det >> n_orb; b; b &= b - 1) {
        unsigned j = __ffsll((long long)b) - 1;
        el += fr_phys(eris, o, j, u, j);
        if (spin == 1) el -= fr_phys(eris, o, j, j, u);
    }
    return el;
}

// FRIES/Hamiltonians/molecule.cpp:26-42
__device__ __forceinline__ double fr_doub_matrel(unsigned o1, unsigned o2, unsigned u1, unsigned u2, const double *eris, unsigned n_orb) {
    int same = (o1 / n_orb) == (o2 / n_orb);
    unsigned s0 = o1 % n_orb, s1 = o2 % n_orb, s2 = u1 % n_orb, s3 = u2 % n_orb;
    double el = fr_phys(eris, s0, s1, s2, s3);
    if (same) el -= fr_phys(eris, s0, s1, s3, s2);
    return el;
}

// ------------------------------------------------------------------ LDS staging
__device__ __forceinline__ void fr_stage_tables(HbTables *dst, const HbTables *src) {
    const uint32_t *s = (const uint32_t *)src;
    uint32_t *d = (uint32_t *)dst;
    for (unsigned i = threadIdx.x; i < sizeof(HbTables) / 4; i += blockDim.x) d[i] = s[i];
    __syncthreads();
}

// ------------------------------------------------------------------ block primitives (256 threads)
// Fixed-shape tree: lanes by xor-butterfly inside a wave, then waves 0..3 in order.
// The association order depends only on the launch geometry, never on timing.
__device__ __forceinline__ double fr_wave_sum(double x) {
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
    return x;
}
__device__ __forceinline__ uint32_t fr_wave_sum_u32(uint32_t x) {
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
    return x;
}
__device__ __forceinline__ double fr_block_sum(double x, double *sh /* >= 4 */) {
    x = fr_wave_sum(x);
    if (fr_lane() == 0) sh[threadIdx.x >> 6] = x;
    __syncthreads();
    double r = ((sh[0] + sh[1]) + sh[2]) + sh[3];
    __syncthreads();
    return r;
}
__device__ __forceinline__ uint32_t fr_block_sum_u32(uint32_t x, uint32_t *sh /* >= 4 */) {
    x = fr_wave_sum_u32(x);
    if (fr_lane() == 0) sh[threadIdx.x >> 6] = x;
    __syncthreads();
    uint32_t r = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return r;
}
// inclusive scan over the 256 threads of a block; returns inclusive value, *total = block total
__device__ __forceinline__ uint32_t fr_block_scan_u32(uint32_t x, uint32_t *sh /* >= 4 */, uint32_t *total) {
    int lane = fr_lane(), w = threadIdx.x >> 6;
    uint32_t v = x;
    for (int off = 1; off < 64; off <<= 1) { uint32_t t = __shfl_up(v, off); if (lane >= off) v += t; }
    if (lane == 63) sh[w] = v;
    __syncthreads();
    uint32_t base = 0;
    for (int k = 0; k < w; k++) base += sh[k];
    *total = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return v + base;
}
// inclusive fp64 scan in storage order: Hillis-Steele inside the wave, wave carries in order.
__device__ __forceinline__ double fr_block_scan_f64(double x, double *sh /* >= 4 */, double *total) {
    int lane = fr_lane(), w = threadIdx.x >> 6;
    double v = x;
    for (int off = 1; off < 64; off <<= 1) { double t = __shfl_up(v, off); if (lane >= off) v += t; }
    if (lane == 63) sh[w] = v;
    __syncthreads();
    double base = 0;
    for (int k = 0; k < w; k++) base += sh[k];
    *total = ((sh[0] + sh[1]) + sh[2]) + sh[3];
    __syncthreads();
    return w ? base + v : v;
}

// Sum of partial[0..n) in index order, every thread of the block gets the result.
// (n <= FR_MAX_PART; each lane sums a strided slice, slices combined by the fixed tree.)
__device__ __forceinline__ double fr_sum_partials(const double *p, unsigned n, double *sh) {
    double x = 0;
    for (unsigned i = threadIdx.x; i < n; i += blockDim.x) x += p[i];
    return fr_block_sum(x, sh);
}
__device__ __forceinline__ uint32_t fr_sum_partials_u32(const uint32_t *p, unsigned n, uint32_t *sh) {
    uint32_t x = 0;
    for (unsigned i = threadIdx.x; i < n; i += blockDim.x) x += p[i];
    return fr_block_sum_u32(x, sh);
}

// ------------------------------------------------------------------ hash table
// hash_fxn with phonon numbers (det_hash.hpp:160-170): occupied orbitals, then the phonon number of every site
__device__ __forceinline__ unsigned long long fr_hh_hash(det_t d, const uint32_t *scr, uint32_t n_sites) {
    unsigned long long hash = 0;
    uint32_t i = 0;
    for (det_t a = d & ((1ull << (2 * n_sites)) - 1ull); a; a &= a - 1, i++) {
        unsigned orb = __ffsll((long long)a) - 1;
        hash = 1099511628211ULL * hash + (uint32_t)((i + 1u) * scr[orb]);
    }
    for (uint32_t s = 0; s < n_sites; s++) {
        uint32_t ph = (uint32_t)(d >> (2 * n_sites + FR_HH_PH_BITS * s)) & ((1u << FR_HH_PH_BITS) - 1u);
        hash = 1099511628211ULL * hash + (uint32_t)((s + 1u) * scr[ph]);
    }
    return hash;
}
// What identifies a stored index.  Molecules: the determinant.  Hubbard-Holstein: what the reference's HashTable can tell
// apart -- the bucket (hash % table size) and the first ceil(2 n_sites / 8) bytes (det_hash.hpp:47, :60-94): states with
// the same electrons, different phonons and the same bucket share one entry there, and so they do here.
__device__ __forceinline__ det_t fr_vec_key(const VecDev &v, det_t d) {
    if (!v.hh_sites) return d;
    unsigned long long bucket = fr_hh_hash(d, v.hh_scr, v.hh_sites) % v.hh_buckets;
    unsigned key_bytes = (2 * v.hh_sites + 7) / 8;
    det_t tmask = (1ull << (8 * key_bytes)) - 1ull;        // key_bytes <= 3
    return (bucket << 24) | (d & tmask);
}
__device__ __forceinline__ uint32_t fr_hash_slot(det_t d, uint32_t hcap) { return (uint32_t)fr_mix64(d) & (hcap - 1); }
// returns the slot holding determinant dd's entry or FR_NOPOS
__device__ __forceinline__ uint32_t fr_hash_find(const VecDev &v, det_t dd) {
    const det_t d = fr_vec_key(v, dd);
    uint32_t s = fr_hash_slot(d, v.hcap);
    for (uint32_t probe = 0; probe < v.hcap; probe++) {
        det_t k = v.hs[s].key;
        if (k == d) return s;
        if (k == FR_EMPTY_KEY) return FR_NOPOS;
        s = (s + 1) & (v.hcap - 1);
    }
    return FR_NOPOS;
}
