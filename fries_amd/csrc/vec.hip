// Device-resident DistVec: open-addressed determinant hash, deterministic annihilating merge,
// ordered deletion.  Reference behaviour being reproduced: FRIES/vec_utils.hpp:418-476,
// 606-641 (add / add_elements / del_at_pos) and FRIES/det_hash.hpp:60-147.
//
// Ordering contract (vec_utils.hpp:618-626): a determinant that enters the vector takes the
// most recently freed position if any (LIFO stack), else the next position at the end, and
// determinants enter in the order their first initiator spawn arrives.  Values accumulate in
// arrival order.  Both orders are reproduced exactly: first-arrival ranks come from an
// atomicMin on the new hash slot plus a prefix sum, and accumulation runs over a stable
// radix sort of (position, pass) keys so each position sums its contributions sequentially.
#include "ctx.hpp"
#include <chrono>
#include <cstring>

// ------------------------------------------------------------------ allocation / state
// every slot empty: key FR_EMPTY_KEY (0), position FR_NOPOS
static __global__ void __launch_bounds__(FR_BLOCK) k_hash_clear(HSlot *hs, uint32_t n) {
    const uint4 e = make_uint4(0u, 0u, FR_NOPOS, 0u);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) ((uint4 *)hs)[i] = e;
}
static void fr_hash_clear(FriesCtx *c, VecDev *v, uint32_t n_slots) {
    static_assert(FR_EMPTY_KEY == 0ull, "k_hash_clear writes the empty key as zero");
    unsigned g = fr_blocks(n_slots, FR_BLOCK);
    FR_LAUNCH(c, "k_hash_clear", k_hash_clear, dim3(g > 2048 ? 2048 : g), dim3(FR_BLOCK), v->hs, n_slots);
}
void fr_vec_alloc(FriesCtx *c, VecDev *v, uint32_t cap) {
    v->cap = cap; v->n_dense = 0;
    uint32_t h = 1024;
    while (h < 2u * cap + 1024u) h <<= 1;
    v->hcap_max = h;
    v->hcap = h < (1u << 14) ? h : (1u << 14);
    v->used_ub = 0;
    v->dets = fr_alloc<det_t>(cap); v->v0 = fr_alloc<double>(cap); v->v1 = fr_alloc<double>(cap);
    v->diag = fr_alloc<double>(cap); v->active = fr_alloc<uint8_t>(cap); v->free_stack = fr_alloc<uint32_t>(cap);
    v->hs = fr_alloc<HSlot>(h);
    v->stat_part = fr_alloc<unsigned long long>((size_t)FR_STAT_STRIPES * FR_STAT_STRIDE);
    FR_HIP(hipMemsetAsync(v->stat_part, 0, 8 * (size_t)FR_STAT_STRIPES * FR_STAT_STRIDE, c->stream));
    v->st = fr_alloc<VecState>(1);
    FR_HIP(hipMemsetAsync(v->dets, 0, sizeof(det_t) * cap, c->stream));
    FR_HIP(hipMemsetAsync(v->v0, 0, 8 * (size_t)cap, c->stream));
    FR_HIP(hipMemsetAsync(v->v1, 0, 8 * (size_t)cap, c->stream));
    FR_HIP(hipMemsetAsync(v->diag, 0xff, 8 * (size_t)cap, c->stream));     // all-ones = NaN
    FR_HIP(hipMemsetAsync(v->active, 0, cap, c->stream));
    fr_hash_clear(c, v, h);
    FR_HIP(hipMemsetAsync(v->st, 0, sizeof(VecState), c->stream));
}

// (word != nullptr: the copy is what the host waits for -- the wave raises the ticket itself once its stores are out, instead of a k_ticket launch behind it)
static __global__ void __launch_bounds__(64) k_readback(const uint32_t *src, uint32_t *dst, unsigned n_words, uint32_t *word, uint32_t ticket) {
    for (unsigned i = threadIdx.x; i < n_words; i += 64) dst[i] = src[i];
    if (word && threadIdx.x == 0) __hip_atomic_store(word, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);      // one wave: the release waits for every lane's stores
}
void fr_rb_init(FriesCtx *c) {
    if (c->h_rb) return;
    void *hp = nullptr, *dp = nullptr;
    const size_t tot = FriesCtx::RB_BYTES + FriesCtx::RB_HELD_BYTES + FriesCtx::MISC_BYTES;
    FR_HIP(hipHostMalloc(&hp, tot, hipHostMallocMapped | hipHostMallocCoherent));
    FR_HIP(hipHostGetDevicePointer(&dp, hp, 0));
    memset(hp, 0, tot);
    c->h_rb = (uint8_t *)hp; c->d_rb = (uint8_t *)dp; c->rb_used = 0;
    c->wait_by_sync = getenv("FRIES_WAIT_SYNC") && atoi(getenv("FRIES_WAIT_SYNC")) != 0;
}
static __global__ void k_ticket(uint32_t *word, uint32_t ticket) { __hip_atomic_store(word, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
// The wait in two halves: the ticket is enqueued where the host needs the stream's results, kernels enqueued after it run while the host
// waits for it (run_stage: the kernels that follow a stage's closing pass, which leave at once if the pass did not settle the stage).
uint32_t fr_stream_ticket(FriesCtx *c) {
    fr_rb_init(c);
    if (c->wait_by_sync) return 0u;
    const uint32_t t = ++c->ticket;
    FR_LAUNCH(c, "k_ticket", k_ticket, dim3(1), dim3(1), c->d_misc(), t);
    return t;
}
// A ticket that a kernel of the caller's raises itself (k_readback at its end; k_seq_sums when it STARTS: everything enqueued before it has finished
// then, which is all a ticket says) -- no launch of its own.  0: the host waits by hipStreamSynchronize, nobody raises anything.
uint32_t fr_ticket_reserve(FriesCtx *c) {
    fr_rb_init(c);
    if (c->wait_by_sync) return 0u;
    return ++c->ticket;
}
void fr_stream_wait_ticket(FriesCtx *c, uint32_t t) {
    if (c->wait_by_sync) { FR_HIP(hipStreamSynchronize(c->stream)); return; }
    volatile uint32_t *w = c->h_misc();
    const auto t0 = std::chrono::steady_clock::now();
    auto reached = [&]() { return (int32_t)(__atomic_load_n((const uint32_t *)w, __ATOMIC_ACQUIRE) - t) >= 0; };      // (a later ticket may already have passed)
    for (uint64_t spin = 0; !reached(); spin++) {
        __builtin_ia32_pause();
        if ((spin & 0xFFFFF) == 0xFFFFF && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {      // a long kernel, or a fault: let the runtime say which
            FR_HIP(hipStreamSynchronize(c->stream));
            if (!reached()) throw FriesError("fr_stream_wait: the stream drained without the ticket kernel having run");
            break;
        }
    }
}
void fr_stream_wait(FriesCtx *c) { fr_stream_wait_ticket(c, fr_stream_ticket(c)); }
static size_t fr_rb_take(FriesCtx *c, size_t bytes, bool held) {
    fr_rb_init(c);
    const size_t need = (bytes + 63) & ~(size_t)63;
    if (bytes == 0 || (bytes & 3) || need > 2048) throw FriesError("fr_readback: bad size");
    if (held) return FriesCtx::RB_BYTES;        // a slot of its own behind the ring: the ring may wrap any number of times before the caller reads it
    if (c->rb_used + need > FriesCtx::RB_BYTES) c->rb_used = 0;
    const size_t off = c->rb_used;
    c->rb_used += need;
    return off;
}
const void *fr_readback(FriesCtx *c, const void *src, size_t bytes, bool held, uint32_t *ticket) {
    const size_t off = fr_rb_take(c, bytes, held);
    const uint32_t t = ticket ? fr_ticket_reserve(c) : 0u;
    if (ticket) *ticket = t;
    FR_LAUNCH(c, "k_readback", k_readback, dim3(1), dim3(64), (const uint32_t *)src, (uint32_t *)(c->d_rb + off), (unsigned)(bytes / 4), t ? c->d_misc() : (uint32_t *)nullptr, t);
    return c->h_rb + off;
}

// folds the striped counters into the state, then hands the state to the host block
static __global__ void __launch_bounds__(64) k_vec_state_out(VecDev V, VecState *dst, uint32_t *word, uint32_t ticket) {
    const int l = threadIdx.x;
    unsigned long long x[3];
    for (int q = 0; q < 3; q++) {
        unsigned long long *p = &V.stat_part[(size_t)l * FR_STAT_STRIDE + q];
        const bool have = V.stat_part && l < FR_STAT_STRIPES;       // (the flat array the pivotal matrix compression wraps in a VecDev has no counters)
        x[q] = have ? *p : 0ull;
        if (have) *p = 0ull;
        for (int off = 32; off > 0; off >>= 1) x[q] += __shfl_xor(x[q], off);
    }
    if (l == 0) {
        VecState s = *V.st;
        s.nonini_occ_add += x[0]; s.n_used += (uint32_t)x[1]; s.n_tomb -= (uint32_t)x[2];
        *V.st = s; *dst = s;
        if (word) __hip_atomic_store(word, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
void fr_vec_sync_state(FriesCtx *c, VecDev *v, VecState *out) {
    const size_t off = fr_rb_take(c, sizeof(VecState), false);
    const uint32_t t = fr_ticket_reserve(c);
    FR_LAUNCH(c, "k_vec_state_out", k_vec_state_out, dim3(1), dim3(64), *v, (VecState *)(c->d_rb + off), t ? c->d_misc() : (uint32_t *)nullptr, t);
    fr_stream_wait_ticket(c, t);
    memcpy(out, c->h_rb + off, sizeof(VecState));
    v->used_ub = out->n_used;
}

void fr_spawn_alloc(FriesCtx *c, uint32_t cap) {
    SpawnBuf &s = c->sp;
    s.cap = cap;
    s.det = fr_alloc<det_t>(cap); s.val = fr_alloc<double>(cap); s.ini = fr_alloc<uint8_t>(cap);
    s.slot = fr_alloc<uint32_t>(cap); s.flag = fr_alloc<uint32_t>(cap);
    for (int h = 0; h < 2; h++) { s.key[h] = fr_alloc<uint32_t>(cap); s.pay[h] = fr_alloc<uint32_t>(cap); }
    s.hist = fr_alloc<uint32_t>((size_t)256 * FR_MAX_PART + 256);
    s.pcnt = fr_alloc<uint32_t>(FR_MAX_PART);
    s.n_spawn = fr_alloc<uint32_t>(1);
    FR_HIP(hipMemsetAsync(s.n_spawn, 0, 4, c->stream));
}

// ------------------------------------------------------------------ merge kernels
// M1: look every spawn up; initiator spawns claim a slot for unseen determinants.
// mode 0: frisys_mol's merge (target column 1, origin column 0).  Modes 1 / 2: perform_add into the
// origin column itself -- first the initiator spawns claim slots (1), then the others look up (2), so
// that a non-initiator never misses a determinant an earlier initiator spawn of the same batch created.
// mode 3: like 0, but the list is ONE arrival order (frifull_hh adds initiators and the others interleaved), no pass bit in the sort key.
__global__ void __launch_bounds__(FR_BLOCK) k_spawn_lookup(VecDev V, SpawnBuf S, uint32_t n_elec, int mode) {
    const uint32_t n = *S.n_spawn;
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    bool created = false, reused = false;   // counters are bumped once per wave at the end (one address, ~1e6 lanes)
    bool bad_nelec = false, hash_full = false;
    if (j < n) {
        const det_t dd = S.det[j];
        bool ini = S.ini[j];
        if (!((mode == 1 && !ini) || (mode == 2 && ini))) {
            const det_t emask = V.hh_sites ? (1ull << (2 * V.hh_sites)) - 1ull : ~0ull;
            if ((uint32_t)__popcll(dd & emask) != n_elec) { bad_nelec = true; S.slot[j] = FR_NOPOS; }
            else {
                const det_t d = fr_vec_key(V, dd);
                uint32_t s = fr_hash_slot(d, V.hcap), found = FR_NOPOS, first_tomb = FR_NOPOS, hv = FR_NEWBIT;
                // A new entry takes the first tombstone of its probe chain once the chain has been walked to its end without finding the key
                // (otherwise a determinant that is deleted and spawned again every iteration -- the rule in fciqmc_fp_mol, common in
                // frisys_mol -- lengthens its own chain by one slot per incarnation until the next rebuild).  Every lane inserting d follows
                // the same rule, so concurrent inserters of d meet at the same slot and the CAS (old == d) merges them.
                for (uint32_t probe = 0; probe < V.hcap; probe++) {
                    const uint4 raw = *(const uint4 *)&V.hs[s];         // key and position in one load
                    const det_t k = (det_t)raw.x | ((det_t)raw.y << 32);
                    if (k == d) { found = s; hv = raw.z; break; }
                    if (k == FR_TOMB_KEY && ini && first_tomb == FR_NOPOS) first_tomb = s;
                    if (k == FR_EMPTY_KEY) {
                        if (!ini) break;
                        const bool use_tomb = first_tomb != FR_NOPOS;
                        const uint32_t tgt = use_tomb ? first_tomb : s;
                        const det_t expect = use_tomb ? FR_TOMB_KEY : FR_EMPTY_KEY;
                        det_t old = atomicCAS((unsigned long long *)&V.hs[tgt].key, (unsigned long long)expect, (unsigned long long)d);
                        if (old == expect) { found = tgt; if (use_tomb) reused = true; else created = true; break; }
                        if (old == d) { found = tgt; hv = V.hs[tgt].val; break; }
                        // another determinant took the slot: probe on from the one after it
                        s = tgt; first_tomb = FR_NOPOS;
                    }
                    s = (s + 1) & (V.hcap - 1);
                }
                if (found == FR_NOPOS) {
                    if (ini) hash_full = true;
                    S.slot[j] = FR_NOPOS;
                }
                else {
                    // S.slot: FR_POSBIT | position for a determinant that was stored before this merge, the hash slot for one that is being
                    // created in it (its position is assigned by k_spawn_assign)
                    if (ini) {
                        if (hv & FR_NEWBIT) { atomicMin(&V.hs[found].val, FR_NEWBIT | j); S.slot[j] = found; }   // being created in this merge
                        else S.slot[j] = FR_POSBIT | hv;
                    }
                    else {
                        // Non-initiator spawns only reach determinants that were present before this merge and are non-zero in the origin
                        // column (vec_utils.hpp:617, 632-637).  A determinant created in this merge starts with zero in both columns
                        // (k_spawn_assign), so both conditions are the one test of the origin column k_seg_sum makes per position.
                        S.slot[j] = (hv & FR_NEWBIT) ? found : (FR_POSBIT | hv);
                    }
                }
            }
        }
    }
    {   // slots taken / tombstones re-used: one striped atomic per workgroup (VecDev::stat_part)
        __shared__ uint32_t shc[8];
        const unsigned long long mc = __ballot(created), mr = __ballot(reused);
        if (fr_lane() == 0) { shc[threadIdx.x >> 6] = (uint32_t)__popcll(mc); shc[4 + (threadIdx.x >> 6)] = (uint32_t)__popcll(mr); }
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t nc = shc[0] + shc[1] + shc[2] + shc[3], nr = shc[4] + shc[5] + shc[6] + shc[7];
            unsigned long long *sp = &V.stat_part[(size_t)(blockIdx.x % FR_STAT_STRIPES) * FR_STAT_STRIDE];
            if (nc) atomicAdd(&sp[1], (unsigned long long)nc);
            if (nr) atomicAdd(&sp[2], (unsigned long long)nr);
        }
    }
    if (__any(bad_nelec) && fr_lane() == 0) atomicOr(&V.st->err, FR_ERR_NELEC);
    if (__any(hash_full) && fr_lane() == 0) atomicOr(&V.st->err, FR_ERR_HASH_FULL);
}

// M2: flag the first arrival of every new determinant
__global__ void __launch_bounds__(FR_BLOCK) k_spawn_first(VecDev V, SpawnBuf S) {
    __shared__ uint32_t shu[4];
    const uint32_t n = *S.n_spawn;
    const unsigned nblk = (n + FR_TILE - 1) / FR_TILE;
    if (blockIdx.x >= nblk) return;
    size_t base = (size_t)blockIdx.x * FR_TILE + (size_t)threadIdx.x * FR_ITEMS;
    uint32_t cnt = 0;
    for (int it = 0; it < FR_ITEMS; it++) {
        size_t j = base + it;
        if (j >= n) break;
        uint32_t s = S.slot[j], f = 0;
        if (s != FR_NOPOS && !(s & FR_POSBIT) && S.ini[j] && V.hs[s].val == (FR_NEWBIT | (uint32_t)j)) f = 1;
        S.flag[j] = f; cnt += f;
    }
    uint32_t bc = fr_block_sum_u32(cnt, shu);
    if (threadIdx.x == 0) S.pcnt[blockIdx.x] = bc;
}

// M3: rank -> position (free stack top first, then append), initialise the new entries
__global__ void __launch_bounds__(FR_BLOCK) k_spawn_assign(VecDev V, SpawnBuf S) {
    __shared__ uint32_t shu[4];
    const uint32_t n = *S.n_spawn;
    const unsigned nblk = (n + FR_TILE - 1) / FR_TILE;
    if (blockIdx.x >= nblk) return;
    uint32_t off;
    { uint32_t x = 0; for (unsigned i = threadIdx.x; i < blockIdx.x; i += blockDim.x) x += S.pcnt[i]; off = fr_block_sum_u32(x, shu); }
    const uint32_t n_free = V.st->n_free, curr = V.st->curr_size;
    size_t base = (size_t)blockIdx.x * FR_TILE + (size_t)threadIdx.x * FR_ITEMS;
    uint32_t f[FR_ITEMS], tsum = 0;
#pragma unroll
    for (int it = 0; it < FR_ITEMS; it++) { size_t j = base + it; f[it] = j < n ? S.flag[j] : 0; tsum += f[it]; }
    uint32_t tot;
    uint32_t incl = fr_block_scan_u32(tsum, shu, &tot);
    uint32_t r = off + incl - tsum;
#pragma unroll
    for (int it = 0; it < FR_ITEMS; it++) {
        size_t j = base + it;
        if (!f[it]) continue;
        uint32_t pos = r < n_free ? V.free_stack[n_free - 1 - r] : curr + (r - n_free);
        r++;
        if (pos >= V.cap) { atomicOr(&V.st->err, FR_ERR_CAP); continue; }
        V.dets[pos] = S.det[j]; V.v0[pos] = 0; V.v1[pos] = 0;
        V.diag[pos] = __longlong_as_double(-1ll);     // NaN: diagonal element not cached yet
        V.active[pos] = 1;
        V.hs[S.slot[j]].val = pos;
    }
}

// M4: sort keys (position, pass) and the vector's bookkeeping
__global__ void __launch_bounds__(FR_BLOCK) k_spawn_resolve(VecDev V, SpawnBuf S, uint32_t *key, uint32_t *pay, uint32_t drop_key, int mode) {
    __shared__ uint32_t shu[4];
    const uint32_t n = *S.n_spawn;
    if (blockIdx.x == 0) {
        const unsigned nblk = (n + FR_TILE - 1) / FR_TILE;
        uint32_t n_new = fr_sum_partials_u32(S.pcnt, nblk, shu);
        if (threadIdx.x == 0) {
            VecState *st = V.st;
            uint32_t from_stack = n_new < st->n_free ? n_new : st->n_free;
            st->n_free -= from_stack;
            st->curr_size += n_new - from_stack;
            st->n_nonz += (int32_t)n_new;
            if (st->curr_size > V.cap) st->err |= FR_ERR_CAP;
        }
    }
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    uint32_t s = S.slot[j];
    uint32_t k = drop_key;
    if (s != FR_NOPOS) { uint32_t pos = (s & FR_POSBIT) ? (s & ~FR_POSBIT) : V.hs[s].val; if (pos < V.cap) k = (pos << 1) | ((mode == 0 && S.ini[j]) ? 1u : 0u); }
    key[j] = k; pay[j] = j;
}

// ------------------------------------------------------------------ stable LSD radix sort, 8-bit digits
__global__ void __launch_bounds__(FR_BLOCK) k_rs_hist(const uint32_t *key, const uint32_t *n_ptr, uint32_t *hist, int shift, uint32_t nblk_alloc) {
    __shared__ uint32_t h[256];
    const uint32_t n = *n_ptr;
    const unsigned nblk = (n + FR_TILE - 1) / FR_TILE;
    if (blockIdx.x >= nblk) return;
    h[threadIdx.x] = 0;
    __syncthreads();
    size_t base = (size_t)blockIdx.x * FR_TILE;
    for (int it = 0; it < FR_ITEMS; it++) {
        size_t j = base + it * FR_BLOCK + threadIdx.x;
        if (j < n) atomicAdd(&h[(key[j] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblk_alloc + blockIdx.x] = h[threadIdx.x];
}

// one block per digit: exclusive scan of that digit's counts over the tiles; total -> hist tail
__global__ void __launch_bounds__(FR_BLOCK) k_rs_scan(const uint32_t *n_ptr, uint32_t *hist, uint32_t nblk_alloc) {
    __shared__ uint32_t shu[4];
    const uint32_t n = *n_ptr;
    const unsigned nblk = (n + FR_TILE - 1) / FR_TILE;
    uint32_t *row = hist + (size_t)blockIdx.x * nblk_alloc;
    uint32_t carry = 0;
    for (unsigned b0 = 0; b0 < nblk; b0 += FR_BLOCK) {
        unsigned b = b0 + threadIdx.x;
        uint32_t x = b < nblk ? row[b] : 0, tot;
        uint32_t incl = fr_block_scan_u32(x, shu, &tot);
        if (b < nblk) row[b] = carry + incl - x;
        carry += tot;
    }
    if (threadIdx.x == 0) hist[(size_t)256 * nblk_alloc + blockIdx.x] = carry;
}

__global__ void __launch_bounds__(FR_BLOCK) k_rs_scatter(const uint32_t *key, const uint32_t *pay, uint32_t *key_out, uint32_t *pay_out,
                                                         const uint32_t *n_ptr, const uint32_t *hist, int shift, uint32_t nblk_alloc) {
    __shared__ uint32_t wcount[4][256];
    __shared__ uint32_t gbase[256];
    __shared__ uint32_t shu[4];
    const uint32_t n = *n_ptr;
    const unsigned nblk = (n + FR_TILE - 1) / FR_TILE;
    if (blockIdx.x >= nblk) return;
    const int lane = fr_lane(), w = threadIdx.x >> 6;
    for (int q = 0; q < 4; q++) wcount[q][threadIdx.x] = 0;
    {   // digit bases: exclusive scan of the 256 totals + this tile's offset within the digit
        uint32_t t = hist[(size_t)256 * nblk_alloc + threadIdx.x], tot;
        uint32_t incl = fr_block_scan_u32(t, shu, &tot);
        gbase[threadIdx.x] = incl - t + hist[(size_t)threadIdx.x * nblk_alloc + blockIdx.x];
    }
    __syncthreads();
    // each wave owns 256 consecutive keys, visited as 4 rounds of 64 consecutive keys
    size_t wbase = (size_t)blockIdx.x * FR_TILE + (size_t)w * 256;
    uint32_t kk[4], pp[4], rk[4];
    for (int r = 0; r < 4; r++) {
        size_t j = wbase + r * 64 + lane;
        bool ok = j < n;
        kk[r] = ok ? key[j] : 0xFFFFFFFFu; pp[r] = ok ? pay[j] : 0;
        uint32_t dg = (kk[r] >> shift) & 255u;
        unsigned long long m = __ballot(ok);
        for (int b = 0; b < 8; b++) {
            unsigned long long bal = __ballot((dg >> b) & 1u);
            m &= ((dg >> b) & 1u) ? bal : ~bal;
        }
        if (!ok) m = 0;
        uint32_t before = __popcll(m & ((1ull << lane) - 1ull));
        uint32_t prev = ok ? wcount[w][dg] : 0;
        rk[r] = prev + before;
        __builtin_amdgcn_wave_barrier();
        if (ok && before == 0) wcount[w][dg] = prev + (uint32_t)__popcll(m);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // cross-wave exclusive prefix per digit (thread d handles digit d)
    {
        uint32_t a0 = wcount[0][threadIdx.x], a1 = wcount[1][threadIdx.x], a2 = wcount[2][threadIdx.x];
        __syncthreads();
        wcount[0][threadIdx.x] = 0; wcount[1][threadIdx.x] = a0; wcount[2][threadIdx.x] = a0 + a1; wcount[3][threadIdx.x] = a0 + a1 + a2;
    }
    __syncthreads();
    for (int r = 0; r < 4; r++) {
        size_t j = wbase + r * 64 + lane;
        if (j < n) {
            uint32_t dg = (kk[r] >> shift) & 255u;
            uint32_t o = gbase[dg] + wcount[w][dg] + rk[r];
            key_out[o] = kk[r]; pay_out[o] = pp[r];
        }
    }
}

// M6: each position sums its contributions sequentially in (pass, arrival) order.
// Every lane fetches its own entry of the sorted list -- key, arrival index, initiator flag, origin-column test and value, all independent
// loads -- into LDS; the lane at the head of a segment then adds the segment up from LDS (walking the list from global memory is three
// dependent loads per addend), and goes back to global memory only where the segment runs past its workgroup's 256 entries.
__global__ void __launch_bounds__(FR_BLOCK) k_seg_sum(VecDev V, SpawnBuf S, const uint32_t *key, const uint32_t *pay, uint32_t drop_key, int mode) {
    __shared__ double sval[FR_BLOCK];
    __shared__ uint32_t skey[FR_BLOCK];
    __shared__ uint8_t sfl[FR_BLOCK];           // bit 0: initiator spawn, bit 1: value fetched
    __shared__ uint32_t sh_cont, sh_done;       // the head whose segment runs past this workgroup's 256 entries (thread + 1; at most one: the last segment)
    const uint32_t n = *S.n_spawn;
    const uint32_t t0 = blockIdx.x * blockDim.x, t = t0 + threadIdx.x;
    if (t0 >= n) return;
    const bool rule_fixed = mode == 0 || mode == 3;     // the initiator rule looks at the origin column as it was before the merge
    if (threadIdx.x == 0) { sh_cont = 0u; sh_done = 0u; }
    // entry g of the sorted list into slot threadIdx.x of the LDS arrays: key, initiator flag, origin-column test and value -- all independent loads
    auto stage = [&](uint32_t g, uint32_t *k_out, bool *ini_out, bool *occ_out) {
        const uint32_t k = g < n ? key[g] : drop_key;
        const bool live = k != drop_key;
        bool ini = false, occ = false;
        uint32_t j = 0;
        if (live) {
            j = pay[g];
            ini = mode == 0 ? (k & 1u) != 0 : S.ini[j] != 0;
            // non-initiator spawns only reach determinants that are non-zero in the origin column (vec_utils.hpp:617, 632-637); one created
            // in this merge has zero there (k_spawn_assign)
            if (rule_fixed) occ = V.v0[k >> 1] != 0;
        }
        const bool fetch = live && (!rule_fixed || ini || occ);
        const double val = fetch ? S.val[j] : 0.0;
        skey[threadIdx.x] = k; sval[threadIdx.x] = val; sfl[threadIdx.x] = (uint8_t)((ini ? 1u : 0u) | (fetch ? 2u : 0u));
        *k_out = k; *ini_out = ini; *occ_out = occ;
    };
    uint32_t k; bool ini, occ;
    stage(t, &k, &ini, &occ);
    const bool live = k != drop_key;
    const uint32_t pos = k >> 1;
    const uint32_t kprev = (t > 0 && live) ? key[t - 1] : drop_key;
    unsigned long long occ_add = (rule_fixed && live && !ini && occ) ? 1ull : 0ull;
    __syncthreads();
    const bool head = live && !(t > 0 && kprev != drop_key && (kprev >> 1) == pos);
    double acc = 0;
    // the segment's entries in the LDS arrays from slot u on; returns the slot it stopped at (FR_BLOCK: the segment may go on)
    auto add_up = [&](uint32_t u) {
        for (; u < FR_BLOCK; u++) {
            const uint32_t ku = skey[u];
            if (ku == drop_key || (ku >> 1) != pos) break;
            const uint32_t f = sfl[u];
            if (rule_fixed) { if (f & 2u) acc += sval[u]; }
            else {
                const bool nonz = acc != 0;
                occ_add += (!(f & 1u) && nonz);
                if ((f & 1u) || nonz) acc += sval[u];
            }
        }
        return u;
    };
    bool mine_goes_on = false;
    if (head) {
        acc = rule_fixed ? V.v1[pos] : V.v0[pos];
        if (add_up(threadIdx.x) == FR_BLOCK && t0 + FR_BLOCK < n) { mine_goes_on = true; sh_cont = threadIdx.x + 1u; }
        else if (rule_fixed) V.v1[pos] = acc; else V.v0[pos] = acc;
    }
    __syncthreads();
    // A segment that runs past the workgroup's entries (a determinant that many spawns reach: one lane walking the list in global memory pays three
    // dependent loads per addend, and that one lane was the kernel's time): the workgroup stages the next 256 entries the same way and the head goes on
    // in LDS.  The workgroup those entries belong to sees no head there and leaves them alone.
    if (sh_cont) {
        for (uint32_t g0 = t0 + FR_BLOCK; g0 < n; g0 += FR_BLOCK) {
            uint32_t k2; bool i2, o2;
            stage(g0 + threadIdx.x, &k2, &i2, &o2);
            __syncthreads();
            if (mine_goes_on && add_up(0u) < FR_BLOCK) sh_done = 1u;
            __syncthreads();
            if (sh_done) break;
        }
        if (mine_goes_on) { if (rule_fixed) V.v1[pos] = acc; else V.v0[pos] = acc; }
    }
    // one counter, ~1e6 segments: added up per workgroup, then one striped atomic (VecDev::stat_part)
    for (int off = 32; off > 0; off >>= 1) occ_add += __shfl_xor(occ_add, off);
    __shared__ unsigned long long sho[4];
    __syncthreads();
    if (fr_lane() == 0) sho[threadIdx.x >> 6] = occ_add;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long tot = sho[0] + sho[1] + sho[2] + sho[3];
        if (tot) atomicAdd(&V.stat_part[(size_t)(blockIdx.x % FR_STAT_STRIPES) * FR_STAT_STRIDE], tot);
    }
}

static int bits_for(uint32_t x) { int b = 0; while ((1ull << b) <= x) b++; return b; }

// Adds the spawn list (c->sp, length *sp.n_spawn <= n_bound) to column 1 of the vector with
// the reference's two-pass initiator rule relative to column 0.
void fr_vec_merge(FriesCtx *c, VecDev *v, uint32_t n_bound, bool same_column, bool arrival_order) {
    if (n_bound == 0) return;
    SpawnBuf &S = c->sp;
    hipStream_t st = c->stream;
    if (n_bound > S.cap) throw FriesError("spawn list exceeds spawn buffer capacity");
    fr_vec_reserve_hash(c, v, n_bound);
    v->used_ub = (uint32_t)((uint64_t)v->used_ub + n_bound < 0xFFFFFFFFull ? v->used_ub + n_bound : 0xFFFFFFFFu);
    unsigned g1 = fr_blocks(n_bound, FR_BLOCK), gt = fr_blocks(n_bound, FR_TILE);
    uint32_t nblk_alloc = FR_MAX_PART;
    int nbits = bits_for(2u * v->cap + 1u);
    uint32_t drop_key = (nbits >= 32) ? 0xFFFFFFFFu : ((1u << nbits) - 1u);
    const int mode = same_column ? 1 : (arrival_order ? 3 : 0);
    FR_LAUNCH(c, "k_spawn_lookup", k_spawn_lookup, dim3(g1), dim3(FR_BLOCK), *v, S, c->n_elec, mode);
    if (same_column) FR_LAUNCH(c, "k_spawn_lookup", k_spawn_lookup, dim3(g1), dim3(FR_BLOCK), *v, S, c->n_elec, 2);
    FR_LAUNCH(c, "k_spawn_first", k_spawn_first, dim3(gt), dim3(FR_BLOCK), *v, S);
    FR_LAUNCH(c, "k_spawn_assign", k_spawn_assign, dim3(gt), dim3(FR_BLOCK), *v, S);
    FR_LAUNCH(c, "k_spawn_resolve", k_spawn_resolve, dim3(g1), dim3(FR_BLOCK), *v, S, S.key[0], S.pay[0], drop_key, mode);
    int src = 0;
    for (int shift = 0; shift < nbits; shift += 8) {
        FR_LAUNCH(c, "k_rs_hist", k_rs_hist, dim3(gt), dim3(FR_BLOCK), S.key[src], S.n_spawn, S.hist, shift, nblk_alloc);
        FR_LAUNCH(c, "k_rs_scan", k_rs_scan, dim3(256), dim3(FR_BLOCK), S.n_spawn, S.hist, nblk_alloc);
        FR_LAUNCH(c, "k_rs_scatter", k_rs_scatter, dim3(gt), dim3(FR_BLOCK), S.key[src], S.pay[src], S.key[src ^ 1], S.pay[src ^ 1], S.n_spawn, S.hist, shift, nblk_alloc);
        src ^= 1;
    }
    FR_LAUNCH(c, "k_seg_sum", k_seg_sum, dim3(g1), dim3(FR_BLOCK), *v, S, S.key[src], S.pay[src], drop_key, mode);
}

// ------------------------------------------------------------------ deletion (vec_utils.hpp:458-476)
__global__ void __launch_bounds__(FR_BLOCK) k_del_count(VecDev V, const uint8_t *flags, uint32_t *pcnt) {
    __shared__ uint32_t shu[4];
    const uint32_t n = V.st->curr_size;
    const unsigned nblk = (n + FR_TILE - 1) / FR_TILE;
    if (blockIdx.x >= nblk) return;
    size_t base = (size_t)blockIdx.x * FR_TILE + (size_t)threadIdx.x * FR_ITEMS;
    uint32_t cnt = 0;
    for (int it = 0; it < FR_ITEMS; it++) {
        size_t i = base + it;
        if (i < n && flags[i] && V.active[i] && V.v0[i] == 0 && V.v1[i] == 0) cnt++;
    }
    uint32_t bc = fr_block_sum_u32(cnt, shu);
    if (threadIdx.x == 0) pcnt[blockIdx.x] = bc;
}
__global__ void __launch_bounds__(FR_BLOCK) k_del_apply(VecDev V, uint8_t *flags, const uint32_t *pcnt) {
    __shared__ uint32_t shu[4];
    const uint32_t n = V.st->curr_size;
    const unsigned nblk = (n + FR_TILE - 1) / FR_TILE;
    if (blockIdx.x >= nblk) return;
    uint32_t off;
    { uint32_t x = 0; for (unsigned i = threadIdx.x; i < blockIdx.x; i += blockDim.x) x += pcnt[i]; off = fr_block_sum_u32(x, shu); }
    const uint32_t n_free = V.st->n_free;
    size_t base = (size_t)blockIdx.x * FR_TILE + (size_t)threadIdx.x * FR_ITEMS;
    uint32_t f[FR_ITEMS], tsum = 0;
#pragma unroll
    for (int it = 0; it < FR_ITEMS; it++) {
        size_t i = base + it;
        f[it] = (i < n && flags[i] && V.active[i] && V.v0[i] == 0 && V.v1[i] == 0) ? 1u : 0u;
        if (i < n) flags[i] = 0;
        tsum += f[it];
    }
    uint32_t tot;
    uint32_t incl = fr_block_scan_u32(tsum, shu, &tot);
    uint32_t r = off + incl - tsum;
#pragma unroll
    for (int it = 0; it < FR_ITEMS; it++) {
        size_t i = base + it;
        if (!f[it]) continue;
        V.free_stack[n_free + r] = (uint32_t)i;      // ascending positions: the highest ends on top
        r++;
        V.active[i] = 0;
        uint32_t s = fr_hash_find(V, V.dets[i]);
        if (s != FR_NOPOS) { V.hs[s].key = FR_TOMB_KEY; V.hs[s].val = FR_NOPOS; }
    }
}
__global__ void k_del_finish(VecDev V, const uint32_t *pcnt) {
    __shared__ uint32_t shu[4];
    const uint32_t n = V.st->curr_size;
    const unsigned nblk = (n + FR_TILE - 1) / FR_TILE;
    uint32_t tot = fr_sum_partials_u32(pcnt, nblk, shu);
    if (threadIdx.x == 0) { V.st->n_free += tot; V.st->n_nonz -= (int32_t)tot; V.st->n_tomb += tot; }
}

void fr_vec_delete_flagged(FriesCtx *c, VecDev *v, const uint8_t *d_flags, uint32_t n_bound) {
    if (n_bound == 0) return;
    unsigned gt = fr_blocks(n_bound, FR_TILE);
    FR_LAUNCH(c, "k_del_count", k_del_count, dim3(gt), dim3(FR_BLOCK), *v, d_flags, c->sp.pcnt);
    FR_LAUNCH(c, "k_del_apply", k_del_apply, dim3(gt), dim3(FR_BLOCK), *v, (uint8_t *)d_flags, c->sp.pcnt);
    FR_LAUNCH(c, "k_del_finish", k_del_finish, dim3(1), dim3(FR_BLOCK), *v, c->sp.pcnt);
}

// ------------------------------------------------------------------ tombstone cleanup
__global__ void k_hash_reinsert(VecDev V) {
    const uint32_t n = V.st->curr_size;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { V.st->n_tomb = 0; V.st->n_used = (uint32_t)V.st->n_nonz; }
    if (i < FR_STAT_STRIPES) { V.stat_part[(size_t)i * FR_STAT_STRIDE + 1] = 0ull; V.stat_part[(size_t)i * FR_STAT_STRIDE + 2] = 0ull; }      // (what the merges counted since the last fold is superseded)
    if (i >= n || !V.active[i]) return;
    det_t d = fr_vec_key(V, V.dets[i]);
    uint32_t s = fr_hash_slot(d, V.hcap);
    for (uint32_t probe = 0; probe < V.hcap; probe++) {
        det_t old = atomicCAS((unsigned long long *)&V.hs[s].key, (unsigned long long)FR_EMPTY_KEY, (unsigned long long)d);
        if (old == FR_EMPTY_KEY) { V.hs[s].val = i; return; }
        s = (s + 1) & (V.hcap - 1);
    }
    atomicOr(&V.st->err, FR_ERR_HASH_FULL);
}

// Clears the table and re-inserts every active position into a table of `want` slots (a power of two <= hcap_max).
static void hash_rebuild(FriesCtx *c, VecDev *v, uint32_t want, uint32_t n_pos_bound) {
    v->hcap = want;
    fr_hash_clear(c, v, v->hcap);
    FR_LAUNCH(c, "k_hash_reinsert", k_hash_reinsert, dim3(fr_blocks(n_pos_bound ? n_pos_bound : 1, FR_BLOCK)), dim3(FR_BLOCK), *v);
}
static uint32_t hash_slots_for(const VecDev *v, uint64_t n_entries) {
    static const int fill = getenv("FRIES_HASH_FILL") ? atoi(getenv("FRIES_HASH_FILL")) : 40;     // percent of the slots the entries may take
    uint64_t want = 1u << 14;
    while (want * (uint64_t)fill < n_entries * 100u && want < v->hcap_max) want <<= 1;
    return (uint32_t)(want < v->hcap_max ? want : v->hcap_max);
}

void fr_vec_maybe_rebuild(FriesCtx *c, VecDev *v) {
    // host copy of the state must be current.  Rebuild when tombstones + entries pass 60 % of the slots; the new size follows the live set.
    if ((uint64_t)c->h_vst.n_used * 10 < (uint64_t)v->hcap * 6) return;
    hash_rebuild(c, v, hash_slots_for(v, c->h_vst.curr_size), c->h_vst.curr_size);
    v->used_ub = c->h_vst.curr_size;
}

// before a merge of up to n_new spawns: room for every one of them to be a new determinant
void fr_vec_reserve_hash(FriesCtx *c, VecDev *v, uint32_t n_new) {
    const uint64_t need = (uint64_t)v->used_ub + n_new;
    if (need * 10 < (uint64_t)v->hcap * 7) return;
    const uint32_t want = hash_slots_for(v, need);
    if (want == v->hcap && want == v->hcap_max) return;          // as large as it gets: the rebuild-on-tombstones rule applies as before
    hash_rebuild(c, v, want > v->hcap ? want : v->hcap, v->cap);       // positions in use are not known on the host here: launch over the capacity
    // used_ub stays: an upper bound is all it has to be (the next sync makes it exact)
}

// ------------------------------------------------------------------ spawn exchange between ranks
// Adder::add routes every pending element to idx_to_proc(det) (vec_utils.hpp:418-423, 360-379) and
// perform_add ships them with MPI_Alltoallv; the receiver walks its receive buffer source rank by
// source rank, each source's elements in the order of its add() calls (:991-1019, :606-641).
// frisys_mol calls this twice per iteration, first for the non-initiator spawns and then for the
// initiator ones (frisys_mol.cpp:430-471).  Here one exchange carries both: a source's segment for
// a destination is [non-initiators in spawn order | initiators in spawn order] as 16-byte
// (determinant, value) records, and the merge's stable (position, pass) sort restores the
// reference's accumulation order (pass 0 of every source before pass 1 of any).
#define FR_XCH_MAXB (2 * FR_MAX_RANKS)

__device__ __forceinline__ uint32_t fr_proc_of(det_t d, const uint32_t *scr, uint32_t n_ranks) {
    unsigned long long hash = 0;
    uint32_t i = 0;
    while (d) {
        unsigned orb = __ffsll((long long)d) - 1;
        d &= d - 1;
        uint32_t term = (i + 1u) * scr[orb];             // unsigned int * uint32_t: wraps at 32 bits (det_hash.hpp:160-170)
        hash = 1099511628211ULL * hash + term;
        i++;
    }
    return (uint32_t)(hash % n_ranks);
}

// bucket = 2 * destination + initiator flag; counts per 256-element tile
#define FR_XCH_MAXR 1022    // perform_add rounds per pass when the Adder fills up
// sel != nullptr: only the spawns of pass sel_pass whose index lies in round sel_round of this rank, [sel[r], sel[r + 1]) of
// sel = bounds + sel_pass * (FR_XCH_MAXR + 2), take part (the others get key 0xFF)
__global__ void __launch_bounds__(FR_BLOCK) k_xch_keys(SpawnBuf S, const uint32_t *scr, uint32_t n_ranks, uint8_t *key, uint32_t *tile_cnt, uint32_t hh_sites, int one_pass,
                                                       const uint32_t *sel, int sel_pass, int sel_round) {
    __shared__ uint32_t wcnt[4][FR_XCH_MAXB];
    const uint32_t n = *S.n_spawn;
    const uint32_t nb = 2 * n_ranks;
    const uint32_t ntile = (n + FR_BLOCK - 1) / FR_BLOCK;
    if (blockIdx.x >= ntile) return;
    uint32_t j = blockIdx.x * FR_BLOCK + threadIdx.x;
    uint32_t k = 0xFFu;
    if (j < n) {
        uint32_t owner = hh_sites ? (uint32_t)(fr_hh_hash(S.det[j], scr, hh_sites) % n_ranks) : fr_proc_of(S.det[j], scr, n_ranks);
        k = 2 * owner + ((S.ini[j] && !one_pass) ? 1u : 0u);
        if (sel && ((int)(k & 1u) != sel_pass || j < sel[sel_round] || j >= sel[sel_round + 1])) k = 0xFFu;
        key[j] = (uint8_t)k;
    }
    const int w = threadIdx.x >> 6;
    for (uint32_t b = 0; b < nb; b++) {
        unsigned long long m = __ballot(k == b);
        if (fr_lane() == 0) wcnt[w][b] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nb; b += FR_BLOCK) tile_cnt[(size_t)blockIdx.x * nb + b] = wcnt[0][b] + wcnt[1][b] + wcnt[2][b] + wcnt[3][b];
}

// exclusive prefix of every bucket over the tiles, bucket totals, bucket bases (one workgroup)
__global__ void __launch_bounds__(FR_BLOCK) k_xch_scan(SpawnBuf S, uint32_t n_ranks, const uint32_t *tile_cnt, uint32_t *tile_off, uint32_t *bucket /* [2][nb]: totals, bases */, uint32_t *msg) {
    __shared__ uint32_t shu[4];
    __shared__ uint32_t s_tot[FR_XCH_MAXB];
    const uint32_t n = *S.n_spawn;
    const uint32_t nb = 2 * n_ranks;
    const uint32_t ntile = (n + FR_BLOCK - 1) / FR_BLOCK;
    for (uint32_t b = 0; b < nb; b++) {
        uint32_t run = 0;
        for (uint32_t t0 = 0; t0 < ntile; t0 += FR_BLOCK) {
            uint32_t t = t0 + threadIdx.x;
            uint32_t x = t < ntile ? tile_cnt[(size_t)t * nb + b] : 0u;
            uint32_t tot;
            uint32_t incl = fr_block_scan_u32(x, shu, &tot);
            if (t < ntile) tile_off[(size_t)t * nb + b] = run + incl - x;
            run += tot;
            __syncthreads();
        }
        if (threadIdx.x == 0) s_tot[b] = run;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t base = 0;
        for (uint32_t b = 0; b < nb; b++) { bucket[b] = s_tot[b]; bucket[nb + b] = base; msg[b] = s_tot[b]; base += s_tot[b]; }
    }
}

struct XchRec { det_t det; double val; };

__global__ void __launch_bounds__(FR_BLOCK) k_xch_scatter(SpawnBuf S, uint32_t n_ranks, const uint8_t *key, const uint32_t *tile_off, const uint32_t *bucket, XchRec *out, uint32_t cap_recs, uint32_t *err, int one_pass) {
    __shared__ uint32_t wcnt[4][FR_XCH_MAXB];
    const uint32_t n = *S.n_spawn;
    const uint32_t nb = 2 * n_ranks;
    const uint32_t ntile = (n + FR_BLOCK - 1) / FR_BLOCK;
    if (blockIdx.x >= ntile) return;
    uint32_t j = blockIdx.x * FR_BLOCK + threadIdx.x;
    uint32_t k = j < n ? key[j] : 0xFFu;
    const int w = threadIdx.x >> 6, lane = fr_lane();
    uint32_t my_rank = 0;
    for (uint32_t b = 0; b < nb; b++) {
        unsigned long long m = __ballot(k == b);
        if (lane == 0) wcnt[w][b] = (uint32_t)__popcll(m);
        if (k == b) my_rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    }
    __syncthreads();
    if (j < n && k < nb) {
        for (int ww = 0; ww < w; ww++) my_rank += wcnt[ww][k];
        uint32_t o = bucket[nb + k] + tile_off[(size_t)blockIdx.x * nb + k] + my_rank;
        if (o < cap_recs) {
            XchRec r; r.det = S.det[j]; r.val = S.val[j];
            // one pass (fciqmc_mol): initiator and non-initiator spawns keep their order, so the flag rides inside the (integer) walker
            // count: 2 v + sgn(v) * flag, exact for |v| < 2^51
            if (one_pass == 1) r.val = 2.0 * r.val + ((S.ini[j] ? 1.0 : 0.0) * (r.val > 0 ? 1.0 : -1.0));
            // one pass with real values (frifull_hh): the flag rides in bit 63 of the index, which no index of <= 31 orbitals / 12 sites uses
            if (one_pass == 2 && S.ini[j]) r.det |= 1ull << 63;
            out[o] = r;
        }
        else atomicOr(err, FR_ERR_SPAWN_CAP);
    }
}

// seg[s] = {first record, number of pass-0 records, number of records} of source s in the receive buffer
struct XchSegs { uint32_t n_src; uint32_t first[FR_MAX_RANKS], n0[FR_MAX_RANKS], cnt[FR_MAX_RANKS]; };

__global__ void __launch_bounds__(FR_BLOCK) k_xch_unpack(SpawnBuf S, const XchRec *in, XchSegs G, uint32_t n_recv, int one_pass) {
    uint32_t j = blockIdx.x * FR_BLOCK + threadIdx.x;
    if (j == 0) *S.n_spawn = n_recv;
    if (j >= n_recv) return;
    uint32_t s = 0;
    while (s + 1 < G.n_src && j >= G.first[s + 1]) s++;
    XchRec r = in[j];
    if (one_pass == 2) {
        S.det[j] = r.det & ~(1ull << 63); S.val[j] = r.val; S.ini[j] = (uint8_t)(r.det >> 63);
        return;
    }
    if (one_pass) {
        const double a = fabs(r.val);
        const double flag = a - 2.0 * floor(a * 0.5);           // 1 for an initiator spawn
        S.det[j] = r.det; S.ini[j] = flag != 0 ? 1 : 0;
        S.val[j] = (r.val - (r.val > 0 ? flag : -flag)) * 0.5;
        return;
    }
    S.det[j] = r.det; S.val[j] = r.val; S.ini[j] = (j - G.first[s]) >= G.n0[s] ? 1 : 0;
}

// Where this rank's perform_add rounds end.  Adder::add returns false when a destination's buffer reaches adder_size; the driver then stops
// adding, every rank calls perform_add, and the walk over the samples goes on where it stopped (vec_utils.hpp:957-971,
// frisys_mol.cpp:430-471): per pass, round r of this rank is the index range [E_r, E_(r+1)) with E_(r+1) - 1 = the first spawn of the
// pass at which SOME destination has received adder_size spawns since E_r.  One workgroup, thread b = bucket (destination, pass):
// counts below an index and the index of the n-th element of a bucket come from the tiles' prefix counts (k_xch_scan) and a walk
// inside one tile of 256 keys.  bounds[pass][0 .. nr] = E_0 = 0, ..., E_nr = n; msg[pass] = nr.
__global__ void __launch_bounds__(FR_BLOCK) k_xch_rounds(const uint8_t *key, const uint32_t *n_ptr, uint32_t nb, const uint32_t *tile_off, const uint32_t *bucket_tot, uint32_t cap,
                                                         uint32_t *bounds, uint32_t *msg, uint32_t *err) {
    __shared__ uint32_t s_min;
    const uint32_t n = *n_ptr;
    const uint32_t ntile = (n + FR_BLOCK - 1) / FR_BLOCK;
    const uint32_t b = threadIdx.x;
    for (uint32_t p = 0; p < 2; p++) {
        uint32_t *bd = bounds + p * (FR_XCH_MAXR + 2);
        uint32_t E = 0, k = 0;
        bool too_many = false;
        if (threadIdx.x == 0) bd[0] = 0;
        while (true) {
            if (threadIdx.x == 0) s_min = 0xFFFFFFFFu;
            __syncthreads();
            if (b < nb && (b & 1u) == p && cap > 0) {
                uint32_t base;                                  // elements of my bucket below E
                const uint32_t t = E / FR_BLOCK;
                if (t >= ntile) base = bucket_tot[b];
                else { base = tile_off[(size_t)t * nb + b]; for (uint32_t j = t * FR_BLOCK; j < E; j++) base += key[j] == b ? 1u : 0u; }
                const uint64_t target = (uint64_t)base + cap;   // ordinal (from 1) of the element that fills the buffer
                if ((uint64_t)bucket_tot[b] >= target) {
                    uint32_t lo = 0, hi = ntile - 1;            // last tile with fewer than `target` elements of the bucket before it
                    while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if ((uint64_t)tile_off[(size_t)mid * nb + b] < target) lo = mid; else hi = mid - 1; }
                    uint32_t cnt = tile_off[(size_t)lo * nb + b], x = 0xFFFFFFFFu;
                    for (uint32_t j = lo * FR_BLOCK; j < n && j < (lo + 1) * FR_BLOCK; j++) if (key[j] == b && (uint64_t)(++cnt) == target) { x = j; break; }
                    if (x != 0xFFFFFFFFu) atomicMin(&s_min, x);
                }
            }
            __syncthreads();
            const uint32_t B = s_min;
            __syncthreads();
            if (B == 0xFFFFFFFFu) break;
            E = B + 1; k++;
            if (k > FR_XCH_MAXR) { too_many = true; k = FR_XCH_MAXR; break; }
            if (threadIdx.x == 0) bd[k] = E;
        }
        k++;
        if (threadIdx.x == 0) { for (uint32_t q = k; q < FR_XCH_MAXR + 2; q++) bd[q] = n; msg[p] = too_many ? 0xFFFFFFFFu : k; }      // (rounds this rank no longer takes part in are empty: [n, n))
        __syncthreads();
    }
}

void fr_xch_alloc(FriesCtx *c, uint32_t cap) {
    SpawnBuf &S = c->sp;
    if (!c->use_comm) return;
    uint32_t ntile = fr_blocks(cap, FR_BLOCK) + 1;
    S.xkey = fr_alloc<uint8_t>(cap);
    S.xcnt = fr_alloc<uint32_t>((size_t)ntile * 2 * c->n_ranks);
    S.xoff = fr_alloc<uint32_t>((size_t)ntile * 2 * c->n_ranks);
    S.xbucket = fr_alloc<uint32_t>(4 * (size_t)c->n_ranks);
    if ((uint64_t)cap * sizeof(XchRec) > c->comm.big_bytes) throw FriesError("fries_comm.big_bytes is smaller than 16 bytes x (mat_nonz + 4096)");
    if ((size_t)2 * c->n_ranks * 4 > 2048) throw FriesError("too many ranks");
}

// One perform_add: ships the spawns of `src` (the whole list, or with sel the spawns of one pass and round) to their owners; on return
// c->sp holds what this rank received, in the reference's arrival order.  Returns the number received; *overflow (when asked for) =
// some (source, destination, pass) has reached the Adder's capacity and nothing was shipped.
static uint32_t xch_once(FriesCtx *c, const SpawnBuf &src, uint32_t n_local, int one_pass, const uint32_t *sel, int sel_pass, int sel_round, bool *overflow) {
    SpawnBuf &S = c->sp;
    hipStream_t st = c->stream;
    const int P = c->n_ranks;
    const uint32_t nb = 2 * P;
    unsigned g = fr_blocks(n_local ? n_local : 1, FR_BLOCK);
    uint32_t cap_recs = (uint32_t)(c->comm.big_bytes / sizeof(XchRec));
    FR_LAUNCH(c, "k_xch_keys", k_xch_keys, dim3(g), dim3(FR_BLOCK), src, c->d_proc_scr, (uint32_t)P, S.xkey, S.xcnt, c->vec.hh_sites, one_pass, sel, sel_pass, sel_round);
    FR_LAUNCH(c, "k_xch_scan", k_xch_scan, dim3(1), dim3(FR_BLOCK), src, (uint32_t)P, S.xcnt, S.xoff, S.xbucket, (uint32_t *)c->comm.small_send);
    FR_LAUNCH(c, "k_xch_scatter", k_xch_scatter, dim3(g), dim3(FR_BLOCK), src, (uint32_t)P, S.xkey, S.xoff, S.xbucket, (XchRec *)c->comm.big_send, cap_recs, c->d_err, one_pass);
    const uint32_t *all = (const uint32_t *)fr_allgather(c, nb * 4);
    std::vector<uint32_t> cnt((size_t)P * nb);
    FR_HIP(hipMemcpyAsync(cnt.data(), all, cnt.size() * 4, hipMemcpyDeviceToHost, st));
    FR_HIP(hipStreamSynchronize(st));
    std::vector<uint64_t> sb(P), rb(P);
    XchSegs G{};
    G.n_src = (uint32_t)P;
    uint64_t n_recv = 0;
    // Every rank holds the whole P x 2P count matrix, so the limits are checked for ALL (source, destination) pairs on every
    // rank: either everybody raises here or everybody enters the all-to-all -- a rank-local check would leave the peers
    // blocked in the collective while one rank unwinds.  (Capacities are equal on all ranks: same parameters.)
    bool full = false;
    for (int dst = 0; dst < P; dst++) {
        uint64_t tot = 0;
        for (int s2 = 0; s2 < P; s2++) {
            const uint32_t n0 = cnt[(size_t)s2 * nb + 2 * dst], n1 = cnt[(size_t)s2 * nb + 2 * dst + 1];
            // Adder::add returns false once adder_size_ elements are pending for one destination (vec_utils.hpp:957-971) and the driver flushes
            // early: the passes then take several perform_add rounds (fr_xch_rounds).  mat_nonz * 4 / n_ranks (capped at 1e6) per pass is that limit.
            if (!sel && (n0 >= c->adder_cap || n1 >= c->adder_cap)) full = true;
            tot += (uint64_t)n0 + n1;
        }
        if (!full && (tot > S.cap || tot > cap_recs)) throw FriesError("received spawns exceed the spawn buffer on some rank");
    }
    if (full) {
        if (!overflow) throw FriesError("a rank has more pending adds for one destination than the reference's Adder holds (this driver's loop has no early perform_add)");
        *overflow = true;
        return 0;
    }
    for (int p = 0; p < P; p++) {
        const uint32_t *mine = &cnt[(size_t)c->rank * nb], *theirs = &cnt[(size_t)p * nb];
        sb[p] = (uint64_t)sizeof(XchRec) * ((uint64_t)mine[2 * p] + mine[2 * p + 1]);
        uint32_t n0 = theirs[2 * c->rank], n1 = theirs[2 * c->rank + 1];
        G.first[p] = (uint32_t)n_recv; G.n0[p] = n0; G.cnt[p] = n0 + n1;
        rb[p] = (uint64_t)sizeof(XchRec) * ((uint64_t)n0 + n1);
        n_recv += (uint64_t)n0 + n1;
    }
    if (c->comm.alltoallv(c->comm.user, sb.data(), rb.data(), (void *)st)) throw FriesError("fries_comm.alltoallv failed");
    c->n_collectives++;
    FR_LAUNCH(c, "k_xch_unpack", k_xch_unpack, dim3(fr_blocks(n_recv ? n_recv : 1, FR_BLOCK)), dim3(FR_BLOCK), S, (const XchRec *)c->comm.big_recv, G, (uint32_t)n_recv, one_pass);
    return (uint32_t)n_recv;
}

// The passes in several perform_add rounds, each merged into the vector before the next is shipped -- what the reference's loop does when
// Adder::add reports a full buffer (frisys_mol.cpp:430-471): arrival order = (pass, round, source rank, order of the adds).  The keys and
// the tiles' prefix counts of the whole list are in S.xkey / S.xoff / S.xbucket (the attempt that found the buffer full).
static uint32_t fr_xch_rounds(FriesCtx *c, uint32_t n_local) {
    SpawnBuf &S = c->sp;
    hipStream_t st = c->stream;
    const int P = c->n_ranks;
    if (!S.bdet) {
        S.bdet = fr_alloc<det_t>(S.cap); S.bval = fr_alloc<double>(S.cap); S.bini = fr_alloc<uint8_t>(S.cap); S.bn = fr_alloc<uint32_t>(1);
        S.xbounds = fr_alloc<uint32_t>(2 * (FR_XCH_MAXR + 2));
    }
    FR_HIP(hipMemcpyAsync(S.bdet, S.det, sizeof(det_t) * (size_t)n_local, hipMemcpyDeviceToDevice, st));
    FR_HIP(hipMemcpyAsync(S.bval, S.val, 8 * (size_t)n_local, hipMemcpyDeviceToDevice, st));
    FR_HIP(hipMemcpyAsync(S.bini, S.ini, (size_t)n_local, hipMemcpyDeviceToDevice, st));
    FR_HIP(hipMemcpyAsync(S.bn, S.n_spawn, 4, hipMemcpyDeviceToDevice, st));
    FR_LAUNCH(c, "k_xch_rounds", k_xch_rounds, dim3(1), dim3(FR_BLOCK), S.xkey, S.bn, (uint32_t)(2 * P), S.xoff, S.xbucket, c->adder_cap, S.xbounds, (uint32_t *)c->comm.small_send, c->d_err);
    const uint32_t *all = (const uint32_t *)fr_allgather(c, 8);
    std::vector<uint32_t> nr((size_t)2 * P);
    FR_HIP(hipMemcpyAsync(nr.data(), all, nr.size() * 4, hipMemcpyDeviceToHost, st));
    FR_HIP(hipStreamSynchronize(st));
    SpawnBuf src = S;
    src.det = S.bdet; src.val = S.bval; src.ini = S.bini; src.n_spawn = S.bn;
    uint64_t total = 0;
    for (int p = 0; p < 2; p++) {
        uint32_t rounds = 0;
        for (int r = 0; r < P; r++) rounds = nr[(size_t)2 * r + p] > rounds ? nr[(size_t)2 * r + p] : rounds;       // the loop runs while any rank still adds
        if (rounds == 0xFFFFFFFFu) throw FriesError("a pass needs more perform_add rounds than FR_XCH_MAXR (every rank raises this)");
        for (uint32_t k = 0; k < rounds; k++) {
            // a rank that has run out of rounds ships nothing: its bounds beyond the last round all read n
            const uint32_t n_recv = xch_once(c, src, n_local, 0, S.xbounds + p * (FR_XCH_MAXR + 2), p, (int)k, nullptr);
            if (n_recv) fr_vec_merge(c, &c->vec, n_recv, false);
            total += n_recv;
            c->n_adder_rounds++;
        }
    }
    return (uint32_t)(total > 0xFFFFFFFFull ? 0xFFFFFFFFull : total);
}

// Ships the n_local spawns in c->sp to their owners; on return c->sp holds what this rank received, in the
// reference's arrival order.  Returns the number received.  merged != nullptr (frisys_mol's two-pass loop): should the Adder fill up,
// the passes run in rounds, every round is merged into the vector here, *merged = true and the return value is the total received.
uint32_t fr_spawn_exchange(FriesCtx *c, uint32_t n_local, int one_pass, bool *merged) {
    SpawnBuf &S = c->sp;
    if (n_local > S.cap) throw FriesError("spawn list exceeds spawn buffer capacity");
    if (n_local == 0) FR_HIP(hipMemsetAsync(S.n_spawn, 0, 4, c->stream));
    if (merged) *merged = false;
    bool full = false;
    const uint32_t n_recv = xch_once(c, S, n_local, one_pass, nullptr, 0, 0, (merged && one_pass == 0) ? &full : nullptr);
    if (!full) return n_recv;
    *merged = true;
    return fr_xch_rounds(c, n_local);
}
