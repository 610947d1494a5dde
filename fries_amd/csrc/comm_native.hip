// Native transports behind include/fries_hip.h's fries_comm -- the reference's MPI calls on MPI_COMM_WORLD
// (FRIES/vec_utils.hpp:991-1019 MPI_Alltoallv inside Adder::perform_add; FRIES/compress_utils.hpp:170-231
// MPI_Allgather inside every sum_mpi), without Python in the loop:
//
//   * RCCL: one process per MI355X, librccl over xGMI.  ncclAllGather on the 2 KB staging block, ncclAllToAllv on the
//     spawn records, both enqueued on the engine's own stream (no host synchronisation added by the transport).
//   * local: the ranks are host threads of ONE process, each with its own context and stream, on any devices (several
//     ranks may share a GPU -- RCCL refuses that).  Device-to-device copies between the ranks' staging buffers with a
//     host barrier on both sides.  This is what the multi-rank parity tests use on a one-GPU box (8 ranks of BASELINE
//     config 4), and it lets one process drive several GPUs.
#include "ctx.hpp"
#include <rccl/rccl.h>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <atomic>
#include <chrono>

struct fries_transport {
    int kind = 0;               // 1 RCCL, 2 local
    int rank = 0, size = 1, device = 0;
    uint64_t big_bytes = 0;
    void *small_send = nullptr, *small_recv = nullptr, *big_send = nullptr, *big_recv = nullptr;
    uint64_t n_allgather = 0, n_alltoallv = 0;
    // RCCL
    ncclComm_t nccl = nullptr;
    std::vector<size_t> sc, sd, rc, rd;
    // local
    struct fries_local_group *grp = nullptr;
    std::vector<uint64_t> send_off;         // byte offset of every destination's segment in my big_send (this collective)
    // host collectives (an MPI program's own MPI_Allgather / MPI_Alltoallv on host memory)
    fries_host_collectives host{};
    uint8_t *h_small = nullptr, *h_small_all = nullptr, *h_big_send = nullptr, *h_big_recv = nullptr;       // pinned
};

struct fries_local_group {
    int size = 0;
    uint64_t big_bytes = 0;
    std::vector<fries_transport *> member;
    std::mutex mu;
    std::condition_variable cv;
    int waiting = 0;
    uint64_t generation = 0;
    std::atomic<int> failed{0};
    void barrier() {
        std::unique_lock<std::mutex> lk(mu);
        const uint64_t gen = generation;
        if (++waiting == size) { waiting = 0; generation++; cv.notify_all(); }
        else if (!cv.wait_for(lk, std::chrono::seconds(300), [&] { return generation != gen; })) {
            // a rank never arrived (its engine raised, or its thread died): fail the group instead of hanging the process
            failed = 1; waiting = 0; generation++; cv.notify_all();
        }
    }
};

static void tr_alloc(fries_transport *t) {
    FR_HIP(hipSetDevice(t->device));
    t->small_send = fr_alloc<uint8_t>(FRIES_COMM_SMALL_BYTES);
    t->small_recv = fr_alloc<uint8_t>((size_t)FRIES_COMM_SMALL_BYTES * t->size);
    t->big_send = fr_alloc<uint8_t>(t->big_bytes);
    t->big_recv = fr_alloc<uint8_t>(t->big_bytes);
    FR_HIP(hipMemset(t->small_send, 0, FRIES_COMM_SMALL_BYTES));
    FR_HIP(hipMemset(t->small_recv, 0, (size_t)FRIES_COMM_SMALL_BYTES * t->size));
}

#define FR_NCCL(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { fr_set_error(std::string(#call) + ": " + ncclGetErrorString(r_)); return 1; } } while (0)

// ------------------------------------------------------------------ RCCL
static int rccl_allgather(void *user, uint64_t bytes, void *stream) {
    fries_transport *t = (fries_transport *)user;
    FR_NCCL(ncclAllGather(t->small_send, t->small_recv, (size_t)bytes, ncclChar, t->nccl, (hipStream_t)stream));
    t->n_allgather++;
    return 0;
}
static int rccl_alltoallv(void *user, const uint64_t *send_bytes, const uint64_t *recv_bytes, void *stream) {
    fries_transport *t = (fries_transport *)user;
    size_t so = 0, ro = 0;
    for (int p = 0; p < t->size; p++) {
        t->sc[p] = (size_t)send_bytes[p]; t->sd[p] = so; so += t->sc[p];
        t->rc[p] = (size_t)recv_bytes[p]; t->rd[p] = ro; ro += t->rc[p];
    }
    if (so > t->big_bytes || ro > t->big_bytes) { fr_set_error("all-to-all segments exceed the staging buffers"); return 1; }
    FR_NCCL(ncclAllToAllv(t->big_send, t->sc.data(), t->sd.data(), t->big_recv, t->rc.data(), t->rd.data(), ncclChar, t->nccl, (hipStream_t)stream));
    t->n_alltoallv++;
    return 0;
}

extern "C" int fries_rccl_unique_id(uint8_t id[128]) {
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId u;
    FR_NCCL(ncclGetUniqueId(&u));
    memcpy(id, &u, 128);
    return 0;
}

extern "C" int fries_rccl_create(fries_transport **out, const uint8_t id[128], int rank, int size, int device, uint64_t big_bytes) {
    try {
        if (size < 1 || rank < 0 || rank >= size || size > FRIES_COMM_MAX_RANKS) throw FriesError("bad rank / size");
        fries_transport *t = new fries_transport();
        t->kind = 1; t->rank = rank; t->size = size; t->device = device; t->big_bytes = big_bytes;
        tr_alloc(t);
        t->sc.resize(size); t->sd.resize(size); t->rc.resize(size); t->rd.resize(size);
        ncclUniqueId u;
        memcpy(&u, id, 128);
        FR_NCCL(ncclCommInitRank(&t->nccl, size, u, rank));
        *out = t;
        return 0;
    } catch (const std::exception &e) { fr_set_error(e.what()); return 1; }
}

// ------------------------------------------------------------------ local (threads of one process)
static int local_allgather(void *user, uint64_t bytes, void *stream) {
    fries_transport *t = (fries_transport *)user;
    fries_local_group *g = t->grp;
    hipStream_t st = (hipStream_t)stream;
    int bad = 0;
    if (hipStreamSynchronize(st) != hipSuccess) bad = 1;                // my block is final
    if (bad) g->failed = 1;
    g->barrier();
    for (int p = 0; p < g->size && !bad; p++)
        if (hipMemcpyAsync((uint8_t *)t->small_recv + (size_t)p * bytes, g->member[p]->small_send, (size_t)bytes, hipMemcpyDefault, st) != hipSuccess) bad = 1;
    if (!bad && hipStreamSynchronize(st) != hipSuccess) bad = 1;
    if (bad) g->failed = 1;
    g->barrier();                                                        // nobody refills its block before everybody has read it
    t->n_allgather++;
    if (g->failed) { fr_set_error("local transport: a device copy failed on some rank"); return 1; }
    return 0;
}
static int local_alltoallv(void *user, const uint64_t *send_bytes, const uint64_t *recv_bytes, void *stream) {
    fries_transport *t = (fries_transport *)user;
    fries_local_group *g = t->grp;
    hipStream_t st = (hipStream_t)stream;
    uint64_t so = 0;
    for (int p = 0; p < g->size; p++) { t->send_off[p] = so; so += send_bytes[p]; }
    int bad = so > t->big_bytes ? 1 : 0;
    if (hipStreamSynchronize(st) != hipSuccess) bad = 1;
    if (bad) g->failed = 1;
    g->barrier();
    uint64_t ro = 0;
    for (int s = 0; s < g->size && !bad; s++) {
        const fries_transport *src = g->member[s];
        if (ro + recv_bytes[s] > t->big_bytes) { bad = 1; break; }
        if (recv_bytes[s] && hipMemcpyAsync((uint8_t *)t->big_recv + ro, (const uint8_t *)src->big_send + src->send_off[t->rank], (size_t)recv_bytes[s], hipMemcpyDefault, st) != hipSuccess) bad = 1;
        ro += recv_bytes[s];
    }
    if (!bad && hipStreamSynchronize(st) != hipSuccess) bad = 1;
    if (bad) g->failed = 1;
    g->barrier();
    t->n_alltoallv++;
    if (g->failed) { fr_set_error("local transport: a device copy failed or a segment exceeds the staging buffers"); return 1; }
    return 0;
}

extern "C" int fries_local_group_create(fries_local_group **out, int size, uint64_t big_bytes) {
    if (size < 1 || size > FRIES_COMM_MAX_RANKS) { fr_set_error("bad group size"); return 1; }
    fries_local_group *g = new fries_local_group();
    g->size = size; g->big_bytes = big_bytes; g->member.assign(size, nullptr);
    *out = g;
    return 0;
}
extern "C" void fries_local_group_destroy(fries_local_group *g) { delete g; }

extern "C" int fries_local_create(fries_transport **out, fries_local_group *g, int rank, int device) {
    try {
        if (!g || rank < 0 || rank >= g->size) throw FriesError("bad rank");
        fries_transport *t = new fries_transport();
        t->kind = 2; t->rank = rank; t->size = g->size; t->device = device; t->big_bytes = g->big_bytes; t->grp = g;
        t->send_off.assign(g->size, 0);
        tr_alloc(t);
        { std::lock_guard<std::mutex> lk(g->mu); g->member[rank] = t; }
        *out = t;
        return 0;
    } catch (const std::exception &e) { fr_set_error(e.what()); return 1; }
}

// ------------------------------------------------------------------ host collectives (kind 3)
// The staging blocks cross to pinned host memory, the caller's collective runs there (MPI on host buffers: no GPU-aware MPI needed),
// the result goes back: stream-ordered on both sides, two host synchronisations per collective.  This is what lets an MPI program
// written against include/FRIES run the engine on one GPU per rank with nothing but its own MPI_COMM_WORLD.
static int host_allgather(void *user, uint64_t bytes, void *stream) {
    fries_transport *t = (fries_transport *)user;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemcpyAsync(t->h_small, t->small_send, (size_t)bytes, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { fr_set_error("host transport: staging copy failed"); return 1; }
    if (t->host.allgather(t->host.user, t->h_small, t->h_small_all, bytes)) { fr_set_error("host transport: the caller's all-gather failed"); return 1; }
    if (hipMemcpyAsync(t->small_recv, t->h_small_all, (size_t)bytes * t->size, hipMemcpyHostToDevice, st) != hipSuccess) { fr_set_error("host transport: staging copy failed"); return 1; }
    t->n_allgather++;
    return 0;
}
static int host_alltoallv(void *user, const uint64_t *send_bytes, const uint64_t *recv_bytes, void *stream) {
    fries_transport *t = (fries_transport *)user;
    hipStream_t st = (hipStream_t)stream;
    uint64_t so = 0, ro = 0;
    for (int p = 0; p < t->size; p++) { so += send_bytes[p]; ro += recv_bytes[p]; }
    if (so > t->big_bytes || ro > t->big_bytes) { fr_set_error("all-to-all segments exceed the staging buffers"); return 1; }
    if ((so && hipMemcpyAsync(t->h_big_send, t->big_send, (size_t)so, hipMemcpyDeviceToHost, st) != hipSuccess) || hipStreamSynchronize(st) != hipSuccess) { fr_set_error("host transport: staging copy failed"); return 1; }
    if (t->host.alltoallv(t->host.user, t->h_big_send, send_bytes, t->h_big_recv, recv_bytes)) { fr_set_error("host transport: the caller's all-to-all failed"); return 1; }
    if (ro && hipMemcpyAsync(t->big_recv, t->h_big_recv, (size_t)ro, hipMemcpyHostToDevice, st) != hipSuccess) { fr_set_error("host transport: staging copy failed"); return 1; }
    t->n_alltoallv++;
    return 0;
}
extern "C" int fries_hostcomm_create(fries_transport **out, const fries_host_collectives *cb, int rank, int size, int device, uint64_t big_bytes) {
    try {
        if (!cb || !cb->allgather || !cb->alltoallv) throw FriesError("both host collectives are required");
        if (size < 1 || rank < 0 || rank >= size || size > FRIES_COMM_MAX_RANKS) throw FriesError("bad rank / size");
        fries_transport *t = new fries_transport();
        t->kind = 3; t->rank = rank; t->size = size; t->device = device; t->big_bytes = big_bytes; t->host = *cb;
        tr_alloc(t);
        FR_HIP(hipHostMalloc((void **)&t->h_small, FRIES_COMM_SMALL_BYTES, hipHostMallocDefault));
        FR_HIP(hipHostMalloc((void **)&t->h_small_all, (size_t)FRIES_COMM_SMALL_BYTES * size, hipHostMallocDefault));
        FR_HIP(hipHostMalloc((void **)&t->h_big_send, big_bytes, hipHostMallocDefault));
        FR_HIP(hipHostMalloc((void **)&t->h_big_recv, big_bytes, hipHostMallocDefault));
        *out = t;
        return 0;
    } catch (const std::exception &e) { fr_set_error(e.what()); return 1; }
}

// ------------------------------------------------------------------ common
extern "C" int fries_transport_comm(fries_transport *t, fries_comm *cm) {
    if (!t || !cm) { fr_set_error("null transport"); return 1; }
    cm->user = t; cm->rank = t->rank; cm->size = t->size;
    cm->small_send = t->small_send; cm->small_recv = t->small_recv; cm->big_send = t->big_send; cm->big_recv = t->big_recv;
    cm->big_bytes = t->big_bytes;
    cm->allgather = t->kind == 1 ? rccl_allgather : (t->kind == 3 ? host_allgather : local_allgather);
    cm->alltoallv = t->kind == 1 ? rccl_alltoallv : (t->kind == 3 ? host_alltoallv : local_alltoallv);
    return 0;
}
extern "C" int fries_transport_counts(fries_transport *t, uint64_t *n_allgather, uint64_t *n_alltoallv) {
    if (!t) return 1;
    if (n_allgather) *n_allgather = t->n_allgather;
    if (n_alltoallv) *n_alltoallv = t->n_alltoallv;
    return 0;
}
extern "C" void fries_transport_destroy(fries_transport *t) {
    if (!t) return;
    hipSetDevice(t->device);
    if (t->nccl) ncclCommDestroy(t->nccl);
    hipFree(t->small_send); hipFree(t->small_recv); hipFree(t->big_send); hipFree(t->big_recv);
    if (t->h_small) { hipHostFree(t->h_small); hipHostFree(t->h_small_all); hipHostFree(t->h_big_send); hipHostFree(t->h_big_recv); }
    delete t;
}
