/* MI355X build only (no counterpart in the reference): what the FRIES/*.hpp headers of this build share -- the device context the
 * solution vector gets bound to, the molecule parse_fcidump read last, and a registry that maps a raw pointer handed out by
 * DistVec::values() (or the indices() matrix passed to apply_HBPP_sys) back to its vector: the reference's free functions take raw
 * pointers (find_preserve(double *, ...), apply_HBPP_sys(Matrix<uint8_t> &all_orbs, Matrix<uint8_t> &all_dets, ...)).
 *
 * Division of labour behind these headers: trial vectors and other small set-up vectors live on the host exactly as in the reference;
 * the solution vector moves to the device (DistVec<double> with two columns at the first apply_HBPP_sys, DistVec<int> at its first
 * perform_add, HubHolVec<double> at the first comp_sub) and from then on every operator on it -- apply_HBPP_sys, comp_sub,
 * Adder::perform_add, add_vecs, zero_vec, find_preserve, sys_comp, dot -- is a call into libfries_hip.so; its host arrays are mirrors
 * that are refreshed when the program asks for a pointer.  The compression operators refuse host vectors instead of falling back to a
 * CPU implementation.
 *
 * Ranks: the program's MPI_COMM_WORLD.  One process per GPU (FRIES_DEVICE, or the launcher's local rank, else rank modulo the number of
 * devices); the host side routes its adds and sums with MPI as the reference does, and the engine's own collectives (the sum_mpi's
 * inside find_keep_sub / find_preserve / sys_comp) reach MPI through a host-collectives transport (fries_hostcomm_create). */
#ifndef FRIES_BACKEND_HPP
#define FRIES_BACKEND_HPP
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>
#include <mpi.h>
#include "../fries_hip.h"

namespace fries_hip {
inline void ck(int rc) { if (rc) throw std::runtime_error(fries_last_error()); }
/* what parse_hh_input read (FRIES/io_utils.hpp), for the set-up of the Hubbard-Holstein vector's device context (FRIES/hh_vec.hpp) */
struct HHParams { unsigned n_elec = 0, lat_len = 0; double eps = 0, U = 0, omega = 0, g = 0, gs_energy = 0; bool set = false; };
inline HHParams &hh_params() { static HHParams p; return p; }
struct DeviceVecBase {
    virtual ~DeviceVecBase() {}
    virtual bool owns(const void *p) const = 0;             // p points into one of the value mirrors
    virtual size_t offset_of(const void *p) const = 0;      // element offset of p inside its column
    virtual const void *indices_key() const = 0;            // address of the indices() matrix
    virtual bool bound() const = 0;
    virtual void bind(uint32_t mat_nonz, bool new_hb) = 0;  // host content -> device; no-op when bound
    virtual void before_device_op() = 0;                    // host mirrors the program may have written through -> device
    virtual void after_device_op(bool col0, bool col1, bool layout) = 0;   // which mirrors are stale now
    virtual size_t dense_size() const = 0;                  // positions in front that belong to the dense (semi-stochastic) space
    virtual bool hh_candidate() const { return false; }     // a Hubbard-Holstein solution vector that has not moved to the device yet
    virtual void hh_budget(uint32_t /*n_samp*/) {}          // ... is told the sample budget of the driver's compressions before it moves
    virtual fries_ctx *ctx() = 0;
};
inline int mpi_rank() { int r = 0; MPI_Comm_rank(MPI_COMM_WORLD, &r); return r; }
inline int mpi_size() { int n = 1; MPI_Comm_size(MPI_COMM_WORLD, &n); return n; }
inline int mpi_allgather_cb(void *, const void *send, void *recv, uint64_t bytes) {
    return MPI_Allgather((void *)send, (int)bytes, MPI_BYTE, recv, (int)bytes, MPI_BYTE, MPI_COMM_WORLD) != MPI_SUCCESS;
}
inline int mpi_alltoallv_cb(void *, const void *send, const uint64_t *send_bytes, void *recv, const uint64_t *recv_bytes) {
    int n = 1;
    MPI_Comm_size(MPI_COMM_WORLD, &n);
    std::vector<int> sc(n), sd(n), rc(n), rd(n);
    int so = 0, ro = 0;
    for (int p = 0; p < n; p++) { sc[p] = (int)send_bytes[p]; sd[p] = so; so += sc[p]; rc[p] = (int)recv_bytes[p]; rd[p] = ro; ro += rc[p]; }
    return MPI_Alltoallv((void *)send, sc.data(), sd.data(), MPI_BYTE, recv, rc.data(), rd.data(), MPI_BYTE, MPI_COMM_WORLD) != MPI_SUCCESS;
}
struct Backend {
    uint32_t n_orb = 0, n_elec = 0;
    std::vector<uint8_t> symm; std::vector<double> hcore, eris;
    const void *eris_obj = nullptr, *hcore_obj = nullptr;   // the objects parse_fcidump returned (identity check in the matrix-element calls)
    bool have_mol = false;
    std::vector<DeviceVecBase *> vecs;
    static Backend &get() { static Backend b; return b; }
    static int device_for_rank(int rank) {
        if (const char *dv = getenv("FRIES_DEVICE")) return atoi(dv);
        const int nd = fries_device_count();
        if (nd <= 0) return 0;
        for (const char *k : {"LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", "MPI_LOCALRANKID", "PMI_LOCAL_RANK"}) if (const char *v = getenv(k)) return atoi(v) % nd;
        return rank % nd;
    }
    fries_ctx *ctx() {
        if (!ctx_) {
            if (!have_mol && !hh_mode) throw std::runtime_error("no Hamiltonian: parse_fcidump (or parse_hh_input) must run before anything that needs the device");
            int rank = 0;
            MPI_Comm_rank(MPI_COMM_WORLD, &rank);
            device_ = device_for_rank(rank);
            ck(fries_ctx_create(&ctx_, device_));
            if (have_mol) ck(fries_set_molecule(ctx_, n_orb, n_elec, symm.data(), hcore.data(), eris.data()));
        }
        return ctx_;
    }
    /* before the driver's setup on the context: with more than one MPI rank the engine gets the program's communicator */
    void attach_comm(uint32_t spawn_cap) {
        int n = 1, rank = 0;
        MPI_Comm_size(MPI_COMM_WORLD, &n);
        MPI_Comm_rank(MPI_COMM_WORLD, &rank);
        if (n <= 1 || transport_) return;
        fries_host_collectives cb{nullptr, mpi_allgather_cb, mpi_alltoallv_cb};
        ck(fries_hostcomm_create(&transport_, &cb, rank, n, device_, 16ull * ((uint64_t)spawn_cap + 4096)));
        fries_comm cm;
        ck(fries_transport_comm(transport_, &cm));
        ck(fries_set_comm(ctx(), &cm));
    }
    bool ctx_taken = false;                                 // one bound vector per context
    /* bulk traffic of the mirrors over PCIe, per direction (FRIES_FACADE_STATS=1 prints the totals when the program ends) */
    uint64_t bytes_to_device = 0, bytes_to_host = 0, device_ops = 0;
    bool hh_mode = false;                                   // parse_hh_input ran: the context runs the Hubbard-Holstein model
    DeviceVecBase *owner_of(const void *p) { for (auto *v : vecs) if (v->owns(p)) return v; return nullptr; }
    DeviceVecBase *by_indices(const void *key) { for (auto *v : vecs) if (v->indices_key() == key) return v; return nullptr; }
    DeviceVecBase *bound_vec() { for (auto *v : vecs) if (v->bound()) return v; return nullptr; }
    void add(DeviceVecBase *v) { vecs.push_back(v); }
    void remove(DeviceVecBase *v) { for (size_t i = 0; i < vecs.size(); i++) if (vecs[i] == v) { vecs.erase(vecs.begin() + i); return; } }
    ~Backend() {
        if (getenv("FRIES_FACADE_STATS")) fprintf(stderr, "fries facade: %llu bytes host->device, %llu bytes device->host, %llu device operators\n",
                                                  (unsigned long long)bytes_to_device, (unsigned long long)bytes_to_host, (unsigned long long)device_ops);
        if (ctx_) fries_ctx_destroy(ctx_);
        if (transport_) fries_transport_destroy(transport_);
    }
private:
    fries_ctx *ctx_ = nullptr;
    fries_transport *transport_ = nullptr;
    int device_ = 0;
};
}  // namespace fries_hip
#endif
