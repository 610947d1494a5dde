/* MI355X build only (no counterpart in the reference): what the FRIES/*.hpp headers of this build share -- the device context the
 * solution vector gets bound to, the molecule parse_fcidump read last, and a registry that maps a raw pointer handed out by
 * DistVec::values() (or the indices() matrix passed to apply_HBPP_sys) back to its vector: the reference's free functions take raw
 * pointers (find_preserve(double *, ...), apply_HBPP_sys(Matrix<uint8_t> &all_orbs, Matrix<uint8_t> &all_dets, ...)).
 *
 * Division of labour behind these headers: trial vectors and other small set-up vectors live on the host exactly as in the reference;
 * the solution vector (DistVec<double> with two value columns) moves to the device at the first apply_HBPP_sys and from then on every
 * operator on it -- apply_HBPP_sys, Adder::perform_add, add_vecs, zero_vec, find_preserve, sys_comp, dot -- is a call into
 * libfries_hip.so; its host arrays are mirrors that are refreshed when the program asks for a pointer.  The compression operators
 * refuse host vectors instead of falling back to a CPU implementation. */
#ifndef FRIES_BACKEND_HPP
#define FRIES_BACKEND_HPP
#include <cstdint>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>
#include "../fries_hip.h"

namespace fries_hip {
inline void ck(int rc) { if (rc) throw std::runtime_error(fries_last_error()); }
struct DeviceVecBase {
    virtual ~DeviceVecBase() {}
    virtual bool owns(const void *p) const = 0;             // p points into one of the value mirrors
    virtual size_t offset_of(const void *p) const = 0;      // element offset of p inside its column
    virtual const void *indices_key() const = 0;            // address of the indices() matrix
    virtual bool bound() const = 0;
    virtual void bind(uint32_t mat_nonz, bool new_hb) = 0;  // host content -> device; no-op when bound
    virtual void before_device_op() = 0;                    // host mirrors the program may have written through -> device
    virtual void after_device_op(bool col0, bool col1, bool layout) = 0;   // which mirrors are stale now
    virtual fries_ctx *ctx() = 0;
};
struct Backend {
    uint32_t n_orb = 0, n_elec = 0;
    std::vector<uint8_t> symm; std::vector<double> hcore, eris;
    const void *eris_obj = nullptr, *hcore_obj = nullptr;   // the objects parse_fcidump returned (identity check in the matrix-element calls)
    bool have_mol = false;
    std::vector<DeviceVecBase *> vecs;
    static Backend &get() { static Backend b; return b; }
    fries_ctx *ctx() {
        if (!ctx_) {
            if (!have_mol) throw std::runtime_error("no molecule: parse_fcidump must run before anything that needs the device");
            const char *dv = getenv("FRIES_DEVICE");
            ck(fries_ctx_create(&ctx_, dv ? atoi(dv) : 0));
            ck(fries_set_molecule(ctx_, n_orb, n_elec, symm.data(), hcore.data(), eris.data()));
        }
        return ctx_;
    }
    bool ctx_taken = false;                                 // one bound vector per context
    DeviceVecBase *owner_of(const void *p) { for (auto *v : vecs) if (v->owns(p)) return v; return nullptr; }
    DeviceVecBase *by_indices(const void *key) { for (auto *v : vecs) if (v->indices_key() == key) return v; return nullptr; }
    void add(DeviceVecBase *v) { vecs.push_back(v); }
    void remove(DeviceVecBase *v) { for (size_t i = 0; i < vecs.size(); i++) if (vecs[i] == v) { vecs.erase(vecs.begin() + i); return; } }
    ~Backend() { if (ctx_) fries_ctx_destroy(ctx_); }
private:
    fries_ctx *ctx_ = nullptr;
};
}  // namespace fries_hip
#endif
