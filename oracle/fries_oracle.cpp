// TEST INFRASTRUCTURE ONLY -- see fries_oracle.hpp.  Sequential CPU restatement
// of the FRIES hot path; every function cites the reference lines it follows.
// Compile with -ffp-contract=off: the parity contract is plain IEEE-754
// double arithmetic in the reference's literal operation order (no FMA).
#include "fries_oracle.hpp"
#include <cstdlib>
#include <cstdio>
#include <mutex>
#include <condition_variable>
#include <thread>
#include <cmath>
#include <cstring>
#include <algorithm>
#include <stdexcept>
#include <iostream>
#include <climits>

namespace fo {

// ------------------------------------------------------------------ in-process ranks
struct CommShared {
    int size;
    std::mutex mu; std::condition_variable cv;
    int waiting = 0; uint64_t gen = 0;
    std::vector<const void *> slot;
    bool failed = false;
    explicit CommShared(int n) : size(n), slot(n, nullptr) {}
    void barrier() {
        std::unique_lock<std::mutex> lk(mu);
        if (failed) throw std::runtime_error("another rank failed");
        uint64_t g = gen;
        if (++waiting == size) { waiting = 0; gen++; cv.notify_all(); }
        else {
            cv.wait(lk, [&] { return gen != g || failed; });
            if (failed && gen == g) throw std::runtime_error("another rank failed");
        }
    }
    void fail() { std::lock_guard<std::mutex> lk(mu); failed = true; cv.notify_all(); }
};
CommShared *comm_create(int size) { return new CommShared(size); }
void comm_destroy(CommShared *s) { delete s; }
const Comm &Comm::self() { static const Comm c; return c; }
void Comm::barrier() const { if (size > 1) sh->barrier(); }
void Comm::allgather(const void *in, void *out, size_t bytes) const {
    if (size == 1) { memcpy(out, in, bytes); return; }
    sh->slot[rank] = in;
    sh->barrier();
    for (int p = 0; p < size; p++) memcpy((uint8_t *)out + (size_t)p * bytes, sh->slot[p], bytes);
    sh->barrier();
}
void Comm::alltoallv(const std::vector<std::vector<uint8_t>> &send, std::vector<std::vector<uint8_t>> &recv) const {
    recv.assign(size, {});
    if (size == 1) { recv[0] = send[0]; return; }
    sh->slot[rank] = &send;
    sh->barrier();
    for (int p = 0; p < size; p++) recv[p] = (*(const std::vector<std::vector<uint8_t>> *)sh->slot[p])[rank];
    sh->barrier();
}
double Comm::sum(double x) const {
    if (size == 1) { double g = 0; g += x; return g; }
    double all[256];
    allgather(&x, all, sizeof(double));
    double g = 0;
    for (int p = 0; p < size; p++) g += all[p];
    return g;
}
int Comm::sum(int x) const {
    if (size == 1) return x;
    int all[256];
    allgather(&x, all, sizeof(int));
    int g = 0;
    for (int p = 0; p < size; p++) g += all[p];
    return g;
}
void run_ranks(int size, const std::function<void(const Comm &)> &fn) {
    if (size == 1) { fn(Comm::self()); return; }
    CommShared *sh = comm_create(size);
    std::vector<std::thread> th;
    std::vector<std::string> err(size);
    for (int r = 0; r < size; r++) th.emplace_back([&, r] {
        Comm c; c.rank = r; c.size = size; c.sh = sh;
        try { fn(c); } catch (std::exception &e) { err[r] = e.what(); sh->fail(); }
    });
    for (auto &t : th) t.join();
    comm_destroy(sh);
    for (int r = 0; r < size; r++) if (!err[r].empty() && err[r] != "another rank failed") throw std::runtime_error("rank " + std::to_string(r) + ": " + err[r]);
}


static inline size_t tri_wdiag(size_t i, size_t j) { return j * (j + 1) / 2 + i; }      // math_utils.h I_J_TO_TRI_WDIAG (i <= j)
static inline size_t tri_nodiag(size_t i, size_t j) { return j * (j - 1) / 2 + i; }     // math_utils.h I_J_TO_TRI_NODIAG (i < j)
static inline bool bit(det_t d, unsigned i) { return (d >> i) & 1ull; }

// ------------------------------------------------------------------ bit strings
int occ_list(det_t det, uint8_t *occ) {
    int n = 0;
    while (det) {
        occ[n++] = (uint8_t)__builtin_ctzll(det);
        det &= det - 1;
    }
    return n;
}

unsigned bits_between(det_t det, unsigned a, unsigned b) {
    unsigned lo = a < b ? a : b, hi = a < b ? b : a;
    if (hi - lo < 2) return 0;
    det_t mask = ((hi >= 64 ? 0ull : (1ull << hi)) - 1ull) & ~((1ull << (lo + 1)) - 1ull);
    return (unsigned)__builtin_popcountll(det & mask);
}

int excite_sign(unsigned cre, unsigned des, det_t det) {
    return (bits_between(det, cre, des) % 2 == 0) ? 1 : -1;
}

int sing_det_parity(det_t *det, const uint8_t *orbs) {
    *det &= ~(1ull << orbs[0]);
    int sign = excite_sign(orbs[0], orbs[1], *det);
    *det |= 1ull << orbs[1];
    return sign;
}

int sing_parity(det_t det, const uint8_t *orbs) { return excite_sign(orbs[0], orbs[1], det); }

int doub_det_parity(det_t *det, const uint8_t *orbs) {
    *det &= ~(1ull << orbs[0]);
    *det &= ~(1ull << orbs[1]);
    int sign = excite_sign(orbs[2], orbs[0], *det);
    sign *= excite_sign(orbs[3], orbs[1], *det);
    *det |= 1ull << orbs[2];
    *det |= 1ull << orbs[3];
    return sign;
}

int doub_parity(det_t det, const uint8_t *orbs) {
    det &= ~(1ull << orbs[0]);
    det &= ~(1ull << orbs[1]);
    int sign = excite_sign(orbs[2], orbs[0], det);
    sign *= excite_sign(orbs[3], orbs[1], det);
    return sign;
}

det_t sing_det(det_t det, const uint8_t *orbs) { return (det & ~(1ull << orbs[0])) | (1ull << orbs[1]); }
det_t doub_det(det_t det, const uint8_t *orbs) {
    det &= ~(1ull << orbs[0]);
    det &= ~(1ull << orbs[1]);
    return det | (1ull << orbs[2]) | (1ull << orbs[3]);
}

det_t gen_hf_det(unsigned n_orb, unsigned n_elec) {
    det_t half = (1ull << (n_elec / 2)) - 1ull;
    return half | (half << n_orb);
}

uint8_t find_nth_virt(const uint8_t *occ, int spin, unsigned n_elec, unsigned n_orb, unsigned n) {
    unsigned virt = n_orb * spin + n;
    // the reference reads occ[orb_idx] before testing orb_idx < n_elec; a sentinel
    // larger than any orbital reproduces what its contiguous rows yield in practice
    for (size_t k = n_elec / 2 * spin; k < n_elec && occ[k] <= virt; k++) {
        if (occ[k] <= virt) virt++;
    }
    return (uint8_t)virt;
}

// ------------------------------------------------------------------ integrals
double Integrals::chem(unsigned i1, unsigned i2, unsigned i3, unsigned i4) const {
    size_t mn1 = i1 < i2 ? i1 : i2, mx1 = i1 < i2 ? i2 : i1;
    size_t p1 = tri_wdiag(mn1, mx1);
    size_t mn2 = i3 < i4 ? i3 : i4, mx2 = i3 < i4 ? i4 : i3;
    size_t p2 = tri_wdiag(mn2, mx2);
    size_t mnp = p1 < p2 ? p1 : p2, mxp = p1 < p2 ? p2 : p1;
    return eri[tri_wdiag(mnp, mxp)];
}

void Symm::init(const uint8_t *irreps, unsigned n) {
    n_orb = n;
    irrep.assign(irreps, irreps + n);
    lookup.assign((size_t)N_IRREPS * (n + 1), 0);
    for (unsigned idx = 0; idx < n; idx++) {
        unsigned s = irreps[idx];
        unsigned cnt = lookup[s * (n + 1)];
        lookup[s * (n + 1) + 1 + cnt] = (uint8_t)idx;
        lookup[s * (n + 1)] = (uint8_t)(cnt + 1);
    }
    max_n_symm = 0;
    for (unsigned s = 0; s < N_IRREPS; s++) if (lk(s, 0) > max_n_symm) max_n_symm = lk(s, 0);
}

double diag_matrel(const uint8_t *occ, const Integrals &in, unsigned n_elec) {
    unsigned n_orbs = in.n_orb;
    double sum = 0;
    unsigned j, k, e1, e2;
    for (j = 0; j < n_elec / 2; j++) {
        e1 = occ[j];
        sum += in.h[e1 * n_orbs + e1];
        for (k = j + 1; k < n_elec / 2; k++) {
            e2 = occ[k];
            sum += in.phys(e1, e2, e1, e2);
            sum -= in.phys(e1, e2, e2, e1);
        }
        for (k = n_elec / 2; k < n_elec; k++) {
            e2 = occ[k] - n_orbs;
            sum += in.phys(e1, e2, e1, e2);
        }
    }
    for (j = n_elec / 2; j < n_elec; j++) {
        e1 = occ[j] - n_orbs;
        sum += in.h[e1 * n_orbs + e1];
        for (k = j + 1; k < n_elec; k++) {
            e2 = occ[k] - n_orbs;
            sum += in.phys(e1, e2, e1, e2);
            sum -= in.phys(e1, e2, e2, e1);
        }
    }
    return sum;
}

double sing_matrel_nosgn(const uint8_t *ex, const uint8_t *occ, const Integrals &in, unsigned n_elec) {
    unsigned n = in.n_orb;
    unsigned o = ex[0] % n, u = ex[1] % n, spin = ex[0] / n;
    double el = in.h[o * n + u];
    for (unsigned j = 0; j < n_elec / 2; j++) {
        el += in.phys(o, occ[j], u, occ[j]);
        if (spin == 0) el -= in.phys(o, occ[j], occ[j], u);
    }
    for (unsigned j = n_elec / 2; j < n_elec; j++) {
        el += in.phys(o, occ[j] - n, u, occ[j] - n);
        if (spin == 1) el -= in.phys(o, occ[j] - n, occ[j] - n, u);
    }
    return el;
}

double doub_matrel_nosgn(const uint8_t *ex, const Integrals &in) {
    unsigned n = in.n_orb;
    int same = (ex[0] / n) == (ex[1] / n);
    unsigned s0 = ex[0] % n, s1 = ex[1] % n, s2 = ex[2] % n, s3 = ex[3] % n;
    double el = in.phys(s0, s1, s2, s3);
    if (same) el -= in.phys(s0, s1, s3, s2);
    return el;
}

size_t sing_ex_symm(det_t det, const uint8_t *occ, unsigned n_elec, unsigned n_orb, std::vector<uint8_t> &out, const uint8_t *irrep) {
    out.clear();
    for (unsigned i = 0; i < n_elec / 2; i++) {
        unsigned io = occ[i];
        for (unsigned a = 0; a < n_orb; a++)
            if (!bit(det, a) && irrep[io] == irrep[a]) { out.push_back(io); out.push_back(a); }
    }
    for (unsigned i = n_elec / 2; i < n_elec; i++) {
        unsigned io = occ[i];
        for (unsigned a = n_orb; a < 2 * n_orb; a++)
            if (!bit(det, a) && irrep[io - n_orb] == irrep[a - n_orb]) { out.push_back(io); out.push_back(a); }
    }
    return out.size() / 2;
}

size_t doub_ex_symm(det_t det, const uint8_t *occ, unsigned ne, unsigned no, std::vector<uint8_t> &out, const uint8_t *sy) {
    out.clear();
    auto push = [&](unsigned a, unsigned b, unsigned c, unsigned d) { out.push_back(a); out.push_back(b); out.push_back(c); out.push_back(d); };
    unsigned i, j, k, l;
    for (i = 0; i < ne / 2; i++) {
        unsigned io = occ[i];
        for (j = ne / 2; j < ne; j++) {
            unsigned jo = occ[j];
            for (k = 0; k < no; k++) if (!bit(det, k))
                for (l = no; l < 2 * no; l++)
                    if (!bit(det, l) && (sy[io] ^ sy[jo - no] ^ sy[k] ^ sy[l - no]) == 0) push(io, jo, k, l);
        }
    }
    for (i = 0; i < ne / 2; i++) {
        unsigned io = occ[i];
        for (j = i + 1; j < ne / 2; j++) {
            unsigned jo = occ[j];
            for (k = 0; k < no; k++) if (!bit(det, k))
                for (l = k + 1; l < no; l++)
                    if (!bit(det, l) && (sy[io] ^ sy[jo] ^ sy[k] ^ sy[l]) == 0) push(io, jo, k, l);
        }
    }
    for (i = ne / 2; i < ne; i++) {
        unsigned io = occ[i];
        for (j = i + 1; j < ne; j++) {
            unsigned jo = occ[j];
            for (k = no; k < 2 * no; k++) if (!bit(det, k))
                for (l = k + 1; l < 2 * no; l++)
                    if (!bit(det, l) && (sy[io - no] ^ sy[jo - no] ^ sy[k - no] ^ sy[l - no]) == 0) push(io, jo, k, l);
        }
    }
    return out.size() / 4;
}

size_t count_singex(det_t det, const uint8_t *occ, unsigned n_elec, const Symm &s) {
    size_t n = 0;
    for (unsigned e = 0; e < n_elec; e++) {
        unsigned orb = occ[e], sy = s.irrep[orb % s.n_orb], spin = orb / s.n_orb;
        for (unsigned k = 0; k < s.lk(sy, 0); k++)
            if (!bit(det, s.lk(sy, k + 1) + s.n_orb * spin)) n++;
    }
    return n;
}

void count_symm_virt(unsigned counts[][2], const uint8_t *occ, unsigned n_elec, const Symm &s) {
    unsigned i;
    for (i = 0; i < N_IRREPS; i++) { counts[i][0] = s.lk(i, 0); counts[i][1] = s.lk(i, 0); }
    for (i = 0; i < n_elec / 2; i++) counts[s.irrep[occ[i]]][0] -= 1;
    for (; i < n_elec; i++) counts[s.irrep[occ[i] - s.n_orb]][1] -= 1;
}

unsigned count_sing_allowed(const uint8_t *occ, unsigned n_elec, const Symm &s, unsigned counts[][2]) {
    unsigned n = 0;
    for (unsigned e = 0; e < n_elec; e++) {
        unsigned sy = s.irrep[occ[e] % s.n_orb];
        if (counts[sy][e / (n_elec / 2)] != 0) n++;
    }
    return n;
}

unsigned count_sing_virt(const uint8_t *occ, unsigned n_elec, const Symm &s, unsigned counts[][2], uint8_t *occ_choice) {
    unsigned n = 0;
    for (unsigned e = 0; e < n_elec; e++) {
        unsigned sy = s.irrep[occ[e] % s.n_orb];
        unsigned va = counts[sy][e / (n_elec / 2)];
        if (va != 0) {
            if (n == *occ_choice) { *occ_choice = (uint8_t)e; return va; }
            n++;
        }
    }
    return 0;
}

uint8_t virt_from_idx(det_t det, const Symm &s, unsigned irrep, unsigned spin_shift, unsigned index) {
    for (unsigned k = 0; k < s.lk(irrep, 0); k++) {
        unsigned orb = spin_shift + s.lk(irrep, 1 + k);
        if (!bit(det, orb)) {
            if (index == 0) return (uint8_t)orb;
            index--;
        }
    }
    return 255;
}

// ------------------------------------------------------------------ HB-PP tensors
void HBInfo::set_up(const Integrals &in) {
    unsigned n = in.n_orb;
    n_orb = n;
    size_t i, j, a, b;
    d_diff.assign((size_t)n * n, 0.0);
    for (i = 0; i < n; i++) for (j = 0; j < n; j++)
        for (a = 0; a < n; a++) for (b = 0; b < n; b++)
            if (i != a && j != b) d_diff[i * n + j] += fabs(in.phys(i, j, a, b));
    d_same.assign((size_t)n * (n - 1) / 2, 0.0);
    size_t tri = 0;
    for (j = 1; j < n; j++) for (i = 0; i < j; i++) {
        for (a = 0; a < n; a++) for (b = 0; b < a; b++)
            if (a != j && a != i && b != j && b != i)
                d_same[tri] += 2 * fabs(in.phys(i, j, a, b) - in.phys(i, j, b, a));
        tri++;
    }
    s_tens.assign(n, 0.0);
    s_norm = 0;
    for (i = 0; i < n; i++) {
        for (j = 0; j < i; j++) s_tens[i] += d_same[tri_nodiag(j, i)];
        for (j = i + 1; j < n; j++) s_tens[i] += d_same[tri_nodiag(i, j)];
        for (j = 0; j < n; j++) s_tens[i] += d_diff[i * n + j];
        s_norm += s_tens[i];
    }
    exch_sqrt.assign((size_t)n * (n - 1) / 2, 0.0);
    tri = 0;
    for (j = 0; j < n; j++) for (i = 0; i < j; i++) { exch_sqrt[tri] = sqrt(fabs(in.phys(i, j, j, i))); tri++; }
    diag_sqrt.assign(n, 0.0);
    for (j = 0; j < n; j++) diag_sqrt[j] = sqrt(fabs(in.phys(j, j, j, j)));
    exch_norms.assign(n, 0.0);
    for (i = 0; i < n; i++) {
        for (j = 0; j < i; j++) exch_norms[i] += exch_sqrt[tri_nodiag(j, i)];
        exch_norms[i] += diag_sqrt[i];
        for (j = i + 1; j < n; j++) exch_norms[i] += exch_sqrt[tri_nodiag(i, j)];
    }
}

double calc_o1_probs(const HBInfo &t, double *p, unsigned n_elec, const uint8_t *occ, int exclude_first) {
    double norm = 0;
    unsigned skip = exclude_first > 0;
    for (unsigned k = skip; k < n_elec / 2; k++) { p[k - skip] = t.s_tens[occ[k]]; norm += p[k - skip]; }
    for (unsigned k = n_elec / 2; k < n_elec; k++) { p[k - skip] = t.s_tens[occ[k] - t.n_orb]; norm += p[k - skip]; }
    double inv = 1. / norm;
    for (unsigned k = skip; k < n_elec; k++) p[k - skip] *= inv;
    norm /= t.s_norm;
    return norm;
}

double calc_o2_probs(const HBInfo &t, double *p, unsigned n_elec, const uint8_t *occ, unsigned o1_idx) {
    double norm = 0;
    unsigned o1 = occ[o1_idx], n = t.n_orb;
    int sp = o1 / n;
    unsigned off = (1 - sp) * n_elec / 2;
    for (unsigned k = off; k < n_elec / 2 + off; k++) { p[k] = t.d_diff[(o1 % n) * n + occ[k] % n]; norm += p[k]; }
    off = sp * n_elec / 2;
    for (unsigned k = off; k < o1_idx; k++) { p[k] = t.d_same[tri_nodiag(occ[k] % n, o1 % n)]; norm += p[k]; }
    for (unsigned k = o1_idx + 1; k < n_elec / 2 + off; k++) { p[k] = t.d_same[tri_nodiag(o1 % n, occ[k] % n)]; norm += p[k]; }
    p[o1_idx] = 0;
    double inv = 1. / norm;
    for (unsigned k = 0; k < n_elec; k++) p[k] *= inv;
    norm /= t.s_tens[o1 % n];
    return norm;
}

double calc_o2_probs_half(const HBInfo &t, double *p, unsigned n_elec, const uint8_t *occ, unsigned o1_idx) {
    double norm = 0;
    unsigned o1 = occ[o1_idx], n = t.n_orb;
    int sp = o1 / n;
    unsigned upper = n_elec / 2 > o1_idx ? o1_idx : n_elec / 2;
    for (unsigned k = 0; k < upper; k++) {
        if (sp == 0) p[k] = t.d_same[tri_nodiag(occ[k], o1)];
        else p[k] = t.d_diff[(o1 - n) * n + occ[k]];
        norm += p[k];
    }
    for (unsigned k = n_elec / 2; k < o1_idx; k++) {
        if (sp == 0) p[k] = t.d_diff[o1 * n + occ[k] - n];
        else p[k] = t.d_same[tri_nodiag(occ[k] - n, o1 - n)];
        norm += p[k];
    }
    double inv = 1. / norm;
    for (unsigned k = 0; k < o1_idx; k++) p[k] *= inv;
    norm /= t.s_tens[o1 % n];
    return norm;
}

double calc_u1_probs(const HBInfo &t, double *p, unsigned o1_orb, const uint8_t *occ, unsigned n_elec, int exclude_first) {
    unsigned n = t.n_orb;
    int sp = o1_orb / n;
    unsigned o1s = o1_orb % n, off = sp * n;
    double norm = 0;
    size_t pi = 0;
    unsigned oi = n_elec / 2 * sp;
    auto occ_at = [&](unsigned k) -> unsigned { return k < n_elec ? occ[k] : 255u; };  // see header note on row overrun
    unsigned cur = occ_at(oi);
    for (unsigned k = 0; k < o1s; k++) {
        if (k + off == cur) { oi++; cur = occ_at(oi); }
        else { p[pi] = t.exch_sqrt[tri_nodiag(k, o1s)]; norm += p[pi]; pi++; }
    }
    oi++;
    cur = occ_at(oi);
    for (unsigned k = o1s + 1; k < n; k++) {
        if (k + off == cur) {
            if (oi < n_elec - 1) { oi++; cur = occ_at(oi); }
        }
        else { p[pi] = t.exch_sqrt[tri_nodiag(o1s, k)]; norm += p[pi]; pi++; }
    }
    if (exclude_first) { norm -= p[0]; p[0] = 0; }
    double inv = 1. / norm;
    for (unsigned k = 0; k < pi; k++) p[k] *= inv;
    norm /= t.exch_norms[o1s];
    return norm;
}

static inline double exch_or_diag(const HBInfo &t, unsigned o, unsigned u) {
    if (o == u) return t.diag_sqrt[o];
    unsigned mn = o < u ? o : u, mx = o > u ? o : u;
    return t.exch_sqrt[tri_nodiag(mn, mx)];
}

double calc_u2_probs(const HBInfo &t, double *p, unsigned o1, unsigned o2, unsigned u1, const Symm &s, uint16_t *len) {
    unsigned n = t.n_orb;
    unsigned o2s = o2 % n, u1s = u1 % n;
    int same = (o1 / n) == (o2 / n);
    unsigned ir = s.irrep[o1 % n] ^ s.irrep[o2s] ^ s.irrep[u1s];
    unsigned num = s.lk(ir, 0);
    *len = (uint16_t)num;
    double norm = 0;
    for (unsigned k = 0; k < num; k++) {
        unsigned u2 = s.lk(ir, k + 1);
        if ((same && u2 != u1s) || !same) { p[k] = exch_or_diag(t, o2s, u2); norm += p[k]; }
        else p[k] = 0;
    }
    if (norm != 0) {
        double inv = 1 / norm;
        for (unsigned k = 0; k < num; k++) {
            unsigned u2 = s.lk(ir, k + 1);
            if ((same && u2 != u1s) || !same) p[k] *= inv;
        }
    }
    norm /= t.exch_norms[o2s];
    return norm;
}

double calc_u2_probs_half(const HBInfo &t, double *p, unsigned o1, unsigned o2, unsigned u1, det_t det, const Symm &s, uint16_t *len) {
    unsigned n = t.n_orb;
    unsigned o2s = o2 % n, u1s = u1 % n;
    int u2_spin = o2 / n;
    int same = (int)(o1 / n) == u2_spin;
    unsigned ir = s.irrep[o1 % n] ^ s.irrep[o2s] ^ s.irrep[u1s];
    unsigned num = s.lk(ir, 0);
    double norm = 0;
    unsigned k;
    for (k = 0; k < num; k++) {
        unsigned u2 = s.lk(ir, k + 1);
        if (same && u2 >= u1s) break;
        if (((same && u2 != u1s) || !same) && !bit(det, u2 + n * u2_spin)) { p[k] = exch_or_diag(t, o2s, u2); norm += p[k]; }
        else p[k] = 0;
    }
    *len = (uint16_t)k;
    if (norm != 0) {
        double inv = 1 / norm;
        for (k = 0; k < *len; k++) p[k] *= inv;
    }
    norm /= t.exch_norms[o2s];
    return norm;
}

double calc_unnorm_wt(const HBInfo &t, const uint8_t *orbs) {
    unsigned n = t.n_orb;
    unsigned o1 = orbs[0] % n, o2 = orbs[1] % n, u1 = orbs[2] % n, u2 = orbs[3] % n;
    unsigned mn11 = o1 < u1 ? o1 : u1, mx11 = o1 > u1 ? o1 : u1;
    unsigned mn22 = o2 < u2 ? o2 : u2, mx22 = o2 > u2 ? o2 : u2;
    int same = (orbs[0] / n) == (orbs[1] / n);
    double w;
    if (same) {
        w = t.d_same[tri_nodiag(o1, o2)] * (t.exch_sqrt[tri_nodiag(mn11, mx11)] * t.exch_sqrt[tri_nodiag(mn22, mx22)]) / t.s_norm / t.exch_norms[o1] / t.exch_norms[o2];
    }
    else {
        w = (t.d_diff[o2 * n + o1]) * t.exch_sqrt[tri_nodiag(mn11, mx11)] * t.exch_sqrt[tri_nodiag(mn22, mx22)] / t.s_norm / t.exch_norms[o1] / t.exch_norms[o2];
    }
    return w;
}

double calc_norm_wt(const HBInfo &t, const uint8_t *orbs, const uint8_t *occ, unsigned n_elec, det_t det, const Symm &sy) {
    unsigned n = t.n_orb;
    unsigned o1 = orbs[0] % n, o2 = orbs[1] % n, u1 = orbs[2] % n, u2 = orbs[3] % n;
    int o1_spin = orbs[0] / n, o2_spin = orbs[1] / n;
    unsigned mn11 = o1 < u1 ? o1 : u1, mx11 = o1 > u1 ? o1 : u1;
    unsigned mn22 = o2 < u2 ? o2 : u2, mx22 = o2 > u2 ? o2 : u2;
    int same = o1_spin == o2_spin;
    size_t k;
    uint8_t os[64];
    for (k = 0; k < n_elec; k++) os[k] = occ[k] % n;
    os[n_elec] = 255;
    double s_denom = 0;
    for (k = 0; k < n_elec; k++) s_denom += t.s_tens[os[k]];
    double d1 = 0;
    unsigned off = (1 - o1_spin) * n_elec / 2;
    for (k = off; k < n_elec / 2 + off; k++) d1 += t.d_diff[o1 * n + os[k]];
    off = o1_spin * n_elec / 2;
    for (k = off; os[k] < o1; k++) d1 += t.d_same[tri_nodiag(os[k], o1)];
    for (k++; k < n_elec / 2 + off; k++) d1 += t.d_same[tri_nodiag(o1, os[k])];
    double d2 = 0;
    off = (1 - o2_spin) * n_elec / 2;
    for (k = off; k < n_elec / 2 + off; k++) d2 += t.d_diff[o2 * n + os[k]];
    off = o2_spin * n_elec / 2;
    for (k = off; os[k] < o2; k++) d2 += t.d_same[tri_nodiag(os[k], o2)];
    for (k++; k < n_elec / 2 + off; k++) d2 += t.d_same[tri_nodiag(o2, os[k])];

    double e1v = 0;
    off = o1_spin * n;
    for (k = 0; k < o1; k++) if (!bit(det, k + off)) e1v += t.exch_sqrt[tri_nodiag(k, o1)];
    for (k = o1 + 1; k < n; k++) if (!bit(det, k + off)) e1v += t.exch_sqrt[tri_nodiag(o1, k)];
    double e2v = 0;
    off = o2_spin * n;
    for (k = 0; k < o2; k++) if (!bit(det, k + off)) e2v += t.exch_sqrt[tri_nodiag(k, o2)];
    for (k = o2 + 1; k < n; k++) if (!bit(det, k + off)) e2v += t.exch_sqrt[tri_nodiag(o2, k)];

    unsigned u1_ir = sy.irrep[u1], u2_ir = sy.irrep[u2];
    double e2s_no1 = 0, e2s_no2 = 0, e1s_no1 = 0, e1s_no2 = 0;
    for (k = 0; k < sy.lk(u2_ir, 0); k++) {
        unsigned so = sy.lk(u2_ir, k + 1);
        if ((same && so != u1) || !same) e2s_no1 += exch_or_diag(t, o2, so);
        if ((same && so != u1) || !same) e1s_no1 += exch_or_diag(t, o1, so);
    }
    for (k = 0; k < sy.lk(u1_ir, 0); k++) {
        unsigned so = sy.lk(u1_ir, k + 1);
        if ((same && so != u2) || !same) e2s_no2 += exch_or_diag(t, o2, so);
        if ((same && so != u2) || !same) e1s_no2 += exch_or_diag(t, o1, so);
    }
    unsigned o1u1 = tri_nodiag(mn11, mx11), o2u2 = tri_nodiag(mn22, mx22);
    double w;
    if (same) {
        unsigned mn12 = o1 < u2 ? o1 : u2, mx12 = o1 > u2 ? o1 : u2;
        unsigned mn21 = o2 < u1 ? o2 : u1, mx21 = o2 > u1 ? o2 : u1;
        unsigned o1o2 = tri_nodiag(o1, o2), o1u2 = tri_nodiag(mn12, mx12), o2u1 = tri_nodiag(mn21, mx21);
        w = t.d_same[o1o2] / s_denom * (
            t.s_tens[o1] / d1 / e1v * (t.exch_sqrt[o1u1] * t.exch_sqrt[o2u2] / e2s_no1 + t.exch_sqrt[o1u2] * t.exch_sqrt[o2u1] / e2s_no2) +
            t.s_tens[o2] / d2 / e2v * (t.exch_sqrt[o2u1] * t.exch_sqrt[o1u2] / e1s_no1 + t.exch_sqrt[o2u2] * t.exch_sqrt[o1u1] / e1s_no2));
    }
    else {
        w = (t.s_tens[o1] * t.d_diff[o1 * n + o2] / d1 / e1v / e2s_no1 + t.s_tens[o2] * t.d_diff[o2 * n + o1] / d2 / e2v / e1s_no2) * t.exch_sqrt[o1u1] * t.exch_sqrt[o2u2] / s_denom;
    }
    return w;
}

// ------------------------------------------------------------------ compression
double find_preserve(const double *values, std::vector<size_t> &srt, std::vector<uint8_t> &keep,
                     size_t count, unsigned *n_samp, double *global_norm, const Comm &cm) {
    double loc = 0, glob = 0;
    size_t heap_count = count;
    for (size_t i = 0; i < count; i++) { loc += fabs(values[i]); srt[i] = i; }
    auto cmp = [values](size_t i, size_t j) { return fabs(values[i]) < fabs(values[j]); };
    std::make_heap(srt.begin(), srt.begin() + heap_count, cmp);
    int loc_sampled, glob_sampled = 1, keep_going = 1;
    double el = 0;
    size_t mx;
    *global_norm = cm.sum(loc);
    bool recalc = false;
    while (glob_sampled > 0) {
        glob = cm.sum(loc);
        loc_sampled = 0;
        while (keep_going && heap_count > 0 && glob >= 0) {
            mx = srt[0];
            el = fabs(values[mx]);
            if (el >= glob / (*n_samp - loc_sampled)) {
                keep[mx] = 1;
                loc_sampled++;
                loc -= el;
                glob -= el;
                heap_count--;
                if (heap_count) std::pop_heap(srt.begin(), srt.begin() + heap_count + 1, cmp);
                else keep_going = 0;
            }
            else keep_going = 0;
        }
        glob_sampled = cm.sum(loc_sampled);
        (*n_samp) -= glob_sampled;
        if (glob_sampled == 0 && !recalc) {
            loc = 0;
            for (size_t i = 0; i < count; i++) if (!keep[i]) loc += fabs(values[i]);
            glob_sampled = 1;
            recalc = true;
        }
        else recalc = false;
        keep_going = 1;
    }
    loc = 0;
    if (glob < 1e-9) *n_samp = 0;
    else for (size_t i = 0; i < count; i++) if (!keep[i]) loc += fabs(values[i]);
    return loc;
}

double seed_sys(const double *norms, double *rn, unsigned n_samp, const Comm &cm) {
    double lbound = 0;
    for (int p = 0; p < cm.rank; p++) lbound += norms[p];
    double global_norm = lbound;
    for (int p = cm.rank; p < cm.size; p++) global_norm += norms[p];
    *rn *= global_norm / n_samp;
    *rn += global_norm / n_samp * (int)(lbound * n_samp / global_norm);
    if (*rn < lbound) *rn += global_norm / n_samp;
    return lbound;
}

void sys_comp(double *vals, size_t len, double *loc_norms, unsigned n_samp, std::vector<uint8_t> &keep, double rn, const Comm &cm) {
    double rn_sys = rn;    // every rank is handed rank 0's draw (MPI_Bcast at :291)
    double tmp_glob = 0;
    for (int p = 0; p < cm.size; p++) tmp_glob += loc_norms[p];
    double lbound;
    if (n_samp > 0) lbound = seed_sys(loc_norms, &rn_sys, n_samp, cm);
    else { lbound = 0; rn_sys = INFINITY; }
    double out_norm = 0;
    for (size_t i = 0; i < len; i++) {
        double v = vals[i];
        if (keep[i]) { out_norm += fabs(v); keep[i] = 0; }
        else if (v != 0) {
            lbound += fabs(v);
            if (rn_sys < lbound) {
                vals[i] = tmp_glob / n_samp * ((v > 0) - (v < 0));
                out_norm += tmp_glob / n_samp;
                rn_sys += tmp_glob / n_samp;
            }
            else { vals[i] = 0; keep[i] = 1; }
        }
    }
    std::vector<double> all(cm.size);
    cm.allgather(&out_norm, all.data(), sizeof(double));     // :326
    for (int p = 0; p < cm.size; p++) loc_norms[p] = all[p];
}

// ---------------------------------------------------------------- pivotal compression
// compress_utils.cpp:389-518.  The vector is cut into consecutive sampling units of weight seg_norm / n_samp; from the
// elements wholly inside a unit (plus the residual piece handed over by the previous unit) one candidate H is drawn, then
// either H or the element straddling the unit's upper border is sampled and the other is handed to the next unit.
void piv_samp_serial(double *vals, size_t len, double seg_norm, uint32_t n_samp, std::vector<uint8_t> &flag, std::mt19937 &mt) {
    if (n_samp == 0) {                                   // :391-403
        for (size_t i = 0; i < len; i++) {
            if (flag[i]) flag[i] = 0;
            else vals[i] = 0;
            if (vals[i] == 0) flag[i] = 1;
        }
        return;
    }
    const double unit = seg_norm / n_samp;
    std::vector<double> wt(16);          // wt[0]: residual piece; wt[1..]: the unit's unpreserved elements in order
    wt[0] = 0;
    size_t pos = 0, resid = 0;
    uint32_t n_done = 0;
    auto sgn_unit = [unit](double v) { return unit * ((v > 0) - (v < 0)); };
    while (pos < len && n_done < n_samp) {
        size_t n_wt = 1, used = 0;
        double cum = wt[0];
        for (; cum < unit && pos + used < len; used++) {               // :414-424
            if (!flag[pos + used]) {
                if (n_wt == wt.size()) wt.resize(2 * wt.size());
                wt[n_wt] = fabs(vals[pos + used]);
                cum += wt[n_wt];
                n_wt++;
            }
        }
        const bool at_end = pos + used == len;
        size_t n_inner = used > 0 ? used - 1 : 0;        // elements before the border element (:425-428)
        if (at_end) n_inner++;
        const double over = cum - unit;                   // b_n: the part of the border element beyond this unit
        if (!at_end) { n_wt--; cum -= wt[n_wt]; }        // :430-433
        const double under = unit - cum;                  // a_n: the part of the border element inside this unit
        double r = mt() / (1. + UINT32_MAX) * cum;      // :437-446
        double run = 0;
        size_t H = 0;
        while (run < r && H < n_wt) { run += wt[H]; H++; }
        if (r > 0) H--;
        if (H != 0 && pos != 0) { vals[resid] = 0; flag[resid] = 1; }       // the residual piece can no longer be drawn (:447-450)
        double p_pass = under / (unit - over);           // :453-456
        if (at_end) p_pass = 0;
        r = mt() / (1. + UINT32_MAX);
        if (r < p_pass) {                                 // border element sampled, H handed on (:458-476)
            size_t k = 1;
            for (size_t o = 0; o < n_inner; o++) {
                if (!flag[pos + o]) {
                    if (k == H) resid = pos + o;
                    else { vals[pos + o] = 0; flag[pos + o] = 1; }
                    k++;
                }
                else flag[pos + o] = 0;
            }
            vals[pos + n_inner] = sgn_unit(vals[pos + n_inner]);
        }
        else {                                            // H sampled, border element handed on (:477-501)
            if (H == 0) vals[resid] = sgn_unit(vals[resid]);
            size_t k = 1;
            for (size_t o = 0; o < n_inner; o++) {
                if (!flag[pos + o]) {
                    if (k != H) { vals[pos + o] = 0; flag[pos + o] = 1; }
                    else vals[pos + o] = sgn_unit(vals[pos + o]);
                    k++;
                }
                else flag[pos + o] = 0;
            }
            resid = pos + n_inner;
        }
        pos += n_inner + 1;
        wt[0] = over;
        n_done++;
    }
    for (; pos < len; pos++) {                            // :506-513
        if (!flag[pos]) { vals[pos] = 0; flag[pos] = 1; }
        else flag[pos] = 0;
    }
    if (resid < len) { vals[resid] = 0; flag[resid] = 1; }     // :515-518
}

// compress_utils.cpp:552-604
uint32_t piv_budget(const double *loc_norms, uint32_t n_samp, std::mt19937 &mt, const Comm &cm) {
    std::vector<uint32_t> budgets(cm.size, 0);
    if (cm.rank == 0) {
        double glob = 0;
        for (int p = 0; p < cm.size; p++) glob += loc_norms[p];
        uint32_t tot = 0, n_frac = 0;
        std::vector<double> frac(cm.size);
        for (int p = 0; p < cm.size; p++) {
            budgets[p] = loc_norms[p] / glob * n_samp;
            tot += budgets[p];
            frac[p] = loc_norms[p] - budgets[p] * glob / n_samp;
            if (frac[p] < 1e-12) frac[p] = 0;
            if (frac[p] > 0) n_frac++;
        }
        if (n_frac == n_samp - tot) {
            for (int p = 0; p < cm.size; p++) if (frac[p] > 0) budgets[p]++;
            tot = n_samp;
        }
        if (tot < n_samp) {
            std::vector<uint8_t> none(cm.size, 0);
            piv_samp_serial(frac.data(), cm.size, glob * (n_samp - tot) / n_samp, n_samp - tot, none, mt);
            for (int p = 0; p < cm.size; p++) if (frac[p] > 0) budgets[p]++;
        }
    }
    std::vector<uint32_t> all((size_t)cm.size * cm.size);      // MPI_Scatter from rank 0
    cm.allgather(budgets.data(), all.data(), sizeof(uint32_t) * cm.size);
    return all[cm.rank];
}

// compress_utils.cpp:606-681
double adjust_probs(double *vals, size_t len, uint32_t *n_samp_loc, double exp_nsamp_loc, uint32_t n_samp_tot, double tot_norm, std::vector<uint8_t> &flag) {
    const double top = ceill(exp_nsamp_loc);
    const double resid = exp_nsamp_loc - (unsigned int)exp_nsamp_loc;
    const double unit = tot_norm / n_samp_tot;
    const double loc_norm = exp_nsamp_loc * unit;
    bool too_big = false;
    for (size_t i = 0; i < len && !too_big; i++) if (!flag[i] && fabs(vals[i]) >= loc_norm / top) too_big = true;
    if (!too_big) return loc_norm;
    double counter = exp_nsamp_loc;
    if (*n_samp_loc > exp_nsamp_loc) {                   // the budget was rounded up: inflate small elements, pin large ones
        for (size_t i = 0; i < len; i++) {
            if (flag[i]) continue;
            int8_t sg = 2 * (vals[i] > 0) - 1;
            double pi = fabs(vals[i]) / unit;
            if (pi < resid) { counter += pi / resid - pi; vals[i] /= resid; }
            else { counter -= pi; vals[i] = sg * unit; flag[i] = 1; (*n_samp_loc)--; }
            if (counter >= *n_samp_loc) { vals[i] += sg * unit * (*n_samp_loc - counter); break; }
        }
    }
    else {                                                // rounded down: shrink
        for (size_t i = 0; i < len; i++) {
            if (flag[i]) continue;
            int8_t sg = 2 * (vals[i] > 0) - 1;
            double pi = fabs(vals[i]) / unit;
            if (pi > resid) { double q = (pi - resid) / (1 - resid); counter += q - pi; vals[i] = sg * q * unit; }
            else { counter -= pi; vals[i] = 0; }
            if (counter <= *n_samp_loc) { vals[i] += sg * unit * (*n_samp_loc - counter); break; }
        }
    }
    return *n_samp_loc * loc_norm / exp_nsamp_loc;
}

// compress_utils.cpp:354-386
void piv_comp_parallel(double *vals, size_t len, uint32_t compress_size, std::vector<size_t> &srt, std::vector<uint8_t> &flag, std::mt19937 &mt, const Comm &cm) {
    std::vector<double> norms(cm.size);
    unsigned n_samp = compress_size;
    double glob;
    double mine = find_preserve(vals, srt, flag, len, &n_samp, &glob, cm);
    cm.allgather(&mine, norms.data(), sizeof(double));
    glob = 0;
    for (int p = 0; p < cm.size; p++) glob += norms[p];
    uint32_t loc_samp = 0;
    double new_norm = 0;
    if (n_samp != 0) {
        loc_samp = piv_budget(norms.data(), n_samp, mt, cm);
        cm.sum((int)loc_samp);                            // the reference's consistency check (a collective)
        new_norm = adjust_probs(vals, len, &loc_samp, n_samp * norms[cm.rank] / glob, n_samp, glob, flag);
    }
    piv_samp_serial(vals, len, new_norm, loc_samp, flag, mt);
}

double find_keep_sub(const double *values, const uint32_t *n_div, SubWts &sw, const uint16_t *sub_sizes,
                     size_t count, unsigned *n_samp, double *wt_remain, const Comm &cm) {
    double loc = 0, glob = 0;
    for (size_t i = 0; i < count; i++) { loc += values[i]; wt_remain[i] = values[i]; }
    int loc_sampled, glob_sampled = 1;
    double sub_magn, sub_remain;
    int last_pass = 0;
    size_t n_sub = sw.cols;
    const size_t coarse = 8;
    size_t n_coarse = count / coarse;
    double cw[8];
    while (glob_sampled > 0) {
        glob = cm.sum(loc);
        if (glob < 0) break;
        loc_sampled = 0;
        for (size_t c = 0; c <= n_coarse; c++) {
            unsigned flags = 0;
            size_t lim = (c == n_coarse) ? count % coarse : coarse;
            double wf = *n_samp - loc_sampled;
            for (size_t f = 0; f < lim; f++) {
                size_t i = c * coarse + f;
                if (wt_remain[i] > 0) {
                    cw[f] = values[i] * wf;
                    if (n_div[i] > 0) cw[f] /= n_div[i];
                    flags += (unsigned)(cw[f] >= glob) << f;
                }
            }
            for (size_t f = 0; f < lim; f++) {
                if (!((flags >> f) & 1)) continue;
                size_t i = c * coarse + f;
                double el = values[i];
                if (n_div[i] > 0) {
                    sw.keep[i] |= 1u;
                    wt_remain[i] = 0;
                    loc_sampled += n_div[i];
                    loc -= el;
                    glob -= el;
                    if (glob < 0) break;
                }
                else {
                    sub_remain = 0;
                    const double *row = sw.row(i);
                    if (sub_sizes) n_sub = sub_sizes[i];
                    double cwt = cw[f];
                    size_t full = (n_sub / 8) * 8;
                    uint32_t kp = sw.keep[i];
                    for (size_t s = 0; s < n_sub; s++) {
                        if ((kp >> s) & 1u) continue;
                        sub_magn = cwt * row[s];
                        double thr = s < full ? 1e-12 : 1e-10;   // compress_utils.cpp:213 vs :233
                        if (sub_magn >= glob && fabs(sub_magn) > thr) { kp |= 1u << s; loc_sampled++; }
                        else sub_remain += sub_magn;
                    }
                    sw.keep[i] = kp;
                    sub_remain /= wf;
                    double change = wt_remain[i] - sub_remain;
                    wt_remain[i] = sub_remain;
                    loc -= change;
                    glob -= change;
                }
            }
        }
        glob_sampled = cm.sum(loc_sampled);
        (*n_samp) -= glob_sampled;
        if (last_pass && glob_sampled) last_pass = 0;
        if (glob_sampled == 0 && !last_pass) {
            last_pass = 1;
            glob_sampled = 1;
            loc = 0;
            for (size_t i = 0; i < count; i++) loc += wt_remain[i];
        }
    }
    loc = 0;
    if (glob / *n_samp < 1e-8) *n_samp = 0;
    else for (size_t i = 0; i < count; i++) loc += wt_remain[i];
    return loc;
}

size_t sys_sub(const double *values, const uint32_t *n_div, SubWts &sw, const uint16_t *sub_sizes,
               size_t count, unsigned n_samp, const double *wt_remain, double *loc_norms, double rn,
               double *new_vals, size_t (*new_idx)[2], const Comm &cm) {
    double rn_sys = rn;
    double tmp_glob = 0;
    for (int p = 0; p < cm.size; p++) tmp_glob += loc_norms[p];
    double lbound;
    if (n_samp > 0) lbound = seed_sys(loc_norms, &rn_sys, n_samp, cm);
    else { lbound = 0; rn_sys = INFINITY; }
    double out_norm = 0;
    size_t num_new = 0, sub_idx;
    size_t n_sub = sw.cols;
    for (size_t i = 0; i < count; i++) {
        double v = values[i];
        if (v == 0) continue;
        lbound += wt_remain[i];
        if (n_div[i] > 0) {
            if (sw.keep[i] & 1u) {
                sw.keep[i] &= ~1u;
                for (sub_idx = 0; sub_idx < n_div[i]; sub_idx++) {
                    new_vals[num_new] = v / n_div[i];
                    new_idx[num_new][0] = i; new_idx[num_new][1] = sub_idx;
                    num_new++;
                }
                out_norm += v;
            }
            else if (v != 0) {
                while (rn_sys < lbound) {
                    sub_idx = (size_t)((lbound - rn_sys) * n_div[i] / v);
                    if (sub_idx < n_div[i]) {
                        new_vals[num_new] = tmp_glob / n_samp;
                        new_idx[num_new][0] = i; new_idx[num_new][1] = sub_idx;
                        num_new++;
                        out_norm += tmp_glob / n_samp;
                    }
                    rn_sys += tmp_glob / n_samp;
                }
            }
        }
        else if (wt_remain[i] < v || rn_sys < lbound) {
            out_norm += (v - wt_remain[i]);
            double sub_lbound = lbound - wt_remain[i];
            if (sub_sizes) n_sub = sub_sizes[i];
            const double *row = sw.row(i);
            uint32_t kp = sw.keep[i];
            for (sub_idx = 0; sub_idx < n_sub; sub_idx++) {
                if (((kp >> sub_idx) & 1u) && row[sub_idx] != 0) {
                    new_vals[num_new] = v * row[sub_idx];
                    new_idx[num_new][0] = i; new_idx[num_new][1] = sub_idx;
                    num_new++;
                }
                else {
                    sub_lbound += v * row[sub_idx];
                    if (rn_sys < sub_lbound && row[sub_idx] != 0) {
                        new_vals[num_new] = tmp_glob / n_samp;
                        new_idx[num_new][0] = i; new_idx[num_new][1] = sub_idx;
                        num_new++;
                        out_norm += tmp_glob / n_samp;
                        rn_sys += tmp_glob / n_samp;
                    }
                }
            }
            sw.keep[i] = 0;   // reference clears bits 0..n_sub-1; no others are ever set
        }
    }
    loc_norms[cm.rank] = out_norm;
    return num_new;
}

size_t comp_sub(const double *values, size_t count, const uint32_t *n_div, SubWts &sw, const uint16_t *sub_sizes,
                unsigned n_samp, double *wt_remain, double rn, double *new_vals, size_t (*new_idx)[2], const Comm &cm) {
    unsigned tmp_nsamp = n_samp;
    std::vector<double> loc_norms(cm.size);
    double mine = find_keep_sub(values, n_div, sw, sub_sizes, count, &tmp_nsamp, wt_remain, cm);
    cm.allgather(&mine, loc_norms.data(), sizeof(double));   // :818
    static const bool dbg = getenv("FRIES_ORACLE_DBG") != nullptr;
    size_t n_out = sys_sub(values, n_div, sw, sub_sizes, count, tmp_nsamp, wt_remain, loc_norms.data(), rn, new_vals, new_idx, cm);
    if (dbg) fprintf(stderr, "[oracle comp_sub rank %d] n_in %zu n_rem %u loc_norm %.17g n_out %zu\n", cm.rank, count, tmp_nsamp, mine, n_out);
    return n_out;
}

void adjust_shift(double *shift, double one_norm, double *last_norm, double target_norm, double damp) {
    if (*last_norm) {
        *shift -= damp * log(one_norm / *last_norm);
        *last_norm = one_norm;
    }
    if (*last_norm == 0 && one_norm > target_norm) *last_norm = one_norm;
}

// ------------------------------------------------------------------ sparse vector
uint64_t hash_fxn(const uint8_t *occ, unsigned n_elec, const uint32_t *scr) {
    uint64_t hash = 0;
    for (unsigned i = 0; i < n_elec; i++) {
        // (i + 1) * scrambler_[...] is unsigned int * uint32_t: 32-bit wraparound
        uint32_t term = (uint32_t)(i + 1) * scr[occ[i]];
        hash = 1099511628211ULL * hash + term;
    }
    return hash;
}

void Vec::init(size_t size, size_t add_size, unsigned n_el, unsigned nv, const Comm &c, const uint32_t *pscr) {
    cm = c; proc_scr = pscr;
    n_elec = n_el; n_vecs = nv; max_size = size; curr_size = 0; adder_cap = add_size; n_nonz = 0; cur = 0;
    dets.assign(size, 0);
    vals.assign(nv, std::vector<double>(size, 0.0));
    occ.assign(size * n_el, 0);
    diag.assign(size, NAN);
    active.assign(size, 0);
    free_stack.clear(); table.clear();
    add_det.assign(cm.size, {}); add_val.assign(cm.size, {}); add_ini.assign(cm.size, {});
}

det_t Vec::key_of(det_t det) const {
    if (!hh_sites) return det;
    uint8_t o[64], ph[64];
    occ_list(det & elec_mask(), o);
    decode_phonons(det, hh_sites, hh_ph_bits, ph);
    uint64_t bucket = hash_fxn_hh(o, n_elec, ph, hh_sites, vec_scr) % (uint64_t)n_buckets;
    unsigned key_bytes = (2 * hh_sites + 7) / 8;
    det_t tmask = key_bytes >= 8 ? ~(det_t)0 : (((det_t)1 << (8 * key_bytes)) - 1);
    return (bucket << 24) | (det & tmask);
}

int Vec::idx_to_proc(det_t det) const {
    if (cm.size == 1) return 0;
    uint8_t o[64];
    unsigned n = (unsigned)occ_list(det & elec_mask(), o);
    if (hh_sites) {         // HubHolVec::idx_to_proc, hh_vec.hpp:56-66
        uint8_t ph[64];
        decode_phonons(det, hh_sites, hh_ph_bits, ph);
        return (int)(hash_fxn_hh(o, n_elec, ph, hh_sites, proc_scr) % (uint64_t)cm.size);
    }
    return (int)(hash_fxn(o, n, proc_scr) % (uint64_t)cm.size);
}

void Vec::expand() {
    size_t nm = max_size * 2;
    dets.resize(nm, 0);
    for (auto &c : vals) c.resize(nm, 0.0);
    occ.resize(nm * n_elec, 0);
    diag.resize(nm, NAN);
    active.resize(nm, 0);
    max_size = nm;
}

bool Vec::add(det_t det, double val, uint8_t ini) {
    if (val != 0) {
        int d = idx_to_proc(det);
        if (add_det[d].size() >= adder_cap) throw std::runtime_error("Too many elements added to Adder - must call perform_add() more frequently.");
        add_det[d].push_back(det); add_val[d].push_back(val); add_ini[d].push_back(ini);
        return add_det[d].size() < adder_cap;
    }
    return true;
}

void Vec::perform_add(size_t origin) {
    // Alltoallv of (index, value, initiator) triples; the receive buffer is ordered by source rank and,
    // within one source, by the order of its add() calls (vec_utils.hpp:991-1019).
    std::vector<det_t> rdet; std::vector<double> rval; std::vector<uint8_t> rini;
    if (cm.size == 1) { rdet.swap(add_det[0]); rval.swap(add_val[0]); rini.swap(add_ini[0]); }
    else {
        std::vector<std::vector<uint8_t>> snd(cm.size), rcv;
        for (int d = 0; d < cm.size; d++) {
            size_t n = add_det[d].size();
            snd[d].resize(n * 17);
            if (n) {
                memcpy(snd[d].data(), add_det[d].data(), n * 8);
                memcpy(snd[d].data() + n * 8, add_val[d].data(), n * 8);
                memcpy(snd[d].data() + n * 16, add_ini[d].data(), n);
            }
        }
        cm.alltoallv(snd, rcv);
        for (int sr = 0; sr < cm.size; sr++) {
            size_t n = rcv[sr].size() / 17, o = rdet.size();
            rdet.resize(o + n); rval.resize(o + n); rini.resize(o + n);
            if (n) {
                memcpy(&rdet[o], rcv[sr].data(), n * 8);
                memcpy(&rval[o], rcv[sr].data() + n * 8, n * 8);
                memcpy(&rini[o], rcv[sr].data() + n * 16, n);
            }
        }
    }
    uint8_t tmp_occ[64];
    for (size_t e = 0; e < rdet.size(); e++) {
        det_t d = rdet[e];
        int ini = rini[e];
        if ((unsigned)occ_list(d & elec_mask(), tmp_occ) != n_elec) throw std::runtime_error("Determinant created with an incorrect number of electrons");
        ptrdiff_t *ptr = nullptr;
        const det_t key = key_of(d);
        auto it = table.find(key);
        if (it != table.end()) ptr = &it->second;
        else if (ini) ptr = &table.emplace(key, (ptrdiff_t)-1).first->second;
        if (ptr && *ptr == -1) {
            if (!free_stack.empty()) { *ptr = (ptrdiff_t)free_stack.back(); free_stack.pop_back(); }
            else {
                if (curr_size >= max_size) expand();
                *ptr = (ptrdiff_t)curr_size;
                curr_size++;
            }
            size_t pos = (size_t)*ptr;
            dets[pos] = d;
            for (unsigned v = 0; v < n_vecs; v++) vals[v][pos] = 0;
            diag[pos] = NAN;
            active[pos] = 1;
            memcpy(&occ[pos * n_elec], tmp_occ, n_elec);
            n_nonz++;
        }
        if (ptr) {
            size_t pos = (size_t)*ptr;
            bool nonz = vals[origin][pos] != 0;
            bool should = ini || nonz;
            nonini_occ_add += !ini && nonz;
            if (should) vals[cur][pos] += rval[e];
        }
    }
    for (int d = 0; d < cm.size; d++) { add_det[d].clear(); add_val[d].clear(); add_ini[d].clear(); }
}

void Vec::del_at_pos(size_t pos) {
    if (!active[pos]) return;
    bool all_zero = true;
    for (unsigned v = 0; v < n_vecs; v++) if (vals[v][pos] != 0) all_zero = false;
    if (all_zero) {
        free_stack.push_back(pos);
        table.erase(key_of(dets[pos]));
        n_nonz--;
        active[pos] = 0;
    }
}

double Vec::local_norm() const {
    double norm = 0;
    for (size_t i = 0; i < curr_size; i++) norm += fabs(vals[cur][i]);
    return norm;
}

double Vec::dot(const std::vector<det_t> &d2, const std::vector<double> &v2) const {
    double numer = 0;
    for (size_t k = 0; k < d2.size(); k++) {
        auto it = table.find(key_of(d2[k]));
        if (it != table.end()) numer += v2[k] * vals[cur][(size_t)it->second];
    }
    return numer;
}

// ------------------------------------------------------------------ deterministic H application, frifull_mol
det_t flip_spins(det_t det, unsigned n_orb) {
    const det_t half = n_orb >= 64 ? ~0ull : (1ull << n_orb) - 1ull;
    det_t out = n_orb >= 32 ? (det >> 32) | (det << 32) : ((det >> n_orb) & half) | ((det & half) << n_orb);
    // The reference's byte loop (fci_utils.c:173-178) is off by one byte when the strings are whole bytes long and at least three of them
    // (n_orb = 24, 32): bytes mid + 1 .. n_bytes - 2 of the result receive alpha byte b - mid - 1 instead of b - mid.  Kept: a vector
    // symmetrised by the reference is symmetrised under THIS map.
    if (n_orb % 8 == 0 && n_orb >= 24) {
        const unsigned mid = n_orb / 8, nb = 2 * mid;
        for (unsigned b = mid + 1; b + 1 < nb; b++) out = (out & ~(0xffull << (8 * b))) | (((det >> (8 * (b - mid - 1))) & 0xffull) << (8 * b));
    }
    return out;
}
int det_memcmp(det_t a, det_t b) {
    const det_t x = a ^ b;
    if (!x) return 0;
    const int byte = __builtin_ctzll(x) >> 3;
    return ((a >> (8 * byte)) & 255ull) > ((b >> (8 * byte)) & 255ull) ? 1 : -1;
}
int tr_doub_connect(const uint8_t *occ, unsigned n_orb, unsigned n_elec, uint8_t *diff_idx) {
    const unsigned half = n_elec / 2;
    bool same = true;
    for (unsigned k = 0; k < half && same; k++) same = occ[k] == occ[half + k] - n_orb;
    if (same) return 0;
    unsigned i1 = 0, i2 = 0;
    bool s1 = false, s2 = false;
    while (i1 < half && i2 < half) {
        const int diff = (int)occ[i1] - (int)(occ[half + i2] - n_orb);
        if (diff == 0) { i1++; i2++; }
        else if (diff > 0) { if (s2) return 2; s2 = true; diff_idx[1] = (uint8_t)(half + i2); i2++; }
        else { if (s1) return 2; s1 = true; diff_idx[0] = (uint8_t)i1; i1++; }
    }
    if (i1 < half) diff_idx[0] = (uint8_t)i1;
    else if (i2 < half) diff_idx[1] = (uint8_t)(half + i2);
    return 1;
}
int adjust_tr(const MolSys &sys, det_t cur, det_t nd, const uint8_t *occ, double *matr_el, int spin_parity, det_t *target, bool unit_matrel,
              const std::function<void(int, const uint8_t *)> &weight_fix) {
    const unsigned n = sys.n_orb;
    const uint8_t *irr = sys.symm.irrep.data();
    double norm = flip_spins(cur, n) == cur ? sqrt(2) : 1;               // i == i'
    const det_t img = flip_spins(nd, n);
    if (img == cur) { *matr_el = 0; return 0; }                            // (part of the diagonal)
    const int cmp = det_memcmp(nd, img);
    if (cmp == 0) {                                                         // j == j'
        if (spin_parity == -1) { *matr_el = 0; return 0; }
        *matr_el *= 2;
        norm *= sqrt(2);
    }
    else {
        const det_t x = cur ^ img;
        const int n_diff = __builtin_popcountll(x);
        uint8_t d[4] = {0, 0, 0, 0};
        if (n_diff <= 4) { int k = 0; for (det_t y = x; y; y &= y - 1) d[k++] = (uint8_t)__builtin_ctzll(y); }
        if (n_diff == 2) {
            if (irr[d[0] % n] == irr[d[1] % n]) {
                if (bit(cur, d[1])) std::swap(d[0], d[1]);
                if (weight_fix) weight_fix(1, d);
                double rev = unit_matrel ? 1.0 : sing_matrel_nosgn(d, occ, sys.ints, sys.n_elec);
                rev *= sing_parity(cur, d);
                *matr_el += rev * spin_parity;
                if (!weight_fix) norm *= 2;             // h_op_offdiag: two excitations give this determinant (molecule.cpp:326); apply_HBPP_piv adds their probabilities instead
            }
        }
        else if (n_diff == 4) {
            // the reference writes `a ^ b ^ c ^ d == 0`, which C++ reads as a ^ b ^ c ^ (d == 0): kept as written
            if ((irr[d[0] % n] ^ irr[d[1] % n] ^ irr[d[2] % n] ^ (unsigned)(irr[d[3] % n] == 0)) != 0) {
                if (bit(cur, d[2])) { if (bit(cur, d[0])) std::swap(d[1], d[2]); else std::swap(d[0], d[2]); }
                if (bit(cur, d[3])) { if (bit(cur, d[0])) std::swap(d[1], d[3]); else std::swap(d[0], d[3]); }
                if (d[0] > d[1]) std::swap(d[0], d[1]);
                if (d[2] > d[3]) std::swap(d[2], d[3]);
                if (weight_fix) weight_fix(2, d);
                double rev = unit_matrel ? 1.0 : doub_matrel_nosgn(d, sys.ints);
                rev *= doub_parity(cur, d);
                *matr_el += rev * spin_parity;
                if (!weight_fix) norm *= 2;
            }
        }
    }
    if (cmp > 0) norm *= spin_parity;
    *matr_el /= norm;
    *target = cmp > 0 ? img : nd;
    return 1;
}

size_t h_op_offdiag(Vec &v, size_t vec_size, const MolSys &sys, unsigned dest, double h_fac, int spin_parity) {
    const unsigned n_elec = sys.n_elec, n_orb = sys.n_orb;
    const unsigned origin = v.cur;
    std::vector<uint8_t> ex;
    size_t n_calls = 0;
    for (int kind = 0; kind < 2; kind++) {            // all singles (:562-608), then all doubles (:610-664)
        int keep_going = 1;
        size_t ex_idx = 0, n_ex = 0, det_idx = 0;
        double cur_el = 0;
        det_t cur_det = 0;
        const uint8_t *occ = nullptr;
        while (keep_going) {
            keep_going = 0;
            const double *before = v.vals[origin].data();
            v.cur = dest;
            while (true) {
                if (ex_idx >= n_ex) {
                    if (det_idx >= vec_size) break;
                    cur_el = before[det_idx];
                    if (cur_el == 0) { det_idx++; continue; }
                    cur_det = v.dets[det_idx];
                    occ = v.orbs_at(det_idx);
                    n_ex = kind == 0 ? sing_ex_symm(cur_det, occ, n_elec, n_orb, ex, sys.symm.irrep.data())
                                     : doub_ex_symm(cur_det, occ, n_elec, n_orb, ex, sys.symm.irrep.data());
                    if (n_ex == 0) { det_idx++; continue; }     // (the reference has this guard for singles only; a molecule always has doubles)
                    ex_idx = 0;
                    det_idx++;
                }
                det_t nd = cur_det;
                double m;
                if (kind == 0) { m = sing_matrel_nosgn(&ex[2 * ex_idx], occ, sys.ints, n_elec); m *= sing_det_parity(&nd, &ex[2 * ex_idx]); }
                else { m = doub_matrel_nosgn(&ex[4 * ex_idx], sys.ints); m *= doub_det_parity(&nd, &ex[4 * ex_idx]); }
                ex_idx++;
                keep_going = 1;
                if (spin_parity) { det_t tgt = nd; if (!adjust_tr(sys, cur_det, nd, occ, &m, spin_parity, &tgt)) continue; nd = tgt; }
                m *= cur_el * h_fac;
                n_calls++;
                if (!v.add(nd, m, 1)) break;
            }
            keep_going = v.cm.sum(keep_going);
            v.perform_add(0);
        }
    }
    return n_calls;
}

void h_op_diag(Vec &v, unsigned dest, double id_fac, double h_fac, const MolSys &sys) {
    const std::vector<double> &src = v.vals[v.cur];
    for (size_t i = 0; i < v.curr_size; i++) {
        double cv = src[i];
        if (cv != 0) {
            if (v.diag[i] != v.diag[i]) v.diag[i] = diag_matrel(v.orbs_at(i), sys.ints, sys.n_elec) - sys.hf_en;      // matr_el_at_pos, vec_utils.hpp:548-556
            v.vals[dest][i] = cv * (id_fac + h_fac * v.diag[i]);
        }
        else v.vals[dest][i] = 0;
    }
    v.cur = dest;
}

void Frifull::setup() {
    const unsigned n_orb = sys.n_orb, n_elec = sys.n_elec;
    uint8_t tmp[64];
    hf_det = gen_hf_det(n_orb, n_elec);
    occ_list(hf_det, tmp);
    sys.hf_en = diag_matrel(tmp, sys.ints, n_elec);
    mt.seed(par.seed);
    proc_scr.resize(2 * n_orb); vec_scr.resize(2 * n_orb);
    for (auto &x : proc_scr) x = mt();     // frifull_mol.cpp:96-98
    for (auto &x : vec_scr) x = mt();      // :104-107
    sol.init(par.max_dets, 1000000, n_elec, 2, Comm::self(), proc_scr.data());     // the Adder's size only sets how often perform_add runs
    trial_det = {hf_det}; trial_val = {1.0};
    sol.add(hf_det, 100, 1);               // :186-190
    sol.perform_add(0);
    srt.resize(sol.max_size); keep.assign(sol.max_size, 0);
    en_shift = 0; last_one_norm = 0; iterat = 0; vec_idx = 0;
}

void Frifull::iterate(unsigned n) {
    const double eps = par.eps;
    const double shift_damping = 0.05;
    const unsigned shift_interval = 10;
    for (unsigned k = 0; k < n; k++, iterat++) {
        IterLog lg{};
        sol.cur = vec_idx;
        double denom = sol.dot(trial_det, trial_val);           // :259-260
        if (srt.size() < sol.max_size) { srt.resize(sol.max_size); keep.resize(sol.max_size, 0); }
        unsigned n_samp = par.vec_nonz;
        double glob_norm;
        double mine = find_preserve(sol.vals[vec_idx].data(), srt, keep, sol.curr_size, &n_samp, &glob_norm);
        lg.nkept = par.vec_nonz - n_samp;
        if ((iterat + 1) % shift_interval == 0) adjust_shift(&en_shift, glob_norm, &last_one_norm, par.target_norm, shift_damping / shift_interval / eps);
        double rn_sys = mt() / (1. + UINT32_MAX);
        sys_comp(sol.vals[vec_idx].data(), sol.curr_size, &mine, n_samp, keep, rn_sys);
        for (size_t i = 0; i < sol.curr_size; i++) if (keep[i]) { sol.del_at_pos(i); keep[i] = 0; }
        h_op_diag(sol, !vec_idx, 1 + eps * en_shift, -eps, sys);      // :288
        sol.cur = vec_idx;
        lg.num_success = h_op_offdiag(sol, sol.curr_size, sys, !vec_idx, -eps);
        vec_idx = !vec_idx;
        sol.cur = vec_idx;
        double numer = sol.dot(trial_det, trial_val);
        numer = ((1 + eps * en_shift) * denom - numer) / eps;       // :295
        lg.numer = numer; lg.denom = denom; lg.shift = en_shift; lg.norm = glob_norm;
        lg.n_nonz = sol.n_nonz; lg.curr_size = sol.curr_size;
        log.push_back(lg);
    }
}

// ------------------------------------------------------------------ apply_HBPP_sys
void HBScratch::init(size_t length, size_t n_subwt) {
    len = length; vec_len = 0;
    vec1.assign(length, 0); vec2.assign(length, 0); wt_remain.assign(length, 0);
    det_idx1.assign(length, 0); det_idx2.assign(length, 0);
    orb1.assign(length * 4, 0); orb2.assign(length * 4, 0);
    nsub.assign(length, 0); ndiv.assign(length, 0);
    comp_idx.assign(length * 2, 0);
    sw.cols = n_subwt; sw.w.assign(length * n_subwt, 0); sw.keep.assign(length, 0);
}

void apply_HBPP_sys(const Vec &v, HBScratch &sc, const MolSys &sys, double p_doub, bool new_hb,
                    const double rn[5], uint32_t n_samp, bool unit_matrel, const Comm &cm) {
    std::vector<double> &vec1 = sc.vec1, &vec2 = sc.vec2;
    SubWts &sw = sc.sw;
    std::vector<uint32_t> &ndiv = sc.ndiv;
    std::vector<uint16_t> &nsub = sc.nsub;
    size_t comp_len = sc.vec_len;
    std::vector<size_t> &di1 = sc.det_idx1, &di2 = sc.det_idx2;
    uint8_t (*oi1)[4] = (uint8_t (*)[4])sc.orb1.data();
    uint8_t (*oi2)[4] = (uint8_t (*)[4])sc.orb2.data();
    size_t (*cidx)[2] = (size_t (*)[2])sc.comp_idx.data();
    double *wtr = sc.wt_remain.data();
    size_t spawn_length = sc.len;
    const unsigned n_elec = sys.n_elec, n_orb = sys.n_orb;
    const HBInfo &hb = sys.hb;
    const Symm &symm = sys.symm;
    unsigned cts[N_IRREPS][2];

    // ---- singles vs doubles (heat_bathPP.cpp:713-734)
    sw.reshape(spawn_length, 2);
    for (size_t d = 0; d < comp_len; d++) {
        double w = fabs(vec1[d]);
        vec1[d] = w;
        if (w > 0) { sw.row(d)[0] = p_doub; sw.row(d)[1] = 1 - p_doub; ndiv[d] = 0; }
        else ndiv[d] = 1;
    }
    comp_len = comp_sub(vec1.data(), comp_len, ndiv.data(), sw, nullptr, n_samp, wtr, rn[0], vec2.data(), cidx, cm);
    sc.stage_len[0] = comp_len;
    if (comp_len > spawn_length) std::cerr << "Error: insufficient memory allocated for matrix compression.\n";

    // ---- first occupied orbital (:736-770)
    sw.reshape(spawn_length, n_elec - new_hb);
    for (size_t s = 0; s < comp_len; s++) {
        size_t d = di1[cidx[s][0]];
        di2[s] = d;
        oi1[s][0] = (uint8_t)cidx[s][1];
        const uint8_t *occ = v.orbs_at(d);
        if (oi1[s][0] == 0) {
            ndiv[s] = 0;
            double tw = calc_o1_probs(hb, sw.row(s), n_elec, occ, new_hb);
            if (new_hb) vec2[s] *= tw;
        }
        else {
            count_symm_virt(cts, occ, n_elec, symm);
            uint32_t n_occ = count_sing_allowed(occ, n_elec, symm, cts);
            if (n_occ == 0) { ndiv[s] = 1; vec2[s] = 0; }
            else ndiv[s] = n_occ;
        }
    }
    comp_len = comp_sub(vec2.data(), comp_len, ndiv.data(), sw, nullptr, n_samp, wtr, rn[1], vec1.data(), cidx, cm);
    sc.stage_len[1] = comp_len;
    if (comp_len > spawn_length) std::cerr << "Error: insufficient memory allocated for matrix compression.\n";

    // ---- unoccupied (single) / 2nd occupied (double) (:772-816)
    for (size_t s = 0; s < comp_len; s++) {
        size_t wi = cidx[s][0];
        size_t d = di2[wi];
        di1[s] = d;
        oi2[s][0] = oi1[wi][0];
        oi2[s][1] = (uint8_t)cidx[s][1];
        if (oi2[s][1] >= n_elec) {
            std::cerr << "Error: chosen occupied orbital (first) is out of bounds\n";
            vec1[s] = 0; ndiv[s] = 1;
            continue;
        }
        const uint8_t *occ = v.orbs_at(d);
        if (oi2[s][0] == 0) {
            ndiv[s] = 0;
            if (new_hb) {
                oi2[s][1]++;
                nsub[s] = oi2[s][1];
                vec1[s] *= calc_o2_probs_half(hb, sw.row(s), n_elec, occ, oi2[s][1]);
            }
            else calc_o2_probs(hb, sw.row(s), n_elec, occ, oi2[s][1]);
        }
        else {
            count_symm_virt(cts, occ, n_elec, symm);
            uint32_t n_virt = count_sing_virt(occ, n_elec, symm, cts, &oi2[s][1]);
            if (n_virt == 0) { ndiv[s] = 1; vec1[s] = 0; }
            else { ndiv[s] = n_virt; oi2[s][3] = (uint8_t)n_virt; }
        }
    }
    comp_len = comp_sub(vec1.data(), comp_len, ndiv.data(), sw, new_hb ? nsub.data() : nullptr, n_samp, wtr, rn[2], vec2.data(), cidx, cm);
    sc.stage_len[2] = comp_len;
    if (comp_len > spawn_length) std::cerr << "Error: insufficient memory allocated for matrix compression.\n";

    // ---- 1st unoccupied (double) (:818-864)
    sw.reshape(spawn_length, n_orb - n_elec / 2);
    for (size_t s = 0; s < comp_len; s++) {
        size_t wi = cidx[s][0];
        size_t d = di1[wi];
        di2[s] = d;
        oi1[s][0] = oi2[wi][0];
        uint8_t o1_idx = oi2[wi][1];
        oi1[s][1] = o1_idx;
        uint8_t o2u1 = (uint8_t)cidx[s][1];
        oi1[s][2] = o2u1;
        if (oi1[s][0] == 0) {
            if (o2u1 >= n_elec) {
                std::cerr << "Error: chosen occupied orbital (second) is out of bounds\n";
                vec2[s] = 0; ndiv[s] = 1;
                continue;
            }
            ndiv[s] = 0;
            const uint8_t *occ = v.orbs_at(d);
            int o1_spin = o1_idx / (n_elec / 2);
            int o2_spin = occ[o2u1] / n_orb;
            uint8_t o1_orb = occ[o1_idx];
            double tw = calc_u1_probs(hb, sw.row(s), o1_orb, occ, n_elec, new_hb && (o1_spin == o2_spin));
            if (new_hb) vec2[s] *= tw;
        }
        else {
            uint8_t n_virt = oi2[wi][3];
            if (o2u1 >= n_virt) { vec1[s] = 0; std::cerr << "Error: index of chosen virtual orbital exceeds maximum\n"; }
            oi1[s][3] = n_virt;
            ndiv[s] = 1;
        }
    }
    comp_len = comp_sub(vec2.data(), comp_len, ndiv.data(), sw, nullptr, n_samp, wtr, rn[3], vec1.data(), cidx, cm);
    sc.stage_len[3] = comp_len;
    if (comp_len > spawn_length) std::cerr << "Error: insufficient memory allocated for matrix compression.\n";

    // ---- 2nd unoccupied (double) (:866-915)
    sw.reshape(spawn_length, symm.max_n_symm);
    for (size_t s = 0; s < comp_len; s++) {
        size_t wi = cidx[s][0];
        size_t d = di2[wi];
        di1[s] = d;
        oi2[s][0] = oi1[wi][0];
        uint8_t o1_idx = oi1[wi][1];
        oi2[s][1] = o1_idx;
        uint8_t o2_idx = oi1[wi][2];
        oi2[s][2] = o2_idx;
        if (oi2[s][0] == 0) {
            const uint8_t *occ = v.orbs_at(d);
            uint8_t u1 = find_nth_virt(occ, o1_idx / (n_elec / 2), n_elec, n_orb, (unsigned)cidx[s][1]);
            det_t cd = v.dets[d];
            if (bit(cd, u1)) {
                std::cerr << "Error: occupied orbital chosen as 1st virtual\n";
                vec1[s] = 0; ndiv[s] = 1;
            }
            else {
                ndiv[s] = 0;
                oi2[s][3] = u1;
                double tw;
                uint8_t o1_orb = occ[o1_idx], o2_orb = occ[o2_idx];
                if (new_hb) tw = calc_u2_probs_half(hb, sw.row(s), o1_orb, o2_orb, u1, cd, symm, &nsub[s]);
                else tw = calc_u2_probs(hb, sw.row(s), o1_orb, o2_orb, u1, symm, &nsub[s]);
                if (new_hb || tw == 0) vec1[s] *= tw;
            }
        }
        else { oi2[s][3] = oi1[wi][3]; ndiv[s] = 1; }
    }
    comp_len = comp_sub(vec1.data(), comp_len, ndiv.data(), sw, nsub.data(), n_samp, wtr, rn[4], vec2.data(), cidx, cm);
    sc.stage_len[4] = comp_len;
    if (comp_len > spawn_length) std::cerr << "Error: insufficient memory allocated for matrix compression.\n";

    // ---- decode, weight, matrix element, parity (:917-991)
    size_t ns = 0;
    for (size_t s = 0; s < comp_len; s++) {
        size_t wi = cidx[s][0];
        size_t d = di1[wi];
        di2[ns] = d;
        const uint8_t *occ = v.orbs_at(d);
        det_t cd = v.dets[d];
        uint8_t o1_idx = oi2[wi][1];
        if (oi2[wi][0] == 0) {
            uint8_t o2_idx = oi2[wi][2];
            uint8_t o1 = occ[o1_idx], o2 = occ[o2_idx], u1 = oi2[wi][3];
            uint8_t u2_ir = symm.irrep[o1 % n_orb] ^ symm.irrep[o2 % n_orb] ^ symm.irrep[u1 % n_orb];
            uint8_t u2 = symm.lk(u2_ir, (unsigned)cidx[s][1] + 1) + n_orb * (o2 / n_orb);
            if (bit(cd, u2)) {
                if (new_hb) std::cerr << "Error: occupied orbital chosen as second virtual in unnormalized heat-bath\n";
                continue;
            }
            if (u1 == u2) { std::cerr << "Error: repeat virtual orbital chosen\n"; continue; }
            if (u1 > u2) std::swap(u1, u2);
            if (o1 > o2) std::swap(o1, o2);
            oi1[ns][0] = o1; oi1[ns][1] = o2; oi1[ns][2] = u1; oi1[ns][3] = u2;
            double tw = new_hb ? calc_unnorm_wt(hb, oi1[ns]) : calc_norm_wt(hb, oi1[ns], occ, n_elec, cd, symm);
            double mel = unit_matrel ? 1.0 : doub_matrel_nosgn(oi1[ns], sys.ints);
            double el = mel * vec2[s] / tw / p_doub;
            if (fabs(el) > 1e-9) {
                el *= doub_parity(cd, oi1[ns]);
                vec1[ns] = el;
                ns++;
            }
        }
        else {
            uint8_t o1 = occ[o1_idx];
            oi1[ns][0] = o1;
            uint8_t u1_ir = symm.irrep[o1 % n_orb];
            uint8_t spin = o1 / n_orb;
            uint8_t u1 = virt_from_idx(cd, symm, u1_ir, n_orb * spin, oi2[wi][2]);
            if (u1 == 255) { std::cerr << "Error: virtual orbital not found\n"; continue; }
            oi1[ns][1] = u1;
            oi1[ns][2] = oi1[ns][3] = 0;
            count_symm_virt(cts, occ, n_elec, symm);
            unsigned n_occ = count_sing_allowed(occ, n_elec, symm, cts);
            double el = unit_matrel ? 1.0 : sing_matrel_nosgn(oi1[ns], occ, sys.ints, n_elec);
            el *= vec2[s] / (1 - p_doub) * n_occ * oi2[wi][3];
            if (fabs(el) > 1e-9) {
                el *= sing_parity(cd, oi1[ns]);
                vec1[ns] = el;
                ns++;
            }
        }
    }
    sc.vec_len = ns;
}

// ------------------------------------------------------------------ apply_HBPP_piv
void HBPivScratch::init(size_t length, size_t n_subwt) {
    len = length; vec_len = 0;
    vec1.assign(length, 0); long_vec.assign(length * n_subwt, 0);
    det_idx1.assign(length, 0); det_idx2.assign(length, 0); srt.assign(length * n_subwt, 0);
    orb1.assign(length * 4, 0); orb2.assign(length * 4, 0); flag.assign(length * n_subwt, 0);
    group.assign(length, 0);
}

// heat_bathPP.cpp:994-1012: the elements the compression did not zero, in order; fn(old_short, new_short, index in the group)
template <class F>
static size_t collapse_long(std::vector<double> &short_vec, const std::vector<double> &long_vec, size_t short_len, std::vector<uint8_t> &flag,
                            const std::vector<uint16_t> &group, F fn) {
    size_t n_short = 0, li = 0;
    for (size_t si = 0; si < short_len; si++) {
        for (size_t g = 0; g < group[si]; g++) {
            if (!flag[li]) { short_vec[n_short] = long_vec[li]; fn(si, n_short, g); n_short++; }
            flag[li] = 0;
            li++;
        }
    }
    return n_short;
}

void apply_HBPP_piv(const Vec &v, HBPivScratch &sc, const MolSys &sys, double p_doub, bool new_hb,
                    std::mt19937 &mt, uint32_t n_samp, bool unit_matrel, const Comm &cm, int spin_parity) {
    if (spin_parity && !new_hb) throw std::runtime_error("Time-reversal symmetry is only implemented for the unnormalized heat-bath distribution");      // :1019-1021
    std::vector<double> &sv = sc.vec1, &lv = sc.long_vec;
    size_t n_short = sc.vec_len;
    std::vector<size_t> &di1 = sc.det_idx1, &di2 = sc.det_idx2;
    uint8_t (*oi1)[4] = (uint8_t (*)[4])sc.orb1.data();
    uint8_t (*oi2)[4] = (uint8_t (*)[4])sc.orb2.data();
    std::vector<uint16_t> &grp = sc.group;
    const unsigned n_elec = sys.n_elec, n_orb = sys.n_orb;
    const HBInfo &hb = sys.hb;
    const Symm &symm = sys.symm;
    unsigned cts[N_IRREPS][2];

    // ---- singles vs doubles (:1047-1067)
    size_t n_long = 0;
    for (size_t s = 0; s < n_short; s++) {
        double w = fabs(sv[s]);
        if (w > 0) { lv[n_long++] = w * p_doub; lv[n_long++] = w * (1 - p_doub); grp[s] = 2; }
        else grp[s] = 0;
    }
    piv_comp_parallel(lv.data(), n_long, n_samp, sc.srt, sc.flag, mt, cm);
    n_short = collapse_long(sv, lv, n_short, sc.flag, grp, [&](size_t o, size_t n, size_t g) { di2[n] = di1[o]; oi1[n][0] = (uint8_t)g; });
    sc.stage_len[0] = n_short;

    // ---- first occupied orbital (:1069-1101)
    n_long = 0;
    for (size_t s = 0; s < n_short; s++) {
        const uint8_t *occ = v.orbs_at(di2[s]);
        if (oi1[s][0] == 0) {
            grp[s] = (uint16_t)(n_elec - new_hb);
            double tw = calc_o1_probs(hb, &lv[n_long], n_elec, occ, new_hb);
            for (size_t k = 0; k < n_elec - new_hb; k++) lv[n_long + k] *= sv[s] * (new_hb ? tw : 1);
            n_long += n_elec - new_hb;
        }
        else {
            count_symm_virt(cts, occ, n_elec, symm);
            uint32_t n_occ = count_sing_allowed(occ, n_elec, symm, cts);
            grp[s] = (uint16_t)n_occ;
            for (size_t k = 0; k < n_occ; k++) lv[n_long + k] = sv[s] / n_occ;
            n_long += n_occ;
        }
    }
    piv_comp_parallel(lv.data(), n_long, n_samp, sc.srt, sc.flag, mt, cm);
    n_short = collapse_long(sv, lv, n_short, sc.flag, grp, [&](size_t o, size_t n, size_t g) { di1[n] = di2[o]; oi2[n][0] = oi1[o][0]; oi2[n][1] = (uint8_t)g; });
    sc.stage_len[1] = n_short;

    // ---- unoccupied (single) / 2nd occupied (double) (:1103-1157)
    n_long = 0;
    for (size_t s = 0; s < n_short; s++) {
        if (oi2[s][1] >= n_elec) { std::cerr << "Error: chosen occupied orbital (first) is out of bounds\n"; grp[s] = 0; continue; }
        const uint8_t *occ = v.orbs_at(di1[s]);
        if (oi2[s][0] == 0) {
            double tw = 1;
            uint16_t n_o2;
            if (new_hb) { oi2[s][1]++; n_o2 = oi2[s][1]; tw = calc_o2_probs_half(hb, &lv[n_long], n_elec, occ, n_o2); }
            else { n_o2 = (uint16_t)n_elec; calc_o2_probs(hb, &lv[n_long], n_elec, occ, oi2[s][1]); }
            for (size_t k = 0; k < n_o2; k++) lv[n_long + k] *= tw * sv[s];
            grp[s] = n_o2;
            n_long += n_o2;
        }
        else {
            count_symm_virt(cts, occ, n_elec, symm);
            uint32_t n_virt = count_sing_virt(occ, n_elec, symm, cts, &oi2[s][1]);
            if (n_virt == 0) grp[s] = 0;
            else {
                grp[s] = (uint16_t)n_virt;
                oi2[s][3] = (uint8_t)n_virt;
                for (size_t k = 0; k < n_virt; k++) lv[n_long + k] = sv[s] / n_virt;
                n_long += n_virt;
            }
        }
    }
    piv_comp_parallel(lv.data(), n_long, n_samp, sc.srt, sc.flag, mt, cm);
    n_short = collapse_long(sv, lv, n_short, sc.flag, grp, [&](size_t o, size_t n, size_t g) {
        di2[n] = di1[o]; oi1[n][0] = oi2[o][0]; oi1[n][1] = oi2[o][1]; oi1[n][2] = (uint8_t)g;
        if (oi2[o][0] == 1) oi1[n][3] = oi2[o][3];
    });
    sc.stage_len[2] = n_short;

    // ---- 1st unoccupied (double) (:1159-1209)
    n_long = 0;
    for (size_t s = 0; s < n_short; s++) {
        uint8_t o2u1 = oi1[s][2];
        if (oi1[s][0] == 0) {
            if (o2u1 >= n_elec) { std::cerr << "Error: chosen occupied orbital (second) is out of bounds\n"; grp[s] = 0; continue; }
            const uint8_t *occ = v.orbs_at(di2[s]);
            uint8_t o1_idx = oi1[s][1];
            int o1_spin = o1_idx / (n_elec / 2);
            int o2_spin = occ[o2u1] / n_orb;
            double tw = calc_u1_probs(hb, &lv[n_long], occ[o1_idx], occ, n_elec, new_hb && (o1_spin == o2_spin));
            uint32_t n_virt = n_orb - n_elec / 2;
            grp[s] = (uint16_t)n_virt;
            for (size_t k = 0; k < n_virt; k++) lv[n_long + k] *= sv[s] * (new_hb ? tw : 1);
            n_long += n_virt;
        }
        else {
            if (o2u1 >= oi1[s][3]) std::cerr << "Error: index of chosen virtual orbital exceeds maximum\n";
            grp[s] = 1;         // the reference resets the group size after the error branch (:1192-1196)
            lv[n_long++] = sv[s];
        }
    }
    piv_comp_parallel(lv.data(), n_long, n_samp, sc.srt, sc.flag, mt, cm);
    n_short = collapse_long(sv, lv, n_short, sc.flag, grp, [&](size_t o, size_t n, size_t g) {
        di1[n] = di2[o]; oi2[n][0] = oi1[o][0]; oi2[n][1] = oi1[o][1]; oi2[n][2] = oi1[o][2];
        oi2[n][3] = oi1[o][0] == 1 ? oi1[o][3] : (uint8_t)g;
    });
    sc.stage_len[3] = n_short;

    // ---- 2nd unoccupied (double) (:1211-1252)
    n_long = 0;
    for (size_t s = 0; s < n_short; s++) {
        size_t d = di1[s];
        uint8_t o1_idx = oi2[s][1];
        if (oi2[s][0] == 0) {
            const uint8_t *occ = v.orbs_at(d);
            uint8_t u1 = find_nth_virt(occ, o1_idx / (n_elec / 2), n_elec, n_orb, oi2[s][3]);
            det_t cd = v.dets[d];
            if (bit(cd, u1)) { std::cerr << "Error: occupied orbital chosen as 1st virtual\n"; grp[s] = 0; }
            else {
                oi2[s][3] = u1;
                uint16_t n_probs;
                double tw;
                if (new_hb) tw = calc_u2_probs_half(hb, &lv[n_long], occ[o1_idx], occ[oi2[s][2]], u1, cd, symm, &n_probs);
                else tw = calc_u2_probs(hb, &lv[n_long], occ[o1_idx], occ[oi2[s][2]], u1, symm, &n_probs);
                for (size_t k = 0; k < n_probs; k++) lv[n_long + k] *= sv[s] * (new_hb ? tw : 1);
                grp[s] = n_probs;
                n_long += n_probs;
            }
        }
        else { grp[s] = 1; lv[n_long++] = sv[s]; }
    }
    piv_comp_parallel(lv.data(), n_long, n_samp, sc.srt, sc.flag, mt, cm);
    sc.stage_len[4] = 0;
    for (size_t li = 0; li < n_long; li++) sc.stage_len[4] += !sc.flag[li];

    // ---- decode, weight, matrix element, parity (:1254-1417, spin_parity == 0)
    size_t old_len = n_short, li = 0;
    n_short = 0;
    for (size_t s = 0; s < old_len; s++) {
        for (size_t g = 0; g < grp[s]; g++, li++) {
            if (sc.flag[li]) { sc.flag[li] = 0; continue; }
            size_t d = di1[s];
            di2[n_short] = d;
            const uint8_t *occ = v.orbs_at(d);
            det_t cd = v.dets[d];
            uint8_t o1_idx = oi2[s][1];
            double tw, mel;
            if (oi2[s][0] == 0) {
                uint8_t o1 = occ[o1_idx], o2 = occ[oi2[s][2]], u1 = oi2[s][3];
                uint8_t u2_ir = symm.irrep[o1 % n_orb] ^ symm.irrep[o2 % n_orb] ^ symm.irrep[u1 % n_orb];
                uint8_t u2 = symm.lk(u2_ir, (unsigned)g + 1) + n_orb * (o2 / n_orb);
                if (bit(cd, u2)) { if (new_hb) std::cerr << "Error: occupied orbital chosen as second virtual in unnormalized heat-bath\n"; continue; }
                if (u1 == u2) { std::cerr << "Error: repeat virtual orbital chosen\n"; continue; }
                if (u1 > u2) std::swap(u1, u2);
                if (o1 > o2) std::swap(o1, o2);
                oi1[n_short][0] = o1; oi1[n_short][1] = o2; oi1[n_short][2] = u1; oi1[n_short][3] = u2;
                tw = new_hb ? calc_unnorm_wt(hb, oi1[n_short]) : calc_norm_wt(hb, oi1[n_short], occ, n_elec, cd, symm);
                tw *= p_doub;
                mel = unit_matrel ? 1.0 : doub_matrel_nosgn(oi1[n_short], sys.ints);
                mel *= doub_parity(cd, oi1[n_short]);
            }
            else {
                uint8_t o1 = occ[o1_idx];
                oi1[n_short][0] = o1;
                uint8_t u1 = virt_from_idx(cd, symm, symm.irrep[o1 % n_orb], n_orb * (o1 / n_orb), oi2[s][2]);
                if (u1 == 255) { std::cerr << "Error: virtual orbital not found\n"; continue; }
                oi1[n_short][1] = u1;
                oi1[n_short][2] = oi1[n_short][3] = 0;
                count_symm_virt(cts, occ, n_elec, symm);
                unsigned n_occ = count_sing_allowed(occ, n_elec, symm, cts);
                tw = (1 - p_doub) / n_occ / oi2[s][3];
                mel = unit_matrel ? 1.0 : sing_matrel_nosgn(oi1[n_short], occ, sys.ints, n_elec);
                mel *= sing_parity(cd, oi1[n_short]);
            }
            if (spin_parity) {          // :1326-1407: the element between the symmetrised functions, the image's probability added to the weight
                det_t nd = oi2[s][0] == 0 ? doub_det(cd, oi1[n_short]) : sing_det(cd, oi1[n_short]), tgt;
                count_symm_virt(cts, occ, n_elec, symm);
                auto fix = [&](int kind, const uint8_t *d) {
                    if (kind == 1) { const unsigned n_occ = count_sing_allowed(occ, n_elec, symm, cts); const unsigned n_virt = cts[symm.irrep[d[0] % n_orb]][0]; tw += (1 - p_doub) / n_occ / n_virt; }
                    else tw += calc_unnorm_wt(hb, d) * p_doub;
                };
                if (!adjust_tr(sys, cd, nd, occ, &mel, spin_parity, &tgt, unit_matrel, fix)) continue;
            }
            double el = lv[li] * mel / tw;
            if (fabs(el) > 1e-12) { sv[n_short] = el; n_short++; }
        }
    }
    sc.vec_len = n_short;
}

// ------------------------------------------------------------------ driver
static inline double uni(std::mt19937 &mt) { return mt() / (1. + UINT32_MAX); }

void Frisys::setup() {
    const unsigned n_orb = sys.n_orb, n_elec = sys.n_elec;
    uint8_t tmp[64];
    hf_det = gen_hf_det(n_orb, n_elec);
    occ_list(hf_det, tmp);
    sys.hf_en = diag_matrel(tmp, sys.ints, n_elec);
    mt.seed(par.seed);
    proc_scr.resize(2 * n_orb); vec_scr.resize(2 * n_orb);
    for (auto &x : proc_scr) x = mt();     // frisys_mol.cpp:133-135
    for (auto &x : vec_scr) x = mt();      // :142-144
    unsigned spawn_length = par.mat_nonz * 4 / cm.size;     // :121
    size_t adder_size = spawn_length > 1000000 ? 1000000 : spawn_length;
    sol.init(par.max_dets, adder_size, n_elec, 2, cm, proc_scr.data());
    hf_proc = sol.idx_to_proc(hf_det);
    size_t n_states = n_elec > (n_orb - n_elec / 2) ? n_elec : n_orb - n_elec / 2;
    sc.init(spawn_length, n_states);

    if (has_ham_shift) sys.hf_en = ham_shift;            // :95-98
    if (!trial_in_det.empty()) {
        // trial vector from file (:157-181): rank 0 adds every entry to trial_vec and htrial_vec, then H * trial by the general
        // h_op_offdiag / h_op_diag / add_vecs (:205-210) and both are replicated (collect_procs)
        size_t n_ex = (size_t)n_orb * n_orb * n_elec * n_elec;
        size_t tot = 2 * trial_in_det.size();
        Vec tv, ht;
        tv.init(tot, tot, n_elec, 1, cm, proc_scr.data());
        ht.init(tot * n_ex / cm.size, tot * n_ex / cm.size, n_elec, 2, cm, proc_scr.data());
        if (cm.rank == 0) for (size_t i = 0; i < trial_in_det.size(); i++) { tv.add(trial_in_det[i], trial_in_val[i], 1); ht.add(trial_in_det[i], trial_in_val[i], 1); }
        tv.perform_add(0); ht.perform_add(0);
        h_op_offdiag(ht, ht.curr_size, sys, 1, 1.0);
        ht.cur = 0;
        h_op_diag(ht, 0, 0, 1, sys);
        ht.add_vecs(0, 1);
        auto gather = [&](Vec &v, std::vector<det_t> &od, std::vector<double> &ov) {
            std::vector<std::vector<uint8_t>> snd(cm.size), rcv;
            size_t n = v.curr_size;
            std::vector<uint8_t> mine(n * 16);
            if (n) { memcpy(mine.data(), v.dets.data(), n * 8); memcpy(mine.data() + n * 8, v.vals[0].data(), n * 8); }
            for (int d = 0; d < cm.size; d++) snd[d] = mine;
            if (cm.size == 1) rcv = snd; else cm.alltoallv(snd, rcv);
            od.clear(); ov.clear();
            for (int sr = 0; sr < cm.size; sr++) {
                size_t k = rcv[sr].size() / 16, o = od.size();
                od.resize(o + k); ov.resize(o + k);
                if (k) { memcpy(&od[o], rcv[sr].data(), k * 8); memcpy(&ov[o], rcv[sr].data() + k * 8, k * 8); }
            }
        };
        gather(tv, trial_det, trial_val);
        gather(ht, htrial_det, htrial_val);
        uint8_t hf_occ[64];
        occ_list(hf_det, hf_occ);
        std::vector<uint8_t> ex;
        size_t n_doub = doub_ex_symm(hf_det, hf_occ, n_elec, n_orb, ex, sys.symm.irrep.data());      // :216-220
        size_t n_sing2 = count_singex(hf_det, hf_occ, n_elec, sys.symm);
        p_doub = (double)n_doub / (n_sing2 + n_doub);
    }
    else {
    // trial = HF; H*trial by full enumeration (:163-214, molecule.cpp:448-665)
    trial_det = {hf_det}; trial_val = {1.0};
    {
        size_t n_ex = (size_t)n_orb * n_orb * n_elec * n_elec;
        Vec ht; ht.init(2 * n_ex, 2 * n_ex, n_elec, 2, cm, proc_scr.data());
        if (cm.rank == hf_proc) ht.add(hf_det, 1, 1);
        ht.perform_add(0);
        std::vector<uint8_t> ex;
        ht.cur = 1;
        // only the rank that owns HF holds a non-zero element; the others join the collectives empty
        uint8_t hf_occ[64];
        occ_list(hf_det, hf_occ);
        const uint8_t *occ = hf_occ;
        double cur_el = cm.rank == hf_proc ? ht.vals[0][0] : 0;
        size_t n_sing = sing_ex_symm(hf_det, occ, n_elec, n_orb, ex, sys.symm.irrep.data());
        for (size_t e = 0; e < n_sing && cm.rank == hf_proc; e++) {
            double m = sing_matrel_nosgn(&ex[2 * e], occ, sys.ints, n_elec);
            det_t nd = hf_det;
            m *= sing_det_parity(&nd, &ex[2 * e]);
            m *= cur_el * 1.0;
            ht.add(nd, m, 1);
        }
        ht.perform_add(0);
        size_t n_doub = doub_ex_symm(hf_det, occ, n_elec, n_orb, ex, sys.symm.irrep.data());
        for (size_t e = 0; e < n_doub && cm.rank == hf_proc; e++) {
            double m = doub_matrel_nosgn(&ex[4 * e], sys.ints);
            det_t nd = hf_det;
            m *= doub_det_parity(&nd, &ex[4 * e]);
            m *= cur_el * 1.0;
            ht.add(nd, m, 1);
        }
        ht.perform_add(0);
        // h_op_diag(htrial, 0, 0, 1) then add_vecs(0, 1)  (molecule.cpp:205-219)
        for (size_t i = 0; i < ht.curr_size; i++) {
            double cv = ht.vals[0][i];
            if (cv != 0) {
                double de = diag_matrel(ht.orbs_at(i), sys.ints, n_elec) - sys.hf_en;
                ht.vals[0][i] = cv * (0 + 1 * de);
            }
            else ht.vals[0][i] = 0;
        }
        ht.add_vecs(0, 1);
        // collect_procs (vec_utils.hpp:920-952): every rank ends with all shards, concatenated in rank order
        {
            std::vector<std::vector<uint8_t>> snd(cm.size), rcv;
            size_t n = ht.curr_size;
            std::vector<uint8_t> mine(n * 16);
            if (n) { memcpy(mine.data(), ht.dets.data(), n * 8); memcpy(mine.data() + n * 8, ht.vals[0].data(), n * 8); }
            for (int d = 0; d < cm.size; d++) snd[d] = mine;
            if (cm.size == 1) rcv = snd; else cm.alltoallv(snd, rcv);
            htrial_det.clear(); htrial_val.clear();
            for (int sr = 0; sr < cm.size; sr++) {
                size_t k = rcv[sr].size() / 16, o = htrial_det.size();
                htrial_det.resize(o + k); htrial_val.resize(o + k);
                if (k) { memcpy(&htrial_det[o], rcv[sr].data(), k * 8); memcpy(&htrial_val[o], rcv[sr].data() + k * 8, k * 8); }
            }
        }
        p_doub = (double)n_doub / (double)(n_sing + n_doub);
        // n_hf_sing comes from count_singex (:219), identical to n_sing by construction
        size_t n_sing2 = count_singex(hf_det, occ, n_elec, sys.symm);
        p_doub = (double)n_doub / (n_sing2 + n_doub);
    }
    }
    n_determ = 0; determ_from.clear(); determ_to.clear(); determ_el.clear();
    if (!det_space.empty()) {
        // init_dense (vec_utils.hpp:858-873): add every determinant with value 1, then zero the values; they stay stored
        if (cm.rank == 0) for (det_t d : det_space) sol.add(d, 1, 1);       // rank 0 reads the file; the adds travel to their owners
        sol.perform_add(0);
        n_determ = sol.curr_size;
        for (auto &col : sol.vals) std::fill(col.begin(), col.begin() + n_determ, 0.0);
    }
    if (!ini_det.empty()) { if (cm.rank == 0) for (size_t i = 0; i < ini_det.size(); i++) sol.add(ini_det[i], ini_val[i], 1); }      // :264-274
    else if (cm.rank == hf_proc) sol.add(hf_det, 100, 1);       // :277-279
    sol.perform_add(0);
    sys.hb.set_up(sys.ints);
    srt.assign(sol.max_size, 0); keep.assign(sol.max_size, 0);
    en_shift = 0; last_one_norm = 0; iterat = 0;
    // H inside the dense space, times -eps (frisys_mol.cpp:347-397): per determinant its singles, then its doubles
    std::vector<uint8_t> ex;
    tot_dense_h = 0;
    for (size_t di = 0; di < n_determ; di++) {
        const det_t cur = sol.dets[di];
        const uint8_t *occ = sol.orbs_at(di);
        size_t n_sing = sing_ex_symm(cur, occ, n_elec, n_orb, ex, sys.symm.irrep.data());
        for (size_t e = 0; e < n_sing; e++) {
            double m = sing_matrel_nosgn(&ex[2 * e], occ, sys.ints, n_elec);
            det_t nd = cur;
            m *= sing_det_parity(&nd, &ex[2 * e]) * -par.eps;
            determ_from.push_back(di); determ_to.push_back(nd); determ_el.push_back(m);
        }
        size_t n_doub = doub_ex_symm(cur, occ, n_elec, n_orb, ex, sys.symm.irrep.data());
        for (size_t e = 0; e < n_doub; e++) {
            double m = doub_matrel_nosgn(&ex[4 * e], sys.ints);
            det_t nd = cur;
            m *= doub_det_parity(&nd, &ex[4 * e]) * -par.eps;
            determ_from.push_back(di); determ_to.push_back(nd); determ_el.push_back(m);
        }
    }
    tot_dense_h = (uint32_t)cm.sum((int)determ_el.size());       // :399
}

void Frisys::iterate(unsigned n_iter) {
    const double eps = par.eps;
    const unsigned shift_interval = 10;
    const double shift_damping = 0.05;
    for (unsigned it = 0; it < n_iter; it++, iterat++) {
        IterLog lg{};
        // :414-421
        std::copy(sol.vals[0].begin() + n_determ, sol.vals[0].begin() + sol.curr_size, sc.vec1.begin());
        for (size_t i = n_determ; i < sol.curr_size; i++) sc.det_idx1[i - n_determ] = i;
        sc.vec_len = sol.curr_size - n_determ;
        double rn[5];
        for (int k = 0; k < 5; k++) rn[k] = uni(mt);     // comp_sub broadcasts rank 0's draw (compress_utils.cpp:806);
                                                           // all ranks seed alike here, so the streams agree
        apply_HBPP_sys(sol, sc, sys, p_doub, par.new_hb, rn, par.mat_nonz - tot_dense_h, false, cm);      // :421 matr_samp - tot_dense_h
        size_t comp_len = sc.vec_len;
        lg.num_success = comp_len;
        for (int k = 0; k < 5; k++) lg.comp_len[k] = sc.stage_len[k];

        std::vector<double> &before = sol.vals[0];
        sol.cur = 1;
        sol.zero_cur();
        size_t vec_size = sol.curr_size;
        const uint8_t (*orbs)[4] = (const uint8_t (*)[4])sc.orb1.data();
        for (int add_ini = 0; add_ini < 2; add_ini++) {       // :430-471
            int num_added = 1;
            size_t s = 0;
            while (num_added > 0) {
                num_added = 0;
                while (s < comp_len) {
                    size_t d = sc.det_idx2[s];
                    double cv = before[d];
                    uint8_t ini = fabs(cv) >= par.init_thresh;
                    if (ini != add_ini) { s++; continue; }
                    double add_el = -eps * sc.vec1[s];
                    if (cv < 0) add_el *= -1;
                    det_t nd = sol.dets[d];
                    if (!(orbs[s][2] == 0 && orbs[s][3] == 0)) nd = doub_det(nd, orbs[s]);
                    else nd = sing_det(nd, orbs[s]);
                    num_added++;
                    s++;
                    if (!sol.add(nd, add_el, ini)) break;
                }
                sol.perform_add(0);
                num_added = cm.sum(num_added);      // :469
            }
        }
        if (sol.max_size > srt.size()) { srt.resize(sol.max_size); keep.resize(sol.max_size, 0); }
        // the dense block of H, exactly (:480-485; the reference's loop bound is the allocated size -- the entries beyond the filled
        // ones are zero pages of a fresh allocation and add nothing)
        for (size_t k = 0; k < determ_el.size(); k++) sol.add(determ_to[k], before[determ_from[k]] * determ_el[k], 1);
        sol.perform_add(0);
        // death / cloning :488-499
        sol.cur = 0;
        for (size_t i = 0; i < vec_size; i++) {
            double &cv = sol.vals[0][i];
            if (cv != 0) {
                if (std::isnan(sol.diag[i])) sol.diag[i] = diag_matrel(sol.orbs_at(i), sys.ints, sys.n_elec) - sys.hf_en;
                cv *= 1 - eps * (sol.diag[i] - en_shift);
            }
        }
        sol.add_vecs(0, 1);
        sol.cur = 1; sol.zero_cur(); sol.cur = 0;
        // vector compression :502-539
        unsigned n_samp = par.vec_nonz;
        double glob_norm;
        std::vector<double> loc_norms(cm.size);
        double mine = find_preserve(sol.vals[0].data() + n_determ, srt, keep, sol.curr_size - n_determ, &n_samp, &glob_norm, cm);
        {   // dense_norm (vec_utils.hpp:903-918)
            double dn = 0;
            for (size_t i = 0; i < n_determ; i++) { double e = sol.vals[0][i]; dn += e >= 0 ? e : -e; }
            glob_norm += cm.sum(dn);
        }
        lg.nkept = par.vec_nonz - n_samp;
        if ((iterat + 1) % shift_interval == 0)
            adjust_shift(&en_shift, glob_norm, &last_one_norm, par.target_norm, shift_damping / shift_interval / eps);
        lg.numer = cm.sum(sol.dot(htrial_det, htrial_val));      // :512-517
        lg.denom = cm.sum(sol.dot(trial_det, trial_val));
        lg.shift = en_shift; lg.norm = glob_norm;
        double rn_sys = uni(mt);    // the reference draws on rank 0 only and broadcasts (:528, compress_utils.cpp:291)
        cm.allgather(&mine, loc_norms.data(), sizeof(double));
        sys_comp(sol.vals[0].data() + n_determ, sol.curr_size - n_determ, loc_norms.data(), n_samp, keep, rn_sys, cm);
        for (size_t i = 0; i < sol.curr_size - n_determ; i++) {
            if (keep[i]) { sol.del_at_pos(i + n_determ); keep[i] = 0; }
        }
        lg.n_nonz = sol.n_nonz; lg.curr_size = sol.curr_size;
        log.push_back(lg);
    }
}


// ------------------------------------------------------------------ Hubbard-Holstein
uint64_t hash_fxn_hh(const uint8_t *occ, unsigned n_elec, const uint8_t *ph, unsigned n_sites, const uint32_t *scr) {
    uint64_t hash = 0;
    for (unsigned i = 0; i < n_elec; i++) hash = 1099511628211ULL * hash + (uint32_t)((i + 1) * scr[occ[i]]);
    for (unsigned i = 0; i < n_sites; i++) hash = 1099511628211ULL * hash + (uint32_t)((i + 1) * scr[ph[i]]);
    return hash;
}

unsigned hub_diag(det_t det, unsigned L) {          // doubly occupied sites
    det_t m = ((det_t)1 << L) - 1;
    return (unsigned)__builtin_popcountll(det & (det >> L) & m);
}

det_t gen_neel_det_1D(unsigned L, unsigned n_elec) {
    // alpha electrons on the even sites 0, 2, ..., beta electrons on the odd sites 1, 3, ... (n_elec / 2 of each); no phonons
    det_t d = 0;
    for (unsigned k = 0; k < n_elec / 2; k++) { d |= (det_t)1 << (2 * k); d |= (det_t)1 << (L + 2 * k + 1); }
    return d;
}

void find_neighbors_1D(det_t det, unsigned L, unsigned n_elec, uint8_t *nb) {
    const det_t E = det & (((det_t)1 << (2 * L)) - 1);
    // row 0: occupied orbitals whose right neighbour (orbital + 1) is empty; the last site of each spin chain cannot hop right
    det_t r0 = E & ~(E >> 1);
    r0 &= ~((det_t)1 << (L - 1)); r0 &= ~((det_t)1 << (2 * L - 1));
    // row 1: occupied orbitals whose left neighbour (orbital - 1) is empty; site 0 of each chain cannot hop left
    det_t r1 = E & (~E << 1);
    r1 &= ~((det_t)1 << L);
    nb[0] = (uint8_t)occ_list(r0, nb + 1);
    nb[n_elec + 1] = (uint8_t)occ_list(r1, nb + n_elec + 2);
}

void decode_phonons(det_t det, unsigned L, unsigned ph_bits, uint8_t *numbers) {
    for (unsigned s = 0; s < L; s++) numbers[s] = (uint8_t)((det >> (2 * L + s * ph_bits)) & (((det_t)1 << ph_bits) - 1));
}

bool det_from_ph(det_t det, det_t *out, unsigned L, unsigned ph_bits, unsigned site, int change) {
    unsigned sh = 2 * L + site * ph_bits;
    det_t mask = (((det_t)1 << ph_bits) - 1) << sh;
    unsigned num = (unsigned)((det & mask) >> sh);
    if (change == 1 && num == (1u << ph_bits) - 1) return false;
    if (change == -1 && num == 0) return false;
    num += change;
    *out = (det & ~mask) | ((det_t)num << sh);
    return true;
}

// hub_holstein.hpp:93-186 restated on one 64-bit word.  The reference walks the electron part byte by byte; two of its
// byte-level details are behaviour and are kept: (1) "the orbital to my right is empty" is taken as true for bit 7 of
// every byte (integer promotion of ~byte >> 1, :150), (2) the open-boundary mask is applied to byte ceil(L / 8) (:165-167).
double calc_ref_ovlp(const det_t *dets, const double *vals, size_t n, det_t ref, unsigned n_elec, unsigned L, unsigned ph_bits, double g_over_t) {
    double result = 0;
    const unsigned nbytes = (2 * L + 7) / 8;
    const det_t emask = ((det_t)1 << (2 * L)) - 1;
    const det_t byte_mask = nbytes >= 8 ? ~(det_t)0 : (((det_t)1 << (8 * nbytes)) - 1);
    for (size_t i = 0; i < n; i++) {
        det_t cur = dets[i];
        uint8_t ph[64];
        decode_phonons(cur, L, ph_bits, ph);
        if (((cur ^ ref) & emask) == 0) {
            unsigned found = 0, site_elecs = 0;
            for (unsigned s = 0; s < L && found < 2; s++) {
                unsigned n_occ = (unsigned)((ref >> s) & 1) + (unsigned)((ref >> (s + L)) & 1);
                if (ph[s] > 1 || (ph[s] == 1 && n_occ == 0)) { site_elecs = 0; break; }
                else if (ph[s] == 1) { site_elecs = n_occ; found++; }
            }
            if (found == 2) site_elecs = 0;
            result -= vals[i] * g_over_t * site_elecs;
        }
        else {
            unsigned num_ph = 0;
            for (unsigned s = 0; s < L; s++) num_ph += ph[s];
            if (num_ph != 0) continue;
            // the reference compares whole bytes, so phonon bits sharing the last electron byte take part (they are zero here)
            det_t c = cur & byte_mask, r = ref & byte_mask;
            det_t not_occ = c & ~r;
            det_t ref_left = c & (r >> 1);
            det_t not_occ_left = ((~c & byte_mask) >> 1) | 0x8080808080808080ULL;
            det_t ref_right = c & (r << 1);
            det_t not_occ_right = (~c << 1);
            // last byte: nothing chains in from a following byte for ref_left (zero), handled by byte_mask above
            unsigned ob = (L + 7) / 8;       // byte the open-boundary mask lands in
            if (ob < nbytes) ref_left &= ~((det_t)1 << (8 * ob + (L - 1) % 8));
            det_t mask = not_occ & ((ref_left & not_occ_left) | (ref_right & not_occ_right)) & byte_mask;
            if ((2 * L) % 8 != 0) mask &= ~(((det_t)0xff << (8 * (nbytes - 1))) & ~emask);      // last byte keeps only electron bits
            // the reference stops at the first byte that brings n_hop above 1, before counting that byte's common bits
            unsigned n_hop = 0, n_common = 0;
            for (unsigned b = 0; b < nbytes && n_hop <= 1; b++) {
                n_hop += (unsigned)__builtin_popcountll((mask >> (8 * b)) & 0xff);
                if (n_hop > 1) break;
                n_common += (unsigned)__builtin_popcountll(((r & c) >> (8 * b)) & 0xff);
            }
            if (n_hop == 1 && n_common == n_elec - 1) result += vals[i];
        }
    }
    return result;
}

void FrisysHH::setup() {
    const unsigned L = par.n_sites, n_elec = par.n_elec;
    mt.seed(par.seed);
    proc_scr.resize(2 * L); vec_scr.resize(2 * L);
    for (auto &x : proc_scr) x = mt();     // frisys_hh.cpp:80-83
    for (auto &x : vec_scr) x = mt();      // :88-91
    unsigned spawn_length = par.vec_nonz * 4 / cm.size;     // :94
    if (full) { size_t sl = (size_t)n_elec * 4 * par.max_dets / cm.size; spawn_length = sl > 200000 ? 200000u : (unsigned)sl; }      // frifull_hh.cpp:91-95
    sol.hh_sites = L; sol.hh_ph_bits = par.ph_bits; sol.vec_scr = vec_scr.data(); sol.n_buckets = par.max_dets;
    sol.init(par.max_dets, spawn_length, n_elec, 2, cm, proc_scr.data());      // the Adder gets spawn_length here (:100), not adder_size
    neel = gen_neel_det_1D(L, n_elec);
    ref_proc = sol.idx_to_proc(neel);
    if (ref_proc == cm.rank) sol.add(neel, 100, 1);       // :113-117
    sol.perform_add(0);
    comp1.assign(spawn_length, 0); comp2.assign(spawn_length, 0); wt_remain.assign(spawn_length, 0);
    ndiv.assign(spawn_length, 0); comp_idx.assign(2 * (size_t)spawn_length, 0); det_indices.assign(spawn_length, 0); ph_ex.assign(spawn_length, 0);
    sw.cols = 2; sw.w.assign((size_t)spawn_length * 2, 0); sw.keep.assign(spawn_length, 0);
    srt.assign(sol.max_size, 0); keep.assign(sol.max_size, 0);
    en_shift = 0; last_one_norm = 0; iterat = 0;
}

void FrisysHH::iterate(unsigned n_iter) {
    const unsigned L = par.n_sites, n_elec = par.n_elec;
    const double hub_t = 1, eps = par.eps;
    const unsigned shift_interval = 10;
    const double shift_damping = 0.05;
    size_t (*cidx)[2] = (size_t (*)[2])comp_idx.data();
    uint8_t nb[2 * 65];
    for (unsigned it = 0; it < n_iter; it++, iterat++) {
        HHLog lg{};
        size_t vec_size = sol.curr_size;
        if (full) {
            // frifull_hh.cpp:187-263: every hop and every phonon move of every stored state, in batches of about one Adder
            size_t det_idx = 0, n_add = 0;
            int num_added = 1;
            const size_t vec_size_f = sol.curr_size;
            const size_t adder_size = sol.adder_cap - n_elec * 4;
            sol.cur = 1;
            sol.zero_cur();
            while (num_added > 0) {
                num_added = 0;
                const std::vector<double> &before = sol.vals[0];
                while (det_idx < vec_size_f && (size_t)num_added < adder_size) {
                    const double cur_el = before[det_idx];
                    if (cur_el == 0) { det_idx++; continue; }
                    const det_t cur = sol.dets[det_idx];
                    const uint8_t ini = fabs(cur_el) > par.init_thresh;
                    find_neighbors_1D(cur, L, n_elec, nb);
                    for (unsigned k = 0; k < nb[0]; k++) {                    // hub_all: hops to the right, then to the left
                        unsigned o = nb[k + 1];
                        sol.add((cur & ~((det_t)1 << o)) | ((det_t)1 << (o + 1)), eps * hub_t * cur_el, ini);
                    }
                    for (unsigned k = 0; k < nb[n_elec + 1]; k++) {
                        unsigned o = nb[n_elec + 1 + k + 1];
                        sol.add((cur & ~((det_t)1 << o)) | ((det_t)1 << (o - 1)), eps * hub_t * cur_el, ini);
                    }
                    num_added += nb[0] + nb[n_elec + 1];
                    uint8_t ph[64];
                    decode_phonons(cur, L, par.ph_bits, ph);
                    const uint8_t *occ = sol.orbs_at(det_idx);
                    det_t nd;
                    for (unsigned e = 0; e < n_elec / 2; e++) {               // spin-up electrons; a doubly occupied site couples twice
                        unsigned site = occ[e];
                        unsigned pn = ph[site];
                        int doubly = (int)((cur >> (site + L)) & 1);
                        if (pn > 0) { det_from_ph(cur, &nd, L, par.ph_bits, site, -1); sol.add(nd, -eps * par.g * sqrt((double)pn) * (doubly + 1) * cur_el, ini); num_added++; }
                        if (pn + 1 < (1u << par.ph_bits)) { det_from_ph(cur, &nd, L, par.ph_bits, site, +1); sol.add(nd, -eps * par.g * sqrt((double)(pn + 1)) * (doubly + 1) * cur_el, ini); num_added++; }
                    }
                    for (unsigned e = n_elec / 2; e < n_elec; e++) {          // spin-down electrons on sites without a spin-up one
                        unsigned site = occ[e] - L;
                        if ((cur >> site) & 1) continue;
                        unsigned pn = ph[site];
                        if (pn > 0) { det_from_ph(cur, &nd, L, par.ph_bits, site, -1); sol.add(nd, -eps * par.g * sqrt((double)pn) * cur_el, ini); num_added++; }
                        if (pn + 1 < (1u << par.ph_bits)) { det_from_ph(cur, &nd, L, par.ph_bits, site, +1); sol.add(nd, -eps * par.g * sqrt((double)(pn + 1)) * cur_el, ini); num_added++; }
                    }
                    det_idx++;
                }
                n_add += (size_t)num_added;
                num_added = cm.sum(num_added);
                sol.perform_add(0);
            }
            lg.num_success = n_add;
        }
        else {
        // :187-204 electron hop vs phonon
        for (size_t i = 0; i < sol.curr_size; i++) {
            double w = fabs(sol.vals[0][i]);
            comp1[i] = w;
            if (w > 0) { sw.row(i)[0] = hub_t; sw.row(i)[1] = par.g; ndiv[i] = 0; }
            else ndiv[i] = 1;
        }
        double rn = mt() / (1. + UINT32_MAX);
        size_t comp_len = comp_sub(comp1.data(), sol.curr_size, ndiv.data(), sw, nullptr, par.vec_nonz, wt_remain.data(), rn, comp2.data(), cidx, cm);
        // :209-220 which hop / which phonon move
        for (size_t s = 0; s < comp_len; s++) {
            size_t d = cidx[s][0];
            det_indices[s] = d;
            ph_ex[s] = (uint8_t)cidx[s][1];
            if (ph_ex[s]) ndiv[s] = 2 * n_elec;
            else { find_neighbors_1D(sol.dets[d], L, n_elec, nb); ndiv[s] = nb[0] + nb[n_elec + 1]; }
            comp2[s] *= ndiv[s];
        }
        rn = mt() / (1. + UINT32_MAX);
        comp_len = comp_sub(comp2.data(), comp_len, ndiv.data(), sw, nullptr, par.vec_nonz, wt_remain.data(), rn, comp1.data(), cidx, cm);
        lg.num_success = comp_len;

        std::vector<double> &before = sol.vals[0];
        sol.cur = 1;
        sol.zero_cur();
        vec_size = sol.curr_size;
        for (int add_ini = 0; add_ini < 2; add_ini++) {       // :233-300
            int num_added = 1;
            size_t s = 0;
            while (num_added > 0) {
                num_added = 0;
                while (s < comp_len) {
                    size_t prev = cidx[s][0];
                    size_t d = det_indices[prev];
                    double cv = before[d];
                    uint8_t ini = fabs(cv) >= par.init_thresh;
                    if (ini != add_ini) { s++; continue; }
                    unsigned exc = (unsigned)(cidx[s][1] & 0xff);
                    det_t cur = sol.dets[d], nd = cur;
                    double el = comp1[s] * -eps;
                    if (cv < 0) el *= -1;
                    if (ph_ex[prev]) {
                        uint8_t ph[64];
                        decode_phonons(cur, L, par.ph_bits, ph);
                        const uint8_t *occ = sol.orbs_at(d);
                        unsigned site = occ[exc % n_elec] % L;
                        unsigned pn = ph[site];
                        if (exc < n_elec && pn > 0) { det_from_ph(cur, &nd, L, par.ph_bits, site, -1); el *= sqrt((double)pn); }
                        else if (exc >= n_elec && pn + 1 < (1u << par.ph_bits)) { det_from_ph(cur, &nd, L, par.ph_bits, site, +1); el *= sqrt((double)(pn + 1)); }
                        else el = 0;
                    }
                    else {
                        find_neighbors_1D(cur, L, n_elec, nb);
                        unsigned orig, dest;
                        if (exc < nb[0]) { orig = nb[exc + 1]; dest = orig + 1; }
                        else { orig = nb[n_elec + 1 + exc - nb[0] + 1]; dest = orig - 1; }
                        nd = (cur & ~((det_t)1 << orig)) | ((det_t)1 << dest);
                        el *= -1;      // hub_t
                    }
                    s++;
                    if (fabs(el) > 1e-9) {
                        num_added++;
                        if (!sol.add(nd, el, ini)) break;
                    }
                }
                sol.perform_add(0);
                num_added = cm.sum(num_added);
            }
        }
        }
        if (sol.max_size > srt.size()) { srt.resize(sol.max_size); keep.resize(sol.max_size, 0); }
        // diagonal :311-320
        sol.cur = 0;
        for (size_t i = 0; i < vec_size; i++) {
            double &cv = sol.vals[0][i];
            if (cv != 0) {
                if (std::isnan(sol.diag[i])) sol.diag[i] = hub_diag(sol.dets[i], L);
                uint8_t ph[64];
                decode_phonons(sol.dets[i], L, par.ph_bits, ph);
                unsigned tot = 0;
                for (unsigned k = 0; k < L; k++) tot += ph[k];
                double phonon_diag = tot * par.omega;
                cv *= 1 - eps * (sol.diag[i] * par.U + phonon_diag - par.hf_en - en_shift);
            }
        }
        sol.add_vecs(0, 1);
        // compression :323-361
        unsigned n_samp = par.vec_nonz;
        double glob_norm;
        std::vector<double> loc_norms(cm.size);
        double mine = find_preserve(sol.vals[0].data(), srt, keep, sol.curr_size, &n_samp, &glob_norm, cm);
        lg.nkept = par.vec_nonz - n_samp;
        if ((iterat + 1) % shift_interval == 0)
            adjust_shift(&en_shift, glob_norm, &last_one_norm, par.target_norm, shift_damping / shift_interval / eps);
        double numer = calc_ref_ovlp(sol.dets.data(), sol.vals[0].data(), sol.curr_size, neel, n_elec, L, par.ph_bits, par.g / hub_t);
        std::vector<double> recv(cm.size);
        cm.allgather(&numer, recv.data(), sizeof(double));      // MPI_Gather to ref_proc (:337)
        lg.numer = 0; lg.denom = 0;
        if (cm.rank == ref_proc) {
            if (std::isnan(sol.diag[0])) sol.diag[0] = hub_diag(sol.dets[0], L);
            double ref_el = sol.vals[0][0];
            double nu = (sol.diag[0] * par.U - par.hf_en) * ref_el;
            for (int p = 0; p < cm.size; p++) nu += recv[p] * -hub_t;
            lg.numer = nu; lg.denom = ref_el;
        }
        lg.shift = en_shift; lg.norm = glob_norm;
        double rn_sys = mt() / (1. + UINT32_MAX);
        cm.allgather(&mine, loc_norms.data(), sizeof(double));
        sys_comp(sol.vals[0].data(), sol.curr_size, loc_norms.data(), n_samp, keep, rn_sys, cm);
        for (size_t i = 0; i < sol.curr_size; i++) {
            if (keep[i] && !(cm.rank == 0 && i == 0)) { sol.del_at_pos(i); keep[i] = 0; }      // :356 (rank 0, not ref_proc)
        }
        lg.n_nonz = sol.n_nonz; lg.curr_size = sol.curr_size;
        log.push_back(lg);
    }
}


// ------------------------------------------------------------------ FCIQMC, near-uniform
uint64_t Rng::mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
void Rng::begin(uint64_t iter, det_t det, uint32_t attempt, uint32_t purpose) {
    if (mt) return;
    uint64_t h = mix(seed ^ 0x9e3779b97f4a7c15ULL);
    h = mix(h ^ (iter * 0xd1b54a32d192ed03ULL));
    h = mix(h ^ det);
    h = mix(h ^ (((uint64_t)attempt << 8) | purpose));
    key = h; ctr = 0;
}
double Rng::uni() {
    if (mt) return (*mt)() / (1. + UINT32_MAX);
    uint64_t h = mix(key + (uint64_t)(++ctr) * 0x9e3779b97f4a7c15ULL);
    return (uint32_t)(h >> 32) / (1. + UINT32_MAX);
}

unsigned bin_sample(unsigned n, double p, Rng &rng) {
    unsigned success = 0;
    for (unsigned i = 0; i < n; i++) success += rng.uni() < p;
    return success;
}
int round_binomially(double p, unsigned n, Rng &rng) {
    int flr = (int)floor(p);
    double prob = p - flr;
    int ret = flr * (int)n;
    for (unsigned i = 0; i < n; i++) ret += rng.uni() < prob;
    return ret;
}
static inline unsigned choose_uint(Rng &rng, unsigned nmax) { return (unsigned)(rng.uni() * nmax); }   // near_uniform.cpp:41-44

bool nu_doub_sample(det_t det, const uint8_t *occ, unsigned n_elec, const Symm &sy, unsigned counts[][2], Rng &rng, uint8_t orbs[4], double *prob) {
    const unsigned n_orb = sy.n_orb;
    // _choose_occ_pair / _tri_to_occ_pair (:46-64)
    unsigned tri = choose_uint(rng, n_elec * (n_elec - 1) / 2);
    unsigned i1 = (unsigned)((sqrt(tri * 8. + 1) - 1) / 2);
    unsigned i2 = (unsigned)(tri - i1 * (i1 + 1.) / 2);
    i1 += 1;
    unsigned orb1 = occ[i1], orb2 = occ[i2];
    unsigned spin1 = i1 / (n_elec / 2), spin2 = i2 / (n_elec / 2);
    unsigned sym_prod = sy.irrep[orb1 % n_orb] ^ sy.irrep[orb2 % n_orb];
    // _count_doub_virt (:66-87)
    int same_symm = sym_prod == 0 && spin1 == spin2;
    unsigned n_allow = spin1 == spin2 ? n_orb - n_elec / 2 : 2 * n_orb - n_elec;
    for (unsigned i = 0; i < N_IRREPS; i++) {
        if (counts[i ^ sym_prod][spin2] == (unsigned)same_symm) n_allow -= counts[i][spin1];
        if (spin1 != spin2 && counts[i ^ sym_prod][spin1] == (unsigned)same_symm) n_allow -= counts[i][spin2];
    }
    if (n_allow == 0) return false;
    // _doub_choose_virt1 (:89-173)
    int virt_choice;
    unsigned a_spin, b_spin, n_virt2, orbital;
    if (n_allow <= 3) {
        virt_choice = (int)choose_uint(rng, n_allow);
        if (spin1 == spin2) { a_spin = spin1; b_spin = a_spin; } else { a_spin = 0; b_spin = 1; }
        orbital = 0;
        while (virt_choice >= 0 && orbital < n_orb) {
            if (!((det >> (orbital + a_spin * n_orb)) & 1)) {
                unsigned a_symm = sy.irrep[orbital];
                n_virt2 = counts[sym_prod ^ a_symm][b_spin] - (sym_prod == 0 && a_spin == b_spin);
                if (n_virt2 != 0) virt_choice -= 1;
            }
            orbital += 1;
        }
        if (virt_choice >= 0) {
            a_spin = 1; b_spin = 0;
            while (virt_choice >= 0 && orbital < 2 * n_orb) {
                if (!((det >> orbital) & 1)) {
                    unsigned a_symm = sy.irrep[orbital - n_orb];
                    n_virt2 = counts[sym_prod ^ a_symm][b_spin] - (sym_prod == 0 && a_spin == b_spin);
                    if (n_virt2 != 0) virt_choice -= 1;
                }
                orbital += 1;
            }
            orbital -= n_orb;
        }
        virt_choice = (int)(orbital - 1 + a_spin * n_orb);
    }
    else {
        n_virt2 = 0;
        while (n_virt2 == 0) {
            if (spin1 == spin2) { a_spin = spin1; b_spin = a_spin; virt_choice = (int)(choose_uint(rng, n_orb) + a_spin * n_orb); }
            else { virt_choice = (int)choose_uint(rng, 2 * n_orb); a_spin = virt_choice / n_orb; b_spin = 1 - a_spin; }
            if (!((det >> virt_choice) & 1)) {
                unsigned a_symm = sy.irrep[virt_choice % n_orb];
                n_virt2 = counts[sym_prod ^ a_symm][b_spin] - (sym_prod == 0 && a_spin == b_spin);
            }
        }
    }
    unsigned unocc1 = (unsigned)virt_choice;
    a_spin = unocc1 / n_orb;
    b_spin = spin1 ^ spin2 ^ a_spin;
    unsigned a_symm = sy.irrep[unocc1 % n_orb], b_symm = sym_prod ^ a_symm;
    unsigned m_a_b = counts[b_symm][b_spin] - (sym_prod == 0 && a_spin == b_spin);
    // _doub_choose_virt2 (:176-191)
    int orb_idx = (int)choose_uint(rng, m_a_b);
    unsigned unocc2 = 0, symm_idx = 1;
    while (orb_idx >= 0) {
        unocc2 = sy.lk(b_symm, symm_idx) + b_spin * n_orb;
        if (!((det >> unocc2) & 1) && unocc2 != unocc1) orb_idx -= 1;
        symm_idx += 1;
    }
    unsigned m_b_a = counts[a_symm][a_spin] - (sym_prod == 0 && a_spin == b_spin);
    *prob = 2. / n_elec / (n_elec - 1) / n_allow * (1. / m_a_b + 1. / m_b_a);
    orbs[0] = (uint8_t)orb2; orbs[1] = (uint8_t)orb1;
    if (unocc1 < unocc2) { orbs[2] = (uint8_t)unocc1; orbs[3] = (uint8_t)unocc2; } else { orbs[2] = (uint8_t)unocc2; orbs[3] = (uint8_t)unocc1; }
    return true;
}

void nu_sing_setup(const uint8_t *occ, unsigned n_elec, const Symm &sy, unsigned counts[][2], unsigned *m_allow, unsigned *delta_s) {
    unsigned ds = 0;
    for (unsigned e = 0; e < n_elec; e++) {
        unsigned na = counts[sy.irrep[occ[e] % sy.n_orb]][e / (n_elec / 2)];
        m_allow[e] = na;
        if (na == 0) ds++;
    }
    *delta_s = ds;
}
void nu_sing_sample(det_t det, const uint8_t *occ, unsigned n_elec, const Symm &sy, const unsigned *m_allow, unsigned delta_s, Rng &rng, uint8_t orbs[2], double *prob) {
    unsigned elec = 0, na = 0;
    while (na == 0) { elec = choose_uint(rng, n_elec); na = m_allow[elec]; }        // _sing_choose_occ (:248-257)
    unsigned occ_orb = occ[elec], occ_symm = sy.irrep[occ_orb % sy.n_orb], spin = occ_orb / sy.n_orb;
    int symm_idx = -1;
    unsigned orbital = 0;
    while (symm_idx == -1) {                                                          // _sing_choose_virt (:260-274)
        symm_idx = (int)choose_uint(rng, sy.lk(occ_symm, 0));
        orbital = spin * sy.n_orb + sy.lk(occ_symm, symm_idx + 1);
        if ((det >> orbital) & 1) symm_idx = -1;
    }
    *prob = 1. / m_allow[elec] / (n_elec - delta_s);
    orbs[0] = (uint8_t)occ_orb; orbs[1] = (uint8_t)orbital;
}


void setup_alias(const double *probs, unsigned *aliases, double *alias_probs, size_t n) {
    size_t n_s = 0, n_b = 0;
    unsigned smaller[256], bigger[256];
    for (unsigned i = 0; i < n; i++) {
        aliases[i] = i;
        alias_probs[i] = n * probs[i];
        if (alias_probs[i] < 1) smaller[n_s++] = i; else bigger[n_b++] = i;
    }
    while (n_s > 0 && n_b > 0) {
        unsigned sm = smaller[n_s - 1], b = bigger[n_b - 1];
        aliases[sm] = b;
        alias_probs[b] += alias_probs[sm] - 1;
        if (alias_probs[b] < 1) { smaller[n_s - 1] = b; n_b--; }
        else n_s--;
    }
}
unsigned sample_alias_one(const unsigned *aliases, const double *alias_probs, size_t n, Rng &rng) {
    uint8_t chosen = (uint8_t)(rng.uni() * n);
    if (rng.uni() < alias_probs[chosen]) return chosen;
    return aliases[chosen];
}

unsigned hb_doub_multi(det_t det, const uint8_t *occ, unsigned n_elec, const Symm &sy, const HBInfo &hb, unsigned num_sampl, Rng &rng, uint64_t iter,
                       uint8_t *orbs, double *prob, uint32_t *att) {
    const unsigned n_orb = hb.n_orb, n_virt = n_orb - n_elec / 2;
    unsigned alias_idx[64]; double alias_probs[64], probs[64];
    // first occupied orbital of every sample (:614-617), then how many samples chose each electron (:619-625)
    calc_o1_probs(hb, probs, n_elec, occ, 0);
    setup_alias(probs, alias_idx, alias_probs, n_elec);
    std::vector<unsigned> o1_samples(n_elec, 0);
    for (unsigned i = 0; i < num_sampl; i++) {
        rng.begin(iter, det, i, RNG_HB_O1);
        o1_samples[sample_alias_one(alias_idx, alias_probs, n_elec, rng)]++;
    }
    unsigned tot = 0;
    std::vector<uint8_t> o2s, u1s;
    for (unsigned e = 0; e < n_elec; e++) {
        unsigned loc = o1_samples[e];
        if (loc == 0) continue;
        o2s.resize(loc); u1s.resize(loc);
        calc_o2_probs(hb, probs, n_elec, occ, e);           // second occupied orbital for the whole group (:636-639)
        unsigned o1 = occ[e];
        setup_alias(probs, alias_idx, alias_probs, n_elec);
        for (unsigned k = 0; k < loc; k++) { rng.begin(iter, det, (e << 20) | k, RNG_HB_O2); o2s[k] = (uint8_t)sample_alias_one(alias_idx, alias_probs, n_elec, rng); }
        calc_u1_probs(hb, probs, o1, occ, n_elec, 0);       // first virtual for the whole group (:641-644)
        setup_alias(probs, alias_idx, alias_probs, n_virt);
        for (unsigned k = 0; k < loc; k++) { rng.begin(iter, det, (e << 20) | k, RNG_HB_U1); u1s[k] = (uint8_t)sample_alias_one(alias_idx, alias_probs, n_virt, rng); }
        for (unsigned k = 0; k < loc; k++) {                // second virtual, one sample at a time (:646-679)
            unsigned o2 = occ[o2s[k]];
            unsigned u1 = find_nth_virt(occ, o1 / n_orb, n_elec, n_orb, u1s[k]);
            unsigned u2_symm = sy.irrep[o1 % n_orb] ^ sy.irrep[o2 % n_orb] ^ sy.irrep[u1 % n_orb];
            uint16_t num_u2 = 0;
            double u2_norm = calc_u2_probs(hb, probs, o1, o2, u1, sy, &num_u2);
            if (u2_norm != 0) {
                setup_alias(probs, alias_idx, alias_probs, num_u2);
                rng.begin(iter, det, (e << 20) | k, RNG_HB_U2);
                unsigned u2 = sample_alias_one(alias_idx, alias_probs, num_u2, rng);
                u2 = sy.lk(u2_symm, u2 + 1) + n_orb * (o2 / n_orb);
                if ((det >> u2) & 1) continue;
                uint8_t *o = &orbs[4 * tot];
                if (o1 > o2) { o[0] = (uint8_t)o2; o[1] = (uint8_t)o1; } else { o[0] = (uint8_t)o1; o[1] = (uint8_t)o2; }
                if (u1 > u2) { o[2] = (uint8_t)u2; o[3] = (uint8_t)u1; } else { o[2] = (uint8_t)u1; o[3] = (uint8_t)u2; }
                prob[tot] = calc_norm_wt(hb, o, occ, n_elec, det, sy);
                att[tot] = (e << 20) | k;
                tot++;
            }
        }
    }
    return tot;
}

void Fciqmc::setup() {
    const unsigned n_orb = sys.n_orb, n_elec = sys.n_elec;
    uint8_t tmp[64];
    hf_det = gen_hf_det(n_orb, n_elec);
    occ_list(hf_det, tmp);
    sys.hf_en = diag_matrel(tmp, sys.ints, n_elec);
    mt.seed(par.seed + (uint32_t)cm.rank);          // fciqmc_mol.cpp:102-104: one generator per process (frimulti_mol.cpp:84-86 alike)
    proc_scr.assign(2 * n_orb, 0); vec_scr.resize(2 * n_orb);
    if (cm.rank == 0) for (auto &x : proc_scr) x = mt();     // :123-131, then MPI_Bcast
    if (cm.size > 1) {
        std::vector<uint32_t> all((size_t)cm.size * 2 * n_orb);
        cm.allgather(proc_scr.data(), all.data(), sizeof(uint32_t) * 2 * n_orb);
        std::copy(all.begin(), all.begin() + 2 * n_orb, proc_scr.begin());
    }
    for (auto &x : vec_scr) x = mt();      // :134-136
    if (par.counter_rng) { rng.mt = nullptr; rng.seed = par.seed; } else rng.mt = &mt;
    unsigned spawn_length = par.target_walkers / cm.size / cm.size * 2;      // :107
    if (par.multi) spawn_length = par.mat_nonz * 2 / cm.size / cm.size;     // frimulti_mol.cpp:89
    sol.init(par.max_dets, spawn_length, n_elec, 1, cm, proc_scr.data());
    hf_proc = sol.idx_to_proc(hf_det);
    if (!trial_in_det.empty()) {
        // --trial_vec (:150-177).  Every entry goes in through `while (!vec.add(...)) vec.perform_add(0)`: add() returns false when
        // the entry it just stored filled the Adder, and the loop then stores it a second time.  trial_vec's Adder holds exactly
        // n_trial entries, so the last entry of the file enters the trial vector twice (htrial_vec's is far larger): reproduced.
        size_t n_trial = trial_in_det.size(), n_ex = (size_t)n_orb * n_orb * n_elec * n_elec;
        Vec tv, ht;
        tv.init(n_trial, n_trial, n_elec, 1, cm, proc_scr.data());
        ht.init(n_trial * n_ex / cm.size, n_trial * n_ex / cm.size, n_elec, 2, cm, proc_scr.data());
        if (cm.rank == 0) for (size_t i = 0; i < n_trial; i++) {
            if (par.multi) {        // frimulti_mol.cpp:149-157: a filled Adder is an error there -- on one rank the last entry of ANY trial file fills trial_vec's
                if (!tv.add(trial_in_det[i], trial_in_val[i], 1)) throw std::runtime_error("Insufficient memory allocated in adder");
                if (!ht.add(trial_in_det[i], trial_in_val[i], 1)) throw std::runtime_error("Insufficient memory allocated in adder");
                continue;
            }
            while (!tv.add(trial_in_det[i], trial_in_val[i], 1)) tv.perform_add(0);
            while (!ht.add(trial_in_det[i], trial_in_val[i], 1)) ht.perform_add(0);
        }
        tv.perform_add(0); ht.perform_add(0);
        h_op_offdiag(ht, ht.curr_size, sys, 1, 1.0);
        ht.cur = 0;
        h_op_diag(ht, 0, 0, 1, sys);
        ht.add_vecs(0, 1);
        auto gather = [&](Vec &v, std::vector<det_t> &od, std::vector<double> &ov) {
            std::vector<std::vector<uint8_t>> snd(cm.size), rcv;
            size_t n = v.curr_size;
            std::vector<uint8_t> mine(n * 16);
            if (n) { memcpy(mine.data(), v.dets.data(), n * 8); memcpy(mine.data() + n * 8, v.vals[0].data(), n * 8); }
            for (int d = 0; d < cm.size; d++) snd[d] = mine;
            if (cm.size == 1) rcv = snd; else cm.alltoallv(snd, rcv);
            od.clear(); ov.clear();
            for (int sr = 0; sr < cm.size; sr++) {
                size_t k = rcv[sr].size() / 16, o = od.size();
                od.resize(o + k); ov.resize(o + k);
                if (k) { memcpy(&od[o], rcv[sr].data(), k * 8); memcpy(&ov[o], rcv[sr].data() + k * 8, k * 8); }
            }
        };
        gather(tv, trial_det, trial_val);
        gather(ht, htrial_det, htrial_val);
        std::vector<uint8_t> ex;
        size_t n_doub = doub_ex_symm(hf_det, tmp, n_elec, n_orb, ex, sys.symm.irrep.data());
        size_t n_sing2 = count_singex(hf_det, tmp, n_elec, sys.symm);
        p_doub = (double)n_doub / (n_sing2 + n_doub);
    }
    else {
    // trial = HF, H trial by enumeration on the rank that owns HF, then replicated (:148-191), as in frisys_mol
    trial_det = {hf_det}; trial_val = {1.0};
    {
        size_t n_ex = (size_t)n_orb * n_orb * n_elec * n_elec;
        Vec ht; ht.init(2 * n_ex, 2 * n_ex, n_elec, 2, cm, proc_scr.data());
        if (cm.rank == hf_proc) ht.add(hf_det, 1, 1);
        ht.perform_add(0);
        std::vector<uint8_t> ex;
        ht.cur = 1;
        double cur_el = cm.rank == hf_proc ? ht.vals[0][0] : 0;
        size_t n_sing = sing_ex_symm(hf_det, tmp, n_elec, n_orb, ex, sys.symm.irrep.data());
        for (size_t e = 0; e < n_sing && cm.rank == hf_proc; e++) {
            double m = sing_matrel_nosgn(&ex[2 * e], tmp, sys.ints, n_elec);
            det_t nd = hf_det;
            m *= sing_det_parity(&nd, &ex[2 * e]);
            ht.add(nd, m * cur_el * 1.0, 1);
        }
        ht.perform_add(0);
        size_t n_doub = doub_ex_symm(hf_det, tmp, n_elec, n_orb, ex, sys.symm.irrep.data());
        for (size_t e = 0; e < n_doub && cm.rank == hf_proc; e++) {
            double m = doub_matrel_nosgn(&ex[4 * e], sys.ints);
            det_t nd = hf_det;
            m *= doub_det_parity(&nd, &ex[4 * e]);
            ht.add(nd, m * cur_el * 1.0, 1);
        }
        ht.perform_add(0);
        for (size_t i = 0; i < ht.curr_size; i++) {
            double cv = ht.vals[0][i];
            if (cv != 0) ht.vals[0][i] = cv * (diag_matrel(ht.orbs_at(i), sys.ints, n_elec) - sys.hf_en);
        }
        ht.add_vecs(0, 1);
        // collect_procs (vec_utils.hpp:920-952): every rank ends with all shards, concatenated in rank order
        std::vector<std::vector<uint8_t>> snd(cm.size), rcv;
        size_t n = ht.curr_size;
        std::vector<uint8_t> mine(n * 16);
        if (n) { memcpy(mine.data(), ht.dets.data(), n * 8); memcpy(mine.data() + n * 8, ht.vals[0].data(), n * 8); }
        for (int d = 0; d < cm.size; d++) snd[d] = mine;
        if (cm.size == 1) rcv = snd; else cm.alltoallv(snd, rcv);
        htrial_det.clear(); htrial_val.clear();
        for (int sr = 0; sr < cm.size; sr++) {
            size_t k = rcv[sr].size() / 16, o = htrial_det.size();
            htrial_det.resize(o + k); htrial_val.resize(o + k);
            if (k) { memcpy(&htrial_det[o], rcv[sr].data(), k * 8); memcpy(&htrial_val[o], rcv[sr].data() + k * 8, k * 8); }
        }
        size_t n_sing2 = count_singex(hf_det, tmp, n_elec, sys.symm);
        p_doub = (double)n_doub / (n_sing2 + n_doub);
    }
    }
    if (!ini_det.empty()) {      // fciqmc_mol.cpp:226-237 / fciqmc_fp_mol.cpp:233-246: `while (!add) perform_add`; frimulti_mol.cpp:205-215: plain add()
        if (cm.rank == 0) for (size_t i = 0; i < ini_det.size(); i++) {
            if (par.multi) sol.add(ini_det[i], ini_val[i], 1);
            else while (!sol.add(ini_det[i], ini_val[i], 1)) sol.perform_add(0);
        }
    }
    else if (cm.rank == hf_proc) sol.add(hf_det, 100, 1);      // :239-243
    sol.perform_add(0);
    if (par.heat_bath) sys.hb.set_up(sys.ints);       // :310-313
    en_shift = 0; last_norm = 0; iterat = 0;
    if (par.multi) {                                   // frimulti_mol.cpp:227-233
        loc_norms.assign(cm.size, 0);
        double mine = sol.local_norm();
        cm.allgather(&mine, loc_norms.data(), sizeof(double));
        glob_norm = 0;
        for (int p = 0; p < cm.size; p++) glob_norm += loc_norms[p];
        srt.assign(sol.max_size, 0); keep.assign(sol.max_size, 0);
    }
}
void Fciqmc::iterate(unsigned n_iter) {
    const unsigned n_elec = sys.n_elec;
    const double eps = par.eps;
    const unsigned shift_interval = 10;
    const double shift_damping = 0.05;
    std::vector<uint8_t> orbs; std::vector<double> probs;
    for (unsigned it = 0; it < n_iter; it++, iterat++) {
        FciqmcLog lg{};
        int n_nonz = 0; uint32_t n_ini = 0; size_t n_spawn = 0;
        for (size_t d = 0; d < sol.curr_size; d++) {          // :331-411
            double &cur = sol.vals[0][d];
            int cur_i = (int)cur;
            unsigned n_walk = (unsigned)abs(cur_i);
            if (par.fp) {                                   // fciqmc_fp_mol.cpp:342: one draw for every stored position, zeros included
                rng.begin(iterat, sol.dets[d], 0, RNG_NWALK);
                n_walk = (unsigned)round_binomially(fabs(cur), 1, rng);
            }
            if (n_walk == 0) continue;
            n_nonz++;
            int ini = n_walk > par.init_thresh;
            n_ini += ini;
            int sign = par.fp ? 1 - 2 * (cur < 0) : (cur_i < 0 ? -1 : 1);
            const det_t det = sol.dets[d];
            const uint8_t *occ = sol.orbs_at(d);
            unsigned counts[N_IRREPS][2];
            count_symm_virt(counts, occ, n_elec, sys.symm);
            rng.begin(iterat, det, 0, RNG_BIN);
            unsigned n_doub = bin_sample(n_walk, p_doub, rng);
            unsigned n_sing = n_walk - n_doub;
            // doubles: all samples first (doub_multin), then the rounding draws (:363-378)
            if (orbs.size() < 4 * (size_t)n_walk) { orbs.resize(4 * (size_t)n_walk); probs.resize(n_walk); }
            unsigned nn = 0;
            std::vector<uint32_t> att(n_doub);       // which attempt produced sample w: counter mode keys the rounding draw by it
            if (par.heat_bath) nn = hb_doub_multi(det, occ, n_elec, sys.symm, sys.hb, n_doub, rng, iterat, orbs.data(), probs.data(), att.data());      // :366-368
            else for (unsigned i = 0; i < n_doub; i++) {
                rng.begin(iterat, det, i, RNG_DOUB);
                if (nu_doub_sample(det, occ, n_elec, sys.symm, counts, rng, &orbs[4 * nn], &probs[nn])) { att[nn] = i; nn++; }
            }
            for (unsigned w = 0; w < nn; w++) {
                double m = doub_matrel_nosgn(&orbs[4 * w], sys.ints);
                m *= eps / probs[w] / p_doub;
                double sp = m;                              // fciqmc_fp_mol.cpp:385-390: only small spawns are rounded
                if (!par.fp || fabs(m) < 0.01) { rng.begin(iterat, det, att[w], RNG_ROUND_D); sp = round_binomially(m, 1, rng); }
                if (sp != 0) {
                    det_t nd = det;
                    sp *= -doub_det_parity(&nd, &orbs[4 * w]) * sign;
                    if (!sol.add(nd, sp, (uint8_t)ini)) throw std::runtime_error("Insufficient memory allocated in adder");
                    n_spawn++;
                }
            }
            // singles (:380-394)
            unsigned m_allow[64], delta_s;
            nu_sing_setup(occ, n_elec, sys.symm, counts, m_allow, &delta_s);
            unsigned ns = delta_s == n_elec ? 0 : n_sing;
            for (unsigned j = 0; j < ns; j++) {
                rng.begin(iterat, det, j, RNG_SING);
                nu_sing_sample(det, occ, n_elec, sys.symm, m_allow, delta_s, rng, &orbs[2 * j], &probs[j]);
            }
            for (unsigned w = 0; w < ns; w++) {
                double m = sing_matrel_nosgn(&orbs[2 * w], occ, sys.ints, n_elec);
                m *= eps / probs[w] / (1 - p_doub);
                double sp = m;
                if (!par.fp || fabs(m) < 0.01) { rng.begin(iterat, det, w, RNG_ROUND_S); sp = round_binomially(m, 1, rng); }
                if (sp != 0) {
                    det_t nd = det;
                    sp *= -sing_det_parity(&nd, &orbs[2 * w]) * sign;
                    if (!sol.add(nd, sp, (uint8_t)ini)) throw std::runtime_error("Insufficient memory allocated in adder");
                    n_spawn++;
                }
            }
            // death / cloning (:396-402).  The reference calls del_at_pos BEFORE it stores the new value (:400-403), while the
            // old one is still non-zero, so del_at_pos declines: a determinant left without walkers keeps its slot and its
            // hash entry with value 0 (only initiator spawns can revive it).  Reproduced: nothing is deleted here.
            if (std::isnan(sol.diag[d])) sol.diag[d] = diag_matrel(occ, sys.ints, n_elec) - sys.hf_en;
            if (par.fp) { cur *= 1 - eps * (sol.diag[d] - en_shift); continue; }        // fciqmc_fp_mol.cpp:423-424
            double m = (1 - eps * (sol.diag[d] - en_shift)) * sign;
            rng.begin(iterat, det, 0, RNG_DEATH);
            int new_val = round_binomially(m, n_walk, rng);
            if (new_val == 0) sol.del_at_pos(d);      // value still non-zero: no effect, as in the reference
            cur = new_val;
        }
        sol.perform_add(0);
        if (par.fp) {                                       // fciqmc_fp_mol.cpp:428-441
            for (size_t d = 0; d < sol.curr_size; d++) {
                double &c = sol.vals[0][d];
                if (c == 0) continue;
                if (fabs(c) < 1) { rng.begin(iterat, sol.dets[d], 0, RNG_COMP); c = round_binomially(c, 1, rng); }
                if (c == 0) sol.del_at_pos(d);
            }
        }
        double glob_norm = 0;
        if ((iterat + 1) % shift_interval == 0) {          // :415-427
            glob_norm = cm.sum(sol.local_norm());
            adjust_shift(&en_shift, glob_norm, &last_norm, par.target_walkers, shift_damping / eps / shift_interval);
            (void)cm.sum(n_nonz);                          // glob_nnonz (:422)
        }
        lg.numer = sol.dot(htrial_det, htrial_val);
        lg.denom = sol.dot(trial_det, trial_val);
        if (cm.size > 1) {                                  // MPI_Gather to the rank that owns HF, added up in rank order (:433-441)
            std::vector<double> all(2 * (size_t)cm.size);
            double mine2[2] = {lg.numer, lg.denom};
            cm.allgather(mine2, all.data(), 16);
            // fciqmc_fp_mol.cpp:461-462 overwrites slot 0 of the gathered terms with the gathering rank's own (rank 0's terms are lost and
            // the HF owner's count twice unless it is rank 0): reproduced as the rank that owns HF sees it
            if (par.fp) { all[0] = all[2 * (size_t)hf_proc]; all[1] = all[2 * (size_t)hf_proc + 1]; }
            lg.numer = 0; lg.denom = 0;
            for (int q = 0; q < cm.size; q++) { lg.numer += all[2 * q]; lg.denom += all[2 * q + 1]; }
        }
        lg.shift = en_shift; lg.norm = glob_norm; lg.n_nonz = n_nonz; lg.n_ini = n_ini; lg.curr_size = sol.curr_size; lg.n_spawn = n_spawn;
        log.push_back(lg);
    }
}

// frimulti_mol.cpp:296-425.  Only --distribution HB gets past the reference's argument check (:38-46: both branches compare with
// "HB"), so the doubles always come from hb_doub_multi and the singles from sing_multin.
void Fciqmc::iterate_multi(unsigned n_iter) {
    const unsigned n_elec = sys.n_elec;
    const double eps = par.eps;
    const unsigned shift_interval = 10;
    const double shift_damping = 0.05;
    std::vector<uint8_t> orbs; std::vector<double> probs; std::vector<uint32_t> att;
    for (unsigned it = 0; it < n_iter; it++, iterat++) {
        FciqmcLog lg{};
        uint32_t n_ini = 0; size_t n_spawn = 0;
        auto bcast0 = [&]() {                              // rank 0 draws, everybody receives (:301-305, :412-414 + compress_utils.cpp:291)
            double mine = cm.rank == 0 ? mt() / (1. + UINT32_MAX) : 0.0;
            if (cm.size == 1) return mine;
            std::vector<double> all(cm.size);
            cm.allgather(&mine, all.data(), sizeof(double));
            return all[0];
        };
        double rn_sys = bcast0();
        const unsigned curr_mat_samp = iterat < 10 ? par.mat_nonz / 10 : par.mat_nonz;
        double lbound = seed_sys(loc_norms.data(), &rn_sys, curr_mat_samp, cm);
        if (sol.max_size > srt.size()) { srt.resize(sol.max_size); keep.resize(sol.max_size, 0); }
        for (size_t d = 0; d < sol.curr_size; d++) {
            double &cur = sol.vals[0][d];
            const double weight = fabs(cur);
            if (weight == 0) continue;
            unsigned n_walk = 0;
            lbound += weight;
            while (rn_sys < lbound) { n_walk++; rn_sys += glob_norm / curr_mat_samp; }
            double colsamp_wt = weight / (glob_norm / curr_mat_samp);
            if (colsamp_wt > 1) colsamp_wt = 1;
            const int ini = weight > par.init_thresh_f;
            n_ini += ini;
            const det_t det = sol.dets[d];
            const uint8_t *occ = sol.orbs_at(d);
            unsigned counts[N_IRREPS][2];
            count_symm_virt(counts, occ, n_elec, sys.symm);
            rng.begin(iterat, det, 0, RNG_BIN);
            unsigned n_doub = bin_sample(n_walk, p_doub, rng);
            unsigned n_sing = n_walk - n_doub;
            if (orbs.size() < 4 * (size_t)n_walk + 4) { orbs.resize(4 * (size_t)n_walk + 4); probs.resize(n_walk + 1); att.resize(n_walk + 1); }
            unsigned nn = hb_doub_multi(det, occ, n_elec, sys.symm, sys.hb, n_doub, rng, iterat, orbs.data(), probs.data(), att.data());
            for (unsigned w = 0; w < nn; w++) {
                double m = doub_matrel_nosgn(&orbs[4 * w], sys.ints);
                if (fabs(m) > 1e-9) {
                    det_t nd = det;
                    m *= -eps / probs[w] / p_doub / n_walk * cur * doub_det_parity(&nd, &orbs[4 * w]) / colsamp_wt;
                    if (!sol.add(nd, m, (uint8_t)ini)) throw std::runtime_error("Insufficient memory allocated in adder.");
                    n_spawn++;
                }
            }
            unsigned m_allow[64], delta_s;
            nu_sing_setup(occ, n_elec, sys.symm, counts, m_allow, &delta_s);
            unsigned ns = delta_s == n_elec ? 0 : n_sing;
            for (unsigned j = 0; j < ns; j++) {
                rng.begin(iterat, det, j, RNG_SING);
                nu_sing_sample(det, occ, n_elec, sys.symm, m_allow, delta_s, rng, &orbs[2 * j], &probs[j]);
            }
            for (unsigned w = 0; w < ns; w++) {
                double m = sing_matrel_nosgn(&orbs[2 * w], occ, sys.ints, n_elec);
                if (fabs(m) > 1e-9) {
                    det_t nd = det;
                    m *= -eps / probs[w] / (1 - p_doub) / n_walk * cur * sing_det_parity(&nd, &orbs[2 * w]) / colsamp_wt;
                    if (!sol.add(nd, m, (uint8_t)ini)) throw std::runtime_error("Insufficient memory allocated in adder.");
                    n_spawn++;
                }
            }
            if (std::isnan(sol.diag[d])) sol.diag[d] = diag_matrel(occ, sys.ints, n_elec) - sys.hf_en;
            cur *= 1 - eps * (sol.diag[d] - en_shift);
        }
        sol.perform_add(0);
        if (sol.max_size > srt.size()) { srt.resize(sol.max_size); keep.resize(sol.max_size, 0); }
        unsigned n_samp = par.vec_nonz;
        double mine = find_preserve(sol.vals[0].data(), srt, keep, sol.curr_size, &n_samp, &glob_norm, cm);
        nkept = par.vec_nonz - n_samp;
        if ((iterat + 1) % shift_interval == 0) adjust_shift(&en_shift, glob_norm, &last_norm, par.target_norm, shift_damping / shift_interval / eps);
        lg.numer = sol.dot(htrial_det, htrial_val);
        lg.denom = sol.dot(trial_det, trial_val);
        if (cm.size > 1) {
            std::vector<double> all(2 * (size_t)cm.size);
            double mine2[2] = {lg.numer, lg.denom};
            cm.allgather(mine2, all.data(), 16);
            lg.numer = 0; lg.denom = 0;
            for (int q = 0; q < cm.size; q++) { lg.numer += all[2 * q]; lg.denom += all[2 * q + 1]; }
        }
        rn_sys = bcast0();
        cm.allgather(&mine, loc_norms.data(), sizeof(double));
        sys_comp(sol.vals[0].data(), sol.curr_size, loc_norms.data(), n_samp, keep, rn_sys, cm);
        // :416-421: the test `indices()[det_idx] != hf_det` compares addresses, so it is always true and HF is deleted like any other
        for (size_t i = 0; i < sol.curr_size; i++) if (keep[i]) { sol.del_at_pos(i); keep[i] = 0; }
        lg.shift = en_shift; lg.norm = glob_norm; lg.n_nonz = sol.n_nonz; lg.n_ini = n_ini; lg.curr_size = sol.curr_size; lg.n_spawn = n_spawn;
        log.push_back(lg);
    }
}

}  // namespace fo
