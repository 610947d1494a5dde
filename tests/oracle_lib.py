"""ctypes wrapper of the CPU oracle (oracle/_build/libfries_oracle.so).  Test infrastructure
only: nothing under fries_amd/ may import this."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "_build", "libfries_oracle.so")

LOG_DTYPE = np.dtype([("numer", "f8"), ("denom", "f8"), ("shift", "f8"), ("norm", "f8"), ("nkept", "u4"), ("n_nonz", "i4"),
                      ("curr_size", "u4"), ("num_success", "u4"), ("comp_len", "u4", (5,)), ("err", "u4")], align=True)

_lib = None


def build():
    subprocess.run(["make", "-C", ORACLE_DIR, "oracle"], check=True, capture_output=True)


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        lib = C.CDLL(LIB_PATH)
        lib.fo_frisys_create.restype = C.c_void_p
        lib.fo_frisys_create.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double,
                                         C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
        lib.fo_frisys_destroy.argtypes = [C.c_void_p]
        lib.fo_vec_digest.restype = C.c_uint64
        lib.fo_vec_digest.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.fo_frisys_iterate.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        lib.fo_frisys_p_doub.restype = C.c_double
        lib.fo_frisys_p_doub.argtypes = [C.c_void_p]
        lib.fo_frisys_hf_en.restype = C.c_double
        lib.fo_frisys_hf_en.argtypes = [C.c_void_p]
        for nm in ("fo_frisys_vec", "fo_frisys_htrial"):
            getattr(lib, nm).restype = C.c_size_t
            getattr(lib, nm).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        lib.fo_frisys_load.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        lib.fo_hb_tensor.restype = C.c_size_t
        lib.fo_hb_tensor.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        lib.fo_set_hb_tensor.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        lib.fo_matrel_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        lib.fo_apply_hbpp_sys.restype = C.c_size_t
        lib.fo_apply_hbpp_sys.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        lib.fo_apply_hbpp_piv.restype = C.c_size_t
        lib.fo_apply_hbpp_piv.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        lib.fo_set_p_doub.argtypes = [C.c_void_p, C.c_double]
        lib.fo_compress_vec.argtypes = [C.c_void_p, C.c_uint32, C.c_double, C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
        lib.fo_vec_add.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        lib.fo_compress_vec_piv.argtypes = [C.c_void_p, C.c_uint32]
        lib.fo_fciqmc_create_ex.restype = C.c_void_p
        lib.fo_fciqmc_create_ex.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                            C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]
        lib.fo_fqranks_create.restype = C.c_void_p
        lib.fo_fqranks_create.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
        lib.fo_fqranks_destroy.argtypes = [C.c_void_p]
        lib.fo_fqranks_iterate.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        lib.fo_fqranks_get.restype = C.c_void_p
        lib.fo_fqranks_get.argtypes = [C.c_void_p, C.c_uint32]
        lib.fo_fqranks_hf_proc.argtypes = [C.c_void_p]
        lib.fo_frisys_create_ex.restype = C.c_void_p
        lib.fo_frisys_create_ex.argtypes = lib.fo_frisys_create.argtypes + [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_double]
        lib.fo_frisys_create_ex2.restype = C.c_void_p
        lib.fo_frisys_create_ex2.argtypes = lib.fo_frisys_create_ex.argtypes + [C.c_void_p, C.c_size_t]
        lib.fo_frifull_create.restype = C.c_void_p
        lib.fo_frifull_create.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_uint32, C.c_uint32, C.c_uint32]
        lib.fo_frifull_destroy.argtypes = [C.c_void_p]
        lib.fo_frifull_iterate.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        lib.fo_frifull_vec.restype = C.c_size_t
        lib.fo_frifull_vec.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        lib.fo_adjust_probs.restype = C.c_double
        lib.fo_adjust_probs.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32), C.c_double, C.c_uint32, C.c_double, C.c_void_p]
        lib.fo_next_draw.restype = C.c_uint32
        lib.fo_next_draw.argtypes = [C.c_void_p]
        lib.fo_piv_comp.restype = C.c_uint32
        lib.fo_piv_comp.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p]
        lib.fo_vec_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.POINTER(C.c_uint32)]
        lib.fo_frisys_restart.argtypes = [C.c_void_p, C.c_uint32, C.c_double, C.c_double, C.c_uint32]
        lib.fo_hash.restype = C.c_uint64
        lib.fo_hash.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        lib.fo_ranks_create.restype = C.c_void_p
        lib.fo_ranks_create.argtypes = [C.c_uint32] + lib.fo_frisys_create.argtypes
        lib.fo_ranks_destroy.argtypes = [C.c_void_p]
        lib.fo_ranks_iterate.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        lib.fo_ranks_compress_piv.argtypes = [C.c_void_p, C.c_uint32]
        lib.fo_ranks_get.restype = C.c_void_p
        lib.fo_ranks_get.argtypes = [C.c_void_p, C.c_uint32]
        lib.fo_ranks_hf_proc.argtypes = [C.c_void_p]
        lib.fo_idx_to_proc.argtypes = [C.c_void_p, C.c_uint64]
        lib.fo_hh_create.restype = C.c_void_p
        lib.fo_hh_create.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32] + [C.c_double] * 7 + [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
        lib.fo_hh_destroy.argtypes = [C.c_void_p]
        lib.fo_hh_iterate.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        lib.fo_hh_vec.restype = C.c_size_t
        lib.fo_hh_vec.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t]
        lib.fo_hh_ref_proc.argtypes = [C.c_void_p]
        lib.fo_hh_neel.restype = C.c_uint64
        lib.fo_hh_neel.argtypes = [C.c_void_p]
        lib.fo_fciqmc_create.restype = C.c_void_p
        lib.fo_fciqmc_create.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
        lib.fo_fciqmc_destroy.argtypes = [C.c_void_p]
        lib.fo_fciqmc_iterate.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        lib.fo_fciqmc_vec.restype = C.c_size_t
        lib.fo_fciqmc_vec.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        lib.fo_fciqmc_load.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_double, C.c_double, C.c_uint32]
        lib.fo_fciqmc_p_doub.restype = C.c_double
        lib.fo_fciqmc_p_doub.argtypes = [C.c_void_p]
        _lib = lib
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def vec_digest(dets, vals) -> int:
    """The digest ref_harness logs per iteration (C loop: a 1e6-element vector takes milliseconds)."""
    d = np.ascontiguousarray(dets, dtype=np.uint64)
    v = np.ascontiguousarray(vals, dtype=np.float64)
    return int(load().fo_vec_digest(_p(d), _p(v), min(d.size, v.size)))


class OracleFrisys:
    """fo::Frisys -- the sequential CPU restatement of frisys_mol (one rank)."""

    def __init__(self, mol, *, epsilon, vec_nonz, mat_nonz, max_dets, target_norm=0.0, initiator=0.0, seed=0, distribution="HB_unnorm",
                 trial=None, ini=None, ham_shift=None, det_space=None):
        self.lib = load()
        self.mol = mol
        irr = np.ascontiguousarray(mol.irreps, dtype=np.uint8)
        hc = np.ascontiguousarray(mol.h_core, dtype=np.float64)
        er = np.ascontiguousarray(mol.eris, dtype=np.float64)
        if trial is None and ini is None and ham_shift is None and det_space is None:
            self.h = self.lib.fo_frisys_create(mol.n_orb, mol.n_elec, _p(irr), _p(hc), _p(er), epsilon, target_norm, initiator,
                                               vec_nonz, mat_nonz, max_dets, seed, 1 if distribution == "HB_unnorm" else 0)
        else:
            td, tv = (np.ascontiguousarray(trial[0], dtype=np.uint64), np.ascontiguousarray(trial[1], dtype=np.float64)) if trial is not None else (np.zeros(1, np.uint64), np.zeros(1))
            idd, iv = (np.ascontiguousarray(ini[0], dtype=np.uint64), np.ascontiguousarray(ini[1], dtype=np.float64)) if ini is not None else (np.zeros(1, np.uint64), np.zeros(1))
            sp = np.ascontiguousarray(det_space, dtype=np.uint64) if det_space is not None else np.zeros(1, np.uint64)
            self.h = self.lib.fo_frisys_create_ex2(mol.n_orb, mol.n_elec, _p(irr), _p(hc), _p(er), epsilon, target_norm, initiator,
                                                   vec_nonz, mat_nonz, max_dets, seed, 1 if distribution == "HB_unnorm" else 0,
                                                   _p(td), _p(tv), td.size if trial is not None else 0, _p(idd), _p(iv), idd.size if ini is not None else 0,
                                                   0 if ham_shift is None else 1, 0.0 if ham_shift is None else float(ham_shift),
                                                   _p(sp), sp.size if det_space is not None else 0)
        self.max_dets = max_dets

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.fo_frisys_destroy(self.h)
            self.h = None

    def iterate(self, n):
        logs = np.zeros(n, dtype=LOG_DTYPE)
        self.lib.fo_frisys_iterate(self.h, n, _p(logs))
        return logs

    @property
    def p_doub(self):
        return self.lib.fo_frisys_p_doub(self.h)

    @property
    def hf_energy(self):
        return self.lib.fo_frisys_hf_en(self.h)

    def vector(self):
        n = self.lib.fo_frisys_vec(self.h, None, None, 0)
        d = np.zeros(max(n, 1), dtype=np.uint64)
        v = np.zeros(max(n, 1))
        self.lib.fo_frisys_vec(self.h, _p(d), _p(v), d.size)
        return d[:n], v[:n]

    def htrial(self):
        n = self.lib.fo_frisys_htrial(self.h, None, None, 0)
        d = np.zeros(max(n, 1), dtype=np.uint64)
        v = np.zeros(max(n, 1))
        self.lib.fo_frisys_htrial(self.h, _p(d), _p(v), d.size)
        return d[:n], v[:n]

    def vec_load(self, dets, vals):
        d = np.ascontiguousarray(dets, dtype=np.uint64)
        v = np.ascontiguousarray(vals, dtype=np.float64)
        self.lib.fo_frisys_load(self.h, _p(d), _p(v), d.size)

    def vec_add(self, dets, vals, ini):
        d = np.ascontiguousarray(dets, dtype=np.uint64)
        v = np.ascontiguousarray(vals, dtype=np.float64)
        f = np.ascontiguousarray(ini, dtype=np.uint8)
        self.lib.fo_vec_add(self.h, _p(d), _p(v), _p(f), d.size)

    def vec_info(self):
        a, b, c = C.c_uint32(), C.c_int32(), C.c_uint32()
        self.lib.fo_vec_info(self.h, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def hb_tensor(self, which):
        out = np.zeros(self.mol.n_orb ** 2 + 1)
        n = self.lib.fo_hb_tensor(self.h, which, _p(out), out.size)
        return out[:n].copy()

    def set_hb_tensor(self, which, arr):
        a = np.ascontiguousarray(arr, dtype=np.float64)
        self.lib.fo_set_hb_tensor(self.h, which, _p(a), a.size)

    def set_p_doub(self, p):
        self.lib.fo_set_p_doub(self.h, p)

    def matrel(self, kind, dets, orbs=None):
        d = np.ascontiguousarray(dets, dtype=np.uint64)
        o = np.ascontiguousarray(orbs, dtype=np.uint8) if orbs is not None else None
        out = np.zeros(d.size)
        sg = np.zeros(d.size, dtype=np.int32)
        self.lib.fo_matrel_batch(self.h, kind, _p(d), _p(o), d.size, _p(out), _p(sg))
        return out, sg

    def apply_hbpp_sys(self, n_samp, rn, unit_matrel=False):
        rn = np.ascontiguousarray(rn, dtype=np.float64)
        cap = 4 * n_samp + 4096
        pos = np.zeros(cap, dtype=np.uint32)
        orbs = np.zeros((cap, 4), dtype=np.uint8)
        vals = np.zeros(cap)
        n = self.lib.fo_apply_hbpp_sys(self.h, n_samp, _p(rn), int(unit_matrel), _p(pos), _p(orbs), _p(vals), cap)
        return pos[:n].copy(), orbs[:n].copy(), vals[:n].copy()

    def set_spin_parity(self, sp):
        self.lib.fo_set_spin_parity.argtypes = [C.c_int]
        self.lib.fo_set_spin_parity(int(sp))

    def h_offdiag_list(self, dets, vals):
        d = np.ascontiguousarray(dets, dtype=np.uint64); v = np.ascontiguousarray(vals, dtype=np.float64)
        cap = d.size * (self.mol.n_orb ** 2 * self.mol.n_elec ** 2 + 2) + 64
        od = np.zeros(cap, dtype=np.uint64); ov = np.zeros(cap)
        self.lib.fo_h_offdiag_list.restype = C.c_size_t
        self.lib.fo_h_offdiag_list.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]
        n = self.lib.fo_h_offdiag_list(self.h, _p(d), _p(v), d.size, _p(od), _p(ov), cap)
        assert n != 2 ** 64 - 1
        return od[:n].copy(), ov[:n].copy()

    def apply_hbpp_piv(self, n_samp, unit_matrel=False, cap=None):
        """apply_HBPP_piv on the stored vector with the handle's generator (restart(seed) first): positions, orbitals, values, stage lengths."""
        cap = cap or (max(n_samp, self.vec_info()[0]) * 2 + 64)
        pos = np.zeros(cap, dtype=np.uint32)
        orbs = np.zeros((cap, 4), dtype=np.uint8)
        vals = np.zeros(cap)
        st = np.zeros(5, dtype=np.uint64)
        n = self.lib.fo_apply_hbpp_piv(self.h, n_samp, int(unit_matrel), _p(pos), _p(orbs), _p(vals), cap, _p(st))
        return pos[:n].copy(), orbs[:n].copy(), vals[:n].copy(), st

    def compress_vec(self, n_samp, rn):
        nk = C.c_uint32()
        gn = C.c_double()
        self.lib.fo_compress_vec(self.h, n_samp, rn, C.byref(nk), C.byref(gn))
        return nk.value, gn.value

    def restart(self, seed, en_shift=0.0, last_one_norm=0.0, iterat=0):
        self.lib.fo_frisys_restart(self.h, seed, en_shift, last_one_norm, iterat)

    def compress_vec_piv(self, n_samp):
        self.lib.fo_compress_vec_piv(self.h, n_samp)

    def next_draw(self):
        return self.lib.fo_next_draw(self.h)


class OracleFull:
    """fo::Frifull -- the sequential CPU restatement of frifull_mol (one rank)."""

    def __init__(self, mol, *, epsilon, vec_nonz, max_dets, target_norm=0.0, seed=0):
        self.lib = load()
        irr = np.ascontiguousarray(mol.irreps, dtype=np.uint8)
        hc = np.ascontiguousarray(mol.h_core, dtype=np.float64)
        er = np.ascontiguousarray(mol.eris, dtype=np.float64)
        self.h = self.lib.fo_frifull_create(mol.n_orb, mol.n_elec, _p(irr), _p(hc), _p(er), epsilon, target_norm, vec_nonz, max_dets, seed)

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.fo_frifull_destroy(self.h)
            self.h = None

    def iterate(self, n):
        logs = np.zeros(n, dtype=LOG_DTYPE)
        self.lib.fo_frifull_iterate(self.h, n, _p(logs))
        return logs

    def vector(self):
        n = self.lib.fo_frifull_vec(self.h, None, None, 0)
        d = np.zeros(max(n, 1), dtype=np.uint64)
        v = np.zeros(max(n, 1))
        self.lib.fo_frifull_vec(self.h, _p(d), _p(v), d.size)
        return d[:n], v[:n]


class OracleRanks:
    """P in-process ranks of fo::Frisys sharing one communicator -- the reference under `mpiexec -n P`
    (hash-sharded determinants, all-to-all spawns, rank-ordered sum_mpi)."""

    def __init__(self, n_ranks, mol, *, epsilon, vec_nonz, mat_nonz, max_dets, target_norm=0.0, initiator=0.0, seed=0, distribution="HB_unnorm", det_space=None):
        self.lib = load()
        self.n_ranks = n_ranks
        irr = np.ascontiguousarray(mol.irreps, dtype=np.uint8)
        hc = np.ascontiguousarray(mol.h_core, dtype=np.float64)
        er = np.ascontiguousarray(mol.eris, dtype=np.float64)
        if det_space is not None:
            sp = np.ascontiguousarray(det_space, dtype=np.uint64)
            self.lib.fo_ranks_set_det_space.argtypes = [C.c_void_p, C.c_size_t]
            self.lib.fo_ranks_set_det_space(_p(sp), sp.size)
        self.h = self.lib.fo_ranks_create(n_ranks, mol.n_orb, mol.n_elec, _p(irr), _p(hc), _p(er), epsilon, target_norm, initiator,
                                          vec_nonz, mat_nonz, max_dets, seed, 1 if distribution == "HB_unnorm" else 0)
        if not self.h:
            raise RuntimeError("oracle ranks failed to set up")

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.fo_ranks_destroy(self.h)
            self.h = None

    def iterate(self, n):
        """-> logs[n_ranks, n]"""
        logs = np.zeros((self.n_ranks, n), dtype=LOG_DTYPE)
        if self.lib.fo_ranks_iterate(self.h, n, _p(logs)):
            raise RuntimeError("oracle ranks failed")
        return logs

    def restart(self, seed):
        """every rank's generator re-seeded (and shift / iteration counters reset), like FriEngine.restart on every rank"""
        for r in range(self.n_ranks):
            self.lib.fo_frisys_restart(self._rank(r), seed, 0.0, 0.0, 0)

    def apply_hbpp_piv(self, n_samp, rank, cap):
        """apply_HBPP_piv over the ranks (collective); this rank's samples.  NOTE: every call advances every rank's generator."""
        self.lib.fo_ranks_apply_hbpp_piv.restype = C.c_size_t
        self.lib.fo_ranks_apply_hbpp_piv.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        pos = np.zeros(cap, dtype=np.uint32); orbs = np.zeros((cap, 4), dtype=np.uint8); vals = np.zeros(cap); st = np.zeros(5, dtype=np.uint64)
        n = self.lib.fo_ranks_apply_hbpp_piv(self.h, n_samp, rank, _p(pos), _p(orbs), _p(vals), cap, _p(st))
        if n == 2 ** 64 - 1:
            raise RuntimeError("oracle apply_hbpp_piv over ranks failed")
        return pos[:n].copy(), orbs[:n].copy(), vals[:n].copy(), st

    def compress_piv(self, n_samp):
        if self.lib.fo_ranks_compress_piv(self.h, n_samp):
            raise RuntimeError("oracle ranks: pivotal compression failed")

    @property
    def hf_proc(self):
        return self.lib.fo_ranks_hf_proc(self.h)

    def _rank(self, r):
        return self.lib.fo_ranks_get(self.h, r)

    def vector(self, r):
        h = self._rank(r)
        n = self.lib.fo_frisys_vec(h, None, None, 0)
        d = np.zeros(max(n, 1), dtype=np.uint64)
        v = np.zeros(max(n, 1))
        self.lib.fo_frisys_vec(h, _p(d), _p(v), d.size)
        return d[:n], v[:n]

    def htrial(self, r=0):
        h = self._rank(r)
        n = self.lib.fo_frisys_htrial(h, None, None, 0)
        d = np.zeros(max(n, 1), dtype=np.uint64)
        v = np.zeros(max(n, 1))
        self.lib.fo_frisys_htrial(h, _p(d), _p(v), d.size)
        return d[:n], v[:n]

    def p_doub(self, r=0):
        return self.lib.fo_frisys_p_doub(self._rank(r))

    def idx_to_proc(self, det):
        return self.lib.fo_idx_to_proc(self._rank(0), int(det))


class OracleHH:
    """fo::FrisysHH -- the CPU restatement of frisys_hh (1-D Hubbard-Holstein), on n_ranks in-process ranks."""

    def __init__(self, *, n_elec, n_sites, eps, U, omega, g, gs_energy, vec_nonz, max_dets, target_norm=0.0, initiator=0.0, seed=0, n_ranks=1, full=False):
        self.lib = load()
        self.n_ranks = n_ranks
        self.h = self.lib.fo_hh_create(n_ranks, n_elec, n_sites, eps, U, omega, g, gs_energy, target_norm, initiator, vec_nonz, max_dets, seed, int(full))
        if not self.h:
            raise RuntimeError("oracle HH setup failed")

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.fo_hh_destroy(self.h)
            self.h = None

    def iterate(self, n):
        logs = np.zeros((self.n_ranks, n), dtype=LOG_DTYPE)
        if self.lib.fo_hh_iterate(self.h, n, _p(logs)):
            raise RuntimeError("oracle HH failed")
        return logs if self.n_ranks > 1 else logs[0]

    def vector(self, rank=0):
        n = self.lib.fo_hh_vec(self.h, rank, None, None, 0)
        d = np.zeros(max(n, 1), dtype=np.uint64)
        v = np.zeros(max(n, 1))
        self.lib.fo_hh_vec(self.h, rank, _p(d), _p(v), d.size)
        return d[:n], v[:n]

    @property
    def ref_proc(self):
        return self.lib.fo_hh_ref_proc(self.h)

    @property
    def neel(self):
        return int(self.lib.fo_hh_neel(self.h))


FQLOG_DTYPE = np.dtype([("numer", "f8"), ("denom", "f8"), ("shift", "f8"), ("norm", "f8"), ("n_nonz", "i4"), ("n_ini", "u4"), ("curr_size", "u4"),
                        ("n_spawn", "u4")], align=True)


def _fq_create_vecs(lib, mol, epsilon, target_walkers, initiator, max_dets, seed, flags, vec_nonz, mat_nonz, initiator_f, target_norm, trial, ini, truncate_ini):
    """fo_fq_create_vecs: fo::Fciqmc in any of its three modes with --trial_vec / --ini_vec; truncate_ini: fciqmc_mol's reader fills an int array."""
    lib.fo_fq_create_vecs.restype = C.c_void_p
    lib.fo_fq_create_vecs.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                      C.c_uint32, C.c_uint32, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]
    irr = np.ascontiguousarray(mol.irreps, dtype=np.uint8)
    hc = np.ascontiguousarray(mol.h_core, dtype=np.float64)
    er = np.ascontiguousarray(mol.eris, dtype=np.float64)
    td, tv = (np.ascontiguousarray(trial[0], dtype=np.uint64), np.ascontiguousarray(trial[1], dtype=np.float64)) if trial is not None else (np.zeros(1, np.uint64), np.zeros(1))
    idd, iv = (np.ascontiguousarray(ini[0], dtype=np.uint64), np.ascontiguousarray(ini[1], dtype=np.float64)) if ini is not None else (np.zeros(1, np.uint64), np.zeros(1))
    if truncate_ini:
        iv = np.trunc(iv)
    h = lib.fo_fq_create_vecs(mol.n_orb, mol.n_elec, _p(irr), _p(hc), _p(er), epsilon, target_walkers, initiator, max_dets, seed, flags, vec_nonz, mat_nonz, initiator_f, target_norm,
                              _p(td), _p(tv), td.size if trial is not None else 0, _p(idd), _p(iv), idd.size if ini is not None else 0)
    if not h:
        lib.fo_last_error.restype = C.c_char_p
        raise RuntimeError(lib.fo_last_error().decode())
    return h


class OracleFciqmc:
    """fo::Fciqmc -- CPU restatement of fciqmc_mol (near-uniform generator, one rank).  counter_rng=False consumes the
    reference's sequential mt19937 stream (pinned against the reference loop); counter_rng=True uses the counter-based
    stream the GPU replays."""

    def __init__(self, mol, *, epsilon, target_walkers, max_dets, initiator=0, seed=0, counter_rng=False, distribution="NU", trial=None, ini=None, fp=False):
        """fp=True: fciqmc_fp_mol (FRIES_bin/fciqmc_fp_mol.cpp), real-valued walkers."""
        self.lib = load()
        if distribution not in ("NU", "HB"):
            raise RuntimeError('"dist_str" argument must be either "NU" or "HB"')
        irr = np.ascontiguousarray(mol.irreps, dtype=np.uint8)
        hc = np.ascontiguousarray(mol.h_core, dtype=np.float64)
        er = np.ascontiguousarray(mol.eris, dtype=np.float64)
        flags = int(counter_rng) | (2 if distribution == "HB" else 0) | (4 if fp else 0)
        if trial is None and ini is None:
            self.h = self.lib.fo_fciqmc_create(mol.n_orb, mol.n_elec, _p(irr), _p(hc), _p(er), epsilon, target_walkers, initiator, max_dets, seed, flags)
        else:
            self.h = _fq_create_vecs(self.lib, mol, epsilon, target_walkers, initiator, max_dets, seed, flags, 0, 0, 0.0, 0.0, trial, ini, truncate_ini=not fp)

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.fo_fciqmc_destroy(self.h)
            self.h = None

    def iterate(self, n):
        logs = np.zeros(n, dtype=FQLOG_DTYPE)
        if self.lib.fo_fciqmc_iterate(self.h, n, _p(logs)):
            raise RuntimeError("oracle fciqmc failed")
        return logs

    def vector(self):
        n = self.lib.fo_fciqmc_vec(self.h, None, None, 0)
        d = np.zeros(max(n, 1), dtype=np.uint64)
        v = np.zeros(max(n, 1))
        self.lib.fo_fciqmc_vec(self.h, _p(d), _p(v), d.size)
        return d[:n], v[:n]

    @property
    def p_doub(self):
        return self.lib.fo_fciqmc_p_doub(self.h)

    def load(self, dets, vals, en_shift=0.0, last_norm=0.0, iterat=0):
        """Replaces the walkers (positions 0..n-1) and the shift / last walker number / iteration count."""
        d = np.ascontiguousarray(dets, dtype=np.uint64)
        v = np.ascontiguousarray(vals, dtype=np.float64)
        self.lib.fo_fciqmc_load(self.h, _p(d), _p(v), d.size, en_shift, last_norm, iterat)


class OracleMulti(OracleFciqmc):
    """fo::Fciqmc in its frimulti_mol mode (FRIES_bin/frimulti_mol.cpp, --distribution HB): counter_rng=False is the reference's mt19937
    stream (pinned against the reference loop), counter_rng=True the counter-based stream the GPU replays."""

    def __init__(self, mol, *, epsilon, vec_nonz, mat_nonz, max_dets, initiator=0.0, target_norm=0.0, seed=0, counter_rng=False, trial=None, ini=None):
        self.lib = load()
        self.lib.fo_frimulti_nkept.restype = C.c_uint32
        self.lib.fo_frimulti_nkept.argtypes = [C.c_void_p]
        if trial is not None or ini is not None:
            self.h = _fq_create_vecs(self.lib, mol, epsilon, 0, 0, max_dets, seed, int(counter_rng) | 8, vec_nonz, mat_nonz, float(initiator), float(target_norm), trial, ini, truncate_ini=False)
            return
        irr = np.ascontiguousarray(mol.irreps, dtype=np.uint8)
        hc = np.ascontiguousarray(mol.h_core, dtype=np.float64)
        er = np.ascontiguousarray(mol.eris, dtype=np.float64)
        self.lib.fo_frimulti_create.restype = C.c_void_p
        self.lib.fo_frimulti_create.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                                C.c_double, C.c_double, C.c_int]
        self.lib.fo_frimulti_nkept.restype = C.c_uint32
        self.lib.fo_frimulti_nkept.argtypes = [C.c_void_p]
        self.h = self.lib.fo_frimulti_create(mol.n_orb, mol.n_elec, _p(irr), _p(hc), _p(er), epsilon, vec_nonz, mat_nonz, max_dets, seed, float(initiator), float(target_norm),
                                             int(counter_rng))

    @property
    def nkept(self):
        return self.lib.fo_frimulti_nkept(self.h)


def piv_comp(vals, compress_size, seed):
    """fo::piv_comp_parallel (one rank) on a copy of vals: (new values, delete flags, the generator's next draw)."""
    lib = load()
    v = np.ascontiguousarray(vals, dtype=np.float64).copy()
    fl = np.zeros(v.size, dtype=np.uint8)
    nxt = lib.fo_piv_comp(v.ctypes.data, v.size, compress_size, seed, fl.ctypes.data)
    return v, fl, nxt


def adjust_probs(vals, n_loc, exp_loc, n_tot, tot_norm):
    """fo::adjust_probs on a copy of vals with nothing preserved: (new values, n_loc, new norm, pinned flags)."""
    lib = load()
    v = np.ascontiguousarray(vals, dtype=np.float64).copy()
    fl = np.zeros(v.size, dtype=np.uint8)
    nl = C.c_uint32(n_loc)
    nn = lib.fo_adjust_probs(v.ctypes.data, v.size, C.byref(nl), exp_loc, n_tot, tot_norm, fl.ctypes.data)
    return v, nl.value, nn, fl


class OracleFciqmcRanks:
    """P in-process ranks of fo::Fciqmc sharing one communicator -- fciqmc_mol under `mpiexec -n P` (a generator per rank, seeded
    seed + rank in mt mode; the counter stream does not depend on the rank)."""

    def __init__(self, n_ranks, mol, *, epsilon, target_walkers, max_dets, initiator=0, seed=0, counter_rng=False, distribution="NU", fp=False):
        self.lib = load()
        self.n_ranks = n_ranks
        irr = np.ascontiguousarray(mol.irreps, dtype=np.uint8)
        hc = np.ascontiguousarray(mol.h_core, dtype=np.float64)
        er = np.ascontiguousarray(mol.eris, dtype=np.float64)
        self.h = self.lib.fo_fqranks_create(n_ranks, mol.n_orb, mol.n_elec, _p(irr), _p(hc), _p(er), epsilon, target_walkers, initiator, max_dets, seed,
                                            int(counter_rng) | (2 if distribution == "HB" else 0) | (4 if fp else 0))
        if not self.h:
            raise RuntimeError("oracle fciqmc ranks: setup failed")

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.fo_fqranks_destroy(self.h)
            self.h = None

    def iterate(self, n):
        """-> logs[n_ranks, n]"""
        logs = np.zeros((self.n_ranks, n), dtype=FQLOG_DTYPE)
        if self.lib.fo_fqranks_iterate(self.h, n, _p(logs)):
            raise RuntimeError("oracle fciqmc ranks failed")
        return logs

    @property
    def hf_proc(self):
        return self.lib.fo_fqranks_hf_proc(self.h)

    def vector(self, r):
        h = self.lib.fo_fqranks_get(self.h, r)
        n = self.lib.fo_fciqmc_vec(h, None, None, 0)
        d = np.zeros(max(n, 1), dtype=np.uint64)
        v = np.zeros(max(n, 1))
        self.lib.fo_fciqmc_vec(h, _p(d), _p(v), d.size)
        return d[:n], v[:n]


class OracleMultiRanks(OracleFciqmcRanks):
    """P in-process ranks of fo::Fciqmc in its frimulti_mol mode -- frimulti_mol under `mpiexec -n P`."""

    def __init__(self, n_ranks, mol, *, epsilon, vec_nonz, mat_nonz, max_dets, initiator=0.0, target_norm=0.0, seed=0, counter_rng=False):
        self.lib = load()
        self.n_ranks = n_ranks
        irr = np.ascontiguousarray(mol.irreps, dtype=np.uint8)
        hc = np.ascontiguousarray(mol.h_core, dtype=np.float64)
        er = np.ascontiguousarray(mol.eris, dtype=np.float64)
        self.lib.fo_multiranks_create.restype = C.c_void_p
        self.lib.fo_multiranks_create.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_uint32, C.c_uint32, C.c_uint32,
                                                  C.c_uint32, C.c_double, C.c_double, C.c_int]
        self.h = self.lib.fo_multiranks_create(n_ranks, mol.n_orb, mol.n_elec, _p(irr), _p(hc), _p(er), epsilon, vec_nonz, mat_nonz, max_dets, seed, float(initiator),
                                               float(target_norm), int(counter_rng))
        if not self.h:
            raise RuntimeError("oracle frimulti ranks: setup failed")
