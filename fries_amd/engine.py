"""Host-side mirror of the reference's driver surface over the C ABI (include/fries_hip.h).

``FriEngine`` plays the role of frisys_mol's ``main`` (FRIES_bin/frisys_mol.cpp): it owns a
context, hands it the parsed FCIDUMP, and runs the iteration loop on the MI355X.  There is no
CPU path: importing works anywhere, creating an engine without libfries_hip.so or without a
HIP device raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FRIES_LIB") or os.path.join(_HERE, "libfries_hip.so")      # FRIES_LIB: another build of the same library (A/B timing of kernel variants)

# every symbol include/fries_hip.h declares
EXPORTS = [
    "fries_last_error", "fries_device_count", "fries_ctx_create", "fries_ctx_destroy", "fries_set_molecule",
    "fries_get_hb_tensor", "fries_set_hb_tensor", "fries_hf_energy", "fries_matrel_batch", "fries_frisys_setup",
    "fries_frisys_iterate", "fries_p_doub", "fries_kernel_launches", "fries_vec_info", "fries_vec_download",
    "fries_vec_add", "fries_vec_load", "fries_htrial_download", "fries_apply_hbpp_sys", "fries_apply_hbpp_piv", "fries_rng_set_state", "fries_rng_get_state", "fries_compress_vec",
    "fries_test_teeth", "fries_test_seqsum", "fries_frisys_restart", "fries_prof_enable", "fries_prof_count", "fries_prof_get", "fries_counters",
    "fries_set_comm", "fries_stream", "fries_idx_to_proc", "fries_hh_setup", "fries_hh_iterate", "fries_get_scramblers", "fries_fciqmc_setup", "fries_fciqmc_iterate", "fries_frimulti_setup", "fries_frimulti_iterate",
    "fries_compress_vec_piv", "fries_next_draw", "fries_test_piv_adjust", "fries_frifull_setup", "fries_frifull_iterate",
    "fries_rccl_unique_id", "fries_rccl_create", "fries_local_group_create", "fries_local_group_destroy", "fries_local_create", "fries_transport_comm", "fries_transport_counts", "fries_transport_destroy",
    "fries_set_proc_scrambler", "fries_tie_margins", "fries_measure_copy_bandwidth", "fries_piv_stats", "fries_set_trial_vector", "fries_set_initial_vector", "fries_set_ham_shift", "fries_vec_add_to", "fries_death_clone", "fries_dots", "fries_find_preserve", "fries_sys_comp",
    "fries_vec_column_download", "fries_vec_column_upload", "fries_vec_column_zero", "fries_vec_diag_download", "fries_vec_dot_list", "fries_vec_add_vecs", "fries_set_det_space", "fries_vec_set_dense", "fries_dense_sizes", "fries_hostcomm_create", "fries_set_vec_scrambler", "fries_hh_comp_sub", "fries_hh_ref_ovlp", "fries_set_spin_parity", "fries_h_offdiag_list",
]


class FrisysParams(C.Structure):
    _fields_ = [("epsilon", C.c_double), ("target_norm", C.c_double), ("initiator", C.c_double),
                ("vec_nonz", C.c_uint32), ("mat_nonz", C.c_uint32), ("max_dets", C.c_uint32),
                ("seed", C.c_uint32), ("hb_unnorm", C.c_int32)]


class FrifullParams(C.Structure):
    _fields_ = [("epsilon", C.c_double), ("target_norm", C.c_double), ("vec_nonz", C.c_uint32), ("max_dets", C.c_uint32),
                ("seed", C.c_uint32), ("spawn_cap", C.c_uint32)]


class HHParams(C.Structure):
    """struct fries_hh_params"""
    _fields_ = [("n_elec", C.c_uint32), ("n_sites", C.c_uint32), ("eps", C.c_double), ("U", C.c_double), ("omega", C.c_double), ("g", C.c_double),
                ("gs_energy", C.c_double), ("target_norm", C.c_double), ("initiator", C.c_double), ("vec_nonz", C.c_uint32), ("max_dets", C.c_uint32),
                ("seed", C.c_uint32), ("full", C.c_uint32)]


class FrimultiParams(C.Structure):
    """struct fries_frimulti_params"""
    _fields_ = [("epsilon", C.c_double), ("target_norm", C.c_double), ("initiator", C.c_double), ("vec_nonz", C.c_uint32), ("mat_nonz", C.c_uint32),
                ("max_dets", C.c_uint32), ("seed", C.c_uint32)]


class FciqmcParams(C.Structure):
    """struct fries_fciqmc_params"""
    _fields_ = [("epsilon", C.c_double), ("target_walkers", C.c_uint32), ("initiator", C.c_uint32), ("max_dets", C.c_uint32), ("seed", C.c_uint32),
                ("heat_bath", C.c_int32), ("real_walkers", C.c_int32)]


FCIQMC_LOG_DTYPE = np.dtype([("numer", "f8"), ("denom", "f8"), ("shift", "f8"), ("norm", "f8"), ("n_nonz", "i4"), ("n_ini", "u4"), ("curr_size", "u4"),
                             ("n_spawn", "u4"), ("n_attempts", "u4"), ("err", "u4")], align=True)


class IterLog(C.Structure):
    _fields_ = [("numer", C.c_double), ("denom", C.c_double), ("shift", C.c_double), ("norm", C.c_double),
                ("nkept", C.c_uint32), ("n_nonz", C.c_int32), ("curr_size", C.c_uint32), ("num_success", C.c_uint32),
                ("comp_len", C.c_uint32 * 5), ("err", C.c_uint32)]


ITERLOG_DTYPE = np.dtype([("numer", "f8"), ("denom", "f8"), ("shift", "f8"), ("norm", "f8"), ("nkept", "u4"), ("n_nonz", "i4"),
                          ("curr_size", "u4"), ("num_success", "u4"), ("comp_len", "u4", (5,)), ("err", "u4")], align=True)

_lib = None


def load_library() -> C.CDLL:
    """Loads libfries_hip.so; raises if it has not been built (no fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
                           "The FRI engine has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    lib.fries_last_error.restype = C.c_char_p
    lib.fries_hf_energy.restype = C.c_double
    lib.fries_p_doub.restype = C.c_double
    lib.fries_kernel_launches.restype = C.c_uint64
    lib.fries_ctx_destroy.restype = None
    for name in ("fries_hf_energy", "fries_p_doub", "fries_kernel_launches", "fries_ctx_destroy"):
        getattr(lib, name).argtypes = [C.c_void_p]
    lib.fries_ctx_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    lib.fries_set_molecule.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.fries_get_hb_tensor.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.fries_set_hb_tensor.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    lib.fries_matrel_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    lib.fries_frisys_setup.argtypes = [C.c_void_p, C.POINTER(FrisysParams)]
    lib.fries_frisys_iterate.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    lib.fries_vec_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.POINTER(C.c_uint32)]
    lib.fries_vec_download.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.fries_htrial_download.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.fries_vec_add.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    lib.fries_vec_column_download.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.fries_vec_column_upload.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    lib.fries_vec_column_zero.argtypes = [C.c_void_p, C.c_int]
    lib.fries_vec_add_vecs.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double]
    lib.fries_set_det_space.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.fries_vec_diag_download.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.fries_vec_dot_list.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_double)]
    lib.fries_vec_load.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    lib.fries_rng_set_state.argtypes = [C.c_void_p, C.c_char_p]
    lib.fries_rng_get_state.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.fries_apply_hbpp_piv.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    lib.fries_apply_hbpp_sys.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                         C.POINTER(C.c_size_t), C.c_void_p]
    lib.fries_compress_vec.argtypes = [C.c_void_p, C.c_uint32, C.c_double, C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
    lib.fries_compress_vec_piv.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
    lib.fries_test_piv_adjust.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_double, C.c_uint32, C.c_double, C.POINTER(C.c_double), C.c_void_p]
    lib.fries_measure_copy_bandwidth.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_double)]
    lib.fries_piv_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.fries_set_trial_vector.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    lib.fries_set_initial_vector.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    lib.fries_set_ham_shift.argtypes = [C.c_void_p, C.c_double]
    lib.fries_next_draw.restype = C.c_uint32
    lib.fries_next_draw.argtypes = [C.c_void_p]
    lib.fries_test_teeth.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    lib.fries_test_seqsum.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_double, C.c_void_p, C.POINTER(C.c_double),
                                      C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.fries_frisys_restart.argtypes = [C.c_void_p, C.c_uint32, C.c_double, C.c_double, C.c_uint32]
    lib.fries_prof_enable.argtypes = [C.c_void_p, C.c_int]
    lib.fries_prof_count.argtypes = [C.c_void_p]
    lib.fries_prof_get.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    lib.fries_counters.argtypes = [C.c_void_p] + [C.POINTER(C.c_uint64)] * 5
    lib.fries_get_scramblers.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    lib.fries_frimulti_setup.argtypes = [C.c_void_p, C.POINTER(FrimultiParams)]
    lib.fries_frimulti_iterate.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    lib.fries_fciqmc_setup.argtypes = [C.c_void_p, C.POINTER(FciqmcParams)]
    lib.fries_fciqmc_iterate.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    lib.fries_frifull_setup.argtypes = [C.c_void_p, C.POINTER(FrifullParams)]
    lib.fries_frifull_iterate.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    lib.fries_hh_setup.argtypes = [C.c_void_p, C.POINTER(HHParams)]
    lib.fries_hh_iterate.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    lib.fries_set_comm.argtypes = [C.c_void_p, C.c_void_p]
    lib.fries_tie_margins.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.fries_rccl_unique_id.argtypes = [C.c_void_p]
    lib.fries_rccl_create.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint64]
    lib.fries_local_group_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_uint64]
    lib.fries_local_group_destroy.argtypes = [C.c_void_p]
    lib.fries_local_group_destroy.restype = None
    lib.fries_local_create.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int]
    lib.fries_transport_comm.argtypes = [C.c_void_p, C.c_void_p]
    lib.fries_transport_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.fries_transport_destroy.argtypes = [C.c_void_p]
    lib.fries_transport_destroy.restype = None
    lib.fries_stream.restype = C.c_void_p
    lib.fries_stream.argtypes = [C.c_void_p]
    lib.fries_idx_to_proc.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    _lib = lib
    return lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class FriEngine:
    """One MI355X running the FRI iteration for one molecular Hamiltonian -- or, with ``comm`` (fries_amd.comm.TorchComm),
    one rank of a hash-sharded run: vec_nonz / mat_nonz / target_norm are then global, max_dets is per rank."""

    def __init__(self, mol, device: int = 0, comm=None):
        self.lib = load_library()
        if self.lib.fries_device_count() <= 0:
            raise RuntimeError("no HIP device visible: the FRI engine runs on MI355X only (no CPU fallback)")
        self.h = C.c_void_p()
        self._ck(self.lib.fries_ctx_create(C.byref(self.h), device))
        self.mol = mol
        if mol is not None:         # mol=None: a lattice-model run (setup_hh) needs no integrals
            irr = np.ascontiguousarray(mol.irreps, dtype=np.uint8)
            hc = np.ascontiguousarray(mol.h_core, dtype=np.float64)
            er = np.ascontiguousarray(mol.eris, dtype=np.float64)
            self._ck(self.lib.fries_set_molecule(self.h, mol.n_orb, mol.n_elec, _ptr(irr), _ptr(hc), _ptr(er)))
        self.max_dets = 0
        self.comm = comm
        if comm is not None:
            self._ck(self.lib.fries_set_comm(self.h, C.byref(comm.struct)))

    def _ck(self, rc):
        if rc != 0:
            raise RuntimeError(self.lib.fries_last_error().decode())

    def close(self):
        if self.h:
            if self.comm is not None:
                self.comm.release()
            self.lib.fries_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- molecule
    @property
    def hf_energy(self) -> float:
        return self.lib.fries_hf_energy(self.h)

    @property
    def p_doub(self) -> float:
        return self.lib.fries_p_doub(self.h)

    @property
    def kernel_launches(self) -> int:
        return int(self.lib.fries_kernel_launches(self.h))

    def hb_tensor(self, which: int) -> np.ndarray:
        n = self.mol.n_orb
        out = np.zeros(max(n * n, 1))
        ln = C.c_size_t()
        self._ck(self.lib.fries_get_hb_tensor(self.h, which, _ptr(out), out.size, C.byref(ln)))
        return out[:ln.value].copy()

    def set_hb_tensor(self, which: int, arr) -> None:
        a = np.ascontiguousarray(arr, dtype=np.float64)
        self._ck(self.lib.fries_set_hb_tensor(self.h, which, _ptr(a), a.size))

    def matrel(self, kind: int, dets, orbs=None):
        d = np.ascontiguousarray(dets, dtype=np.uint64)
        o = np.ascontiguousarray(orbs, dtype=np.uint8) if orbs is not None else None
        out = np.zeros(d.size)
        sg = np.zeros(d.size, dtype=np.int32)
        self._ck(self.lib.fries_matrel_batch(self.h, kind, _ptr(d), _ptr(o), d.size, _ptr(out), _ptr(sg)))
        return out, sg

    # ---- frisys_mol
    def setup(self, *, epsilon, vec_nonz, mat_nonz, max_dets, target_norm=0.0, initiator=0.0, seed=0, distribution="HB_unnorm",
              trial=None, ini=None, ham_shift=None, det_space=None):
        """frisys_mol setup.  trial / ini: (dets, vals) of --trial_vec / --ini_vec; ham_shift: the diagonal offset that replaces
        the HF energy (--ham_shift minus the core energy); det_space: the determinants of --det_space (semi-stochastic dense space)."""
        if distribution not in ("HB", "HB_unnorm"):
            raise RuntimeError('"dist_str" argument must be either "HB" or "HB_unnorm"')
        for pair, fn in ((trial, self.lib.fries_set_trial_vector), (ini, self.lib.fries_set_initial_vector)):
            if pair is not None:
                d = np.ascontiguousarray(pair[0], dtype=np.uint64)
                v = np.ascontiguousarray(pair[1], dtype=np.float64)
                self._ck(fn(self.h, _ptr(d), _ptr(v), min(d.size, v.size)))
        if ham_shift is not None:
            self._ck(self.lib.fries_set_ham_shift(self.h, float(ham_shift)))
        if det_space is not None:
            d = np.ascontiguousarray(det_space, dtype=np.uint64)
            self._ck(self.lib.fries_set_det_space(self.h, _ptr(d), d.size))
        if self.comm is not None and self.comm.big_bytes < 16 * (mat_nonz + 4096):
            raise RuntimeError("TorchComm was sized for a smaller mat_nonz")
        p = FrisysParams(epsilon, target_norm, initiator, vec_nonz, mat_nonz, max_dets, seed, 1 if distribution == "HB_unnorm" else 0)
        self._ck(self.lib.fries_frisys_setup(self.h, C.byref(p)))
        self.max_dets = max_dets

    # ---- fciqmc_mol
    def setup_fciqmc(self, *, epsilon, target_walkers, max_dets, initiator=0, seed=0, distribution="NU", trial=None, ini=None, fp=False):
        """fciqmc_mol with the near-uniform excitation generator (FRIES_bin/fciqmc_mol.cpp, --distribution NU): HF trial vector,
        100 walkers on HF to start; uniforms from a counter-based stream (see csrc/fciqmc.hip)."""
        if distribution not in ("NU", "HB"):
            raise RuntimeError('"dist_str" argument must be either "NU" or "HB"')
        for pair, fn in ((trial, self.lib.fries_set_trial_vector), (ini, self.lib.fries_set_initial_vector)):
            if pair is not None:
                d = np.ascontiguousarray(pair[0], dtype=np.uint64)
                v = np.ascontiguousarray(pair[1], dtype=np.float64)
                self._ck(fn(self.h, _ptr(d), _ptr(v), min(d.size, v.size)))
        p = FciqmcParams(epsilon, target_walkers, initiator, max_dets, seed, 1 if distribution == "HB" else 0, 1 if fp else 0)      # fp: fciqmc_fp_mol
        self._ck(self.lib.fries_fciqmc_setup(self.h, C.byref(p)))
        self.max_dets = max_dets

    def iterate_fciqmc(self, n_iter: int):
        logs = np.zeros(n_iter, dtype=FCIQMC_LOG_DTYPE)
        self._ck(self.lib.fries_fciqmc_iterate(self.h, n_iter, _ptr(logs)))
        return logs

    # ---- frimulti_mol
    def setup_multi(self, *, epsilon, vec_nonz, mat_nonz, max_dets, target_norm=0.0, initiator=0.0, seed=0, trial=None, ini=None):
        """frimulti_mol (FRIES_bin/frimulti_mol.cpp, --distribution HB): multinomial matrix compression, systematic vector compression.
        ini: (dets, vals) of --ini_vec; trial: --trial_vec, which the reference refuses ("Insufficient memory allocated in adder") and so does this."""
        for pair, fn in ((trial, self.lib.fries_set_trial_vector), (ini, self.lib.fries_set_initial_vector)):
            if pair is not None:
                d = np.ascontiguousarray(pair[0], dtype=np.uint64)
                v = np.ascontiguousarray(pair[1], dtype=np.float64)
                self._ck(fn(self.h, _ptr(d), _ptr(v), min(d.size, v.size)))
        p = FrimultiParams(epsilon, target_norm, initiator, vec_nonz, mat_nonz, max_dets, seed)
        self._ck(self.lib.fries_frimulti_setup(self.h, C.byref(p)))
        self.max_dets = max_dets

    def iterate_multi(self, n_iter: int):
        logs = np.zeros(n_iter, dtype=FCIQMC_LOG_DTYPE)
        self._ck(self.lib.fries_frimulti_iterate(self.h, n_iter, _ptr(logs)))
        return logs

    # ---- frifull_mol
    def setup_full(self, *, epsilon, vec_nonz, max_dets, target_norm=0.0, seed=0, spawn_cap=0):
        """frifull_mol (FRIES_bin/frifull_mol.cpp): vector compression + the Hamiltonian applied in full, start from 100 x HF."""
        p = FrifullParams(epsilon, target_norm, vec_nonz, max_dets, seed, spawn_cap)
        self._ck(self.lib.fries_frifull_setup(self.h, C.byref(p)))
        self.max_dets = max_dets

    def iterate_full(self, n_iter: int, want_logs: bool = True):
        logs = np.zeros(n_iter, dtype=ITERLOG_DTYPE) if want_logs else None
        self._ck(self.lib.fries_frifull_iterate(self.h, n_iter, _ptr(logs) if want_logs else None))
        return logs

    # ---- frisys_hh
    def setup_hh(self, *, n_elec, n_sites, eps, U, omega, g, gs_energy, vec_nonz, max_dets, target_norm=0.0, initiator=0.0, seed=0, full=False):
        """frisys_hh (FRIES_bin/frisys_hh.cpp): 1-D Hubbard-Holstein chain, open boundaries, t = 1, start from 100 x Neel.
        full=True: frifull_hh (FRIES_bin/frifull_hh.cpp), the Hamiltonian applied in full instead of compressed."""
        p = HHParams(n_elec, n_sites, eps, U, omega, g, gs_energy, target_norm, initiator, vec_nonz, max_dets, seed, 1 if full else 0)
        self._ck(self.lib.fries_hh_setup(self.h, C.byref(p)))
        self.max_dets = max_dets

    def iterate_hh(self, n_iter: int, want_logs: bool = True):
        logs = np.zeros(n_iter, dtype=ITERLOG_DTYPE) if want_logs else None
        self._ck(self.lib.fries_hh_iterate(self.h, n_iter, _ptr(logs) if want_logs else None))
        return logs

    def idx_to_proc(self, dets) -> np.ndarray:
        """DistVec::idx_to_proc: the rank that owns each determinant."""
        d = np.ascontiguousarray(dets, dtype=np.uint64)
        out = np.zeros(d.size, dtype=np.int32)
        self._ck(self.lib.fries_idx_to_proc(self.h, _ptr(d), d.size, _ptr(out)))
        return out

    def iterate(self, n_iter: int, want_logs: bool = True):
        logs = np.zeros(n_iter, dtype=ITERLOG_DTYPE) if want_logs else None
        assert ITERLOG_DTYPE.itemsize == C.sizeof(IterLog)
        self._ck(self.lib.fries_frisys_iterate(self.h, n_iter, _ptr(logs) if want_logs else None))
        return logs

    def vec_info(self):
        a, b, c = C.c_uint32(), C.c_int32(), C.c_uint32()
        self._ck(self.lib.fries_vec_info(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def vector(self):
        n = self.vec_info()[0]
        d = np.zeros(max(n, 1), dtype=np.uint64)
        v = np.zeros(max(n, 1))
        m = C.c_size_t()
        self._ck(self.lib.fries_vec_download(self.h, _ptr(d), _ptr(v), d.size, C.byref(m)))
        return d[:m.value], v[:m.value]

    def htrial(self):
        cap = 4 * self.mol.n_orb ** 2 * self.mol.n_elec ** 2 + 16
        d = np.zeros(cap, dtype=np.uint64)
        v = np.zeros(cap)
        m = C.c_size_t()
        self._ck(self.lib.fries_htrial_download(self.h, _ptr(d), _ptr(v), cap, C.byref(m)))
        return d[:m.value].copy(), v[:m.value].copy()

    def set_spin_parity(self, spin_parity: int):
        """Time-reversal symmetrised vectors (spin_parity = +-1; 0 = off): see fries_set_spin_parity."""
        self._ck(self.lib.fries_set_spin_parity(self.h, int(spin_parity)))

    def h_offdiag_list(self, dets, vals):
        """h_op_offdiag (dest column 1, h_fac 1, the context's spin parity) on a fresh vector holding (dets, vals): stored determinants, column 1."""
        d = np.ascontiguousarray(dets, dtype=np.uint64)
        v = np.ascontiguousarray(vals, dtype=np.float64)
        cap = d.size * (self.mol.n_orb ** 2 * self.mol.n_elec ** 2 + 2) + 64
        od = np.zeros(cap, dtype=np.uint64)
        ov = np.zeros(cap)
        m = C.c_size_t()
        self._ck(self.lib.fries_h_offdiag_list(self.h, _ptr(d), _ptr(v), d.size, _ptr(od), _ptr(ov), cap, C.byref(m)))
        return od[:m.value].copy(), ov[:m.value].copy()

    def vec_add(self, dets, vals, ini):
        d = np.ascontiguousarray(dets, dtype=np.uint64)
        v = np.ascontiguousarray(vals, dtype=np.float64)
        f = np.ascontiguousarray(ini, dtype=np.uint8)
        self._ck(self.lib.fries_vec_add(self.h, _ptr(d), _ptr(v), _ptr(f), d.size))

    def vec_load(self, dets, vals):
        d = np.ascontiguousarray(dets, dtype=np.uint64)
        v = np.ascontiguousarray(vals, dtype=np.float64)
        self._ck(self.lib.fries_vec_load(self.h, _ptr(d), _ptr(v), d.size))

    def apply_hbpp_sys(self, n_samp: int, rn, unit_matrel: bool = False):
        rn = np.ascontiguousarray(rn, dtype=np.float64)
        cap = n_samp + 4096
        pos = np.zeros(cap, dtype=np.uint32)
        orbs = np.zeros((cap, 4), dtype=np.uint8)
        vals = np.zeros(cap)
        m = C.c_size_t()
        cl = np.zeros(5, dtype=np.uint32)
        self._ck(self.lib.fries_apply_hbpp_sys(self.h, n_samp, _ptr(rn), int(unit_matrel), _ptr(pos), _ptr(orbs), _ptr(vals), cap, C.byref(m), _ptr(cl)))
        k = m.value
        return pos[:k].copy(), orbs[:k].copy(), vals[:k].copy(), cl

    def rng_state(self) -> str:
        """The driver generator in std::mt19937's text form."""
        need = C.c_size_t()
        self._ck(self.lib.fries_rng_get_state(self.h, None, 0, C.byref(need)))
        buf = C.create_string_buffer(need.value)
        self._ck(self.lib.fries_rng_get_state(self.h, buf, need.value, None))
        return buf.value.decode()

    def set_rng_state(self, text: str):
        self._ck(self.lib.fries_rng_set_state(self.h, text.encode()))

    def apply_hbpp_piv(self, n_samp: int, unit_matrel: bool = False):
        """apply_HBPP_piv (heat_bathPP.cpp:1014-1419) on the stored vector; the uniforms come from the engine's generator (restart(seed))."""
        cap = n_samp + 4096
        pos = np.zeros(cap, dtype=np.uint32)
        orbs = np.zeros((cap, 4), dtype=np.uint8)
        vals = np.zeros(cap)
        m = C.c_size_t()
        sl = np.zeros(5, dtype=np.uint32)
        self._ck(self.lib.fries_apply_hbpp_piv(self.h, n_samp, int(unit_matrel), _ptr(pos), _ptr(orbs), _ptr(vals), cap, C.byref(m), _ptr(sl)))
        k = m.value
        return pos[:k].copy(), orbs[:k].copy(), vals[:k].copy(), sl

    def compress_vec(self, n_samp: int, rn: float):
        nk = C.c_uint32()
        gn = C.c_double()
        self._ck(self.lib.fries_compress_vec(self.h, n_samp, rn, C.byref(nk), C.byref(gn)))
        return nk.value, gn.value

    def compress_vec_piv(self, n_samp: int):
        """compress_vecs with one vector (vec_utils.cpp:9-32): pivotal compression of column 0 to at most n_samp
        non-zeros, drawing from the engine's mt19937; returns (elements preserved exactly, one-norm before)."""
        nk = C.c_uint32()
        gn = C.c_double()
        self._ck(self.lib.fries_compress_vec_piv(self.h, n_samp, C.byref(nk), C.byref(gn)))
        return nk.value, gn.value

    def test_piv_adjust(self, n_loc: int, exp_loc: float, n_tot: int, tot_norm: float):
        nl = C.c_uint32(n_loc)
        nn = C.c_double()
        fl = np.zeros(max(self.vec_info()[0], 1), dtype=np.uint8)
        self._ck(self.lib.fries_test_piv_adjust(self.h, C.byref(nl), exp_loc, n_tot, tot_norm, C.byref(nn), _ptr(fl)))
        return nl.value, nn.value, fl

    def copy_bandwidth(self, nbytes: int = 1 << 30, reps: int = 5) -> float:
        out = C.c_double()
        self._ck(self.lib.fries_measure_copy_bandwidth(self.h, nbytes, reps, C.byref(out)))
        return out.value

    def piv_stats(self):
        a, b = C.c_uint64(), C.c_uint64()
        self.last_piv_reason = self.lib.fries_piv_stats(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def next_draw(self) -> int:
        return int(self.lib.fries_next_draw(self.h))

    def test_teeth(self, r0: float, unit: float, n: int, queries):
        q = np.ascontiguousarray(queries, dtype=np.float64)
        pos = np.zeros(max(n, 1))
        below = np.zeros(max(q.size, 1), dtype=np.uint32)
        self._ck(self.lib.fries_test_teeth(self.h, r0, unit, n, _ptr(pos), _ptr(q), q.size, _ptr(below)))
        return pos[:n], below[:q.size]

    def test_seqsum(self, vals, start: float = 0.0):
        a = np.ascontiguousarray(vals, dtype=np.float64)
        out = np.zeros(max(a.size, 1))
        tot = C.c_double()
        dt, ds = C.c_uint32(), C.c_uint32()
        self._ck(self.lib.fries_test_seqsum(self.h, _ptr(a), a.size, start, _ptr(out), C.byref(tot), C.byref(dt), C.byref(ds)))
        return out[:a.size], tot.value, dt.value, ds.value

    def restart(self, seed: int, en_shift: float = 0.0, last_one_norm: float = 0.0, iterat: int = 0):
        self._ck(self.lib.fries_frisys_restart(self.h, seed, en_shift, last_one_norm, iterat))

    def tie_margins(self, enable: bool = True):
        """(smallest relative margin of a find_keep_sub comparison, of a find_preserve comparison) since the last call."""
        a, b = C.c_double(), C.c_double()
        self._ck(self.lib.fries_tie_margins(self.h, int(enable), C.byref(a), C.byref(b)))
        return a.value, b.value

    def prof_enable(self, on: bool = True):
        self._ck(self.lib.fries_prof_enable(self.h, int(on)))

    def prof_report(self):
        """{kernel: (total_ms, calls)} from HIP events on the engine's stream."""
        out = {}
        buf = C.create_string_buffer(128)
        for i in range(self.lib.fries_prof_count(self.h)):
            ms, calls = C.c_double(), C.c_uint64()
            self._ck(self.lib.fries_prof_get(self.h, i, buf, 128, C.byref(ms), C.byref(calls)))
            out[buf.value.decode()] = (ms.value, calls.value)
        return out

    def counters(self):
        v = [C.c_uint64() for _ in range(5)]
        self.lib.fries_counters(self.h, *[C.byref(x) for x in v])
        return dict(zip(("iters", "spawns", "launches", "fks_replays", "stage_elems"), (x.value for x in v)))
