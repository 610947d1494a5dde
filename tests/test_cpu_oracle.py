"""CPU suite (-m "not gpu"): the oracle against the golden vectors the real reference produced, the
host logic, and the C-ABI library's exports."""
import ctypes
import hashlib
import os
import re
import struct

import numpy as np
import pytest

import golden_io
from fries_amd import fcidump

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_params(r):
    return dict(epsilon=r["epsilon"], vec_nonz=r["vec_nonz"], mat_nonz=r["mat_nonz"], max_dets=r["max_dets"], target_norm=r["target_norm"],
                initiator=r["initiator"], seed=r["seed"], distribution=r["distribution"])


@pytest.mark.parametrize("name", sorted(golden_io.manifest()["runs"]))
def test_oracle_reproduces_reference_trajectory(oracle, mols, name):
    """Bit-for-bit: every logged scalar and the digest of (position, determinant, value) per iteration."""
    r = golden_io.manifest()["runs"][name]
    g = golden_io.read_traj(name)
    orc = oracle.OracleFrisys(mols(r["shape"]), **_run_params(r))
    assert orc.p_doub == g["p_doub"] and orc.hf_energy == g["hf_en"]
    n_check = min(len(g["rows"]), 40)
    for row in g["rows"][:n_check]:
        lg = orc.iterate(1)[0]
        for f in ("numer", "denom", "norm", "shift"):
            assert float(lg[f]) == row[f], (name, row["it"], f)
        for f in ("nkept", "n_nonz", "curr_size", "num_success"):
            assert int(lg[f]) == row[f], (name, row["it"], f)
        if row["it"] % 10 == 9 or row["it"] < 3:
            d, v = orc.vector()
            assert golden_io.vec_hash(d, v) == row["hash"], (name, row["it"])


def _extras(r):
    kw = {}
    if "trial" in r:
        kw["trial"] = golden_io.read_text_vector(r["trial"])
    if "ini" in r:
        kw["ini"] = golden_io.read_text_vector(r["ini"])
    if "ham_shift" in r:
        kw["ham_shift"] = r["ham_shift"]            # the synthetic FCIDUMPs carry no core energy
    return kw


@pytest.mark.parametrize("name", sorted(golden_io.manifest()["extra_runs"]))
def test_oracle_driver_options_reproduce_reference(oracle, mols, name):
    """--trial_vec / --ini_vec (text vectors through the reference's own reader) and --ham_shift: the restatement against the
    reference's trajectory, bit for bit."""
    r = golden_io.manifest()["extra_runs"][name]
    g = golden_io.read_traj(name)
    orc = oracle.OracleFrisys(mols(r["shape"]), **_run_params(r), **_extras(r))
    assert orc.p_doub == g["p_doub"] and orc.hf_energy == g["hf_en"]
    for row in g["rows"]:
        lg = orc.iterate(1)[0]
        for f in ("numer", "denom", "norm", "shift"):
            assert float(lg[f]) == row[f], (name, row["it"], f)
        for f in ("nkept", "n_nonz", "curr_size", "num_success"):
            assert int(lg[f]) == row[f], (name, row["it"], f)
    d, v = orc.vector()
    assert golden_io.vec_hash(d, v) == g["rows"][-1]["hash"]


@pytest.mark.parametrize("name", sorted(golden_io.manifest()["full_runs"]))
def test_oracle_frifull_reproduces_reference(oracle, mols, name):
    """fo::Frifull against what the reference's frifull_mol loop logged (deterministic H application + vector compression)."""
    r = golden_io.manifest()["full_runs"][name]
    g = golden_io.read_traj(name)
    orc = oracle.OracleFull(mols(r["shape"]), epsilon=r["epsilon"], vec_nonz=r["vec_nonz"], max_dets=r["max_dets"], target_norm=r["target_norm"], seed=r["seed"])
    for row in g["rows"][:15]:
        lg = orc.iterate(1)[0]
        for f in ("numer", "denom", "norm", "shift"):
            assert float(lg[f]) == row[f], (name, row["it"], f)
        for f in ("nkept", "n_nonz", "curr_size", "num_success"):
            assert int(lg[f]) == row[f], (name, row["it"], f)
        if row["it"] % 5 == 4:
            d, v = orc.vector()
            assert golden_io.vec_hash(d, v) == row["hash"], (name, row["it"])


@pytest.mark.parametrize("name", sorted(golden_io.manifest()["mpi_runs"]))
def test_oracle_ranks_reproduce_reference_under_mpiexec(oracle, mols, name):
    """The multi-rank oracle (P in-process ranks) against what every rank of the real reference logged under
    `mpiexec -n P`: bit-for-bit scalars, shard sizes and shard digests."""
    r = golden_io.manifest()["mpi_runs"][name]
    P = r["n_ranks"]
    g = [golden_io.read_traj(name, rank=k) for k in range(P)]
    orc = oracle.OracleRanks(P, mols(r["shape"]), **_run_params(r))
    assert orc.hf_proc == g[0]["hf_proc"] and orc.htrial()[0].size == g[0]["n_htrial"]
    assert orc.p_doub() == g[0]["p_doub"]
    n_it = len(g[0]["rows"])
    logs = orc.iterate(n_it)
    for k in range(P):
        for i, row in enumerate(g[k]["rows"]):
            lg = logs[k, i]
            for f in ("numer", "denom", "norm", "shift"):
                assert float(lg[f]) == row[f], (name, k, i, f)
            for f in ("nkept", "n_nonz", "curr_size", "num_success"):
                assert int(lg[f]) == row[f], (name, k, i, f)
        d, v = orc.vector(k)
        assert golden_io.vec_hash(d, v) == g[k]["rows"][-1]["hash"], (name, k)


def test_comm_layer_world_size_2_gloo(tmp_path):
    """fries_amd.comm.TorchComm under torch.distributed (gloo, 2 ranks, CPU tensors), driven through the fries_comm
    function pointers like the engine drives it."""
    import socket
    import subprocess
    import sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "comm_worker.py")]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert res.stdout.count("comm ok") == 2, res.stdout[-2000:]        # the two ranks' lines may interleave


def test_sharding_hash_matches_reference_ranks(oracle, mols):
    """idx_to_proc of the oracle (checked against the reference's hf_proc and shard sizes through the rank goldens)
    partitions a vector the way the shards of a 4-rank run hold it."""
    name = "n2_m10000_unnorm_p4"
    r = golden_io.manifest()["mpi_runs"][name]
    orc = oracle.OracleRanks(4, mols(r["shape"]), **_run_params(r))
    orc.iterate(5)
    for k in range(4):
        d, v = orc.vector(k)
        nz = d[v != 0]
        assert nz.size > 0 and all(orc.idx_to_proc(x) == k for x in nz[:200])


@pytest.mark.parametrize("name", sorted(golden_io.manifest()["fciqmc_runs"]) + sorted(golden_io.manifest()["fciqmc_fp_runs"]))
def test_oracle_fciqmc_reproduces_reference(oracle, mols, name):
    """fciqmc_mol (near-uniform and heat-bath generators) and fciqmc_fp_mol (real-valued walkers, fciqmc_fp_*) restated, consuming the
    reference's mt19937 stream: every logged scalar, the walker counts and the digest of (position, determinant, walkers) against the
    reference's own loops."""
    man = golden_io.manifest()
    r = man["fciqmc_runs"][name] if name in man["fciqmc_runs"] else man["fciqmc_fp_runs"][name]
    if r.get("fp"):
        kw_fp = dict(fp=True)
    else:
        kw_fp = {}
    rows = []
    with open(os.path.join(golden_io.GOLD, name + ".traj")) as f:
        for ln in f:
            if ln.startswith("#"):
                continue
            t = ln.split()
            rows.append((float.fromhex(t[1]), float.fromhex(t[2]), float.fromhex(t[3]), float.fromhex(t[4]), int(t[5]), int(t[6]), int(t[7]), int(t[8]), int(t[9], 16)))
    kw = {}
    if "trial" in r:        # --trial_vec / --ini_vec: text vectors through the reference's reader; the last trial entry counts twice there
        kw["trial"] = golden_io.read_text_vector(r["trial"])
    if "ini" in r:
        kw["ini"] = golden_io.read_text_vector(r["ini"])
    orc = oracle.OracleFciqmc(mols(r["shape"]), epsilon=r["epsilon"], target_walkers=r["target_walkers"], max_dets=r["max_dets"], initiator=r["initiator"],
                              seed=r["seed"], counter_rng=False, distribution=r["distribution"], **kw, **kw_fp)
    logs = orc.iterate(r["n_iter"])
    for i, row in enumerate(rows):
        lg = logs[i]
        assert (float(lg["numer"]), float(lg["denom"]), float(lg["norm"]), float(lg["shift"])) == row[:4], (name, i)
        assert (int(lg["n_nonz"]), int(lg["n_ini"]), int(lg["curr_size"]), int(lg["n_spawn"])) == row[4:8], (name, i)
    d, v = orc.vector()
    assert golden_io.vec_hash(d, v) == rows[-1][8]


def _read_fq_rows(name, rank=None):
    rows = []
    with open(os.path.join(golden_io.GOLD, name + ".traj" + ("" if rank is None else f".r{rank}"))) as f:
        for ln in f:
            if ln.startswith("#"):
                continue
            t = ln.split()
            rows.append((float.fromhex(t[1]), float.fromhex(t[2]), float.fromhex(t[3]), float.fromhex(t[4]), int(t[5]), int(t[6]), int(t[7]), int(t[8]), int(t[9], 16)))
    return rows


@pytest.mark.parametrize("name", sorted(golden_io.manifest()["fciqmc_mpi_runs"]) + sorted(golden_io.manifest().get("fciqmc_fp_mpi_runs", {})))
def test_oracle_fciqmc_ranks_reproduce_reference_under_mpiexec(oracle, mols, name):
    """fciqmc_mol sharded over ranks (a generator per process, one all-to-all of the spawns per iteration, walker totals and
    projections summed in rank order): every rank of the in-process oracle against what the same rank of the reference logged
    under mpiexec -- shard sizes, local counts, digests; shift and walker number everywhere; the projected energy on the rank
    that owns HF (the others keep their own terms in the reference)."""
    man = golden_io.manifest()
    r = man["fciqmc_mpi_runs"][name] if name in man["fciqmc_mpi_runs"] else man["fciqmc_fp_mpi_runs"][name]      # fciqmc_fp_*: real-valued walkers
    P = r["n_ranks"]
    orc = oracle.OracleFciqmcRanks(P, mols(r["shape"]), epsilon=r["epsilon"], target_walkers=r["target_walkers"], max_dets=r["max_dets"],
                                   initiator=r["initiator"], seed=r["seed"], counter_rng=False, distribution=r["distribution"], fp=bool(r.get("fp")))
    logs = orc.iterate(r["n_iter"])
    hf = orc.hf_proc
    for k in range(P):
        rows = _read_fq_rows(name, k)
        for i, row in enumerate(rows):
            lg = logs[k, i]
            assert (float(lg["norm"]), float(lg["shift"])) == row[2:4], (name, k, i)
            assert (int(lg["n_nonz"]), int(lg["n_ini"]), int(lg["curr_size"]), int(lg["n_spawn"])) == row[4:8], (name, k, i)
            if k == hf:
                assert (float(lg["numer"]), float(lg["denom"])) == row[:2], (name, k, i)
        d, v = orc.vector(k)
        assert golden_io.vec_hash(d, v) == rows[-1][8], (name, k)


@pytest.mark.parametrize("dist", ["NU", "HB"])
def test_fciqmc_counter_stream_is_statistically_the_reference_stream(oracle, mols, dist):
    """the counter-based uniform stream the GPU replays cannot reproduce the reference's mt19937
    trajectory, so besides the function-level and lockstep pins the two streams must agree in distribution: projected energies
    (ratio of time-averaged numerator and denominator after equilibration) over independent seeds, within four standard errors."""
    import numpy as np
    mol = mols("Ne")

    def energy(seed, counter):
        o = oracle.OracleFciqmc(mol, epsilon=0.02, target_walkers=4000, max_dets=60000, initiator=0, seed=seed, counter_rng=counter, distribution=dist)
        lg = o.iterate(2000)
        return float(np.sum(lg["numer"][1000:]) / np.sum(lg["denom"][1000:]))

    a = np.array([energy(s, False) for s in range(1, 8)])
    b = np.array([energy(s, True) for s in range(1, 8)])
    se = np.sqrt(a.var(ddof=1) / a.size + b.var(ddof=1) / b.size)
    assert abs(a.mean() - b.mean()) < 4 * se, (a.mean(), b.mean(), se)
    assert se < 5e-4 and abs(a.mean() - b.mean()) < 1e-3


def test_heat_bath_probabilities_are_the_sampling_frequencies(oracle, mols):
    """the probability calc_norm_wt reports for a heat-bath double excitation (heat_bathPP.cpp:413-481) is the
    frequency with which hb_doub_multi draws it; the accepted fraction is the listed total (the rest is draws it rejects)."""
    import ctypes as C
    import numpy as np
    lib = oracle.load()
    lib.fo_hb_sample_hist.restype = C.c_uint64
    lib.fo_hb_sample_hist.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    mol = mols("Ne")
    orc = oracle.OracleFrisys(mol, epsilon=0.01, vec_nonz=10, mat_nonz=10, max_dets=100, seed=1, distribution="HB")
    hd, _ = orc.htrial()
    n = 400000
    for det in (int(hd[0]), int(hd[7])):
        cnt = np.zeros(4096, dtype=np.uint64); rep = np.zeros(4096); lst = np.zeros(4096)
        acc = lib.fo_hb_sample_hist(orc.h, det, n, 17, cnt.ctypes.data, rep.ctypes.data, lst.ctypes.data, cnt.size)
        assert 0 < acc <= n
        k = int(np.flatnonzero(lst > 0)[-1]) + 1
        p, c = lst[:k], cnt[:k].astype(float)
        assert np.all((rep[:k] == 0) | (rep[:k] == p))                      # what a draw reports is the listed probability, bit for bit
        assert abs(acc / n - p.sum()) < 5 * np.sqrt(p.sum() * (1 - p.sum()) / n)
        z = (c / n - p) / np.sqrt(np.maximum(p * (1 - p), 1e-300) / n)
        assert np.max(np.abs(z)) < 5.5, float(np.max(np.abs(z)))
        assert abs(np.mean(z ** 2) - 1) < 0.25                              # chi-square per bin ~ 1


def _hh_params(r):
    return dict(n_elec=r["n_elec"], n_sites=r["n_sites"], eps=r["eps"], U=r["U"], omega=r["omega"], g=r["g"], gs_energy=r["gs_energy"],
                vec_nonz=r["vec_nonz"], max_dets=r["max_dets"], target_norm=r["target_norm"], initiator=r["initiator"], seed=r["seed"])


_HH_ALL = dict(golden_io.manifest()["hh_runs"], **golden_io.manifest()["hhfull_runs"])


@pytest.mark.parametrize("name", sorted(_HH_ALL))
def test_oracle_hubbard_holstein_reproduces_reference(oracle, name):
    """frisys_hh and frifull_hh (1-D Hubbard-Holstein) restated against the reference's own loops, one rank and under mpiexec -n 3:
    every logged scalar bit for bit and the digest of every shard."""
    r = _HH_ALL[name]
    P = r["n_ranks"]
    orc = oracle.OracleHH(n_ranks=P, full=name in golden_io.manifest()["hhfull_runs"], **_hh_params(r))
    logs = orc.iterate(r["n_iter"])
    if P == 1:
        logs = logs[None, :]
    for k in range(P):
        g = golden_io.read_traj(name, rank=None if P == 1 else k)
        for i, row in enumerate(g["rows"]):
            lg = logs[k, i]
            for f in ("numer", "denom", "norm", "shift"):
                assert float(lg[f]) == row[f], (name, k, i, f)
            for f in ("nkept", "n_nonz", "curr_size", "num_success"):
                assert int(lg[f]) == row[f], (name, k, i, f)
        d, v = orc.vector(k)
        assert golden_io.vec_hash(d, v) == g["rows"][-1]["hash"], (name, k)


def test_oracle_snapshot_matches_reference(oracle, mols):
    name = "ne_m2000_unnorm"
    r = golden_io.manifest()["runs"][name]
    g = golden_io.read_traj(name)
    it, ent = sorted(g["snaps"].items())[-1]
    orc = oracle.OracleFrisys(mols(r["shape"]), **_run_params(r))
    orc.iterate(it + 1)
    d, v = orc.vector()
    nz = np.nonzero(v != 0)[0]
    assert len(ent) == nz.size
    for (pos, det, val), i in zip(ent, nz):
        assert pos == i and det == int(d[i]) and val == v[i]


def test_oracle_new_hb_all_known_answer(oracle, mols):
    """[new_hb_all] (reference tests/test_hamiltonian.cpp:454-520): n_samp = n_ex returns every excitation
    with |value| == 1, in the reference's order."""
    g = golden_io.read_hbpp_all()
    mol = mols("Ne")
    mol2 = fcidump.MolInput(22, 8, np.array(golden_io.HBPP_ALL_SYMM, dtype=np.uint8), mol.h_core, mol.eris, 0.0, "D2h")
    n_ex = 22 * 22 * 8 * 8
    orc = oracle.OracleFrisys(mol2, epsilon=0.01, vec_nonz=10, mat_nonz=n_ex, max_dets=64, seed=0, distribution="HB_unnorm")
    for k, v in g["tens"].items():
        orc.set_hb_tensor(golden_io.TENSOR_ID[k], np.array(v))
    orc.set_p_doub(0.95)
    hf = (1 << 4) - 1 | (((1 << 4) - 1) << 22)
    orc.vec_load(np.array([hf], dtype=np.uint64), np.array([1.0]))
    pos, orbs, vals = orc.apply_hbpp_sys(n_ex, g["rn"], unit_matrel=True)
    assert pos.size == g["n"] == 984
    assert np.array_equal(orbs, g["orbs"]) and np.array_equal(vals, g["vals"])
    assert np.all(np.abs(np.abs(vals) - 1) < 1e-7)
    # second half of the reference test (:507-519): apply_HBPP_piv with the same budget returns the same excitations
    assert g["piv_n"] == 984 and np.array_equal(g["piv_orbs"], g["orbs"])
    ppos, porbs, pvals, _ = orc.apply_hbpp_piv(n_ex, unit_matrel=True)
    assert np.array_equal(porbs, g["piv_orbs"]) and pvals.tobytes() == g["piv_vals"].tobytes() and np.all(ppos == 0)


def test_fcidump_parser_matches_reference_parser(mols, tmp_path):
    """The binary image the reference's parse_fcidump produced (sha256 in the manifest) equals ours."""
    man = golden_io.manifest()["ints"]
    for shape, info in man.items():
        mol = mols(shape)
        path = tmp_path / (shape + ".FCIDUMP")
        fcidump.write_fcidump(str(path), mol)
        m2 = fcidump.parse_fcidump(str(path), info["point_group"])
        blob = struct.pack("<II", m2.n_orb, m2.n_elec) + m2.irreps.tobytes() + struct.pack("<d", m2.core_en) + \
            np.ascontiguousarray(m2.h_core).tobytes() + np.ascontiguousarray(m2.eris).tobytes()
        assert hashlib.sha256(blob).hexdigest() == info["sha256"], shape


def test_fcidump_errors():
    with pytest.raises(RuntimeError):
        fcidump.convert_symm([9], "D2h")
    with pytest.raises(RuntimeError):
        fcidump.convert_symm([1], "Oh")


def test_symmetry_label_maps():
    # reference tests/test_hamiltonian.cpp:642-707 (convert_symm)
    assert list(fcidump.convert_symm([1, 2, 3, 4, 5, 6, 7, 8], "D2h")) == [0, 7, 6, 1, 5, 2, 3, 4]
    assert list(fcidump.convert_symm([1, 2, 3, 4], "C2v")) == [0, 2, 3, 1]
    assert list(fcidump.convert_symm([1, 2, 3, 4], "D2")) == [0, 3, 2, 1]
    assert list(fcidump.convert_symm([1, 2], "Cs")) == [0, 1]


def test_oracle_compression_is_identity_when_budget_exceeds_nnz(oracle, mols):
    """reference tests/test_compression.cpp:62-117."""
    orc = oracle.OracleFrisys(mols("Ne"), epsilon=0.01, vec_nonz=100, mat_nonz=100, max_dets=4000, seed=3)
    rng = np.random.RandomState(0)
    dets, _ = orc.vector()
    hd, hv = orc.htrial()
    vals = rng.standard_normal(hd.size)
    orc.vec_load(hd, vals)
    nk, gn = orc.compress_vec(hd.size + 10, 0.3)
    d2, v2 = orc.vector()
    assert np.array_equal(v2, vals) and np.isclose(gn, np.abs(vals).sum())


def test_oracle_pivotal_compression_known_answers(oracle):
    """fo::piv_comp_parallel against what the reference's own piv_comp_parallel produced (compress_utils.cpp:354-386):
    values, delete flags and generator position, bit for bit; plus the budget invariants."""
    import numpy as np
    cases = golden_io.read_piv_cases()
    assert len(cases) == 4
    for cs in cases:
        v, fl, nxt = oracle.piv_comp(np.array(cs["inp"]), cs["compress_size"], cs["seed"])
        assert v.tobytes() == np.array(cs["out"]).tobytes()
        assert fl.tolist() == cs["flag"]
        assert nxt == cs["next"]
        assert np.count_nonzero(v) <= cs["compress_size"]
        assert np.all((v != 0) | (fl == 1) | (np.array(cs["inp"]) == 0))


@pytest.mark.parametrize("name", sorted(golden_io.manifest()["hbpiv_runs"]))
def test_oracle_apply_hbpp_piv_matches_reference(oracle, mols, name):
    """fo::apply_HBPP_piv (pivotal compression of every HB-PP factor) against what the reference's apply_HBPP_piv returned
    (heat_bathPP.cpp:1014-1419) on the vector of a golden frisys_mol run: positions, orbitals and values bit for bit, stage lengths."""
    import numpy as np
    h = golden_io.manifest()["hbpiv_runs"][name]
    r = golden_io.manifest()["runs"][h["run"]]
    mol = fcidump.synthetic(r["shape"])
    orc = oracle.OracleFrisys(mol, epsilon=r["epsilon"], vec_nonz=r["vec_nonz"], mat_nonz=r["mat_nonz"], max_dets=r["max_dets"], target_norm=r["target_norm"],
                              initiator=r["initiator"], seed=r["seed"], distribution=r["distribution"])
    orc.iterate(h["n_iter"])
    cases = golden_io.read_hbpiv(name)
    assert [[c["n_samp"], c["seed"]] for c in cases] == h["cases"]
    for c in cases:
        orc.restart(c["seed"])
        pos, orbs, vals, st = orc.apply_hbpp_piv(c["n_samp"])
        assert len(pos) == c["n_out"] and st.tolist() == c["stage_len"]
        assert np.array_equal(pos, c["pos"]) and np.array_equal(orbs, c["orbs"]) and vals.tobytes() == c["val"].tobytes()


@pytest.mark.parametrize("name", sorted(golden_io.manifest().get("tr_runs", {})))
def test_oracle_time_reversal_h_op_offdiag_matches_reference(oracle, mols, name):
    """Time-reversal symmetry (spin_parity = +-1): fo::h_op_offdiag with the adjust_tr rule against what the reference's h_op_offdiag
    (molecule.cpp:298-369, 448-665) left in the vector, for both parities: stored determinants in order and values bit for bit.  (The
    harness that wrote the fixture also ran flip_spins on 400 random strings for every n_orb in 5 .. 32 and tr_doub_connect on 4000 occupied
    lists, reference against restatement.)"""
    import numpy as np
    r = golden_io.manifest()["tr_runs"][name]
    orc = oracle.OracleFrisys(mols(r["shape"]), epsilon=0.01, vec_nonz=100, mat_nonz=100, max_dets=1000, seed=1, distribution="HB_unnorm")
    sd, sv, out = golden_io.read_tr(name)
    for sp in (1, -1):
        orc.set_spin_parity(sp)
        d, v = orc.h_offdiag_list(sd, sv)
        assert np.array_equal(d, out[sp][0]) and v.tobytes() == out[sp][1].tobytes(), (name, sp)
    orc.set_spin_parity(0)


@pytest.mark.parametrize("name", sorted(golden_io.manifest().get("hbpiv_tr_runs", {})))
def test_oracle_apply_hbpp_piv_time_reversal_matches_reference(oracle, mols, name):
    """fo::apply_HBPP_piv with spin_parity = +-1 (heat_bathPP.cpp:1326-1407) against the reference's function on the vector of a golden run."""
    import numpy as np
    h = golden_io.manifest()["hbpiv_tr_runs"][name]
    r = golden_io.manifest()["runs"][h["run"]]
    orc = oracle.OracleFrisys(mols(r["shape"]), epsilon=r["epsilon"], vec_nonz=r["vec_nonz"], mat_nonz=r["mat_nonz"], max_dets=r["max_dets"], target_norm=r["target_norm"],
                              initiator=r["initiator"], seed=r["seed"], distribution=r["distribution"])
    orc.iterate(h["n_iter"])
    orc.set_spin_parity(h["spin_parity"])
    try:
        for c in golden_io.read_hbpiv(name):
            orc.restart(c["seed"])
            pos, orbs, vals, st = orc.apply_hbpp_piv(c["n_samp"])
            assert len(pos) == c["n_out"] and st.tolist() == c["stage_len"]
            assert np.array_equal(pos, c["pos"]) and np.array_equal(orbs, c["orbs"]) and vals.tobytes() == c["val"].tobytes()
    finally:
        orc.set_spin_parity(0)


@pytest.mark.parametrize("name", sorted(golden_io.manifest()["multi_runs"]))
def test_oracle_frimulti_matches_reference(oracle, name):
    """fo::Fciqmc::iterate_multi on the reference's mt19937 stream against the trajectory of the reference's frimulti_mol loop
    (tests/golden/multi_*.traj): every scalar bit for bit, the counts, and the digest of the stored vector."""
    r = golden_io.manifest()["multi_runs"][name]
    rows = golden_io.read_multi_traj(name)
    mol = fcidump.synthetic(r["shape"])
    kw = {}
    if "ini" in r:          # --ini_vec (frimulti_mol.cpp:205-215): real values through the reference's text reader
        kw["ini"] = golden_io.read_text_vector(r["ini"])
    orc = oracle.OracleMulti(mol, epsilon=r["epsilon"], vec_nonz=r["vec_nonz"], mat_nonz=r["mat_nonz"], max_dets=r["max_dets"], initiator=r["initiator"],
                             target_norm=r["target_norm"], seed=r["seed"], **kw)
    for row in rows:
        lg = orc.iterate(1)[0]
        for f in ("numer", "denom", "norm", "shift"):
            assert float(lg[f]) == row[f], (name, row["it"], f)
        for f in ("n_nonz", "curr_size", "n_spawn", "n_ini"):
            assert int(lg[f]) == row[f], (name, row["it"], f)
        assert orc.nkept == row["nkept"]
    d, v = orc.vector()
    assert golden_io.vec_hash(d, v) == rows[-1]["hash"]


def test_oracle_frimulti_refuses_a_trial_vector_like_the_reference(oracle):
    """frimulti_mol --trial_vec on one rank: trial_vec's Adder holds n_trial entries, add() reports the entry that fills it, and this driver throws on that
    report (frimulti_mol.cpp:149-157) where fciqmc_mol flushes -- every trial file is refused.  The reference run that recorded the message: gen_golden.py."""
    r = golden_io.manifest()["multi_trial_one_rank_error"]
    mol = fcidump.synthetic(r["shape"])
    with pytest.raises(RuntimeError, match=r["error"]):
        oracle.OracleMulti(mol, epsilon=0.01, vec_nonz=5000, mat_nonz=20000, max_dets=200000, initiator=1.0, target_norm=2500.0, seed=3, trial=golden_io.read_text_vector(r["trial"]))


@pytest.mark.parametrize("name", sorted(golden_io.manifest().get("multi_mpi_runs", {})))
def test_oracle_frimulti_ranks_reproduce_reference_under_mpiexec(oracle, name):
    """frimulti_mol sharded over ranks (rank 0's comb offsets broadcast, the comb running through the ranks' norms in rank order, one
    all-to-all of the spawns, collective vector compression): every rank of the in-process oracle against what the same rank of the
    reference logged under mpiexec; the projections on the rank that owns HF."""
    r = golden_io.manifest()["multi_mpi_runs"][name]
    P = r["n_ranks"]
    mol = fcidump.synthetic(r["shape"])
    orc = oracle.OracleMultiRanks(P, mol, epsilon=r["epsilon"], vec_nonz=r["vec_nonz"], mat_nonz=r["mat_nonz"], max_dets=r["max_dets"], initiator=r["initiator"],
                                  target_norm=r["target_norm"], seed=r["seed"])
    logs = orc.iterate(r["n_iter"])
    hf = orc.hf_proc
    for k in range(P):
        rows = golden_io.read_multi_traj(name + ".traj.r%d" % k, raw=True)
        for i, row in enumerate(rows):
            lg = logs[k, i]
            assert (float(lg["norm"]), float(lg["shift"])) == (row["norm"], row["shift"]), (name, k, i)
            assert (int(lg["n_nonz"]), int(lg["curr_size"]), int(lg["n_spawn"]), int(lg["n_ini"])) == (row["n_nonz"], row["curr_size"], row["n_spawn"], row["n_ini"]), (name, k, i)
            if k == hf:
                assert (float(lg["numer"]), float(lg["denom"])) == (row["numer"], row["denom"]), (name, k, i)
        d, v = orc.vector(k)
        assert golden_io.vec_hash(d, v) == rows[-1]["hash"], (name, k)


def test_library_exports_every_declared_symbol():
    """libfries_hip.so loads on a GPU-less host and exports exactly what include/fries_hip.h declares."""
    from fries_amd import engine
    hdr = open(os.path.join(ROOT, "include", "fries_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(fries_[a-z_0-9]+)\s*\(", hdr)))
    assert declared == sorted(engine.EXPORTS)
    if not os.path.exists(engine.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    dll = ctypes.CDLL(engine.LIB_PATH)
    for s in declared:
        assert hasattr(dll, s), s


def test_reference_headers_and_drivers_compile_on_the_host(tmp_path):
    """include/FRIES (the reference's own header names over the C ABI) and the C++ drivers are plain host C++17: they must compile without
    hipcc, against the one-rank MPI stand-in and -- where the image has one -- against the real <mpi.h>."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "t.cpp"
    src.write_text('#include <FRIES/Hamiltonians/heat_bathPP.hpp>\n#include <FRIES/Hamiltonians/hub_holstein.hpp>\n#include <FRIES/hh_vec.hpp>\n'
                   'int main() { Matrix<double> m(2, 2); m(1, 1) = 1; double s = 0, l = 0; adjust_shift(&s, 2.0, &l, 1.0, 0.1); std::mt19937 mt(1); return (int)m(0, 0) + round_binomially(0.0, 3, mt); }\n')
    inc = os.path.join(root, "include")
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-I", inc, "-I", os.path.join(inc, "FRIES", "compat"), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    if os.path.exists("/opt/conda/include/mpi.h"):
        r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", inc, "-I", "/opt/conda/include", str(src)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
    for drv in ("frisys_mol_hip", "fciqmc_mol_hip", "frisys_hh_hip", "frifull_mol_hip"):
        r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", os.path.join(root, "fries_amd", "drivers", drv + ".cpp")], capture_output=True, text=True)
        assert r.returncode == 0, (drv, r.stderr[-2000:])


def test_engine_fails_loudly_without_gpu(mols):
    """No CPU fallback: without a HIP device the engine refuses to start."""
    from fries_amd import engine
    lib = engine.load_library()
    if lib.fries_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError):
        engine.FriEngine(mols("Ne"))


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "fries_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "oracle" not in txt.lower() or fn == "engine.py" and False, os.path.join(dirpath, fn)


@pytest.mark.skipif(not os.path.isdir("/root/reference/N2_load"), reason="the reference's own checkpoint dumps live only in the build container")
def test_reference_checkpoint_dumps_shard_by_our_proc_hash():
    """f-3: the checkpoint an 8-rank run of the REFERENCE left in N2_load/ (dets<rank>.dat: 7-byte indices of 26 orbitals; hash.dat: the
    proc scrambler, raw uint32[52]).  Every index stored by rank r must hash to r under DistVec::idx_to_proc as restated here
    (hash_fxn with the 32-bit-wrapped term, det_hash.hpp:160-170; % 8) for most of its entries (see below) and all of them carry
    5 + 5 electrons.  Files that are whole multiples of 7 bytes only (two of the blobs are truncated)."""
    import oracle_lib
    lib = oracle_lib.load()
    d = "/root/reference/N2_load/"
    scr = np.fromfile(d + "hash.dat", dtype=np.uint32)
    assert scr.size == 52
    n_orb, nb, P = 26, 7, 8
    checked = 0
    for r in range(1, 8):
        raw = np.fromfile(d + f"dets{r}.dat", dtype=np.uint8)
        if raw.size % nb:
            continue
        n = raw.size // nb
        dets = np.zeros(n, dtype=np.uint64)
        for b in range(nb):
            dets |= raw.reshape(n, nb)[:, b].astype(np.uint64) << np.uint64(8 * b)
        dets = dets[dets != 0]                                    # never-used tail slots
        pc = np.zeros(dets.size, dtype=np.int64)
        h = np.zeros(dets.size, dtype=np.uint64)
        k = np.zeros(dets.size, dtype=np.uint64)                  # occupied orbitals seen so far
        with np.errstate(over="ignore"):
            for o in range(2 * n_orb):
                bit = ((dets >> np.uint64(o)) & np.uint64(1)).astype(bool)
                term = (((k + np.uint64(1)) * np.uint64(scr[o])) & np.uint64(0xFFFFFFFF))
                h = np.where(bit, np.uint64(1099511628211) * h + term, h)
                k = k + bit.astype(np.uint64)
                pc += bit
        assert np.all(pc == 10), (r, np.unique(pc))
        alpha = np.zeros(dets.size, dtype=np.int64)
        for o in range(n_orb):
            alpha += ((dets >> np.uint64(o)) & np.uint64(1)).astype(np.int64)
        assert np.all(alpha == 5), r
        # The dump itself is not self-consistent: ~27 % of every rank's indices hash elsewhere under this hash.dat (the run was restarted
        # from an older checkpoint -- params.txt: "Restarting calculation from ../N2_load/" -- and DistVec::load re-hashes locally but
        # never re-shards), so the pin is: the rank is by far the most frequent owner, not the only one.
        owners = np.bincount((h % np.uint64(P)).astype(np.int64), minlength=P)
        assert int(np.argmax(owners)) == r and owners[r] > 0.6 * dets.size, (r, owners.tolist())
        # the restatement's C hash agrees on a sample
        for i in range(0, dets.size, max(1, dets.size // 50)):
            occ = np.array([o for o in range(52) if (int(dets[i]) >> o) & 1], dtype=np.uint8)
            assert int(lib.fo_hash(occ.ctypes.data_as(ctypes.c_void_p), 10, scr.ctypes.data_as(ctypes.c_void_p))) == int(h[i])
        checked += dets.size
    assert checked > 2_000_000


def test_legacy_hf_directory_reader_matches_reference(tmp_path):
    """f-3: the legacy HF-output directory (--hf_path of frifull_mol / frimulti_mol).  fries_amd.fcidump.write_hf_dir writes one; the
    C++ reader of the drivers (parse_hf_dir, fries_amd/drivers/driver_common.hpp) must return the integrals it was written from, and --
    where the reference is built -- so must the reference's own parse_hf_input (oracle/_ref/ref_harness hfdir) on the same files."""
    import subprocess
    mol = fcidump.synthetic("Ne")
    d = str(tmp_path) + "/"
    fcidump.write_hf_dir(d, mol, eps=0.0125, hf_energy=-1.5)
    n = mol.n_orb
    src = tmp_path / "t.cpp"
    src.write_text('#include "%s"\nint main(int c, char **v) { HfDir h = parse_hf_dir(v[1]); fwrite(h.mol.symm.data(), 1, h.mol.symm.size(), stdout); fwrite(h.mol.hcore.data(), 8, h.mol.hcore.size(), stdout);'
                   ' fwrite(h.mol.eris.data(), 8, h.mol.eris.size(), stdout); fprintf(stderr, "%%u %%u %%.17g %%.17g", h.mol.n_orb, h.mol.n_elec, h.eps, h.hf_en); return 0; }\n'
                   % os.path.join(ROOT, "fries_amd", "drivers", "driver_common.hpp"))
    exe = str(tmp_path / "t")
    subprocess.run(["g++", "-std=c++17", "-O1", "-o", exe, str(src)], check=True)
    r = subprocess.run([exe, d], capture_output=True)
    assert r.stderr.decode().split() == [str(n), str(mol.n_elec), "0.012500000000000001", "-1.5"]
    assert np.array_equal(np.frombuffer(r.stdout[:n], dtype=np.uint8), np.asarray(mol.irreps, dtype=np.uint8))
    a = np.frombuffer(r.stdout[n:], dtype=np.float64)
    assert np.array_equal(a[:n * n], np.asarray(mol.h_core).reshape(-1)) and np.array_equal(a[n * n:], np.asarray(mol.eris))
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    if os.path.exists(harness):
        out = str(tmp_path / "ref.bin")
        subprocess.run([harness, "hfdir", d, out], check=True, capture_output=True)
        blob = open(out, "rb").read()
        hdr = np.frombuffer(blob[:12], dtype=np.uint32)
        assert hdr.tolist() == [n, mol.n_elec, 0]
        assert np.array_equal(np.frombuffer(blob[12:12 + n], dtype=np.uint8), np.asarray(mol.irreps, dtype=np.uint8))
        b = np.frombuffer(blob[12 + n:], dtype=np.float64)
        assert b[0] == 0.0125 and b[1] == -1.5
        assert np.array_equal(b[2:2 + n * n], np.asarray(mol.h_core).reshape(-1)) and np.array_equal(b[2 + n * n:], np.asarray(mol.eris))


@pytest.mark.parametrize("name", sorted(golden_io.manifest().get("dense_runs", {})))
def test_oracle_dense_space_matches_reference_golden(name):
    """--det_space (semi-stochastic): the restatement with a dense space -- init_dense, H inside the space applied exactly, the
    compressions restricted to the rest, dense_norm -- against what the real reference logged (`ref_harness frisys` with FRIES_DETSPACE:
    DistVec::init_dense and the driver's own loop bound over the allocated dense-H arrays)."""
    import oracle_lib
    r = golden_io.manifest()["dense_runs"][name]
    g = golden_io.read_traj(name)
    space = np.array([int(x) for x in open(os.path.join(golden_io.GOLD, r["det_space"])).read().split()], dtype=np.uint64)
    assert space.size == r["n_dense"]
    orc = oracle_lib.OracleFrisys(fcidump.synthetic(r["shape"]), epsilon=r["epsilon"], vec_nonz=r["vec_nonz"], mat_nonz=r["mat_nonz"], max_dets=r["max_dets"],
                                  target_norm=r["target_norm"], initiator=r["initiator"], seed=r["seed"], distribution=r["distribution"], det_space=space)
    n_it = min(r["n_iter"], 30)
    lo = orc.iterate(n_it)
    for i in range(n_it):
        row = g["rows"][i]
        for f in ("nkept", "n_nonz", "curr_size", "num_success"):
            assert int(lo[f][i]) == row[f], (i, f)
        assert float(lo["norm"][i]) == row["norm"] and float(lo["shift"][i]) == row["shift"] and float(lo["numer"][i]) == row["numer"] and float(lo["denom"][i]) == row["denom"], i
    d, v = orc.vector()
    assert golden_io.vec_hash(d, v) == g["rows"][n_it - 1]["hash"]
    assert np.array_equal(d[:space.size], space)         # the dense space sits in front, in file order, whatever its values


def test_fries_headers_host_logic(tmp_path):
    """include/FRIES (the reference's headers of this build): everything that runs on the HOST -- bit strings, fermionic signs, excitation
    lists in the reference's order, SymmInfo, the rank / vector hash, the host DistVec + Adder (positions, LIFO re-use of freed
    positions, initiator rule, order of additions, dot, local_norm), Matrix / Matrix<bool> / SymmERIs, adjust_shift and the
    command-line parser -- against the oracle's restatement on random inputs (tests/cpp/test_fries_headers.cpp).  No device call."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import oracle_lib
    oracle_lib.load()                   # builds oracle/_build/libfries_oracle.so if needed
    from fries_amd import build
    assert os.path.exists(build.LIB), "libfries_hip.so has not been built (the headers reference the C ABI)"
    exe = str(tmp_path / "test_fries_headers")
    cmd = ["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "include", "FRIES", "compat"),
           "-I" + os.path.join(root, "oracle"), os.path.join(root, "tests", "cpp", "test_fries_headers.cpp"), "-o", exe,
           os.path.join(root, "oracle", "_build", "libfries_oracle.so"), "-L" + os.path.dirname(build.LIB), "-lfries_hip",
           "-Wl,-rpath," + os.path.join(root, "oracle", "_build"), "-Wl,-rpath," + os.path.dirname(build.LIB), "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib", "-pthread"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "fails=0" in r.stdout, (r.stdout[-3000:], r.stderr[-1000:])


@pytest.mark.parametrize("name", sorted(golden_io.manifest().get("dense_mpi_runs", {})))
def test_oracle_dense_space_over_ranks(oracle, mols, name):
    """--det_space under `mpiexec -n P`: rank 0 reads the file, the dense determinants travel to their owners and every rank keeps its
    share in front of its shard; tot_dense_h and dense_norm are sums over the ranks.  The restatement's in-process ranks against what
    every rank of the real reference logged."""
    r = golden_io.manifest()["dense_mpi_runs"][name]
    P = r["n_ranks"]
    g = [golden_io.read_traj(name, rank=k) for k in range(P)]
    space = np.array([int(x) for x in open(os.path.join(golden_io.GOLD, r["det_space"])).read().split()], dtype=np.uint64)
    orc = oracle.OracleRanks(P, mols(r["shape"]), det_space=space, **_run_params(r))
    n_it = min(len(g[0]["rows"]), 30)
    logs = orc.iterate(n_it)
    for k in range(P):
        for i in range(n_it):
            row, lg = g[k]["rows"][i], logs[k, i]
            for f in ("numer", "denom", "norm", "shift"):
                assert float(lg[f]) == row[f], (name, k, i, f)
            for f in ("nkept", "n_nonz", "curr_size", "num_success"):
                assert int(lg[f]) == row[f], (name, k, i, f)
        d, v = orc.vector(k)
        assert golden_io.vec_hash(d, v) == g[k]["rows"][n_it - 1]["hash"], (name, k)
