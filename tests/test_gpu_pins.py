"""-m gpu: the BASELINE sizes against the REAL reference (tests/golden/*.pin, written by `ref_harness pin`), and the native
C++ transports (csrc/comm_native.hip).

  * pin_n2_m1e6      BASELINE config 2 = bench.py's workload: N2-shaped, m = 1e6, filler + restart + 100 iterations.
  * pin_h2o_m1e7_p8  BASELINE config 4: H2O-shaped, m = 1e7 over 8 ranks (`mpiexec -n 8` in the reference), every rank's shard.
    The 8 ranks are threads of one process on this box's one GPU, exchanging through the native "local" transport.
  * the 2 / 3 / 4-rank goldens of tests/golden (mpiexec) again, through the same C++ transport instead of torch.distributed.
  * librccl directly (ncclAllGather / ncclAllToAllv under the engine's stream) with a world of one.
"""
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

import golden_io
import pin_replay
from fries_amd import fcidump

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PINS = golden_io.manifest().get("pin_runs", {})


@pytest.mark.parametrize("name", sorted(PINS))
def test_pinned_baseline_size_matches_reference(name):
    res = pin_replay.replay(name)
    for k, (fails, info) in enumerate(res):
        assert not fails, (name, k, fails[:6])
        assert info["run_iters"] == PINS[name]["n_iter"]
    print(name, [i for _, i in res][:2])


def _rank_thread_run(name, r, P):
    """The golden run `name` (mpiexec -n P in the reference) with P rank threads over the native local transport."""
    from fries_amd.comm import LocalGroup
    from fries_amd.engine import FriEngine
    mol = fcidump.synthetic(r["shape"])
    grp = LocalGroup(P, r["mat_nonz"])
    comms = [grp.comm(k, 0) for k in range(P)]
    out = [None] * P

    def work(k):
        fails = []
        try:
            g = golden_io.read_traj(name, rank=k)
            eng = FriEngine(mol, device=0, comm=comms[k])
            eng.setup(epsilon=r["epsilon"], vec_nonz=r["vec_nonz"], mat_nonz=r["mat_nonz"], max_dets=r["max_dets"], target_norm=r["target_norm"],
                      initiator=r["initiator"], seed=r["seed"], distribution=r["distribution"])
            for row in g["rows"]:
                lg = eng.iterate(1)[0]
                pin_replay._check_row(lg, row, fails, "mpi")
            d, v = eng.vector()
            if golden_io.vec_hash(d, v) != g["rows"][-1]["hash"]:
                fails.append(("digest",))
            eng.close()
            out[k] = (fails, dict(n_allgather=comms[k].n_allgather, n_alltoallv=comms[k].n_alltoallv, iters=len(g["rows"])))
        except Exception as e:
            print("rank", k, "raised", repr(e), file=sys.stderr, flush=True)     # (the other ranks then wait in their next collective)
            out[k] = ([("exception", repr(e))], {})

    th = [threading.Thread(target=work, args=(k,)) for k in range(P)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    grp.destroy()
    return out


@pytest.mark.parametrize("name", sorted(golden_io.manifest()["mpi_runs"]))
def test_rank_goldens_through_native_local_transport(name):
    r = golden_io.manifest()["mpi_runs"][name]
    res = _rank_thread_run(name, r, r["n_ranks"])
    for k, (fails, info) in enumerate(res):
        assert not fails, (name, k, fails[:6])
        assert info["n_alltoallv"] == info["iters"]         # one spawn exchange per iteration


@pytest.mark.parametrize("name", sorted(golden_io.manifest().get("adder_runs", {})))
def test_adder_rounds_match_reference(name, monkeypatch):
    """Adder::add reports a full buffer, the spawning loop flushes early and a pass takes several perform_add rounds
    (vec_utils.hpp:957-971, frisys_mol.cpp:430-471): the reference under mpiexec with a small Adder (FRIES_ADDER_SIZE = the adder_size
    argument of its DistVec) against the engine's rounds (vec.hip fr_xch_rounds), rank by rank."""
    r = golden_io.manifest()["adder_runs"][name]
    monkeypatch.setenv("FRIES_ADDER_SIZE", str(r["adder_size"]))
    res = _rank_thread_run(name, r, r["n_ranks"])
    for k, (fails, info) in enumerate(res):
        assert not fails, (name, k, fails[:6])
        assert info["n_alltoallv"] > 2 * info["iters"], info         # several exchanges per iteration: the rounds did take place


@pytest.mark.parametrize("name", sorted(golden_io.manifest().get("full_mpi_runs", {})))
def test_frifull_over_ranks_matches_reference(name):
    """frifull_mol under mpiexec -n P (FRIES_bin/frifull_mol.cpp:258-304 with h_op_offdiag's two passes of adds, molecule.cpp:553-660): rank threads
    over the native local transport against what every rank of the reference logged -- counts, norm and shift as doubles, numerator and
    denominator to 1e-10, the shard's digest every iteration."""
    from fries_amd.comm import LocalGroup
    from fries_amd.engine import FriEngine
    r = golden_io.manifest()["full_mpi_runs"][name]
    P = r["n_ranks"]
    mol = fcidump.synthetic(r["shape"])
    cap = 3000000
    grp = LocalGroup(P, cap)
    comms = [grp.comm(k, 0) for k in range(P)]
    out = [None] * P

    def work(k):
        fails = []
        try:
            g = golden_io.read_traj(name, rank=k)
            eng = FriEngine(mol, device=0, comm=comms[k])
            eng.setup_full(epsilon=r["epsilon"], vec_nonz=r["vec_nonz"], max_dets=r["max_dets"], target_norm=r["target_norm"], seed=r["seed"], spawn_cap=cap)
            for row in g["rows"]:
                lg = eng.iterate_full(1)[0]
                row = dict(row, num_success=int(lg["num_success"]))         # (the golden does not record the number of adds)
                pin_replay._check_row(lg, row, fails, "full")
                d, v = eng.vector()
                if golden_io.vec_hash(d, v) != row["hash"]:
                    fails.append(("full", row["it"], "digest"))
                if len(fails) > 6:
                    break
            eng.close()
        except Exception as e:
            print("rank", k, "raised", repr(e), file=sys.stderr, flush=True)
            fails.append(("exception", repr(e)))
        out[k] = fails

    th = [threading.Thread(target=work, args=(k,)) for k in range(P)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    grp.destroy()
    for k in range(P):
        assert not out[k], (name, k, out[k][:6])


_RCCL_ONE = r"""
import sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
import golden_io
from fries_amd import fcidump
from fries_amd.comm import RcclComm
from fries_amd.engine import FriEngine
name = "n2_m10000_unnorm_ini0"
r = golden_io.manifest()["runs"][name]
g = golden_io.read_traj(name)
comm = RcclComm(r["mat_nonz"], 0)
eng = FriEngine(fcidump.synthetic(r["shape"]), device=0, comm=comm)
eng.setup(epsilon=r["epsilon"], vec_nonz=r["vec_nonz"], mat_nonz=r["mat_nonz"], max_dets=r["max_dets"], target_norm=r["target_norm"],
          initiator=r["initiator"], seed=r["seed"], distribution=r["distribution"])
import pin_replay
fails = []
for row in g["rows"]:
    pin_replay._check_row(eng.iterate(1)[0], row, fails, "rccl1")
d, v = eng.vector()
assert not fails, fails[:5]
assert golden_io.vec_hash(d, v) == g["rows"][-1]["hash"]
assert comm.n_alltoallv == len(g["rows"]) and comm.n_allgather > 5 * len(g["rows"])
print("RCCL1 OK", comm.n_allgather / len(g["rows"]), "all-gathers per iteration")
eng.close(); comm.destroy()
"""


def test_native_rccl_transport_world_of_one():
    """ncclAllGather / ncclAllToAllv from C++ under the engine's stream (no torch in the process), one rank: every collective
    of the iteration is a real RCCL call and the run must still equal the reference's one-rank golden."""
    code = _RCCL_ONE.format(root=ROOT, tests=os.path.join(ROOT, "tests"))
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "RCCL1 OK" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]


@pytest.mark.parametrize("dist,n_it", [("NU", 50), ("HB", 50)])
def test_fciqmc_1e6_walkers_matches_oracle(oracle, dist, n_it):
    """BASELINE config 3 (fciqmc_mol, N2-shaped, 1e6 walkers): the device against the CPU restatement on the shared counter stream
    for 50 iterations at that size -- walker numbers, positions, spawn counts, shift, every stored value.  The population is grown
    on the device from 2e5 walkers on HF (300 iterations spread them over ~1e5 determinants) and multiplied up to >= 1e6 walkers, then both sides restart from that
    vector (DistVec::load semantics: non-zero entries into positions 0..n-1, shift 0, iteration 0)."""
    from fries_amd.engine import FriEngine
    mol = fcidump.synthetic("N2")
    par = dict(epsilon=0.02, target_walkers=1000000, max_dets=4000000, initiator=3, seed=1, distribution=dist)
    hf = (1 << (mol.n_elec // 2)) - 1
    hf = hf | (hf << mol.n_orb)
    eng = FriEngine(mol)
    eng.setup_fciqmc(ini=(np.array([hf], dtype=np.uint64), np.array([2.0e5])), **par)     # the reference's per-determinant spawn buffer holds 5e5 (fciqmc_mol.cpp:144)
    lg = eng.iterate_fciqmc(300)
    assert int(lg["err"].max()) == 0
    d, v = eng.vector()
    keep = v != 0
    d, v = d[keep], v[keep]
    v = v * np.ceil(1.0e6 / np.abs(v).sum())         # the same distribution with an integer multiple of the walkers: >= 1e6
    walkers = float(np.abs(v).sum())
    assert 1e6 <= walkers < 2.0e6 and d.size > 30000, (walkers, d.size)
    eng.vec_load(d, v)
    eng.restart(0, 0.0, 0.0, 0)
    orc = oracle.OracleFciqmc(mol, counter_rng=True, **par)
    orc.load(d, v, 0.0, 0.0, 0)
    assert eng.p_doub == orc.p_doub
    lg = eng.iterate_fciqmc(n_it)
    lo = orc.iterate(n_it)
    assert int(lg["err"].max()) == 0
    for f in ("n_nonz", "n_ini", "curr_size", "n_spawn"):
        assert np.array_equal(lg[f].astype(np.int64), lo[f].astype(np.int64)), (f, np.nonzero(lg[f].astype(np.int64) != lo[f].astype(np.int64))[0][:5])
    assert np.array_equal(lg["shift"], lo["shift"]) and np.array_equal(lg["norm"], lo["norm"]) and np.array_equal(lg["denom"], lo["denom"])
    assert np.all(np.abs(lg["numer"] - lo["numer"]) <= 1e-10 * np.maximum(1.0, np.abs(lo["numer"])))
    gd, gv = eng.vector()
    od, ov = orc.vector()
    assert gd.size == od.size and np.array_equal(gv, ov)
    nz = ov != 0
    assert np.array_equal(gd[nz], od[nz])
    print(dist, "walkers", int(np.abs(gv).sum()), "determinants", int(np.count_nonzero(gv)), "attempts/iteration", int(lg["n_attempts"][-1]), "spawns/iteration", int(lg["n_spawn"][-1]))
    eng.close()


def test_frisys_hh_budget_1e6_matches_reference():
    """The 1-D stand-in for BASELINE config 5 (frisys_hh; the reference has no 2-D lattice): L = 12 sites at half filling, budget
    vec_nonz = 1e6, from 100 x Neel through the start-up regime, the regime where the second compression's comb needs ~1e4
    repairs per stage (k_sys_prop) and into the compressed regime, against what the REAL reference logged
    (tests/golden/hh_l12_m1e6.traj, `ref_harness hh`): counts, norms, shifts every iteration, the digest of the vector every 20."""
    from fries_amd.engine import FriEngine
    g = golden_io.read_traj("hh_l12_m1e6")
    r = golden_io.manifest()["hh_scale_runs"]["hh_l12_m1e6"]
    eng = FriEngine(None)
    eng.setup_hh(n_elec=r["n_elec"], n_sites=r["n_sites"], eps=r["eps"], U=r["U"], omega=r["omega"], g=r["g"], gs_energy=r["gs_energy"], vec_nonz=r["vec_nonz"],
                 max_dets=r["max_dets"], target_norm=r["target_norm"], initiator=r["initiator"], seed=r["seed"])
    fails = []
    for row in g["rows"]:
        lg = eng.iterate_hh(1)[0]
        pin_replay._check_row(lg, row, fails, "hh")
        if row["it"] % 20 == 19 or row is g["rows"][-1]:
            d, v = eng.vector()
            if golden_io.vec_hash(d, v) != row["hash"]:
                fails.append(("hh", row["it"], "digest"))
        assert not fails, fails[:6]
    assert g["rows"][-1]["n_nonz"] > 500000 and max(x["num_success"] for x in g["rows"]) == r["vec_nonz"]
    eng.close()


def _read_ckpt(dirname, rank, n_orb):
    nb = (2 * n_orb + 7) // 8
    raw = np.fromfile(dirname + f"dets{rank}.dat", dtype=np.uint8)
    n = raw.size // nb
    vals = np.fromfile(dirname + f"vals{rank}.dat", dtype=np.float64)
    dets = np.zeros(n, dtype=np.uint64)
    for b in range(nb):
        dets |= raw.reshape(n, nb)[:, b].astype(np.uint64) << np.uint64(8 * b)
    return dets, vals[:n]


def _cli_base(r, fc, mol, out):
    from fries_amd import build
    return [build.DRIVER, "--fcidump_path", fc, "--point_group", mol.point_group, "--distribution", r["distribution"], "--vec_nonz", str(r["vec_nonz"]),
            "--mat_nonz", str(r["mat_nonz"]), "--max_dets", str(r["max_dets"]), "--target", repr(r["target_norm"]), "--initiator", repr(r["initiator"]),
            "--epsilon", repr(r["epsilon"]), "--seed", str(r["seed"]), "--max_iter", str(r["n_iter"]), "--result_dir", out]


@pytest.mark.parametrize("name", ["n2_m10000_unnorm_p2", "h2o_m5000_hb_p3"])
def test_cpp_driver_thread_ranks_match_mpiexec_goldens(name, tmp_path):
    """frisys_mol_hip --ranks P (P rank threads of one C++ process over the native local transport, no Python anywhere) against
    what every rank of the reference logged under mpiexec -n P: the projected energy the HF owner writes, every shift and norm,
    and every rank's final shard (dets<r>.dat / vals<r>.dat) by digest."""
    r = golden_io.manifest()["mpi_runs"][name]
    P = r["n_ranks"]
    mol = fcidump.synthetic(r["shape"])
    fc = str(tmp_path / "mol.FCIDUMP")
    fcidump.write_fcidump(fc, mol)
    out = str(tmp_path / "run") + "/"
    os.makedirs(out)
    res = subprocess.run(_cli_base(r, fc, mol, out) + ["--ranks", str(P)], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "Exception" not in res.stderr, res.stderr[-3000:]
    gs = [golden_io.read_traj(name, rank=k) for k in range(P)]
    rows = gs[gs[0]["hf_proc"]]["rows"]
    num = np.loadtxt(out + "projnum.txt"); den = np.loadtxt(out + "projden.txt"); nk = np.loadtxt(out + "nkept.txt")
    sh = np.loadtxt(out + "S.txt").reshape(-1); nm = np.loadtxt(out + "norm.txt").reshape(-1)
    assert num.size == r["n_iter"]
    for i, row in enumerate(rows):
        assert abs(num[i] / den[i] - row["numer"] / row["denom"]) < 1e-10 and int(nk[i]) == row["nkept"], i
    for k in range(r["n_iter"] // 10):
        assert sh[k] == rows[10 * k + 9]["shift"] and nm[k] == rows[10 * k + 9]["norm"], k
    for k in range(P):
        d, v = _read_ckpt(out, k, mol.n_orb)
        assert d.size == gs[k]["rows"][-1]["curr_size"] and golden_io.vec_hash(d, v) == gs[k]["rows"][-1]["hash"], k
    assert open(out + "dense.txt").read() == ",".join(["0"] * P) + "\n"


def test_frifull_cli_thread_ranks_match_mpiexec_golden(tmp_path):
    """frifull_mol_hip --ranks 2 (two rank threads of one C++ process over the native local transport) against what the HF owner of the
    reference's frifull_mol loop wrote under mpiexec -n 2: projected energy, preserved counts, shift and norm every ten iterations."""
    from fries_amd import build
    name = "full_ne_m300_p2"
    r = golden_io.manifest()["full_mpi_runs"][name]
    P = r["n_ranks"]
    mol = fcidump.synthetic(r["shape"])
    fc = str(tmp_path / "mol.FCIDUMP")
    fcidump.write_fcidump(fc, mol)
    out = str(tmp_path / "run") + "/"
    os.makedirs(out)
    cmd = [build.DRIVERS["frifull_mol_hip"], "--fcidump_path", fc, "--point_group", mol.point_group, "--epsilon", repr(r["epsilon"]), "--vec_nonz", str(r["vec_nonz"]),
           "--max_dets", str(r["max_dets"]), "--target", repr(r["target_norm"]), "--seed", str(r["seed"]), "--max_iter", str(r["n_iter"]), "--result_dir", out,
           "--ranks", str(P), "--spawn_cap", "3000000"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "Exception" not in res.stderr, res.stderr[-3000:]
    gs = [golden_io.read_traj(name, rank=k) for k in range(P)]
    rows = gs[gs[0]["hf_proc"]]["rows"]
    num = np.loadtxt(out + "projnum.txt"); den = np.loadtxt(out + "projden.txt"); nk = np.loadtxt(out + "nkept.txt")
    sh = np.loadtxt(out + "S.txt").reshape(-1); nm = np.loadtxt(out + "norm.txt").reshape(-1)
    assert num.size == r["n_iter"]
    for i, row in enumerate(rows):
        assert abs(num[i] - row["numer"]) <= 1e-10 * max(1.0, abs(row["numer"])) and abs(den[i] - row["denom"]) <= 1e-10 * abs(row["denom"]) and int(nk[i]) == row["nkept"], i
    for k in range(r["n_iter"] // 10):
        assert sh[k] == rows[10 * k + 9]["shift"] and nm[k] == rows[10 * k + 9]["norm"], k


def test_cpp_driver_rccl_process_rank(tmp_path):
    """frisys_mol_hip launched as a rank (RANK / WORLD_SIZE in the environment): librccl communicator from the id file in the
    result directory, every collective of the iteration a real RCCL call from C++; world of one here (one GPU per box)."""
    name = "n2_m10000_unnorm_ini0"
    r = golden_io.manifest()["runs"][name]
    g = golden_io.read_traj(name)
    mol = fcidump.synthetic(r["shape"])
    fc = str(tmp_path / "mol.FCIDUMP")
    fcidump.write_fcidump(fc, mol)
    out = str(tmp_path / "run") + "/"
    os.makedirs(out)
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    res = subprocess.run(_cli_base(r, fc, mol, out), capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0 and "Exception" not in res.stderr, res.stderr[-3000:]
    num = np.loadtxt(out + "projnum.txt"); den = np.loadtxt(out + "projden.txt")
    for i, row in enumerate(g["rows"]):
        assert abs(num[i] / den[i] - row["numer"] / row["denom"]) < 1e-10, i
    d, v = _read_ckpt(out, 0, mol.n_orb)
    assert golden_io.vec_hash(d, v) == g["rows"][-1]["hash"]
    assert not os.path.exists(out + ".rccl_id")


def test_fciqmc_cli_reads_the_references_checkpoint_and_its_own(tmp_path):
    """f-3: fciqmc_mol_hip --load_dir on a checkpoint the REFERENCE wrote (tests/golden/fciqmc_ne_ck: DistVec<int>::save + hash.dat +
    S.txt after 120 iterations of its loop) -- the walkers, the non-zero determinants and the shift must come back -- and on its own
    checkpoint: the restarted run starts from the walker number the first run ended with."""
    from fries_amd import build
    mol = fcidump.synthetic("Ne")
    fc = str(tmp_path / "mol.FCIDUMP")
    fcidump.write_fcidump(fc, mol)
    ck = os.path.join(golden_io.GOLD, "fciqmc_ne_ck") + "/"
    meta = open(ck + "meta.txt").read().split()
    ref_nonzero, ref_walkers = int(meta[meta.index("nonzero") + 1]), int(meta[meta.index("walkers") + 1])
    exe = build.DRIVERS["fciqmc_mol_hip"]
    base = [exe, "--fcidump_path", fc, "--point_group", mol.point_group, "--distribution", "NU", "--target", "5000", "--max_dets", "20000", "--epsilon", "0.002",
            "--initiator", "3", "--seed", "5"]
    out1 = str(tmp_path / "r1") + "/"
    os.makedirs(out1)
    res = subprocess.run(base + ["--max_iter", "1", "--result_dir", out1, "--load_dir", ck], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "Exception" not in res.stderr, res.stderr[-2000:]
    assert f"loaded {ref_walkers} walkers" in res.stdout, res.stdout[-500:]
    assert np.array_equal(np.fromfile(out1 + "hash.dat", dtype=np.uint32), np.fromfile(ck + "hash.dat", dtype=np.uint32)[:2 * mol.n_orb])
    # own checkpoint round trip
    out2, out3 = str(tmp_path / "r2") + "/", str(tmp_path / "r3") + "/"
    os.makedirs(out2); os.makedirs(out3)
    res = subprocess.run(base + ["--max_iter", "150", "--result_dir", out2], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "Exception" not in res.stderr, res.stderr[-2000:]
    nb = (2 * mol.n_orb + 7) // 8
    n = os.path.getsize(out2 + "dets0.dat") // nb
    iv = np.fromfile(out2 + "vals0.dat", dtype=np.int32)
    assert iv.size == n and np.abs(iv).sum() > 100
    res = subprocess.run(base + ["--max_iter", "10", "--result_dir", out3, "--load_dir", out2], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "Exception" not in res.stderr, res.stderr[-2000:]
    assert f"loaded {int(np.abs(iv).sum())} walkers" in res.stdout
    assert int(np.loadtxt(out3 + "nnonz.txt").reshape(-1)[0]) > 0


def test_frifull_cli_from_legacy_hf_directory(tmp_path):
    """f-3: frifull_mol_hip --hf_path (the legacy HF-output directory the reference's frifull_mol takes) reproduces the reference's
    frifull_mol trajectory of tests/golden/full_ne_m300.traj, like the FCIDUMP route."""
    from fries_amd import build
    name = "full_ne_m300"
    r = golden_io.manifest()["full_runs"][name]
    g = golden_io.read_traj(name)
    mol = fcidump.synthetic(r["shape"])
    hf = str(tmp_path / "hf") + "/"
    out = str(tmp_path / "out") + "/"
    os.makedirs(hf); os.makedirs(out)
    fcidump.write_hf_dir(hf, mol, eps=r["epsilon"])
    cmd = [build.DRIVERS["frifull_mol_hip"], "--hf_path", hf, "--vec_nonz", str(r["vec_nonz"]), "--max_dets", str(r["max_dets"]), "--target", repr(r["target_norm"]),
           "--seed", str(r["seed"]), "--max_iter", str(r["n_iter"]), "--result_dir", out]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "Exception" not in res.stderr, res.stderr[-2000:]
    num = np.loadtxt(out + "projnum.txt"); den = np.loadtxt(out + "projden.txt"); nk = np.loadtxt(out + "nkept.txt")
    assert num.size == r["n_iter"]
    for i, row in enumerate(g["rows"]):
        assert abs(num[i] / den[i] - row["numer"] / row["denom"]) < 1e-10 and den[i] == row["denom"] and int(nk[i]) == row["nkept"], i


@pytest.mark.parametrize("name", sorted(golden_io.manifest().get("dense_mpi_runs", {})))
def test_dense_space_over_ranks_matches_reference(name):
    """--det_space over ranks (rank threads of this process on the native local transport): every rank keeps its share of the dense
    determinants in front of its shard, the dense block of H is exchanged as a perform_add of its own, tot_dense_h and dense_norm are
    sums over the ranks -- per rank against what that rank of the real reference logged under `mpiexec -n P`."""
    import threading
    from fries_amd.comm import LocalGroup
    from fries_amd.engine import FriEngine
    r = golden_io.manifest()["dense_mpi_runs"][name]
    P = r["n_ranks"]
    space = np.array([int(x) for x in open(os.path.join(golden_io.GOLD, r["det_space"])).read().split()], dtype=np.uint64)
    mol = fcidump.synthetic(r["shape"])
    grp = LocalGroup(P, r["mat_nonz"])
    comms = [grp.comm(k, 0) for k in range(P)]
    out = [None] * P

    def work(k):
        fails = []
        try:
            g = golden_io.read_traj(name, rank=k)
            eng = FriEngine(mol, device=0, comm=comms[k])
            eng.setup(epsilon=r["epsilon"], vec_nonz=r["vec_nonz"], mat_nonz=r["mat_nonz"], max_dets=r["max_dets"], target_norm=r["target_norm"],
                      initiator=r["initiator"], seed=r["seed"], distribution=r["distribution"], det_space=space)
            for row in g["rows"]:
                lg = eng.iterate(1)[0]
                pin_replay._check_row(lg, row, fails, "dense")
                if row["it"] % 10 == 9 or row is g["rows"][-1]:
                    d, v = eng.vector()
                    if golden_io.vec_hash(d, v) != row["hash"]:
                        fails.append(("dense", row["it"], "digest"))
                if len(fails) > 6:
                    break
            eng.close()
        except Exception as e:      # the other ranks then time out in their next collective instead of hanging
            fails.append(("exception", repr(e)))
        out[k] = fails

    th = [threading.Thread(target=work, args=(k,)) for k in range(P)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    grp.destroy()
    for k in range(P):
        assert not out[k], (name, k, out[k][:6])
