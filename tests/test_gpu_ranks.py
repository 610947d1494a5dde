"""-m gpu: the hash-sharded engine with 2, 3 and 4 ranks against what every rank of the real reference logged
under `mpiexec -n P` (tests/golden/*.traj.r<rank>).  The ranks are separate processes sharing this box's GPU and
talking through torch.distributed (gloo here; bench.py uses nccl = RCCL with one GPU per rank)."""
import json
import os
import socket
import subprocess
import sys

import pytest

import golden_io

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_ranks(cmd, env, timeout):
    """Runs the launcher in its own process group and kills the whole group on timeout (a stuck rank must fail the
    test, never hang it)."""
    import signal
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGKILL)
        out, err = proc.communicate()
        pytest.fail(f"ranks did not finish within {timeout} s\n" + out[-2000:] + err[-4000:])
    return subprocess.CompletedProcess(cmd, proc.returncode, out, err)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


_RANK_RUNS = dict(golden_io.manifest()["mpi_runs"], **{k: v for k, v in golden_io.manifest()["hh_runs"].items() if v["n_ranks"] > 1},
                  **{k: v for k, v in golden_io.manifest()["hhfull_runs"].items() if v["n_ranks"] > 1})


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(_RANK_RUNS))
def test_sharded_engine_matches_reference_ranks(name, tmp_path):
    r = _RANK_RUNS[name]
    P = r["n_ranks"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={P}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "ranks_worker.py"), name, "gloo", str(tmp_path)]
    res = _run_ranks(cmd, env, 300)
    reports = []
    for k in range(P):
        fn = tmp_path / f"rank{k}.json"
        assert fn.exists(), res.stdout[-2000:] + res.stderr[-4000:]
        reports.append(json.loads(fn.read_text()))
    for rep in reports:
        assert rep["ok"], rep["fails"]
        assert rep["iters"] == r["n_iter"] and rep["n_alltoallv"] == r["n_iter"]
    assert res.returncode == 0, res.stderr[-4000:]


@pytest.mark.gpu
def test_sharded_engine_with_the_norms_in_a_message_of_their_own(tmp_path):
    """FRIES_FKS_NO_MERGED_NORM=1: the ranks' remaining norms gathered after the host has seen the closing pass's flag (two all-gathers per stage, the
    sequence until round 3) instead of riding with the flag -- the same trajectories per rank."""
    name = "n2_m10000_unnorm_p2"
    r = _RANK_RUNS[name]
    P = r["n_ranks"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", FRIES_FKS_NO_MERGED_NORM="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={P}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "ranks_worker.py"), name, "gloo", str(tmp_path)]
    res = _run_ranks(cmd, env, 300)
    for k in range(P):
        fn = tmp_path / f"rank{k}.json"
        assert fn.exists(), res.stdout[-2000:] + res.stderr[-4000:]
        rep = json.loads(fn.read_text())
        assert rep["ok"], rep["fails"]
    assert res.returncode == 0, res.stderr[-4000:]


@pytest.mark.gpu
def test_one_rank_communicator_over_rccl(tmp_path):
    """backend "nccl" (RCCL) with a world of one: every all-gather and the spawn all-to-all run as real RCCL
    collectives enqueued under the engine's stream, and the run must still equal the one-rank golden."""
    name = "n2_m10000_unnorm_ini0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "ranks_worker.py"), name, "nccl", str(tmp_path)]
    res = _run_ranks(cmd, dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1"), 300)
    fn = tmp_path / "rank0.json"
    assert fn.exists(), res.stdout[-2000:] + res.stderr[-4000:]
    rep = json.loads(fn.read_text())
    assert rep["ok"], rep["fails"]
    assert rep["n_alltoallv"] == rep["iters"] and rep["n_allgather"] > 10 * rep["iters"]


@pytest.mark.gpu
@pytest.mark.parametrize("name,budget", [("n2_m10000_unnorm_p2", 4000), ("h2o_m5000_hb_p3", 2500)])
def test_apply_hbpp_piv_over_ranks(name, budget, tmp_path):
    """apply_HBPP_piv over a sharded vector: the five pivotal compressions inside are collective (preservation rounds, the budget
    apportioned by rank 0, per-shard sampling); every rank's samples against the in-process rank oracle, bit for bit."""
    r = golden_io.manifest()["mpi_runs"][name]
    P = r["n_ranks"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", FRIES_RANKS_HBPIV=str(budget))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={P}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "ranks_worker.py"), name, "gloo", str(tmp_path), "12"]
    res = _run_ranks(cmd, env, 300)
    total = 0
    for k in range(P):
        fn = tmp_path / f"rank{k}.json"
        assert fn.exists(), res.stdout[-2000:] + res.stderr[-4000:]
        rep = json.loads(fn.read_text())
        assert rep["ok"], rep["fails"]
        total += rep["hbpiv_stages"][0]
    assert 0 < total <= budget          # what the first compression kept over all ranks
    assert res.returncode == 0, res.stderr[-4000:]


@pytest.mark.gpu
@pytest.mark.parametrize("name,budget", [("n2_m10000_unnorm_p2", 3000), ("h2o_m5000_hb_p3", 1500)])
def test_pivotal_compression_over_ranks(name, budget, tmp_path):
    """compress_vecs over a sharded vector (piv_comp_parallel with piv_budget apportioning the samples among the ranks,
    adjust_probs re-weighting each shard, per-rank pivotal sampling) against the in-process rank oracle, shard by shard."""
    r = golden_io.manifest()["mpi_runs"][name]
    P = r["n_ranks"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", FRIES_RANKS_PIV=str(budget))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={P}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "ranks_worker.py"), name, "gloo", str(tmp_path), "12"]
    res = _run_ranks(cmd, env, 300)
    total = 0
    for k in range(P):
        fn = tmp_path / f"rank{k}.json"
        assert fn.exists(), res.stdout[-2000:] + res.stderr[-4000:]
        rep = json.loads(fn.read_text())
        assert rep["ok"], rep["fails"]
        total += rep["piv_nonzero"]
    assert 0 < total <= budget
    assert res.returncode == 0, res.stderr[-4000:]


_FQ_RANK_RUNS = dict(golden_io.manifest()["fciqmc_mpi_runs"], **golden_io.manifest().get("fciqmc_fp_mpi_runs", {}), **golden_io.manifest().get("multi_mpi_runs", {}))


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(_FQ_RANK_RUNS))
def test_fciqmc_over_ranks(name, tmp_path):
    """fciqmc_mol, fciqmc_fp_mol (fciqmc_fp_*: real-valued walkers) and frimulti_mol (multi_*) hash-sharded over 2 and 3 ranks (near-uniform and heat-bath
    generators): one all-to-all of the spawns per iteration with initiator and non-initiator spawns in their original order, walker
    totals and projections summed in rank order; every rank's shard equals the in-process rank oracle's (which is pinned against the
    reference under mpiexec in the CPU suite)."""
    r = _FQ_RANK_RUNS[name]
    P = r["n_ranks"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={P}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "ranks_worker.py"), name, "gloo", str(tmp_path)]
    res = _run_ranks(cmd, env, 300)
    for k in range(P):
        fn = tmp_path / f"rank{k}.json"
        assert fn.exists(), res.stdout[-2000:] + res.stderr[-4000:]
        rep = json.loads(fn.read_text())
        assert rep["ok"], rep["fails"]
        assert rep["n_alltoallv"] == rep["iters"]
    assert res.returncode == 0, res.stderr[-4000:]


@pytest.mark.gpu
def test_fciqmc_one_rank_communicator_over_rccl(tmp_path):
    """backend "nccl" (RCCL) with a world of one: the order-preserving spawn exchange and the walker-total all-gather run as real
    RCCL collectives under the engine's stream, and the run must equal the one-rank oracle."""
    name = "fciqmc_n2_p2"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "ranks_worker.py"), name, "nccl", str(tmp_path)]
    res = _run_ranks(cmd, dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1"), 300)
    fn = tmp_path / "rank0.json"
    assert fn.exists(), res.stdout[-2000:] + res.stderr[-4000:]
    rep = json.loads(fn.read_text())
    assert rep["ok"], rep["fails"]
    assert rep["n_alltoallv"] == rep["iters"]
