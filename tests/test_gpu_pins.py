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
