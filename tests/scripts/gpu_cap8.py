"""Ad-hoc: the allocations one rank of the 8-GPU bench makes (vec_nonz = mat_nonz = 8e6 global, max_dets = 4e6 per rank) on one GPU,
with a 1e6-element shard loaded: setup, a few iterations, memory in use."""
import os, sys, time
_TESTS = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _TESTS); sys.path.insert(0, os.path.dirname(_TESTS))      # tests/ (golden_io, oracle_lib) and the repository root (bench, fries_amd)
import numpy as np
import bench
from fries_amd import fcidump
from fries_amd.engine import FriEngine

mol = fcidump.synthetic("N2")
m = 1000000
dets, vals = bench.build_state(mol, m, 4 * m, 20250215, 0, None, None)
print("state", dets.size, flush=True)
eng = FriEngine(mol)
t0 = time.time()
eng.setup(epsilon=0.01, vec_nonz=8 * m, mat_nonz=8 * m, max_dets=4 * m, target_norm=float(8 * m), initiator=1.0, seed=1, distribution="HB_unnorm")
print(f"setup {time.time() - t0:.2f} s", flush=True)
eng.vec_load(dets, vals)
eng.restart(777)
lg = eng.iterate(5)
print("n_nonz", lg["n_nonz"].tolist(), "num_success", lg["num_success"].tolist(), "err", lg["err"].tolist(), flush=True)
t0 = time.time()
eng.iterate(10, want_logs=False)
eng.vec_info()
print(f"{10 / (time.time() - t0):.1f} it/s with everything preserved (budget 8e6 > elements)")
