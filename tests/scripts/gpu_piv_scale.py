"""Ad-hoc: pivotal compression (fries_compress_vec_piv) at 1e6 stored elements: per-kernel time and the CPU restatement beside it."""
import os, sys, time
_TESTS = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _TESTS); sys.path.insert(0, os.path.dirname(_TESTS))      # tests/ (golden_io, oracle_lib) and the repository root (bench, fries_amd)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # tests/ (golden_io, oracle_lib)
import numpy as np
from fries_amd import fcidump
from fries_amd.engine import FriEngine
import oracle_lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
budget = int(sys.argv[2]) if len(sys.argv) > 2 else 400000
mol = fcidump.synthetic("N2")
rng = np.random.RandomState(4)
vals = np.exp(6 * rng.random_sample(n)) * np.where(rng.random_sample(n) < 0.5, 1.0, -1.0)
dets = (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(2654435761)) | np.uint64(1 << 40)
eng = FriEngine(mol)
eng.setup(epsilon=0.01, vec_nonz=100, mat_nonz=100, max_dets=n + 1000, seed=3)
for rep in range(3):
    eng.vec_load(dets, vals)
    eng.restart(77)
    if rep == 2:
        eng.prof_enable(True)
    t0 = time.time()
    eng.compress_vec_piv(budget)
    dt = time.time() - t0
    print(f"rep {rep}: {dt * 1e3:.2f} ms wall, n_nonz after {eng.vec_info()[1]}")
rep = eng.prof_report()
for k, (ms, calls) in sorted(rep.items(), key=lambda kv: -kv[1][0])[:12]:
    print(f"  {k:20s} {ms:8.3f} ms  {calls} calls")
t0 = time.time()
ov, ofl, _ = oracle_lib.piv_comp(vals, budget, 77)
print(f"CPU restatement (1 core): {(time.time() - t0) * 1e3:.1f} ms")
_, v = eng.vector()
print("identical:", bool(np.array_equal(v, ov)))
