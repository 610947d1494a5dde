"""Ad-hoc: frimulti_mol on the device at scale: growth from 100 x HF, then timed iterations; per-kernel time.
usage: gpu_multi_scale.py [vec_nonz] [mat_nonz] [n_grow] [n_timed]"""
import os, sys, time
_TESTS = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _TESTS); sys.path.insert(0, os.path.dirname(_TESTS))      # tests/ (golden_io, oracle_lib) and the repository root (bench, fries_amd)
import numpy as np
from fries_amd import fcidump
from fries_amd.engine import FriEngine

m = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
mat = int(sys.argv[2]) if len(sys.argv) > 2 else m
n_grow = int(sys.argv[3]) if len(sys.argv) > 3 else 60
n_timed = int(sys.argv[4]) if len(sys.argv) > 4 else 50
mol = fcidump.synthetic("N2")
eng = FriEngine(mol)
eng.setup_multi(epsilon=0.01, vec_nonz=m, mat_nonz=mat, max_dets=4 * m, target_norm=float(m) / 4, initiator=0.0, seed=5)
t0 = time.time()
done = 0
while done < n_grow:
    lg = eng.iterate_multi(10)
    done += 10
    print(f"{done} n_nonz {int(lg['n_nonz'][-1])} n_spawn {int(lg['n_spawn'][-1])} samples {int(lg['n_attempts'][-1])} norm {float(lg['norm'][-1]):.6g} shift {float(lg['shift'][-1]):.4f} "
          f"energy {float(lg['numer'][-1] / lg['denom'][-1]):.6f} err {int(lg['err'].max())}", flush=True)
    if int(lg["n_nonz"][-1]) >= m:
        break
print(f"growth: {time.time() - t0:.1f} s")
eng.prof_enable(True)
t0 = time.time()
lg = eng.iterate_multi(n_timed)
dt = time.time() - t0
print(f"timed: {n_timed / dt:.1f} it/s, {lg['n_attempts'].sum() / dt:.3e} samples/s, n_nonz {int(lg['n_nonz'][-1])}, err {int(lg['err'].max())}")
rep = eng.prof_report()
tot = sum(ms for ms, _ in rep.values())
print(f"kernel time {tot / n_timed:.3f} ms per iteration")
for k, (ms, calls) in sorted(rep.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  {k:20s} {ms / n_timed:8.3f} ms/it  {calls / n_timed:.1f} calls/it")
