"""Ad-hoc: apply_HBPP_piv (pivotal compression of every HB-PP factor) on a full vector: device against the CPU restatement on the same
vector and generator, per-kernel time.  usage: gpu_hbpiv_scale.py [m] [n_samp] [n_iter] [HB|HB_unnorm]"""
import os, sys, time
_TESTS = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _TESTS); sys.path.insert(0, os.path.dirname(_TESTS))      # tests/ (golden_io, oracle_lib) and the repository root (bench, fries_amd)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # tests/ (golden_io, oracle_lib)
import numpy as np
from fries_amd import fcidump
from fries_amd.engine import FriEngine
import oracle_lib

m = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
n_samp = int(sys.argv[2]) if len(sys.argv) > 2 else m
n_iter = int(sys.argv[3]) if len(sys.argv) > 3 else 12
dist = sys.argv[4] if len(sys.argv) > 4 else "HB_unnorm"
mol = fcidump.synthetic("N2")
par = dict(epsilon=0.01, vec_nonz=m, mat_nonz=m, max_dets=8 * m, target_norm=0.3 * m, initiator=0.0, seed=5, distribution=dist)
eng = FriEngine(mol)
eng.setup(**par)
eng.iterate(n_iter, want_logs=False)
d, v = eng.vector()
nz = v != 0
d, v = d[nz], v[nz]
print(f"vector: {d.size} non-zero elements after {n_iter} iterations, budget {n_samp}", flush=True)
eng.vec_load(d, v)
orc = oracle_lib.OracleFrisys(mol, **{k: par[k] for k in par})
orc.vec_load(d, v)
for rep in range(3):
    eng.restart(123)
    if rep == 2:
        eng.prof_enable(True)
    t0 = time.time()
    pos, orbs, vals, sl = eng.apply_hbpp_piv(n_samp)
    dt = time.time() - t0
    print(f"rep {rep}: {dt * 1e3:.2f} ms wall, {len(pos)} samples, stage lengths {sl.tolist()}, pivotal stats {eng.piv_stats()}", flush=True)
rep = eng.prof_report()
tot = sum(ms for ms, _ in rep.values())
print(f"kernel time {tot:.3f} ms")
for k, (ms, calls) in sorted(rep.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  {k:20s} {ms:8.3f} ms  {calls} calls")
orc.restart(123)
t0 = time.time()
opos, oorbs, ovals, osl = orc.apply_hbpp_piv(n_samp)
print(f"CPU restatement (1 core): {(time.time() - t0) * 1e3:.1f} ms")
same = len(pos) == len(opos) and np.array_equal(pos, opos) and np.array_equal(orbs, oorbs) and vals.tobytes() == ovals.tobytes() and sl.tolist() == osl.tolist()
print("identical to the CPU restatement:", bool(same), "next draw equal:", eng.next_draw() == orc.next_draw())
sys.exit(0 if same else 1)
