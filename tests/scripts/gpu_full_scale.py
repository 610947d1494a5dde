"""Ad-hoc: frifull_mol on the device at N2, vec_nonz = 1e3: excitations added per second, per-kernel time, and the CPU
restatement (1 core) on the same run beside it."""
import os, sys, time
_TESTS = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _TESTS); sys.path.insert(0, os.path.dirname(_TESTS))      # tests/ (golden_io, oracle_lib) and the repository root (bench, fries_amd)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # tests/ (golden_io, oracle_lib)
import numpy as np
from fries_amd import fcidump
from fries_amd.engine import FriEngine
import oracle_lib

m = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n_it = int(sys.argv[2]) if len(sys.argv) > 2 else 8
mol = fcidump.synthetic("N2")
par = dict(epsilon=0.01, vec_nonz=m, max_dets=8000000, target_norm=float(m), seed=7)
eng = FriEngine(mol)
eng.setup_full(**par)
lg = eng.iterate_full(6)            # grow: 1 -> ~1.7e3 -> ... determinants before compression bites
eng.prof_enable(True)
t0 = time.time()
lg = eng.iterate_full(n_it)
dt = time.time() - t0
adds = int(lg["num_success"].astype(np.int64).sum())
print(f"GPU: {n_it} iterations in {dt:.3f} s = {n_it / dt:.2f} it/s, {adds / dt / 1e6:.1f} M excitations/s, stored determinants {int(lg['curr_size'][-1])}, nkept {int(lg['nkept'][-1])}")
rep = eng.prof_report()
tot = sum(v[0] for v in rep.values())
for k, (ms, calls) in sorted(rep.items(), key=lambda kv: -kv[1][0])[:10]:
    print(f"  {k:18s} {ms / n_it:9.3f} ms/iter  {calls / n_it:7.1f} calls/iter  {100 * ms / tot:5.1f} %")
orc = oracle_lib.OracleFull(mol, **par)
orc.iterate(6)
t0 = time.time()
lo = orc.iterate(3)
dto = time.time() - t0
print(f"CPU restatement (1 core): {3 / dto:.3f} it/s, {int(lo['num_success'].astype(np.int64).sum()) / dto / 1e6:.2f} M excitations/s")
print("identical so far:", bool(np.array_equal(lo["norm"], lg["norm"][:3]) and np.array_equal(lo["curr_size"], lg["curr_size"][:3])))
