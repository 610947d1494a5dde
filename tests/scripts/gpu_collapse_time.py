"""The start-up ("collapse") regime at the bench's size: the filler run from the Hartree-Fock determinant with a budget of m, per-iteration
wall time and a digest of the vector at the end (scratch; run once per setting of FRIES_FKS_SEQ_WALK to compare the one-wave walk with the
parallel form of the in-order sweep, fks_seq.hpp)."""
import os, sys, time, hashlib
_TESTS = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _TESTS); sys.path.insert(0, os.path.dirname(_TESTS))      # tests/ (golden_io, oracle_lib) and the repository root (bench, fries_amd)
import numpy as np
from fries_amd import fcidump
from fries_amd.engine import FriEngine
m = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
n_it = int(sys.argv[2]) if len(sys.argv) > 2 else 30
mol = fcidump.synthetic("N2")
eng = FriEngine(mol)
eng.setup(epsilon=0.01, vec_nonz=m, mat_nonz=m, max_dets=4 * m, target_norm=0.0, initiator=0.0, seed=20250215, distribution="HB_unnorm")
ts = []
prof = os.environ.get("COLLAPSE_PROF") == "1"
if prof:
    eng.prof_enable(True)
for it in range(n_it):
    t0 = time.perf_counter()
    lg = eng.iterate(1)
    ts.append((time.perf_counter() - t0) * 1e3)
    print(f"it {it:3d}  {ts[-1]:9.2f} ms  n_nonz {int(lg['n_nonz'][-1]):8d}  nkept {int(lg['nkept'][-1]):8d}", flush=True)
if prof:
    rep = eng.prof_report()
    for name, (ms, calls) in sorted(rep.items(), key=lambda kv: -kv[1][0])[:14]:
        print(f"  {name:24s} {ms:9.2f} ms  {calls:7d} calls  {1e3 * ms / max(calls, 1):9.1f} us each")
d, v = eng.vector()
k = v != 0
print("total ms", round(sum(ts), 1), " digest", hashlib.sha256(d[k].tobytes() + v[k].tobytes()).hexdigest()[:16], " walk_only", os.environ.get("FRIES_FKS_SEQ_WALK", "0"))
eng.close()
