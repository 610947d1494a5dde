"""Ad-hoc: grow an FCIQMC population on the GPU and time iterations at scale."""
import os, sys, time
_TESTS = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _TESTS); sys.path.insert(0, os.path.dirname(_TESTS))      # tests/ (golden_io, oracle_lib) and the repository root (bench, fries_amd)
import numpy as np
from fries_amd import fcidump
from fries_amd.engine import FriEngine
shape = sys.argv[1] if len(sys.argv) > 1 else "N2"
eps = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
target = int(sys.argv[3]) if len(sys.argv) > 3 else 1000000
fp = len(sys.argv) > 4 and sys.argv[4] == "fp"       # fciqmc_fp_mol: real-valued walkers
mol = fcidump.synthetic(shape)
eng = FriEngine(mol)
eng.setup_fciqmc(epsilon=eps, target_walkers=target, max_dets=8 * target, initiator=3, seed=1, fp=fp)
t0 = time.time(); it = 0
while it < 40000:
    lg = eng.iterate_fciqmc(500); it += 500
    d, v = eng.vector()
    w = np.abs(v).sum()
    print(it, "walkers", int(w), "n_nonz", int(lg["n_nonz"][-1]), "attempts", int(lg["n_attempts"][-1]), "spawns", int(lg["n_spawn"][-1]), "shift", float(lg["shift"][-1]),
          "en", float(lg["numer"][-1] / lg["denom"][-1]), "%.1f it/s" % (500 / (time.time() - t0)), flush=True)
    t0 = time.time()
    if w >= target and it >= 2000 and lg["shift"][-1] != 0:
        break
t0 = time.time(); lg = eng.iterate_fciqmc(200); dt = time.time() - t0
print("steady: %.1f it/s, %.3g attempts/s" % (200 / dt, lg["n_attempts"].sum() / dt))
# the same distribution with `mult` times the walkers (DistVec::load of a scaled vector), to time a 1e6-walker population
d, v = eng.vector()
w = np.abs(v).sum()
mult = max(1, int(round(target / w)))
keep = v != 0
eng.vec_load(d[keep], v[keep] * mult)
lg = eng.iterate_fciqmc(50)
t0 = time.time(); lg = eng.iterate_fciqmc(200); dt = time.time() - t0
d, v = eng.vector()
print("x%d: walkers %d n_nonz %d: %.1f it/s, %.3g attempts/s, %.3g spawns/s" % (mult, int(np.abs(v).sum()), int(lg["n_nonz"][-1]), 200 / dt, lg["n_attempts"].sum() / dt, lg["n_spawn"].sum() / dt))
if os.environ.get("FQ_PROF") == "1":          # where a 1e6-walker iteration goes (HIP events per kernel)
    eng.prof_enable(True)
    eng.iterate_fciqmc(100)
    for name, (ms, calls) in sorted(eng.prof_report().items(), key=lambda kv: -kv[1][0])[:14]:
        print(f"  {name:24s} {ms / 100:9.4f} ms per iteration  {calls / 100:6.1f} calls  {1e3 * ms / max(calls, 1):9.1f} us each")
