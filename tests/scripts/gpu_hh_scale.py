"""Ad-hoc: frisys_hh on the GPU at a large budget (BASELINE config 5 is a 4x4 lattice the reference cannot run; this is its 1-D model)."""
import os, sys, time
_TESTS = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _TESTS); sys.path.insert(0, os.path.dirname(_TESTS))      # tests/ (golden_io, oracle_lib) and the repository root (bench, fries_amd)
import numpy as np
from fries_amd.engine import FriEngine
L = int(sys.argv[1]) if len(sys.argv) > 1 else 12
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
gs = float(sys.argv[3]) if len(sys.argv) > 3 else -7.5     # -1: the population grows (the golden hh_l12_m1e6 run); -7.5: it dies out at 900 states
step = int(sys.argv[4]) if len(sys.argv) > 4 else 50
eng = FriEngine(None)
eng.setup_hh(n_elec=L, n_sites=L, eps=0.005, U=4.0, omega=1.0, g=0.7, gs_energy=gs, vec_nonz=m, max_dets=8 * m, target_norm=float(m) / 4, initiator=1.0, seed=3)
t0 = time.time(); it = 0
while it < 3000:
    lg = eng.iterate_hh(step); it += step
    print(it, "n_nonz", int(lg["n_nonz"][-1]), "num_success", int(lg["num_success"][-1]), "norm", float(lg["norm"][-1]), "shift", float(lg["shift"][-1]), "%.1f it/s" % (step / (time.time() - t0)), flush=True)
    t0 = time.time()
    if lg["n_nonz"][-1] > 0.9 * m and lg["shift"][-1] != 0:
        break
t0 = time.time(); lg = eng.iterate_hh(100); dt = time.time() - t0
print("steady: n_nonz %d, %.1f it/s, %.3g samples/s" % (int(lg["n_nonz"][-1]), 100 / dt, lg["num_success"].sum() / dt))
