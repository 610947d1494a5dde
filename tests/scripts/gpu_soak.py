"""Ad-hoc: long runs at bench size -- no replay / repair / capacity error, population under shift control, timings stable."""
import os, sys, time
_TESTS = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _TESTS); sys.path.insert(0, os.path.dirname(_TESTS))      # tests/ (golden_io, oracle_lib) and the repository root (bench, fries_amd)
import numpy as np
import bench
from fries_amd import fcidump
from fries_amd.engine import FriEngine
m = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
n_it = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
mol = fcidump.synthetic("N2")
dets, vals = bench.build_state(mol, m, 4 * m, 20250215, 0, None, None)
eng = FriEngine(mol)
eng.setup(epsilon=0.01, vec_nonz=m, mat_nonz=m, max_dets=4 * m, target_norm=float(m), initiator=1.0, seed=20250215, distribution="HB_unnorm")
eng.vec_load(dets, vals)
eng.restart(777, 0.0, 0.0, 0)
t_all = time.time()
for blk in range(n_it // 500):
    t0 = time.time()
    lg = eng.iterate(500)
    dt = time.time() - t0
    en = lg["numer"] / lg["denom"]
    print(f"iterations {500 * blk}-{500 * blk + 499}: {500 / dt:.1f} it/s, norm {lg['norm'][-1]:.4g}, shift {lg['shift'][-1]:.5f}, n_nonz {int(lg['n_nonz'][-1])}, "
          f"curr_size {int(lg['curr_size'][-1])}, E {en.mean():.6f} +- {en.std() / np.sqrt(en.size):.6f}, err {int(lg['err'].max())}", flush=True)
print("total", time.time() - t_all, "s; counters", eng.counters())
