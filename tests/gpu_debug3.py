import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fries_amd import fcidump
from fries_amd.engine import FriEngine
import oracle_lib
shape, m, n_pre, dist = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
mol = fcidump.synthetic(shape)
par = dict(epsilon=0.01, vec_nonz=m, mat_nonz=m, max_dets=10 * m, target_norm=m / 2, initiator=1.0, seed=20250215, distribution=dist)
eng = FriEngine(mol); orc = oracle_lib.OracleFrisys(mol, **par); eng.setup(**par)
eng.iterate(n_pre); orc.iterate(n_pre)
gd, gv = eng.vector(); cd, cv = orc.vector()
print("pre-state equal:", np.array_equal(gd, cd), np.array_equal(gv, cv), np.abs(gv-cv).max())
eng.vec_load(cd, cv); orc.vec_load(cd, cv)
# replicate the rn the drivers would draw next: both mt streams are aligned, but apply API takes rn explicitly
rng = np.random.RandomState(11)
for trial in range(4):
    rn = rng.random_sample(5)
    gp, go, gvv, cl = eng.apply_hbpp_sys(m, rn)
    cp, co, cvv = orc.apply_hbpp_sys(m, rn)
    n = min(gp.size, cp.size)
    neq = (gp[:n] != cp[:n]) | (go[:n] != co[:n]).any(axis=1)
    print("trial", trial, "n", gp.size, cp.size, list(cl), "first mismatch", (np.nonzero(neq)[0][:3] if neq.any() else None))
    if gp.size != cp.size or neq.any():
        i = np.nonzero(neq)[0][0] if neq.any() else n - 1
        for k in range(max(0, i - 2), min(n, i + 3)):
            print("   ", k, "gpu", gp[k], go[k], gvv[k], "| cpu", cp[k], co[k], cvv[k])
