import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fries_amd import fcidump
from fries_amd.engine import FriEngine
import oracle_lib

shape = sys.argv[1] if len(sys.argv) > 1 else "Ne"
m = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
n_pre = int(sys.argv[3]) if len(sys.argv) > 3 else 5
mol = fcidump.synthetic(shape)
par = dict(epsilon=0.01, vec_nonz=m, mat_nonz=m, max_dets=10 * m, target_norm=m / 2, initiator=1.0, seed=20250215, distribution="HB_unnorm")
eng = FriEngine(mol); orc = oracle_lib.OracleFrisys(mol, **par); eng.setup(**par)
orc.iterate(n_pre)
d, v = orc.vector()
eng.vec_load(d, v); orc.vec_load(d, v)
rng = np.random.RandomState(7)
for trial in range(6):
    rn = rng.random_sample(5)
    gp, go, gv, cl = eng.apply_hbpp_sys(m, rn)
    cp, co, cv = orc.apply_hbpp_sys(m, rn)
    n = min(gp.size, cp.size)
    same = gp.size == cp.size and np.array_equal(gp, cp) and np.array_equal(go, co)
    print("trial", trial, "n", gp.size, cp.size, "comp_len", list(cl), "idx same", same)
    if not same:
        bad = np.nonzero((gp[:n] != cp[:n]) | (go[:n] != co[:n]).any(axis=1))[0]
        print(" first diffs at", bad[:10])
        for b in bad[:5]:
            print("  ", b, "gpu", gp[b], go[b], gv[b], "cpu", cp[b], co[b], cv[b])
    else:
        rel = np.abs(gv - cv) / np.abs(cv)
        print("  max rel val diff", rel.max(), "n_bitdiff", int((gv != cv).sum()))
