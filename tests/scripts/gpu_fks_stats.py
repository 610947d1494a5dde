"""FRIES_DBG=3: per-replay statistics of the find_keep_sub replay at the bench's workload (scratch)"""
import os, sys
_TESTS = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _TESTS); sys.path.insert(0, os.path.dirname(_TESTS))      # tests/ (golden_io, oracle_lib) and the repository root (bench, fries_amd)
import numpy as np
import bench
from fries_amd import fcidump
from fries_amd.engine import FriEngine
m = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
mol = fcidump.synthetic("N2")
os.environ.pop("FRIES_DBG", None)
dets, vals = bench.build_state(mol, m, 4 * m, 20250215, 0, None, None)
os.environ["FRIES_DBG"] = "3"
eng = FriEngine(mol)
eng.setup(epsilon=0.01, vec_nonz=m, mat_nonz=m, max_dets=4 * m, target_norm=float(m), initiator=1.0, seed=20250215, distribution="HB_unnorm")
eng.vec_load(dets, vals); eng.restart(777, 0.0, 0.0, 0)
eng.iterate(3)
eng.close()
