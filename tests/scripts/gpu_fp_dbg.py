"""Ad-hoc: which kernel of fciqmc_fp_mol grows with the iteration count."""
import os, sys, time
_TESTS = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _TESTS); sys.path.insert(0, os.path.dirname(_TESTS))      # tests/ (golden_io, oracle_lib) and the repository root (bench, fries_amd)
import numpy as np
from fries_amd import fcidump
from fries_amd.engine import FriEngine
mol = fcidump.synthetic("N2")
eng = FriEngine(mol)
eng.setup_fciqmc(epsilon=0.02, target_walkers=1000000, max_dets=8000000, initiator=3, seed=1, fp=True)
for phase in range(4):
    eng.iterate_fciqmc(1500)
    eng.prof_enable(True)
    t0 = time.time()
    lg = eng.iterate_fciqmc(100)
    dt = time.time() - t0
    rep = eng.prof_report()
    eng.prof_enable(False)
    info = eng.vec_info()
    print(f"after {(phase + 1) * 1600} iterations: {100 / dt:.1f} it/s, curr_size {info[0]} n_nonz {info[1]} n_free {info[2]}, kernel ms/it {sum(ms for ms, _ in rep.values()) / 100:.3f}")
    for k, (ms, calls) in sorted(rep.items(), key=lambda kv: -kv[1][0])[:6]:
        print(f"   {k:22s} {ms / 100:8.4f} ms/it {calls / 100:.1f} calls/it")
