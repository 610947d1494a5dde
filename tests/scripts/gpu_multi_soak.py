"""Ad-hoc: frimulti_mol on the device over thousands of iterations with the shift engaged: error flags, norm, shift, energy."""
import os, sys, time
_TESTS = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _TESTS); sys.path.insert(0, os.path.dirname(_TESTS))      # tests/ (golden_io, oracle_lib) and the repository root (bench, fries_amd)
import numpy as np
from fries_amd import fcidump
from fries_amd.engine import FriEngine
m = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
n_tot = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
target = float(sys.argv[3]) if len(sys.argv) > 3 else 2000.0
mol = fcidump.synthetic("N2")
eng = FriEngine(mol)
eng.setup_multi(epsilon=0.01, vec_nonz=m, mat_nonz=m, max_dets=6 * m, target_norm=target, initiator=1.0, seed=7)
done = 0; worst = 0
t0 = time.time()
while done < n_tot:
    lg = eng.iterate_multi(250); done += 250
    worst |= int(lg["err"].max())
    en = lg["numer"] / lg["denom"]
    print(f"{done} n_nonz {int(lg['n_nonz'][-1])} curr_size {int(lg['curr_size'][-1])} spawns {int(lg['n_spawn'][-1])} norm {float(lg['norm'][-1]):.6g} shift {float(lg['shift'][-1]):.5f} "
          f"energy mean {float(en.mean()):.6f} err {int(lg['err'].max())} {250 / (time.time() - t0):.1f} it/s", flush=True)
    t0 = time.time()
print("error flags over the run:", worst)
sys.exit(0 if worst == 0 else 1)
