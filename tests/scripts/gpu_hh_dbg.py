import os, sys
_TESTS = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _TESTS); sys.path.insert(0, os.path.dirname(_TESTS))      # tests/ (golden_io, oracle_lib) and the repository root (bench, fries_amd); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # tests/ (golden_io, oracle_lib)
import numpy as np
from fries_amd.engine import FriEngine
import oracle_lib
L, m, ini = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
par = dict(n_elec=L, n_sites=L, eps=0.005, U=4.0, omega=1.0, g=0.7, gs_energy=-4.0, vec_nonz=m, max_dets=4 * m, target_norm=float(m), initiator=ini, seed=3)
eng = FriEngine(None); eng.setup_hh(**par)
orc = oracle_lib.OracleHH(**par)
for it in range(int(sys.argv[4])):
    lo = orc.iterate(1)[0]
    try:
        lg = eng.iterate_hh(1)[0]
    except RuntimeError as e:
        print(it, "GPU error:", e, "| oracle n_nonz", int(lo["n_nonz"]), "succ", int(lo["num_success"]), "nkept", int(lo["nkept"])); break
    same = all(int(lg[f]) == int(lo[f]) for f in ("n_nonz", "curr_size", "num_success", "nkept")) and float(lg["norm"]) == float(lo["norm"])
    print(it, "n_nonz", int(lg["n_nonz"]), "succ", int(lg["num_success"]), "comp_len", lg["comp_len"][:2].tolist(), "nkept", int(lg["nkept"]), "same" if same else "DIFF", flush=True)
