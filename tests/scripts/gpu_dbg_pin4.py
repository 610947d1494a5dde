"""debug: H2O m=1e7 filler, P thread ranks on the GPU vs the oracle's rank mode, first iterations"""
import sys, os, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # tests/ (golden_io, oracle_lib); _TESTS = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _TESTS); sys.path.insert(0, os.path.dirname(_TESTS))      # tests/ (golden_io, oracle_lib) and the repository root (bench, fries_amd)
import numpy as np
import oracle_lib
from fries_amd import fcidump
from fries_amd.comm import LocalGroup
from fries_amd.engine import FriEngine

P = int(sys.argv[1]); m = int(sys.argv[2]); n_it = int(sys.argv[3])
maxd = int(4 * m / P * 1.15) + 65536
mol = fcidump.synthetic("H2O")
par = dict(epsilon=0.01, vec_nonz=m, mat_nonz=m, max_dets=maxd, target_norm=0.0, initiator=0.0, seed=20250215, distribution="HB_unnorm")
orc = oracle_lib.OracleRanks(P, mol, **par) if P > 1 else oracle_lib.OracleFrisys(mol, **par)
grp = LocalGroup(P, m) if P > 1 else None
comms = [grp.comm(k, 0) for k in range(P)] if P > 1 else [None]
logs = [[] for _ in range(P)]
vecs = [None] * P
bar = threading.Barrier(P)

def work(k):
    eng = FriEngine(mol, device=0, comm=comms[k])
    eng.setup(**par)
    for it in range(n_it):
        logs[k].append(eng.iterate(1)[0].copy())
    vecs[k] = eng.vector()
    eng.close()

th = [threading.Thread(target=work, args=(k,)) for k in range(P)]
[t.start() for t in th]; [t.join() for t in th]
for it in range(n_it):
    lo = orc.iterate(1)
    for k in range(P):
        o = lo[k][0] if P > 1 else lo[0]
        g = logs[k][it]
        bad = [f for f in ("nkept", "n_nonz", "curr_size", "num_success") if int(g[f]) != int(o[f])]
        if float(g["norm"]) != float(o["norm"]): bad.append("norm")
        print("it", it, "rank", k, "OK" if not bad else "MISMATCH %s" % bad, "gpu", [int(g[f]) for f in ("nkept", "n_nonz", "curr_size", "num_success")], list(map(int, g["comp_len"])), "err", int(g["err"]),
              "oracle", [int(o[f]) for f in ("nkept", "n_nonz", "curr_size", "num_success")], list(map(int, o["comp_len"])), flush=True)
