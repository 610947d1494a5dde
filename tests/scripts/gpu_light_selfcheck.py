"""Ad-hoc: the light replays against "every wave decides in every replay" over a long run at bench size.  Both must give the reference's
trajectory, so they must give the same one: counts, norm and shift bit for bit every iteration, the vector digest at the end.
usage: gpu_light_selfcheck.py [m] [iterations]"""
import os, sys, subprocess, json, hashlib
_TESTS = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _TESTS); sys.path.insert(0, os.path.dirname(_TESTS))      # tests/ (golden_io, oracle_lib) and the repository root (bench, fries_amd)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # tests/ (golden_io, oracle_lib)
import numpy as np

m = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
n_it = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
if len(sys.argv) > 3 and sys.argv[3] == "worker":
    import bench
    from fries_amd import fcidump
    from fries_amd.engine import FriEngine
    mol = fcidump.synthetic("N2")
    dets, vals = bench.build_state(mol, m, 4 * m, 20250215, 0, None, None)
    eng = FriEngine(mol)
    eng.setup(epsilon=0.01, vec_nonz=m, mat_nonz=m, max_dets=4 * m, target_norm=float(m), initiator=1.0, seed=20250215, distribution="HB_unnorm")
    eng.vec_load(dets, vals)
    eng.restart(777, 0.0, 0.0, 0)
    h = hashlib.sha256()
    for blk in range(n_it // 100):
        lg = eng.iterate(100)
        for f in ("nkept", "n_nonz", "curr_size", "num_success", "norm", "shift", "numer", "denom"):
            h.update(np.ascontiguousarray(lg[f]).tobytes())
        assert int(lg["err"].max()) == 0
    d, v = eng.vector()
    h.update(d.tobytes()); h.update(v.tobytes())
    print(json.dumps({"digest": h.hexdigest(), "replays": eng.counters()["fks_replays"], "n_nonz": int(lg["n_nonz"][-1])}))
    sys.exit(0)
out = {}
for name, env in (("light", {}), ("every wave", {"FRIES_FKS_NO_LIGHT": "1"}), ("no sweep-count allowance, per-chunk start", {"FRIES_FKS_NO_EXT": "1", "FRIES_GROUP_WARM_ALL": "0"})):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), str(m), str(n_it), "worker"], env=e, capture_output=True, text=True, timeout=900)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not line:
        print(name, "FAILED", r.stderr[-2000:]); sys.exit(1)
    out[name] = json.loads(line[-1])
    print(f"{name:44s} {out[name]}", flush=True)
same = len({v["digest"] for v in out.values()}) == 1
print("identical trajectories and final vectors over", n_it, "iterations at m =", m, ":", same)
sys.exit(0 if same else 1)
