"""Ad-hoc: where k_fks_sweep's time goes.  FRIES_DBG=4 adds a loads-only launch (k_fks_prologue) beside every sweep;
FRIES_DBG=2 runs the sweeps without row evaluations (results are then not the reference's: timing only)."""
import os, sys, json, subprocess
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for dbg in ("0", "4", "2"):
    env = dict(os.environ, FRIES_DBG=dbg)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--cpu-iters", "0", "--steps", "20", "--warmup", "5"], env=env, capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print("FRIES_DBG=" + dbg, "it/s %.1f" % d["value"], "replays/iter", d["fks_replays_per_iter"], {k: (round(v["ms_per_iter"], 3), v["calls_per_iter"]) for k, v in d["top_kernels"].items()})
    except Exception as e:
        print("FRIES_DBG=" + dbg, "failed:", r.stderr[-600:])
