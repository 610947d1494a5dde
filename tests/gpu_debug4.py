import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fries_amd import fcidump
from fries_amd.engine import FriEngine
eng = FriEngine(fcidump.synthetic("Ne"))
rng = np.random.RandomState(3)
for n, kind in [(5, "u"), (1000, "u"), (1024, "u"), (1025, "e"), (70000, "e"), (1000003, "u"), (1000003, "e"), (300000, "same"), (200000, "tie"), (4000000, "e"), (100000, "zeros")]:
    if kind == "u": a = rng.random_sample(n)
    elif kind == "e": a = np.exp(8 * rng.random_sample(n)) * (rng.random_sample(n) > 0.1)
    elif kind == "same": a = np.full(n, 0.1234567)
    elif kind == "tie": a = np.ldexp(rng.randint(1, 8, n).astype(float), -3)   # multiples of 1/8: many exact ties once the sum is large
    else: a = np.where(rng.random_sample(n) > 0.9, rng.random_sample(n), 0.0)
    ref = np.cumsum(a)          # numpy accumulates left to right
    # double check numpy's order on a slice with a python loop
    s = 0.0
    for x in a[:2000]: s = s + x
    assert s == ref[min(n, 2000) - 1]
    t0 = time.time()
    out, tot, dt, ds = eng.test_seqsum(a)
    bad = int((out != ref).sum())
    print(f"n={n:8d} {kind:5s} mismatches={bad} total_ok={tot == ref[-1]} dirty_tiles={dt} dirty_subs={ds} first_bad={np.nonzero(out != ref)[0][:3] if bad else None}  {time.time()-t0:.3f}s", flush=True)
