"""scratch: wall-clock of frisys_hh (L = 12, budget 1e6) iteration by iteration, and of apply_HBPP_piv at n_samp = 1e6 on the bench's state"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # tests/ (golden_io, oracle_lib); _TESTS = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _TESTS); sys.path.insert(0, os.path.dirname(_TESTS))      # tests/ (golden_io, oracle_lib) and the repository root (bench, fries_amd)
import numpy as np
import golden_io
from fries_amd.engine import FriEngine
what = sys.argv[1] if len(sys.argv) > 1 else "hh"
if what == "hh":
    P0 = int(sys.argv[2]) if len(sys.argv) > 2 else 140
    P1 = int(sys.argv[3]) if len(sys.argv) > 3 else 150
    r = golden_io.manifest()["hh_scale_runs"]["hh_l12_m1e6"]
    eng = FriEngine(None)
    eng.setup_hh(n_elec=r["n_elec"], n_sites=r["n_sites"], eps=r["eps"], U=r["U"], omega=r["omega"], g=r["g"], gs_energy=r["gs_energy"], vec_nonz=r["vec_nonz"],
                 max_dets=r["max_dets"], target_norm=r["target_norm"], initiator=r["initiator"], seed=r["seed"])
    ts = []
    for i in range(150):
        if i == P0: eng.prof_enable(True)
        if i == P1: break
        t0 = time.perf_counter(); lg = eng.iterate_hh(1)[0]; ts.append(time.perf_counter() - t0)
        if i % 10 == 9: print("it", i, "n_nonz", int(lg["n_nonz"]), "num_success", int(lg["num_success"]), "ms/it (last 10)", round(1e3 * float(np.mean(ts[-10:])), 2), flush=True)
    print("hh L=12 m=1e6: last 10 iterations %.2f ms/it = %.1f it/s" % (1e3 * np.mean(ts[-10:]), 1.0 / np.mean(ts[-10:])))
    rep = eng.prof_report()
    tot = sum(v[0] for v in rep.values())
    print("kernel time of iterations %d-%d: %.1f ms total (with event overhead)" % (P0, P1 - 1, tot))
    for name, (ms, calls) in sorted(rep.items(), key=lambda x: -x[1][0])[:16]: print("  %-28s %9.2f ms %7d calls  %.1f us/call" % (name, ms, calls, 1e3 * ms / max(1, calls)))
else:
    import bench
    from fries_amd import fcidump
    m = 1000000
    mol = fcidump.synthetic("N2")
    dets, vals = bench.build_state(mol, m, 4 * m, 20250215, 0, None, None)
    eng = FriEngine(mol)
    eng.setup(epsilon=0.01, vec_nonz=m, mat_nonz=m, max_dets=4 * m, target_norm=float(m), initiator=1.0, seed=20250215, distribution="HB_unnorm")
    eng.vec_load(dets, vals); eng.restart(777, 0.0, 0.0, 0)
    for k in range(10):
        if k == 9: eng.prof_enable(True)
        t0 = time.perf_counter(); out = eng.apply_hbpp_piv(m); dt = time.perf_counter() - t0
        print("apply_HBPP_piv n_samp=1e6 call", k, "%.1f ms" % (1e3 * dt), "emitted", len(out[0]) if isinstance(out, tuple) else out, "piv_stats", eng.piv_stats() if hasattr(eng, "piv_stats") else None, flush=True)
    rep = eng.prof_report()
    print("kernels of the last call: %.1f ms" % sum(v[0] for v in rep.values()))
    for name, (ms, calls) in sorted(rep.items(), key=lambda x: -x[1][0])[:18]: print("  %-28s %9.2f ms %7d calls  %.1f us/call" % (name, ms, calls, 1e3 * ms / max(1, calls)))
