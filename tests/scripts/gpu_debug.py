"""Ad-hoc GPU bring-up script (not collected by pytest): prints where the HIP path and the oracle part ways."""
import os
import sys
import time

import numpy as np

_TESTS = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _TESTS); sys.path.insert(0, os.path.dirname(_TESTS))      # tests/ (golden_io, oracle_lib) and the repository root (bench, fries_amd)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # tests/ (golden_io, oracle_lib)
from fries_amd import fcidump  # noqa: E402
from fries_amd.engine import FriEngine  # noqa: E402
import oracle_lib  # noqa: E402


def rand_dets(rng, n_orb, n_elec, n):
    out = np.zeros(n, dtype=np.uint64)
    for i in range(n):
        d = 0
        for sp in range(2):
            for o in rng.choice(n_orb, n_elec // 2, replace=False):
                d |= 1 << (int(o) + sp * n_orb)
        out[i] = d
    return out


def main():
    shape = sys.argv[1] if len(sys.argv) > 1 else "Ne"
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    n_iter = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    dist = sys.argv[4] if len(sys.argv) > 4 else "HB_unnorm"
    mol = fcidump.synthetic(shape)
    par = dict(epsilon=0.01, vec_nonz=m, mat_nonz=m, max_dets=10 * m, target_norm=m / 2, initiator=1.0, seed=20250215, distribution=dist)
    t0 = time.time()
    eng = FriEngine(mol)
    print("engine created", time.time() - t0, flush=True)
    orc = oracle_lib.OracleFrisys(mol, **par)
    print("hf_en", eng.hf_energy, orc.hf_energy, eng.hf_energy == orc.hf_energy)
    for w in range(7):
        a, b = eng.hb_tensor(w), orc.hb_tensor(w)
        print("hb tensor", w, a.size, b.size, "bitwise" if np.array_equal(a, b) else f"MAXDIFF {np.abs(a - b).max()}")
    rng = np.random.RandomState(1)
    dets = rand_dets(rng, mol.n_orb, mol.n_elec, 2000)
    a, _ = eng.matrel(0, dets)
    b, _ = orc.matrel(0, dets)
    print("diag matrel bitwise:", np.array_equal(a, b), np.abs(a - b).max())
    # teeth
    r0, unit, n = 0.37 * 1.234567e-3, 1.234567e-3, 200000
    q = np.sort(rng.random_sample(5000) * unit * n)
    pos, below = eng.test_teeth(r0, unit, n, q)
    ref = np.empty(n)
    x = r0
    for k in range(n):
        ref[k] = x
        x = x + unit
    print("teeth bitwise:", np.array_equal(pos, ref), "n_mismatch", int((pos != ref).sum()))
    print("teeth_below ok:", np.array_equal(below, np.searchsorted(ref, q, side="left")))
    eng.setup(**par)
    print("p_doub", eng.p_doub, orc.p_doub, eng.p_doub == orc.p_doub)
    hd, hv = eng.htrial()
    od, ov = orc.htrial()
    print("htrial", hd.size, od.size, np.array_equal(hd, od), np.array_equal(hv, ov), np.abs(hv - ov).max() if hv.size == ov.size else None)
    t0 = time.time()
    for it in range(n_iter):
        lg = eng.iterate(1)[0]
        lo = orc.iterate(1)[0]
        gd, gv = eng.vector()
        cd, cv = orc.vector()
        same_len = gd.size == cd.size
        nz = cv != 0
        dets_ok = same_len and np.array_equal(gd[nz], cd[nz]) and np.array_equal(gv != 0, nz)
        vdiff = np.abs(gv - cv).max() / max(np.abs(cv).max(), 1e-300) if same_len else None
        ok = dets_ok and lg["num_success"] == lo["num_success"] and lg["n_nonz"] == lo["n_nonz"]
        print(f"it {it}: succ {lg['num_success']}/{lo['num_success']} nnz {lg['n_nonz']}/{lo['n_nonz']} size {lg['curr_size']}/{lo['curr_size']} "
              f"nkept {lg['nkept']}/{lo['nkept']} comp {list(lg['comp_len'])} dets_ok {dets_ok} vrel {vdiff} "
              f"num {lg['numer']:.12g}/{lo['numer']:.12g} den {lg['denom']:.12g}/{lo['denom']:.12g} norm {lg['norm']:.12g}/{lo['norm']:.12g} err {lg['err']}", flush=True)
        if not ok:
            print("DIVERGED")
            break
    print("loop time", time.time() - t0, "kernel launches", eng.kernel_launches)


if __name__ == "__main__":
    main()
