"""What the reference's own driver SOURCE costs on the engine at the bench's size (run on the GPU box; writes plain text to stdout).

oracle/_ref/frisys_mol_refsrc_on_hip is /root/reference/FRIES_bin/frisys_mol.cpp compiled unmodified against include/FRIES (oracle/Makefile,
target refdrv).  Its loop keeps the reference's host-side structure: it copies values() into the compression scratch, walks the compressed
samples on the host to form the spawned determinants, buffers them in the Adder, and applies the diagonal through operator[] -- so the
mirrors of the device vector cross PCIe every iteration.  This script restarts that binary from the bench's state (a checkpoint in the
reference's format: hash.dat, dets0.dat, vals0.dat, dense.txt, S.txt) at vec_nonz = mat_nonz = target = m and reports iterations/s and
PCIe bytes per iteration (FRIES_FACADE_STATS=1, include/FRIES/backend.hpp), beside the engine's own loop (fries_iterate) on the same state.

usage: python tests/scripts/gpu_facade_cost.py [m] [iterations]
"""
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    m = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
    n_it = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    import bench
    from fries_amd import fcidump
    from fries_amd.engine import FriEngine

    exe = os.path.join(ROOT, "oracle", "_ref", "frisys_mol_refsrc_on_hip")
    if not os.path.exists(exe):
        raise SystemExit("oracle/_ref/frisys_mol_refsrc_on_hip is not built (make -C oracle refdrv, in the container that holds /root/reference)")
    mol = fcidump.synthetic("N2")
    max_dets = 4 * m
    dets, vals = bench.build_state(mol, m, max_dets, 20250215, 0, None, None)
    print(f"state: {len(vals)} determinants, one-norm {np.abs(vals).sum():.1f}", flush=True)

    tmp = tempfile.mkdtemp(prefix="fries_facade_")
    ck = os.path.join(tmp, "ck") + "/"
    os.makedirs(ck)
    fc = os.path.join(tmp, "mol.FCIDUMP")
    fcidump.write_fcidump(fc, mol)
    det_size = (2 * mol.n_orb + 7) // 8
    words = np.ascontiguousarray(dets, dtype=np.uint64)
    words.view(np.uint8).reshape(-1, 8)[:, :det_size].tofile(ck + "dets0.dat")
    np.concatenate([np.asarray(vals, dtype=np.float64), np.zeros(len(vals))]).tofile(ck + "vals0.dat")
    np.random.default_rng(5).integers(0, 2 ** 32, size=2 * mol.n_orb, dtype=np.uint32).tofile(ck + "hash.dat")
    open(ck + "dense.txt", "w").write("0\n")
    open(ck + "S.txt", "w").write("0\n")

    def run(iters):
        out = os.path.join(tmp, f"out{iters}") + "/"
        os.makedirs(out)
        cmd = [exe, "--fcidump_path", fc, "--point_group", mol.point_group, "--distribution", "HB_unnorm", "--vec_nonz", str(m), "--mat_nonz", str(m),
               "--max_dets", str(max_dets), "--target", repr(float(m)), "--initiator", "1", "--epsilon", "0.01", "--max_iter", str(iters),
               "--load_dir", ck, "--result_dir", out]
        env = dict(os.environ, FRIES_FACADE_STATS="1", FRIES_DEVICE="0")
        t0 = time.perf_counter()
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        dt = time.perf_counter() - t0
        if p.returncode != 0:
            raise SystemExit(f"driver failed ({p.returncode}):\n{p.stdout[-2000:]}\n{p.stderr[-2000:]}")
        mm = re.search(r"fries facade: (\d+) bytes host->device, (\d+) bytes device->host", p.stderr)
        up, down = (int(mm.group(1)), int(mm.group(2))) if mm else (None, None)
        nk = np.loadtxt(out + "nkept.txt", ndmin=1)
        return dt, up, down, len(nk)

    a = run(n_it // 3)
    b = run(n_it)
    d_it = b[3] - a[3]
    per_it = (b[0] - a[0]) / d_it
    print(f"reference driver source on the engine (frisys_mol.cpp unmodified, include/FRIES), m = {m}:")
    print(f"  runs of {a[3]} and {b[3]} iterations: {a[0]:.2f} s and {b[0]:.2f} s  ->  {per_it * 1e3:.1f} ms per iteration = {1.0 / per_it:.2f} iterations/s (set-up excluded by the difference)")
    if b[1] is not None:
        print(f"  PCIe per iteration: {(b[1] - a[1]) / d_it / 1e6:.1f} MB host->device, {(b[2] - a[2]) / d_it / 1e6:.1f} MB device->host")

    # the engine's own loop from the same state
    eng = FriEngine(mol, device=0)
    eng.setup(epsilon=0.01, vec_nonz=m, mat_nonz=m, max_dets=max_dets, target_norm=float(m), initiator=1.0, seed=20250215, distribution="HB_unnorm")
    eng.vec_load(dets, vals)
    eng.restart(777, 0.0, 0.0, 0)
    eng.iterate(10, want_logs=False)
    eng.vec_info()
    t0 = time.perf_counter()
    eng.iterate(100, want_logs=False)
    eng.vec_info()
    dt = time.perf_counter() - t0
    eng.close()
    print(f"engine loop on the same state (fries_iterate, what bench.py times): {dt * 10:.2f} ms per iteration = {100 / dt:.1f} iterations/s, no bulk PCIe traffic")
    print(f"ratio: {per_it / (dt / 100):.1f} x -- the price of keeping the reference's host loop (sample walk, Adder buffers, operator[] on the diagonal) around the device operators")


if __name__ == "__main__":
    main()
