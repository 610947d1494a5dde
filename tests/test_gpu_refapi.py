"""The reference's C++ interface as a boundary (include/FRIES/*.hpp over libfries_hip.so).

Two programs drive FRI through DistVec / Adder / HBCompressSys / apply_HBPP_sys / find_preserve / sys_comp / adjust_shift exactly as
FRIES_bin/frisys_mol.cpp does:

  * fries_amd/frisys_mol_ref_api -- committed source, written against include/FRIES with a --seed option and 17-digit output;
  * oracle/_ref/frisys_mol_refsrc_on_hip -- the reference's OWN frisys_mol.cpp, compiled where it lies in /root/reference with only the
    include path changed (oracle/Makefile `refdrv`, built by __graft_entry__.build() in the container; the binary travels to the GPU
    box like the other files of oracle/_ref).  It seeds from the clock, so the test pins std::chrono::system_clock::now() with an
    LD_PRELOAD shim (tests/cpp/fixed_clock.cpp) to the golden run's seed.

Both must reproduce the golden trajectories the real reference (CPU) logged: sample counts and the final stored vector bit for bit,
shift / one-norm / projected-energy numerator and denominator to the digits the program prints."""
import os
import subprocess

import numpy as np
import pytest

import golden_io
from fries_amd import fcidump

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFSRC = os.path.join(ROOT, "oracle", "_ref", "frisys_mol_refsrc_on_hip")
SHIM = os.path.join(ROOT, "tests", "cpp", "libfixed_clock.so")


def _cmd(exe, fc, mol, r, n_it, out):
    return [exe, "--fcidump_path", fc, "--point_group", mol.point_group, "--distribution", r["distribution"], "--vec_nonz", str(r["vec_nonz"]),
            "--mat_nonz", str(r["mat_nonz"]), "--max_dets", str(r["max_dets"]), "--target", repr(r["target_norm"]), "--initiator", repr(r["initiator"]),
            "--epsilon", repr(r["epsilon"]), "--max_iter", str(n_it), "--result_dir", out]


def _saved_vector(out, mol):
    nb = (2 * mol.n_orb + 7) // 8
    raw = np.fromfile(out + "dets0.dat", dtype=np.uint8)
    n_saved = raw.size // nb
    vals = np.fromfile(out + "vals0.dat", dtype=np.float64)
    assert vals.size == 2 * n_saved and np.all(vals[n_saved:] == 0)
    dets = np.zeros(n_saved, dtype=np.uint64)
    for b in range(nb):
        dets |= raw.reshape(n_saved, nb)[:, b].astype(np.uint64) << np.uint64(8 * b)
    return dets, vals[:n_saved]


@pytest.mark.parametrize("name", ["ne_m2000_unnorm", "h2o_m5000_hb", "n2_m30000_unnorm"])
def test_ref_api_driver_matches_reference(name, tmp_path):
    from fries_amd import build
    r = golden_io.manifest()["runs"][name]
    g = golden_io.read_traj(name)
    mol = fcidump.synthetic(r["shape"])
    fc = str(tmp_path / "mol.FCIDUMP")
    fcidump.write_fcidump(fc, mol)
    out = str(tmp_path / "run") + "/"
    os.makedirs(out)
    assert os.path.exists(build.REF_API_DRIVER), "frisys_mol_ref_api has not been built"
    n_it = min(r["n_iter"], 40)
    res = subprocess.run(_cmd(build.REF_API_DRIVER, fc, mol, r, n_it, out) + ["--seed", str(r["seed"])], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "Exception" not in res.stderr, res.stderr[-2000:]
    num = np.loadtxt(out + "projnum.txt"); den = np.loadtxt(out + "projden.txt"); nk = np.loadtxt(out + "nkept.txt")
    sh = np.loadtxt(out + "S.txt").reshape(-1); nm = np.loadtxt(out + "norm.txt").reshape(-1)
    assert num.size == n_it
    for i in range(n_it):
        row = g["rows"][i]
        assert int(nk[i]) == row["nkept"], i
        # DistVec::dot of this build sums in list order like the reference's loop: the same doubles, not merely close ones
        assert num[i] == row["numer"] and den[i] == row["denom"], (i, num[i], row["numer"], den[i], row["denom"])
    for k in range(n_it // 10):
        row = g["rows"][10 * k + 9]
        assert sh[k] == row["shift"] and nm[k] == row["norm"], k
    dets, vals = _saved_vector(out, mol)
    assert dets.size == g["rows"][n_it - 1]["curr_size"]
    assert golden_io.vec_hash(dets, vals) == g["rows"][n_it - 1]["hash"]


@pytest.mark.parametrize("name", ["ne_m2000_unnorm", "n2_m10000_unnorm_ini0"])
def test_reference_driver_source_runs_on_the_engine(name, tmp_path):
    if not os.path.exists(REFSRC):
        pytest.skip("oracle/_ref/frisys_mol_refsrc_on_hip is built from /root/reference by __graft_entry__.build() in the container")
    assert os.path.exists(SHIM), "tests/cpp/libfixed_clock.so has not been built"
    r = golden_io.manifest()["runs"][name]
    g = golden_io.read_traj(name)
    mol = fcidump.synthetic(r["shape"])
    fc = str(tmp_path / "mol.FCIDUMP")
    fcidump.write_fcidump(fc, mol)
    out = str(tmp_path / "run") + "/"
    os.makedirs(out)
    n_it = min(r["n_iter"], 40)
    env = dict(os.environ, LD_PRELOAD=SHIM, FRIES_FIXED_CLOCK_NS=str(r["seed"]))
    res = subprocess.run(_cmd(REFSRC, fc, mol, r, n_it, out), capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0 and "Exception" not in res.stderr, res.stderr[-2000:]
    assert "seed on process 0 is %d" % r["seed"] in res.stdout
    num = np.loadtxt(out + "projnum.txt"); den = np.loadtxt(out + "projden.txt"); nk = np.loadtxt(out + "nkept.txt")
    sh = np.loadtxt(out + "S.txt").reshape(-1); nm = np.loadtxt(out + "norm.txt").reshape(-1)
    assert num.size == n_it
    six = lambda x: float("%.6g" % x)       # the reference writes its text files with the stream's default 6 significant digits
    for i in range(n_it):
        row = g["rows"][i]
        assert int(nk[i]) == row["nkept"], i
        assert num[i] == six(row["numer"]) and den[i] == six(row["denom"]), (i, num[i], row["numer"], den[i], row["denom"])
    for k in range(n_it // 10):
        row = g["rows"][10 * k + 9]
        assert sh[k] == six(row["shift"]) and nm[k] == six(row["norm"]), k
    # the binary checkpoint carries every bit: stored positions, determinants and values equal to the reference's vector
    dets, vals = _saved_vector(out, mol)
    assert dets.size == g["rows"][n_it - 1]["curr_size"]
    assert golden_io.vec_hash(dets, vals) == g["rows"][n_it - 1]["hash"]
    assert np.fromfile(out + "hash.dat", dtype=np.uint32).size == 2 * mol.n_orb


@pytest.mark.parametrize("name", sorted(golden_io.manifest()["extra_runs"]))
def test_reference_driver_source_options_on_the_engine(name, tmp_path):
    """The unmodified reference driver with --trial_vec / --ini_vec (load_vec_txt, a 25-determinant trial vector through the host
    h_op_offdiag / h_op_diag of include/FRIES/Hamiltonians/molecule.hpp) and --ham_shift (the diagonal offset recovered from the
    driver's diag_shortcut lambda when the vector moves to the device), against the reference's trajectories."""
    if not os.path.exists(REFSRC):
        pytest.skip("oracle/_ref/frisys_mol_refsrc_on_hip is built from /root/reference by __graft_entry__.build() in the container")
    r = golden_io.manifest()["extra_runs"][name]
    g = golden_io.read_traj(name)
    mol = fcidump.synthetic(r["shape"])
    fc = str(tmp_path / "mol.FCIDUMP")
    fcidump.write_fcidump(fc, mol)
    out = str(tmp_path / "run") + "/"
    os.makedirs(out)
    n_it = r["n_iter"]
    cmd = _cmd(REFSRC, fc, mol, r, n_it, out)
    if "trial" in r:
        cmd += ["--trial_vec", os.path.join(golden_io.GOLD, r["trial"])]
    if "ini" in r:
        cmd += ["--ini_vec", os.path.join(golden_io.GOLD, r["ini"])]
    if "ham_shift" in r:
        cmd += ["--ham_shift", repr(r["ham_shift"])]
    env = dict(os.environ, LD_PRELOAD=SHIM, FRIES_FIXED_CLOCK_NS=str(r["seed"]))
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0 and "Exception" not in res.stderr, res.stderr[-2000:]
    num = np.loadtxt(out + "projnum.txt"); den = np.loadtxt(out + "projden.txt"); nk = np.loadtxt(out + "nkept.txt")
    six = lambda x: float("%.6g" % x)
    for i in range(n_it):
        row = g["rows"][i]
        assert int(nk[i]) == row["nkept"], i
        assert num[i] == six(row["numer"]) and den[i] == six(row["denom"]), (i, num[i], row["numer"], den[i], row["denom"])
    dets, vals = _saved_vector(out, mol)
    assert golden_io.vec_hash(dets, vals) == g["rows"][n_it - 1]["hash"]


@pytest.mark.parametrize("name", sorted(golden_io.manifest()["hbpiv_runs"]))
def test_ref_api_apply_hbpp_piv(name, tmp_path):
    """The reference's 12-argument apply_HBPP_piv (HBCompressPiv, the caller's std::mt19937 lent to the device and taken back) through
    include/FRIES/Hamiltonians/heat_bathPP.hpp, on the vector the ref-API driver holds after a golden run's iterations: positions,
    orbitals and values of the reference's own apply_HBPP_piv (tests/golden/hbpiv_*.txt), and the generator left in the right state."""
    from fries_amd import build
    h = golden_io.manifest()["hbpiv_runs"][name]
    r = golden_io.manifest()["runs"][h["run"]]
    cases = golden_io.read_hbpiv(name)
    mol = fcidump.synthetic(r["shape"])
    fc = str(tmp_path / "mol.FCIDUMP")
    fcidump.write_fcidump(fc, mol)
    out = str(tmp_path / "run") + "/"
    os.makedirs(out)
    spec = ",".join("%d:%d" % (c["seed"], c["n_samp"]) for c in cases)
    res = subprocess.run(_cmd(build.REF_API_DRIVER, fc, mol, r, h["n_iter"], out) + ["--seed", str(r["seed"]), "--piv_after", spec], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "Exception" not in res.stderr, res.stderr[-2000:]
    for k, c in enumerate(cases):
        raw = np.fromfile(out + "piv%d.bin" % k, dtype=np.uint8)
        n_out = int(raw[:8].view(np.uint64)[0])
        assert n_out == c["n_out"], (k, n_out, c["n_out"])
        pos = raw[16:16 + 8 * n_out].view(np.uint64)
        orbs = raw[16 + 8 * n_out:16 + 12 * n_out].reshape(-1, 4)
        vals = raw[16 + 12 * n_out:16 + 20 * n_out].view(np.float64)
        assert np.array_equal(pos, c["pos"].astype(np.uint64)) and np.array_equal(orbs, c["orbs"]) and vals.tobytes() == c["val"].tobytes(), (name, k)
