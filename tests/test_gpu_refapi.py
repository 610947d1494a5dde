"""The reference's C++ interface as a boundary (include/FRIES/*.hpp over libfries_hip.so).

Two programs drive FRI through DistVec / Adder / HBCompressSys / apply_HBPP_sys / find_preserve / sys_comp / adjust_shift exactly as
FRIES_bin/frisys_mol.cpp does:

  * tests/cpp/frisys_mol_ref_api -- a test program written against include/FRIES with a --seed option and 17-digit output;
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


# ------------------------------------------------------------------ the other reference drivers, unmodified, on the engine
FQ_REFSRC = os.path.join(ROOT, "oracle", "_ref", "fciqmc_mol_refsrc_on_hip")
HH_REFSRC = os.path.join(ROOT, "oracle", "_ref", "frisys_hh_refsrc_on_hip")
MPI_REFSRC = os.path.join(ROOT, "oracle", "_ref", "frisys_mol_refsrc_on_hip_mpi")
MPIEXEC = "/opt/conda/bin/mpiexec"


def _read_fq_rows(name):
    rows = []
    with open(os.path.join(golden_io.GOLD, name + ".traj")) as f:
        for ln in f:
            if ln.startswith("#"):
                continue
            t = ln.split()
            rows.append(dict(numer=float.fromhex(t[1]), denom=float.fromhex(t[2]), norm=float.fromhex(t[3]), shift=float.fromhex(t[4]), n_nonz=int(t[5]), n_ini=int(t[6]),
                             curr_size=int(t[7]), n_spawn=int(t[8]), hash=int(t[9], 16)))
    return rows


@pytest.mark.parametrize("name", ["fciqmc_ne", "fciqmc_n2_hb", "fciqmc_h2o_ini0"])
def test_reference_fciqmc_driver_source_runs_on_the_engine(name, tmp_path):
    """FRIES_bin/fciqmc_mol.cpp, compiled where it lies against include/FRIES (oracle/Makefile refdrv): its DistVec<int> walker vector moves
    to the device at the first perform_add (annihilating merge, diagonal elements, projections there), its per-determinant samplers
    (bin_sample, doub_multin / hb_doub_multi, sing_multin, round_binomially) are this build's host forms on the driver's own mt19937.  With
    the golden run's seed it must walk the reference's trajectory: initiator counts, walker numbers, shift and the projections to the six
    digits the driver prints, the stored vector bit for bit."""
    if not os.path.exists(FQ_REFSRC):
        pytest.skip("oracle/_ref/fciqmc_mol_refsrc_on_hip is built from /root/reference by __graft_entry__.build() in the container")
    r = golden_io.manifest()["fciqmc_runs"][name]
    rows = _read_fq_rows(name)
    mol = fcidump.synthetic(r["shape"])
    fc = str(tmp_path / "mol.FCIDUMP")
    fcidump.write_fcidump(fc, mol)
    out = str(tmp_path / "run") + "/"
    os.makedirs(out)
    n_it = min(r["n_iter"], 120)
    cmd = [FQ_REFSRC, "--fcidump_path", fc, "--point_group", mol.point_group, "--distribution", r["distribution"], "--target", str(r["target_walkers"]),
           "--max_dets", str(r["max_dets"]), "--initiator", str(r["initiator"]), "--epsilon", repr(r["epsilon"]), "--max_iter", str(n_it), "--result_dir", out]
    env = dict(os.environ, LD_PRELOAD=SHIM, FRIES_FIXED_CLOCK_NS=str(r["seed"]))
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0 and "Exception" not in res.stderr, res.stderr[-2000:]
    num = np.loadtxt(out + "projnum.txt").reshape(-1); den = np.loadtxt(out + "projden.txt").reshape(-1); nini = np.loadtxt(out + "nini.txt").reshape(-1)
    assert num.size == n_it
    six = lambda x: float("%.6g" % x)
    for i in range(n_it):
        assert int(nini[i]) == rows[i]["n_ini"], i
        assert num[i] == six(rows[i]["numer"]) and den[i] == six(rows[i]["denom"]), (i, num[i], rows[i]["numer"], den[i], rows[i]["denom"])
    sh = np.loadtxt(out + "S.txt").reshape(-1); nw = np.loadtxt(out + "N.txt").reshape(-1); nz = np.loadtxt(out + "nnonz.txt").reshape(-1)
    for k in range(n_it // 10):
        row = rows[10 * k + 9]
        assert sh[k] == six(row["shift"]) and int(nw[k]) == int(row["norm"]) and int(nz[k]) == row["n_nonz"], k
    nb = (2 * mol.n_orb + 7) // 8
    raw = np.fromfile(out + "dets0.dat", dtype=np.uint8)
    n_saved = raw.size // nb
    vals = np.fromfile(out + "vals0.dat", dtype=np.int32)
    assert vals.size == n_saved == rows[n_it - 1]["curr_size"]
    dets = np.zeros(n_saved, dtype=np.uint64)
    for b in range(nb):
        dets |= raw.reshape(n_saved, nb)[:, b].astype(np.uint64) << np.uint64(8 * b)
    assert golden_io.vec_hash(dets, vals.astype(np.float64)) == rows[n_it - 1]["hash"]


@pytest.mark.parametrize("name", ["n2_m10000_unnorm_p2", "h2o_m5000_hb_p3"])
def test_reference_driver_source_under_mpiexec_on_the_engine(name, tmp_path):
    """The same frisys_mol.cpp against the image's real MPI instead of include/FRIES/compat: `mpiexec -n P`, one engine context per rank
    (sharing the one GPU here), Adder::perform_add routing the adds with MPI_Alltoallv, sum_mpi with MPI_Allgather, and the engine's own
    collectives (the sum_mpi's inside find_keep_sub / find_preserve / sys_comp) reaching MPI_COMM_WORLD through the host-collectives
    transport (fries_hostcomm_create).  Every rank's final shard must equal what the same rank of the reference wrote under mpiexec."""
    if not os.path.exists(MPI_REFSRC):
        pytest.skip("oracle/_ref/frisys_mol_refsrc_on_hip_mpi is built from /root/reference by __graft_entry__.build() in the container")
    if not os.path.exists(MPIEXEC):
        pytest.skip("no mpiexec on this machine")
    r = golden_io.manifest()["mpi_runs"][name]
    P = r["n_ranks"]
    mol = fcidump.synthetic(r["shape"])
    fc = str(tmp_path / "mol.FCIDUMP")
    fcidump.write_fcidump(fc, mol)
    out = str(tmp_path / "run") + "/"
    os.makedirs(out)
    n_it = r["n_iter"]
    env = dict(os.environ, LD_PRELOAD=SHIM, FRIES_FIXED_CLOCK_NS=str(r["seed"]), FRIES_DEVICE="0")
    res = subprocess.run([MPIEXEC, "-n", str(P)] + _cmd(MPI_REFSRC, fc, mol, r, n_it, out), capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0 and "Exception" not in res.stderr, (res.stdout[-1500:], res.stderr[-2000:])
    g0 = golden_io.read_traj(name, 0)
    hf_proc = g0["hf_proc"]
    gh = golden_io.read_traj(name, hf_proc)
    nk = np.loadtxt(out + "nkept.txt").reshape(-1); num = np.loadtxt(out + "projnum.txt").reshape(-1); den = np.loadtxt(out + "projden.txt").reshape(-1)
    assert nk.size == n_it
    six = lambda x: float("%.6g" % x)
    for i in range(n_it):
        row = gh["rows"][i]
        assert int(nk[i]) == row["nkept"], i
        assert num[i] == six(row["numer"]) and den[i] == six(row["denom"]), (i, num[i], row["numer"], den[i], row["denom"])
    nb = (2 * mol.n_orb + 7) // 8
    for k in range(P):
        g = golden_io.read_traj(name, k)
        raw = np.fromfile(out + f"dets{k}.dat", dtype=np.uint8)
        n_saved = raw.size // nb
        vals = np.fromfile(out + f"vals{k}.dat", dtype=np.float64)
        assert n_saved == g["rows"][n_it - 1]["curr_size"], k
        dets = np.zeros(n_saved, dtype=np.uint64)
        for b in range(nb):
            dets |= raw.reshape(n_saved, nb)[:, b].astype(np.uint64) << np.uint64(8 * b)
        assert golden_io.vec_hash(dets, vals[:n_saved]) == g["rows"][n_it - 1]["hash"], k


@pytest.mark.parametrize("name", ["hh_l6_m2000", "hh_l8_m5000"])
def test_reference_frisys_hh_driver_source_runs_on_the_engine(name, tmp_path):
    """FRIES_bin/frisys_hh.cpp, compiled where it lies against include/FRIES: its HubHolVec moves to the device at the first comp_sub
    (fries_hh_setup from the parameter file parse_hh_input read, the driver's own scramblers), both comp_sub calls, the merge of the adds the
    driver forms on the host from the neighbour / phonon mirrors, add_vecs, find_preserve, sys_comp and calc_ref_ovlp run there.  Against the
    reference's golden trajectory: denominators to the printed digits, numerators to 1e-6 relative (six printed digits), shift and one-norm,
    and the stored states with their values bit for bit."""
    if not os.path.exists(HH_REFSRC):
        pytest.skip("oracle/_ref/frisys_hh_refsrc_on_hip is built from /root/reference by __graft_entry__.build() in the container")
    r = golden_io.manifest()["hh_runs"][name]
    g = golden_io.read_traj(name)
    pf = tmp_path / "hh_params.txt"
    pf.write_text("n_elec\n%d\nlat_len\n%d\nn_dim\n1\neps\n%r\nU\n%r\nomega\n%r\ng\n%r\ngs_energy\n%r\n" % (r["n_elec"], r["n_sites"], r["eps"], r["U"], r["omega"], r["g"], r["gs_energy"]))
    out = str(tmp_path / "run") + "/"
    os.makedirs(out)
    n_it = r["n_iter"]
    cmd = [HH_REFSRC, "--params_path", str(pf), "--vec_nonz", str(r["vec_nonz"]), "--max_dets", str(r["max_dets"]), "--target", repr(r["target_norm"]),
           "--initiator", repr(r["initiator"]), "--max_iter", str(n_it), "--result_dir", out]
    env = dict(os.environ, LD_PRELOAD=SHIM, FRIES_FIXED_CLOCK_NS=str(r["seed"]))
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0 and "Exception" not in res.stderr, res.stderr[-2000:]
    num = np.loadtxt(out + "projnum.txt").reshape(-1); den = np.loadtxt(out + "projden.txt").reshape(-1)
    sh = np.loadtxt(out + "S.txt").reshape(-1); nm = np.loadtxt(out + "norm.txt").reshape(-1)
    assert num.size == n_it
    six = lambda x: float("%.6g" % x)
    for i in range(n_it):
        row = g["rows"][i]
        assert den[i] == six(row["denom"]), (i, den[i], row["denom"])
        assert abs(num[i] - row["numer"]) <= 1e-5 * max(1.0, abs(row["numer"])), (i, num[i], row["numer"])       # (six printed digits)
    for k in range(n_it // 10):
        row = g["rows"][10 * k + 9]
        assert sh[k] == six(row["shift"]) and nm[k] == six(row["norm"]), k
    L = r["n_sites"]
    nb = (5 * L + 7) // 8
    raw = np.fromfile(out + "dets0.dat", dtype=np.uint8)
    n_saved = raw.size // nb
    vals = np.fromfile(out + "vals0.dat", dtype=np.float64)
    assert vals.size == 2 * n_saved and n_saved == g["rows"][n_it - 1]["curr_size"]
    dets = np.zeros(n_saved, dtype=np.uint64)
    for b in range(nb):
        dets |= raw.reshape(n_saved, nb)[:, b].astype(np.uint64) << np.uint64(8 * b)
    assert golden_io.vec_hash(dets, vals[:n_saved]) == g["rows"][n_it - 1]["hash"]


@pytest.mark.parametrize("name", ["ne_m2000_dense", "ne_m2000_dense_p2"])
def test_reference_driver_source_with_det_space_on_the_engine(name, tmp_path):
    """--det_space through the reference's own driver: DistVec::init_dense on the host vector (the adds travel to their owners), the
    dense space declared to the device when the vector moves there (fries_vec_set_dense), the driver's own exact multiplication by the dense
    block of H through add / perform_add, find_preserve / sys_comp behind the dense positions, dense_norm.  One rank (MPI stand-in) and
    `mpiexec -n 2` (real MPI), against the reference's trajectories."""
    man = golden_io.manifest()
    r = man["dense_runs"][name] if name in man["dense_runs"] else man["dense_mpi_runs"][name]
    P = r.get("n_ranks", 1)
    exe = REFSRC if P == 1 else MPI_REFSRC
    if not os.path.exists(exe):
        pytest.skip("the reference driver binaries are built from /root/reference by __graft_entry__.build() in the container")
    if P > 1 and not os.path.exists(MPIEXEC):
        pytest.skip("no mpiexec on this machine")
    mol = fcidump.synthetic(r["shape"])
    fc = str(tmp_path / "mol.FCIDUMP")
    fcidump.write_fcidump(fc, mol)
    out = str(tmp_path / "run") + "/"
    os.makedirs(out)
    n_it = min(r["n_iter"], 40)
    cmd = _cmd(exe, fc, mol, r, n_it, out) + ["--det_space", os.path.join(golden_io.GOLD, r["det_space"])]
    if P > 1:
        cmd = [MPIEXEC, "-n", str(P)] + cmd
    env = dict(os.environ, LD_PRELOAD=SHIM, FRIES_FIXED_CLOCK_NS=str(r["seed"]), FRIES_DEVICE="0")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0 and "Exception" not in res.stderr, (res.stdout[-1500:], res.stderr[-2000:])
    g0 = golden_io.read_traj(name, None if P == 1 else 0)
    hf_proc = g0["hf_proc"] if P > 1 else 0
    gh = golden_io.read_traj(name, None if P == 1 else hf_proc)
    nk = np.loadtxt(out + "nkept.txt").reshape(-1); num = np.loadtxt(out + "projnum.txt").reshape(-1); den = np.loadtxt(out + "projden.txt").reshape(-1)
    six = lambda x: float("%.6g" % x)
    for i in range(n_it):
        row = gh["rows"][i]
        assert int(nk[i]) == row["nkept"], i
        assert num[i] == six(row["numer"]) and den[i] == six(row["denom"]), (i, num[i], row["numer"], den[i], row["denom"])
    nb = (2 * mol.n_orb + 7) // 8
    for k in range(P):
        g = golden_io.read_traj(name, None if P == 1 else k)
        raw = np.fromfile(out + f"dets{k}.dat", dtype=np.uint8)
        n_saved = raw.size // nb
        vals = np.fromfile(out + f"vals{k}.dat", dtype=np.float64)
        assert n_saved == g["rows"][n_it - 1]["curr_size"], k
        dets = np.zeros(n_saved, dtype=np.uint64)
        for b in range(nb):
            dets |= raw.reshape(n_saved, nb)[:, b].astype(np.uint64) << np.uint64(8 * b)
        assert golden_io.vec_hash(dets, vals[:n_saved]) == g["rows"][n_it - 1]["hash"], k
    assert open(out + "dense.txt").read().strip().rstrip(",") != ""
