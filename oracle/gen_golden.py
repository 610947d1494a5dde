#!/usr/bin/env python3
"""Regenerates tests/golden/ from the REAL reference (oracle/_ref, built by `make -C oracle ref`).
Runs only where /root/reference exists.  Inputs are the seeded synthetic FCIDUMPs of
fries_amd/fcidump.py; outputs are small text fixtures (C99 hex floats) plus a manifest."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fries_amd import fcidump  # noqa: E402

HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
GOLD = os.path.join(ROOT, "tests", "golden")

# name: shape, point group flag, n_iter, seed, eps, vec_nonz, mat_nonz, max_dets, initiator, target, distribution, snapshot every
RUNS = {
    "ne_m2000_unnorm": ("Ne", 80, 20250215, 0.01, 2000, 2000, 20000, 1.0, 1000.0, "HB_unnorm", 80),
    "n2_m10000_unnorm_ini0": ("N2", 40, 7, 0.01, 10000, 10000, 80000, 0.0, 5000.0, "HB_unnorm", 0),
    "h2o_m5000_hb": ("H2O", 60, 99, 0.005, 5000, 8000, 80000, 3.0, 2000.0, "HB", 0),
    "n2_m30000_unnorm": ("N2", 30, 31, 0.01, 30000, 30000, 200000, 0.5, 10000.0, "HB_unnorm", 0),
    # edge shapes: 32 spatial orbitals (the whole 64-bit index), 2 electrons in 4 orbitals
    "max32_m3000_unnorm": ("MAX32", 25, 3, 0.01, 3000, 3000, 60000, 1.0, 1500.0, "HB_unnorm", 0),
    "max32_m3000_hb": ("MAX32", 25, 3, 0.01, 3000, 3000, 60000, 1.0, 1500.0, "HB", 0),
    "min4_m50_unnorm": ("MIN4", 25, 3, 0.01, 50, 50, 1000, 1.0, 25.0, "HB_unnorm", 0),
}

# driver options beyond the defaults: --trial_vec / --ini_vec text vectors (written below from H|HF>) and --ham_shift
EXTRA_RUNS = {
    "n2_m5000_trial_ini": (("N2", 30, 13, 0.01, 5000, 5000, 60000, 0.5, 3000.0, "HB_unnorm"), dict(trial="n2_trial_", ini="n2_ini_")),
    "ne_m2000_ham_shift": (("Ne", 30, 21, 0.01, 2000, 2000, 20000, 1.0, 1000.0, "HB_unnorm"), dict(ham_shift=-44.3)),
}

# binary checkpoints written by the reference itself (DistVec::save): run name -> after how many iterations
CHECKPOINTS = {"ne_m2000_unnorm": 40}

# multi-rank runs under mpiexec -n P: name -> (n_ranks, same tuple as RUNS without the snapshot field)
MPI_RUNS = {
    "n2_m10000_unnorm_p2": (2, ("N2", 40, 7, 0.01, 10000, 10000, 80000, 0.0, 5000.0, "HB_unnorm")),
    "n2_m10000_unnorm_p4": (4, ("N2", 40, 7, 0.01, 10000, 10000, 80000, 1.0, 5000.0, "HB_unnorm")),
    "h2o_m5000_hb_p3": (3, ("H2O", 40, 99, 0.005, 5000, 8000, 80000, 3.0, 2000.0, "HB")),
}
# the same with a smaller Adder (FRIES_ADDER_SIZE = the adder_size argument of the reference's DistVec): Adder::add reports a full buffer and the
# loop of frisys_mol.cpp:430-471 takes several perform_add rounds per pass.  name -> (n_ranks, adder_size, tuple as above)
ADDER_RUNS = {
    "n2_m10000_unnorm_p2_adder300": (2, 300, ("N2", 30, 7, 0.01, 10000, 10000, 80000, 1.0, 5000.0, "HB_unnorm")),
    "h2o_m5000_hb_p3_adder25": (3, 25, ("H2O", 30, 99, 0.005, 5000, 8000, 80000, 3.0, 2000.0, "HB")),
}
MPIEXEC = "/opt/conda/bin/mpiexec"

# fciqmc_mol, near-uniform excitation generator, one rank: name -> (shape, n_iter, seed, eps, target_walkers, max_dets, initiator)
FCIQMC_RUNS = {
    "fciqmc_ne": ("Ne", 200, 5, 0.002, 5000, 20000, 3, "NU"),
    "fciqmc_n2": ("N2", 300, 9, 0.004, 20000, 100000, 2, "NU"),
    "fciqmc_h2o_ini0": ("H2O", 150, 11, 0.002, 10000, 100000, 0, "NU"),
    "fciqmc_n2_hb": ("N2", 200, 9, 0.004, 20000, 100000, 2, "HB"),
    "fciqmc_h2o_hb_ini0": ("H2O", 150, 11, 0.002, 10000, 100000, 0, "HB"),
}

# fciqmc_mol under mpiexec -n P (one generator per rank, seeded seed + rank): name -> (n_ranks, same tuple as FCIQMC_RUNS)
FCIQMC_MPI_RUNS = {
    "fciqmc_n2_p2": (2, ("N2", 150, 9, 0.004, 20000, 100000, 2, "NU")),
    "fciqmc_h2o_hb_p3": (3, ("H2O", 120, 11, 0.002, 10000, 100000, 0, "HB")),
}

# frifull_mol (Hamiltonian applied in full), one rank: name -> (shape, n_iter, seed, eps, vec_nonz, max_dets, target)
FULL_RUNS = {
    "full_ne_m300": ("Ne", 40, 5, 0.01, 300, 400000, 120.0),
    "full_n2_m400": ("N2", 25, 7, 0.01, 400, 2000000, 150.0),
    "full_h2o_m200": ("H2O", 30, 11, 0.005, 200, 1000000, 110.0),
}

# frifull_mol under mpiexec -n P (frifull_mol.cpp's own Adder size; the excitations of a pass go through several perform_add rounds when a buffer fills):
# name -> (n_ranks, tuple as FULL_RUNS)
FULL_MPI_RUNS = {
    "full_ne_m300_p2": (2, ("Ne", 30, 5, 0.01, 300, 400000, 120.0)),
    "full_h2o_m200_p3": (3, ("H2O", 24, 11, 0.005, 200, 1000000, 110.0)),
}


def gen_full_mpi(manifest):
    manifest["full_mpi_runs"] = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, (n_ranks, (shape, n_iter, seed, eps, vnz, maxd, tgt)) in FULL_MPI_RUNS.items():
            mol = fcidump.synthetic(shape)
            path = os.path.join(tmp, shape + ".FCIDUMP")
            fcidump.write_fcidump(path, mol)
            out = os.path.join(GOLD, name + ".traj")
            subprocess.run([MPIEXEC, "-n", str(n_ranks), HARNESS, "frifull_mpi", path, mol.point_group, str(n_iter), str(seed), repr(eps), str(vnz), str(maxd), repr(tgt),
                            str(mol.n_orb), str(mol.n_elec), out], check=True)
            manifest["full_mpi_runs"][name] = dict(n_ranks=n_ranks, shape=shape, n_iter=n_iter, seed=seed, epsilon=eps, vec_nonz=vnz, max_dets=maxd, target_norm=tgt)


# frisys_hh (1-D Hubbard-Holstein): name -> (n_ranks, n_iter, seed, n_elec, n_sites, eps, U, omega, g, gs_energy, vec_nonz, max_dets, initiator, target)
HH_RUNS = {
    "hh_l6_m2000": (1, 60, 5, 6, 6, 0.01, 2.0, 0.5, 0.3, -3.0, 2000, 20000, 1.0, 1000.0),
    "hh_l8_m5000": (1, 40, 7, 8, 8, 0.01, 4.0, 1.0, 0.5, -2.0, 5000, 50000, 1.0, 2500.0),
    "hh_l10_m8000_ini0": (1, 40, 9, 10, 10, 0.005, 4.0, 1.0, 0.7, -4.0, 8000, 80000, 0.0, 4000.0),
    "hh_l8_m5000_p3": (3, 40, 7, 8, 8, 0.01, 4.0, 1.0, 0.5, -2.0, 5000, 50000, 1.0, 2500.0),
}

# frifull_hh (Hubbard-Holstein Hamiltonian applied in full): same tuple as HH_RUNS
HHFULL_RUNS = {
    "hhfull_l6_m300": (1, 60, 5, 6, 6, 0.01, 2.0, 0.5, 0.3, -3.0, 300, 200000, 1.0, 150.0),
    "hhfull_l8_m500_ini0": (1, 40, 7, 8, 8, 0.01, 4.0, 1.0, 0.5, -2.0, 500, 400000, 0.0, 250.0),
    # under mpiexec -n P (one Adder round per iteration at these sizes)
    "hhfull_l8_m500_p2": (2, 40, 7, 8, 8, 0.01, 4.0, 1.0, 0.5, -2.0, 500, 400000, 1.0, 250.0),
    "hhfull_l6_m300_ini0_p3": (3, 50, 5, 6, 6, 0.01, 2.0, 0.5, 0.3, -3.0, 300, 200000, 0.0, 150.0),
}


# apply_HBPP_piv (pivotal matrix compression) on the vector a golden frisys_mol run holds after n_iter iterations:
# name -> (run in RUNS, n_iter, [(n_samp, generator seed), ...])
HBPIV_RUNS = {
    "hbpiv_ne_unnorm": ("ne_m2000_unnorm", 30, [(1500, 11), (400, 12), (6000, 13), (1, 14)]),
    "hbpiv_h2o_hb": ("h2o_m5000_hb", 25, [(3000, 5), (700, 6)]),
    "hbpiv_n2_unnorm": ("n2_m10000_unnorm_ini0", 20, [(8000, 1), (25000, 2)]),
}


# Time-reversal symmetry (spin_parity = +-1).  tr_*: h_op_offdiag with both parities on a small source vector (ref_harness tr; the harness also
# checks flip_spins / tr_doub_connect there): name -> (shape, seed, n_src).  hbpiv_*_trp / _trm: apply_HBPP_piv with spin_parity +1 / -1
# (unnormalised heat bath only, heat_bathPP.cpp:1019): name -> (run in RUNS, n_iter, parity, cases).  H2O-shaped systems (24 orbitals) are left out:
# the reference's flip_spins is off by one byte for n_orb = 24 and 32 and its own h_op_offdiag then aborts on an invalid determinant.
TR_RUNS = {"tr_n2": ("N2", 5, 5), "tr_ne": ("Ne", 9, 6)}
HBPIV_TR_RUNS = {
    "hbpiv_ne_unnorm_trp": ("ne_m2000_unnorm", 30, 1, [(1500, 11), (400, 12)]),
    "hbpiv_ne_unnorm_trm": ("ne_m2000_unnorm", 30, -1, [(1500, 11), (3000, 13)]),
    "hbpiv_n2_unnorm_trp": ("n2_m10000_unnorm_ini0", 20, 1, [(3000, 1)]),
}

# frimulti_mol (multinomial matrix compression, --distribution HB), one rank: name -> (shape, n_iter, seed, eps, vec_nonz, mat_nonz, max_dets, initiator, target)
MULTI_RUNS = {
    "multi_ne_m1000": ("Ne", 60, 11, 0.01, 1000, 5000, 50000, 1.0, 500.0),
    "multi_n2_m5000_ini0": ("N2", 40, 3, 0.01, 5000, 20000, 200000, 0.0, 2500.0),
}


# fciqmc_fp_mol (real-valued walkers), one rank: name -> (shape, n_iter, seed, eps, target_walkers, max_dets, initiator, distribution)
FCIQMC_FP_RUNS = {
    "fciqmc_fp_ne": ("Ne", 250, 5, 0.005, 5000, 50000, 3, "NU"),
    "fciqmc_fp_n2_hb": ("N2", 200, 9, 0.006, 20000, 200000, 2, "HB"),
    "fciqmc_fp_h2o_ini0": ("H2O", 150, 11, 0.004, 10000, 200000, 0, "NU"),
}


# fciqmc_fp_mol under mpiexec -n P: name -> (n_ranks, same tuple as FCIQMC_FP_RUNS)
FCIQMC_FP_MPI_RUNS = {
    "fciqmc_fp_n2_p2": (2, ("N2", 150, 9, 0.006, 20000, 200000, 2, "NU")),
    "fciqmc_fp_h2o_hb_p3": (3, ("H2O", 120, 11, 0.004, 10000, 200000, 0, "HB")),
}


def gen_fp(manifest):
    manifest["fciqmc_fp_mpi_runs"] = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, (n_ranks, (shape, n_iter, seed, eps, tw, maxd, ini, dist)) in FCIQMC_FP_MPI_RUNS.items():
            mol = fcidump.synthetic(shape)
            path = os.path.join(tmp, shape + ".FCIDUMP")
            fcidump.write_fcidump(path, mol)
            out = os.path.join(GOLD, name + ".traj")
            subprocess.run([MPIEXEC, "-n", str(n_ranks), HARNESS, "fciqmc_fp", path, mol.point_group, str(n_iter), str(seed), repr(eps), str(tw), str(maxd), str(ini), out, dist], check=True)
            manifest["fciqmc_fp_mpi_runs"][name] = dict(n_ranks=n_ranks, shape=shape, n_iter=n_iter, seed=seed, epsilon=eps, target_walkers=tw, max_dets=maxd, initiator=ini,
                                                        distribution=dist, fp=True)
    manifest["fciqmc_fp_runs"] = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, (shape, n_iter, seed, eps, tw, maxd, ini, dist) in FCIQMC_FP_RUNS.items():
            mol = fcidump.synthetic(shape)
            path = os.path.join(tmp, shape + ".FCIDUMP")
            fcidump.write_fcidump(path, mol)
            out = os.path.join(GOLD, name + ".traj")
            subprocess.run([HARNESS, "fciqmc_fp", path, mol.point_group, str(n_iter), str(seed), repr(eps), str(tw), str(maxd), str(ini), out, dist], check=True)
            manifest["fciqmc_fp_runs"][name] = dict(shape=shape, n_iter=n_iter, seed=seed, epsilon=eps, target_walkers=tw, max_dets=maxd, initiator=ini, distribution=dist, fp=True)
        # fciqmc_fp_mol with --trial_vec / --ini_vec (fciqmc_fp_mol.cpp:157-185, 233-246): the 25-determinant N2 trial fixture and the real-valued start vector of the
        # frisys_mol run with the same options
        name = "fciqmc_fp_n2_trial_ini"
        mol = fcidump.synthetic("N2")
        path = os.path.join(tmp, "N2.FCIDUMP")
        fcidump.write_fcidump(path, mol)
        out = os.path.join(GOLD, name + ".traj")
        env = dict(os.environ, FRIES_TRIAL=os.path.join(GOLD, "n2_trial_"), FRIES_INI=os.path.join(GOLD, "n2_ini_"))
        subprocess.run([HARNESS, "fciqmc_fp", path, mol.point_group, "150", "9", "0.004", "20000", "100000", "2", out, "NU"], check=True, env=env)
        manifest["fciqmc_fp_runs"][name] = dict(shape="N2", n_iter=150, seed=9, epsilon=0.004, target_walkers=20000, max_dets=100000, initiator=2, distribution="NU", fp=True,
                                                trial="n2_trial_", ini="n2_ini_")


def gen_hhfull(manifest):
    manifest["hhfull_runs"] = {}
    for name, (n_ranks, n_iter, seed, n_elec, n_sites, eps, U, omega, g, gs, vnz, maxd, ini, tgt) in HHFULL_RUNS.items():
        out = os.path.join(GOLD, name + ".traj")
        cmd = [HARNESS, "hh", str(n_iter), str(seed), str(n_elec), str(n_sites), repr(eps), repr(U), repr(omega), repr(g), repr(gs), str(vnz), str(maxd), repr(ini), repr(tgt), out]
        if n_ranks > 1:
            cmd = [MPIEXEC, "-n", str(n_ranks)] + cmd
        subprocess.run(cmd, check=True, env=dict(os.environ, FRIES_HH_FULL="1"))
        manifest["hhfull_runs"][name] = dict(n_ranks=n_ranks, n_iter=n_iter, seed=seed, n_elec=n_elec, n_sites=n_sites, eps=eps, U=U, omega=omega, g=g,
                                              gs_energy=gs, vec_nonz=vnz, max_dets=maxd, initiator=ini, target_norm=tgt)


# frimulti_mol under mpiexec -n P: name -> (n_ranks, same tuple as MULTI_RUNS)
MULTI_MPI_RUNS = {
    "multi_n2_m5000_p2": (2, ("N2", 40, 3, 0.01, 5000, 20000, 200000, 1.0, 2500.0)),
    "multi_h2o_m3000_ini0_p3": (3, ("H2O", 40, 17, 0.005, 3000, 12000, 100000, 0.0, 1500.0)),
}


def gen_multi(manifest):
    manifest["multi_mpi_runs"] = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, (n_ranks, (shape, n_iter, seed, eps, vnz, mnz, maxd, ini, tgt)) in MULTI_MPI_RUNS.items():
            mol = fcidump.synthetic(shape)
            path = os.path.join(tmp, shape + ".FCIDUMP")
            fcidump.write_fcidump(path, mol)
            out = os.path.join(GOLD, name + ".traj")
            subprocess.run([MPIEXEC, "-n", str(n_ranks), HARNESS, "frimulti", path, mol.point_group, str(n_iter), str(seed), repr(eps), str(vnz), str(mnz), str(maxd), repr(ini), repr(tgt), out], check=True)
            manifest["multi_mpi_runs"][name] = dict(n_ranks=n_ranks, shape=shape, n_iter=n_iter, seed=seed, epsilon=eps, vec_nonz=vnz, mat_nonz=mnz, max_dets=maxd,
                                                    initiator=ini, target_norm=tgt)
    manifest["multi_runs"] = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, (shape, n_iter, seed, eps, vnz, mnz, maxd, ini, tgt) in MULTI_RUNS.items():
            mol = fcidump.synthetic(shape)
            path = os.path.join(tmp, shape + ".FCIDUMP")
            fcidump.write_fcidump(path, mol)
            out = os.path.join(GOLD, name + ".traj")
            subprocess.run([HARNESS, "frimulti", path, mol.point_group, str(n_iter), str(seed), repr(eps), str(vnz), str(mnz), str(maxd), repr(ini), repr(tgt), out], check=True)
            manifest["multi_runs"][name] = dict(shape=shape, n_iter=n_iter, seed=seed, epsilon=eps, vec_nonz=vnz, mat_nonz=mnz, max_dets=maxd, initiator=ini, target_norm=tgt)
        # frimulti_mol with --ini_vec (frimulti_mol.cpp:205-215): the real-valued start vector of the frisys_mol run with that option
        name = "multi_n2_ini"
        mol = fcidump.synthetic("N2")
        path = os.path.join(tmp, "N2.FCIDUMP")
        fcidump.write_fcidump(path, mol)
        out = os.path.join(GOLD, name + ".traj")
        env = dict(os.environ, FRIES_INI=os.path.join(GOLD, "n2_ini_"))
        subprocess.run([HARNESS, "frimulti", path, mol.point_group, "40", "3", "0.01", "5000", "20000", "200000", "1.0", "2500.0", out], check=True, env=env)
        manifest["multi_runs"][name] = dict(shape="N2", n_iter=40, seed=3, epsilon=0.01, vec_nonz=5000, mat_nonz=20000, max_dets=200000, initiator=1.0, target_norm=2500.0, ini="n2_ini_")
        # ... and --trial_vec (:139-163).  On ONE rank the reference refuses every trial file: trial_vec's Adder holds n_trial entries, add() reports the entry that
        # fills it, and this driver throws on that report ("Insufficient memory allocated in adder") where fciqmc_mol flushes.  Recorded as the expected error.  (Over
        # several ranks load_vec_txt returns 0 on the non-root ranks, io_utils.cpp:410-444, which then build zero-sized vectors and abort inside MPI_Alltoallv: no golden.)
        r = subprocess.run([HARNESS, "frimulti", path, mol.point_group, "5", "3", "0.01", "5000", "20000", "200000", "1.0", "2500.0", os.path.join(tmp, "x.traj")],
                           env=dict(os.environ, FRIES_TRIAL=os.path.join(GOLD, "n2_trial_")), capture_output=True, text=True)
        assert r.returncode != 0 and "Insufficient memory allocated in adder" in r.stderr, (r.returncode, r.stderr[-300:])
        manifest["multi_trial_one_rank_error"] = dict(shape="N2", trial="n2_trial_", error="Insufficient memory allocated in adder")


# BASELINE sizes pinned by the reference (ref_harness pin: bench.py's filler + restart, then n_iter iterations with digests):
# name -> (n_ranks, shape, seed, eps, m, max_dets per rank, distribution, run_seed, n_iter)
PIN_RUNS = {
    "pin_n2_m1e6": (1, "N2", 20250215, 0.01, 1000000, 4000000, "HB_unnorm", 777, 100),           # BASELINE config 2 = bench.py's workload
    "pin_h2o_m1e7_p8": (8, "H2O", 20250215, 0.01, 10000000, 5815536, "HB_unnorm", 777, 24),      # BASELINE config 4: mpiexec -n 8
    "pin_h2o_m1e7_p2": (2, "H2O", 20250215, 0.01, 10000000, 23065536, "HB_unnorm", 777, 6),       # the same on 2 ranks: > 1e6 spawns per (source, destination, pass) fill the Adder
}


# frisys_hh at a large budget, one rank, from 100 x Neel through the start-up regime into the compressed one (the 1-D stand-in for
# BASELINE config 5): same tuple as HH_RUNS.  gs_energy -1 makes the population grow (with -7.5 it dies out at 900 states).
# 150 iterations: zero-valued entries are never released in this driver (frisys_hh.cpp:321 vs frisys_mol.cpp:498), the table keeps
# growing past the budget, and the reference itself segfaults at iteration 164 of this run.
HH_SCALE_RUNS = {
    "hh_l12_m1e6": (1, 150, 3, 12, 12, 0.005, 4.0, 1.0, 0.7, -1.0, 1000000, 8000000, 1.0, 250000.0),
}


def gen_hh_scale(manifest):
    manifest["hh_scale_runs"] = {}
    for name, (n_ranks, n_iter, seed, n_elec, n_sites, eps, U, omega, g, gs, vnz, maxd, ini, tgt) in HH_SCALE_RUNS.items():
        out = os.path.join(GOLD, name + ".traj")
        cmd = [HARNESS, "hh", str(n_iter), str(seed), str(n_elec), str(n_sites), repr(eps), repr(U), repr(omega), repr(g), repr(gs), str(vnz), str(maxd), repr(ini), repr(tgt), out]
        subprocess.run(cmd, check=True)
        manifest["hh_scale_runs"][name] = dict(n_ranks=n_ranks, n_iter=n_iter, seed=seed, n_elec=n_elec, n_sites=n_sites, eps=eps, U=U, omega=omega, g=g,
                                               gs_energy=gs, vec_nonz=vnz, max_dets=maxd, initiator=ini, target_norm=tgt)


# --load_dir through the reference (ref_harness reload): name -> (shape, n1, n2, seed, eps, vec_nonz, mat_nonz, max_dets, initiator, target, distribution)
RELOAD_RUNS = {
    "ne_m2000_reload": ("Ne", 60, 30, 20250215, 0.01, 2000, 2000, 20000, 1.0, 150.0, "HB_unnorm"),
}
# the same with a dense space: the first run gets --det_space, the restarted one takes the space from the checkpoint's dense.txt (frisys_mol.cpp:234, :258);
# name -> (tuple as above, det-space file of DENSE_RUNS)
RELOAD_DENSE_RUNS = {
    "ne_m2000_dense_reload": (("Ne", 40, 30, 33, 0.01, 2000, 20000, 20000, 1.0, 150.0, "HB_unnorm"), "ne_m2000_dense_space.txt"),
}


# --det_space (semi-stochastic): name -> ((shape, n_iter, seed, eps, vec_nonz, mat_nonz, max_dets, initiator, target, dist), n_dense): the dense space is the
# first n_dense determinants of H|HF> (HF first), written as the integers DistVec::init_dense reads
DENSE_RUNS = {
    # (mat_nonz is the budget INCLUDING the dense block of H: frisys_mol.cpp:421 hands mat_nonz - tot_dense_h to the compression)
    "ne_m2000_dense": (("Ne", 60, 33, 0.01, 2000, 20000, 20000, 1.0, 1000.0, "HB_unnorm"), 12),
    "n2_m10000_dense_hb": (("N2", 40, 5, 0.01, 10000, 120000, 80000, 2.0, 5000.0, "HB"), 40),
}


DENSE_RANKS = {"ne_m2000_dense": (2, 3)}


def gen_dense(manifest):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    manifest["dense_runs"] = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, ((shape, n_iter, seed, eps, vnz, mnz, maxd, ini, tgt, dist), n_dense) in DENSE_RUNS.items():
            mol = fcidump.synthetic(shape)
            path = os.path.join(tmp, shape + ".FCIDUMP")
            fcidump.write_fcidump(path, mol)
            orc = oracle_lib.OracleFrisys(mol, epsilon=0.01, vec_nonz=10, mat_nonz=10, max_dets=100, seed=1)
            hd, _ = orc.htrial()
            space = name + "_space.txt"
            with open(os.path.join(GOLD, space), "w") as f:
                f.write("".join("%d\n" % int(d) for d in hd[:n_dense]))
            env = dict(os.environ, FRIES_DETSPACE=os.path.join(GOLD, space), FRIES_DETSPACE_DIR=tmp + "/")
            subprocess.run([HARNESS, "frisys", path, mol.point_group, str(n_iter), str(seed), repr(eps), str(vnz), str(mnz), str(maxd), repr(ini), repr(tgt), dist,
                            os.path.join(GOLD, name + ".traj")], check=True, env=env)
            manifest["dense_runs"][name] = dict(shape=shape, n_iter=n_iter, seed=seed, epsilon=eps, vec_nonz=vnz, mat_nonz=mnz, max_dets=maxd, initiator=ini,
                                                target_norm=tgt, distribution=dist, det_space=space, n_dense=n_dense)
            # the same run under mpiexec: rank 0 reads the file, the dense determinants travel to their owners, every rank holds its share
            for n_ranks in DENSE_RANKS.get(name, ()):
                rname = "%s_p%d" % (name, n_ranks)
                subprocess.run([MPIEXEC, "-n", str(n_ranks), HARNESS, "frisys_mpi", path, mol.point_group, str(n_iter), str(seed), repr(eps), str(vnz), str(mnz), str(maxd),
                                repr(ini), repr(tgt), dist, os.path.join(GOLD, rname + ".traj")], check=True, env=env)
                manifest.setdefault("dense_mpi_runs", {})[rname] = dict(manifest["dense_runs"][name], n_ranks=n_ranks)


def gen_reload(manifest):
    manifest["reload_runs"] = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, (shape, n1, n2, seed, eps, vnz, mnz, maxd, ini, tgt, dist) in RELOAD_RUNS.items():
            mol = fcidump.synthetic(shape)
            path = os.path.join(tmp, shape + ".FCIDUMP")
            fcidump.write_fcidump(path, mol)
            ck = os.path.join(tmp, name + "_ck") + "/"
            os.makedirs(ck)
            subprocess.run([HARNESS, "reload", path, mol.point_group, str(n1), str(seed), repr(eps), str(vnz), str(mnz), str(maxd), repr(ini), repr(tgt), dist,
                            os.path.join(GOLD, name + ".traj"), str(n2), ck], check=True)
            manifest["reload_runs"][name] = dict(shape=shape, n1=n1, n2=n2, seed=seed, epsilon=eps, vec_nonz=vnz, mat_nonz=mnz, max_dets=maxd, initiator=ini,
                                                 target_norm=tgt, distribution=dist)
        for name, ((shape, n1, n2, seed, eps, vnz, mnz, maxd, ini, tgt, dist), space) in RELOAD_DENSE_RUNS.items():
            mol = fcidump.synthetic(shape)
            path = os.path.join(tmp, shape + ".FCIDUMP")
            fcidump.write_fcidump(path, mol)
            ck = os.path.join(tmp, name + "_ck") + "/"
            os.makedirs(ck)
            env = dict(os.environ, FRIES_DETSPACE=os.path.join(GOLD, space), FRIES_DETSPACE_DIR=tmp + "/")
            subprocess.run([HARNESS, "reload", path, mol.point_group, str(n1), str(seed), repr(eps), str(vnz), str(mnz), str(maxd), repr(ini), repr(tgt), dist,
                            os.path.join(GOLD, name + ".traj"), str(n2), ck], check=True, env=env)
            manifest["reload_runs"][name] = dict(shape=shape, n1=n1, n2=n2, seed=seed, epsilon=eps, vec_nonz=vnz, mat_nonz=mnz, max_dets=maxd, initiator=ini,
                                                 target_norm=tgt, distribution=dist, det_space=space)


def gen_adder(manifest, tmp):
    manifest["adder_runs"] = {}
    for name, (n_ranks, adder, (shape, n_iter, seed, eps, vnz, mnz, maxd, ini, tgt, dist)) in ADDER_RUNS.items():
        mol = fcidump.synthetic(shape)
        path = os.path.join(tmp, shape + ".FCIDUMP")
        if not os.path.exists(path):
            fcidump.write_fcidump(path, mol)
        out = os.path.join(GOLD, name + ".traj")
        cmd = [MPIEXEC, "-n", str(n_ranks), HARNESS, "frisys_mpi", path, mol.point_group, str(n_iter), str(seed), repr(eps), str(vnz), str(mnz),
               str(maxd), repr(ini), repr(tgt), dist, out]
        subprocess.run(cmd, check=True, env=dict(os.environ, FRIES_ADDER_SIZE=str(adder)))
        manifest["adder_runs"][name] = dict(n_ranks=n_ranks, adder_size=adder, shape=shape, n_iter=n_iter, seed=seed, epsilon=eps, vec_nonz=vnz, mat_nonz=mnz,
                                            max_dets=maxd, initiator=ini, target_norm=tgt, distribution=dist)


def gen_pin(manifest, only=None):
    manifest.setdefault("pin_runs", {})
    with tempfile.TemporaryDirectory() as tmp:
        for name, (n_ranks, shape, seed, eps, m, maxd, dist, run_seed, n_iter) in PIN_RUNS.items():
            if only and name not in only:
                continue
            mol = fcidump.synthetic(shape)
            path = os.path.join(tmp, shape + ".FCIDUMP")
            fcidump.write_fcidump(path, mol)
            out = os.path.join(GOLD, name + ".pin")
            cmd = [HARNESS, "pin", path, mol.point_group, str(seed), repr(eps), str(m), str(maxd), dist, str(run_seed), str(n_iter), out]
            if n_ranks > 1:
                cmd = [MPIEXEC, "-n", str(n_ranks)] + cmd
            r = subprocess.run(cmd, check=True, capture_output=True, text=True)
            timing = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
            manifest["pin_runs"][name] = dict(n_ranks=n_ranks, shape=shape, seed=seed, epsilon=eps, m=m, max_dets=maxd, distribution=dist, run_seed=run_seed,
                                              n_iter=n_iter, reference_iters_per_s_build_container=timing["iters_per_s"], filler_iters=timing["filler_iters"])
            print("pin", name, timing, flush=True)


def gen_hbpiv(manifest):
    manifest["hbpiv_runs"] = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, (run, n_iter, cases) in HBPIV_RUNS.items():
            shape, _, seed, eps, vnz, mnz, maxd, ini, tgt, dist, _ = RUNS[run]
            mol = fcidump.synthetic(shape)
            path = os.path.join(tmp, shape + ".FCIDUMP")
            fcidump.write_fcidump(path, mol)
            out = os.path.join(GOLD, name + ".txt")
            cmd = [HARNESS, "hbpiv", path, mol.point_group, str(n_iter), str(seed), repr(eps), str(vnz), str(mnz), str(maxd), repr(ini), repr(tgt), dist, out]
            for ns, ps in cases:
                cmd += [str(ns), str(ps)]
            subprocess.run(cmd, check=True)
            manifest["hbpiv_runs"][name] = dict(run=run, n_iter=n_iter, cases=[list(c) for c in cases])


def gen_tr(manifest):
    manifest["tr_runs"] = {}
    manifest["hbpiv_tr_runs"] = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, (shape, seed, n_src) in TR_RUNS.items():
            mol = fcidump.synthetic(shape)
            path = os.path.join(tmp, shape + ".FCIDUMP")
            fcidump.write_fcidump(path, mol)
            subprocess.run([HARNESS, "tr", path, mol.point_group, str(seed), str(n_src), os.path.join(GOLD, name + ".txt")], check=True)
            manifest["tr_runs"][name] = dict(shape=shape, seed=seed, n_src=n_src)
        for name, (run, n_iter, parity, cases) in HBPIV_TR_RUNS.items():
            shape, _, seed, eps, vnz, mnz, maxd, ini, tgt, dist, _ = RUNS[run]
            mol = fcidump.synthetic(shape)
            path = os.path.join(tmp, shape + ".FCIDUMP")
            fcidump.write_fcidump(path, mol)
            cmd = [HARNESS, "hbpiv", path, mol.point_group, str(n_iter), str(seed), repr(eps), str(vnz), str(mnz), str(maxd), repr(ini), repr(tgt), dist, os.path.join(GOLD, name + ".txt")]
            for ns, ps in cases:
                cmd += [str(ns), str(ps)]
            subprocess.run(cmd, check=True, env=dict(os.environ, FRIES_SPIN_PARITY=str(parity)))
            manifest["hbpiv_tr_runs"][name] = dict(run=run, n_iter=n_iter, spin_parity=parity, cases=[list(c) for c in cases])


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--only-reload":
        with open(os.path.join(GOLD, "manifest.json")) as f:
            manifest = json.load(f)
        gen_reload(manifest)
        with open(os.path.join(GOLD, "manifest.json"), "w") as f:
            json.dump(manifest, f, indent=1, sort_keys=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--only-dense":
        with open(os.path.join(GOLD, "manifest.json")) as f:
            manifest = json.load(f)
        gen_dense(manifest)
        with open(os.path.join(GOLD, "manifest.json"), "w") as f:
            json.dump(manifest, f, indent=1, sort_keys=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--only-hh-scale":
        with open(os.path.join(GOLD, "manifest.json")) as f:
            manifest = json.load(f)
        gen_hh_scale(manifest)
        with open(os.path.join(GOLD, "manifest.json"), "w") as f:
            json.dump(manifest, f, indent=1, sort_keys=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--only-full-mpi":
        manifest = json.load(open(os.path.join(GOLD, "manifest.json")))
        gen_full_mpi(manifest)
        json.dump(manifest, open(os.path.join(GOLD, "manifest.json"), "w"), indent=1, sort_keys=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "--only-adder":      # the early-flush runs alone
        manifest = json.load(open(os.path.join(GOLD, "manifest.json")))
        with tempfile.TemporaryDirectory() as tmp:
            gen_adder(manifest, tmp)
        json.dump(manifest, open(os.path.join(GOLD, "manifest.json"), "w"), indent=1, sort_keys=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "--only-pin":      # minutes of CPU each: the BASELINE sizes
        with open(os.path.join(GOLD, "manifest.json")) as f:
            manifest = json.load(f)
        gen_pin(manifest, only=sys.argv[2:] or None)
        with open(os.path.join(GOLD, "manifest.json"), "w") as f:
            json.dump(manifest, f, indent=1, sort_keys=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] in ("--only-multi", "--only-fp", "--only-hhfull"):
        with open(os.path.join(GOLD, "manifest.json")) as f:
            manifest = json.load(f)
        {"--only-multi": gen_multi, "--only-fp": gen_fp, "--only-hhfull": gen_hhfull}[sys.argv[1]](manifest)
        with open(os.path.join(GOLD, "manifest.json"), "w") as f:
            json.dump(manifest, f, indent=1, sort_keys=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--only-tr":
        with open(os.path.join(GOLD, "manifest.json")) as f:
            manifest = json.load(f)
        gen_tr(manifest)
        with open(os.path.join(GOLD, "manifest.json"), "w") as f:
            json.dump(manifest, f, indent=1, sort_keys=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--only-hbpiv":     # add these fixtures to the existing manifest without re-running the rest
        with open(os.path.join(GOLD, "manifest.json")) as f:
            manifest = json.load(f)
        subprocess.run([HARNESS, "hbpp_all", os.path.join(GOLD, "hbpp_all.txt")], check=True)
        gen_hbpiv(manifest)
        with open(os.path.join(GOLD, "manifest.json"), "w") as f:
            json.dump(manifest, f, indent=1, sort_keys=True)
        return
    os.makedirs(GOLD, exist_ok=True)
    manifest = {"runs": {}, "ints": {}}
    subprocess.run([HARNESS, "unit"], check=True)
    subprocess.run([HARNESS, "hbpp_all", os.path.join(GOLD, "hbpp_all.txt")], check=True)
    subprocess.run([HARNESS, "piv", os.path.join(GOLD, "piv_comp.txt")], check=True)
    with tempfile.TemporaryDirectory() as tmp:
        for shape in ("Ne", "N2", "H2O", "MAX32", "MIN4"):
            mol = fcidump.synthetic(shape)
            path = os.path.join(tmp, shape + ".FCIDUMP")
            fcidump.write_fcidump(path, mol)
            dump = os.path.join(tmp, shape + ".ints")
            subprocess.run([HARNESS, "dump_ints", path, mol.point_group, dump], check=True)
            manifest["ints"][shape] = {"sha256": hashlib.sha256(open(dump, "rb").read()).hexdigest(), "point_group": mol.point_group}
        for name, (shape, n_iter, seed, eps, vnz, mnz, maxd, ini, tgt, dist, snap) in RUNS.items():
            mol = fcidump.synthetic(shape)
            path = os.path.join(tmp, shape + ".FCIDUMP")
            out = os.path.join(GOLD, name + ".traj")
            cmd = [HARNESS, "frisys", path, mol.point_group, str(n_iter), str(seed), repr(eps), str(vnz), str(mnz), str(maxd), repr(ini), repr(tgt), dist, out]
            if snap:
                cmd.append(str(snap))
            env = dict(os.environ)
            if name in CHECKPOINTS:     # DistVec::save of the reference after that many iterations
                ckdir = os.path.join(tmp, name + "_ck") + "/"
                os.makedirs(ckdir)
                env.update(FRIES_SAVE_DIR=ckdir, FRIES_SAVE_AT=str(CHECKPOINTS[name]))
            subprocess.run(cmd, check=True, env=env)
            if name in CHECKPOINTS:
                manifest.setdefault("checkpoints", {})[name] = dict(
                    after_iterations=CHECKPOINTS[name], dense_txt=open(ckdir + "dense.txt").read(),
                    **{fn.replace(".", "_") + "_sha256": hashlib.sha256(open(ckdir + fn, "rb").read()).hexdigest() for fn in ("dets0.dat", "vals0.dat")},
                    **{fn.replace(".", "_") + "_bytes": os.path.getsize(ckdir + fn) for fn in ("dets0.dat", "vals0.dat")})
            manifest["runs"][name] = dict(shape=shape, n_iter=n_iter, seed=seed, epsilon=eps, vec_nonz=vnz, mat_nonz=mnz, max_dets=maxd,
                                          initiator=ini, target_norm=tgt, distribution=dist)
        # text vectors for --trial_vec / --ini_vec: the first 25 entries of H|HF> of the N2 shape (HF first), committed as fixtures
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        import numpy as np
        mol = fcidump.synthetic("N2")
        orc = oracle_lib.OracleFrisys(mol, epsilon=0.01, vec_nonz=10, mat_nonz=10, max_dets=100, seed=1)
        hd, hv = orc.htrial()
        with open(os.path.join(GOLD, "n2_trial_dets"), "w") as f:
            f.write("".join("%d\n" % int(d) for d in hd[:25]))
        with open(os.path.join(GOLD, "n2_trial_vals"), "w") as f:
            f.write("".join("%r\n" % (1.0 if i == 0 else float(-0.05 * np.sign(hv[i]) * (1 + 0.01 * i))) for i in range(25)))
        with open(os.path.join(GOLD, "n2_ini_dets"), "w") as f:
            f.write("".join("%d\n" % int(d) for d in hd[:25][::-1]))
        with open(os.path.join(GOLD, "n2_ini_vals"), "w") as f:
            f.write("".join("%r\n" % (float(60.0 if i == 24 else -3.0 * np.sign(hv[24 - i]))) for i in range(25)))
        manifest["extra_runs"] = {}
        for name, ((shape, n_iter, seed, eps, vnz, mnz, maxd, ini, tgt, dist), extra) in EXTRA_RUNS.items():
            mol = fcidump.synthetic(shape)
            path = os.path.join(tmp, shape + ".FCIDUMP")
            out = os.path.join(GOLD, name + ".traj")
            env = dict(os.environ)
            if "trial" in extra:
                env["FRIES_TRIAL"] = os.path.join(GOLD, extra["trial"])
            if "ini" in extra:
                env["FRIES_INI"] = os.path.join(GOLD, extra["ini"])
            if "ham_shift" in extra:
                env["FRIES_HAM_SHIFT"] = repr(extra["ham_shift"])
            subprocess.run([HARNESS, "frisys", path, mol.point_group, str(n_iter), str(seed), repr(eps), str(vnz), str(mnz), str(maxd), repr(ini), repr(tgt), dist, out],
                           check=True, env=env)
            manifest["extra_runs"][name] = dict(shape=shape, n_iter=n_iter, seed=seed, epsilon=eps, vec_nonz=vnz, mat_nonz=mnz, max_dets=maxd,
                                                initiator=ini, target_norm=tgt, distribution=dist, **extra)
        manifest["mpi_runs"] = {}
        for name, (n_ranks, (shape, n_iter, seed, eps, vnz, mnz, maxd, ini, tgt, dist)) in MPI_RUNS.items():
            mol = fcidump.synthetic(shape)
            path = os.path.join(tmp, shape + ".FCIDUMP")
            out = os.path.join(GOLD, name + ".traj")
            cmd = [MPIEXEC, "-n", str(n_ranks), HARNESS, "frisys_mpi", path, mol.point_group, str(n_iter), str(seed), repr(eps), str(vnz), str(mnz),
                   str(maxd), repr(ini), repr(tgt), dist, out]
            subprocess.run(cmd, check=True)
            manifest["mpi_runs"][name] = dict(n_ranks=n_ranks, shape=shape, n_iter=n_iter, seed=seed, epsilon=eps, vec_nonz=vnz, mat_nonz=mnz,
                                              max_dets=maxd, initiator=ini, target_norm=tgt, distribution=dist)
        gen_adder(manifest, tmp)
        gen_full_mpi(manifest)
        manifest["fciqmc_runs"] = {}
        for name, (shape, n_iter, seed, eps, tw, maxd, ini, dist) in FCIQMC_RUNS.items():
            mol = fcidump.synthetic(shape)
            path = os.path.join(tmp, shape + ".FCIDUMP")
            out = os.path.join(GOLD, name + ".traj")
            subprocess.run([HARNESS, "fciqmc", path, mol.point_group, str(n_iter), str(seed), repr(eps), str(tw), str(maxd), str(ini), out, dist], check=True)
            manifest["fciqmc_runs"][name] = dict(shape=shape, n_iter=n_iter, seed=seed, epsilon=eps, target_walkers=tw, max_dets=maxd, initiator=ini, distribution=dist)
        # fciqmc_mol with --trial_vec / --ini_vec: the N2 trial fixture above and an integer start vector over its first 12 determinants
        with open(os.path.join(GOLD, "n2_fq_ini_dets"), "w") as f:
            f.write("".join("%d\n" % int(d) for d in hd[:12]))
        with open(os.path.join(GOLD, "n2_fq_ini_vals"), "w") as f:
            f.write("".join("%d\n" % (40 if i == 0 else (-3 if i % 2 else 5)) for i in range(12)))
        name = "fciqmc_n2_trial_ini"
        mol = fcidump.synthetic("N2")
        path = os.path.join(tmp, "N2.FCIDUMP")
        out = os.path.join(GOLD, name + ".traj")
        env = dict(os.environ, FRIES_TRIAL=os.path.join(GOLD, "n2_trial_"), FRIES_INI=os.path.join(GOLD, "n2_fq_ini_"))
        subprocess.run([HARNESS, "fciqmc", path, mol.point_group, "150", "9", "0.004", "20000", "100000", "2", out, "NU"], check=True, env=env)
        manifest["fciqmc_runs"][name] = dict(shape="N2", n_iter=150, seed=9, epsilon=0.004, target_walkers=20000, max_dets=100000, initiator=2, distribution="NU",
                                             trial="n2_trial_", ini="n2_fq_ini_")
        manifest["fciqmc_mpi_runs"] = {}
        for name, (n_ranks, (shape, n_iter, seed, eps, tw, maxd, ini, dist)) in FCIQMC_MPI_RUNS.items():
            mol = fcidump.synthetic(shape)
            path = os.path.join(tmp, shape + ".FCIDUMP")
            out = os.path.join(GOLD, name + ".traj")
            subprocess.run([MPIEXEC, "-n", str(n_ranks), HARNESS, "fciqmc", path, mol.point_group, str(n_iter), str(seed), repr(eps), str(tw), str(maxd), str(ini), out, dist], check=True)
            manifest["fciqmc_mpi_runs"][name] = dict(n_ranks=n_ranks, shape=shape, n_iter=n_iter, seed=seed, epsilon=eps, target_walkers=tw, max_dets=maxd, initiator=ini, distribution=dist)
        manifest["full_runs"] = {}
        for name, (shape, n_iter, seed, eps, vnz, maxd, tgt) in FULL_RUNS.items():
            mol = fcidump.synthetic(shape)
            path = os.path.join(tmp, shape + ".FCIDUMP")
            out = os.path.join(GOLD, name + ".traj")
            subprocess.run([HARNESS, "frifull", path, mol.point_group, str(n_iter), str(seed), repr(eps), str(vnz), str(maxd), repr(tgt), out], check=True)
            manifest["full_runs"][name] = dict(shape=shape, n_iter=n_iter, seed=seed, epsilon=eps, vec_nonz=vnz, max_dets=maxd, target_norm=tgt)
        manifest["hh_runs"] = {}
        for name, (n_ranks, n_iter, seed, n_elec, n_sites, eps, U, omega, g, gs, vnz, maxd, ini, tgt) in HH_RUNS.items():
            out = os.path.join(GOLD, name + ".traj")
            cmd = [HARNESS, "hh", str(n_iter), str(seed), str(n_elec), str(n_sites), repr(eps), repr(U), repr(omega), repr(g), repr(gs), str(vnz), str(maxd),
                   repr(ini), repr(tgt), out]
            if n_ranks > 1:
                cmd = [MPIEXEC, "-n", str(n_ranks)] + cmd
            subprocess.run(cmd, check=True)
            manifest["hh_runs"][name] = dict(n_ranks=n_ranks, n_iter=n_iter, seed=seed, n_elec=n_elec, n_sites=n_sites, eps=eps, U=U, omega=omega, g=g,
                                             gs_energy=gs, vec_nonz=vnz, max_dets=maxd, initiator=ini, target_norm=tgt)
    gen_hhfull(manifest)
    gen_hbpiv(manifest)
    gen_multi(manifest)
    gen_fp(manifest)
    gen_pin(manifest)
    gen_hh_scale(manifest)
    gen_reload(manifest)
    gen_tr(manifest)
    with open(os.path.join(GOLD, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("golden fixtures written to", GOLD)


if __name__ == "__main__":
    main()
