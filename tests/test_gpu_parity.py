"""GPU suite (-m gpu): the hand-written HIP path, called through the C ABI, against the CPU oracle
on the same seeded inputs and against the golden vectors the real reference produced.
Bar: integers (positions, determinants, counts) and stored doubles bit-exact; projected energy
numer/denom identical on one rank (the products are added in list order), within 1e-10 with ranks and for the drivers that gather partial sums."""
import os

import numpy as np
import pytest

import golden_io
from fries_amd import fcidump

pytestmark = pytest.mark.gpu
ENERGY_TOL = 1e-10


@pytest.fixture(scope="module")
def Engine():
    from fries_amd.engine import FriEngine
    return FriEngine


def rand_dets(rng, n_orb, n_elec, n):
    out = np.zeros(n, dtype=np.uint64)
    for i in range(n):
        d = 0
        for sp in range(2):
            for o in rng.choice(n_orb, n_elec // 2, replace=False):
                d |= 1 << (int(o) + sp * n_orb)
        out[i] = d
    return out


def compare_iter(lg, lo, eng, orc, check_values=True):
    for f in ("num_success", "n_nonz", "curr_size", "nkept"):
        assert int(lg[f]) == int(lo[f]), f
    assert float(lg["norm"]) == float(lo["norm"]) and float(lg["shift"]) == float(lo["shift"])
    assert abs(lg["numer"] / lg["denom"] - lo["numer"] / lo["denom"]) < ENERGY_TOL
    assert int(lg["err"]) == 0
    if check_values:
        gd, gv = eng.vector()
        cd, cv = orc.vector()
        nz = cv != 0
        assert gd.size == cd.size
        assert np.array_equal(gv, cv)
        assert np.array_equal(gd[nz], cd[nz])


@pytest.mark.parametrize("shape", ["Ne", "N2", "H2O"])
def test_system_tables_and_matrix_elements(Engine, oracle, mols, shape):
    mol = mols(shape)
    eng = Engine(mol)
    orc = oracle.OracleFrisys(mol, epsilon=0.01, vec_nonz=100, mat_nonz=100, max_dets=1000, seed=1)
    assert eng.hf_energy == orc.hf_energy
    for w in range(7):
        assert np.array_equal(eng.hb_tensor(w), orc.hb_tensor(w)), w
    rng = np.random.RandomState(5)
    dets = rand_dets(rng, mol.n_orb, mol.n_elec, 3000)
    a, _ = eng.matrel(0, dets)
    b, _ = orc.matrel(0, dets)
    assert np.array_equal(a, b)
    # singles / doubles with random (not nec. symmetry-allowed) orbitals: arithmetic parity only
    n = mol.n_orb
    orbs = np.zeros((dets.size, 4), dtype=np.uint8)
    for i, d in enumerate(dets):
        occ = [k for k in range(2 * n) if (int(d) >> k) & 1]
        vir = [k for k in range(2 * n) if not (int(d) >> k) & 1]
        o = sorted(rng.choice(occ, 2, replace=False))
        u = sorted(rng.choice([v for v in vir if v // n == o[0] // n] if i % 2 else vir, 2, replace=False))
        orbs[i] = [o[0], o[1], u[0], u[1]]
    so = orbs.copy()
    for i in range(dets.size):                      # single: same spin
        sp = so[i, 0] // n
        cand = [k for k in range(sp * n, sp * n + n) if not (int(dets[i]) >> k) & 1]
        so[i, 1] = cand[rng.randint(len(cand))]
    for kind, ob in ((1, so), (2, orbs)):
        a, sa = eng.matrel(kind, dets, ob)
        b, sb = orc.matrel(kind, dets, ob)
        assert np.array_equal(a, b) and np.array_equal(sa, sb), kind
    eng.close()


def test_comb_teeth_are_the_reference_sequence(Engine, mols):
    """Tooth k must equal fl(fl(r0 + u) + u ...) (compress_utils.cpp:318), not r0 + k*u."""
    eng = Engine(mols("Ne"))
    rng = np.random.RandomState(2)
    for r0f, unit, n in [(0.37, 1.234567e-3, 300000), (0.0, 0.1, 5000), (0.999, 7.7e-7, 100000), (0.5, 1.0, 70000), (1e-9, 3.3, 1000)]:
        r0 = r0f * unit
        ref = np.empty(n)
        x = r0
        for k in range(n):
            ref[k] = x
            x = x + unit
        q = np.sort(np.concatenate([rng.random_sample(3000) * ref[-1] * 1.01, ref[rng.randint(0, n, 200)]]))
        pos, below = eng.test_teeth(r0, unit, n, q)
        assert np.array_equal(pos, ref)
        assert np.array_equal(below, np.searchsorted(ref, q, side="left"))
    eng.close()


def test_exact_sequential_sums(Engine, mols):
    """seqsum.hpp against numpy's left-to-right cumsum, bit for bit."""
    eng = Engine(mols("Ne"))
    rng = np.random.RandomState(3)
    cases = [rng.random_sample(5), rng.random_sample(1024), np.exp(8 * rng.random_sample(70001)) * (rng.random_sample(70001) > 0.1),
             rng.random_sample(1000003), np.full(300000, 0.1234567), np.ldexp(rng.randint(1, 8, 200000).astype(float), -3),
             np.where(rng.random_sample(100000) > 0.9, rng.random_sample(100000), 0.0), np.exp(20 * rng.random_sample(2000000))]
    for a in cases:
        out, tot, dirty_tiles, dirty_subs = eng.test_seqsum(a)
        ref = np.cumsum(a)
        assert np.array_equal(out, ref) and tot == ref[-1]
        assert dirty_subs < 200
    eng.close()


def test_new_hb_all_known_answer_on_gpu(Engine, mols):
    """Reference tests/test_hamiltonian.cpp:454-520 through the HIP path: every excitation once, |value| = 1,
    orbitals in the reference's order."""
    g = golden_io.read_hbpp_all()
    mol = mols("Ne")
    mol2 = fcidump.MolInput(22, 8, np.array(golden_io.HBPP_ALL_SYMM, dtype=np.uint8), mol.h_core, mol.eris, 0.0, "D2h")
    n_ex = 22 * 22 * 8 * 8
    eng = Engine(mol2)
    eng.setup(epsilon=0.01, vec_nonz=10, mat_nonz=n_ex, max_dets=64, seed=0, distribution="HB_unnorm")
    for k, v in g["tens"].items():
        eng.set_hb_tensor(golden_io.TENSOR_ID[k], np.array(v))
    hf = (1 << 4) - 1 | (((1 << 4) - 1) << 22)
    eng.vec_load(np.array([hf], dtype=np.uint64), np.array([1.0]))
    # p_doub of this test is 0.95, not the HF ratio: go through the oracle-free path by scaling afterwards is not
    # possible, so the engine exposes it via setup only; reproduce it by loading the tensors and comparing orbitals
    pos, orbs, vals, _ = eng.apply_hbpp_sys(n_ex, g["rn"], unit_matrel=True)
    assert pos.size == 984 and np.all(pos == 0)
    assert sorted(map(tuple, orbs.tolist())) == sorted(map(tuple, g["orbs"].tolist()))
    # second half of the reference test (:507-519): the pivotal variant with the same budget keeps everything, too
    ppos, porbs, pvals, sl = eng.apply_hbpp_piv(n_ex, unit_matrel=True)
    assert ppos.size == 984 and np.all(ppos == 0) and np.all(np.abs(np.abs(pvals) - 1) < 1e-7)
    assert np.array_equal(porbs, orbs)
    eng.close()


@pytest.mark.parametrize("name", sorted(golden_io.manifest()["hbpiv_runs"]))
def test_apply_hbpp_piv_matches_reference(Engine, name):
    """apply_HBPP_piv on the device (every HB-PP factor multiplied out, pivotal compression, collapse) against what the reference's
    function returned on the vector of a golden frisys_mol run (tests/golden/hbpiv_*.txt): positions, orbitals, values bit for bit and
    the length after every compression; budgets below, near and above the number of elements."""
    h = golden_io.manifest()["hbpiv_runs"][name]
    r = golden_io.manifest()["runs"][h["run"]]
    mol = fcidump.synthetic(r["shape"])
    eng = Engine(mol)
    eng.setup(epsilon=r["epsilon"], vec_nonz=r["vec_nonz"], mat_nonz=r["mat_nonz"], max_dets=r["max_dets"],
              target_norm=r["target_norm"], initiator=r["initiator"], seed=r["seed"], distribution=r["distribution"])
    logs = eng.iterate(h["n_iter"])
    g = golden_io.read_traj(h["run"])
    assert float(logs[-1]["norm"]) == g["rows"][h["n_iter"] - 1]["norm"]
    for c in golden_io.read_hbpiv(name):
        eng.restart(c["seed"])
        pos, orbs, vals, sl = eng.apply_hbpp_piv(c["n_samp"])
        assert sl.tolist() == c["stage_len"], (name, c["n_samp"], sl.tolist(), c["stage_len"])
        assert len(pos) == c["n_out"]
        assert np.array_equal(pos, c["pos"]) and np.array_equal(orbs, c["orbs"])
        assert vals.tobytes() == c["val"].tobytes(), (name, c["n_samp"], int(np.sum(vals != c["val"])))
    eng.close()


@pytest.mark.parametrize("name", sorted(golden_io.manifest().get("tr_runs", {})))
def test_time_reversal_h_op_offdiag_matches_reference(Engine, name):
    """Time-reversal symmetry on the device (fries_set_spin_parity; k_enum applies the adjust_tr rule per candidate, csrc/hbpp_rows.hpp:
    fr_adjust_tr): the full off-diagonal action on the fixture's source vector for spin parity +1 and -1 against what the reference's
    h_op_offdiag left in its vector -- stored determinants in order, values bit for bit."""
    r = golden_io.manifest()["tr_runs"][name]
    eng = Engine(fcidump.synthetic(r["shape"]))
    eng.setup(epsilon=0.01, vec_nonz=100, mat_nonz=100, max_dets=1000, seed=1, distribution="HB_unnorm")
    sd, sv, out = golden_io.read_tr(name)
    for sp in (1, -1):
        eng.set_spin_parity(sp)
        d, v = eng.h_offdiag_list(sd, sv)
        assert np.array_equal(d, out[sp][0]), (name, sp, d.size, out[sp][0].size)
        assert v.tobytes() == out[sp][1].tobytes(), (name, sp, int(np.sum(v != out[sp][1])))
    eng.set_spin_parity(0)
    d0, v0 = eng.h_offdiag_list(sd, sv)
    assert d0.size > out[1][0].size        # without the symmetry both members of every pair are stored
    eng.close()


@pytest.mark.parametrize("name", sorted(golden_io.manifest().get("hbpiv_tr_runs", {})))
def test_apply_hbpp_piv_time_reversal_matches_reference(Engine, name):
    """apply_HBPP_piv with spin_parity = +-1 on the device against the reference's function (heat_bathPP.cpp:1326-1407) on the vector of a
    golden frisys_mol run: positions, orbitals, values bit for bit."""
    h = golden_io.manifest()["hbpiv_tr_runs"][name]
    r = golden_io.manifest()["runs"][h["run"]]
    eng = Engine(fcidump.synthetic(r["shape"]))
    eng.setup(epsilon=r["epsilon"], vec_nonz=r["vec_nonz"], mat_nonz=r["mat_nonz"], max_dets=r["max_dets"],
              target_norm=r["target_norm"], initiator=r["initiator"], seed=r["seed"], distribution=r["distribution"])
    eng.iterate(h["n_iter"])
    eng.set_spin_parity(h["spin_parity"])
    for c in golden_io.read_hbpiv(name):
        eng.restart(c["seed"])
        pos, orbs, vals, sl = eng.apply_hbpp_piv(c["n_samp"])
        assert sl.tolist() == c["stage_len"] and len(pos) == c["n_out"], (name, c["n_samp"], len(pos), c["n_out"])
        assert np.array_equal(pos, c["pos"]) and np.array_equal(orbs, c["orbs"])
        assert vals.tobytes() == c["val"].tobytes(), (name, c["n_samp"], int(np.sum(vals != c["val"])))
    eng.close()


RUNS = [
    ("Ne", dict(epsilon=0.01, vec_nonz=2000, mat_nonz=2000, max_dets=20000, target_norm=1000.0, initiator=1.0, seed=20250215, distribution="HB_unnorm"), 70),
    ("N2", dict(epsilon=0.01, vec_nonz=10000, mat_nonz=10000, max_dets=80000, target_norm=5000.0, initiator=0.0, seed=7, distribution="HB_unnorm"), 40),
    ("H2O", dict(epsilon=0.005, vec_nonz=5000, mat_nonz=8000, max_dets=80000, target_norm=2000.0, initiator=3.0, seed=99, distribution="HB"), 50),
    ("N2", dict(epsilon=0.01, vec_nonz=100000, mat_nonz=100000, max_dets=600000, target_norm=30000.0, initiator=0.0, seed=5, distribution="HB_unnorm"), 25),
    ("N2", dict(epsilon=0.01, vec_nonz=50000, mat_nonz=120000, max_dets=600000, target_norm=30000.0, initiator=0.5, seed=6, distribution="HB"), 25),
    # edge cases: 32 spatial orbitals (every bit of the 64-bit index in use), and 2 electrons in 4 orbitals (one electron per spin)
    ("MAX32", dict(epsilon=0.01, vec_nonz=3000, mat_nonz=3000, max_dets=60000, target_norm=1500.0, initiator=1.0, seed=3, distribution="HB_unnorm"), 20),
    ("MAX32", dict(epsilon=0.01, vec_nonz=3000, mat_nonz=3000, max_dets=60000, target_norm=1500.0, initiator=1.0, seed=3, distribution="HB"), 20),
    ("MIN4", dict(epsilon=0.01, vec_nonz=50, mat_nonz=50, max_dets=1000, target_norm=25.0, initiator=1.0, seed=3, distribution="HB_unnorm"), 20),
    ("MIN4", dict(epsilon=0.01, vec_nonz=50, mat_nonz=50, max_dets=1000, target_norm=25.0, initiator=1.0, seed=3, distribution="HB"), 20),
]


@pytest.mark.parametrize("case", range(len(RUNS)))
def test_frisys_trajectory_matches_oracle(Engine, oracle, mols, case):
    shape, par, n_iter = RUNS[case]
    mol = mols(shape)
    eng = Engine(mol)
    eng.setup(**par)
    orc = oracle.OracleFrisys(mol, **par)
    assert eng.p_doub == orc.p_doub
    hd, hv = eng.htrial()
    od, ov = orc.htrial()
    assert np.array_equal(hd, od) and np.array_equal(hv, ov)
    for it in range(n_iter):
        lg, lo = eng.iterate(1)[0], orc.iterate(1)[0]
        compare_iter(lg, lo, eng, orc, check_values=(it % 5 == 4 or it < 3 or it == n_iter - 1))
    eng.close()


@pytest.mark.parametrize("name", sorted(golden_io.manifest()["runs"]))
def test_frisys_trajectory_matches_reference_golden(Engine, mols, name):
    """Directly against what the real reference logged (no oracle in between)."""
    r = golden_io.manifest()["runs"][name]
    g = golden_io.read_traj(name)
    eng = Engine(mols(r["shape"]))
    eng.setup(epsilon=r["epsilon"], vec_nonz=r["vec_nonz"], mat_nonz=r["mat_nonz"], max_dets=r["max_dets"], target_norm=r["target_norm"],
              initiator=r["initiator"], seed=r["seed"], distribution=r["distribution"])
    assert eng.p_doub == g["p_doub"] and eng.hf_energy == g["hf_en"]
    for row in g["rows"]:
        lg = eng.iterate(1)[0]
        for f in ("nkept", "n_nonz", "curr_size", "num_success"):
            assert int(lg[f]) == row[f], (row["it"], f)
        assert float(lg["norm"]) == row["norm"] and float(lg["shift"]) == row["shift"]
        # one rank: the products of the dot are added in list order on the device -> the reference's doubles, not merely close ones
        assert float(lg["numer"]) == row["numer"] and float(lg["denom"]) == row["denom"], (row["it"], float(lg["numer"]).hex(), row["numer"].hex())
        if row["it"] % 10 == 9:
            d, v = eng.vector()
            assert golden_io.vec_hash(d, v) == row["hash"], row["it"]
    eng.close()


REPLAY_KNOBS = [
    {"FRIES_FKS_NO_LIGHT": "1"},            # every wave decides in every replay (no margins used)
    {"FRIES_FKS_NO_EXT": "1"},              # a changed number of sweeps sends every wave back to deciding
    {"FRIES_GROUP_WARM_ALL": "0"},          # stages 2-5 start from the per-chunk profile
    {"FRIES_NO_GROUP_WARM": "1"},
    {"FRIES_FKS_FUSE_TOTALS": "1"},         # totals by the last workgroup of the scan
    {"FRIES_FKS_LIGHT_FULL_GRID": "0"},     # light replays on the persistent grid instead of one workgroup per tile
    {"FRIES_FKS_WARM_EXTRAP": "1.0"},
    {"FRIES_FKS_REC_AT": "3"},              # margins recorded by replay 3 instead of replay 1
    {"FRIES_FKS_NO_CLOSING": "1"},          # no closing pass: a confirming replay and a final pass of their own
    {"FRIES_FKS_NO_SPECULATION": "1"},      # nothing enqueued behind the closing pass before the host has seen its flag
    {"FRIES_WAIT_SYNC": "1"},               # the host waits with hipStreamSynchronize instead of polling the ticket word
    {"FRIES_FKS_GRID": "512", "FRIES_FKS_GRID0": "700"},     # other persistent grids (more tiles per workgroup)
    {"FRIES_NO_STAGING": "1"},              # k_sys_write replays sys_sub instead of copying the emissions k_sys_count staged
    {"FRIES_FKS_SEQ": "1"},                 # every stage in the reference's own order: guesses, exact chain, comparison (fks_seq.hpp)
    {"FRIES_FKS_SEQ": "1", "FRIES_FKS_SEQ_WALK": "1"},          # ... by the one-wave walk
    {"FRIES_FKS_SEQ": "1", "FRIES_FSQ_GUESS_ROUNDS": "1"},      # ... the chain run on the first guess
    {"FRIES_FKS_SEQ": "1", "FRIES_FSQ_GUESS_ROUNDS": "0"},      # ... and on no guess at all
    {"FRIES_FKS_SEQ": "1", "FRIES_FSQ_MAPS": "0", "FRIES_FSQ_SPARSE_MAX": "128"},      # ... every tile of the chain walked pair by pair
    {"FRIES_FKS_SEQ": "1", "FRIES_FSQ_MAPS": "0"},              # ... the chain element by element (no integer steps)
    {"FRIES_FKS_SEQ": "1", "FRIES_FSQ_CHECK": "1"},             # ... both forms of the chain, every block entry compared bit for bit
    {"FRIES_FKS_SEQ": "1", "FRIES_FSQ_GUESS_ROUNDS": "1", "FRIES_FSQ_EXACT_ROUNDS": "0"},      # ... the walk taking over at the first tile the comparison rejects
]


@pytest.mark.parametrize("knob", range(len(REPLAY_KNOBS)))
@pytest.mark.parametrize("name", ["n2_m30000_unnorm", "h2o_m5000_hb"])
def test_every_replay_strategy_reproduces_the_reference(Engine, mols, name, knob, monkeypatch):
    """find_keep_sub's fixed point is unique: whichever way the replay reaches it (which waves it lets stand, where it starts from, who adds
    up the totals), the trajectory is the reference's.  The knobs are read when the context is created."""
    for k, v in REPLAY_KNOBS[knob].items():
        monkeypatch.setenv(k, v)
    r = golden_io.manifest()["runs"][name]
    g = golden_io.read_traj(name)
    eng = Engine(mols(r["shape"]))
    eng.setup(epsilon=r["epsilon"], vec_nonz=r["vec_nonz"], mat_nonz=r["mat_nonz"], max_dets=r["max_dets"], target_norm=r["target_norm"],
              initiator=r["initiator"], seed=r["seed"], distribution=r["distribution"])
    for row in g["rows"]:
        lg = eng.iterate(1)[0]
        for f in ("nkept", "n_nonz", "curr_size", "num_success"):
            assert int(lg[f]) == row[f], (REPLAY_KNOBS[knob], row["it"], f)
        assert float(lg["norm"]) == row["norm"] and float(lg["shift"]) == row["shift"], (REPLAY_KNOBS[knob], row["it"])
        assert int(lg["err"]) == 0
    d, v = eng.vector()
    last = g["rows"][-1]
    if last["it"] % 10 == 9:
        assert golden_io.vec_hash(d, v) == last["hash"]
    eng.close()


def test_capacity_beyond_the_old_tile_limit_changes_nothing(Engine, mols):
    """max_dets = 4e7 (39 063 tiles of 1024 per array; FR_MAX_PART was 32 768 until round 3 and refused this) with the golden run's other parameters: the
    reference's trajectory, bit for bit -- the capacity sizes arrays and nothing else."""
    name = "n2_m30000_unnorm"
    r = golden_io.manifest()["runs"][name]
    g = golden_io.read_traj(name)
    eng = Engine(mols(r["shape"]))
    eng.setup(epsilon=r["epsilon"], vec_nonz=r["vec_nonz"], mat_nonz=r["mat_nonz"], max_dets=40_000_000, target_norm=r["target_norm"],
              initiator=r["initiator"], seed=r["seed"], distribution=r["distribution"])
    for row in g["rows"][:12]:
        lg = eng.iterate(1)[0]
        for f in ("nkept", "n_nonz", "curr_size", "num_success"):
            assert int(lg[f]) == row[f], (row["it"], f)
        assert float(lg["norm"]) == row["norm"] and float(lg["shift"]) == row["shift"], row["it"]
        assert int(lg["err"]) == 0
    eng.close()


@pytest.mark.parametrize("name", sorted(golden_io.manifest()["extra_runs"]))
def test_frisys_driver_options_match_reference_golden(Engine, mols, name):
    """--trial_vec (a 25-determinant trial vector: H * trial by full enumeration of every entry), --ini_vec and --ham_shift on
    the device against the reference's trajectory."""
    r = golden_io.manifest()["extra_runs"][name]
    g = golden_io.read_traj(name)
    kw = {}
    if "trial" in r:
        kw["trial"] = golden_io.read_text_vector(r["trial"])
    if "ini" in r:
        kw["ini"] = golden_io.read_text_vector(r["ini"])
    if "ham_shift" in r:
        kw["ham_shift"] = r["ham_shift"]
    eng = Engine(mols(r["shape"]))
    eng.setup(epsilon=r["epsilon"], vec_nonz=r["vec_nonz"], mat_nonz=r["mat_nonz"], max_dets=r["max_dets"], target_norm=r["target_norm"],
              initiator=r["initiator"], seed=r["seed"], distribution=r["distribution"], **kw)
    assert eng.p_doub == g["p_doub"]
    for row in g["rows"]:
        lg = eng.iterate(1)[0]
        for f in ("nkept", "n_nonz", "curr_size", "num_success"):
            assert int(lg[f]) == row[f], (row["it"], f)
        assert float(lg["norm"]) == row["norm"] and float(lg["shift"]) == row["shift"]
        assert float(lg["numer"]) == row["numer"] and float(lg["denom"]) == row["denom"], (row["it"], float(lg["numer"]).hex(), row["numer"].hex())
    d, v = eng.vector()
    assert golden_io.vec_hash(d, v) == g["rows"][-1]["hash"]
    eng.close()


@pytest.mark.parametrize("name", sorted(golden_io.manifest()["full_runs"]))
def test_frifull_trajectory_matches_reference_golden(Engine, mols, name):
    """frifull_mol on the device (vector compression, then every single and double excitation of every determinant merged in
    the reference's order) against what the real reference logged: counts, norms, shifts and stored values bit for bit."""
    r = golden_io.manifest()["full_runs"][name]
    g = golden_io.read_traj(name)
    eng = Engine(mols(r["shape"]))
    eng.setup_full(epsilon=r["epsilon"], vec_nonz=r["vec_nonz"], max_dets=r["max_dets"], target_norm=r["target_norm"], seed=r["seed"], spawn_cap=300000)
    for row in g["rows"]:
        lg = eng.iterate_full(1)[0]
        for f in ("nkept", "n_nonz", "curr_size", "num_success"):
            assert int(lg[f]) == row[f], (row["it"], f, int(lg[f]), row[f])
        assert float(lg["norm"]) == row["norm"] and float(lg["shift"]) == row["shift"] and float(lg["denom"]) == row["denom"], row["it"]
        assert abs(lg["numer"] / lg["denom"] - row["numer"] / row["denom"]) < ENERGY_TOL
        if row["it"] % 5 == 4:
            d, v = eng.vector()
            assert golden_io.vec_hash(d, v) == row["hash"], row["it"]
    eng.close()


def test_vector_add_and_annihilation_semantics(Engine, oracle, mols):
    """DistVec add / perform_add rules (reference tests/test_vector.cpp:192-224, vec_utils.hpp:606-641):
    initiator spawns create determinants, non-initiator spawns only reach occupied ones, opposite signs cancel,
    freed positions are reused last-in-first-out."""
    mol = mols("Ne")
    par = dict(epsilon=0.01, vec_nonz=500, mat_nonz=500, max_dets=5000, seed=4)
    eng = Engine(mol)
    eng.setup(**par)
    orc = oracle.OracleFrisys(mol, **par)
    rng = np.random.RandomState(9)
    pool = rand_dets(rng, mol.n_orb, mol.n_elec, 300)
    for rnd in range(6):
        idx = rng.randint(0, pool.size, 400)
        vals = np.round(rng.standard_normal(400), 1)          # many exact cancellations and zeros
        ini = (rng.random_sample(400) < 0.5).astype(np.uint8)
        eng.vec_add(pool[idx], vals, ini)
        orc.vec_add(pool[idx], vals, ini)
        gd, gv = eng.vector()
        cd, cv = orc.vector()
        assert eng.vec_info() == orc.vec_info()
        nz = cv != 0
        assert np.array_equal(gd[nz], cd[nz]) and np.array_equal(gv, cv)
        # compress so that positions are freed and later reused
        a = eng.compress_vec(60, 0.25 + 0.1 * rnd)
        b = orc.compress_vec(60, 0.25 + 0.1 * rnd)
        assert a[0] == b[0]
        assert eng.vec_info() == orc.vec_info()
    eng.close()


def test_compression_is_identity_when_budget_exceeds_nnz(Engine, mols):
    """reference tests/test_compression.cpp:62-117."""
    mol = mols("Ne")
    eng = Engine(mol)
    eng.setup(epsilon=0.01, vec_nonz=100, mat_nonz=100, max_dets=4000, seed=3)
    hd, _ = eng.htrial()
    vals = np.random.RandomState(0).standard_normal(hd.size)
    eng.vec_load(hd, vals)
    nk, gn = eng.compress_vec(hd.size + 10, 0.3)
    d2, v2 = eng.vector()
    assert np.array_equal(v2, vals) and gn == np.cumsum(np.abs(vals))[-1]
    eng.close()


def _distinct_dets(n):
    """n distinct determinant labels for vector-level operator tests (no Hamiltonian involved)."""
    return (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(2654435761)) | np.uint64(1 << 40)


def test_pivotal_compression_known_answers(Engine, mols):
    """fries_compress_vec_piv against what the reference's piv_comp_parallel returned (tests/golden/piv_comp.txt):
    values, deleted positions and the generator's position, bit for bit."""
    mol = mols("N2")
    for cs in golden_io.read_piv_cases():
        eng = Engine(mol)
        eng.setup(epsilon=0.01, vec_nonz=100, mat_nonz=100, max_dets=4000, seed=3)
        inp = np.array(cs["inp"])
        eng.vec_load(_distinct_dets(inp.size), inp)
        eng.restart(cs["seed"])
        eng.compress_vec_piv(cs["compress_size"])
        _, v = eng.vector()
        assert v.tobytes() == np.array(cs["out"]).tobytes(), cs["len"]
        assert eng.vec_info()[1] == inp.size - int(np.sum(cs["flag"]))
        assert eng.next_draw() == cs["next"]
        eng.close()


@pytest.mark.parametrize("chain", [False, True])
@pytest.mark.parametrize("n,budget,style,seed", [(5000, 1200, 0, 1), (200000, 50000, 1, 2), (200000, 150000, 0, 3), (1000000, 400000, 2, 4),
                                                 (1000000, 3000, 0, 5), (300000, 400000, 1, 6), (64, 10, 0, 7), (65, 64, 1, 8)])
def test_pivotal_compression_matches_oracle(Engine, oracle, mols, n, budget, style, seed, chain):
    """the device operator against the CPU restatement (pinned to the reference) at sizes up to 1e6 elements, through the
    parallel certified cut-point search and through the sequential one; also the operator's invariants: at most `budget`
    non-zeros, sampled magnitudes all equal, preserved elements untouched."""
    mol = mols("N2")
    rng = np.random.RandomState(seed)
    u = rng.random_sample(n)
    mag = np.exp(6 * u) if style == 0 else (u if style == 1 else np.exp(14 * u))
    vals = np.where(rng.random_sample(n) < 0.85, mag * np.where(rng.random_sample(n) < 0.5, 1.0, -1.0), 0.0)
    eng = Engine(mol)
    eng.setup(epsilon=0.01, vec_nonz=100, mat_nonz=100, max_dets=n + 1000, seed=3)
    eng.vec_load(_distinct_dets(n), vals)
    eng.restart(1000 + seed)
    if chain:
        os.environ["FRIES_PIV_CHAIN"] = "1"
    try:
        nk, gn = eng.compress_vec_piv(budget)
    finally:
        os.environ.pop("FRIES_PIV_CHAIN", None)
    cert, fall = eng.piv_stats()
    assert (cert == 0 and fall <= 1) if chain else (cert + fall <= 1)
    if not chain and n >= 1000 and cert + fall == 1:      # (no sampling at all when the budget covers every non-zero)
        assert cert == 1, ("the parallel cut-point search should settle a generic vector", eng.last_piv_reason)
    _, v = eng.vector()
    ov, ofl, onext = oracle.piv_comp(vals, budget, 1000 + seed)
    assert np.array_equal(v, ov), int(np.sum(v != ov))
    assert eng.next_draw() == onext
    assert eng.vec_info()[1] == n - int(ofl.sum())
    nz = v != 0
    assert int(nz.sum()) <= budget
    changed = nz & (v != vals)
    if changed.any():
        assert np.unique(np.abs(v[changed])).size == 1
    eng.close()


def test_pivotal_compression_falls_back_when_a_cut_point_sits_on_a_border(Engine, oracle, mols):
    """equal magnitudes make every fourth running sum land exactly on a sampling-unit border: the parallel search cannot
    certify that against the rounding of the reference's running sum, reports it, and the sequential search settles it."""
    mol = mols("N2")
    n = 6000
    rng = np.random.RandomState(11)
    vals = np.where(np.arange(n) < 4096, np.where(rng.random_sample(n) < 0.5, 1.0, -1.0), 0.0)
    eng = Engine(mol)
    eng.setup(epsilon=0.01, vec_nonz=100, mat_nonz=100, max_dets=n + 1000, seed=3)
    eng.vec_load(_distinct_dets(n), vals)
    eng.restart(42)
    eng.compress_vec_piv(1024)
    cert, fall = eng.piv_stats()
    assert (cert, fall) == (0, 1) and eng.last_piv_reason & 1
    _, v = eng.vector()
    ov, ofl, onext = oracle.piv_comp(vals, 1024, 42)
    assert np.array_equal(v, ov) and eng.next_draw() == onext and int(np.count_nonzero(v)) == 1024
    eng.close()


@pytest.mark.parametrize("n,seed,up", [(3000, 2, 0), (3000, 2, 1), (3000, 18, 1), (100000, 5, 0), (100000, 5, 1), (100000, 23, 1), (70, 5, 1)])
def test_pivotal_adjust_probs_matches_oracle(Engine, oracle, mols, n, seed, up):
    """adjust_probs (the re-weighting a rank applies when its integer sample budget was rounded up or down from its
    expected share) on the device against the restatement: values, pinned elements, budget and returned norm."""
    mol = mols("N2")
    rng = np.random.RandomState(seed)
    vals = rng.random_sample(n) * np.where(rng.random_sample(n) < 0.5, 1.0, -1.0) * (rng.random_sample(n) < 0.9)
    norm = float(np.abs(vals).sum())
    share = 0.2 + 0.6 * rng.random_sample()
    n_tot = int(rng.randint(n, 2 * n))
    tot_norm = norm / share
    exp_loc = n_tot * norm / tot_norm
    n_loc = int(exp_loc) + up
    eng = Engine(mol)
    eng.setup(epsilon=0.01, vec_nonz=100, mat_nonz=100, max_dets=n + 1000, seed=3)
    eng.vec_load(_distinct_dets(n), vals)
    nl, nn, fl = eng.test_piv_adjust(n_loc, exp_loc, n_tot, tot_norm)
    _, v = eng.vector()
    ov, onl, onn, ofl = oracle.adjust_probs(vals, n_loc, exp_loc, n_tot, tot_norm)
    assert not np.array_equal(ov, vals)            # the case does exercise the re-weighting
    assert np.array_equal(v, ov) and nl == onl and nn == onn and np.array_equal(fl[:n], ofl)
    eng.close()


def test_full_size_invariants_m1e6(Engine, mols):
    """BASELINE.json's full size (m = 1e6) through size-independent properties: the compression conserves the
    one-norm, keeps at most vec_nonz elements, every stored determinant has n_elec electrons and is unique, and a
    second engine fed the same state and seed reproduces the run bit for bit."""
    mol = mols("N2")
    m = 1_000_000
    par = dict(epsilon=0.01, vec_nonz=m, mat_nonz=m, max_dets=3 * m, target_norm=0.0, initiator=0.0, seed=13, distribution="HB_unnorm")
    eng = Engine(mol)
    eng.setup(**par)
    for _ in range(40):
        lg = eng.iterate(1)[0]
        if lg["n_nonz"] >= m:
            break
    logs = eng.iterate(4)
    assert int(logs["err"].max()) == 0
    d, v = eng.vector()
    nz = v != 0
    assert nz.sum() <= m + 1 and nz.sum() == logs["n_nonz"][-1]
    dn = d[nz]
    assert np.unique(dn).size == dn.size
    pop = np.zeros(dn.size, dtype=np.int64)
    x = dn.copy()
    for _ in range(64):
        pop += (x & np.uint64(1)).astype(np.int64)
        x >>= np.uint64(1)
    assert np.all(pop == mol.n_elec)
    # one-norm conservation of find_preserve + sys_comp
    before = np.abs(v).sum()
    nk, gn = eng.compress_vec(7 * m // 10, 0.4321)
    d2, v2 = eng.vector()
    assert abs(np.abs(v2).sum() - before) < 1e-9 * before and abs(gn - before) < 1e-9 * before
    assert (v2 != 0).sum() <= 7 * m // 10 + 1
    # determinism: replay the same state and seed in a fresh engine
    eng2 = Engine(mol)
    eng2.setup(**par)
    eng2.vec_load(d2, v2)
    eng.vec_load(d2, v2)
    eng.restart(99)
    eng2.restart(99)
    la, lb = eng.iterate(3), eng2.iterate(3)
    for f in ("num_success", "n_nonz", "curr_size", "nkept", "norm", "numer", "denom"):
        assert np.array_equal(la[f], lb[f]), f
    assert np.array_equal(eng.vector()[1], eng2.vector()[1])
    eng.close()
    eng2.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(k for k, v in golden_io.manifest()["hh_runs"].items() if v["n_ranks"] == 1)
                         + sorted(k for k, v in golden_io.manifest()["hhfull_runs"].items() if v["n_ranks"] == 1))
def test_hubbard_holstein_matches_reference(name):
    """frisys_hh (hh_*) and frifull_hh (hhfull_*) on the device against the reference's own loops (tests/golden/hh*.traj): counts,
    norms, shifts and the stored shard bit for bit; the projected-energy numerator (a block-parallel sum of signed terms) to 1e-10."""
    from fries_amd.engine import FriEngine
    full = name in golden_io.manifest()["hhfull_runs"]
    r = golden_io.manifest()["hhfull_runs" if full else "hh_runs"][name]
    g = golden_io.read_traj(name)
    eng = FriEngine(None)
    eng.setup_hh(n_elec=r["n_elec"], n_sites=r["n_sites"], eps=r["eps"], U=r["U"], omega=r["omega"], g=r["g"], gs_energy=r["gs_energy"],
                 vec_nonz=r["vec_nonz"], max_dets=r["max_dets"], target_norm=r["target_norm"], initiator=r["initiator"], seed=r["seed"], full=full)
    logs = eng.iterate_hh(r["n_iter"])
    for i, row in enumerate(g["rows"]):
        lg = logs[i]
        assert int(lg["err"]) == 0
        for f in ("nkept", "n_nonz", "curr_size", "num_success"):
            assert int(lg[f]) == row[f], (name, i, f, int(lg[f]), row[f])
        for f in ("norm", "shift", "denom"):
            assert float(lg[f]) == row[f], (name, i, f)
        assert abs(float(lg["numer"]) - row["numer"]) <= 1e-10 * max(1.0, abs(row["numer"])), (name, i)
    d, v = eng.vector()
    assert golden_io.vec_hash(d, v) == g["rows"][-1]["hash"]
    eng.close()


@pytest.mark.gpu
def test_cli_driver_reproduces_reference_outputs(tmp_path):
    """fries_amd/frisys_mol_hip (C++ host driver over the C ABI) with the reference's command line: projnum / projden /
    nkept / S / norm files against the golden trajectory, then the binary checkpoint through --load_dir."""
    import subprocess
    from fries_amd import build
    name = "ne_m2000_unnorm"
    r = golden_io.manifest()["runs"][name]
    g = golden_io.read_traj(name)
    mol = fcidump.synthetic(r["shape"])
    fc = str(tmp_path / "mol.FCIDUMP")
    fcidump.write_fcidump(fc, mol)
    out1 = str(tmp_path / "run1") + "/"
    os.makedirs(out1)
    exe = build.DRIVER
    assert os.path.exists(exe), "frisys_mol_hip has not been built"
    n_it = 40
    cmd = [exe, "--fcidump_path", fc, "--point_group", mol.point_group, "--distribution", r["distribution"], "--vec_nonz", str(r["vec_nonz"]),
           "--mat_nonz", str(r["mat_nonz"]), "--max_dets", str(r["max_dets"]), "--target", repr(r["target_norm"]), "--initiator", repr(r["initiator"]),
           "--epsilon", repr(r["epsilon"]), "--max_iter", str(n_it), "--result_dir", out1, "--seed", str(r["seed"])]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "Exception" not in res.stderr, res.stderr[-2000:]
    num = np.loadtxt(out1 + "projnum.txt"); den = np.loadtxt(out1 + "projden.txt"); nk = np.loadtxt(out1 + "nkept.txt")
    sh = np.loadtxt(out1 + "S.txt"); nm = np.loadtxt(out1 + "norm.txt")
    assert num.size == n_it and sh.size == n_it // 10
    for i in range(n_it):
        row = g["rows"][i]
        assert abs(num[i] / den[i] - row["numer"] / row["denom"]) < 1e-10
        assert int(nk[i]) == row["nkept"]
    for k in range(n_it // 10):
        row = g["rows"][10 * k + 9]
        assert sh[k] == row["shift"] and nm[k] == row["norm"]        # written with 17 significant digits
    # checkpoint: dets0.dat holds ceil(2 n_orb / 8) bytes per stored index, vals0.dat two value columns (DistVec::save)
    nb = (2 * mol.n_orb + 7) // 8
    raw = np.fromfile(out1 + "dets0.dat", dtype=np.uint8)
    n_saved = raw.size // nb
    assert n_saved == g["rows"][n_it - 1]["curr_size"]
    vals = np.fromfile(out1 + "vals0.dat", dtype=np.float64)
    assert vals.size == 2 * n_saved and np.all(vals[n_saved:] == 0)
    dets = np.zeros(n_saved, dtype=np.uint64)
    for b in range(nb):
        dets |= raw.reshape(n_saved, nb)[:, b].astype(np.uint64) << np.uint64(8 * b)
    assert golden_io.vec_hash(dets, vals[:n_saved]) == g["rows"][n_it - 1]["hash"]
    assert np.fromfile(out1 + "hash.dat", dtype=np.uint32).size == 2 * mol.n_orb
    # the same files as the reference's own DistVec::save wrote after the same 40 iterations, byte for byte
    import hashlib
    ck = golden_io.manifest()["checkpoints"][name]
    assert ck["after_iterations"] == n_it
    for fn in ("dets0.dat", "vals0.dat"):
        blob = open(out1 + fn, "rb").read()
        assert len(blob) == ck[fn.replace(".", "_") + "_bytes"], fn
        assert hashlib.sha256(blob).hexdigest() == ck[fn.replace(".", "_") + "_sha256"], fn
    assert open(out1 + "dense.txt").read() == ck["dense_txt"]
    # --load_dir: the stored non-zeros come back in file order (DistVec::load) and the run continues
    out2 = str(tmp_path / "run2") + "/"
    os.makedirs(out2)
    cmd2 = cmd[:]
    cmd2[cmd2.index("--result_dir") + 1] = out2
    cmd2[cmd2.index("--max_iter") + 1] = "5"
    cmd2 += ["--load_dir", out1]
    res2 = subprocess.run(cmd2, capture_output=True, text=True, timeout=300)
    assert res2.returncode == 0 and "Exception" not in res2.stderr, res2.stderr[-2000:]
    assert np.loadtxt(out2 + "projnum.txt").size == 5
    keep = vals[:n_saved] != 0
    assert np.fromfile(out2 + "dets0.dat", dtype=np.uint8).size // nb > 0 and int(keep.sum()) > 0
    # --precision 6: the text files in the reference's own format (its drivers stream doubles with the default precision, "%g" with six digits)
    out3 = str(tmp_path / "run3") + "/"
    os.makedirs(out3)
    cmd3 = cmd[:]
    cmd3[cmd3.index("--result_dir") + 1] = out3
    cmd3[cmd3.index("--max_iter") + 1] = "20"
    res3 = subprocess.run(cmd3 + ["--precision", "6"], capture_output=True, text=True, timeout=300)
    assert res3.returncode == 0 and "Exception" not in res3.stderr, res3.stderr[-2000:]
    assert open(out3 + "projnum.txt").read().split() == ["%g" % x for x in num[:20]]
    assert open(out3 + "projden.txt").read().split() == ["%g" % x for x in den[:20]]
    assert open(out3 + "S.txt").read().split() == ["%g" % x for x in sh[:2]] and open(out3 + "norm.txt").read().split() == ["%g" % x for x in nm[:2]]


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(golden_io.manifest().get("reload_runs", {})))
def test_cli_load_dir_matches_reference_restart(name, tmp_path):
    """--load_dir, pinned (tests/golden/*_reload.traj from `ref_harness reload`): the reference runs n1 iterations, DistVec::save,
    then -- as frisys_mol restarted with --load_dir and the same seed flag does -- DistVec::load into a fresh vector (entries
    with |v| <= 1e-9 dropped, the rest compacted and re-hashed), the shift of S.txt, last_one_norm = 0 (the shift stays put
    until a shift iteration sees the norm above the target), the generator continued after the vec scrambler's draws, and n2
    more iterations crossing shift updates.  frisys_mol_hip does the same through its own checkpoint files."""
    import subprocess
    from fries_amd import build
    r = golden_io.manifest()["reload_runs"][name]
    rows, loaded = [], None
    with open(os.path.join(golden_io.GOLD, name + ".traj")) as f:
        for ln in f:
            t = ln.split()
            if ln.startswith("LOADED"):
                loaded = dict(n=int(t[1]), digest=int(t[3], 16), saved=int(t[5]), shift=float.fromhex(t[7]))
            elif t and not ln.startswith("#"):
                rows.append(dict(numer=float.fromhex(t[1]), denom=float.fromhex(t[2]), norm=float.fromhex(t[3]), shift=float.fromhex(t[4]), nkept=int(t[5]),
                                 n_nonz=int(t[6]), curr_size=int(t[7]), hash=int(t[9], 16)))
    mol = fcidump.synthetic(r["shape"])
    fc = str(tmp_path / "mol.FCIDUMP")
    fcidump.write_fcidump(fc, mol)
    out1, out2 = str(tmp_path / "run1") + "/", str(tmp_path / "run2") + "/"
    os.makedirs(out1); os.makedirs(out2)
    base = [build.DRIVER, "--fcidump_path", fc, "--point_group", mol.point_group, "--distribution", r["distribution"], "--vec_nonz", str(r["vec_nonz"]),
            "--mat_nonz", str(r["mat_nonz"]), "--max_dets", str(r["max_dets"]), "--target", repr(r["target_norm"]), "--initiator", repr(r["initiator"]),
            "--epsilon", repr(r["epsilon"]), "--seed", str(r["seed"])]
    # a dense (semi-stochastic) space: the first run is given --det_space, the restarted one takes it from the checkpoint's dense.txt
    # like the reference (frisys_mol.cpp:234, :258; DistVec::load keeps the first n_dense entries whatever their values)
    first = ["--det_space", os.path.join(golden_io.GOLD, r["det_space"])] if r.get("det_space") else []
    res = subprocess.run(base + first + ["--max_iter", str(r["n1"]), "--result_dir", out1], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "Exception" not in res.stderr, res.stderr[-2000:]
    if first:
        assert open(out1 + "dense.txt").read().strip() not in ("", "0")
    nb = (2 * mol.n_orb + 7) // 8
    assert os.path.getsize(out1 + "dets0.dat") // nb == loaded["saved"]
    assert np.loadtxt(out1 + "S.txt").reshape(-1)[-1] == loaded["shift"]
    res = subprocess.run(base + ["--max_iter", str(r["n2"]), "--result_dir", out2, "--load_dir", out1], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "Exception" not in res.stderr, res.stderr[-2000:]
    num = np.loadtxt(out2 + "projnum.txt"); den = np.loadtxt(out2 + "projden.txt"); nk = np.loadtxt(out2 + "nkept.txt")
    sh = np.loadtxt(out2 + "S.txt").reshape(-1); nm = np.loadtxt(out2 + "norm.txt").reshape(-1)
    assert num.size == r["n2"]
    for i, row in enumerate(rows):
        assert abs(num[i] / den[i] - row["numer"] / row["denom"]) < ENERGY_TOL, i
        assert int(nk[i]) == row["nkept"], i
    for k in range(r["n2"] // 10):
        assert sh[k] == rows[10 * k + 9]["shift"] and nm[k] == rows[10 * k + 9]["norm"], k
    assert len(set(sh.tolist())) > 1 and sh[0] == loaded["shift"]       # frozen at first, then it moves: the update was crossed
    raw = np.fromfile(out2 + "dets0.dat", dtype=np.uint8)
    n_saved = raw.size // nb
    assert n_saved == rows[-1]["curr_size"]
    vals = np.fromfile(out2 + "vals0.dat", dtype=np.float64)
    dets = np.zeros(n_saved, dtype=np.uint64)
    for b in range(nb):
        dets |= raw.reshape(n_saved, nb)[:, b].astype(np.uint64) << np.uint64(8 * b)
    assert golden_io.vec_hash(dets, vals[:n_saved]) == rows[-1]["hash"]


@pytest.mark.gpu
@pytest.mark.parametrize("shape,eps,target,ini,seed,n_it,dist", [("Ne", 0.004, 20000, 3, 5, 300, "NU"), ("N2", 0.006, 50000, 2, 9, 350, "NU"), ("H2O", 0.004, 30000, 0, 11, 250, "NU"),
                                                                 ("N2", 0.006, 50000, 2, 9, 300, "HB"), ("H2O", 0.004, 30000, 0, 11, 250, "HB")])
def test_fciqmc_matches_oracle_counter_stream(oracle, mols, shape, eps, target, ini, seed, n_it, dist):
    """fciqmc_mol (near-uniform and heat-bath generators) on the device against the CPU oracle, both on the counter-based uniform stream: walker
    numbers, positions, spawn counts and shifts identical at every iteration.  (The oracle on the reference's own mt19937 stream
    is pinned against the reference loop in the CPU suite.)"""
    from fries_amd.engine import FriEngine
    mol = mols(shape)
    orc = oracle.OracleFciqmc(mol, epsilon=eps, target_walkers=target, max_dets=200000, initiator=ini, seed=seed, counter_rng=True, distribution=dist)
    eng = FriEngine(mol)
    eng.setup_fciqmc(epsilon=eps, target_walkers=target, max_dets=200000, initiator=ini, seed=seed, distribution=dist)
    assert eng.p_doub == orc.p_doub
    lo = orc.iterate(n_it)
    lg = eng.iterate_fciqmc(n_it)
    assert int(lg["err"].max()) == 0
    for f in ("n_nonz", "n_ini", "curr_size", "n_spawn"):
        assert np.array_equal(lg[f].astype(np.int64), lo[f].astype(np.int64)), (shape, f, np.nonzero(lg[f].astype(np.int64) != lo[f].astype(np.int64))[0][:5])
    assert np.array_equal(lg["shift"], lo["shift"]) and np.array_equal(lg["norm"], lo["norm"]) and np.array_equal(lg["denom"], lo["denom"])
    assert np.all(np.abs(lg["numer"] - lo["numer"]) <= 1e-10 * np.maximum(1.0, np.abs(lo["numer"])))
    gd, gv = eng.vector()
    od, ov = orc.vector()
    assert gd.size == od.size and np.array_equal(gv, ov)
    nz = ov != 0
    assert np.array_equal(gd[nz], od[nz])
    assert int(lo["n_nonz"][-1]) > 30 and int(lo["n_spawn"].sum()) > 200         # walkers did spread and spawn
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(golden_io.manifest()["fciqmc_fp_runs"]))
def test_fciqmc_fp_matches_oracle_counter_stream(oracle, name, tmp_path):
    """fciqmc_fp_mol (real-valued walkers) on the device against the CPU restatement, both on the counter-based uniform stream, in the
    configurations whose mt19937 runs are pinned against the reference loop (tests/golden/fciqmc_fp_*.traj, CPU suite): attempts per
    walker, real and rounded spawns, the +-1 rounding of small values and the deletes -- every stored double bit for bit."""
    from fries_amd.engine import FriEngine
    r = golden_io.manifest()["fciqmc_fp_runs"][name]
    mol = fcidump.synthetic(r["shape"])
    par = dict(epsilon=r["epsilon"], target_walkers=r["target_walkers"], max_dets=r["max_dets"], initiator=r["initiator"], seed=r["seed"], distribution=r["distribution"], fp=True)
    n_it = r["n_iter"]
    if "trial" in r:        # --trial_vec / --ini_vec (fciqmc_fp_mol.cpp:157-185, 233-246)
        par["trial"] = golden_io.read_text_vector(r["trial"])
    if "ini" in r:
        par["ini"] = golden_io.read_text_vector(r["ini"])
    orc = oracle.OracleFciqmc(mol, counter_rng=True, **par)
    eng = FriEngine(mol)
    eng.setup_fciqmc(**par)
    lo = orc.iterate(n_it)
    lg = eng.iterate_fciqmc(n_it)
    assert int(lg["err"].max()) == 0
    for f in ("n_nonz", "n_ini", "curr_size", "n_spawn"):
        assert np.array_equal(lg[f].astype(np.int64), lo[f].astype(np.int64)), (name, f, np.nonzero(lg[f].astype(np.int64) != lo[f].astype(np.int64))[0][:5])
    assert np.array_equal(lg["shift"], lo["shift"]) and np.array_equal(lg["norm"], lo["norm"]) and np.array_equal(lg["denom"], lo["denom"])
    assert np.all(np.abs(lg["numer"] - lo["numer"]) <= 1e-10 * np.maximum(1.0, np.abs(lo["numer"])))
    gd, gv = eng.vector()
    od, ov = orc.vector()
    assert gd.size == od.size and np.array_equal(gv, ov)
    nz = ov != 0
    assert np.array_equal(gd[nz], od[nz])
    assert int(lo["n_nonz"][-1]) > 30 and np.any(ov != np.round(ov))          # the walkers spread, and some are not integers
    eng.close()
    if name not in ("fciqmc_fp_ne", "fciqmc_fp_n2_trial_ini"):
        return
    import subprocess
    from fries_amd import build
    fc = str(tmp_path / "mol.FCIDUMP")
    fcidump.write_fcidump(fc, mol)
    out = str(tmp_path / "out") + "/"
    os.makedirs(out)
    vec_flags = (["--trial_vec", os.path.join(golden_io.GOLD, r["trial"])] if "trial" in r else []) + (["--ini_vec", os.path.join(golden_io.GOLD, r["ini"])] if "ini" in r else [])
    res = subprocess.run([build.DRIVERS["fciqmc_mol_hip"], "--fcidump_path", fc, "--point_group", mol.point_group, "--distribution", r["distribution"], "--target", str(r["target_walkers"]),
                          "--max_dets", str(r["max_dets"]), "--epsilon", repr(r["epsilon"]), "--initiator", str(r["initiator"]), "--max_iter", str(n_it), "--result_dir", out,
                          "--seed", str(r["seed"]), "--fp", "1"] + vec_flags, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "Exception" not in res.stderr, res.stderr[-2000:]
    num = np.loadtxt(out + "projnum.txt"); den = np.loadtxt(out + "projden.txt"); nini = np.loadtxt(out + "nini.txt")
    assert np.array_equal(den, lo["denom"]) and np.all(np.abs(num - lo["numer"]) <= 1e-10 * np.maximum(1.0, np.abs(lo["numer"])))
    assert np.array_equal(nini.astype(np.int64), lo["n_ini"].astype(np.int64))


@pytest.mark.gpu
@pytest.mark.parametrize("name,n_it", [("multi_ne_m1000", 60), ("multi_n2_m5000_ini0", 40), ("multi_n2_ini", 40)])
def test_frimulti_matches_oracle_counter_stream(oracle, name, n_it):
    """frimulti_mol (multinomial matrix compression) on the device against the CPU restatement, both on the counter-based uniform stream,
    in the configurations whose mt19937 runs are pinned against the reference loop (tests/golden/multi_*.traj, CPU suite): samples per
    column from the comb, every spawn weight, the compressed vector -- norms, shifts, counts and the stored values bit for bit."""
    from fries_amd.engine import FriEngine
    r = golden_io.manifest()["multi_runs"][name]
    mol = fcidump.synthetic(r["shape"])
    par = dict(epsilon=r["epsilon"], vec_nonz=r["vec_nonz"], mat_nonz=r["mat_nonz"], max_dets=r["max_dets"], initiator=r["initiator"], target_norm=r["target_norm"], seed=r["seed"])
    if "ini" in r:          # --ini_vec (frimulti_mol.cpp:205-215)
        par["ini"] = golden_io.read_text_vector(r["ini"])
    orc = oracle.OracleMulti(mol, counter_rng=True, **par)
    eng = FriEngine(mol)
    eng.setup_multi(**par)
    assert eng.p_doub == orc.p_doub
    lo = orc.iterate(n_it)
    lg = eng.iterate_multi(n_it)
    assert int(lg["err"].max()) == 0
    for f in ("n_nonz", "n_ini", "curr_size", "n_spawn"):
        assert np.array_equal(lg[f].astype(np.int64), lo[f].astype(np.int64)), (name, f, np.nonzero(lg[f].astype(np.int64) != lo[f].astype(np.int64))[0][:5])
    assert np.array_equal(lg["norm"], lo["norm"]) and np.array_equal(lg["shift"], lo["shift"]) and np.array_equal(lg["denom"], lo["denom"])
    assert np.all(np.abs(lg["numer"] - lo["numer"]) <= 1e-10 * np.maximum(1.0, np.abs(lo["numer"])))
    gd, gv = eng.vector()
    od, ov = orc.vector()
    assert gd.size == od.size and np.array_equal(gv, ov)
    nz = ov != 0
    assert np.array_equal(gd[nz], od[nz])
    assert int(lo["n_nonz"][-1]) == r["vec_nonz"] and int(lo["n_spawn"].sum()) > 10 * r["vec_nonz"]
    eng.close()
    if name != "multi_ne_m1000":
        return
    # the command-line driver writes the same numbers
    import subprocess, tempfile
    from fries_amd import build
    with tempfile.TemporaryDirectory() as tmp:
        fc = os.path.join(tmp, "mol.FCIDUMP")
        fcidump.write_fcidump(fc, mol)
        out = os.path.join(tmp, "out") + "/"
        os.makedirs(out)
        res = subprocess.run([build.DRIVERS["frimulti_mol_hip"], "--fcidump_path", fc, "--point_group", mol.point_group, "--distribution", "HB", "--vec_nonz", str(r["vec_nonz"]),
                              "--mat_nonz", str(r["mat_nonz"]), "--max_dets", str(r["max_dets"]), "--epsilon", repr(r["epsilon"]), "--target", repr(r["target_norm"]),
                              "--initiator", repr(r["initiator"]), "--max_iter", str(n_it), "--result_dir", out, "--seed", str(r["seed"])], capture_output=True, text=True, timeout=300)
        assert res.returncode == 0 and "Exception" not in res.stderr, res.stderr[-2000:]
        num = np.loadtxt(out + "projnum.txt"); den = np.loadtxt(out + "projden.txt"); nm = np.loadtxt(out + "norm.txt"); nini = np.loadtxt(out + "nini.txt")
        assert np.array_equal(den, lo["denom"]) and np.all(np.abs(num - lo["numer"]) <= 1e-10 * np.maximum(1.0, np.abs(lo["numer"])))
        assert np.array_equal(nm, lo["norm"][9::10]) and np.array_equal(nini.astype(np.int64), lo["n_ini"].astype(np.int64))


@pytest.mark.gpu
def test_frimulti_refuses_a_trial_vector_like_the_reference(tmp_path):
    """frimulti_mol --trial_vec: the reference throws "Insufficient memory allocated in adder" for every trial file on one rank (frimulti_mol.cpp:149-157; the
    run that recorded it: oracle/gen_golden.py, manifest key multi_trial_one_rank_error).  The engine and the command-line driver end the same way."""
    import subprocess
    from fries_amd import build
    from fries_amd.engine import FriEngine
    r = golden_io.manifest()["multi_trial_one_rank_error"]
    mol = fcidump.synthetic(r["shape"])
    eng = FriEngine(mol)
    with pytest.raises(RuntimeError, match=r["error"]):
        eng.setup_multi(epsilon=0.01, vec_nonz=5000, mat_nonz=20000, max_dets=200000, initiator=1.0, target_norm=2500.0, seed=3, trial=golden_io.read_text_vector(r["trial"]))
    eng.close()
    fc = str(tmp_path / "mol.FCIDUMP")
    fcidump.write_fcidump(fc, mol)
    res = subprocess.run([build.DRIVERS["frimulti_mol_hip"], "--fcidump_path", fc, "--point_group", mol.point_group, "--distribution", "HB", "--vec_nonz", "5000", "--mat_nonz", "20000",
                          "--max_dets", "200000", "--epsilon", "0.01", "--max_iter", "2", "--result_dir", str(tmp_path) + "/", "--seed", "3",
                          "--trial_vec", os.path.join(golden_io.GOLD, r["trial"])], capture_output=True, text=True, timeout=120)
    assert r["error"] in res.stderr


@pytest.mark.gpu
def test_cli_drivers_hh_and_fciqmc(oracle, mols, tmp_path):
    """frisys_hh_hip against the reference's golden trajectory (its parameter-file format included) and fciqmc_mol_hip against the
    CPU restatement on the counter stream: the files the drivers write."""
    import subprocess
    from fries_amd import build
    # ---- frisys_hh_hip
    name = "hh_l6_m2000"
    r = golden_io.manifest()["hh_runs"][name]
    g = golden_io.read_traj(name)
    pf = tmp_path / "hh_params.txt"
    pf.write_text("n_elec\n%d\nlat_len\n%d\nn_dim\n1\neps\n%r\nU\n%r\nomega\n%r\ng\n%r\ngs_energy\n%r\n" % (r["n_elec"], r["n_sites"], r["eps"], r["U"], r["omega"], r["g"], r["gs_energy"]))
    out = str(tmp_path / "hh") + "/"
    os.makedirs(out)
    n_it = 30
    res = subprocess.run([build.DRIVERS["frisys_hh_hip"], "--params_path", str(pf), "--vec_nonz", str(r["vec_nonz"]), "--max_dets", str(r["max_dets"]), "--target",
                          repr(r["target_norm"]), "--initiator", repr(r["initiator"]), "--max_iter", str(n_it), "--result_dir", out, "--seed", str(r["seed"])],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "Exception" not in res.stderr, res.stderr[-2000:]
    num = np.loadtxt(out + "projnum.txt"); den = np.loadtxt(out + "projden.txt"); sh = np.loadtxt(out + "S.txt"); nm = np.loadtxt(out + "norm.txt")
    for i in range(n_it):
        row = g["rows"][i]
        assert den[i] == row["denom"] and abs(num[i] - row["numer"]) <= 1e-10 * max(1.0, abs(row["numer"]))
    for k in range(n_it // 10):
        assert sh[k] == g["rows"][10 * k + 9]["shift"] and nm[k] == g["rows"][10 * k + 9]["norm"]
    # ---- the same driver as frifull_hh (--full 1)
    name = "hhfull_l6_m300"
    r = golden_io.manifest()["hhfull_runs"][name]
    g = golden_io.read_traj(name)
    pf.write_text("n_elec\n%d\nlat_len\n%d\nn_dim\n1\neps\n%r\nU\n%r\nomega\n%r\ng\n%r\ngs_energy\n%r\n" % (r["n_elec"], r["n_sites"], r["eps"], r["U"], r["omega"], r["g"], r["gs_energy"]))
    outf = str(tmp_path / "hhfull") + "/"
    os.makedirs(outf)
    res = subprocess.run([build.DRIVERS["frisys_hh_hip"], "--params_path", str(pf), "--vec_nonz", str(r["vec_nonz"]), "--max_dets", str(r["max_dets"]), "--target",
                          repr(r["target_norm"]), "--initiator", repr(r["initiator"]), "--max_iter", str(n_it), "--result_dir", outf, "--seed", str(r["seed"]), "--full", "1"],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "Exception" not in res.stderr, res.stderr[-2000:]
    num = np.loadtxt(outf + "projnum.txt"); den = np.loadtxt(outf + "projden.txt"); nm = np.loadtxt(outf + "norm.txt")
    for i in range(n_it):
        row = g["rows"][i]
        assert den[i] == row["denom"] and abs(num[i] - row["numer"]) <= 1e-10 * max(1.0, abs(row["numer"]))
    for k in range(n_it // 10):
        assert nm[k] == g["rows"][10 * k + 9]["norm"]
    # ---- fciqmc_mol_hip
    mol = mols("Ne")
    fc = str(tmp_path / "ne.FCIDUMP")
    fcidump.write_fcidump(fc, mol)
    out2 = str(tmp_path / "fq") + "/"
    os.makedirs(out2)
    n_it = 60
    res = subprocess.run([build.DRIVERS["fciqmc_mol_hip"], "--fcidump_path", fc, "--point_group", mol.point_group, "--distribution", "NU", "--target", "20000",
                          "--max_dets", "100000", "--epsilon", "0.004", "--initiator", "3", "--max_iter", str(n_it), "--result_dir", out2, "--seed", "5"],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "Exception" not in res.stderr, res.stderr[-2000:]
    orc = oracle.OracleFciqmc(mol, epsilon=0.004, target_walkers=20000, max_dets=100000, initiator=3, seed=5, counter_rng=True)
    lo = orc.iterate(n_it)
    num = np.loadtxt(out2 + "projnum.txt"); den = np.loadtxt(out2 + "projden.txt"); nini = np.loadtxt(out2 + "nini.txt"); nnz = np.loadtxt(out2 + "nnonz.txt")
    assert np.array_equal(den, lo["denom"]) and np.array_equal(nini.astype(np.int64), lo["n_ini"].astype(np.int64))
    assert np.all(np.abs(num - lo["numer"]) <= 1e-10 * np.maximum(1.0, np.abs(lo["numer"])))
    assert np.array_equal(nnz.astype(np.int64), lo["n_nonz"][9::10].astype(np.int64))


def test_cli_driver_frifull(mols, tmp_path):
    """frifull_mol_hip: the files it writes against the reference's golden trajectory."""
    import subprocess
    from fries_amd import build
    name = "full_ne_m300"
    r = golden_io.manifest()["full_runs"][name]
    g = golden_io.read_traj(name)
    mol = mols(r["shape"])
    fc = str(tmp_path / "mol.FCIDUMP")
    fcidump.write_fcidump(fc, mol)
    out = str(tmp_path / "full") + "/"
    os.makedirs(out)
    n_it = 30
    res = subprocess.run([build.DRIVERS["frifull_mol_hip"], "--fcidump_path", fc, "--point_group", mol.point_group, "--epsilon", repr(r["epsilon"]), "--vec_nonz",
                          str(r["vec_nonz"]), "--max_dets", str(r["max_dets"]), "--target", repr(r["target_norm"]), "--max_iter", str(n_it), "--result_dir", out,
                          "--seed", str(r["seed"])], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "Exception" not in res.stderr, res.stderr[-2000:]
    num = np.loadtxt(out + "projnum.txt"); den = np.loadtxt(out + "projden.txt"); sh = np.loadtxt(out + "S.txt"); nm = np.loadtxt(out + "norm.txt"); nk = np.loadtxt(out + "nkept.txt")
    for i in range(n_it):
        row = g["rows"][i]
        assert den[i] == row["denom"] and abs(num[i] - row["numer"]) <= 1e-10 * max(1.0, abs(row["numer"])) and int(nk[i]) == row["nkept"]
    for k in range(n_it // 10):
        assert sh[k] == g["rows"][10 * k + 9]["shift"] and nm[k] == g["rows"][10 * k + 9]["norm"]
    bad = subprocess.run([build.DRIVERS["frifull_mol_hip"], "--hf_path", "x/"], capture_output=True, text=True, timeout=60)
    assert bad.returncode == 1 and "missing required option" in bad.stderr


def test_cli_driver_text_vectors_and_ham_shift(mols, tmp_path):
    """frisys_mol_hip --trial_vec / --ini_vec read the reference's text vector files; --ham_shift: the written files against the
    reference's trajectories."""
    import subprocess
    from fries_amd import build
    for name in sorted(golden_io.manifest()["extra_runs"]):
        r = golden_io.manifest()["extra_runs"][name]
        g = golden_io.read_traj(name)
        mol = mols(r["shape"])
        fc = str(tmp_path / (name + ".FCIDUMP"))
        fcidump.write_fcidump(fc, mol)
        out = str(tmp_path / name) + "/"
        os.makedirs(out)
        n_it = 20
        cmd = [build.DRIVER, "--fcidump_path", fc, "--point_group", mol.point_group, "--distribution", r["distribution"], "--vec_nonz", str(r["vec_nonz"]),
               "--mat_nonz", str(r["mat_nonz"]), "--max_dets", str(r["max_dets"]), "--target", repr(r["target_norm"]), "--initiator", repr(r["initiator"]),
               "--epsilon", repr(r["epsilon"]), "--max_iter", str(n_it), "--result_dir", out, "--seed", str(r["seed"])]
        if "trial" in r:
            cmd += ["--trial_vec", os.path.join(golden_io.GOLD, r["trial"])]
        if "ini" in r:
            cmd += ["--ini_vec", os.path.join(golden_io.GOLD, r["ini"])]
        if "ham_shift" in r:
            cmd += ["--ham_shift", repr(r["ham_shift"])]
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        assert res.returncode == 0 and "Exception" not in res.stderr, res.stderr[-2000:]
        num = np.loadtxt(out + "projnum.txt"); den = np.loadtxt(out + "projden.txt"); nk = np.loadtxt(out + "nkept.txt")
        for i in range(n_it):
            row = g["rows"][i]
            assert abs(num[i] / den[i] - row["numer"] / row["denom"]) < 1e-10 and int(nk[i]) == row["nkept"]


def test_fciqmc_trial_and_initial_vectors(oracle, mols, tmp_path):
    """fciqmc_mol --trial_vec / --ini_vec on the device (engine and command-line driver) against the restatement on the counter
    stream; the restatement's mt19937 mode reproduces the reference run with the same files (CPU suite), including the last
    trial entry counting twice in the denominators."""
    import subprocess
    from fries_amd import build
    from fries_amd.engine import FriEngine
    mol = mols("N2")
    trial = golden_io.read_text_vector("n2_trial_")
    ini = golden_io.read_text_vector("n2_fq_ini_")
    par = dict(epsilon=0.004, target_walkers=20000, max_dets=100000, initiator=2, seed=9, distribution="NU")
    orc = oracle.OracleFciqmc(mol, counter_rng=True, trial=trial, ini=ini, **par)
    n_it = 120
    lo = orc.iterate(n_it)
    eng = FriEngine(mol)
    eng.setup_fciqmc(trial=trial, ini=ini, **par)
    lg = eng.iterate_fciqmc(n_it)
    for f in ("n_nonz", "n_ini", "curr_size", "n_spawn"):
        assert np.array_equal(lg[f].astype(np.int64), lo[f].astype(np.int64)), f
    # (a 25-term projection is summed block-parallel on the device: 1e-12 relative instead of bit for bit)
    assert np.all(np.abs(lg["denom"] - lo["denom"]) <= 1e-12 * np.abs(lo["denom"])) and np.array_equal(lg["shift"], lo["shift"])
    assert np.all(np.abs(lg["numer"] - lo["numer"]) <= 1e-10 * np.maximum(1.0, np.abs(lo["numer"])))
    d, v = eng.vector(); od, ov = orc.vector()
    assert np.array_equal(v, ov) and np.array_equal(d[v != 0], od[ov != 0])
    eng.close()
    # the driver reads the same files
    fc = str(tmp_path / "n2.FCIDUMP")
    fcidump.write_fcidump(fc, mol)
    out = str(tmp_path / "fq") + "/"
    os.makedirs(out)
    res = subprocess.run([build.DRIVERS["fciqmc_mol_hip"], "--fcidump_path", fc, "--point_group", mol.point_group, "--distribution", "NU", "--target", "20000",
                          "--max_dets", "100000", "--epsilon", "0.004", "--initiator", "2", "--max_iter", "60", "--result_dir", out, "--seed", "9",
                          "--trial_vec", os.path.join(golden_io.GOLD, "n2_trial_"), "--ini_vec", os.path.join(golden_io.GOLD, "n2_fq_ini_")],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "Exception" not in res.stderr, res.stderr[-2000:]
    den = np.loadtxt(out + "projden.txt"); num = np.loadtxt(out + "projnum.txt")
    assert np.all(np.abs(den - lo["denom"][:60]) <= 1e-12 * np.abs(lo["denom"][:60])) and np.all(np.abs(num - lo["numer"][:60]) <= 1e-10 * np.maximum(1.0, np.abs(lo["numer"][:60])))


@pytest.mark.parametrize("name", sorted(golden_io.manifest().get("dense_runs", {})))
def test_frisys_dense_space_matches_reference_golden(Engine, mols, name):
    """--det_space on the device (fries_set_det_space): the dense determinants in front of the vector, H inside the space applied exactly
    as a perform_add of its own, the H compression and find_preserve / sys_comp restricted to the rest with mat_nonz - tot_dense_h
    samples, dense_norm added to the one-norm -- every count, norm, shift, projected-energy numerator / denominator and the stored
    vector against the reference's trajectory."""
    r = golden_io.manifest()["dense_runs"][name]
    g = golden_io.read_traj(name)
    space = np.array([int(x) for x in open(os.path.join(golden_io.GOLD, r["det_space"])).read().split()], dtype=np.uint64)
    eng = Engine(mols(r["shape"]))
    eng.setup(epsilon=r["epsilon"], vec_nonz=r["vec_nonz"], mat_nonz=r["mat_nonz"], max_dets=r["max_dets"], target_norm=r["target_norm"],
              initiator=r["initiator"], seed=r["seed"], distribution=r["distribution"], det_space=space)
    for row in g["rows"]:
        lg = eng.iterate(1)[0]
        assert int(lg["err"]) == 0
        for f in ("nkept", "n_nonz", "curr_size", "num_success"):
            assert int(lg[f]) == row[f], (row["it"], f, int(lg[f]), row[f])
        assert float(lg["norm"]) == row["norm"] and float(lg["shift"]) == row["shift"], row["it"]
        assert float(lg["numer"]) == row["numer"] and float(lg["denom"]) == row["denom"], row["it"]
        if row["it"] % 10 == 9 or row is g["rows"][-1]:
            d, v = eng.vector()
            assert golden_io.vec_hash(d, v) == row["hash"], row["it"]
            assert np.array_equal(d[:space.size], space)
    eng.close()


def test_cli_det_space(tmp_path):
    """frisys_mol_hip --det_space FILE (the integers DistVec::init_dense reads) against the reference's dense-space trajectory."""
    import subprocess
    from fries_amd import build
    name = "ne_m2000_dense"
    r = golden_io.manifest()["dense_runs"][name]
    g = golden_io.read_traj(name)
    mol = fcidump.synthetic(r["shape"])
    fc = str(tmp_path / "mol.FCIDUMP")
    fcidump.write_fcidump(fc, mol)
    out = str(tmp_path / "run") + "/"
    os.makedirs(out)
    n_it = 30
    cmd = [build.DRIVER, "--fcidump_path", fc, "--point_group", mol.point_group, "--distribution", r["distribution"], "--vec_nonz", str(r["vec_nonz"]),
           "--mat_nonz", str(r["mat_nonz"]), "--max_dets", str(r["max_dets"]), "--target", repr(r["target_norm"]), "--initiator", repr(r["initiator"]),
           "--epsilon", repr(r["epsilon"]), "--max_iter", str(n_it), "--result_dir", out, "--seed", str(r["seed"]), "--det_space", os.path.join(golden_io.GOLD, r["det_space"])]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "Exception" not in res.stderr, res.stderr[-2000:]
    num = np.loadtxt(out + "projnum.txt"); den = np.loadtxt(out + "projden.txt"); nk = np.loadtxt(out + "nkept.txt")
    sh = np.loadtxt(out + "S.txt").reshape(-1); nm = np.loadtxt(out + "norm.txt").reshape(-1)
    for i in range(n_it):
        row = g["rows"][i]
        assert num[i] == row["numer"] and den[i] == row["denom"] and int(nk[i]) == row["nkept"], i
    for k in range(n_it // 10):
        assert sh[k] == g["rows"][10 * k + 9]["shift"] and nm[k] == g["rows"][10 * k + 9]["norm"], k
