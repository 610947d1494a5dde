"""Replays a `ref_harness pin` record (tests/golden/*.pin) on the device: the reference's filler run from 100 x HF, the
restart (non-zero entries compacted, scaled to one-norm m, initiator 1, generator re-seeded) and the measured iterations --
bench.py's workload, iteration by iteration, against what the REAL reference logged at the BASELINE sizes.

Bar per iteration: counts (nkept, n_nonz, curr_size, num_success) equal, one-norm and shift bit-identical, projected-energy
numerator / denominator identical with one rank (the device adds the products in list order), within 1e-10 relative with ranks, the shard's digest (position,
determinant, value bits of every non-zero entry) equal on sampled iterations and on the last one.

Ranks are threads of this process talking through the native "local" transport (csrc/comm_native.hip), so 8 ranks can share
the box's one GPU."""
import threading

import numpy as np

import golden_io
from fries_amd import fcidump

REL_TOL = 1e-10


def _check_row(lg, row, fails, phase, exact=False):
    for f in ("nkept", "n_nonz", "curr_size", "num_success"):
        if int(lg[f]) != row[f]:
            fails.append((phase, row["it"], f, int(lg[f]), row[f]))
    for f in ("norm", "shift"):
        if float(lg[f]) != row[f]:
            fails.append((phase, row["it"], f, float(lg[f]).hex(), row[f].hex()))
    for f in ("numer", "denom"):
        # one rank: the device adds the products in list order like the reference's loop -> the same doubles; with ranks the reference's
        # gathered H * trial list is ordered by rank, ours by enumeration, so the partial sums round differently
        if (float(lg[f]) != row[f]) if exact else (abs(float(lg[f]) - row[f]) > REL_TOL * max(1.0, abs(row[f]))):
            fails.append((phase, row["it"], f, float(lg[f]), row[f]))
    if int(lg["err"]):
        fails.append((phase, row["it"], "err", int(lg["err"])))


class _Shared:
    def __init__(self, n):
        self.bar = threading.Barrier(n, timeout=600)
        self.norms = [0.0] * n


def replay_rank(name, r, rank, comm, device, shared, digest_every=10, max_run_iters=None):
    """-> (fails, info) for one rank.  r: manifest["pin_runs"][name]."""
    from fries_amd.engine import FriEngine
    P = r["n_ranks"]
    g = golden_io.read_pin(name, rank if P > 1 else None)
    mol = fcidump.synthetic(r["shape"])
    m = r["m"]
    fails = []
    info = {}
    eng = FriEngine(mol, device=device, comm=comm)
    eng.setup(epsilon=r["epsilon"], vec_nonz=m, mat_nonz=m, max_dets=r["max_dets"], target_norm=0.0, initiator=0.0, seed=r["seed"], distribution=r["distribution"])
    if eng.p_doub != g["p_doub"] or eng.hf_energy != g["hf_en"]:
        fails.append(("setup", eng.p_doub, g["p_doub"], eng.hf_energy, g["hf_en"]))
    for row in g["fill"]:
        lg = eng.iterate(1)[0]
        _check_row(lg, row, fails, "F", exact=(P == 1))
        if row["it"] % digest_every == digest_every - 1 or row is g["fill"][-1]:
            d, v = eng.vector()
            if golden_io.vec_hash(d, v) != row["hash"]:
                fails.append(("F", row["it"], "digest"))
        if len(fails) > 6:
            break
    d, v = eng.vector()
    eng.close()
    keep = v != 0
    d, v = d[keep], v[keep]
    loc = float(np.cumsum(np.abs(v))[-1]) if v.size else 0.0        # left-to-right, like the harness's loop
    shared.norms[rank] = loc
    shared.bar.wait()
    glob = 0.0
    for p in range(P):
        glob += shared.norms[p]                                       # sum_mpi: rank order
    rs = g["restart"]
    if (d.size, loc, glob) != (rs["n"], rs["loc_norm"], rs["glob_norm"]):
        fails.append(("restart", d.size, rs["n"], loc.hex(), rs["loc_norm"].hex(), glob.hex(), rs["glob_norm"].hex()))
    scale = float(m) / glob
    v = v * scale
    eng = FriEngine(mol, device=device, comm=comm)
    eng.setup(epsilon=r["epsilon"], vec_nonz=m, mat_nonz=m, max_dets=r["max_dets"], target_norm=float(m), initiator=1.0, seed=r["seed"], distribution=r["distribution"])
    eng.vec_load(d, v)
    eng.restart(r["run_seed"], 0.0, 0.0, 0)
    rows = g["run"][:max_run_iters] if max_run_iters else g["run"]
    c0 = eng.counters()
    for row in rows:
        lg = eng.iterate(1)[0]
        _check_row(lg, row, fails, "R", exact=(P == 1))
        if row["it"] % digest_every == digest_every - 1 or row is rows[-1]:
            dd, vv = eng.vector()
            if golden_io.vec_hash(dd, vv) != row["hash"]:
                fails.append(("R", row["it"], "digest"))
        if len(fails) > 6:
            break
    c1 = eng.counters()
    info.update(filler_iters=len(g["fill"]), run_iters=len(rows), n_restart=int(d.size), replays_per_iter=(c1["fks_replays"] - c0["fks_replays"]) / max(1, len(rows)),
                n_nonz_last=int(rows[-1]["n_nonz"]))
    eng.close()
    return fails, info


def replay(name, device=0, digest_every=10, max_run_iters=None):
    """Runs every rank of the pinned run `name`; -> list of (fails, info) per rank."""
    r = golden_io.manifest()["pin_runs"][name]
    P = r["n_ranks"]
    shared = _Shared(P)
    if P == 1:
        return [replay_rank(name, r, 0, None, device, shared, digest_every, max_run_iters)]
    from fries_amd.comm import LocalGroup
    grp = LocalGroup(P, r["m"])
    comms = [grp.comm(k, device) for k in range(P)]
    out = [None] * P

    def work(k):
        try:
            out[k] = replay_rank(name, r, k, comms[k], device, shared, digest_every, max_run_iters)
        except Exception as e:      # the other ranks then time out in their next collective instead of hanging
            out[k] = ([("exception", repr(e))], {})
            shared.bar.abort()

    th = [threading.Thread(target=work, args=(k,)) for k in range(P)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    grp.destroy()
    return out
