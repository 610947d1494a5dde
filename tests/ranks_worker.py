"""One rank of a hash-sharded FRI run (launched by torch.distributed.run from test_gpu_ranks.py / by hand).

Runs the golden configuration `name` (tests/golden/manifest.json "mpi_runs") with as many ranks as the reference
was run with under mpiexec, and compares what THIS rank logs with what the same rank of the reference logged:
scalars bit for bit, shard sizes, and the digest of the shard (position, determinant, value).
All ranks may share one GPU (backend gloo) -- the point is the exchange logic, not bandwidth.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    import torch
    import torch.distributed as dist
    import golden_io
    from fries_amd import fcidump
    from fries_amd.comm import TorchComm
    from fries_amd.engine import FriEngine

    name, backend, out_dir = sys.argv[1], sys.argv[2], sys.argv[3]
    n_check = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"]); local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    dev = local % max(ndev, 1)
    torch.cuda.set_device(dev)
    import datetime
    dist.init_process_group(backend, timeout=datetime.timedelta(seconds=180))     # a dead peer must surface as an error, not a hang
    man = golden_io.manifest()
    if name in man.get("fciqmc_mpi_runs", {}):
        return fciqmc_ranks(name, man["fciqmc_mpi_runs"][name], rank, world, dev, out_dir, dist, torch)
    if name in man.get("multi_mpi_runs", {}):           # frimulti_mol
        return multi_ranks(name, man["multi_mpi_runs"][name], rank, world, dev, out_dir, dist, torch)
    if name in man.get("fciqmc_fp_mpi_runs", {}):       # fciqmc_fp_mol: real-valued walkers
        return fciqmc_ranks(name, man["fciqmc_fp_mpi_runs"][name], rank, world, dev, out_dir, dist, torch)
    hh_full = name in man.get("hhfull_runs", {})
    hh = hh_full or name in man.get("hh_runs", {})
    if hh:
        r = man["hhfull_runs" if hh_full else "hh_runs"][name]
        assert r["n_ranks"] == world, (r["n_ranks"], world)
        g = golden_io.read_traj(name, rank=rank if world > 1 else None)
    elif name in man["mpi_runs"]:
        r = man["mpi_runs"][name]
        assert r["n_ranks"] == world, (r["n_ranks"], world)
        g = golden_io.read_traj(name, rank=rank)
    else:       # a one-rank golden driven through a one-rank communicator: same answers, every collective exercised
        r = man["runs"][name]
        assert world == 1
        g = golden_io.read_traj(name)
    res = dict(rank=rank, ok=True, fails=[])
    if hh:
        # frifull_hh ships every hop and phonon move: up to 4 n_elec adds per stored state
        comm = TorchComm(4 * r["n_elec"] * (r["vec_nonz"] + 64) + 4096 if hh_full else r["vec_nonz"], torch.device("cuda", dev))
        eng = FriEngine(None, device=dev, comm=comm)
        eng.setup_hh(n_elec=r["n_elec"], n_sites=r["n_sites"], eps=r["eps"], U=r["U"], omega=r["omega"], g=r["g"], gs_energy=r["gs_energy"],
                     vec_nonz=r["vec_nonz"], max_dets=r["max_dets"], target_norm=r["target_norm"], initiator=r["initiator"], seed=r["seed"], full=hh_full)
        step = eng.iterate_hh
    else:
        mol = fcidump.synthetic(r["shape"])
        comm = TorchComm(r["mat_nonz"], torch.device("cuda", dev))
        eng = FriEngine(mol, device=dev, comm=comm)
        eng.setup(epsilon=r["epsilon"], vec_nonz=r["vec_nonz"], mat_nonz=r["mat_nonz"], max_dets=r["max_dets"], target_norm=r["target_norm"],
                  initiator=r["initiator"], seed=r["seed"], distribution=r["distribution"])
        step = eng.iterate
        if eng.p_doub != g["p_doub"]:
            res["fails"].append(("p_doub", eng.p_doub, g["p_doub"]))
    rows = g["rows"][:n_check] if n_check else g["rows"]
    for row in rows:
        lg = step(1)[0]
        for f in ("norm", "shift"):
            if float(lg[f]) != row[f]:
                res["fails"].append((row["it"], f, float(lg[f]), row[f]))
        for f in ("numer", "denom"):       # block-parallel dot products: 1e-10 relative, not bit-exact
            if abs(float(lg[f]) - row[f]) > 1e-10 * max(1.0, abs(row[f])):
                res["fails"].append((row["it"], f, float(lg[f]), row[f]))
        for f in ("nkept", "n_nonz", "curr_size", "num_success"):
            if int(lg[f]) != row[f]:
                res["fails"].append((row["it"], f, int(lg[f]), row[f]))
        if int(lg["err"]):
            res["fails"].append((row["it"], "err", int(lg["err"])))
        if len(res["fails"]) > 8:
            break
    d, v = eng.vector()
    if not res["fails"] and golden_io.vec_hash(d, v) != rows[-1]["hash"]:
        res["fails"].append(("digest",))
    # optionally: pivotal compression of the sharded vector (piv_comp_parallel over the ranks) against the in-process rank oracle
    # optionally: apply_HBPP_piv over the ranks (every compression inside is collective) against the in-process rank oracle
    hbpiv = int(os.environ.get("FRIES_RANKS_HBPIV", "0"))
    if hbpiv and not hh and not res["fails"]:
        import oracle_lib
        orc = oracle_lib.OracleRanks(world, mol, epsilon=r["epsilon"], vec_nonz=r["vec_nonz"], mat_nonz=r["mat_nonz"], max_dets=r["max_dets"],
                                     target_norm=r["target_norm"], initiator=r["initiator"], seed=r["seed"], distribution=r["distribution"])
        orc.iterate(len(rows))
        eng.restart(4242)
        orc.restart(4242)
        pos, orbs, vals, sl = eng.apply_hbpp_piv(hbpiv)
        opos, oorbs, ovals, osl = orc.apply_hbpp_piv(hbpiv, rank, hbpiv + 4096)
        if not (len(pos) == len(opos) and np.array_equal(pos, opos) and np.array_equal(orbs, oorbs) and vals.tobytes() == ovals.tobytes()):
            res["fails"].append(("apply_hbpp_piv over ranks", len(pos), len(opos)))
        if sl.tolist() != osl.tolist():
            res["fails"].append(("apply_hbpp_piv stage lengths", sl.tolist(), osl.tolist()))
        res["hbpiv_samples"] = int(len(pos)); res["hbpiv_stages"] = sl.tolist()
    piv_budget = int(os.environ.get("FRIES_RANKS_PIV", "0"))
    if piv_budget and not hh and not res["fails"]:
        import oracle_lib
        orc = oracle_lib.OracleRanks(world, mol, epsilon=r["epsilon"], vec_nonz=r["vec_nonz"], mat_nonz=r["mat_nonz"], max_dets=r["max_dets"],
                                     target_norm=r["target_norm"], initiator=r["initiator"], seed=r["seed"], distribution=r["distribution"])
        orc.iterate(len(rows))
        od0, ov0 = orc.vector(rank)
        if not (np.array_equal(ov0, v) and np.array_equal(od0[ov0 != 0], d[v != 0])):
            res["fails"].append(("oracle ranks differ before the compression",))
        eng.restart(4242)
        orc.restart(4242)
        eng.compress_vec_piv(piv_budget)
        orc.compress_piv(piv_budget)
        d2, v2 = eng.vector()
        od, ov = orc.vector(rank)
        if not np.array_equal(ov, v2):
            res["fails"].append(("pivotal compression", int(np.sum(ov != v2)), int(np.count_nonzero(v2)), int(np.count_nonzero(ov))))
        res["piv_nonzero"] = int(np.count_nonzero(v2))
        res["piv_stats"] = list(eng.piv_stats())
    res["ok"] = not res["fails"]
    res["n_allgather"] = comm.n_allgather; res["n_alltoallv"] = comm.n_alltoallv; res["iters"] = len(rows)
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump(res, f, default=str)
    dist.barrier()
    eng.close()
    dist.destroy_process_group()
    sys.exit(0 if res["ok"] else 1)


def fciqmc_ranks(name, r, rank, world, dev, out_dir, dist, torch):
    """fciqmc_mol sharded over the ranks on the device against the in-process rank oracle, both on the counter-based uniform stream
    (the oracle's mt19937 mode is what is pinned against the reference under mpiexec): this rank's counts, shard and shift at
    every iteration, the projected energy within 1e-10."""
    from fries_amd import fcidump
    from fries_amd.comm import TorchComm
    from fries_amd.engine import FriEngine
    import oracle_lib
    assert r["n_ranks"] == world or world == 1        # world 1: the same configuration through a one-rank communicator (RCCL test)
    mol = fcidump.synthetic(r["shape"])
    n_it = min(r["n_iter"], 100)
    comm = TorchComm(2 * r["target_walkers"] + 4096, torch.device("cuda", dev))
    eng = FriEngine(mol, device=dev, comm=comm)
    fp = bool(r.get("fp"))
    eng.setup_fciqmc(epsilon=r["epsilon"], target_walkers=r["target_walkers"], max_dets=r["max_dets"], initiator=r["initiator"], seed=r["seed"],
                     distribution=r["distribution"], fp=fp)
    orc = oracle_lib.OracleFciqmcRanks(world, mol, epsilon=r["epsilon"], target_walkers=r["target_walkers"], max_dets=r["max_dets"], initiator=r["initiator"],
                                       seed=r["seed"], counter_rng=True, distribution=r["distribution"], fp=fp)
    lo = orc.iterate(n_it)[rank]
    res = dict(rank=rank, ok=True, fails=[])
    lg = eng.iterate_fciqmc(n_it)
    for i in range(n_it):
        for f in ("n_nonz", "n_ini", "curr_size", "n_spawn"):
            if int(lg[f][i]) != int(lo[f][i]):
                res["fails"].append((i, f, int(lg[f][i]), int(lo[f][i])))
        for f in ("shift", "norm", "denom"):
            if float(lg[f][i]) != float(lo[f][i]):
                res["fails"].append((i, f, float(lg[f][i]), float(lo[f][i])))
        if abs(float(lg["numer"][i]) - float(lo["numer"][i])) > 1e-10 * max(1.0, abs(float(lo["numer"][i]))):
            res["fails"].append((i, "numer", float(lg["numer"][i]), float(lo["numer"][i])))
        if len(res["fails"]) > 8:
            break
    d, v = eng.vector()
    od, ov = orc.vector(rank)
    if not res["fails"] and not (np.array_equal(v, ov) and np.array_equal(d[v != 0], od[ov != 0])):
        res["fails"].append(("shard differs",))
    res["ok"] = not res["fails"]
    res["n_allgather"] = comm.n_allgather; res["n_alltoallv"] = comm.n_alltoallv; res["iters"] = n_it
    res["walkers"] = float(np.abs(v).sum())
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump(res, f, default=str)
    dist.barrier()
    eng.close()
    dist.destroy_process_group()
    sys.exit(0 if res["ok"] else 1)


def multi_ranks(name, r, rank, world, dev, out_dir, dist, torch):
    """frimulti_mol sharded over the ranks on the device against the in-process rank oracle, both on the counter-based uniform stream (the
    oracle's mt19937 mode is what is pinned against the reference under mpiexec): this rank's counts, shard, norms and shift at every
    iteration, the projected energy within 1e-10."""
    from fries_amd import fcidump
    from fries_amd.comm import TorchComm
    from fries_amd.engine import FriEngine
    import oracle_lib
    assert r["n_ranks"] == world
    mol = fcidump.synthetic(r["shape"])
    n_it = r["n_iter"]
    par = dict(epsilon=r["epsilon"], vec_nonz=r["vec_nonz"], mat_nonz=r["mat_nonz"], max_dets=r["max_dets"], initiator=r["initiator"], target_norm=r["target_norm"], seed=r["seed"])
    comm = TorchComm(2 * r["mat_nonz"] + 4096, torch.device("cuda", dev))
    eng = FriEngine(mol, device=dev, comm=comm)
    eng.setup_multi(**par)
    orc = oracle_lib.OracleMultiRanks(world, mol, counter_rng=True, **par)
    lo = orc.iterate(n_it)[rank]
    res = dict(rank=rank, ok=True, fails=[])
    lg = eng.iterate_multi(n_it)
    for i in range(n_it):
        for f in ("n_nonz", "n_ini", "curr_size", "n_spawn"):
            if int(lg[f][i]) != int(lo[f][i]):
                res["fails"].append((i, f, int(lg[f][i]), int(lo[f][i])))
        for f in ("shift", "norm", "denom"):
            if float(lg[f][i]) != float(lo[f][i]):
                res["fails"].append((i, f, float(lg[f][i]), float(lo[f][i])))
        if abs(float(lg["numer"][i]) - float(lo["numer"][i])) > 1e-10 * max(1.0, abs(float(lo["numer"][i]))):
            res["fails"].append((i, "numer", float(lg["numer"][i]), float(lo["numer"][i])))
        if int(lg["err"][i]):
            res["fails"].append((i, "err", int(lg["err"][i])))
        if len(res["fails"]) > 8:
            break
    d, v = eng.vector()
    od, ov = orc.vector(rank)
    if not res["fails"] and not (np.array_equal(v, ov) and np.array_equal(d[v != 0], od[ov != 0])):
        res["fails"].append(("shard differs",))
    res["ok"] = not res["fails"]
    res["n_allgather"] = comm.n_allgather; res["n_alltoallv"] = comm.n_alltoallv; res["iters"] = n_it
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump(res, f, default=str)
    dist.barrier()
    eng.close()
    dist.destroy_process_group()
    sys.exit(0 if res["ok"] else 1)


if __name__ == "__main__":
    main()
