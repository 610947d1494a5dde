#!/usr/bin/env python3
"""Headline benchmark: FRI iterations/s (and spawns/s) at fixed m on MI355X.

Workload (BASELINE.json configs[1]): N2/cc-pVDZ-shaped synthetic FCIDUMP (26 orbitals, 10
electrons, D2h), frisys_mol with HB_unnorm, vec_nonz = mat_nonz = target = m = 1e6,
initiator 1, epsilon 0.01, one MI355X.  A "step" is one FRI iteration (frisys_mol.cpp:405-552)
with the vector full; every array is resident in HBM when the timed region starts.

Steady state is reached by a restart, like the reference's --load_dir: a filler run with
initiator 0 populates m determinants in a few dozen iterations, its vector is rescaled to the
target norm and loaded into the measured engine (and, for the CPU baseline, into the oracle).

`--gpus N`: STRONG scaling -- the same global m hash-sharded over N ranks, one per MI355X, RCCL; `value` is the global
iterations/s.  Without a launcher around it (WORLD_SIZE unset) and N > 1 the script starts its own N ranks with
torch.distributed.run before touching a GPU; it refuses to run when fewer than N GPUs are visible or when the process group's
size is not N.  At N == 8 (or --config4 1) BASELINE config 4 (H2O-shaped, m = 1e7 over the ranks) is timed as well (key `config4`).

Prints ONE JSON line (rank 0).  Extra keys: spawns_per_s, roofline, cpu_baseline, parity.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)

# ALGORITHMIC bytes per unit for the kernels that can dominate (DESIGN.md section 4;
# V = 8 B value, D = 8 B determinant, 8 B = parent index + orbital code).
ALG_BYTES = {
    # one replay of find_keep_sub over a stage: read value + (parent idx, code) + parent determinant
    "k_fks_sweep": lambda units: 24.0 * units,
    "k_sys_count": lambda units: 24.0 * units,
    "k_sys_write": lambda units: 40.0 * units,
    "k_prep": lambda units: 40.0 * units,
}


STARTUP = {}      # the filler run from the Hartree-Fock determinant up to m determinants: the start-up ("collapse") regime, outside the timed region


def build_state(mol, m_glob, max_dets, seed, device, comm, dist):
    """Filler run (initiator 0) -> this rank's shard of a vector with ~m_glob determinants at norm ~ m_glob."""
    from fries_amd.engine import FriEngine
    eng = FriEngine(mol, device=device, comm=comm)
    eng.setup(epsilon=0.01, vec_nonz=m_glob, mat_nonz=m_glob, max_dets=max_dets, target_norm=0.0, initiator=0.0, seed=seed, distribution="HB_unnorm")

    def glob(x):
        if dist is None:
            return float(x)
        import torch
        t = torch.tensor([float(x)], dtype=torch.float64, device="cuda")
        dist.all_reduce(t)
        return float(t.item())

    t0 = time.perf_counter()
    n_fill = 0
    for _ in range(200):
        lg = eng.iterate(5)
        n_fill += 5
        if glob(lg["n_nonz"][-1]) >= m_glob:
            break
    # (iterate() returns after the device has finished: each call reads its log back)
    STARTUP.update(iterations=n_fill, seconds=time.perf_counter() - t0)
    eng.iterate(10)
    dets, vals = eng.vector()
    eng.close()
    keep = vals != 0
    dets, vals = dets[keep], vals[keep]
    vals = vals * (float(m_glob) / glob(np.abs(vals).sum()))       # norm == target: about half the elements are initiators
    return dets, vals


def cpu_baseline(args, mol, par, dets, vals, run_seed, m):
    """-> (cpu_baseline dict, per-iteration log of the CPU run).  Bounded to roughly 20 s of CPU work."""
    import struct
    import subprocess
    import tempfile
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    from fries_amd import fcidump
    n_cpu = args.cpu_iters if args.cpu_iters > 0 else max(10, min(40, int(14.0 * 1.0e6 / m)))
    if os.path.exists(harness):
        try:
            tmp = tempfile.mkdtemp(prefix="fries_bench_")
            fc = os.path.join(tmp, "mol.FCIDUMP")
            fcidump.write_fcidump(fc, mol)
            st = os.path.join(tmp, "state.bin")
            with open(st, "wb") as f:
                f.write(struct.pack("<Q", dets.size)); f.write(dets.astype("<u8").tobytes()); f.write(vals.astype("<f8").tobytes())
            log = os.path.join(tmp, "log.txt")
            base = [fc, mol.point_group, str(n_cpu), str(par["seed"]), repr(par["epsilon"]), str(par["vec_nonz"]), str(par["mat_nonz"]), str(par["max_dets"]),
                    repr(par["initiator"]), repr(par["target_norm"]), par["distribution"], st, str(run_seed)]
            out = subprocess.run([harness, "restart"] + base + [log], capture_output=True, text=True, timeout=600)
            line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
            r = json.loads(line)
            rows = [ln.split() for ln in open(log)]
            ref_log = {"numer": [float.fromhex(t[1]) for t in rows], "denom": [float.fromhex(t[2]) for t in rows], "norm": [float.fromhex(t[3]) for t in rows],
                       "nkept": [int(t[5]) for t in rows], "n_nonz": [int(t[6]) for t in rows], "curr_size": [int(t[7]) for t in rows], "num_success": [int(t[8]) for t in rows]}
            cb = {"value": r["iters_per_s"], "unit": "iterations/s", "cores": 1, "kind": "reference",
                  "sample": f"{n_cpu} iterations of the reference's frisys_mol loop (oracle/_ref/ref_harness restart, 1 MPI rank) from the same restart state and seed"}
            # the reference under MPI on this box's cores (its only parallelism), same state: reported beside, not as `value`
            mpiexec = "/opt/conda/bin/mpiexec"
            ncore = min(8, os.cpu_count() or 1)
            if os.path.exists(mpiexec) and ncore > 1:
                try:
                    o2 = subprocess.run([mpiexec, "-n", str(ncore), harness, "restart"] + base, capture_output=True, text=True, timeout=600)
                    r2 = json.loads([ln for ln in o2.stdout.splitlines() if ln.startswith("{")][-1])
                    cb["mpi"] = {"value": r2["iters_per_s"], "cores": ncore, "sample": f"mpiexec -n {ncore}, {n_cpu} iterations, same state"}
                except Exception as e:       # MPI launcher unusable on this box: keep the 1-rank figure
                    cb["mpi"] = {"value": None, "error": str(e)[:200]}
            return cb, ref_log
        except Exception as e:
            sys.stderr.write(f"reference harness unusable here ({e}); timing the oracle port instead\n")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    orc = oracle_lib.OracleFrisys(mol, **par)
    orc.vec_load(dets, vals)
    orc.restart(run_seed, 0.0, 0.0, 0)
    t0 = time.perf_counter()
    lo = orc.iterate(n_cpu)
    cpu_dt = time.perf_counter() - t0
    cb = {"value": n_cpu / cpu_dt, "unit": "iterations/s", "cores": 1, "kind": "port",
          "sample": f"{n_cpu} iterations of the same restart state and seed (oracle/fries_oracle.cpp, 1 thread)"}
    return cb, {f: [x for x in lo[f]] for f in ("numer", "denom", "norm", "nkept", "n_nonz", "curr_size", "num_success")}


def run_workload(args, shape, m_glob, world, rank, device, dist, steps, warmup, primary):
    """One timed run of the frisys_mol loop on a hash-sharded vector of m_glob non-zeros over `world` ranks.  Strong scaling:
    m_glob is the GLOBAL budget (vec_nonz = mat_nonz = target), whatever the number of ranks -- the run
    `mpiexec -n N frisys_mol --vec_nonz m --mat_nonz m` does, one rank per MI355X.  -> result dict on rank 0, None elsewhere."""
    from fries_amd import fcidump
    from fries_amd.engine import FriEngine

    mol = fcidump.synthetic(shape)
    seed = 20250215
    # per-rank capacity: 4 m / N slots plus slack for the hash's load imbalance (a few sigma of a binomial) -- the reference
    # sizes every rank's table with the same --max_dets
    max_dets = int(4 * m_glob / world * (1.0 if world == 1 else 1.15)) + (0 if world == 1 else 65536)
    comm = None
    if dist is not None:
        import torch
        if args.transport == "rccl":
            # native librccl transport; should its communicator fail to come up on some rank, every rank falls back to the
            # torch.distributed callbacks (also RCCL underneath) -- the ranks agree on that through the torch group
            from fries_amd.comm import RcclComm
            err = None
            try:
                comm = RcclComm(m_glob, device, dist)
            except Exception as e:      # noqa: BLE001 -- reported below
                err = e
            flag = torch.tensor([1.0 if err is not None else 0.0], device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if float(flag.item()) > 0:
                sys.stderr.write(f"rank {rank}: native RCCL transport unavailable ({err}); using the torch.distributed transport\n")
                args.transport = "torch"
                comm = None
        if args.transport != "rccl":
            from fries_amd.comm import TorchComm
            comm = TorchComm(m_glob, torch.device("cuda", device))
    dets, vals = build_state(mol, m_glob, max_dets, seed, device, comm, dist)
    par = dict(epsilon=0.01, vec_nonz=m_glob, mat_nonz=m_glob, max_dets=max_dets, target_norm=float(m_glob), initiator=1.0, seed=seed, distribution="HB_unnorm")
    eng = FriEngine(mol, device=device, comm=comm)
    eng.setup(**par)
    eng.vec_load(dets, vals)            # this rank's shard
    run_seed = 777                      # every rank draws the same uniforms (the reference broadcasts rank 0's)
    eng.restart(run_seed, 0.0, 0.0, 0)

    def barrier():
        if dist is not None:
            dist.barrier()
            import torch
            torch.cuda.synchronize()

    first_logs = eng.iterate(warmup) if warmup else None
    c0 = eng.counters()
    coll0 = comm.n_collectives() if comm is not None else 0
    barrier()
    t0 = time.perf_counter()
    eng.iterate(steps, want_logs=False)
    eng.vec_info()                      # drains the engine's stream
    barrier()
    dt = time.perf_counter() - t0
    c1 = eng.counters()
    n_coll = (comm.n_collectives() - coll0) if comm is not None else 0
    if dist is not None:
        import torch
        t = torch.tensor([dt], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        sp = torch.tensor([float(c1["spawns"] - c0["spawns"])], device="cuda")
        dist.all_reduce(sp)
        spawns = float(sp.item())
        world_seen, backend_seen = dist.get_world_size(), dist.get_backend()
    else:
        spawns = float(c1["spawns"] - c0["spawns"])
        world_seen, backend_seen = 1, None
    iters_per_s = steps / dt            # GLOBAL iterations per second at fixed global m (BASELINE.json's metric)
    if world == 1:
        par_str = "1 GPU, 1 rank"
    else:
        par_str = (f"{world_seen} ranks ({backend_seen}, transport {args.transport}), one rank per MI355X, one hash-sharded vector of {m_glob} non-zeros "
                   f"(~{m_glob // world} per GPU): one all-to-all of the spawns + rank-ordered all-gathers per iteration")
    result = {
        "metric": "fri_iterations_per_s", "value": iters_per_s, "unit": "iterations/s",
        "n_gpus": world_seen, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * dt / steps,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{shape} cc-pVDZ-shaped synthetic FCIDUMP, frisys_mol HB_unnorm, vec_nonz=mat_nonz=target={m_glob} (global, fixed as N grows), initiator 1, eps 0.01, restart from a full vector",
                   "m": m_glob, "m_per_gpu": m_glob / world, "parallelism": par_str},
        "spawns_per_s": spawns / dt,
        "kernel_launches_per_iter": (c1["launches"] - c0["launches"]) / steps,
        "fks_replays_per_iter": (c1["fks_replays"] - c0["fks_replays"]) / steps,
    }
    if comm is not None:
        result["collectives_per_iter"] = n_coll / max(1, steps)
    if STARTUP:
        result["startup"] = {"iterations": STARTUP["iterations"], "ms_per_iter": 1e3 * STARTUP["seconds"] / max(1, STARTUP["iterations"]),
                             "note": "filler run from the HF determinant to m determinants (initiator 0), find_keep_sub in the reference's own order where its running norm collapses (DESIGN.md section 2); not part of `value`"}
    if not primary:
        eng.close()
        return result if rank == 0 else None

    # ---- roofline of the dominant kernel, HIP events on the engine's stream (rank 0 times; every rank iterates)
    m = m_glob
    info = eng.vec_info()
    cA = eng.counters()
    if rank == 0:
        eng.prof_enable(True)
    eng.iterate(args.profile_steps, want_logs=False)
    eng.vec_info()
    if rank == 0:
        rep = eng.prof_report()
        eng.prof_enable(False)
        cB = eng.counters()
        tot_ms = sum(v[0] for v in rep.values())
        # the dominant COMPUTE kernel: with ranks, the small kernels that follow a collective also absorb the wait for it in their
        # event times and would otherwise come out on top without saying anything about the hardware
        # (k_fks_sweep_light is not a pass over the stage: it re-decides the few waves k_fks_check marks and has no per-element byte count)
        known = {k: v for k, v in rep.items() if any(k.startswith(b) for b in ALG_BYTES) and k != "k_fks_sweep_light"}
        dom = max((known or rep).items(), key=lambda kv: kv[1][0])
        name, (ms, calls) = dom
        avg_s = ms / calls * 1e-3
        stage_elems = (cB["stage_elems"] - cA["stage_elems"]) / args.profile_steps      # elements over the five stages, per iteration (this rank)
        base = next((k for k in ALG_BYTES if name.startswith(k)), None)
        units = stage_elems / 5.0 if base else None      # elements one launch passes over (stage average)
        ach = ALG_BYTES[base](units) / avg_s / 1e9 if base else None
        # HBM bytes per launch from the PMC passes kept under profiles/ (separate --pmc FETCH_SIZE / WRITE_SIZE runs of this
        # command, corrected as MI355X_MICROARCH.md prescribes); mean over the stage instantiations of the kernel
        traffic = None
        traffic_source = None
        for prof in ("r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
            try:
                with open(os.path.join(ROOT, "profiles", prof)) as f:
                    tk = json.load(f)["kernels"]
                # the bench's label "k_fks_sweep" is the lean replay = template instantiations <stage, *, 0> of the kernel
                want_mode = {"k_fks_sweep": ", 0>", "k_fks_sweep_rec": ", 1>", "k_fks_sweep_light": ", 3>", "k_fks_final": ", 2>", "k_fks_close": ", 4>"}.get(name)
                kname = "k_fks_sweep" if want_mode else name
                vals_t = [v["bytes_per_launch"] for k, v in tk.items() if (k.startswith(kname + "<") and (want_mode is None or k.endswith(want_mode))) or k == kname]
                if vals_t and m == 1_000_000 and world == 1:
                    traffic = float(np.mean(vals_t))
                    traffic_source = f"profiles/{prof}: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, NOT measured in this run"
                    break
            except (OSError, KeyError, ValueError):
                pass
        result["roofline"] = {"bound": "hbm", "kernel": name, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": (ach / HBM_PEAK_GBS) if ach else None, "traffic": traffic,
                              "traffic_source": traffic_source,
                              "avg_launch_us": avg_s * 1e6, "calls_per_iter": calls / args.profile_steps,
                              "share_of_kernel_time": ms / tot_ms, "n_nonz": info[1]}
        # the other find_keep_sub launches beside the dominant one.  The light replays and the closing pass re-decide only the waves whose
        # inputs moved beyond their margins, so they have no per-element byte count: reported as time per launch and launches per iteration.
        result["roofline"]["find_keep_sub_launches"] = {k: {"avg_launch_us": v[0] / v[1] * 1e3, "calls_per_iter": v[1] / args.profile_steps,
                                                "achieved_GBs": (ALG_BYTES["k_fks_sweep"](units) / (v[0] / v[1] * 1e-3) / 1e9) if (units and k in ("k_fks_sweep", "k_fks_sweep_rec")) else None}
                                            for k, v in rep.items() if k.startswith("k_fks_") and v[1] > 0}
        # the measured device copy bandwidth of this box beside the nominal peak (read + write of 1 GiB on the
        # engine's stream, HIP events), and the whole iteration / the spawn term against it
        try:
            copy_gbs = eng.copy_bandwidth(1 << 30, 5)
            result["roofline"]["measured_copy_GBs"] = copy_gbs
            result["roofline"]["frac_of_measured_copy"] = (ach / copy_gbs) if ach else None
            result["iteration_frac_of_measured_copy"] = 312.0 * m * iters_per_s / world / 1e9 / copy_gbs
            result["spawn_term_GBs"] = 64.0 * result["spawns_per_s"] / world / 1e9       # 64 B per spawn (assembly + merge), per GPU
        except Exception as ex:      # the measurement is informative only
            result["roofline"]["measured_copy_GBs"] = None
            print("copy-bandwidth measurement skipped: %r" % (ex,), file=sys.stderr)
        result["kernel_time_ms_per_iter"] = tot_ms / args.profile_steps
        result["top_kernels"] = {k: {"ms_per_iter": v[0] / args.profile_steps, "calls_per_iter": v[1] / args.profile_steps}
                                 for k, v in sorted(rep.items(), key=lambda kv: -kv[1][0])[:int(os.environ.get("FRIES_BENCH_TOPK", "8"))]}
        # whole-iteration algorithmic traffic (~312 B per nonzero per iteration), per GPU
        result["iteration_alg_GBs"] = 312.0 * m * iters_per_s / world / 1e9

        # ---- CPU baseline on this box's host cores: the REAL reference (oracle/_ref, built from /root/reference by
        # oracle/Makefile and shipped as a binary) advanced from the same restart state and seed; the oracle port is the
        # fallback where that binary cannot run.  Either way it is a checker / yardstick, never the product path.
        if args.cpu_iters != 0 and world == 1:
            result["cpu_baseline"], ref_log = cpu_baseline(args, mol, par, dets, vals, run_seed, m)
            if ref_log is not None:
                # the parity leg is a replay of its own: a second engine from the same restart state and seed, as many iterations as the
                # reference was given (at least 10 at the default sample), logged one by one -- independent of --warmup
                k = len(ref_log["numer"])
                eng2 = FriEngine(mol, device=device, comm=None)
                eng2.setup(**par)
                eng2.vec_load(dets, vals)
                eng2.restart(run_seed, 0.0, 0.0, 0)
                first_logs = eng2.iterate(k)
                eng2.close()
                same = all(int(first_logs[f][i]) == int(ref_log[f][i]) for i in range(k) for f in ("num_success", "n_nonz", "curr_size", "nkept"))
                en_g = first_logs["numer"][:k] / first_logs["denom"][:k]
                en_r = np.asarray(ref_log["numer"][:k]) / np.asarray(ref_log["denom"][:k])
                result["parity"] = {"against": result["cpu_baseline"]["kind"], "iterations_compared": k, "counts_identical": bool(same),
                                    "energy_within_1e-10": bool(np.all(np.abs(en_g - en_r) < 1e-10)),
                                    "norm_bit_identical": bool(all(float(first_logs["norm"][i]) == float(ref_log["norm"][i]) for i in range(k)))}
    eng.close()
    return result if rank == 0 else None


def launch_ranks(args, argv, json_fd):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start N fresh worker processes (one per GPU, RCCL)
    BEFORE this process touches a GPU, relay rank 0's JSON line, return the launcher's exit code."""
    import socket
    import subprocess
    import torch        # device_count() does not initialise the GPU on this image
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus:
        if not os.environ.get("FRIES_BENCH_SHARE_GPU"):
            sys.stderr.write(f"bench.py: --gpus {args.gpus} but only {n_dev} GPU(s) are visible; refusing to report a smaller job as n_gpus={args.gpus}\n")
            return 3
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, env=env, stdout=json_fd)       # the ranks get the real stdout: rank 0 writes the one JSON line there
    return r.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--m", type=int, default=int(os.environ.get("FRIES_BENCH_M", "1000000")), help="GLOBAL non-zeros (vec_nonz = mat_nonz = target), fixed as N grows")
    ap.add_argument("--shape", default="N2")
    ap.add_argument("--cpu-iters", type=int, default=-1, help="oracle iterations for cpu_baseline (-1: sized to ~20 s; 0: skip)")
    ap.add_argument("--profile-steps", type=int, default=5)
    ap.add_argument("--backend", default=os.environ.get("FRIES_BENCH_BACKEND", "nccl"), help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--transport", default=os.environ.get("FRIES_BENCH_TRANSPORT", "rccl"), choices=["torch", "rccl"],
                    help="how the engine's collectives travel for N > 1: torch.distributed callbacks, or the native librccl transport (csrc/comm_rccl)")
    ap.add_argument("--config4", type=int, default=-1, help="also time BASELINE config 4 (H2O-shaped, m = 1e7 sharded over the ranks): -1 = only when N == 8")
    args = ap.parse_args()

    if args.gpus < 1:
        sys.stderr.write("bench.py: --gpus must be >= 1\n")
        sys.exit(2)
    # stdout carries ONE JSON line and nothing else: librccl prints a version banner to file descriptor 1 when a communicator
    # is created, so everything written to fd 1 from here on goes to stderr and the result line is written to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args, sys.argv[1:], json_fd))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks\n")
        sys.exit(3)
    if world > 1:
        import torch
        import torch.distributed as dist
        share = bool(os.environ.get("FRIES_BENCH_SHARE_GPU"))      # REHEARSAL of the N > 1 code path on a one-GPU box (gloo + torch transport): not a measurement
        if torch.cuda.device_count() < world and not share:
            sys.stderr.write(f"bench.py: {world} ranks but {torch.cuda.device_count()} GPU(s) visible: one rank per GPU is the contract\n")
            sys.exit(3)
        device = 0 if share else local_rank
        torch.cuda.set_device(device)
        dist.init_process_group(args.backend)
        if dist.get_world_size() != args.gpus:
            raise RuntimeError(f"process group has {dist.get_world_size()} ranks, --gpus {args.gpus}")
    elif os.environ.get("FRIES_BENCH_FORCE_COMM"):
        # one rank, but every collective goes through RCCL: what the callbacks cost without any link latency
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
        device = 0
        torch.cuda.set_device(0)
        dist.init_process_group(args.backend, rank=0, world_size=1)
    else:
        dist = None
        device = 0

    result = run_workload(args, args.shape, args.m, world, rank, device, dist, args.steps, args.warmup, primary=True)
    # BASELINE config 4 beside the headline: H2O-shaped, m = 1e7 hash-sharded over the ranks (default: only at N == 8)
    want4 = args.config4 > 0 or (args.config4 < 0 and world == 8)
    if want4:
        r4 = run_workload(args, "H2O", int(os.environ.get("FRIES_BENCH_M4", "10000000")), world, rank, device, dist, max(5, args.steps // 2), min(args.warmup, 5), primary=False)
        if rank == 0:
            result["config4"] = {k: r4[k] for k in ("value", "unit", "ms_per_step", "spawns_per_s", "steps", "warmup", "config", "kernel_launches_per_iter", "collectives_per_iter") if k in r4}
    if rank == 0:
        if os.environ.get("FRIES_BENCH_SHARE_GPU") and world > 1:
            result["data"] = "synthetic -- REHEARSAL: the ranks share ONE GPU (FRIES_BENCH_SHARE_GPU), not a multi-GPU measurement"
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    os.close(json_fd)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
