"""Readers for tests/golden (fixtures emitted by oracle/gen_golden.py from the real reference)."""
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def manifest():
    with open(os.path.join(GOLD, "manifest.json")) as f:
        return json.load(f)


def read_traj(name, rank=None):
    """rank=None: a single-rank .traj; rank=r: the .traj.r<r> file rank r wrote under mpiexec."""
    rows, snaps = [], {}
    p_doub = hf_en = None
    n_htrial = hf_proc = None
    with open(os.path.join(GOLD, name + ".traj" + ("" if rank is None else f".r{rank}"))) as f:
        lines = f.read().splitlines()
    i = 0
    while i < len(lines):
        ln = lines[i]
        if ln.startswith("# p_doub"):
            t = ln.split()
            p_doub, hf_en = float.fromhex(t[2]), float.fromhex(t[4])
            if len(t) > 8:
                n_htrial, hf_proc = int(t[6]), int(t[8])
        elif ln.startswith("SNAP"):
            it = int(ln.split()[1])
            ent = []
            i += 1
            while lines[i] != "ENDSNAP":
                a, b, c = lines[i].split()
                ent.append((int(a), int(b, 16), float.fromhex(c)))
                i += 1
            snaps[it] = ent
        elif ln and not ln.startswith("#"):
            t = ln.split()
            rows.append(dict(it=int(t[0]), numer=float.fromhex(t[1]), denom=float.fromhex(t[2]), norm=float.fromhex(t[3]), shift=float.fromhex(t[4]),
                             nkept=int(t[5]), n_nonz=int(t[6]), curr_size=int(t[7]), num_success=int(t[8]), hash=int(t[9], 16)))
        i += 1
    return dict(rows=rows, snaps=snaps, p_doub=p_doub, hf_en=hf_en, n_htrial=n_htrial, hf_proc=hf_proc)


def vec_hash(dets, vals):
    """The FNV-style digest oracle/ref_harness.cpp writes per iteration: non-zero entries by position."""
    if len(vals) > 20000:       # the C loop of the test-side oracle library; the Python loop below is the definition
        import oracle_lib
        return oracle_lib.vec_digest(dets, vals)
    mask = (1 << 64) - 1
    h = 1469598103934665603
    p = 1099511628211
    vb = np.ascontiguousarray(vals, dtype=np.float64).view(np.uint64)
    for i in np.nonzero(vals != 0)[0]:
        h = ((h ^ int(dets[i])) * p) & mask
        h = ((h ^ int(vb[i])) * p) & mask
        h = ((h ^ int(i)) * p) & mask
    return h


def read_hbpp_all():
    out = dict(orbs=[], vals=[], piv_orbs=[], piv_vals=[], tens={})
    ko, kv = "orbs", "vals"
    with open(os.path.join(GOLD, "hbpp_all.txt")) as f:
        for ln in f:
            t = ln.split()
            if not t or t[0].startswith("#"):
                continue
            if t[0] == "RN":
                out["rn"] = [float.fromhex(x) for x in t[1:6]]
            elif t[0] == "N":
                out["n"] = int(t[1])
            elif t[0] == "PIV":      # the pivotal half of the reference test: same samples from apply_HBPP_piv
                out["piv_n"] = int(t[1]); ko, kv = "piv_orbs", "piv_vals"
            elif t[0] == "HB":
                out["tens"]["s_norm"] = [float.fromhex(t[1])]
            elif t[0] in ("s_tens", "d_diff", "d_same", "exch_sqrt", "diag_sqrt", "exch_norms"):
                out["tens"][t[0]] = [float.fromhex(x) for x in t[2:]]
            else:
                out[ko].append([int(x) for x in t[:4]])
                out[kv].append(float.fromhex(t[4]))
    for k in ("orbs", "piv_orbs"):
        out[k] = np.array(out[k], dtype=np.uint8).reshape(-1, 4)
    for k in ("vals", "piv_vals"):
        out[k] = np.array(out[k])
    return out


# [new_hb_all] configuration (tests/test_hamiltonian.cpp:454-487 of the reference): Ne-like, 22 orbitals, 8 unfrozen electrons
HBPP_ALL_SYMM = [0, 5, 6, 7, 0, 5, 6, 7, 0, 0, 1, 2, 3, 5, 6, 7, 0, 0, 0, 1, 2, 3]
TENSOR_ID = {"s_tens": 0, "d_same": 1, "d_diff": 2, "exch_sqrt": 3, "diag_sqrt": 4, "exch_norms": 5, "s_norm": 6}


def read_piv_cases():
    """tests/golden/piv_comp.txt: what the reference's piv_comp_parallel returned (one rank)."""
    cases = []
    with open(os.path.join(GOLD, "piv_comp.txt")) as f:
        cur = None
        for line in f:
            t = line.split()
            if not t or t[0].startswith("#"):
                continue
            if t[0] == "case":
                cur = dict(len=int(t[1]), compress_size=int(t[2]), seed=int(t[3]), inp=[], out=[], flag=[])
                cases.append(cur)
            elif t[0] == "next":
                cur["next"] = int(t[1])
            else:
                cur["inp"].append(float.fromhex(t[0])); cur["out"].append(float.fromhex(t[1])); cur["flag"].append(int(t[2]))
    return cases


def read_multi_traj(name, raw=False):
    """tests/golden/multi_*.traj: the reference's frimulti_mol loop (oracle/ref_harness.cpp: run_frimulti).  raw: name is the file name."""
    rows = []
    with open(os.path.join(GOLD, name if raw else name + ".traj")) as f:
        for line in f:
            t = line.split()
            if not t or t[0].startswith("#"):
                continue
            rows.append(dict(it=int(t[0]), numer=float.fromhex(t[1]), denom=float.fromhex(t[2]), norm=float.fromhex(t[3]), shift=float.fromhex(t[4]), nkept=int(t[5]),
                             n_nonz=int(t[6]), curr_size=int(t[7]), n_spawn=int(t[8]), n_ini=int(t[9]), hash=int(t[10], 16)))
    return rows


def read_hbpiv(name):
    """tests/golden/<name>.txt: what the reference's apply_HBPP_piv returned on the vector of a golden run (oracle/ref_harness.cpp: run_hbpiv)."""
    cases = []
    with open(os.path.join(GOLD, name + ".txt")) as f:
        cur = None
        for line in f:
            t = line.split()
            if not t or t[0].startswith("#"):
                continue
            if t[0] == "CASE":
                cur = dict(n_samp=int(t[1]), seed=int(t[2]), n_out=int(t[3]), stage_len=[int(x) for x in t[4:9]], pos=[], orbs=[], val=[])
                cases.append(cur)
            else:
                cur["pos"].append(int(t[0])); cur["orbs"].append([int(x) for x in t[1:5]]); cur["val"].append(float.fromhex(t[5]))
    for c in cases:
        c["pos"] = np.array(c["pos"], dtype=np.uint32); c["orbs"] = np.array(c["orbs"], dtype=np.uint8).reshape(-1, 4); c["val"] = np.array(c["val"])
    return cases


def read_tr(name):
    """tests/golden/<name>.txt (oracle/ref_harness.cpp: run_tr): the source vector and, for spin parity +1 and -1, what the reference's
    h_op_offdiag left in column 1 of the vector (stored determinants in position order)."""
    src_d, src_v, out, cur = [], [], {}, None
    with open(os.path.join(GOLD, name + ".txt")) as f:
        for line in f:
            t = line.split()
            if not t or t[0].startswith("#"):
                continue
            if t[0] == "SRC":
                src_d.append(int(t[1])); src_v.append(float.fromhex(t[2]))
            elif t[0] == "PARITY":
                cur = int(t[1]); out[cur] = ([], [])
            else:
                out[cur][0].append(int(t[0])); out[cur][1].append(float.fromhex(t[1]))
    return np.array(src_d, dtype=np.uint64), np.array(src_v), {k: (np.array(d, dtype=np.uint64), np.array(v)) for k, (d, v) in out.items()}


def read_text_vector(prefix):
    """<prefix>dets / <prefix>vals under tests/golden: the reference's text vector format (io_utils.cpp:447-482, 565-586)."""
    with open(os.path.join(GOLD, prefix + "dets")) as f:
        dets = np.array([int(t) for t in f.read().split()], dtype=np.int64).astype(np.uint64)
    with open(os.path.join(GOLD, prefix + "vals")) as f:
        vals = np.array([float(t) for t in f.read().split()])
    n = min(dets.size, vals.size)
    return dets[:n], vals[:n]


def read_pin(name, rank=None):
    """tests/golden/<name>.pin[.r<rank>] from `ref_harness pin`: the reference's filler run ("F" rows), the restart record
    (entries kept, local / global one-norm, scale factor) and the measured run ("R" rows)."""
    fill, run, restart, hdr = [], [], None, {}
    with open(os.path.join(GOLD, name + ".pin" + ("" if rank is None else f".r{rank}"))) as f:
        for ln in f:
            t = ln.split()
            if not t:
                continue
            if ln.startswith("# p_doub"):
                hdr = dict(p_doub=float.fromhex(t[2]), hf_en=float.fromhex(t[4]), n_htrial=int(t[6]), hf_proc=int(t[8]))
            elif t[0] == "RESTART":
                restart = dict(n=int(t[1]), loc_norm=float.fromhex(t[2]), glob_norm=float.fromhex(t[3]), scale=float.fromhex(t[4]))
            elif t[0] in ("F", "R"):
                row = dict(it=int(t[1]), numer=float.fromhex(t[2]), denom=float.fromhex(t[3]), norm=float.fromhex(t[4]), shift=float.fromhex(t[5]),
                           nkept=int(t[6]), n_nonz=int(t[7]), curr_size=int(t[8]), num_success=int(t[9]), hash=int(t[10], 16))
                (fill if t[0] == "F" else run).append(row)
    return dict(fill=fill, run=run, restart=restart, **hdr)
