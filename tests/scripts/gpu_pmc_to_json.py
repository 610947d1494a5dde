"""Ad-hoc: merges the two rocprofv3 PMC passes of bench.py (--pmc FETCH_SIZE, --pmc WRITE_SIZE) into profiles/rNN_pmc_traffic.json.
usage: gpu_pmc_to_json.py <fetch pass dir> <write pass dir> <out.json>"""
import csv, sys, collections, glob, json


def summarise(d, frac=0.2):
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(fn)):
            k = row["Kernel_Name"].split("(")[0][:60].replace("void ", "")
            rows[k][row["Counter_Name"]].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
    out = {}
    for k in rows:
        for c, lst in rows[k].items():
            lst.sort()
            tail = lst[int(len(lst) * (1 - frac)):]
            out.setdefault(k, {})[c] = (sum(v for _, v in tail) / len(tail), len(tail))
    return out


f, w = summarise(sys.argv[1]), summarise(sys.argv[2])
ker = {}
for k in sorted(set(f) | set(w)):
    fs = f.get(k, {}).get("FETCH_SIZE", (0.0, 0))
    ws = w.get(k, {}).get("WRITE_SIZE", (0.0, 0))
    ker[k] = {"FETCH_SIZE_KB": round(fs[0], 1), "WRITE_SIZE_KB": round(ws[0], 1), "bytes_per_launch": int(2 * fs[0] * 1024 + ws[0] * 1024),
              "dispatches_averaged": max(fs[1], ws[1])}
doc = {"_about": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) on `python3 bench.py --steps 6 --warmup 2 --cpu-iters 0 "
                 "--profile-steps 1`, m = 1e6, one MI355X; mean over the last 20 % of each kernel's dispatches (the full-size measured part). Units: KB as "
                 "reported. bytes_per_launch = 2 * FETCH_SIZE * 1024 (gfx950 correction of MI355X_MICROARCH.md, calibrated there for 16 B/lane streams only; "
                 "these kernels load 4-8 B per lane, so the read half is an upper estimate) + WRITE_SIZE * 1024.", "kernels": ker}
json.dump(doc, open(sys.argv[3], "w"), indent=0, sort_keys=True)
print("kernels:", len(ker), "k_fks_sweep entries:", {k: v["bytes_per_launch"] for k, v in ker.items() if "k_fks_sweep" in k})
