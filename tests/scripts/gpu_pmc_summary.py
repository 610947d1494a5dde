"""Ad-hoc: summarise a rocprofv3 --pmc counter_collection.csv per kernel name.  Only the LAST `frac` of each kernel's
dispatches is averaged (the measured, full-size part of a bench.py run; the filler run before it launches smaller grids)."""
import csv, sys, collections, glob, json
files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in files:
    for row in csv.DictReader(open(fn)):
        k = row["Kernel_Name"].split("(")[0][:60].replace("void ", "")
        rows[k][row["Counter_Name"]].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
out = {}
for k in rows:
    out[k] = {}
    for c, lst in rows[k].items():
        lst.sort()
        tail = lst[int(len(lst) * (1 - frac)):]
        out[k][c] = {"mean": sum(v for _, v in tail) / len(tail), "n": len(tail)}
print(json.dumps(out, indent=1, sort_keys=True))
