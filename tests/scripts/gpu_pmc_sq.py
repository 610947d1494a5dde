"""Ad-hoc: per-kernel means of the SQ counters of one or more rocprofv3 --pmc passes of bench.py (tail 20 % of each kernel's dispatches).
usage: gpu_pmc_sq.py <out.txt> <pass dir> [<pass dir> ...]"""
import csv, sys, collections, glob

rows = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[2:]:
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(fn)):
            k = row["Kernel_Name"].replace("void ", "")
            k = k.split("(")[0][:64]
            rows[k][row["Counter_Name"]].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
names = sorted({c for k in rows for c in rows[k]})
with open(sys.argv[1], "w") as out:
    out.write("kernel " + " ".join(names) + "\n")
    for k in sorted(rows):
        vals = []
        for c in names:
            lst = sorted(rows[k].get(c, []))
            tail = lst[int(len(lst) * 0.8):] or lst
            vals.append("%.4g" % (sum(v for _, v in tail) / len(tail)) if tail else "-")
        out.write(k + " " + " ".join(vals) + "\n")
print(open(sys.argv[1]).read()[:200])
