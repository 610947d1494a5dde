"""Ad-hoc: summarise a rocprofv3 --pmc counter_collection.csv per kernel name (mean per dispatch)."""
import csv, sys, collections, glob
files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for fn in files:
    for row in csv.DictReader(open(fn)):
        k = row["Kernel_Name"].split("(")[0][:60]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
want = sys.argv[2:] or None
for k in sorted(agg, key=lambda k: -sum(agg[k].values())):
    if want and not any(w in k for w in want):
        continue
    print(k, {c: round(agg[k][c] / cnt[k][c], 1) for c in sorted(agg[k])}, "n=", max(cnt[k].values()))
