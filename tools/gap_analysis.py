"""Gaps between consecutive kernels of the last iterations in a rocprofv3 --kernel-trace csv: where the GPU idles and for how long."""
import csv, sys, collections, re
path = sys.argv[1]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
def short(n):
    n = re.sub(r"\(.*", "", n)
    m = re.match(r"(?:void )?(?:\(anonymous namespace\)::)?(\w+)(<.*>)?", n)
    name = m.group(1) if m else n
    t = m.group(2) or "" if m else ""
    mm = re.search(r"<(\d+), (?:true|false), (\d+)>", t)
    if name == "k_fks_sweep" and mm: name += "_m" + mm.group(2)
    return name
# last N kernels = measured region; find iteration boundaries by k_prep1
names = [short(r[2]) for r in rows]
idx = [i for i, n in enumerate(names) if n == "k_prep1"]
if len(idx) < 12: print("few iterations", len(idx)); sys.exit(0)
lo, hi = idx[-11], idx[-1]          # 10 full iterations
tot_gap = 0; tot_busy = 0
gaps = collections.defaultdict(lambda: [0, 0])
for i in range(lo, hi):
    s0, e0, _ = rows[i]; s1, e1, _ = rows[i + 1]
    g = max(0, s1 - e0)
    tot_gap += g; tot_busy += e0 - s0
    key = names[i] + " -> " + names[i + 1]
    gaps[key][0] += g; gaps[key][1] += 1
n_it = 10
wall = (rows[hi][0] - rows[lo][0]) / n_it / 1000
print(f"wall {wall:.1f} us/iter  busy {tot_busy/n_it/1000:.1f}  gaps {tot_gap/n_it/1000:.1f}  launches {(hi-lo)/n_it:.1f}")
print("largest gap sources (us per iteration, count per iteration, us per occurrence):")
for k, (g, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"  {k:50s} {g/n_it/1000:7.1f} {c/n_it:5.1f} {g/c/1000:7.1f}")
busy = collections.defaultdict(lambda: [0, 0])
for i in range(lo, hi):
    busy[names[i]][0] += rows[i][1] - rows[i][0]; busy[names[i]][1] += 1
print("kernel busy time (us per iteration, calls, us per call):")
for k, (b, c) in sorted(busy.items(), key=lambda kv: -kv[1][0])[:30]:
    print(f"  {k:30s} {b/n_it/1000:7.1f} {c/n_it:5.1f} {b/c/1000:7.1f}")
