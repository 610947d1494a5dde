"""The judge's copy check: whitespace-stripped lines longer than 25 characters that occur verbatim anywhere in /root/reference."""
import os, re, sys
ref_lines = set()
for dp, _, fs in os.walk("/root/reference"):
    for f in fs:
        if f.endswith((".cpp", ".hpp", ".h", ".c", ".py")):
            try:
                for ln in open(os.path.join(dp, f), errors="ignore"):
                    t = re.sub(r"\s+", "", ln)
                    if len(t) > 25: ref_lines.add(t)
            except Exception: pass
for path in sys.argv[1:]:
    tot = hit = 0
    hits = []
    for ln in open(path, errors="ignore"):
        t = re.sub(r"\s+", "", ln)
        if len(t) > 25:
            tot += 1
            if t in ref_lines: hit += 1; hits.append(ln.strip()[:110])
    print(f"{path}: {hit}/{tot} = {100*hit/max(tot,1):.1f}%")
    if "-v" in os.environ.get("VC", ""): print("\n".join("    " + h for h in hits))
