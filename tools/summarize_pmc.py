"""Per-kernel averages of the counters collected by tools/pmc.sh.  Usage: python tools/summarize_pmc.py <tag> [kernel substr ...]"""
import csv, glob, re, sys, collections
tag = sys.argv[1]; filt = sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(f"gpurun_out/pmc_{tag}/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"\b(kb?_\w+)", r["Kernel_Name"])
        if not m: continue
        k = m.group(1)
        if filt and not any(s in k for s in filt): continue
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        s, n = acc[k][c]
        print(f"   {c:42s} {s / n:16.1f}   (n={n})")
