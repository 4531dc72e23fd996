"""Per-kernel averages of the counters collected by tools/pmc.sh + derived VALU utilisation.
Usage: python tools/summarize_pmc.py <tag> [kernel substr ...]"""
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
print(f"{'kernel':18s} {'waves':>8s} {'VALU/wave':>10s} {'SALU/wave':>10s} {'LDS/wave':>9s} {'VMEM/wave':>10s} {'busy us':>8s} {'VALU util':>9s} {'bankconf%':>9s}")
for k in sorted(acc, key=lambda k: -acc[k].get('SQ_BUSY_CYCLES', [0, 1])[0] / max(acc[k].get('SQ_BUSY_CYCLES', [0, 1])[1], 1)):
    g = lambda c: acc[k][c][0] / acc[k][c][1] if c in acc[k] and acc[k][c][1] else float('nan')
    waves = g('SQ_WAVES'); busy = g('SQ_BUSY_CYCLES') / 32          # summed over 32 shader engines
    util = g('SQ_INSTS_VALU') * 4 / (busy * 1024) if busy == busy else float('nan')   # 4 cycles per wave64 VALU op, 1024 SIMDs
    print(f"{k:18s} {waves:8.0f} {g('SQ_INSTS_VALU')/waves:10.0f} {g('SQ_INSTS_SALU')/waves:10.0f} {g('SQ_INSTS_LDS')/waves:9.0f} "
          f"{(g('SQ_INSTS_VMEM_RD')+g('SQ_INSTS_VMEM_WR'))/waves:10.1f} {busy/2400:8.1f} {util:9.2f} {100*g('SQ_LDS_BANK_CONFLICT')/max(g('SQ_ACTIVE_INST_LDS'),1):9.1f}")
