"""Device-idle gaps and longest kernels of a rocprofv3 --kernel-trace run (CSV output), used for profiles/r05_idle_gap_trace.txt.
Usage: python tools/gap_trace.py <rocprofv3 output directory>"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
print("kernels", len(rows))
# long kernels
long = sorted(rows, key=lambda r: r[1] - r[0], reverse=True)[:8]
for s, e, n, q in long: print("long", round((e - s) / 1e6, 3), "ms", n, q, "t=", round((s - rows[0][0]) / 1e6, 1))
# device-idle gaps: sweep over end times
cur_end = rows[0][1]
gaps = []
for i, (s, e, n, q) in enumerate(rows[1:], 1):
    if s > cur_end + 5_000_000: gaps.append((s - cur_end, cur_end, i))
    cur_end = max(cur_end, e)
for g, at, i in gaps[:20]:
    print("gap", round(g / 1e6, 2), "ms at t=", round((at - rows[0][0]) / 1e6, 1), "before", rows[i][2], "after", rows[i - 1][2])
