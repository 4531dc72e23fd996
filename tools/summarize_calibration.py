#!/usr/bin/env python3
"""gpurun_out/calib_<tag>/ -> profiles/<tag>_counter_calibration.{md,json}: known bytes / counter bytes per access pattern."""
import collections
import csv
import glob
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
MiB = 1 << 20
known = {"calib_x16": 64 * MiB, "calib_x4": 64 * MiB, "calib_x1": 16 * MiB, "calib_s80": ((64 * MiB // 16) // 320) * 320 * 16, "calib_g8": 350000 * 8,
         "calib_w16": 64 * MiB, "calib_w4": 64 * MiB, "calib_w1": 16 * MiB}
what = {"calib_x16": "16 B/lane coalesced loads", "calib_x4": "4 B/lane coalesced loads", "calib_x1": "1 B/lane coalesced loads",
        "calib_s80": "5 x 16 B per lane at an 80-B lane stride (k_fuse hot records)", "calib_g8": "8-B gathers at random positions of a 2.4 MB table (useful bytes)",
        "calib_w16": "16 B/lane stores", "calib_w4": "4 B/lane stores", "calib_w1": "1 B/lane stores"}
out = {}
for sub, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/calib_{tag}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == ctr:
                k = r["Kernel_Name"].split("(")[0].strip()
                agg[k].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if k not in known:
            continue
        if (ctr == "FETCH_SIZE") != (not k.startswith("calib_w")):
            continue
        kib = sum(v[1:]) / max(len(v) - 1, 1) if len(v) > 1 else v[0]      # first launch of a read pattern may hit a cold cache: skip it
        out[k] = {"pattern": what[k], "known_bytes": known[k], "counter": ctr, "counter_KiB": round(kib, 1), "factor": round(known[k] / (kib * 1024), 3) if kib else None}
json.dump(out, open(f"profiles/{tag}_counter_calibration.json", "w"), indent=1)
with open(f"profiles/{tag}_counter_calibration.md", "w") as f:
    f.write(f"# FETCH_SIZE / WRITE_SIZE calibration `{tag}` (tools/micro/fetch_calib.hip under rocprofv3 --pmc, one counter per pass)\n\n")
    f.write("factor = known bytes / (counter KiB x 1024): multiply a kernel's raw counter by the factor of its access pattern.\n\n")
    f.write("| kernel | pattern | known bytes | counter | KiB per launch | factor |\n|---|---|---|---|---|---|\n")
    for k, e in out.items():
        f.write(f"| {k} | {e['pattern']} | {e['known_bytes']} | {e['counter']} | {e['counter_KiB']} | {e['factor']} |\n")
print(open(f"profiles/{tag}_counter_calibration.md").read())
