#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (rocprofv3 --kernel-trace --stats and the two --pmc passes of tools/profile.sh) into profiles/<name>_*.

    python tools/summarize_profile.py <tag> [<name>]        name defaults to tag; e.g. r03_config2 -> profiles/r03_config2_summary.json

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB per dispatch.  Calibrated on known byte counts in this library's access shapes
(profiles/r03_counter_calibration.md, tools/micro/fetch_calib.hip): FETCH_SIZE reports exactly half of the bytes of coalesced reads of 1, 4 and
16 B per lane and of k_fuse's 80-B-stride record loads (factor 2.0, as MI355X_MICROARCH.md says for wide streams), WRITE_SIZE is exact (factor 1.0).
"""
import csv
import hashlib
import json
import os
import re
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
name = sys.argv[2] if len(sys.argv) > 2 else tag
src = f"gpurun_out/prof_{tag}"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs("profiles", exist_ok=True)


def kname(s):
    m = re.search(r"\b(kb?_\w+)", s)
    return m.group(1) if m else s.split("(")[0][-48:]


def kernel_source_hash():   # = bench.py's: a PMC summary is quoted only for the kernel sources it was measured on
    h = hashlib.sha256()
    d = os.path.join(ROOT, "manhattanslam_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


stats = list(csv.DictReader(open(f"{src}/trace/bench_kernel_stats.csv")))
shutil.copy(f"{src}/trace/bench_kernel_stats.csv", f"profiles/{name}_kernel_stats.csv")
pmc = {}
for sub, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    p = f"{src}/{sub}.json"
    if os.path.exists(p):
        pmc[ctr] = json.load(open(p))
summary = {}
for r in stats:
    k = kname(r["Name"])
    e = summary.setdefault(k, {"calls": 0, "total_ns": 0})
    e["calls"] += int(r["Calls"]); e["total_ns"] += int(r["TotalDurationNs"])
for k, e in summary.items():
    e["avg_us"] = round(e["total_ns"] / e["calls"] / 1e3, 2)
    f = pmc.get("FETCH_SIZE", {}).get(k); w = pmc.get("WRITE_SIZE", {}).get(k)
    if f:
        e["fetch_KiB_raw"] = round(f["avg_KiB_per_launch"], 1)
        e["fetch_bytes_corrected"] = int(2 * 1024 * f["avg_KiB_per_launch"])
    if w:
        e["write_bytes"] = int(1024 * w["avg_KiB_per_launch"])
tot = sum(e["total_ns"] for e in summary.values())
for e in summary.values():
    e["pct"] = round(100.0 * e["total_ns"] / tot, 2)
cmd = ""
if os.path.exists(f"{src}/trace.log"):
    cmd = open(f"{src}/trace.log").read()[-3000:]
line = None
for ln in reversed(cmd.splitlines()):
    if ln.startswith("{") and '"metric"' in ln:
        line = json.loads(ln)
        break
doc = dict(sorted(summary.items(), key=lambda kv: -kv[1]["total_ns"]))
doc["_meta"] = {"kernel_source_hash": kernel_source_hash(), "tag": tag,
                "bench_line_under_rocprofv3": {k: line[k] for k in ("value", "roofline", "config")} if line else None}
json.dump(doc, open(f"profiles/{name}_summary.json", "w"), indent=1)
with open(f"profiles/{name}_summary.md", "w") as f:
    f.write(f"# rocprofv3 summary `{name}` (tools/profile.sh: bench.py --cpu-frames 0 --no-breakdown --steps 3 --warmup 1, kernel trace + statistics; "
            "FETCH_SIZE / WRITE_SIZE from separate --pmc passes)\n\n")
    if line:
        r = line["roofline"]
        f.write(f"bench.py's own line inside the traced run: value {line['value']} frames/s, {r['kernel']} {r['avg_launch_us']} us per launch by the HIP events "
                f"carried by its dispatch (frac {r['frac']}).\n\n")
    f.write("| kernel | calls | avg us | % GPU time | FETCH_SIZE KiB/launch (raw) | HBM read B/launch (x2, calibrated) | WRITE_SIZE B/launch (x1, calibrated) |\n|---|---|---|---|---|---|---|\n")
    for k, e in sorted(summary.items(), key=lambda kv: -kv[1]["total_ns"]):
        f.write(f"| {k} | {e['calls']} | {e['avg_us']} | {e['pct']} | {e.get('fetch_KiB_raw', '')} | {e.get('fetch_bytes_corrected', '')} | {e.get('write_bytes', '')} |\n")
print(open(f"profiles/{name}_summary.md").read())
