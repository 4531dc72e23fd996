#!/bin/bash
# rocprofv3 passes for bench.py (run on the GPU box through gpurun).  Usage: tools/profile.sh <tag> [bench args]
#   trace : --kernel-trace --stats of `bench.py --cpu-frames 0 --no-breakdown --no-parity-gate --steps 3 --warmup 1 [args]` (the stationary workload of the default
#           command: every pass starts from the same map, so 3 steps show the same kernels as 20)
#   pmc   : FETCH_SIZE and WRITE_SIZE in separate passes (never combined with sys / hip trace domains) on --steps 1 --warmup 1 --passes-per-step 2
TAG=${1:-r03}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--cpu-frames 0 --no-breakdown --no-parity-gate $@"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py $ARGS --steps 3 --warmup 1 > $OUT/trace.log 2>&1
rm -f $OUT/trace/*kernel_trace.csv      # hundreds of MB of per-dispatch rows; the statistics are what is kept
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o bench -- python $R/bench.py $ARGS --steps 1 --warmup 1 --passes-per-step 2 > $OUT/pmc_fetch.log 2>&1
rm -f $OUT/pmc_fetch/*kernel_trace.csv
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o bench -- python $R/bench.py $ARGS --steps 1 --warmup 1 --passes-per-step 2 > $OUT/pmc_write.log 2>&1
rm -f $OUT/pmc_write/*kernel_trace.csv
python3 - <<P
# the per-dispatch counter rows are large too: keep per-kernel averages only
import collections, csv, glob, json, re
for sub, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob("$OUT/%s/*counter_collection.csv" % sub):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == ctr:
                m = re.search(r"\b(kb?_\w+)", r["Kernel_Name"])
                a = agg[m.group(1) if m else r["Kernel_Name"].split("(")[0][-48:]]
                a[0] += 1; a[1] += float(r["Counter_Value"])
    json.dump({k: {"launches": n, "avg_KiB_per_launch": v / n} for k, (n, v) in agg.items()}, open("$OUT/%s.json" % sub, "w"), indent=1)
P
rm -f $OUT/pmc_fetch/*counter_collection.csv $OUT/pmc_write/*counter_collection.csv
tail -1 $OUT/trace.log | cut -c1-400
find $OUT -type f | head -30
