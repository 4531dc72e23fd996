#!/bin/bash
# rocprofv3 --kernel-trace --stats of a short bench.py run under two settings of an environment variable (A/B on one box); prints the average
# duration of the map-stage kernels.  Usage: tools/kstat_ab.sh VAR valueA valueB [bench.py args]      (default args: --config 3)
VAR=${1:-MSL_SF_DEAL}; A=${2:-1}; B=${3:-0}; shift 3
ARGS=${@:---config 3}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/kstat_ab; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in $A $B; do
  rm -rf $OUT/v$v
  env $VAR=$v timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/v$v -o p -- python $R/bench.py $ARGS --cpu-frames 0 --no-breakdown --no-parity-gate --steps 2 --warmup 1 --passes-per-step 2 > $OUT/v$v.log 2>&1
  rm -f $OUT/v$v/*kernel_trace.csv $OUT/v$v/*/*kernel_trace.csv
  python3 - <<P
import csv, glob, re
for f in glob.glob("$OUT/v$v/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"\b(kb?_\w+)", r["Name"])
        if m and m.group(1) in ("k_fuse", "k_compact", "k_deal", "k_replay", "k_defer_tail", "kb_seed_plane", "kb_assign", "kb_update_seeds", "k_empty", "k_fast"):
            print("$VAR=$v", m.group(1), "calls", r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 2), "min", round(float(r["MinNs"]) / 1e3, 2), "max", round(float(r["MaxNs"]) / 1e3, 2))
P
  tail -1 $OUT/v$v.log | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('$VAR=$v value', d['value'], 'frac', d['roofline']['frac'], 'k_fuse event us', d['roofline']['avg_launch_us'])
except Exception as e: print('no json', e)"
done
