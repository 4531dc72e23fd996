#!/bin/bash
# A/B of k_fuse's fabric traffic: FETCH_SIZE / WRITE_SIZE and the TCC request counters of `bench.py --config 3` (one pass) under two settings of an
# environment variable.  Usage (on the GPU box through gpurun): tools/fetch_ab.sh VAR valueA valueB      -> gpurun_out/fetch_ab.txt
VAR=${1:-MSL_SF_DEAL}; A=${2:-1}; B=${3:-0}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/fetch_ab; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in $A $B; do
  i=0
  while read -r GROUP; do
    [ -z "$GROUP" ] && continue
    i=$((i+1))
    env $VAR=$v timeout 100 rocprofv3 --pmc $GROUP --kernel-trace --output-format csv -d $OUT/v$v/g$i -o p -- python $R/bench.py --config 3 --cpu-frames 0 --no-breakdown --no-parity-gate --steps 1 --warmup 1 --passes-per-step 1 > $OUT/v${v}_g$i.log 2>&1
  done <<'EOG'
FETCH_SIZE
WRITE_SIZE
TCC_REQ TCC_HIT TCC_MISS TCC_READ
EOG
done
python3 - <<P > $R/gpurun_out/fetch_ab.txt
import csv, glob, re, collections
for v in ("$A", "$B"):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob("$OUT/v%s/**/*counter_collection.csv" % v, recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"\b(kb?_\w+)", r["Kernel_Name"])
            if not m: continue
            a = acc[m.group(1)][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k in ("k_fuse", "k_compact", "k_deal"):
        if k in acc: print("$VAR=%s" % v, k, {c: round(x[0] / x[1], 1) for c, x in sorted(acc[k].items())})
P
find $OUT -name "*.csv" -delete
cat $R/gpurun_out/fetch_ab.txt
