# L2 (TCC) / vector-cache (TCP) request counters of the surfel kernels on bench.py --config 3: three rocprofv3 --pmc passes (kernel trace only), per-launch
# averages on stdout (profiles/r05_cache_counters.txt).  Run on the GPU box through gpurun.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r06}
OUT=$R/gpurun_out/tcc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
while read -r GROUP; do
  [ -z "$GROUP" ] && continue
  i=$((i+1))
  timeout 70 rocprofv3 --pmc $GROUP --kernel-trace --output-format csv -d $OUT/g$i -o p -- python $R/bench.py --config 3 --cpu-frames 0 --no-breakdown --no-parity-gate --steps 1 --warmup 1 --passes-per-step 1 > $OUT/g$i.log 2>&1
  echo "group $i rc=$?"
done <<'EOG'
TCC_REQ TCC_HIT TCC_MISS TCC_READ
TCC_WRITE TCC_EA_RDREQ TCC_EA_WRREQ TCC_TAG_STALL
TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TCP_PENDING_STALL_CYCLES
EOG
python3 - <<P
import csv, glob, re, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"\b(kb?_\w+)", r["Kernel_Name"])
        if not m: continue
        a = acc[m.group(1)][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in ("k_fuse", "k_compact", "kb_seed_plane", "kb_update_seeds", "kb_assign"):
    if k in acc: print(k, {c: round(v[0] / v[1]) for c, v in sorted(acc[k].items())})
P
find $OUT -name "*.csv" -delete
