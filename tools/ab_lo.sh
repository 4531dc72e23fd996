#!/bin/bash
# Round-6 A/B of kernel-variant libraries (scratch/libmsl_<name>.so; first used for the k_fuse phase-B variants): parity on the map tests, the default bench line, and rocprofv3 kernel statistics
# per variant on one box.   tools/ab_lo.sh name1 name2 ...
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
for v in "$@"; do
  L=$R/scratch/libmsl_$v.so; [ $v == full ] && L=$R/manhattanslam_amd/libmsl.so
  echo "== $v"
  if [ "${AB_PARITY:-1}" == "1" ]; then
    MSL_LIB=$L timeout 900 python -m pytest tests/test_surfel_gpu.py tests/test_properties_gpu.py tests/test_golden_gpu.py -q -m gpu -x 2>&1 | tail -2
  fi
  MSL_LIB=$L timeout 300 python bench.py --cpu-frames 0 --no-breakdown --no-parity-gate --steps 8 2> gpurun_out/ab_lo_$v.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$v', 'value', d['value'], 'frac', r['frac'], 'k_fuse us', r['avg_launch_us'], 'raw', r['avg_launch_us_event_pair_raw'], 'surfel_only', d.get('surfel_only_keyframes_per_sec'))"
  OUT=$R/gpurun_out/ab_lo_prof/$v; rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && export TMPDIR=/tmp && MSL_LIB=$L timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python $R/bench.py --cpu-frames 0 --no-breakdown --no-parity-gate --steps 2 --warmup 1 > $OUT.log 2>&1)
  rm -f $OUT/*kernel_trace.csv $OUT/*/*kernel_trace.csv
  python3 - <<P
import csv, glob, re
for f in glob.glob("$OUT/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"\b(kb?_\w+)", r["Name"])
        if m and m.group(1) in ("k_fuse", "k_compact", "k_empty", "kb_seed_plane", "kb_assign", "kb_update_seeds"):
            print("$v rocprof", m.group(1), "calls", r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 2), "min", round(float(r["MinNs"]) / 1e3, 2))
P
  if [ "${AB_ISO:-1}" == "1" ]; then MSL_LIB=$L timeout 200 python tools/fuse_iso.py 3 2>/dev/null | tail -1; fi
  if [ "${AB_CONFIG3:-0}" == "1" ]; then MSL_LIB=$L timeout 300 python bench.py --config 3 --cpu-frames 0 --no-breakdown --no-parity-gate --steps 8 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'config3 value', d['value'])"; fi
done
