#!/bin/bash
# A/B of the default bench line (and optionally --config 3) under several settings of one environment variable, on one box.
# Usage: tools/ab_env.sh VAR v1 v2 ... [-- bench args]
VAR=$1; shift
VALS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do VALS+=("$1"); shift; done; [ "$1" == "--" ] && shift
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
for v in "${VALS[@]}"; do
  env $VAR=$v timeout 300 python bench.py --cpu-frames 0 --no-breakdown --no-parity-gate --steps 8 "$@" > gpurun_out/ab_${VAR}_$v.json 2> gpurun_out/ab_${VAR}_$v.err
  python -c "
import json; d=json.load(open('gpurun_out/ab_${VAR}_$v.json')); r=d['roofline']; print('$VAR=$v', 'value', d['value'], 'frac', r['frac'], 'k_fuse us', r['avg_launch_us'], 'raw', r['avg_launch_us_event_pair_raw'])"
done
