#!/bin/bash
# A/B of the default bench line between kernel-variant libraries (scratch/libmsl_<name>.so from tools/build_variant.sh; "full" = the product library),
# alternating on one box.  Usage: tools/ab_lib.sh name1 name2 ... [-- bench args]
NAMES=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do NAMES+=("$1"); shift; done; [ "$1" == "--" ] && shift
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
for v in "${NAMES[@]}"; do
  L=$R/scratch/libmsl_$v.so; [ $v == full ] && L=$R/manhattanslam_amd/libmsl.so
  MSL_LIB=$L timeout 300 python bench.py --cpu-frames 0 --no-breakdown --no-parity-gate --steps 8 "$@" 2> gpurun_out/ab_lib_$v.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$v', 'value', d['value'], 'frac', r['frac'], 'k_fuse us', r['avg_launch_us'], 'raw', r['avg_launch_us_event_pair_raw'])"
done
