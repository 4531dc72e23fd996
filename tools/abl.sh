#!/bin/bash
# k_fuse ablation timings (analysis builds scratch/libmsl_abl<n>.so, -DMSL_ABL=<n>): one stream, event pair minus the empty-kernel pair
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for v in "$@"; do
  L=$R/scratch/libmsl_$v.so; [ $v == full ] && L=$R/manhattanslam_amd/libmsl.so
  MSL_LIB=$L MSL_SF_DEFER=0 FUSE_ISO_ONLY=one timeout 120 python tools/fuse_iso.py 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); o=d['one_stream']; print('$v', 'k_fuse', round(o['k_fuse_us']-o['k_empty_us'],2), 'raw', o['k_fuse_us'], 'compact', o['k_compact_us'], 'kf/s', o['kf_per_s'], 'upd', o['ctr']['n_updated'])"
done
