#!/bin/bash
# Hardware-counter passes (one rocprofv3 run per counter group; never combined with sys/hip traces).  Only SQ counters:
# a TCP/TCC group once ran into the box's time limit.  Each pass is bounded by `timeout`.
# Usage: tools/pmc.sh <tag> <python script + args>      -> gpurun_out/pmc_<tag>/g<i>/...counter_collection.csv
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
i=0
while read -r GROUP; do
  [ -z "$GROUP" ] && continue
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $GROUP --kernel-trace --output-format csv -d $OUT/g$i -o p -- python $R/$@ > $OUT/g$i.log 2>&1
done <<'EOG'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS
SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_LDS
EOG
find $OUT -name "*counter_collection.csv" | head
