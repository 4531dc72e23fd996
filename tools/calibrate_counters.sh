#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts (run on the GPU box through gpurun): tools/micro/fetch_calib.hip under two
# separate --pmc passes; tools/summarize_calibration.py turns the two CSVs into profiles/<tag>_counter_calibration.{md,json}.
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/calib_$TAG
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/micro/fetch_calib.hip -o $OUT/fetch_calib_bench || exit 1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o c -- $OUT/fetch_calib_bench > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o c -- $OUT/fetch_calib_bench > $OUT/write.log 2>&1
rm -f $OUT/fetch_calib_bench
find $OUT -name "*counter_collection.csv" | head
