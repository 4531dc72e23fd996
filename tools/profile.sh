#!/bin/bash
# rocprofv3 passes for bench.py (run on the GPU box through gpurun). Usage: tools/profile.sh <tag> [bench args]
TAG=${1:-r01}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the default bench run (20 steps x 256 frames after 3 warm-up steps), so the kernel statistics are those of the command bench.py's roofline is quoted on
ARGS="--cpu-frames 0 --no-breakdown $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o bench -- python $R/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o bench -- python $R/bench.py $ARGS > $OUT/pmc_write.log 2>&1
find $OUT -type f | head -30
tail -2 $OUT/trace.log
