#!/bin/bash
# All rocprofv3 evidence of one round (run on the GPU box through gpurun): kernel trace + stats, FETCH_SIZE / WRITE_SIZE passes (separate
# --pmc runs, never combined with the sys / hip trace domains), two SQ counter groups, and the bench line of the same build.
# Usage: tools/profile_all.sh <tag>
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
tools/profile.sh $TAG > gpurun_out/profile_$TAG.log 2>&1
tools/pmc.sh $TAG bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-breakdown > gpurun_out/pmc_$TAG.log 2>&1
cd $R && python bench.py --cpu-frames 0 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
ls gpurun_out/prof_$TAG gpurun_out/pmc_$TAG | head -20
tail -c 600 gpurun_out/bench_$TAG.json
