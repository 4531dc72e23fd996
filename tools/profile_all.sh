#!/bin/bash
# All rocprofv3 evidence of one round (run on the GPU box through gpurun, ~10 min): the default command (kernel trace + statistics, FETCH_SIZE /
# WRITE_SIZE passes), kernel statistics of configurations 2-5, the SQ counter groups, and the bench lines of the same build (default with its
# CPU baseline, configurations 2-5 with theirs, the 8 M-surfel run whose map exceeds the 256 MB Infinity Cache).
# Usage: tools/profile_all.sh <tag>
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
tools/profile.sh $TAG > gpurun_out/profile_$TAG.log 2>&1
# round 4: the same passes on the area-uniform map of rounds 1-3 (~6 % in view), the scene the earlier rooflines were quoted on
tools/profile.sh ${TAG}_sparse --map sparse > gpurun_out/profile_${TAG}_sparse.log 2>&1
for c in 2 3 4 5; do
  O=$R/gpurun_out/prof_${TAG}_config$c; mkdir -p $O
  (cd /tmp && TMPDIR=/tmp timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --config $c --cpu-frames 0 --no-breakdown --steps 2 --warmup 1 > $O/trace.log 2>&1; rm -f $O/trace/*kernel_trace.csv)
done
tools/pmc.sh $TAG bench.py --steps 1 --warmup 1 --passes-per-step 2 --cpu-frames 0 --no-breakdown > gpurun_out/pmc_$TAG.log 2>&1
python tools/summarize_pmc.py $TAG > gpurun_out/pmc_$TAG/sq_counters.txt 2>&1
find gpurun_out/pmc_$TAG -name "*.csv" -delete
python bench.py --io host > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
for c in 2 3 4 5; do python bench.py --config $c > gpurun_out/bench_${TAG}_config$c.json 2> gpurun_out/bench_${TAG}_config$c.err; done
python bench.py --surfels 8000000 --cpu-frames 0 --steps 5 --passes-per-step 3 > gpurun_out/bench_${TAG}_8M.json 2> gpurun_out/bench_${TAG}_8M.err
# round 4: the sparse map, the dense map in random array order, two sequences per GPU, SurfelFusion alone on the sparse map
python bench.py --map sparse --cpu-frames 0 > gpurun_out/bench_${TAG}_sparse.json 2> gpurun_out/bench_${TAG}_sparse.err
python bench.py --map-order random --cpu-frames 0 --steps 8 > gpurun_out/bench_${TAG}_random_order.json 2> gpurun_out/bench_${TAG}_random_order.err
python bench.py --sequences-per-gpu 2 --cpu-frames 0 --steps 8 --no-breakdown > gpurun_out/bench_${TAG}_two_sequences.json 2> gpurun_out/bench_${TAG}_two_sequences.err
python bench.py --config 3 --map sparse --cpu-frames 0 --steps 8 > gpurun_out/bench_${TAG}_config3_sparse.json 2> gpurun_out/bench_${TAG}_config3_sparse.err
MSL_PEAC_CLUSTER=device python bench.py --config 4 --cpu-frames 0 --steps 4 --no-breakdown > gpurun_out/bench_${TAG}_config4_device_cluster.json 2> gpurun_out/bench_${TAG}_config4_device_cluster.err
# round 5: the classic two-launch chain instead of the deferred compaction, SurfelFusion alone and the whole front end; the moving-camera regime
MSL_SF_DEFER=0 python bench.py --config 3 --cpu-frames 0 --steps 8 --no-breakdown > gpurun_out/bench_${TAG}_config3_classic.json 2> gpurun_out/bench_${TAG}_config3_classic.err
MSL_SF_DEFER=0 python bench.py --cpu-frames 0 --steps 8 --no-breakdown > gpurun_out/bench_${TAG}_classic.json 2> gpurun_out/bench_${TAG}_classic.err
python bench.py --map moving --cpu-frames 0 --steps 8 > gpurun_out/bench_${TAG}_moving.json 2> gpurun_out/bench_${TAG}_moving.err
python bench.py --config 3 --map moving --cpu-frames 0 --steps 8 > gpurun_out/bench_${TAG}_config3_moving.json 2> gpurun_out/bench_${TAG}_config3_moving.err
ls gpurun_out/prof_$TAG gpurun_out/pmc_$TAG | head -20
for f in gpurun_out/bench_$TAG*.json; do echo $f; tail -c 300 $f; echo; done
