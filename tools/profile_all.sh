#!/bin/bash
# All rocprofv3 evidence of one round (run on the GPU box through gpurun, ~17 min): the default command (kernel trace + statistics, FETCH_SIZE /
# WRITE_SIZE passes), kernel statistics of configuration 3, the SQ counter groups, and the bench lines of the same build (default with its
# CPU baseline, configurations 2-5 with theirs, the 8 M-surfel run whose map exceeds the 256 MB Infinity Cache, the moving-camera regime, the
# deferred compaction against the classic chain).
# Usage: tools/profile_all.sh <tag>
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
tools/profile.sh $TAG > gpurun_out/profile_$TAG.log 2>&1
for c in 3; do
  O=$R/gpurun_out/prof_${TAG}_config$c; mkdir -p $O
  (cd /tmp && TMPDIR=/tmp timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --config $c --cpu-frames 0 --no-breakdown --no-parity-gate --steps 2 --warmup 1 > $O/trace.log 2>&1; rm -f $O/trace/*kernel_trace.csv)
done
tools/pmc.sh $TAG bench.py --steps 1 --warmup 1 --passes-per-step 2 --cpu-frames 0 --no-breakdown --no-parity-gate > gpurun_out/pmc_$TAG.log 2>&1
python tools/summarize_pmc.py $TAG > gpurun_out/pmc_$TAG/sq_counters.txt 2>&1
find gpurun_out/pmc_$TAG -name "*.csv" -delete
python bench.py --io host > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
for c in 2 3 4 5; do timeout 400 python bench.py --config $c > gpurun_out/bench_${TAG}_config$c.json 2> gpurun_out/bench_${TAG}_config$c.err; done
timeout 300 python bench.py --surfels 8000000 --cpu-frames 0 --steps 5 --passes-per-step 3 > gpurun_out/bench_${TAG}_8M.json 2> gpurun_out/bench_${TAG}_8M.err
# round 5: the moving-camera regime; the deferred compaction forced on (the handle's own two streams keep the classic chain by default) and,
# on ONE stream -- where the map chain is the critical path -- deferred against classic (tools/fuse_iso.py)
timeout 300 python bench.py --map moving --cpu-frames 0 --steps 8 > gpurun_out/bench_${TAG}_moving.json 2> gpurun_out/bench_${TAG}_moving.err
MSL_SF_DEFER=1 timeout 300 python bench.py --cpu-frames 0 --steps 8 --no-breakdown > gpurun_out/bench_${TAG}_deferred.json 2> gpurun_out/bench_${TAG}_deferred.err
MSL_SF_DEFER=1 timeout 300 python bench.py --config 3 --cpu-frames 0 --steps 8 --no-breakdown > gpurun_out/bench_${TAG}_config3_deferred.json 2> gpurun_out/bench_${TAG}_config3_deferred.err
for d in 0 1; do MSL_SF_DEFER=$d timeout 200 python tools/fuse_iso.py 3 2>/dev/null | tail -1 > gpurun_out/fuse_iso_${TAG}_defer$d.json; done
# round 6: the screen-position dealing of k_fuse's sub-blocks against array order (fabric traffic and L2 counters, A/B on this box), the L2 / vector-cache
# request counters of the surfel kernels, two sequences per GPU (bench line + SQ counters of that shape), the random-order map
bash tools/fetch_ab.sh MSL_SF_DEAL 1 0 > gpurun_out/fetch_ab_$TAG.txt 2>&1
bash tools/cache_counters.sh $TAG > gpurun_out/cache_counters_$TAG.txt 2>&1
timeout 300 python bench.py --sequences-per-gpu 2 --cpu-frames 0 --steps 8 --no-breakdown > gpurun_out/bench_${TAG}_two_sequences.json 2> gpurun_out/bench_${TAG}_two_sequences.err
tools/pmc.sh ${TAG}_2seq bench.py --sequences-per-gpu 2 --steps 1 --warmup 1 --passes-per-step 2 --cpu-frames 0 --no-breakdown --no-parity-gate > gpurun_out/pmc_${TAG}_2seq.log 2>&1
python tools/summarize_pmc.py ${TAG}_2seq > gpurun_out/pmc_${TAG}_2seq/sq_counters.txt 2>&1
find gpurun_out/pmc_${TAG}_2seq -name "*.csv" -delete
timeout 300 python bench.py --map-order random --cpu-frames 0 --steps 8 --no-breakdown > gpurun_out/bench_${TAG}_random_order.json 2> gpurun_out/bench_${TAG}_random_order.err
bash tools/kstat_ab.sh MSL_SF_DEAL 1 0 > gpurun_out/kstat_deal_$TAG.txt 2>&1
ls gpurun_out/prof_$TAG gpurun_out/pmc_$TAG | head -20
for f in gpurun_out/bench_$TAG*.json; do echo $f; tail -c 300 $f; echo; done
