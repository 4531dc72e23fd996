#!/bin/bash
# Everything a maintainer wants to see on an MI355X box after a change (run through gpurun or on the box itself, from the repo root):
# the GPU parity suite, the smoke test, and one bench line per BASELINE configuration.  Output goes to gpurun_out/.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" && mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/gpu_tests.log 2>&1; tail -1 gpurun_out/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
for c in frontend 2 3 4 5; do
    timeout 600 python bench.py --config $c --cpu-frames 0 2> gpurun_out/bench_$c.err | tail -1 > gpurun_out/bench_$c.json
    python -c "import json,sys; d=json.load(open('gpurun_out/bench_$c.json')); print('config $c:', d['value'], d['unit'], 'roofline', d['roofline'].get('frac'))"
done
