timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/profile_all.sh r02 > gpurun_out/profile_all.log 2>&1
for c in 2 3 4 5; do timeout 300 python bench.py --config $c --cpu-frames 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config $c', d['value'], d['unit'], d['ms_per_step'], d['roofline'].get('avg_launch_us'), d['roofline'].get('frac'))"; done
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 1500 gpurun_out/bench_default.json
