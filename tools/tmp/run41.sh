cp tools/tmp/bench_old.py ./bench_old.py
timeout 300 python bench_old.py --config 4 --cpu-frames 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('old', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d.get('kernel_breakdown'))"
timeout 300 python bench.py --config 4 --cpu-frames 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d.get('kernel_breakdown'))"
MSL_PEAC_TIMING=1 timeout 300 python bench.py --config 4 --cpu-frames 0 2>&1 | grep "batch of" | awk '{print $NF, $(NF-1)}' | sort | uniq -c | sort -rn | head -5
MSL_PEAC_TIMING=1 timeout 300 python bench.py --config 4 --cpu-frames 0 2>&1 | grep "batch of" | awk '{print $(NF-1)}' | tr '\n' ' '
rm bench_old.py
