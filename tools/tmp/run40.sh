BENCH_DEBUG=1 timeout 300 python bench.py --config 4 --cpu-frames 0 --steps 6 --warmup 2 2>&1 | grep "dbg" | tail -6
