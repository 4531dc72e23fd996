timeout 600 python -m pytest tests/test_peac_gpu.py -x -q -m gpu 2>&1 | tail -2
timeout 300 python bench.py --config 4 --cpu-frames 0 2>&1 | tail -1 | cut -c1-130
MSL_PEAC_TIMING=1 timeout 300 python bench.py --config 4 --cpu-frames 0 --steps 4 --warmup 2 2>&1 | grep "batch of" | tail -4
