timeout 600 python -m pytest tests/test_peac_gpu.py -x -q -m gpu 2>&1 | tail -8
MSL_PEAC_TIMING=1 python tools/tmp/peac_loop.py 2>&1 | grep -v amdgpu | grep "alone\|batch of" | tail -4
timeout 300 python bench.py --config 4 --cpu-frames 0 2>&1 | tail -1 | cut -c1-130
