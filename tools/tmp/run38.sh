timeout 600 python -m pytest tests/test_peac_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python bench.py --config 4 --cpu-frames 0 2>&1 | tail -1 | cut -c1-330
timeout 300 python bench.py --config 4 --cpu-frames 0 2>&1 | tail -1 | cut -c1-130
