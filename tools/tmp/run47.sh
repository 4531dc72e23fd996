python tools/tmp/peac_loop.py 2>&1 | grep -v amdgpu
timeout 300 python bench.py --config 4 --cpu-frames 0 2>&1 | tail -1 | cut -c1-130
timeout 300 python bench.py --config 4 --cpu-frames 0 2>&1 | tail -1 | cut -c1-130
