set -x
timeout 600 python -m pytest tests/test_surfel_gpu.py -x -q -m gpu 2>&1 | tail -15
timeout 200 python tools/quick_sf_bench.py 2>&1 | tail -3
timeout 300 python bench.py --steps 20 --warmup 3 --frames-per-step 32 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('F32', d['value'], d['roofline']['avg_us'] if 'avg_us' in d['roofline'] else d['roofline'], )"
timeout 300 python bench.py 2>&1 | tail -1 | cut -c1-900
