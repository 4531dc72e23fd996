timeout 900 python -m pytest tests/test_orb_gpu.py -x -q -m gpu 2>&1 | tail -4
for m in fused levels fused levels; do MSL_ORB_PYRAMID=$m timeout 300 python bench.py --config 2 --cpu-frames 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config 2 $m', d['value'], d['ms_per_step'])"; done
for m in fused levels; do MSL_ORB_PYRAMID=$m timeout 300 python bench.py --cpu-frames 0 --frames-per-step 32 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('frontend F32 $m', d['value'], d['ms_per_step'])"; done
