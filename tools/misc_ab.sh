cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
for d in 1 0; do MSL_SF_DEAL=$d timeout 400 python bench.py --config 5 --cpu-frames 0 --no-breakdown --no-parity-gate --steps 6 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('config5 deal=$d', d['value'], r['frac'], r['avg_launch_us'])"; done
for d in 1 0; do MSL_SF_DEAL=$d timeout 400 python bench.py --sequences-per-gpu 2 --cpu-frames 0 --no-breakdown --no-parity-gate --steps 6 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('2seq deal=$d', d['value'], r['frac'], r['avg_launch_us'])"; done
for d in 1 0; do MSL_SF_DEAL=$d timeout 400 python bench.py --map moving --cpu-frames 0 --no-breakdown --no-parity-gate --steps 6 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('moving deal=$d', d['value'], r['frac'], r['avg_launch_us'])"; done
for d in 1 0; do MSL_SF_DEAL=$d timeout 400 python bench.py --surfels 8000000 --cpu-frames 0 --no-breakdown --no-parity-gate --steps 4 --passes-per-step 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('8M deal=$d', d['value'], r['frac'], r['avg_launch_us'])"; done
for d in 1 0; do MSL_SF_DEAL=$d timeout 400 python bench.py --map-order random --cpu-frames 0 --no-breakdown --no-parity-gate --steps 6 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('random-order deal=$d', d['value'], r['frac'], r['avg_launch_us'])"; done
