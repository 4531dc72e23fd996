run() { echo "== $*"; env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --frames-per-step 32 --cpu-frames 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('F32', d['value'], d['roofline']['avg_launch_us'])"; }
run A=1
run MSL_MAP_WAVE_PRIO=0
run MSL_MAP_WAVE_PRIO=0 MSL_MAP_STREAM_PRIO=lo
run MSL_MAP_WAVE_PRIO=1 MSL_MAP_STREAM_PRIO=lo
run MSL_MAP_WAVE_PRIO=0 MSL_MAP_STREAM_PRIO=mid
run MSL_MAP_WAVE_PRIO=0 MSL_MAP_STREAM_PRIO=lo MSL_FUSE_GRID=512
run MSL_MAP_WAVE_PRIO=0 MSL_MAP_STREAM_PRIO=lo MSL_FUSE_GRID=1024
run MSL_MAP_WAVE_PRIO=1 MSL_FUSE_GRID=512
run MSL_MAP_WAVE_PRIO=1 MSL_FUSE_GRID=256
