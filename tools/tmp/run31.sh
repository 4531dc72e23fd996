run() { echo "== $*"; env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --frames-per-step 32 --cpu-frames 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('F32', d['value'], d['roofline']['avg_launch_us'])"; }
rund() { echo "== default $*"; env "$@" timeout 300 python bench.py --cpu-frames 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('F256', d['value'], d['roofline']['avg_launch_us'])"; }
OLD=MSL_LIB=$PWD/manhattanslam_amd/variants/libmsl_old.so
run A=1
run $OLD
run MSL_FUSE_GRID=512
run A=1
run $OLD
run MSL_FUSE_GRID=512
rund A=1
rund $OLD
rund MSL_FUSE_GRID=512
echo == quick; python tools/quick_sf_bench.py 2>&1 | tail -1 | cut -c1-120
env $OLD python tools/quick_sf_bench.py 2>&1 | tail -1 | cut -c1-160
MSL_FUSE_GRID=512 python tools/quick_sf_bench.py 2>&1 | tail -1 | cut -c1-120
