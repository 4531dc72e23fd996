export MSL_LIB=$PWD/manhattanslam_amd/variants/libmsl_old.so
NMAP=1000 python tools/tmp/floor.py 2>&1 | grep nmap
NMAP=1000000 python tools/tmp/floor.py 2>&1 | grep nmap
python bench.py --config 2 --cpu-frames 0 2>&1 | tail -1 | cut -c1-200
