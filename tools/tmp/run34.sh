export MSL_LIB=$PWD/manhattanslam_amd/variants/libmsl_old.so
python tools/tmp/floor2.py 2>&1 | grep -v amdgpu.ids
