#!/bin/bash
# Build a whole-library variant with extra preprocessor flags (all translation units): tools/build_full_variant.sh <name> [hipcc flags]
# -> scratch/libmsl_<name>.so (MSL_LIB selects it).  Experiments only.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); NAME=$1; shift
C=$R/manhattanslam_amd/csrc; O=$R/scratch/var_$NAME; mkdir -p $O
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -fno-gpu-flush-denormals-to-zero -I$R/include -I$C"
for o in msl_common msl_orb msl_sf_superpixel msl_sf_map msl_surfel msl_peac msl_match; do
  case $o in msl_sf_superpixel|msl_sf_map|msl_surfel) /opt/rocm/bin/hipcc $FLAGS "$@" -c $C/$o.hip -o $O/$o.o & ;; *) cp $C/$o.o $O/$o.o ;; esac
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $O/*.o -o $R/scratch/libmsl_$NAME.so
echo built scratch/libmsl_$NAME.so
