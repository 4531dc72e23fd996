#!/bin/bash
# Build a kernel-variant library for A/B experiments:
#   tools/build_variant.sh <name> <file.hip> <sed-expression on that file> [extra hipcc flags]
# -> scratch/libmsl_<name>.so (select it with MSL_LIB=...).  Experiments only; the product is manhattanslam_amd/libmsl.so.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; FILE=$2; SED=$3; shift 3
C=$R/manhattanslam_amd/csrc
mkdir -p $R/scratch/var_$NAME
sed "$SED" $C/$FILE > $R/scratch/var_$NAME/$FILE
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -fno-gpu-flush-denormals-to-zero -I$R/include -I$C"
/opt/rocm/bin/hipcc $FLAGS "$@" -c $R/scratch/var_$NAME/$FILE -o $R/scratch/var_$NAME/${FILE%.hip}.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A9 "Name: .*\(k_fuse\|kb_seed_plane\|kb_update_seeds\|kb_assign\|k_compact\|k_replay\)" | grep -E "Name|VGPRs:|SGPRs Spill|Scratch|Occupancy" | sed 's/.*remark: *//;s/\[-Rpass.*//;s/.*Name: //' | paste - - - - - | cut -c1-200
OBJS=""
for o in msl_common msl_orb msl_sf_superpixel msl_sf_map msl_surfel msl_peac msl_match; do
  if [ "$o.hip" == "$FILE" ]; then OBJS="$OBJS $R/scratch/var_$NAME/$o.o"; else OBJS="$OBJS $C/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $R/scratch/libmsl_$NAME.so
echo built scratch/libmsl_$NAME.so
