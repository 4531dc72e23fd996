#!/bin/bash
# Build a kernel-variant library for A/B experiments: tools/build_variant.sh <name> <sed-expression on msl_surfel.hip> [extra hipcc flags]
# -> scratch/libmsl_<name>.so (select it with MSL_LIB=...).  Experiments only; the product is manhattanslam_amd/libmsl.so.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; SED=$2; shift 2
mkdir -p $R/scratch/var_$NAME
sed "$SED" $R/manhattanslam_amd/csrc/msl_surfel.hip > $R/scratch/var_$NAME/msl_surfel.hip
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -fno-gpu-flush-denormals-to-zero -I$R/include -I$R/manhattanslam_amd/csrc"
/opt/rocm/bin/hipcc $FLAGS "$@" -c $R/scratch/var_$NAME/msl_surfel.hip -o $R/scratch/var_$NAME/msl_surfel.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A8 "Function Name: .*\(k_fuse\|kb_seed_plane\|kb_update_seeds\|k_compact\)" | grep -E "Name|VGPRs:|Scratch|Occupancy" | sed 's/.*remark: *//;s/\[-Rpass.*//' | paste - - - - 
C=$R/manhattanslam_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $C/msl_common.o $C/msl_orb.o $R/scratch/var_$NAME/msl_surfel.o $C/msl_peac.o $C/msl_match.o -o $R/scratch/libmsl_$NAME.so
echo built scratch/libmsl_$NAME.so
