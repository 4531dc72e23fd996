#!/bin/bash
# Kernel resource usage of one .hip file (gfx950): name, VGPRs, SGPR spills, VGPR spills, scratch, LDS.   tools/kres.sh <file.hip> [name filter]
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/manhattanslam_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -fno-gpu-flush-denormals-to-zero -I$C -I$R/include"
/opt/rocm/bin/hipcc $FLAGS -Rpass-analysis=kernel-resource-usage -c $1 -o /dev/null 2>&1 | python3 -c "
import re, sys
cur = None; rows = {}
for l in sys.stdin:
    m = re.search(r'remark: *(?:Function )?Name: (\S+)', l)
    if m: cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r'remark: *([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)', l)
    if m and cur: rows[cur][m.group(1).strip()] = int(m.group(2))
    if 'error' in l: print(l.rstrip())
flt = sys.argv[1] if len(sys.argv) > 1 else ''
for k, v in rows.items():
    if flt in k: print('%-70s VGPR %3d  sgprSpill %3d  vgprSpill %3d  scratch %4d  LDS %6d  occ %d' % (k[:70], v.get('VGPRs', -1), v.get('SGPRs Spill', -1), v.get('VGPRs Spill', -1), v.get('ScratchSize', -1), v.get('LDS Size', -1), v.get('Occupancy', -1)))
" "$2"
