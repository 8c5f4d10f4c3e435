#!/bin/bash
# scripts/kernel_usage.sh <reg id | vit> -- VGPRs / scratch / occupancy of every kernel of one translation unit
cd "$(dirname "$0")/../viterbidecodercpp_amd/csrc"
if [ "$1" = "vit" ]; then SRC=vit_hip.hip; DEF=""; else SRC=reg_inst.hip; DEF="-DVIT_REG_ID=$1"; fi
hipcc -O3 -std=c++17 --offload-arch=gfx950 $DEF -S --cuda-device-only -Rpass-analysis=kernel-resource-usage -o /dev/null $SRC 2>&1 |
  python3 -c "
import re,sys
name=None
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: name=m.group(1); d={}
    m=re.search(r'remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)',l)
    if m and name:
        d[m.group(1).split(' ')[0]]=int(m.group(2))
        if m.group(1).startswith('LDS'): print(f\"{d.get('VGPRs'):4d} vgpr {d.get('AGPRs',0):3d} agpr {d.get('ScratchSize'):4d} scratch  occ {d.get('Occupancy')}  lds {d.get('LDS')}  {name[:110]}\")
"
