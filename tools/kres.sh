#!/bin/bash
# kernel resource usage (VGPRs / occupancy / LDS) of the HIP core: tools/kres.sh [name filter regex]
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -c pyslam_amd/csrc/ps_core.hip -o /tmp/ps_core.o \
  -Rpass-analysis=kernel-resource-usage -Wno-unused-value -Wno-unused-result 2>/tmp/res.txt
python - "$1" <<'PY'
import re, sys
flt = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] else '.'
for b in open('/tmp/res.txt').read().split('Function Name: ')[1:]:
    name = b.split('\n')[0]
    if not re.search(flt, name): continue
    g = lambda k: (re.search(k + r': (\S+)', b) or [None, '?'])[1]
    print('%-60s VGPR %4s AGPR %3s occ %2s LDS %6s spill %s' % (name[:60], g('VGPRs'), g('AGPRs'), g(r'Occupancy \[waves/SIMD\]'), g(r'LDS Size \[bytes/block\]'), g('VGPRs Spill')))
PY
