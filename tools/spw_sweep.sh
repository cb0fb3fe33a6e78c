for w in 4 5 6; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread -Wno-unused-value -Wno-unused-result -DPS_SP_WAVES=$w \
     -Iinclude pyslam_amd/csrc/ps_core.hip -o pyslam_amd/lib/libpyslam_hip.so 2>/dev/null
  for rep in 1 2; do
  python bench.py --no-cpu-baseline --no-c4 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('waves $w', d['value'], 'schur', d['stage_ms']['schur_pairs'], 'frac', d['roofline']['frac'])
"
  done
done
