# pose pass at C3 / C4: chunk size (PS_POSE_CHUNK) x reduction form (rebuild with -DPS_POSE_TRANSPOSE=0/1)
# (PS_* measurement switches exist only in the measurement build: python -c "import __graft_entry__ as g; g.build_measure()" first)
export PYSLAM_AMD_MEASURE=1
for tr in 1 0; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread -Wno-unused-value -Wno-unused-result -DPS_POSE_TRANSPOSE=$tr \
     -Iinclude pyslam_amd/csrc/ps_core.hip -o pyslam_amd/lib/libpyslam_hip.so 2>/dev/null
  for ch in 256 512 1024; do
    PS_POSE_CHUNK=$ch python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('transpose $tr chunk $ch: C3', d['value'], 'pose', d['stage_ms']['pose_pass'], '| C4', d['c4_single_gpu']['ms'], 'pose', d['c4_single_gpu']['stage_ms']['pose_pass'])
"
  done
done
