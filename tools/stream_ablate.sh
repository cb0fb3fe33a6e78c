# (PS_* measurement switches exist only in the measurement build: python -c "import __graft_entry__ as g; g.build_measure()" first)
export PYSLAM_AMD_MEASURE=1
PS_SCHUR_STREAM=1 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys, time
sys.path.insert(0, '.')
import torch
from pyslam_amd import synthetic
from pyslam_amd.device import DeviceProblem
lp, _ = synthetic.stereo_ba(200, 50000, 10, 20, seed=0)
dev = DeviceProblem(lp)
dev.snapshot()
for ab in (0, 1, 2, 4, 5, 7):
    dev.set_option('schur_ablate', ab)
    dev.set_profiling(2)
    for _ in range(3):
        dev.restore(); dev.linearize(0.)
    torch.cuda.synchronize()
    st = dev.stage_times(reset=True)
    print('ablate', ab, 'schur ms %.4f' % (st['schur_pairs'][0] / st['schur_pairs'][1]))
PY
