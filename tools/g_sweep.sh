python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys, os, time
sys.path.insert(0, '.')
import torch
from pyslam_amd import synthetic
from pyslam_amd.device import DeviceProblem
lp, _ = synthetic.stereo_ba(200, 50000, 10, 20, seed=0)
for G in (-1, 9, 10, 12, 13, 14, 15):
    dev = DeviceProblem(lp)
    dev.set_option('coarse_groups', G)
    dev.snapshot()
    for _ in range(8):
        dev.restore(); out = dev.gn_iteration(0., 1e-12, 1000, True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100):
        dev.restore(); out = dev.gn_iteration(0., 1e-12, 1000, True)
    torch.cuda.synchronize(); print('G', G, 'wall us/iter %.1f' % ((time.perf_counter() - t0) * 1e4), 'pcg iters', out[2])
    dev.close()
PY
