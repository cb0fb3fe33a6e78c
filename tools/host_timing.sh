mkdir -p gpurun_out/r02d
for lx in 1 0; do
PS_HOST_TIMING=1 LAGX=$lx python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys, os, time
sys.path.insert(0, '.')
import torch
from pyslam_amd import synthetic
from pyslam_amd.device import DeviceProblem
lp, _ = synthetic.stereo_ba(200, 50000, 10, 20, seed=0)
dev = DeviceProblem(lp)
dev.set_option('coarse_lag_x', int(os.environ['LAGX']))
dev.snapshot()
for _ in range(10):
    dev.restore(); dev.gn_iteration(0., 1e-12, 1000, True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100):
    dev.restore(); dev.gn_iteration(0., 1e-12, 1000, True)
torch.cuda.synchronize(); print('lagx', os.environ['LAGX'], 'wall us/iter', (time.perf_counter() - t0) * 1e4)
dev.close()
PY
done
