"""C3: time per iteration vs cg_margin (launches enqueued beyond the previous CG iteration count)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyslam_amd import synthetic
from pyslam_amd.device import DeviceProblem
lp, _ = synthetic.stereo_ba(num_kf=200, num_lm=50000, obs_per_lm=10, half_window=20, seed=0)
dev = DeviceProblem(lp, stream=torch.cuda.current_stream().cuda_stream)
dev.snapshot()
for m in [4, 3, 2, 1, 0]:
    dev.set_option('cg_margin', m)
    for _ in range(3):
        dev.restore(); out = dev.gn_iteration(0.0, 1e-12, 1000, True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30):
        dev.restore(); out = dev.gn_iteration(0.0, 1e-12, 1000, True)
    torch.cuda.synchronize()
    print('margin', m, 'ms/iter %.4f' % ((time.perf_counter() - t0) / 30 * 1e3), out)
