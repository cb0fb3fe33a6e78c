"""C3: ms/iteration and CG iterations vs the number of coarse hat intervals ("coarse_groups")."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyslam_amd import synthetic
from pyslam_amd.device import DeviceProblem
lp, _ = synthetic.stereo_ba(num_kf=200, num_lm=50000, obs_per_lm=10, half_window=20, seed=0)
for G in [int(a) for a in sys.argv[1:]] or [8, 10, 12, 14, 15]:
    dev = DeviceProblem(lp, stream=torch.cuda.current_stream().cuda_stream)
    dev.set_option('coarse_groups', G)
    dev.snapshot()
    for _ in range(3):
        dev.restore(); out = dev.gn_iteration(0.0, 1e-12, 1000, True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30):
        dev.restore(); out = dev.gn_iteration(0.0, 1e-12, 1000, True)
    torch.cuda.synchronize()
    print('G', G, 'ms/iter %.4f' % ((time.perf_counter() - t0) / 30 * 1e3), 'cg iters', out[2])
    del dev
