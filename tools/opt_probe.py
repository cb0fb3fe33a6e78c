"""C3: ms/iteration with one option toggled: tools/opt_probe.py <option> <v1> <v2> ..."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyslam_amd import synthetic
from pyslam_amd.device import DeviceProblem
lp, _ = synthetic.stereo_ba(num_kf=200, num_lm=50000, obs_per_lm=10, half_window=20, seed=0)
opt = sys.argv[1]
for v in [float(a) for a in sys.argv[2:]]:
    dev = DeviceProblem(lp, stream=torch.cuda.current_stream().cuda_stream)
    dev.set_option(opt, v)
    dev.snapshot()
    for _ in range(5):
        dev.restore(); out = dev.gn_iteration(0.0, 1e-12, 1000, True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        dev.restore(); out = dev.gn_iteration(0.0, 1e-12, 1000, True)
    torch.cuda.synchronize()
    print(opt, v, 'ms/iter %.4f' % ((time.perf_counter() - t0) / 50 * 1e3), out)
    del dev
