"""Time of ps_problem_create (host-side structure build + upload) at C3 and C4."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyslam_amd import synthetic
from pyslam_amd.device import DeviceProblem
for kf, lm in ((200, 50000), (2000, 500000)):
    t0 = time.perf_counter()
    lp, _ = synthetic.stereo_ba(num_kf=kf, num_lm=lm, obs_per_lm=10, half_window=20, seed=0)
    t1 = time.perf_counter()
    dev = DeviceProblem(lp)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for rep in range(2):        # again: without the process-wide one-off costs (first allocations, code objects)
        dev.close()
        ta = time.perf_counter()
        dev = DeviceProblem(lp)
        torch.cuda.synchronize()
        print('kf {} lm {}: ps_problem_create again: {:.1f} ms'.format(kf, lm, (time.perf_counter() - ta) * 1e3))
    t2b = time.perf_counter()
    t2s = time.perf_counter()
    out = dev.gn_iteration(0.0, 1e-12, 4000, True)
    t3 = time.perf_counter()
    out = dev.gn_iteration(0.0, 1e-12, 4000, True)
    t4 = time.perf_counter()
    print('kf {} lm {}: synthetic tables {:.2f} s, ps_problem_create {:.2f} s, first iteration {:.1f} ms, second {:.1f} ms'.format(
        kf, lm, t1 - t0, t2 - t1, (t3 - t2s) * 1e3, (t4 - t3) * 1e3), out)
    dev.close()
