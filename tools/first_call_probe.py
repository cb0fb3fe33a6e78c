"""Where does the first whole-iteration call of a process spend its time?  A fresh process: [optionally a small warm-up problem
first], then the C3 problem: create, first, second, third call.   python tools/first_call_probe.py [warm_kf]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyslam_amd import synthetic
from pyslam_amd.device import DeviceProblem
torch.cuda.synchronize()
warm = int(sys.argv[1]) if len(sys.argv) > 1 else 0
if warm:
    lpw, _ = synthetic.stereo_ba(num_kf=warm, num_lm=warm * 20, obs_per_lm=5, half_window=4, seed=1)
    t = time.perf_counter(); w = DeviceProblem(lpw); torch.cuda.synchronize(); t1 = time.perf_counter()
    w.gn_iteration(0., 1e-12, 500, True); t2 = time.perf_counter()
    w.gn_iteration(0., 1e-12, 500, True); t3 = time.perf_counter()
    w.close()
    print('warm-up problem (%d keyframes, %d reduced unknowns): create %.1f ms, first call %.2f ms, second %.2f ms' % (warm, w.nr * 6, (t1 - t) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
lp, _ = synthetic.stereo_ba(num_kf=200, num_lm=50000, obs_per_lm=10, half_window=20, seed=0)
t = time.perf_counter(); dev = DeviceProblem(lp); torch.cuda.synchronize(); t1 = time.perf_counter()
c = dev.eval_cost(True); t2 = time.perf_counter()
ts = []
for k in range(3):
    t0 = time.perf_counter(); dev.gn_iteration(0., 1e-12, 2000, True); ts.append((time.perf_counter() - t0) * 1e3)
print('C3: create %.1f ms, start cost %.2f ms, calls %s ms' % ((t1 - t) * 1e3, (t2 - t1) * 1e3, ['%.3f' % x for x in ts]))
dev.close()
