"""C4 on one GPU under solver options: 8 consecutive iterations of a real solve, wall clock and CG iterations per call.
usage: python tools/c4_options_probe.py "opt=value,opt=value" ["..." ...]    (one line per option set; '' = defaults)"""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from pyslam_amd import synthetic
from pyslam_amd.device import DeviceProblem
kf, lm = int(os.environ.get('KF', 2000)), int(os.environ.get('LM', 500000))
lp, _ = synthetic.stereo_ba(kf, lm, 10, 20, seed=0)
for opts in (sys.argv[1:] or ['']):
    dev = DeviceProblem(lp)
    for kv in opts.split(','):
        if kv:
            dev.set_option(kv.split('=')[0], float(kv.split('=')[1]))
    ms, its = [], []
    for it in range(8):
        torch.cuda.synchronize()
        t = time.perf_counter(); out = dev.gn_iteration(0., 1e-12, 3000, True); ms.append((time.perf_counter() - t) * 1e3); its.append(out[2])
    print('%-40s ms %s its %s final cost %.9e' % (opts or '(defaults)', [round(m, 3) for m in ms], its, out[0]), flush=True)
    dev.close()
