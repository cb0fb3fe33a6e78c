"""The one-launch solves (k_cg_persist, k_xcg_persist) exchange data between workgroups INSIDE a launch: the order in which workgroups
run must not show in the result.  Repeated cold solves of the same problem on one handle and on fresh handles: cost histories and final
parameters compared bit for bit.   python tools/probes/persist_determinism.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from pyslam_amd import synthetic
from pyslam_amd.device import DeviceProblem
from pyslam_amd.problem import device_solve

for name, kw, reps in (('C3 200 x 50 000', dict(num_kf=200, num_lm=50000, obs_per_lm=10, half_window=20, seed=0), 60),
                       ('BA 1 000 x 60 000', dict(num_kf=1000, num_lm=60000, obs_per_lm=10, half_window=20, seed=1), 30),
                       ('C4 2 000 x 500 000', dict(num_kf=2000, num_lm=500000, obs_per_lm=10, half_window=20, seed=1), 12)):
    lp, _ = synthetic.stereo_ba(**kw)
    start = (lp.poses.copy(), lp.points.copy())
    ref, same, handles = None, 0, 0
    for fresh in range(2):
        dev = DeviceProblem(lp, stream=torch.cuda.current_stream().cuda_stream)
        handles += 1
        for r in range(reps // 2):
            dev.reset_solver_state(); dev.set_params(*start)
            hist, stats = device_solve(dev, bench.example_options())
            p = dev.get_params()
            got = (hist, [s[0] for s in stats], p[0].tobytes(), p[1].tobytes())
            if ref is None:
                ref = got
            same += got == ref
        counts = dev.cg_persist_counts()
        dev.close()
    print('%-20s %d solves on %d handles, %d bit-identical to the first (history, CG iterations %s, poses, points); one-launch solves / time-outs on the last handle: %s'
          % (name, reps // 2 * 2, handles, same, ref[1], counts))
