"""When should a call TRY the lagged dense inverse?  Sweep of option "ldi_cost_tol" (relative distance, in cost, between the
point the inverse was built at and the point the call linearises at): whole trajectories from the perturbed start, wall clock.
usage: python tools/ldi_tol_sweep.py   (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyslam_amd import synthetic, losses
from pyslam_amd.device import DeviceProblem

PROBLEMS = {
    'C3': lambda: synthetic.stereo_ba(num_kf=200, num_lm=50000, obs_per_lm=10, half_window=20, seed=0)[0],
    'BA 100 kf': lambda: synthetic.stereo_ba(num_kf=100, num_lm=20000, obs_per_lm=10, half_window=20, seed=4)[0],
    'BA 250 kf huber': lambda: synthetic.stereo_ba(num_kf=250, num_lm=60000, obs_per_lm=10, half_window=20, seed=5, loss=losses.HuberLoss(1.5))[0],
    'BA 60 kf cauchy': lambda: synthetic.stereo_ba(num_kf=60, num_lm=8000, obs_per_lm=8, half_window=12, seed=6, loss=losses.CauchyLoss(3.0))[0],
    'SE3 graph 200': lambda: synthetic.pose_graph(num_poses=200, num_loops=800, dof=6, seed=2, loss=losses.HuberLoss(1.0))[0],
}
for name, make in PROBLEMS.items():
    lp = make()
    for tol in (0.05, 0.01, 0.002, 0.0005, -1.0):
        dev = DeviceProblem(lp)
        if tol < 0:
            dev.set_option('lagged_inverse', 0)
        else:
            dev.set_option('ldi_cost_tol', tol)
        dev.eval_cost(True); dev.snapshot()
        best = None
        for rep in range(3):                                     # (the first repetition also pays first-call effects)
            dev.restore(); torch.cuda.synchronize()
            rows = []
            for _ in range(8):
                t0 = time.perf_counter(); out = dev.gn_iteration(0., 1e-12, 1000, True)
                rows.append(((time.perf_counter() - t0) * 1e3, out[2]))
            tot = sum(r[0] for r in rows)
            if best is None or tot < best[0]:
                best = (tot, rows)
        i = dev.get_info()
        print('%-16s tol %-7s total %.3f ms  ms %s its %s  solves %d fallbacks %d seeds %d' % (
            name, 'off' if tol < 0 else tol, best[0], [round(r[0], 3) for r in best[1]], [r[1] for r in best[1]],
            i['ldi_solves'], i['ldi_fallbacks'], i['ldi_seeds']), flush=True)
        dev.close()
