"""Explicit two-level PCG: three launches per iteration (xcg_fused = 0) against the one-launch form / its automatic choice (1)
and the two-launch form (2) over problem sizes -- the measurement behind the selection rule in ps_host_iteration.h."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.getcwd()))
import numpy as np, torch
from pyslam_amd import synthetic, losses
from pyslam_amd.device import DeviceProblem
cases = []
for P in (1000, 1500, 2000, 5000):
    cases.append(('pg6_%d' % P, (lambda P=P: synthetic.pose_graph(num_poses=P, num_loops=4 * P + 1, dof=6, seed=2)[0])))
for P in (1500, 4000):
    cases.append(('pg3_%d' % P, (lambda P=P: synthetic.pose_graph(num_poses=P, num_loops=4 * P + 1, dof=3, seed=2)[0])))
for K in (1000, 4000):
    cases.append(('ba_%d' % K, (lambda K=K: synthetic.stereo_ba(num_kf=K, num_lm=40 * K, obs_per_lm=6, half_window=8, seed=1)[0])))
for name, make in cases:
    lp = make()
    for mode in (0, 1, 2):
        dev = DeviceProblem(lp)
        dev.set_option('xcg_fused', mode); dev.set_option('lagged_inverse', 0)
        dev.eval_cost(True)
        rows = []
        for _ in range(6):
            torch.cuda.synchronize(); t0 = time.perf_counter(); out = dev.gn_iteration(0., 1e-12, 4000, True); rows.append(((time.perf_counter() - t0) * 1e3, out[2]))
        i = dev.get_info()
        print('%-9s nr %5d xcg_fused=%d last3 %.3f ms  ms %s its %s fused %d fallbacks %d' % (name, dev.nr, mode, np.mean([r[0] for r in rows[3:]]),
              [round(r[0], 3) for r in rows], [r[1] for r in rows], i['xcg_fused_solves'], i['xcg_fused_fallbacks']), flush=True)
        dev.close()
