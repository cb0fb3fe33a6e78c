#!/usr/bin/env python
"""One-launch explicit two-level PCG (option "xcg_fused") against the three-launch form: iterations, timings, agreement."""
import os, sys, time, ctypes as C
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from pyslam_amd import synthetic, _native as nat
from pyslam_amd.device import DeviceProblem

def info(dev):
    i = nat.ProblemInfo(); nat.check(dev._lib.ps_get_info(dev._h, C.byref(i)))
    return i.xcg_fused_solves, i.xcg_fused_fallbacks

def run(name, lp, iters=6, force_explicit=False):
    out = {}
    for mode in (1, 0):
        dev = DeviceProblem(lp, stream=torch.cuda.current_stream().cuda_stream)
        dev.set_option('xcg_fused', mode)
        if force_explicit:
            dev.set_option('cg_explicit_min_rows', 0); dev.set_option('cg_split_min_rows', 0)
        dev.eval_cost(True)
        rows = []
        for it in range(iters):
            dev.set_profiling(2); dev.stage_times(reset=True)
            t0 = time.perf_counter(); c, n, its, rel = dev.gn_iteration(0., 1e-12, 4000, True); dt = (time.perf_counter() - t0) * 1e3
            st = dev.stage_times(reset=True)
            rows.append((round(dt, 3), its, float('%.2e' % rel), round(st['pcg'][0] / max(st['pcg'][1], 1), 3), c))
        dev.set_profiling(0)
        # un-profiled timing of settled iterations
        tt = []
        for it in range(4):
            t0 = time.perf_counter(); dev.gn_iteration(0., 1e-12, 4000, True); tt.append((time.perf_counter() - t0) * 1e3)
        out[mode] = (rows, dev.get_params(), info(dev), tt)
        dev.close()
    pa, pb = out[1][1], out[0][1]
    print(name, 'params agree: poses %.2e points %.2e' % (np.abs(pa[0] - pb[0]).max(), np.abs(pa[1] - pb[1]).max() if pa[1].size else 0.))
    for mode in (1, 0):
        rows, _, inf, tt = out[mode]
        print('  xcg_fused=%d (fused solves, fallbacks) %s' % (mode, inf))
        print('     ms   ', [r[0] for r in rows], 'settled', [round(t, 3) for t in tt])
        print('     its  ', [r[1] for r in rows], 'relres', [r[2] for r in rows])
        print('     pcg  ', [r[3] for r in rows])
        print('     cost ', ['%.8e' % r[4] for r in rows])
    sys.stdout.flush()

if __name__ == '__main__':
    which = sys.argv[1:] or ['mid', 'c4', 'c2']
    from pyslam_amd import losses
    if 'mid' in which:
        lp, _ = synthetic.stereo_ba(num_kf=120, num_lm=12000, obs_per_lm=8, half_window=12, seed=7); run('BA 120 kf (explicit forced)', lp, force_explicit=True)
        lp, _ = synthetic.pose_graph(num_poses=600, num_loops=2401, dof=6, seed=2, loss=losses.HuberLoss(1.0)); run('SE3 pose graph 600', lp)
        lp, _ = synthetic.pose_graph(num_poses=800, num_loops=3201, dof=3, seed=3, loss=losses.HuberLoss(1.0)); run('SE2 pose graph 800', lp)
    if 'c4' in which:
        lp, _ = synthetic.stereo_ba(num_kf=2000, num_lm=500000, obs_per_lm=10, half_window=20, seed=1); run('C4', lp, iters=5)
    if 'c2' in which:
        lp, _ = synthetic.pose_graph(num_poses=10000, num_loops=40001, dof=6, seed=2); run('C2', lp, iters=6)
