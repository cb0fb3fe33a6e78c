#!/usr/bin/env python
"""Lagged dense inverse (option "lagged_inverse"): iteration counts, timings and agreement with the standard solver."""
import os, sys, time, json
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from pyslam_amd import synthetic
from pyslam_amd.device import DeviceProblem
from pyslam_amd import _native as nat
import ctypes as C

def info(dev):
    i = nat.ProblemInfo(); nat.check(dev._lib.ps_get_info(dev._h, C.byref(i)))
    return dict(solves=i.ldi_solves, fallbacks=i.ldi_fallbacks, seeds=i.ldi_seeds)

def steady(dev, steps=20, warm=6):
    for _ in range(warm):
        dev.restore(); out = dev.gn_iteration(0., 1e-12, 1000, True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        dev.restore(); out = dev.gn_iteration(0., 1e-12, 1000, True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, out

def traj(dev, iters=6):
    dev.restore(); torch.cuda.synchronize()
    rows = []
    for _ in range(iters):
        t0 = time.perf_counter()
        cost, nrm, its, rel = dev.gn_iteration(0., 1e-12, 1000, True)
        rows.append((round((time.perf_counter() - t0) * 1e3, 4), its, float('%.3e' % rel), cost))
    return rows

def run(name, lp):
    res = {}
    for mode in (1, 0):
        dev = DeviceProblem(lp, stream=torch.cuda.current_stream().cuda_stream)
        dev.set_option('lagged_inverse', mode)
        dev.eval_cost(True); dev.snapshot()
        ms, out = steady(dev)
        tr = traj(dev)
        tr2 = traj(dev)
        dev.set_profiling(2)
        for _ in range(5):
            dev.restore(); dev.gn_iteration(0., 1e-12, 1000, True)
        st = dev.stage_times(reset=True); dev.set_profiling(0)
        res[mode] = dict(steady_ms=round(ms, 4), steady_its=out[2], relres=out[3], traj=tr, traj_again=tr2, info=info(dev),
                         pcg_ms=round(st['pcg'][0] / max(st['pcg'][1], 1), 4))
        # agreement of the step at the snapshot point
        dev.restore(); dev.gn_iteration(0., 1e-12, 1000, True)
        res[mode]['dx'] = np.concatenate([a.ravel() for a in dev.get_dx()])
        res[mode]['params'] = np.concatenate([a.ravel() for a in dev.get_params()])
        dev.close()
    d = np.linalg.norm(res[1]['dx'] - res[0]['dx']) / np.linalg.norm(res[0]['dx'])
    for m in (1, 0):
        del res[m]['dx']; del res[m]['params']
    print(name, 'step agreement LDI vs standard: %.2e' % d)
    for m in (1, 0):
        print('  lagged_inverse =', m, json.dumps(res[m]))
    sys.stdout.flush()

if __name__ == '__main__':
    which = sys.argv[1:] or ['c3', 'kf100', 'pg200', 'kf250']
    if 'c3' in which:
        lp, _ = synthetic.stereo_ba(num_kf=200, num_lm=50000, obs_per_lm=10, half_window=20, seed=0); run('C3', lp)
    if 'kf100' in which:
        lp, _ = synthetic.stereo_ba(num_kf=100, num_lm=20000, obs_per_lm=10, half_window=20, seed=4); run('BA 100 kf', lp)
    if 'kf250' in which:
        lp, _ = synthetic.stereo_ba(num_kf=250, num_lm=60000, obs_per_lm=10, half_window=20, seed=5); run('BA 250 kf', lp)
    if 'pg200' in which:
        from pyslam_amd import losses
        lp, _ = synthetic.pose_graph(num_poses=200, num_loops=800, dof=6, seed=2, loss=losses.HuberLoss(1.0)); run('SE3 pose graph 200', lp)
    if 'pg2d' in which:
        from pyslam_amd import losses
        lp, _ = synthetic.pose_graph(num_poses=300, num_loops=1200, dof=3, seed=2, loss=losses.HuberLoss(1.0)); run('SE2 pose graph 300', lp)
