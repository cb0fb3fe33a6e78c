#!/usr/bin/env python
"""Wall time of ONE per-frame motion-only Problem, the way the reference's sparse VO pipeline uses config 5
(pyslam/pipelines/sparse.py:153-161): Problem(); add_residual_block(ReprojectionMotionOnlyBatchResidual); 
initialize_params(); solve().  Split into lowering / ps_problem_create / iterations / write-back.  Process and HIP
start-up are excluded by one warm-up frame."""
import json, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def frame(num_pts, reps=30):
    from liegroups import SE3
    from pyslam.problem import Options, Problem
    from pyslam.sensors import StereoCamera
    from pyslam.residuals import ReprojectionMotionOnlyBatchResidual
    from pyslam.losses import CauchyLoss
    from pyslam_amd import synthetic, problem as pmod, device as dmod
    opt = Options()
    opt.allow_nondecreasing_steps, opt.max_nondecreasing_steps = True, 5
    opt.min_cost_decrease, opt.max_iters, opt.linesearch_max_iters = 0.99, 30, 0
    cam = StereoCamera(*synthetic.STEREO_BA_CAMERA)
    stiff = None
    split = {'lower': 0., 'create': 0., 'iterate': 0., 'write_back': 0.}
    # instrument by wrapping the three stages
    orig_lower, orig_make, orig_wb = pmod.Problem._lower, pmod.Problem._make_device, pmod.Problem._write_back
    def t_lower(self, *a, **k):
        t0 = time.perf_counter(); r = orig_lower(self, *a, **k); split['lower'] += time.perf_counter() - t0; return r
    def t_make(self, *a, **k):
        t0 = time.perf_counter(); r = orig_make(self, *a, **k); split['create'] += time.perf_counter() - t0; return r
    def t_wb(self, *a, **k):
        t0 = time.perf_counter(); r = orig_wb(self, *a, **k); split['write_back'] += time.perf_counter() - t0; return r
    pmod.Problem._lower, pmod.Problem._make_device, pmod.Problem._write_back = t_lower, t_make, t_wb
    walls, iters = [], []
    try:
        for rep in range(reps + 2):
            lp, aux = synthetic.motion_only(num_pts=num_pts, seed=100 + rep)
            S = lp.stiff3[0].reshape(3, 3)
            if rep == 2:
                for k in split: split[k] = 0.
            t0 = time.perf_counter()
            problem = Problem(opt)
            problem.add_residual_block(ReprojectionMotionOnlyBatchResidual(cam, aux['obs_1'], aux['obs_2'], S),
                                       ['T_1_0'], CauchyLoss(3.0))
            problem.initialize_params({'T_1_0': SE3.identity()})
            out = problem.solve()
            w = time.perf_counter() - t0
            if rep >= 2:
                walls.append(w); iters.append(len(problem._cost_history) - 1)
            del problem
    finally:
        pmod.Problem._lower, pmod.Problem._make_device, pmod.Problem._write_back = orig_lower, orig_make, orig_wb
    n = len(walls)
    res = {'num_pts': num_pts, 'frames': n, 'wall_ms_mean': 1e3 * float(np.mean(walls)), 'wall_ms_median': 1e3 * float(np.median(walls)),
           'gn_iterations_mean': float(np.mean(iters)),
           'lower_ms': 1e3 * split['lower'] / n, 'create_ms': 1e3 * split['create'] / n, 'write_back_ms': 1e3 * split['write_back'] / n}
    res['iterate_and_rest_ms'] = res['wall_ms_mean'] - res['lower_ms'] - res['create_ms'] - res['write_back_ms']
    return res


if __name__ == '__main__':
    out = [frame(256), frame(2048)]
    print(json.dumps(out))
