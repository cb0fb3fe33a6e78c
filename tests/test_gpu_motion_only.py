"""GPU: the whole Problem.solve() loop of a one-pose motion-only problem in one launch (ps_motion_only_solve,
Options.fused_solve_loop; config C5's per-frame Problem, reference pipelines/sparse.py:153-161) against the same loop
driven from the host one iteration at a time, and against the oracle's restatement of the reference loop
(reference problem.py:130-178)."""
import numpy as np
import pytest

from oracle import gn_oracle as orc
from pyslam_amd import synthetic

pytestmark = pytest.mark.gpu

PIPELINE = dict(allow_nondecreasing_steps=True, max_nondecreasing_steps=5, min_cost_decrease=0.99, max_iters=30,
                linesearch_max_iters=0)      # reference pipelines/sparse.py:34-39


def solve_through_the_api(aux, lp, options, fused, loss=None):
    from liegroups import SE3
    from pyslam.problem import Options, Problem
    from pyslam.sensors import StereoCamera
    from pyslam.residuals import ReprojectionMotionOnlyBatchResidual
    from pyslam.losses import CauchyLoss
    opt = Options()
    for k, v in options.items():
        setattr(opt, k, v)
    opt.fused_solve_loop = fused
    problem = Problem(opt)
    cam = StereoCamera(*synthetic.STEREO_BA_CAMERA)
    problem.add_residual_block(
        ReprojectionMotionOnlyBatchResidual(cam, aux['obs_1'], aux['obs_2'], lp.stiff3[0].reshape(3, 3)),
        ['T_2_1'], loss if loss is not None else CauchyLoss(3.0))
    problem.initialize_params({'T_2_1': SE3.identity()})
    out = problem.solve()
    return np.array(problem._cost_history), out['T_2_1'].as_matrix(), list(problem.solver_stats)


OPTION_SETS = [
    PIPELINE,
    dict(PIPELINE, linesearch_max_iters=10),                                  # cost after the step
    dict(max_iters=30, min_cost_decrease=0.99, linesearch_max_iters=10),      # reference defaults: stop at the first bad step
    dict(max_iters=30, min_cost_decrease=0.99, linesearch_max_iters=0),
    dict(PIPELINE, max_iters=3),                                              # the iteration limit ends it
    dict(PIPELINE, max_nondecreasing_steps=2, min_cost_decrease=0.9),         # best parameters restored early
    dict(PIPELINE, min_update_norm=1e-3),                                     # the step norm ends it
    dict(PIPELINE, min_cost=1e6),                                             # the cost threshold ends it at once
]


@pytest.mark.parametrize('num_pts,seed', [(256, 3), (2048, 4), (40, 5)])
@pytest.mark.parametrize('k', range(len(OPTION_SETS)))
def test_one_launch_solve_equals_the_iteration_by_iteration_loop(num_pts, seed, k):
    """Same decisions, same numbers: cost history from entry 1 on and the final pose are bit-identical (entry 0, the start
    cost, comes from the cost kernel in the host-driven loop and from the iteration kernel's own sum here: equal to
    rounding), and both follow the oracle's restatement of the reference loop."""
    options = OPTION_SETS[k]
    lp, aux = synthetic.motion_only(num_pts=num_pts, seed=seed)
    h1, T1, s1 = solve_through_the_api(aux, lp, options, True)
    h0, T0, s0 = solve_through_the_api(aux, lp, options, False)
    assert len(h1) == len(h0), (h1, h0)
    assert np.array_equal(h1[1:], h0[1:]) and abs(h1[0] - h0[0]) <= 1e-14 * h0[0]
    assert np.array_equal(T1, T0) and s1 == s0
    final, ref = orc.solve(lp, options, points_first=False)
    assert len(h1) == len(ref['cost_history']) and np.allclose(h1, ref['cost_history'], rtol=1e-10)
    from pyslam_amd.lowering import pack_pose_matrices
    assert np.abs(pack_pose_matrices(T1[None])[0] - final.poses[0]).max() < 1e-9


def test_only_single_pose_motion_only_problems_take_the_one_launch_loop():
    """Anything else answers "iterate yourself" (return code 1, nothing touched): two poses, variable landmarks."""
    from pyslam_amd.device import DeviceProblem
    from pyslam.problem import Options
    lp, _ = synthetic.stereo_ba(num_kf=4, num_lm=40, obs_per_lm=3, half_window=2, seed=7)
    dev = DeviceProblem(lp)
    before = dev.get_params()
    assert dev.motion_only_solve(Options(), True) is None
    after = dev.get_params()
    assert np.array_equal(before[0], after[0]) and np.array_equal(before[1], after[1])
    lp1, _ = synthetic.motion_only(num_pts=64, seed=8)
    dev1 = DeviceProblem(lp1)
    opt = Options()
    opt.max_iters = 400                                       # history longer than the pinned block holds
    assert dev1.motion_only_solve(opt, True) is None
    opt.max_iters = 20
    hist, its, dxn, pose = dev1.motion_only_solve(opt, True)
    assert len(hist) == its + 1 and hist[-1] <= hist[0]
    assert np.array_equal(pose, dev1.get_params()[0][0])
