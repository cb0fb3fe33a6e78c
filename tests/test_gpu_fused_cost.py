"""The cost after a step IS the cost at the next linearisation point (reference pyslam/problem.py:159-161: `cost` returned by
solve_one_iter is evaluated at the updated parameters, and :338-360 evaluates the same residuals again for the next
iteration's Jacobians).  The core evaluates every observation once: the tail of an iteration that expects a successor runs the
successor's landmark pass, which sums the cost on its way (csrc/ps_k_packed.h: k_landmark_pass_packed<.., COST>, option
fuse_cost; ps_host_cg.h: gn_tail).  Held here: a cost is the same number, bit for bit, whichever kernel summed it; solves are the
same with the fusion on and off; a landmark block that is not positive definite at the point run ahead is reported by the call
that linearises there, not earlier."""
import numpy as np
import pytest

from oracle import gn_oracle as orc
from pyslam_amd import synthetic, losses
from test_gpu_packed import ragged_ba

pytestmark = pytest.mark.gpu


def example_options(**kw):
    from pyslam_amd.problem import Options
    opt = Options()
    opt.allow_nondecreasing_steps = True
    opt.max_nondecreasing_steps = 3
    opt.pcg_tol, opt.pcg_max_iters = 1e-12, 2000
    for k, v in kw.items():
        setattr(opt, k, v)
    return opt


CASES = [
    ('uniform10', lambda: synthetic.stereo_ba(num_kf=40, num_lm=3000, obs_per_lm=10, half_window=8, seed=2)[0]),
    ('ragged16_cauchy', lambda: ragged_ba(5, const_point_fraction=0.0, loss=losses.CauchyLoss(2.0))),
    ('ragged7_huber', lambda: ragged_ba(6, max_obs=7, const_point_fraction=0.0, loss=losses.HuberLoss(1.5))),
    ('two_obs', lambda: synthetic.stereo_ba(num_kf=12, num_lm=500, obs_per_lm=2, half_window=3, seed=9)[0]),
]


@pytest.mark.parametrize('name,make', CASES, ids=[c[0] for c in CASES])
def test_a_cost_is_one_number_whichever_kernel_summed_it(name, make):
    """fuse_cost 1: ps_eval_cost runs the landmark pass itself (COST instantiation); 2: the cost-only pass in the same structure
    (k_cost_packed); 0: the grid-stride pass of rounds 1-4 (k_cost_reproj: another summation order).  Then whole iterations:
    the tail's fused pass (expect_next) against the cost-only pass."""
    from pyslam_amd.device import DeviceProblem
    lp = make()
    costs, traces, params = {}, {}, {}
    for mode in (1, 2, 0):
        dev = DeviceProblem(lp)
        dev.set_option('fuse_cost', mode)
        dev.set_option('lagged_inverse', 0)
        costs[mode] = dev.eval_cost(True)
        tr = []
        for k in range(3):
            dev.set_expect_next(k < 2)
            tr.append(dev.gn_iteration(0.0, 1e-12, 2000, True))
        dev.set_expect_next(False)
        tr.append(dev.gn_iteration(0.0, 1e-12, 2000, True))         # (its landmark pass ran in the third call's tail ...)
        traces[mode] = tr
        costs[(mode, 'end')] = dev.eval_cost(True)                  # (... and this is the cost the fourth call returned)
        params[mode] = dev.get_params()
        dev.close()
    assert costs[1] == costs[2]
    assert abs(costs[1] - costs[0]) <= 1e-13 * abs(costs[0])
    assert abs(costs[1] - orc.eval_cost(lp)) <= 1e-10 * abs(costs[1])
    for a, b, c in zip(traces[1], traces[2], traces[0]):
        assert a[0] == b[0] and a[2] == b[2]                        # cost, CG iterations: bit for bit
        assert abs(a[0] - c[0]) <= 1e-12 * abs(c[0]) and a[2] == c[2]
    for mode in (1, 2, 0):
        assert costs[(mode, 'end')] == traces[mode][-1][0]
    assert np.array_equal(params[1][0], params[2][0]) and np.array_equal(params[1][1], params[2][1])
    assert np.array_equal(params[1][0], params[0][0]) and np.array_equal(params[1][1], params[0][1])   # (the cost steers nothing here)


def test_solves_with_the_fused_cost_pass_and_without():
    """ps_solve and the Python loop over ps_gn_iteration, fuse_cost on / tails only / off: the same cost history (bit for bit
    between the loops and between 1 and 2, 1e-12 against the other summation order), the same iterations, the same parameters."""
    from pyslam_amd.device import DeviceProblem
    from pyslam_amd.problem import device_solve
    lp, _ = synthetic.stereo_ba(num_kf=60, num_lm=6000, obs_per_lm=8, half_window=10, seed=7)
    start = (lp.poses.copy(), lp.points.copy())
    out = {}
    for mode in (1, 2, 0):
        for core in (True, False):
            dev = DeviceProblem(lp)
            dev.set_option('fuse_cost', mode)
            for rep in range(2):                                    # (a second solve on the live handle: nothing stale survives)
                dev.reset_solver_state(); dev.set_params(*start)
                hist, stats = device_solve(dev, example_options(), use_core_loop=core)
            out[(mode, core)] = (hist, [s[0] for s in stats], dev.get_params())
            dev.close()
    ref = out[(1, True)]
    assert len(ref[0]) >= 4
    for key in ((1, False), (2, True), (2, False)):
        assert out[key][0] == ref[0] and out[key][1] == ref[1]
        assert np.array_equal(out[key][2][0], ref[2][0]) and np.array_equal(out[key][2][1], ref[2][1])
    for key in ((0, True), (0, False)):
        assert out[key][1] == ref[1] and len(out[key][0]) == len(ref[0])
        assert np.allclose(out[key][0], ref[0], rtol=1e-12, atol=0.0)
        assert np.abs(out[key][2][0] - ref[2][0]).max() <= 1e-9 and np.abs(out[key][2][1] - ref[2][1]).max() <= 1e-8
    _, info = orc.solve(lp, dict(allow_nondecreasing_steps=True, max_nondecreasing_steps=3), points_first=False)
    assert len(info['cost_history']) == len(ref[0]) and np.allclose(info['cost_history'], ref[0], rtol=1e-9)


def test_a_failure_found_ahead_belongs_to_the_call_that_linearises_there():
    """Vanishing Tukey weights leave H_ll = 0 (tests/test_gpu_edges.py).  ps_eval_cost runs the landmark pass of that point and
    must NOT raise -- the cost is fine; the whole-iteration call that takes the pass over must raise and apply nothing.  The same
    through ps_solve, whose start cost rides on the first landmark pass."""
    from pyslam_amd._native import NativeError
    from pyslam_amd.device import DeviceProblem
    from pyslam_amd.problem import device_solve
    lp, _ = synthetic.stereo_ba(num_kf=30, num_lm=1200, obs_per_lm=4, half_window=5, seed=2, loss=losses.TukeyLoss(1e-9))
    dev = DeviceProblem(lp)
    c = dev.eval_cost(True)
    assert np.isfinite(c) and abs(c - orc.eval_cost(lp)) <= 1e-10 * abs(c)
    before = dev.get_params()
    with pytest.raises(NativeError, match='landmark block'):
        dev.gn_iteration(0., 1e-12, 500, True)
    after = dev.get_params()
    assert np.array_equal(before[0], after[0]) and np.array_equal(before[1], after[1])
    with pytest.raises(NativeError, match='landmark block'):
        device_solve(dev, example_options())
    dev.close()
    # ... and a healthy problem on a handle that has seen the failure word stamped does not inherit it
    lp2, _ = synthetic.stereo_ba(num_kf=30, num_lm=1200, obs_per_lm=4, half_window=5, seed=2)
    dev = DeviceProblem(lp2)
    hist, _ = device_solve(dev, example_options())
    assert hist[-1] < hist[0]
    dev.close()


def test_parameters_replaced_after_a_fused_tail_are_linearised_afresh():
    """The pass run ahead is of the point the tail left: set_params / restore in between must invalidate it."""
    from pyslam_amd.device import DeviceProblem
    lp, _ = synthetic.stereo_ba(num_kf=40, num_lm=3000, obs_per_lm=10, half_window=8, seed=2)
    start = (lp.poses.copy(), lp.points.copy())
    dev = DeviceProblem(lp)
    dev.set_option('lagged_inverse', 0)
    dev.set_expect_next(True)
    first = dev.gn_iteration(0.0, 1e-12, 2000, True)
    dev.set_params(*start)                                          # back to the start: the pass run ahead is of another point
    again = dev.gn_iteration(0.0, 1e-12, 2000, True)
    # (the second call's coarse level is the lagged one: the same step to the CG's tolerance, not to the bit)
    assert abs(again[0] - first[0]) <= 1e-9 * first[0] and abs(again[1] - first[1]) <= 1e-7 * first[1]
    dev.snapshot()
    third = dev.gn_iteration(0.0, 1e-12, 2000, True)
    dev.restore()
    fourth = dev.gn_iteration(0.0, 1e-12, 2000, True)
    assert abs(fourth[0] - third[0]) <= 1e-9 * third[0] and abs(fourth[1] - third[1]) <= 1e-6 * third[1]
    dev.close()
