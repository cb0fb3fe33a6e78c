"""ps_solve (the loop of Problem.solve in the core) against the Python loop over ps_gn_iteration it restates
(pyslam_amd/problem.py: device_solve, reference pyslam/problem.py:130-178), and ps_reset_solver_state: a solve on a used
handle equals the solve on a fresh one, bit for bit."""
import numpy as np
import pytest

from pyslam_amd import synthetic
from pyslam_amd.problem import Options, device_solve

pytestmark = pytest.mark.gpu


def _options(**kw):
    opt = Options()
    opt.pcg_tol = 1e-12
    for k, v in kw.items():
        setattr(opt, k, v)
    return opt


CASES = [
    dict(),                                                                   # the reference's defaults: stops at the first step that does not cut the cost by 10 %
    dict(allow_nondecreasing_steps=True, max_nondecreasing_steps=3),          # examples/stereo_ba.py:38-40
    dict(allow_nondecreasing_steps=True, max_nondecreasing_steps=1),
    dict(max_iters=2, allow_nondecreasing_steps=True, max_nondecreasing_steps=5),
    dict(max_iters=0),
    dict(linesearch_max_iters=0, allow_nondecreasing_steps=True, max_nondecreasing_steps=2),
    dict(min_cost=1e30),                                                      # stops on min_cost after one iteration
    dict(min_update_norm=1e30),
    dict(lm_lambda=1e-3, allow_nondecreasing_steps=True, max_nondecreasing_steps=2),
]


def _problems():
    yield 'ba', synthetic.stereo_ba(num_kf=24, num_lm=600, obs_per_lm=6, half_window=5, seed=4)[0]
    yield 'ba_small', synthetic.stereo_ba(num_kf=6, num_lm=80, obs_per_lm=4, half_window=3, seed=2)[0]
    yield 'pg', synthetic.pose_graph(num_poses=60, num_loops=90, dof=6, seed=3)[0]
    yield 'pg2', synthetic.pose_graph(num_poses=40, num_loops=50, dof=3, seed=5)[0]


@pytest.mark.parametrize('name,lp', list(_problems()), ids=lambda v: v if isinstance(v, str) else '')
def test_core_loop_equals_the_python_loop(name, lp):
    from pyslam_amd.device import DeviceProblem
    for kw in CASES:
        opt = _options(**kw)
        dev = DeviceProblem(lp)
        h_c, s_c = device_solve(dev, opt, use_core_loop=True)
        p_c = dev.get_params()
        dev.close()
        dev = DeviceProblem(lp)
        h_p, s_p = device_solve(dev, opt, use_core_loop=False)
        p_p = dev.get_params()
        dev.close()
        assert len(h_c) == len(h_p), (name, kw, h_c, h_p)
        # (the start cost is the same kernel's sum in both; every iteration is the same ps_gn_iteration)
        assert h_c == h_p, (name, kw, h_c, h_p)
        assert [a for a, _ in s_c] == [a for a, _ in s_p]
        assert np.array_equal(p_c[0], p_p[0]) and np.array_equal(p_c[1], p_p[1])


def test_a_solve_on_a_used_handle_equals_the_solve_on_a_fresh_one():
    """ps_reset_solver_state: lagged coarse operators, the lagged inverse, predictions and cost history of earlier calls leave no
    trace in the next solve (reference problem.py:130-141: every solve starts from scratch)."""
    from pyslam_amd.device import DeviceProblem
    lp, _ = synthetic.stereo_ba(num_kf=40, num_lm=1500, obs_per_lm=8, half_window=6, seed=9)
    opt = _options(allow_nondecreasing_steps=True, max_nondecreasing_steps=3)
    fresh = DeviceProblem(lp)
    want_h, want_s = device_solve(fresh, opt)
    want_p = fresh.get_params()
    fresh.close()
    used = DeviceProblem(lp)
    for _ in range(7):                                       # a history: iterations that settle, seed and use the lagged operators
        used.gn_iteration(0.0, 1e-12, 2000, True)
    used.set_params(lp.poses, lp.points)
    got_h, got_s = device_solve(used, opt)
    got_p = used.get_params()
    assert got_h == want_h and got_s == want_s
    assert np.array_equal(got_p[0], want_p[0]) and np.array_equal(got_p[1], want_p[1])
    # ... and once more on the same handle, through the Python loop
    used.set_params(lp.poses, lp.points)
    again_h, again_s = device_solve(used, opt, use_core_loop=False)
    assert again_h == want_h and [a for a, _ in again_s] == [a for a, _ in want_s]
    used.close()


def test_a_linearisation_enqueued_ahead_does_not_survive_a_staged_tail():
    """ps_solve may leave the NEXT linearisation in the queue (enqueued behind a converged tail while the host waited).  Anything
    that moves the parameters afterwards -- here the staged tail ps_gn_finish, which applies the last step once more -- must
    invalidate it, or the next ps_gn_iteration at the same damping solves a stale system (round-4 ADVICE).  Held against a
    handle on which set_params forces the fresh linearisation."""
    from pyslam_amd.device import DeviceProblem
    lp, _ = synthetic.stereo_ba(num_kf=40, num_lm=1500, obs_per_lm=8, half_window=6, seed=9)
    opt = _options(max_iters=2, allow_nondecreasing_steps=True, max_nondecreasing_steps=5)
    outs = []
    for force in (False, True):
        dev = DeviceProblem(lp)
        device_solve(dev, opt)
        dev.gn_finish(True)                                  # moves the parameters (the last dx again), no linearize in front
        if force:
            dev.set_params(*dev.get_params())                # the same values: clears whatever was linearised ahead
        outs.append((dev.gn_iteration(0.0, 1e-12, 2000, True), dev.get_params()))
        dev.close()
    (a, pa), (b, pb) = outs
    assert a[0] == b[0] and a[1] == b[1]
    assert np.array_equal(pa[0], pb[0]) and np.array_equal(pa[1], pb[1])
    # round 6: once a landmark pass has run at the moved point (ps_eval_cost sums the cost in that pass and rewrites Z, C^-1, c) the
    # staged tail of the OLD linearisation is refused instead of back-substituting with another point's buffers
    from pyslam_amd._native import NativeError
    dev = DeviceProblem(lp)
    device_solve(dev, opt)
    dev.eval_cost(True)
    with pytest.raises(NativeError, match='no longer belong'):
        dev.gn_finish(True)
    dev.close()


def test_build_sha_matches_the_sources_on_disk():
    import __graft_entry__ as ge
    from pyslam_amd import _native as nat
    assert nat.load().ps_build_sha().decode() == ge.source_sha()


def test_the_solves_final_restore_consumes_the_snapshot():
    """ps_solve hands the best parameters back by exchanging the parameter tables with the snapshot's (no copy: nothing is left
    for the caller's synchronisation to wait for).  The result equals the Python loop's copy-restore (the parametrised test above);
    here: the snapshot is gone afterwards -- a restore without a new snapshot fails loudly instead of bringing back the iterate
    that was given up -- and a snapshot taken after the solve works as before."""
    from pyslam_amd._native import NativeError
    from pyslam_amd.device import DeviceProblem
    lp, _ = synthetic.stereo_ba(num_kf=30, num_lm=1500, obs_per_lm=6, half_window=6, seed=5)
    dev = DeviceProblem(lp)
    hist, _ = device_solve(dev, _options(allow_nondecreasing_steps=True, max_nondecreasing_steps=3))
    assert len(hist) >= 4                                    # (ended on max_nondecreasing_steps: the restore ran)
    best = dev.get_params()
    with pytest.raises(NativeError, match='no snapshot'):
        dev.restore()
    assert np.array_equal(dev.get_params()[0], best[0]) and np.array_equal(dev.get_params()[1], best[1])
    dev.snapshot()
    dev.gn_iteration(0., 1e-12, 500, True)
    dev.restore()
    after = dev.get_params()
    assert np.array_equal(after[0], best[0]) and np.array_equal(after[1], best[1])
    # (what the reference keeps as best_params, problem.py:163-178: the parameters of the iteration at whose end the count of
    #  non-decreasing steps was still zero -- i.e. after the FIRST step that failed to cut the cost by min_cost_decrease; not
    #  the lowest cost seen)
    k_last_good = max(i for i in range(1, len(hist)) if hist[i] < 0.9 * hist[i - 1])
    assert dev.eval_cost(True) == hist[k_last_good + 1]
    dev.close()
