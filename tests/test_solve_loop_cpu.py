"""device_solve (pyslam_amd/problem.py) -- the Python statement of Problem.solve's loop (reference pyslam/problem.py:130-178) -- on a
scripted device: the stopping rules, the best-parameter bookkeeping and what the loop tells the core about its own future
(set_solve_horizon, set_expect_next: the rules ps_solve applies in C, csrc/ps_abi_solver.h).  No GPU."""
import pytest

from pyslam_amd.problem import Options, device_solve, solve_horizon


class ScriptedDevice:
    """Returns a prescribed cost sequence; records every call."""

    def __init__(self, costs, dx_norm=1.0):
        self.costs, self.dx_norm, self.k, self.log = list(costs), dx_norm, 0, []

    def reset_solver_state(self):
        self.log.append(('reset',))

    def eval_cost(self, include_all=True):
        self.log.append(('eval',))
        return self.costs[0]

    def set_solve_horizon(self, n):
        self.log.append(('horizon', n))

    def set_expect_next(self, flag):
        self.log.append(('expect', bool(flag)))

    def gn_iteration(self, lam, tol, max_iters, linesearch):
        self.k += 1
        self.log.append(('iter', self.k))
        return self.costs[self.k], self.dx_norm, 7, 1e-13

    def snapshot(self):
        self.log.append(('snapshot', self.k))

    def restore(self):
        self.log.append(('restore',))


def reference_loop(costs, opt, dx_norm=1.0):
    """The reference's loop on the same cost sequence: (history, iteration at which best_params was taken or None)."""
    cost, hist, it, nd, best, done = costs[0], [costs[0]], 0, 0, None, False
    while not done:
        it += 1
        prev, cost = cost, costs[it]
        hist.append(cost)
        done = it > opt.max_iters or dx_norm < opt.min_update_norm or cost < opt.min_cost
        if opt.allow_nondecreasing_steps:
            if nd == 0:
                best = it
            nd = nd + 1 if cost >= opt.min_cost_decrease * prev else 0
            if nd >= opt.max_nondecreasing_steps:
                done = True
        else:
            done = done or cost >= opt.min_cost_decrease * prev
    restored = opt.allow_nondecreasing_steps and nd >= opt.max_nondecreasing_steps
    return hist, (best if restored else None)


CASES = [
    (dict(allow_nondecreasing_steps=True, max_nondecreasing_steps=3), [100., 10., 9.99, 9.989, 9.9889, 9.98889]),     # examples/stereo_ba.py
    (dict(), [100., 10., 9.99, 9.98]),                                                                               # defaults: first bad step ends it
    (dict(allow_nondecreasing_steps=True, max_nondecreasing_steps=2, max_iters=3), [100., 50., 20., 5., 1., 0.5, 0.1]),
    (dict(allow_nondecreasing_steps=True, max_nondecreasing_steps=3), [100., 99., 10., 9.9, 1.0, 0.99, 0.98, 0.97]),  # the count restarts
    (dict(min_cost=50.), [100., 10., 1.]),
]


@pytest.mark.parametrize('kw,costs', CASES)
def test_python_loop_follows_the_reference_rules_and_announces_its_next_call(kw, costs):
    opt = Options()
    for k, v in kw.items():
        setattr(opt, k, v)
    dev = ScriptedDevice(costs)
    hist, stats = device_solve(dev, opt, use_core_loop=False)
    want, best_at = reference_loop(costs, opt)
    assert hist == want and len(stats) == len(hist) - 1
    snaps = [e[1] for e in dev.log if e[0] == 'snapshot']
    assert (('restore',) in dev.log) == (best_at is not None)
    if best_at is not None:
        assert snaps[-1] == best_at                      # what is restored is the last snapshot: taken where the reference takes best_params
    # before every iteration: the horizon, then whether another iteration will follow -- the rule of ps_solve:
    #   it <= max_iters and (allow_nondecreasing ? horizon >= 1 : (it >= 2 and last cost ratio < 0.5)); switched off at the end
    calls = [e for e in dev.log if e[0] in ('horizon', 'expect', 'iter')]
    nd, ratio = 0, 1.0
    for it in range(1, len(hist)):
        h = solve_horizon(opt, it, nd)
        expect = it <= opt.max_iters and (h >= 1 if opt.allow_nondecreasing_steps else (it >= 2 and ratio < 0.5))
        assert calls[3 * (it - 1):3 * it] == [('horizon', h), ('expect', expect), ('iter', it)]
        ratio = hist[it] / hist[it - 1]
        if opt.allow_nondecreasing_steps:
            nd = nd + 1 if hist[it] >= opt.min_cost_decrease * hist[it - 1] else 0
    assert len(calls) == 3 * (len(hist) - 1) + 1 and calls[-1] == ('expect', False)


def test_solve_horizon_table():
    opt = Options()
    assert solve_horizon(opt, 1, 0) == 0                                   # without allow_nondecreasing_steps nothing is promised
    opt.allow_nondecreasing_steps, opt.max_nondecreasing_steps, opt.max_iters = True, 3, 100
    assert [solve_horizon(opt, 1, nd) for nd in (0, 1, 2)] == [2, 1, 0]
    opt.max_iters = 2
    assert [solve_horizon(opt, it, 0) for it in (1, 2, 3)] == [2, 1, 0]   # the loop stops once its counter exceeds max_iters
