"""Pins the numpy oracle (oracle/gn_oracle.py) against golden vectors produced
by the verbatim reference (oracle/gen_golden.py).  CPU only."""
import numpy as np
import os
import pytest

from conftest import load_golden, REPO, golden_lp, golden_options, SOLVE_CASES, rel_err
from oracle import gn_oracle as orc


@pytest.mark.parametrize('name', SOLVE_CASES)
def test_first_iteration_normal_equations(name):
    g = load_golden(name)
    lp = golden_lp(g)
    pf = bool(g.get('points_first', True))
    P, b, cost = orc.normal_equations(lp, pf)
    assert rel_err(b, g['information']) < 1e-12
    assert abs(cost - float(g['lin_cost'])) <= 1e-12 * abs(float(g['lin_cost']))
    x = np.random.default_rng(7).standard_normal((P.shape[0], 3))
    assert rel_err(P.dot(x), g['probe']) < 1e-12
    if 'precision' in g:
        assert rel_err(P.toarray(), g['precision']) < 1e-12
    assert abs(orc.eval_cost(lp) - g['cost_history'][0]) <= 1e-12 * g['cost_history'][0]


@pytest.mark.parametrize('name', SOLVE_CASES)
def test_first_iteration_dx(name):
    g = load_golden(name)
    lp = golden_lp(g)
    pf = bool(g.get('points_first', True))
    dx, _ = orc.gauss_newton_step(lp, pf)
    assert rel_err(dx, g['iter_dx'][0]) < 1e-9
    dx_s, _ = orc.gauss_newton_step(lp, pf, linear_solver='schur')
    assert rel_err(dx_s, g['iter_dx'][0]) < 1e-8


@pytest.mark.parametrize('name', SOLVE_CASES)
def test_solve_trace(name):
    g = load_golden(name)
    lp = golden_lp(g)
    pf = bool(g.get('points_first', True))
    final, trace = orc.solve(lp, golden_options(g), pf)
    ref = g['cost_history']
    assert len(trace['cost_history']) == len(ref)
    # costs that have collapsed to rounding noise (1e-20 and below) are not comparable
    big = ref > 1e-9 * ref[0]
    assert np.allclose(trace['cost_history'][big], ref[big], rtol=1e-7)
    if 'final_poses' in g:
        assert np.abs(final.poses - g['final_poses']).max() < 1e-8
    if 'final_points' in g:
        n = g['final_points'].shape[0]
        assert np.abs(final.points[:n] - g['final_points']).max() < 1e-7


def test_blocks_known_answers():
    g = load_golden('blocks')
    from pyslam_amd.lowering import LoweredProblem, pack_pose_matrices
    n = g['rpn_T'].shape[0]
    lp = LoweredProblem(dof=6, poses=pack_pose_matrices(g['rpn_T']), pose_rid=np.arange(n),
                        points=g['rpn_pt'], point_vid=np.arange(n),
                        obs_pose=np.arange(n), obs_point=np.arange(n), obs_uvd=g['rpn_obs'],
                        cams=[[640., 480., 1000., 1000., 0.25]], stiff3=g['rpn_S'].reshape(1, 9),
                        obs_groups=[[0, 0, 0, 0]]).finalize()
    r, Jp, Jl = orc.eval_reproj(lp)
    assert rel_err(r, g['rpn_r']) < 1e-13
    assert rel_err(Jp, g['rpn_Jpose']) < 1e-13
    assert rel_err(Jl, g['rpn_Jpt']) < 1e-13
    for dof in (6, 3):
        t = 'pp{}_'.format(dof)
        m = g[t + 'T1'].shape[0]
        Tobs_inv = np.linalg.inv(g[t + 'Tobs'])
        lp = LoweredProblem(dof=dof, poses=pack_pose_matrices(np.concatenate([g[t + 'T1'], g[t + 'T2']])),
                            pose_rid=np.arange(2 * m), e_i=np.arange(m), e_j=m + np.arange(m),
                            e_Tobs_inv=pack_pose_matrices(Tobs_inv), u_i=m + np.arange(m),
                            u_Tobs_inv=pack_pose_matrices(Tobs_inv), u_grp=np.zeros(m),
                            stiffd=g[t + 'S'].reshape(1, -1), edge_groups=[[0, 0, 0]]).finalize()
        r, J1, J2 = orc.eval_edges(lp)
        assert rel_err(r, g[t + 'r'], 1e-3) < 1e-11
        assert rel_err(J1, g[t + 'J1']) < 1e-12
        assert rel_err(J2, g[t + 'J2']) < 1e-13
        rp, _ = orc.eval_priors(lp)
        assert rel_err(rp, g[t + 'r_prior']) < 1e-11


def test_losses_tables():
    g = load_golden('losses_sensors')
    x = g['x']
    for lid, (name, k) in enumerate([('l2', 0.), ('l1', 0.), ('cauchy', 3.), ('huber', 1.5),
                                     ('tukey', 3.), ('tdist', 5.)]):
        assert np.allclose(orc.loss_rho(lid, k, x), g[name + '_loss'], rtol=1e-14, atol=0, equal_nan=True)
        assert np.allclose(orc.loss_weight(lid, k, x), g[name + '_weight'], rtol=1e-14, atol=0, equal_nan=True)


def test_fuzz_api_case_701050_is_not_determined_by_its_inputs():
    """tests/fuzz_api.py case 701050 (round-1 VERDICT): device and oracle cost histories agree to 1e-11, 3e-10, 7e-4 over
    three iterations of a solve restarted at a converged state.  Diagnosis, pinned here on the tables the failing solve
    started from: the ORACLE's own history moves by as much when its landmarks are perturbed by 1e-13 (relative) --
    undamped Gauss-Newton oscillating about a minimum, normal matrix condition number 2.5e14 -- so the third iterate is
    not a quantity two implementations can be compared on.  (tools/diag_fuzz_api.py prints the whole analysis.)"""
    from pyslam_amd.lowering import LoweredProblem
    g = load_golden('fuzz_api_case_701050')
    lp = LoweredProblem(dof=int(g['lp_dof']))
    for k, v in g.items():
        if k.startswith('lp_') and k != 'lp_dof':
            setattr(lp, k[3:], np.array(v))
    lp = lp.finalize()
    opts = dict(allow_nondecreasing_steps=True, max_nondecreasing_steps=3, linesearch_max_iters=10, max_iters=20,
                min_cost_decrease=0.99)
    pf = bool(g['pf'])
    _, a = orc.solve(lp, opts, points_first=pf)
    pert = lp.copy()
    pert.points = pert.points * (1. + 1e-13 * np.random.default_rng(1).standard_normal(pert.points.shape))
    _, b = orc.solve(pert, opts, points_first=pf)
    ha, hb, hd = np.asarray(a['cost_history']), np.asarray(b['cost_history']), g['device_history']
    assert len(ha) == len(hb) == len(hd) == 4
    dev = np.abs(hd - ha) / ha                       # device vs oracle (recorded on the MI355X)
    own = np.abs(hb - ha) / ha                       # oracle vs oracle, inputs perturbed by 1e-13
    assert dev[0] < 1e-15 and dev[1] < 1e-10 and dev[2] < 1e-9
    assert own[3] > 1e-4 and own[3] > dev[3]          # the reference's own sensitivity exceeds the discrepancy
    assert own[1] < 1e-9                              # ... and the early iterates ARE comparable (and agree)


def test_pipelined_cg_residual_leaves_the_range_of_the_folded_system():
    """Root cause of the pipelined CG's breakdowns on the folded (singular, consistent) two-level system, emulated in
    numpy (tools/cg_drift.py): in float64 the recurrence residual drifts out of range(V^T) -- its coarse part stops
    being X^T times its fine part -- by the time the true residual has dropped ten orders; 80-bit arithmetic keeps the
    drift four orders smaller, restoring the invariant removes it.  The device answers a breakdown by restarting from
    the true residual, which IS that projection."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('cg_drift', os.path.join(REPO, 'tools', 'cg_drift.py'))
    cgd = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cgd)
    lp, H = cgd.build(0, P=120)
    S, Linv, X = cgd.setup(H, lp.dof, 12)
    e = np.zeros(H.shape[0]); e[60 * lp.dof] = 1.
    bf = Linv @ e
    f64 = cgd.cg_cgear(S, X, bf, np.float64)
    prj = cgd.cg_cgear(S, X, bf, np.float64, fix='project')
    f80 = cgd.cg_cgear(S, X, bf, np.longdouble)
    assert f64[-1][2] > 1e-3 and f64[-1][2] > 100 * f80[-1][2]      # O(1e-2..1) in double, orders less in long double
    assert prj[-1][2] == 0. and prj[-1][3] < 1e-12                  # projected: invariant exact, converged
    assert f64[-1][3] < 1e-11                                       # (this column still converges: drift, not yet breakdown)
