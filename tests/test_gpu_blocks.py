"""GPU parity, block level (round-2 VERDICT "parity pinholes"): the pose-factor kernel's own r~, J~_1, J~_2
against the reference-run fixtures, and config C5 at the one-launch kernel's limit of 2 048 observations per pose
and just above it (where the general multi-kernel path takes over)."""
import numpy as np
import pytest

from conftest import load_golden, rel_err
from oracle import gn_oracle as orc

pytestmark = pytest.mark.gpu

TOL_BLOCK = 1e-12      # r~, J~ blocks


def _factor_problem(g, dof):
    from pyslam_amd.lowering import LoweredProblem, pack_pose_matrices
    t = 'pp{}_'.format(dof)
    m = g[t + 'T1'].shape[0]
    Tobs_inv = np.linalg.inv(g[t + 'Tobs'])
    return LoweredProblem(dof=dof, poses=pack_pose_matrices(np.concatenate([g[t + 'T1'], g[t + 'T2']])),
                          pose_rid=np.arange(2 * m), e_i=np.arange(m), e_j=m + np.arange(m),
                          e_Tobs_inv=pack_pose_matrices(Tobs_inv), u_i=m + np.arange(m),
                          u_Tobs_inv=pack_pose_matrices(Tobs_inv), u_grp=np.zeros(m),
                          stiffd=g[t + 'S'].reshape(1, -1), edge_groups=[[0, 0, 0]]).finalize(), m


@pytest.mark.parametrize('dof', [6, 3])
def test_pose_factor_blocks_match_the_reference_fixtures(dof):
    """k_factor_pass (edges + priors) vs tests/golden/blocks.npz `pp6_*` / `pp3_*`: PoseToPoseResidual.evaluate and
    PoseResidual.evaluate of the verbatim reference (pyslam/residuals/pose_to_pose_residual.py:12-32,
    pose_residual.py:12-27) on 12 pose pairs incl. relative rotations of 1e-10 (the `np.isclose(angle, 0)` branches of
    log and of the inverse left Jacobian) -- L2 loss, so the IRLS scale is 1 and the blocks are the reference's own."""
    from pyslam_amd.device import DeviceProblem
    g = load_golden('blocks')
    lp, m = _factor_problem(g, dof)
    t = 'pp{}_'.format(dof)
    dev = DeviceProblem(lp)
    r, j1, j2 = dev.debug_factor_blocks()
    assert r.shape == (2 * m, dof)
    # device vs oracle: the bar of SURVEY 8d
    ro, j1o, j2o = orc.eval_edges(lp)
    rpo, jpo = orc.eval_priors(lp)
    assert rel_err(r[:m], ro) < TOL_BLOCK and rel_err(j1[:m], j1o) < TOL_BLOCK and rel_err(j2[:m], j2o) < TOL_BLOCK
    assert rel_err(r[m:], rpo) < TOL_BLOCK and rel_err(j2[m:], jpo) < TOL_BLOCK
    assert np.all(j1[m:] == 0.)
    # device vs the reference's blocks.  Residuals: per pair, relative to |r| with the floor the oracle's own pin uses
    # (tests/test_oracle.py: pairs whose relative pose differs from the measurement by 1e-10 have |r| ~ 1e-9, where the
    # product T_2 T_1^-1 T_obs^-1 itself carries 1e-16 absolute rounding)
    assert rel_err(r[:m], g[t + 'r'], 1e-3) < TOL_BLOCK
    assert rel_err(r[m:], g[t + 'r_prior'], 1e-3) < TOL_BLOCK
    assert rel_err(j1[:m], g[t + 'J1']) < TOL_BLOCK
    assert rel_err(j2[:m], g[t + 'J2']) < TOL_BLOCK
    for k in range(m):                                       # every pair on its own as well, small-angle ones included
        assert rel_err(j1[k], g[t + 'J1'][k]) < TOL_BLOCK, k
        assert np.abs(r[k] - g[t + 'r'][k]).max() <= 1e-12 * max(1., np.abs(g[t + 'r'][k]).max()), k
    # the tap must not disturb the solver state: a linearisation afterwards gives the oracle's system
    dev.linearize(0.)
    S, gg = dev.reduced_dense()
    P, b, _ = orc.normal_equations(lp, points_first=False)
    assert rel_err(S, P.toarray()) < 1e-11 and rel_err(gg, b) < 1e-11


def test_pose_factor_blocks_carry_the_irls_scale():
    """Huber loss on a pose graph: the tap returns J~ = diag(sqrt(w)) J and r~ = sqrt(w) r (reference problem.py:351-360)."""
    from pyslam_amd.device import DeviceProblem
    from conftest import golden_lp
    lp = golden_lp(load_golden('pg_small_huber'))
    dev = DeviceProblem(lp)
    r, j1, j2 = dev.debug_factor_blocks()
    E = lp.num_edges
    ro, j1o, j2o = orc.eval_edges(lp)
    s = np.sqrt(orc._by_group(lp.edge_groups, lp.e_grp, 1, 2, orc.loss_weight, ro))
    assert (s < 1.).any()                                    # the loss is active somewhere
    assert rel_err(r[:E], s * ro) < TOL_BLOCK
    assert rel_err(j1[:E], s[:, :, None] * j1o) < TOL_BLOCK
    assert rel_err(j2[:E], s[:, :, None] * j2o) < TOL_BLOCK
    rpo, jpo = orc.eval_priors(lp)
    sp = np.sqrt(orc._by_group(lp.edge_groups, lp.u_grp, 1, 2, orc.loss_weight, rpo))
    assert rel_err(r[E:], sp * rpo, 1e-6) < TOL_BLOCK and rel_err(j2[E:], sp[:, :, None] * jpo) < TOL_BLOCK


C5_OPTIONS = dict(allow_nondecreasing_steps=True, max_nondecreasing_steps=5, min_cost_decrease=0.99, max_iters=30,
                  linesearch_max_iters=0)     # reference pipelines/sparse.py:34-39


@pytest.mark.parametrize('num_pts', [2048, 2049])
def test_c5_at_and_above_the_one_launch_limit(num_pts):
    """BASELINE config 5 at N = 2 048 (the documented limit of k_motion_only_iteration: one workgroup per pose, at most
    2 048 observations each) and at 2 049 (the general landmark-free path).  Cauchy loss, 20 % outliers.  Device step vs
    the oracle's step; whole solve() through the public API (one ReprojectionMotionOnlyBatchResidual block, the
    pipeline's options) vs the oracle's solve: iteration count, cost history, final pose."""
    from pyslam_amd import synthetic
    from pyslam_amd.device import DeviceProblem
    from pyslam_amd.lowering import pack_pose
    lp, aux = synthetic.motion_only(num_pts=num_pts, seed=3)
    assert lp.num_obs == num_pts
    dev = DeviceProblem(lp)
    c0 = dev.eval_cost(True)
    assert abs(c0 - orc.eval_cost(lp)) <= 1e-10 * abs(c0)
    dx_ref, lin_cost = orc.gauss_newton_step(lp, points_first=False)
    for linesearch in (False, True):
        dev.set_params(lp.poses, None)
        cost, nrm, its, rel = dev.gn_iteration(0., 1e-12, 100, linesearch)
        xp, _ = dev.get_dx()
        assert rel_err(xp.ravel(), dx_ref) < 1e-8
        assert abs(nrm - np.linalg.norm(dx_ref)) <= 1e-8 * np.linalg.norm(dx_ref)
        want = orc.eval_cost(orc.apply_update(lp, dx_ref, False)) if linesearch else lin_cost
        assert abs(cost - want) <= 1e-10 * abs(want)
    # the same step from the other kernel path
    other = DeviceProblem(lp)
    other.set_option('fused_motion_only', 0)
    other.gn_iteration(0., 1e-12, 100, False)
    assert rel_err(other.get_dx()[0].ravel(), dx_ref) < 1e-8

    # whole solve through the reference's API
    from liegroups import SE3
    from pyslam.problem import Options, Problem
    from pyslam.sensors import StereoCamera
    from pyslam.residuals import ReprojectionMotionOnlyBatchResidual
    from pyslam.losses import CauchyLoss
    opt = Options()
    for k, v in C5_OPTIONS.items():
        setattr(opt, k, v)
    problem = Problem(opt)
    cam = StereoCamera(*synthetic.STEREO_BA_CAMERA)
    problem.add_residual_block(
        ReprojectionMotionOnlyBatchResidual(cam, aux['obs_1'], aux['obs_2'], lp.stiff3[0].reshape(3, 3)),
        ['T_2_1'], CauchyLoss(3.0))
    problem.initialize_params({'T_2_1': SE3.identity()})
    out = problem.solve()
    final, ref = orc.solve(lp, C5_OPTIONS, points_first=False)
    hist = np.array(problem._cost_history)
    assert len(hist) == len(ref['cost_history']), (hist, ref['cost_history'])
    assert np.allclose(hist, ref['cost_history'], rtol=1e-10)
    assert np.abs(pack_pose(out['T_2_1']) - final.poses[0]).max() < 1e-9
    # and the solve found the motion the scene was rendered with (outliers rejected by the Cauchy weights)
    assert np.abs(out['T_2_1'].as_matrix() - aux['poses'][0]).max() < 5e-3
