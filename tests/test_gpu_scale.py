"""GPU checks at (or near) the BASELINE sizes through size-independent properties:
the device step must satisfy the ORACLE's normal equations, costs must agree, landmark
shards must sum to the unsharded reduced system, results must be bitwise reproducible."""
import numpy as np
import pytest

from oracle import gn_oracle as orc
from pyslam_amd import synthetic
from pyslam_amd.distributed import shard_landmarks, pose_pair_keys

pytestmark = pytest.mark.gpu


def device(lp, **kw):
    from pyslam_amd.device import DeviceProblem
    return DeviceProblem(lp, **kw)


def device_dx_posefirst(dev):
    xp, xl = dev.get_dx()
    return np.concatenate([xp.ravel(), xl.ravel()])


@pytest.fixture(scope='module')
def c3():
    """BASELINE config C3: 200 keyframes, 50 000 landmarks, 500 000 reprojection blocks."""
    lp, _ = synthetic.stereo_ba(200, 50000, 10, 20, seed=0)
    return lp


def test_c3_step_solves_the_reference_normal_equations(c3):
    dev = device(c3)
    c0 = dev.eval_cost(True)
    assert abs(c0 - orc.eval_cost(c3)) <= 1e-11 * c0
    dev.linearize(0.)
    its, rel = dev.solve_reduced(1e-12, 2000)
    dev.backsub()
    dx = device_dx_posefirst(dev)
    P, b, lin_cost = orc.normal_equations(c3, points_first=False)        # reference algebra (oracle)
    res = P.dot(dx) - b
    assert np.linalg.norm(res) <= 1e-9 * np.linalg.norm(b), (its, rel)
    # block-wise too: the landmark part is an exact (direct) solve, the pose part is PCG
    n = 6 * c3.num_reduced
    assert np.linalg.norm(res[n:]) <= 1e-11 * np.linalg.norm(b[n:])
    assert abs(dev.eval_cost(False) - lin_cost) <= 1e-11 * lin_cost
    # applying the step reproduces the oracle's post-step cost
    dev.apply_update(1.0)
    c1 = dev.eval_cost(True)
    assert abs(c1 - orc.eval_cost(orc.apply_update(c3, dx, points_first=False))) <= 1e-9 * c1
    assert c1 < 0.05 * c0


def test_c3_gn_iteration_matches_staged_calls_bitwise(c3):
    a, b = device(c3), device(c3)
    cost, nrm, its, rel = a.gn_iteration(0., 1e-12, 2000, True)
    b.linearize(0.); b.solve_reduced(1e-12, 2000)
    cost2, p2, l2 = b.gn_finish(True)
    assert cost == cost2 and nrm == np.sqrt(p2 + l2)
    pa, la = a.get_params()
    pb, lb = b.get_params()
    assert np.array_equal(pa, pb) and np.array_equal(la, lb)


def test_c3_all_solver_variants_agree(c3):
    ref = None
    # (large-system modes forced onto C3: folded two-level CG in split mode, and the explicitly applied preconditioner)
    for variant, groups, split in ((1, -1, None), (1, 0, None), (0, 0, None), (1, 24, 'folded'), (1, 24, 'explicit')):
        dev = device(c3)
        dev.set_option('pcg_variant', variant)
        dev.set_option('coarse_groups', groups)
        if split:
            dev.set_option('cg_split_min_rows', 0)
            dev.set_option('cg_explicit', float(split == 'explicit'))
        dev.linearize(0.)
        its, rel = dev.solve_reduced(1e-13, 3000)
        dev.backsub()
        dx = device_dx_posefirst(dev)
        if ref is None:
            ref = dx
        else:
            assert np.linalg.norm(dx - ref) <= 1e-9 * np.linalg.norm(ref), (variant, groups, split, its, rel)


def test_lagged_coarse_factor_gives_the_same_trajectory(c3):
    """From the second whole-iteration call on, the two-level CG is built with the coarse factor of
    the PREVIOUS iteration (the current one is factored on a side stream).  The reduced solve must
    still reach the same solution: three Gauss-Newton iterations with and without the lag agree."""
    out = {}
    for lag in (1, 0):
        dev = device(c3)
        dev.set_option('coarse_lag', lag)
        trace = [dev.gn_iteration(0., 1e-12, 2000, True) for _ in range(3)]
        out[lag] = (trace, dev.get_params())
    for (c1, n1, i1, r1), (c0, n0, i0, r0) in zip(out[1][0], out[0][0]):
        assert abs(c1 - c0) <= 1e-9 * abs(c0) and abs(n1 - n0) <= 1e-8 * n0
        assert r1 <= 1e-12 and i1 <= i0 + 8            # a stale factor may cost a few CG iterations, not more
    for a, b in zip(out[1][1], out[0][1]):
        assert np.max(np.abs(a - b)) <= 1e-8


def test_landmark_shards_sum_to_the_unsharded_reduced_system():
    """What the multi-GPU all-reduce relies on, checked on ONE GPU: every shard built with the
    union block pattern, S and g summed on the host, compared with the unsharded system."""
    lp, _ = synthetic.stereo_ba(60, 6000, 8, 10, seed=4)
    full = device(lp)
    full.linearize(0.)
    rp, ci, vals, g = full.reduced_system()
    union = pose_pair_keys(lp)
    acc_v, acc_g, cost = np.zeros_like(vals), np.zeros_like(g), 0.
    world = 4
    for r in range(world):
        sh = shard_landmarks(lp, r, world)
        extra = np.setdiff1d(union, pose_pair_keys(sh))
        dev = device(sh, extra_pairs=((extra >> 32).astype(np.int32), (extra & 0xFFFFFFFF).astype(np.int32)))
        dev.linearize(0.)
        rp2, ci2, v2, g2 = dev.reduced_system()
        assert np.array_equal(rp, rp2) and np.array_equal(ci, ci2)       # identical pattern on every rank
        acc_v += v2; acc_g += g2
        cost += dev.eval_cost(True)
    assert np.abs(acc_v - vals).max() <= 1e-11 * np.abs(vals).max()
    assert np.abs(acc_g - g).max() <= 1e-11 * np.abs(g).max()
    assert abs(cost - full.eval_cost(True)) <= 1e-12 * cost


def test_marquardt_damping_matches_oracle():
    lp, _ = synthetic.stereo_ba(12, 300, 5, 4, seed=9)
    lam = 0.37
    dev = device(lp)
    dev.linearize(lam)
    dev.solve_reduced(1e-13, 500)
    dev.backsub()
    P, b, _ = orc.normal_equations(lp, points_first=False, lm_lambda=lam)
    dx = np.linalg.solve(P.toarray(), b)
    assert np.linalg.norm(device_dx_posefirst(dev) - dx) <= 1e-9 * np.linalg.norm(dx)


def test_marquardt_damping_on_the_small_problem_paths():
    """lambda * diag(J^T J) through the direct (<= 90 unknowns) solve and through the one-launch
    motion-only iteration: the step equals the oracle's damped normal-equation solve."""
    lam = 0.21
    # 11 reduced poses x 6 = 66 unknowns: direct Cholesky path (whole-iteration call)
    lp, _ = synthetic.stereo_ba(12, 300, 5, 4, seed=9)
    dev = device(lp)
    dev.gn_iteration(lam, 1e-13, 500, True)
    P, b, _ = orc.normal_equations(lp, points_first=False, lm_lambda=lam)
    dx = np.linalg.solve(P.toarray(), b)
    assert np.linalg.norm(device_dx_posefirst(dev) - dx) <= 1e-9 * np.linalg.norm(dx)
    # motion-only (all landmarks constant): fused kernel
    lp, _ = synthetic.stereo_ba(6, 200, 4, 3, seed=4, const_point_fraction=1.0)
    dev = device(lp)
    dev.gn_iteration(lam, 1e-13, 500, True)
    P, b, _ = orc.normal_equations(lp, points_first=False, lm_lambda=lam)
    dx = np.linalg.solve(P.toarray(), b)
    assert np.linalg.norm(dev.get_dx()[0].ravel() - dx) <= 1e-10 * np.linalg.norm(dx)


def test_medium_pose_graph_solve_matches_oracle_solve():
    """C2 shape at 1/5 scale: 2 000 SE(3) poses, 10 000 edges, Huber loss, prior on pose 0."""
    lp, _ = synthetic.pose_graph(num_poses=2000, num_loops=8001, dof=6, seed=2)
    opts = dict(allow_nondecreasing_steps=True, max_nondecreasing_steps=3)
    final, trace = orc.solve(lp, opts, points_first=True)
    dev = device(lp)
    hist = [dev.eval_cost(True)]
    non = 0
    for _ in range(30):                      # the reference's termination rules (problem.py:159-178)
        prev = hist[-1]
        cost, nrm, its, rel = dev.gn_iteration(0., 1e-13, 4000, True)
        hist.append(cost)
        if non == 0:
            dev.snapshot()
        non = non + 1 if cost >= 0.9 * prev else 0
        if nrm < 1e-6 or cost < 1e-12 or non >= 3:
            if non >= 3:
                dev.restore()
            break
    ref = trace['cost_history']
    assert len(hist) == len(ref)
    assert np.allclose(hist, ref, rtol=1e-6)
    poses, _ = dev.get_params()
    assert np.abs(poses - final.poses).max() < 1e-6


def test_reference_covariance_identity_se3():
    """Reference tests/test_problem.py:294-321: Sigma_T1 = Sigma_odom + Ad(odom) Sigma_T0 Ad(odom)^T."""
    from liegroups import SE3
    from pyslam.problem import Options, Problem
    from pyslam.residuals import PoseResidual, PoseToPoseResidual
    from pyslam.utils import invsqrt
    options = Options()
    options.allow_nondecreasing_steps = True
    options.max_nondecreasing_steps = 3
    problem = Problem(options)
    odom = SE3.exp(0.1 * np.ones(6))
    odom_stiffness = invsqrt(1e-3 * np.eye(6))
    T0_stiffness = invsqrt(1e-6 * np.eye(6))
    odom_covar = np.linalg.inv(np.dot(odom_stiffness, odom_stiffness))
    T0_covar = np.linalg.inv(np.dot(T0_stiffness, T0_stiffness))
    problem.add_residual_block(PoseResidual(SE3.identity(), T0_stiffness), 'T0')
    problem.add_residual_block(PoseToPoseResidual(odom, odom_stiffness), ['T0', 'T1'])
    problem.initialize_params({'T0': SE3.identity(), 'T1': SE3.identity()})
    problem.solve()
    problem.compute_covariance()
    expected = odom_covar + odom.adjoint().dot(T0_covar.dot(odom.adjoint().T))
    assert np.allclose(problem.get_covariance_block('T1', 'T1'), expected)
    assert problem.get_covariance_block('T1', 'nope') is None


def test_solve_one_iter_api_does_not_move_parameters():
    from conftest import load_golden, golden_lp
    from test_host_api import build_namespace
    g = load_golden('stereo_ba_example')
    problem = synthetic.to_objects(golden_lp(g), build_namespace())
    before = {k: np.array(v) if isinstance(v, np.ndarray) else v.as_matrix().copy()
              for k, v in problem.param_dict.items()}
    dx, cost = problem.solve_one_iter()
    assert np.linalg.norm(dx - g['iter_dx'][0]) <= 1e-8 * np.linalg.norm(dx)
    assert abs(cost - g['iter_cost'][0]) <= 1e-9 * cost
    for k, v in problem.param_dict.items():
        now = v if isinstance(v, np.ndarray) else v.as_matrix()
        assert np.array_equal(now, before[k])

def test_c3_covariance_columns_solve_the_reference_precision(c3):
    """Covariance columns at BASELINE size (151 194 unknowns): H x = e_k must hold for the oracle's
    (reference algebra) precision matrix H, for a pose and a landmark column."""
    dev = device(c3)
    dev.covariance_begin()
    P, _, _ = orc.normal_equations(c3, points_first=False)
    d, nr = c3.dof, c3.num_reduced
    for kind, index, comp in ((0, 57, 4), (1, 31337, 1)):
        xp, xl = dev.covariance_column(kind, index, comp)
        x = np.concatenate([xp.ravel(), xl.ravel()])
        e = np.zeros_like(x)
        e[(index * d + comp) if kind == 0 else (nr * d + index * 3 + comp)] = 1.
        res = P.dot(x) - e
        # scale-free check: the residual measured in the norm of the diagonal of H
        dg = np.sqrt(P.diagonal())
        assert np.linalg.norm(res / dg) <= 1e-8 * np.linalg.norm(x * dg), (kind, index)
        k = (index * d + comp) if kind == 0 else (nr * d + index * 3 + comp)
        assert x[k] > 0.                                       # a variance

def accurate_sparse_solve(H, b):
    """The arbiter at sizes where a dense solve is out of reach: Jacobi-scaled sparse LU + iterative refinement with
    long-double residuals (cond(H) ~ 1e12 with the 1e-12 prior next to unit loop closures: SuperLU on the unscaled
    matrix is itself the less accurate side)."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    d = 1. / np.sqrt(H.diagonal())
    Dm = sp.diags(d)
    Hs = (Dm @ H @ Dm).tocsc()
    bs = b * d
    lu = spl.splu(Hs)
    x = lu.solve(bs)
    Hl = Hs.tocsr().astype(np.longdouble)
    for _ in range(3):
        res = bs.astype(np.longdouble) - Hl.dot(x.astype(np.longdouble))
        x = x + lu.solve(np.asarray(res, dtype=float))
    return x * d


def test_c2_full_size_first_steps_match_the_reference_algebra():
    """BASELINE config C2 at full size: 10 000 SE(3) poses, 50 001 edges, Huber, prior on pose 0.  Two Gauss-Newton
    steps on the device (two-level PCG, rigid-motion-aware coarse basis) in the form of the C3 / C4 checks, SURVEY 8d's
    tolerances: every step satisfies the oracle's normal equations at the device's own linearisation point,
    |H dx - b| <= 1e-9 |b| (Jacobi-scaled norms: block scales differ by 1e12), equals the accurate solve to 1e-8, and the
    post-step cost equals the oracle's to 1e-10."""
    import copy
    lp, _ = synthetic.pose_graph(num_poses=10000, num_loops=40001, dof=6, seed=2)
    dev = device(lp)
    cur = lp
    assert abs(dev.eval_cost(True) - orc.eval_cost(cur)) <= 1e-10 * orc.eval_cost(cur)
    for _ in range(2):
        H, b, _ = orc.normal_equations(cur, points_first=False)       # the oracle at the DEVICE's current parameters
        dx_ref = accurate_sparse_solve(H.tocsr(), b)
        cost, nrm, its, rel = dev.gn_iteration(0., 0., 4000, True)     # (tolerance 0 = Options().pcg_tol = None: the core's default, 1e-14 on pose graphs)
        assert rel <= 1e-14 and its < 4000                       # the CG converges (it did not before the Ad-aware basis)
        xp, _ = dev.get_dx()
        dx = xp.ravel()                                          # pose-first order, no landmarks: the oracle's order
        dj = 1. / np.sqrt(H.diagonal())
        assert np.linalg.norm((H.dot(dx) - b) * dj) <= 1e-9 * np.linalg.norm(b * dj)
        assert np.linalg.norm(dx - dx_ref) <= 1e-8 * np.linalg.norm(dx_ref), np.linalg.norm(dx - dx_ref) / np.linalg.norm(dx_ref)
        assert abs(nrm - np.linalg.norm(dx_ref)) <= 1e-8 * np.linalg.norm(dx_ref)
        nxt = orc.apply_update(cur, dx_ref, False)
        assert abs(cost - orc.eval_cost(nxt)) <= 1e-10 * cost
        poses, _ = dev.get_params()
        assert np.abs(poses - nxt.poses).max() < 1e-9
        cur = copy.copy(nxt)
        cur.poses = poses                                        # the next step is compared at the device's own point


def test_explicit_and_folded_two_level_cg_agree_on_a_long_chain():
    """Long sparse chains run the two-level PCG with the preconditioner APPLIED (k_xcg_*: restrict, dense coarse
    solve, prolong) instead of folded into the matrix.  Same operator: same step, similar iteration counts at the
    same number of coarse intervals; and the much finer coarse level it makes affordable needs far fewer."""
    lp, _ = synthetic.pose_graph(num_poses=3000, num_loops=12001, dof=6, seed=5)
    out = {}
    for name, explicit, groups in (('folded', 0, 48), ('explicit', 1, 48), ('explicit_fine', 1, 150)):
        dev = device(lp)
        dev.set_option('cg_explicit', explicit)
        dev.set_option('coarse_groups', groups)
        dev.linearize(0.)
        its, rel = dev.solve_reduced(1e-13, 4000)
        assert rel <= 1e-13
        out[name] = (its, device_dx_posefirst(dev))
    ref = out['folded'][1]
    for name in ('explicit', 'explicit_fine'):
        assert np.linalg.norm(out[name][1] - ref) <= 1e-8 * np.linalg.norm(ref), name
    assert abs(out['explicit'][0] - out['folded'][0]) <= 0.1 * out['folded'][0] + 5
    assert out['explicit_fine'][0] < 0.6 * out['explicit'][0]


@pytest.mark.parametrize('kf,lm,with_edges', [(560, 14000, False), (640, 12000, True)])
def test_mid_size_ba_uses_the_explicit_pcg_and_matches_the_oracle(kf, lm, with_edges):
    """Beyond 540 reduced poses bundle adjustment runs the explicitly applied two-level preconditioner by default
    (second iteration: with the lagged side-stream coarse inverse); two Gauss-Newton iterations against the oracle's
    CPU Schur solve, also with pose-pose edges and a prior over the same keyframes."""
    from pyslam_amd import losses
    loss = losses.HuberLoss(1.5) if with_edges else losses.L2Loss()
    lp, truth = synthetic.stereo_ba(num_kf=kf, num_lm=lm, obs_per_lm=6, half_window=12, seed=kf, loss=loss)
    if with_edges:
        lp = synthetic.with_pose_edges(lp, kf, kf + 1, loss=loss, truth_poses=truth['poses'])
    dev = device(lp)
    cur = lp
    for it in range(2):
        cost, nrm, its, rel = dev.gn_iteration(0., 1e-13, 4000, True)
        P, b, _ = orc.normal_equations(cur, points_first=False)
        dx = orc.schur_solve(cur, P, b, points_first=False)
        cur = orc.apply_update(cur, dx, points_first=False)
        want = orc.eval_cost(cur, True)
        poses, points = dev.get_params()
        assert 0 < its < 200 and rel <= 1e-12
        assert abs(cost - want) <= 1e-9 * want and abs(nrm - np.linalg.norm(dx)) <= 1e-8 * np.linalg.norm(dx)
        assert np.abs(poses - cur.poses).max() < 1e-9 and np.abs(points - cur.points).max() < 1e-8


# ---------------------------------------------------------------------------
# C4: the north star's target size (BASELINE.json configs[3]): 2 000 keyframes, 500 000 landmarks,
# 5 M reprojection blocks = 15 M residual rows, 1 511 994 unknowns.  The reference cannot run it;
# parity is through the size-independent properties, against the oracle's (reference algebra) matrix.
# ---------------------------------------------------------------------------
@pytest.fixture(scope='module')
def c4():
    lp, _ = synthetic.stereo_ba(2000, 500000, 10, 20, seed=1)
    return lp


def test_c4_step_solves_the_reference_normal_equations(c4):
    dev = device(c4)
    assert dev.info['num_obs'] == 5000000 and dev.info['num_reduced'] == 1999
    c0 = dev.eval_cost(True)
    assert abs(c0 - orc.eval_cost(c4)) <= 1e-11 * c0
    cost, nrm, its, rel = dev.gn_iteration(0., 1e-12, 2000, True)       # the whole-iteration call the bench times
    dx = device_dx_posefirst(dev)
    P, b, lin_cost = orc.normal_equations(c4, points_first=False)        # reference algebra (oracle), 185 M non-zeros
    res = P.dot(dx) - b
    assert np.linalg.norm(res) <= 1e-9 * np.linalg.norm(b), (its, rel)
    n = 6 * c4.num_reduced
    assert np.linalg.norm(res[n:]) <= 1e-11 * np.linalg.norm(b[n:])       # landmark part: exact (direct) solve
    assert abs(nrm - np.linalg.norm(dx)) <= 1e-10 * nrm
    assert abs(cost - orc.eval_cost(orc.apply_update(c4, dx, points_first=False))) <= 1e-9 * cost
    assert cost < 0.05 * c0
    # a second iteration (lagged side-stream coarse inverse, predicted launch count) still converges to tolerance
    cost2, nrm2, its2, rel2 = dev.gn_iteration(0., 1e-12, 2000, True)
    assert rel2 <= 1e-12 and cost2 < cost


def test_c4_four_landmark_shards_sum_to_the_unsharded_reduced_system(c4):
    """The 4-GPU split of the FIXED C4 problem (bench.py --gpus 4), checked on ONE GPU: every shard
    built with the union block pattern; S, g and the cost summed on the host equal the unsharded ones."""
    full = device(c4)
    full.linearize(0.)
    rp, ci, vals, g = full.reduced_system()
    cost_full = full.eval_cost(True)
    full.close()
    union = pose_pair_keys(c4)
    acc_v, acc_g, cost, nobs = np.zeros_like(vals), np.zeros_like(g), 0., 0
    world = 4
    for r in range(world):
        sh = shard_landmarks(c4, r, world)
        nobs += sh.num_obs
        assert abs(sh.num_obs - c4.num_obs / world) <= 0.01 * c4.num_obs   # balanced by observation count
        extra = np.setdiff1d(union, pose_pair_keys(sh))
        dev = device(sh, extra_pairs=((extra >> 32).astype(np.int32), (extra & 0xFFFFFFFF).astype(np.int32)))
        dev.linearize(0.)
        rp2, ci2, v2, g2 = dev.reduced_system()
        assert np.array_equal(rp, rp2) and np.array_equal(ci, ci2)       # identical pattern on every rank
        acc_v += v2; acc_g += g2
        cost += dev.eval_cost(True)
        dev.close()
    assert nobs == c4.num_obs
    assert np.abs(acc_v - vals).max() <= 1e-11 * np.abs(vals).max()
    assert np.abs(acc_g - g).max() <= 1e-11 * np.abs(g).max()
    assert abs(cost - cost_full) <= 1e-12 * cost


def test_coarse_inverse_refreshed_every_second_iteration_gives_the_same_trajectory():
    """"coarse_refresh_every" = 2 (for coarse levels whose factorisation outlasts an iteration): the explicit two-level PCG holds its
    lagged coarse inverse for two iterations and assembles / factors A_c only every second one.  The preconditioner is
    staler, the solves are the same: four Gauss-Newton iterations agree with the default schedule to 1e-9."""
    lp, _ = synthetic.stereo_ba(num_kf=640, num_lm=12000, obs_per_lm=6, half_window=12, seed=31)
    out = {}
    for every in (1, 2):
        dev = device(lp)
        dev.set_option('coarse_refresh_every', every)
        out[every] = ([dev.gn_iteration(0., 1e-12, 2000, True) for _ in range(4)], dev.get_params())
        dev.close()
    for (c1, n1, i1, r1), (c2, n2, i2, r2) in zip(out[1][0], out[2][0]):
        assert abs(c1 - c2) <= 1e-9 * abs(c1) and abs(n1 - n2) <= 1e-7 * n1 + 1e-12
        assert r2 <= 1e-12 and i2 <= i1 + 25
    assert np.abs(out[1][1][0] - out[2][1][0]).max() <= 1e-8


@pytest.mark.parametrize('shape', ['ba', 'pg'])
def test_coarse_inverse_held_while_the_solve_has_settled_gives_the_same_trajectory(shape):
    """"coarse_auto_hold" (default): once an iteration changes the cost by less than 1e-4 relative, the explicit two-level
    PCG keeps its lagged coarse inverse for up to three set-ups instead of assembling and factoring A_c again.  It only
    preconditions: eight Gauss-Newton iterations agree with the always-refresh schedule, CG counts within a few."""
    if shape == 'ba':
        lp, _ = synthetic.stereo_ba(num_kf=640, num_lm=12000, obs_per_lm=6, half_window=12, seed=32)
    else:
        lp, _ = synthetic.pose_graph(num_poses=1200, num_loops=4801, dof=6, seed=33)
    out = {}
    for hold in (0, 1):
        dev = device(lp)
        dev.set_option('coarse_auto_hold', hold)
        out[hold] = ([dev.gn_iteration(0., 1e-12, 3000, True) for _ in range(8)], dev.get_params())
        dev.close()
    for (c1, n1, i1, r1), (c2, n2, i2, r2) in zip(out[0][0], out[1][0]):
        assert abs(c1 - c2) <= 1e-9 * abs(c1) and abs(n1 - n2) <= 1e-6 * n1 + 1e-10
        assert r2 <= 1e-12 and i2 <= i1 + 10
    for a, b in zip(out[0][1], out[1][1]):
        if a.size:
            assert np.abs(a - b).max() <= 1e-8
