"""CPU tests of the host layer: API surface, lowering, C-ABI symbols, fail-loud."""
import ctypes
import os
import re
import types

import numpy as np
import pytest

from conftest import load_golden, golden_lp, REPO


def build_namespace():
    import pyslam.problem as P
    import pyslam.residuals as R
    import pyslam.losses as Ls
    import pyslam.sensors as S
    import liegroups as G
    return types.SimpleNamespace(
        Problem=P.Problem, Options=P.Options, StereoCamera=S.StereoCamera,
        PoseResidual=R.PoseResidual, PoseToPoseResidual=R.PoseToPoseResidual,
        PoseToPoseOrientationResidual=R.PoseToPoseOrientationResidual,
        ReprojectionResidual=R.ReprojectionResidual,
        L2Loss=Ls.L2Loss, L1Loss=Ls.L1Loss, CauchyLoss=Ls.CauchyLoss, HuberLoss=Ls.HuberLoss,
        TukeyLoss=Ls.TukeyLoss, TDistributionLoss=Ls.TDistributionLoss,
        SE3=G.SE3, SO3=G.SO3, SE2=G.SE2, SO2=G.SO2)


def test_reference_module_names_import():
    from pyslam.problem import Options, Problem  # noqa: F401
    from pyslam.residuals import (PoseResidual, PoseToPoseResidual, ReprojectionResidual,  # noqa: F401
                                  ReprojectionMotionOnlyResidual, ReprojectionMotionOnlyBatchResidual,
                                  ReprojectionResidualFrameToFrame, QuadraticResidual,
                                  PoseToPoseOrientationResidual)
    from pyslam.losses import L2Loss, L1Loss, CauchyLoss, HuberLoss, TukeyLoss, TDistributionLoss  # noqa: F401
    from pyslam.sensors import StereoCamera, RGBDCamera  # noqa: F401
    from pyslam.utils import invsqrt  # noqa: F401
    from liegroups import SE2, SE3, SO2, SO3  # noqa: F401


class TestBookkeeping:
    """Reference tests/test_problem.py:10-43 (registry semantics)."""

    def test_residual_blocks(self):
        from pyslam.problem import Problem
        from pyslam.residuals import QuadraticResidual
        problem = Problem()
        keys = ['a', 'b', 'c']
        problem.add_residual_block(QuadraticResidual(2., 4., 1.), keys)
        assert keys == problem.block_param_keys[0]
        problem.add_residual_block(QuadraticResidual(2., 4., 1.), 'a')
        assert problem.block_param_keys[1] == ['a']

    def test_param_dict(self):
        from pyslam.problem import Problem
        problem = Problem()
        params = {'a': 1, 'b': 2, 'c': 3}
        problem.initialize_params(params)
        assert problem.param_dict == params
        params.update({'d': 4})
        problem.initialize_params({'d': 4})
        assert problem.param_dict == params

    def test_constant_params(self):
        from pyslam.problem import Problem
        problem = Problem()
        problem.set_parameters_constant('a')
        assert problem.constant_param_keys == ['a']
        problem.set_parameters_constant(['a', 'b_param'])
        assert problem.constant_param_keys == ['a', 'b_param']
        problem.set_parameters_variable('a')
        assert problem.constant_param_keys == ['b_param']
        problem.set_parameters_variable('c')
        assert problem.constant_param_keys == ['b_param']
        problem.set_parameters_variable(['a', 'b_param', 'c'])
        assert problem.constant_param_keys == []

    def test_summary_errors_and_format(self):
        from pyslam.problem import Problem
        problem = Problem()
        with pytest.raises(ValueError):
            problem.summary()
        problem._cost_history = [3.735817e+05, 5.045051e-26]
        assert problem.summary() == 'Iterations:   2 | Cost: 3.735817e+05 --> 5.045051e-26'
        assert problem.summary(format='full').startswith(' Iter | Initial cost -->   Final cost | Rel change\n')
        with pytest.raises(ValueError):
            problem.summary(format='nope')

    def test_partition_dict_order(self):
        from pyslam.problem import Problem
        from liegroups import SE3, SE2
        problem = Problem()
        problem.initialize_params({'p': np.zeros(3), 'T': SE3.identity(), 's': 1.0, 'U': SE2.identity()})
        problem.set_parameters_constant('s')
        part = problem._get_update_partition_dict()
        assert part == {'p': range(0, 3), 'T': range(3, 9), 'U': range(9, 12)}


class TestResidualContracts:
    """Reference tests/test_costs.py (shapes, None for constant params, zero at truth)."""

    def test_quadratic(self):
        from pyslam.residuals import QuadraticResidual
        res = QuadraticResidual(2., 3., 1.)
        assert res.evaluate([1., -2., 3.]) == 0.
        assert res.evaluate([0., 3., 1.]) != 0.
        _, j1 = res.evaluate([1., -2., 3.], [True, True, True])
        _, j2 = res.evaluate([1., -2., 3.], [False, False, False])
        assert len(j1) == len(j2) == 3
        assert np.allclose(j1, [4., 2., 1.])
        assert not any(j2)

    @pytest.mark.parametrize('group,dof', [('SE2', 3), ('SE3', 6)])
    def test_pose_blocks(self, group, dof):
        import liegroups
        from pyslam.residuals import PoseResidual, PoseToPoseResidual
        G = getattr(liegroups, group)
        T1 = G.exp(np.arange(1, dof + 1, dtype=float))
        T2 = G.exp(np.arange(dof + 1, 2 * dof + 1, dtype=float))
        pr = PoseResidual(T1, np.eye(dof))
        assert np.allclose(pr.evaluate([T1]), 0.)
        assert not np.allclose(pr.evaluate([T2]), 0.)
        _, J = pr.evaluate([T2], [True])
        assert len(J) == 1 and J[0].shape == (dof, dof)
        pp = PoseToPoseResidual(G.identity(), np.eye(dof))
        assert np.allclose(pp.evaluate([T1, T1]), 0.)
        _, Ja = pp.evaluate([T1, T2], [True, True])
        _, Jb = pp.evaluate([T1, T2], [True, False])
        _, Jc = pp.evaluate([T1, T2], [False, True])
        assert Ja[0].shape == Ja[1].shape == (dof, dof)
        assert Jb[1] is None and Jc[0] is None and Jb[0].shape == Jc[1].shape == (dof, dof)

    def test_reprojection(self):
        from liegroups import SE3
        from pyslam.sensors import StereoCamera
        from pyslam.residuals import ReprojectionResidual
        res = ReprojectionResidual(StereoCamera(100., 100., 200., 200., 1., 200, 200), [40, 60, 10], np.eye(3))
        T = SE3.exp([1, 2, 3, 4, 5, 6])
        good = T.inv().dot(res.camera.triangulate(res.obs))
        assert np.allclose(res.evaluate([T, good]), 0.)
        assert not np.allclose(res.evaluate([T, good + [1, 1, 1]]), 0.)
        _, J = res.evaluate([T, good], [True, True])
        assert J[0].shape == (3, 6) and J[1].shape == (3, 3)
        _, J = res.evaluate([T, good], [False, True])
        assert J[0] is None and J[1].shape == (3, 3)

    def test_blocks_match_reference_golden(self):
        from liegroups import SE3, SO3, SE2, SO2
        from pyslam.sensors import StereoCamera
        from pyslam.residuals import ReprojectionResidual, PoseToPoseResidual, PoseResidual
        g = load_golden('blocks')
        cam = StereoCamera(640., 480., 1000., 1000., 0.25, 1280, 960)
        for M, p, o, r, Jp, Jl in zip(g['rpn_T'], g['rpn_pt'], g['rpn_obs'], g['rpn_r'], g['rpn_Jpose'], g['rpn_Jpt']):
            T = SE3(SO3(M[:3, :3]), M[:3, 3])
            rr, J = ReprojectionResidual(cam, o, g['rpn_S']).evaluate([T, p], [True, True])
            assert np.allclose(rr, r, rtol=1e-13, atol=1e-12)
            assert np.allclose(J[0], Jp, rtol=1e-13) and np.allclose(J[1], Jl, rtol=1e-13)
        for dof, SE, SO in ((6, SE3, SO3), (3, SE2, SO2)):
            t, n = 'pp{}_'.format(dof), dof // 3 + 1
            mk = lambda M: SE(SO(M[:n, :n]), M[:n, n])
            for A, B, O, r, J1, rp in zip(g[t + 'T1'], g[t + 'T2'], g[t + 'Tobs'], g[t + 'r'], g[t + 'J1'], g[t + 'r_prior']):
                rr, J = PoseToPoseResidual(mk(O), g[t + 'S']).evaluate([mk(A), mk(B)], [True, True])
                assert np.allclose(rr, r, rtol=1e-12, atol=1e-13)
                assert np.allclose(J[0], J1, rtol=1e-13) and np.allclose(J[1], g[t + 'S'])
                assert np.allclose(PoseResidual(mk(O), g[t + 'S']).evaluate([mk(B)]), rp, rtol=1e-12, atol=1e-13)


def test_losses_and_sensors_match_reference_golden():
    import pyslam.losses as L
    from pyslam.sensors import StereoCamera
    from pyslam.utils import invsqrt
    g = load_golden('losses_sensors')
    x = g['x']
    for name, loss in (('l2', L.L2Loss()), ('l1', L.L1Loss()), ('cauchy', L.CauchyLoss(3.0)),
                       ('huber', L.HuberLoss(1.5)), ('tukey', L.TukeyLoss(3.0)),
                       ('tdist', L.TDistributionLoss(5.0))):
        assert np.allclose(loss.loss(x), g[name + '_loss'], rtol=1e-14, atol=0, equal_nan=True)
        assert np.allclose(loss.weight(x), g[name + '_weight'], rtol=1e-14, atol=0, equal_nan=True)
        if name != 'huber':
            assert np.allclose(loss.influence(x), g[name + '_influence'], rtol=1e-14, atol=0, equal_nan=True)
    assert callable(L.HuberLoss(1.).influence(x))        # reference quirk, losses.py:83-84
    cam = StereoCamera(640., 480., 1000., 1000., 0.25, 1280, 960)
    uvd, J = cam.project(g['cam_pts'], True)
    xyz, Jt = cam.triangulate(g['cam_uvd'], True)
    assert np.allclose(uvd, g['cam_uvd'], rtol=1e-15) and np.allclose(J, g['cam_J'], rtol=1e-15)
    assert np.allclose(xyz, g['cam_xyz'], rtol=1e-14) and np.allclose(Jt, g['cam_Jt'], rtol=1e-14)
    assert np.array_equal(cam.is_valid_measurement(g['cam_uvd']), g['cam_valid'])
    assert np.allclose(invsqrt(np.array([[4., 1., 0.], [1., 3., .5], [0., .5, 2.]])), g['invsqrt_3x3'])
    assert invsqrt(4.) == 0.5


def test_sensor_known_answers():
    """Reference tests/test_sensors.py:15-136 exact Jacobians."""
    from pyslam.sensors import StereoCamera, RGBDCamera
    s = StereoCamera(150., 100., 250., 200., 1., 300, 200)
    assert [bool(s.is_valid_measurement(u)) for u in
            [[110., 120., 10.], [-10., 100., 10.], [0., -10., 10.], [0., 0., -5.]]] == [True, False, False, False]
    _, J = s.project([1., 2., 10.], True)
    assert np.allclose(J, [[25., 0., -2.5], [0., 20., -4.], [0., 0., -2.5]])
    _, J = s.triangulate([110., 120., 10.], True)
    assert np.allclose(J, [[0.1, 0., 0.4], [0., 0.125, -0.25], [0., 0., -2.5]])
    assert np.allclose(s.triangulate(s.project([1., 2., 10.])), [1., 2., 10.])
    uvd, J = s.project(np.array([[1., 2., 10.], [2., 1., -20.]]), True)
    assert uvd.shape == (2, 3) and J.shape == (2, 3, 3)
    r = RGBDCamera(150., 100., 250., 200., 300, 200)
    _, J = r.project([1., 2., 10.], True)
    assert np.allclose(J, [[25., 0., -2.5], [0., 20., -4.], [0., 0., 1.]])
    _, J = r.triangulate([110., 120., 10.], True)
    assert np.allclose(J, [[0.04, 0., -0.16], [0., 0.05, 0.1], [0., 0., 1.]])


def test_lowering_roundtrip():
    """tables -> objects -> lowering reproduces the tables (up to inv(inv(T)))."""
    import pyslam_amd.synthetic as synthetic
    from pyslam_amd import lowering
    for name in ('ba_tiny_huber', 'pg2d_small_huber', 'posegraph_3d_example'):
        lp = golden_lp(load_golden(name))
        problem = synthetic.to_objects(lp, build_namespace())
        lp2 = problem._lower()
        assert lp2.dof == lp.dof
        # parameters were inserted points-first: same tables, same order
        assert np.array_equal(lp2.poses, lp.poses) and np.array_equal(lp2.points, lp.points)
        assert np.array_equal(lp2.pose_rid, lp.pose_rid) and np.array_equal(lp2.point_vid, lp.point_vid)
        assert np.array_equal(lp2.obs_pose, lp.obs_pose) and np.array_equal(lp2.obs_point, lp.obs_point)
        assert np.array_equal(lp2.obs_uvd, lp.obs_uvd)
        assert np.array_equal(lp2.e_i, lp.e_i) and np.array_equal(lp2.e_j, lp.e_j)
        assert np.allclose(lp2.e_Tobs_inv, lp.e_Tobs_inv, atol=1e-14)
        assert np.array_equal(lp2.u_i, lp.u_i)


def test_lowering_motion_only_blocks():
    from liegroups import SE3
    from pyslam.problem import Problem
    from pyslam.sensors import StereoCamera
    from pyslam.residuals import ReprojectionMotionOnlyBatchResidual, ReprojectionMotionOnlyResidual
    from pyslam.losses import CauchyLoss
    g = load_golden('motion_only_cauchy')
    cam = StereoCamera(640., 480., 1000., 1000., 0.25, 1280, 960)
    S = g['lp_stiff3'].reshape(3, 3)
    problem = Problem()
    problem.add_residual_block(ReprojectionMotionOnlyBatchResidual(cam, g['obs_1'], g['obs_2'], S),
                               ['T_2_1'], CauchyLoss(3.0))
    problem.add_residual_block(ReprojectionMotionOnlyResidual(cam, g['obs_1'][0], g['obs_2'][0], S), 'T_2_1')
    problem.initialize_params({'T_2_1': SE3.identity()})
    lp = problem._lower()
    assert lp.num_obs == 257 and lp.num_points == 257 and (lp.point_vid < 0).all()
    assert np.allclose(lp.points[:256], g['pts_1']) and np.array_equal(lp.obs_uvd[:256], g['obs_2'])
    assert lp.obs_groups.shape[0] == 2 and lp.obs_groups[0, 2] == 2 and lp.obs_groups[1, 2] == 0


def test_unlowerable_falls_to_generic_marker():
    from pyslam.problem import Problem
    from pyslam.residuals import QuadraticResidual
    from pyslam_amd.lowering import NotLowerable
    problem = Problem()
    problem.add_residual_block(QuadraticResidual(1., 2., 1.), ['a', 'b', 'c'])
    problem.initialize_params({'a': 0., 'b': 0., 'c': 0.})
    with pytest.raises(NotLowerable):
        problem._lower()


def test_c_abi_exports_every_declared_symbol():
    from pyslam_amd import _native
    header = open(os.path.join(REPO, 'include', 'pyslam_hip.h')).read()
    declared = set(re.findall(r'\b(ps_[a-z0-9_]+)\s*\(', header))
    assert declared == set(_native.SIGNATURES), declared ^ set(_native.SIGNATURES)
    lib = ctypes.CDLL(_native.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert ctypes.sizeof(_native.ProblemDesc) == 272 and ctypes.sizeof(_native.ProblemInfo) == 184  # gcc sizeof


def test_the_list_of_measurement_switches_is_the_sources_list():
    """_native.load() warns about measurement switches the product library ignores -- about those the sources read through
    ps_env(), not about every PS_* variable of the environment (procps' PS_FORMAT ...: round-5 ADVICE)."""
    from pyslam_amd import _native
    src = os.path.join(REPO, 'pyslam_amd', 'csrc')
    found = set()
    for f in os.listdir(src):
        found |= set(re.findall(r'ps_env\("(PS_[A-Z0-9_]+)"\)', open(os.path.join(src, f)).read()))
    assert found == _native._MEASURE_ENV, found ^ _native._MEASURE_ENV
    assert not (_native._MEASURE_ENV & _native._CREATE_ENV) and 'PS_FORMAT' not in _native._MEASURE_ENV


def test_unknown_descriptor_flags_are_rejected_before_anything_is_touched():
    """ps_problem_desc.flags: bits other than PS_DESC_DEVICE_PARAMS | PS_DESC_DEVICE_TABLES are an error (checked first, so
    it can be exercised without a GPU and without valid tables)."""
    from pyslam_amd import _native
    lib = _native.load()
    d = _native.ProblemDesc()
    d.dof, d.flags = 6, 4
    h = _native.H()
    assert lib.ps_problem_create(ctypes.byref(d), None, ctypes.byref(h)) < 0
    assert b'flags' in lib.ps_last_error()
    header = open(os.path.join(REPO, 'include', 'pyslam_hip.h')).read()
    assert re.search(r'PS_DESC_DEVICE_PARAMS = 1u, PS_DESC_DEVICE_TABLES = 2u', header)
    assert (_native.PS_DESC_DEVICE_PARAMS, _native.PS_DESC_DEVICE_TABLES) == (1, 2)


def test_solver_fails_loudly_without_gpu():
    """No CPU fallback: on a machine without an MI355X the product raises."""
    from pyslam_amd import _native
    if _native.load().ps_device_count() > 0:
        pytest.skip('GPU present')
    from pyslam.problem import Problem
    from pyslam.residuals import QuadraticResidual
    problem = Problem()
    problem.add_residual_block(QuadraticResidual(1., 2., 1.), ['a', 'b', 'c'])
    problem.initialize_params({'a': 0., 'b': 0., 'c': 0.})
    with pytest.raises(_native.NativeError):
        problem.solve()
    lp = golden_lp(load_golden('stereo_ba_example'))
    from pyslam_amd.device import DeviceProblem
    with pytest.raises(_native.NativeError):
        DeviceProblem(lp)


def test_orientation_residual_lowers_to_a_rotation_only_edge():
    """PoseToPoseOrientationResidual (reference pose_to_pose_orientation_residual.py:4-38) becomes a
    pose-pose edge with T_obs = (C_obs, 0) and the 3x3 stiffness in the rotational corner of a 6x6 one;
    the lowered tables reproduce the class's own residual and Jacobians through the numpy oracle."""
    from liegroups import SE3, SO3
    from pyslam.residuals import PoseToPoseOrientationResidual
    from pyslam.losses import L2Loss
    from pyslam_amd import lowering
    from oracle import gn_oracle as orc
    rng = np.random.default_rng(3)
    T1, T2 = SE3.exp(rng.standard_normal(6) * 0.3), SE3.exp(rng.standard_normal(6) * 0.3)
    C_obs = SO3.exp(rng.standard_normal(3) * 0.3)
    S3 = np.diag([2., 3., 5.]) + 0.1 * rng.standard_normal((3, 3))
    block = PoseToPoseOrientationResidual(C_obs, S3)
    lp = lowering.lower({'a': T1, 'b': T2}, [block], [['a', 'b']], [L2Loss()], [])
    assert lp.num_edges == 1 and lp.dof == 6
    S6 = lp.stiffd[int(lp.edge_groups[lp.e_grp[0], 0])].reshape(6, 6)
    assert not S6[:3].any() and not S6[:, :3].any() and np.array_equal(S6[3:, 3:], S3)
    r, J1, J2 = orc.eval_edges(lp)
    r_ref, (j1_ref, j2_ref) = block.evaluate([T1, T2], [True, True])
    assert np.allclose(r[0, :3], 0.) and np.allclose(r[0, 3:], r_ref, atol=1e-14)
    assert np.allclose(J1[0, 3:], j1_ref, atol=1e-14) and np.allclose(J2[0, 3:], j2_ref, atol=1e-14)
    assert not J1[0, :3].any() and not J2[0, :3].any()


def test_loss_subclass_and_group_overflow_are_not_lowered():
    """A user subclass of a built-in loss may override loss() / weight(): it must not be lowered to the parent's
    device formula.  More (camera, stiffness, loss) groups than the 8-bit group field holds: NotLowerable
    (the host-evaluated path then runs), not an error out of ps_problem_create."""
    from liegroups import SE3
    from pyslam.losses import HuberLoss, L2Loss
    from pyslam.sensors import StereoCamera
    from pyslam.residuals import ReprojectionResidual
    from pyslam_amd import lowering
    from pyslam_amd.lowering import NotLowerable

    class MyHuber(HuberLoss):
        def weight(self, x):
            return np.ones_like(np.asarray(x, dtype=float))
    cam = StereoCamera(640., 480., 1000., 1000., 0.25, 1280, 960)
    params = {'T': SE3.identity(), 'p': np.array([0., 0., 10.])}
    block = ReprojectionResidual(cam, np.array([640., 480., 25.]), np.eye(3))
    assert lowering.lower(params, [block], [['T', 'p']], [HuberLoss(1.)], []).obs_groups[0, 2] == 3
    with pytest.raises(NotLowerable):
        lowering.lower(params, [block], [['T', 'p']], [MyHuber(1.)], [])
    # one stiffness per observation: lowerable whatever the count (the device splits rows beyond 255 into (camera, loss)
    # classes + a per-observation stiffness index); more than 255 CLASSES are not
    blocks = [ReprojectionResidual(cam, np.array([640., 480., 25.]), (1. + 0.01 * i) * np.eye(3)) for i in range(300)]
    loss = L2Loss()
    wide = lowering.lower(params, blocks, [['T', 'p']] * 300, [loss] * 300, [])
    assert wide.obs_groups.shape[0] == 300 and wide.stiff3.shape[0] == 300
    ok = lowering.lower(params, blocks[:255], [['T', 'p']] * 255, [loss] * 255, [])
    assert ok.obs_groups.shape[0] == 255
    with pytest.raises(NotLowerable):
        lowering.lower(params, blocks, [['T', 'p']] * 300, [HuberLoss(1. + 0.01 * i) for i in range(300)], [])


def test_same_tables_sees_every_edit_but_parameter_values():
    """Problem keeps its HBM tables between calls only while nothing but the parameter VALUES changed."""
    import pyslam_amd.synthetic as synthetic
    lp = golden_lp(load_golden('ba_tiny_huber'))
    problem = synthetic.to_objects(lp, build_namespace())
    a = problem._lower()
    assert a.same_tables(problem._lower())
    problem.param_dict[a.point_keys[0]][0] += 0.5                      # a parameter value: tables still valid
    problem.param_dict[a.pose_keys[1]].perturb(0.01 * np.ones(6))
    b = problem._lower()
    assert a.same_tables(b) and not np.array_equal(a.points, b.points) and not np.array_equal(a.poses, b.poses)
    problem.block_loss_functions[0].k *= 2.                            # loss parameter
    assert not a.same_tables(problem._lower())
    problem.block_loss_functions[0].k /= 2.
    assert a.same_tables(problem._lower())
    problem.residual_blocks[3].obs[1] += 1.                            # a measurement, edited in place
    assert not a.same_tables(problem._lower())
    problem.residual_blocks[3].obs[1] -= 1.
    problem.residual_blocks[0], problem.residual_blocks[1] = problem.residual_blocks[1], problem.residual_blocks[0]
    assert not a.same_tables(problem._lower())                         # same counts, different connectivity
    problem.residual_blocks[0], problem.residual_blocks[1] = problem.residual_blocks[1], problem.residual_blocks[0]
    problem.set_parameters_constant(a.point_keys[2])
    assert not a.same_tables(problem._lower())


def test_fast_se3_odot_and_its_shape_constant_are_exported():
    """reference reprojection_motion_only_residual.py:9-32 (surfaced by the residuals package's pkgutil walk):
    fast_se3_odot(pts, SE3_ODOT_SHAPE) == SE3.odot(pts) for one point and for a stack."""
    import pyslam.residuals as R
    from liegroups import SE3
    assert R.SE3_ODOT_SHAPE.shape == (6,)
    rng = np.random.default_rng(0)
    p = rng.standard_normal(3)
    pts = rng.standard_normal((17, 3))
    assert np.array_equal(R.fast_se3_odot(p, R.SE3_ODOT_SHAPE), SE3.odot(p))
    assert np.array_equal(R.fast_se3_odot(pts, R.SE3_ODOT_SHAPE), SE3.odot(pts))
    assert R.fast_se3_odot(pts, R.SE3_ODOT_SHAPE).shape == (17, 3, 6)
    for name in ('stackmul', 'bilinear_interpolate', 'SE3'):
        assert name in R.__all__ and hasattr(R, name)


def test_update_partition_is_rebuilt_on_first_use_after_solve_invalidated_it():
    """reference problem.py:132 rebuilds `_update_partition_dict` at the top of solve(); here solve() only invalidates it (the
    device path never reads it: one Python statement per parameter saved) and the attribute rebuilds itself when read."""
    import pyslam.problem as P
    problem = P.Problem()
    problem.initialize_params({'a': np.zeros(3), 'b': np.zeros(3), 'c': np.zeros(2)})
    problem.set_parameters_constant('b')
    assert problem._update_partition_dict == {}                       # as constructed (reference problem.py:60)
    problem._partition = None                                         # what solve() does
    assert problem._update_partition_dict == {'a': range(0, 3), 'c': range(3, 5)}
    problem._update_partition_dict = {'x': range(0, 1)}               # assignable, as in the reference
    assert problem._update_partition_dict == {'x': range(0, 1)}
