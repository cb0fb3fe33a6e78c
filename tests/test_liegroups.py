"""Independent pins for the liegroups restatement (the reference's third-party
dependency is absent and unpinned): scipy expm/logm and Rotation known answers,
plus the conventions the hot path relies on."""
import copy

import numpy as np
import scipy.linalg as sl
from scipy.spatial.transform import Rotation

from liegroups import SE2, SE3, SO2, SO3


def test_se3_exp_log_against_expm_logm():
    rng = np.random.default_rng(0)
    for scale in (1e-10, 1e-4, 0.3, 1.0, 2.5):
        for _ in range(5):
            xi = scale * rng.standard_normal(6)
            T = SE3.exp(xi)
            assert np.abs(T.as_matrix() - sl.expm(SE3.wedge(xi))).max() < 1e-12
            if np.linalg.norm(xi[3:]) < 3.0:              # log is the principal branch (|phi| < pi)
                assert np.abs(T.log() - xi).max() < 1e-9 * max(1., scale)
                assert np.abs(SE3.vee(np.real(sl.logm(T.as_matrix()))) - xi).max() < 1e-8


def test_se2_exp_log_against_expm():
    rng = np.random.default_rng(1)
    for scale in (1e-10, 1e-3, 1.0, 3.0):
        xi = scale * rng.standard_normal(3)
        T = SE2.exp(xi)
        assert np.abs(T.as_matrix() - sl.expm(SE2.wedge(xi))).max() < 1e-13
        assert np.abs(T.log() - xi).max() < 1e-10


def test_so3_against_scipy_rotation():
    rng = np.random.default_rng(2)
    for _ in range(10):
        phi = rng.standard_normal(3)
        assert np.abs(SO3.exp(phi).as_matrix() - Rotation.from_rotvec(phi).as_matrix()).max() < 1e-14
        assert np.abs(SO3.exp(phi).log() - Rotation.from_rotvec(phi).as_rotvec()).max() < 1e-12
    assert np.allclose(SO3.rotz(0.3).as_matrix(), Rotation.from_euler('z', 0.3).as_matrix())
    assert np.allclose(SO2.from_angle(0.7).to_angle(), 0.7)


def test_left_jacobians_are_inverse_pairs_and_match_series():
    rng = np.random.default_rng(3)
    for phi in (rng.standard_normal(3), 1e-9 * np.ones(3)):
        J, Ji = SO3.left_jacobian(phi), SO3.inv_left_jacobian(phi)
        assert np.abs(J @ Ji - np.eye(3)).max() < 1e-9
        W = SO3.wedge(phi)
        series = sum(np.linalg.matrix_power(W, n) / np.math.factorial(n + 1) for n in range(25)) \
            if hasattr(np, 'math') else None
        if series is not None:
            assert np.abs(J - series).max() < 1e-12


def test_left_perturbation_adjoint_and_odot_conventions():
    rng = np.random.default_rng(4)
    T = SE3.exp(rng.standard_normal(6))
    xi = 1e-6 * rng.standard_normal(6)
    p = rng.standard_normal(3)
    T2 = copy.deepcopy(T)
    T2.perturb(xi)                                      # T <- exp(xi) T
    assert np.abs(T2.as_matrix() - SE3.exp(xi).dot(T).as_matrix()).max() < 1e-15
    # d(T p)/dxi = odot(T p) under left perturbation
    assert np.abs((T2.dot(p) - T.dot(p)) - SE3.odot(T.dot(p)) @ xi).max() < 1e-10
    # T exp(xi) T^-1 = exp(Ad(T) xi)
    lhs = T.dot(SE3.exp(xi)).dot(T.inv()).as_matrix()
    assert np.abs(lhs - SE3.exp(T.adjoint() @ xi).as_matrix()).max() < 1e-12
    U = SE2.exp([0.3, -0.2, 0.9])
    x2 = 1e-6 * rng.standard_normal(3)
    lhs = U.dot(SE2.exp(x2)).dot(U.inv()).as_matrix()
    assert np.abs(lhs - SE2.exp(U.adjoint() @ x2).as_matrix()).max() < 1e-12
    assert np.allclose(SE3.odot(p), np.hstack([np.eye(3), -SO3.wedge(p)]))
    assert SE3.odot(rng.standard_normal((5, 3))).shape == (5, 3, 6)


def test_api_surface():
    T = SE3.identity()
    assert SE3.dof == 6 and SE3.dim == 4 and SE2.dof == 3 and SE2.dim == 3 and SO3.dof == 3 and SO2.dof == 1
    assert np.allclose(SE3.log(T), 0.) and np.allclose(SE2.log(SE2.identity()), 0.)
    assert np.allclose((T * np.array([1., 2., 3.])), [1., 2., 3.])
    assert T.dot(np.ones((4, 3))).shape == (4, 3)
    M = SE3.exp(np.arange(6) / 10.).as_matrix()
    assert np.allclose(SE3.from_matrix(M).as_matrix(), M)
    R = SO3.exp([0.1, 0.2, 0.3])
    R.mat = R.mat + 1e-9
    R.normalize()
    assert np.allclose(R.mat.T @ R.mat, np.eye(3), atol=1e-14)
