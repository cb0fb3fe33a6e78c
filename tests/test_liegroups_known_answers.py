"""Known-answer pins of the liegroups arithmetic at the places where three restatements of one upstream formula
(pyslam_amd/liegroups, oracle/gn_oracle.py, csrc/ps_math.h -- all written here, the upstream source is not available
offline) could agree with each other and still be wrong: rotation angles at the ``np.isclose(angle, 0.)`` branch
boundary (|theta| = 1e-8) and next to pi.

The answers do not come from any of the three: exp is the matrix exponential series summed in long double (80-bit),
and log / J_l / J_l^-1 are checked through identities against that series (log(exp(xi)) = xi, t = J_l(phi) rho,
J_l J_l^-1 = I), with the conditioning of the upstream formulas (log: eps / (pi - theta)^2 near pi; J_l: eps / theta just
above the small-angle switch) written into the tolerances.  Reference call sites: pose_to_pose_residual.py:17-24,
pose_residual.py:16, problem.py:260, 406.
"""
import numpy as np
import pytest

from liegroups import SE2, SE3, SO3
from oracle import gn_oracle as orc

LD = np.longdouble
THETAS = [1e-9, 9.9e-9, 1e-8, 1.01e-8, 1e-7, 1e-3, 1.0, np.pi - 1e-3, np.pi - 1e-6]
AXES = [np.array([1., 0., 0.]), np.array([1., 2., 3.]) / np.sqrt(14.), np.array([-0.3, 0.5, -0.8]) / np.sqrt(0.98)]
RHO = np.array([0.3, -0.2, 0.5])


def expm_ld(A):
    """Matrix exponential by its Taylor series in long double (|A| <= ~4: 60 terms reach 1e-19)."""
    A = np.asarray(A, dtype=LD)
    out = np.identity(A.shape[0], dtype=LD)
    term = np.identity(A.shape[0], dtype=LD)
    for n in range(1, 70):
        term = term.dot(A) / LD(n)
        out = out + term
    return out


def wedge6(xi):
    W = np.zeros((4, 4), dtype=LD)
    p = np.asarray(xi[3:], dtype=LD)
    W[:3, :3] = np.array([[0, -p[2], p[1]], [p[2], 0, -p[0]], [-p[1], p[0], 0]], dtype=LD)
    W[:3, 3] = np.asarray(xi[:3], dtype=LD)
    return W


def cases():
    for th in THETAS:
        for ax in AXES:
            yield th, np.concatenate([RHO, th * ax])


def jl_tol(theta, scale=1.):
    """Absolute tolerance on J_l(phi) rho against the series.  Upstream's left Jacobian carries the term
    (1 - cos theta) / theta * axis^; in double precision 1 - cos theta has an absolute error of
    1.1e-16, and for 1e-8 < theta < 1.5e-8 -- just above the np.isclose switch to I + phi^ / 2 -- it is exactly 0: the
    formula loses phi^ / 2 there (relative error theta / 2 = 5e-9 in the translation).  That is the reference's
    arithmetic, reproduced on purpose by all three restatements; the pin is the conditioning bound 2.3e-16 / theta."""
    return scale * (4e-16 + (3e-16 / theta if theta > 1e-8 else 0.))


def log_tol(theta):
    """Absolute tolerance on log(exp(xi)) - xi.  Upstream: theta = arccos((tr C - 1) / 2), phi = theta / (2 sin theta)
    vee(C - C^T).  With d = pi - theta the arccos amplifies the rounding of the trace by 1 / d, and sin(theta) = sin(d)
    then carries that error relative to d: eps / d^2 overall (2.5e-11 observed at d = 1e-3, ~1e-4 at d = 1e-6).  The
    formula is the reference's; near pi it pins little more than the branch, and the test says so."""
    d = np.pi - theta
    return 5e-16 + (6e-16 / (d * d) if theta > 1. else 0.)


def test_small_angle_switch_is_npisclose():
    """np.isclose(angle, 0.) == (|angle| <= 1e-8): 1e-8 itself takes the first-order branch, 1.01e-8 does not."""
    assert np.isclose(1e-8, 0.) and not np.isclose(1.01e-8, 0.)
    from pyslam_amd.liegroups._base import is_small
    assert is_small(1e-8) and is_small(-1e-8) and not is_small(1.01e-8)
    assert orc.SMALL == 1e-8


def test_host_module_exp_log_jacobians_against_the_long_double_series():
    for theta, xi in cases():
        T_ld = expm_ld(wedge6(xi))
        T = SE3.exp(xi)
        # the first-order branch (theta <= 1e-8) drops theta^2 / 2 <= 5e-17: below one ulp of the entries
        assert np.abs(T.as_matrix()[:3, :3] - T_ld[:3, :3].astype(float)).max() <= 4e-16, (theta, xi)
        assert np.abs(np.asarray(T.trans) - T_ld[:3, 3].astype(float)).max() <= jl_tol(theta), (theta, xi)
        assert np.abs(SO3.exp(xi[3:]).as_matrix() - T_ld[:3, :3].astype(float)).max() <= 4e-16
        # log of the ROUNDED long-double matrix gives xi back
        Tin = SE3.from_matrix(T_ld.astype(float), normalize=False) if theta > 1e-6 else \
            SE3(SO3(T_ld[:3, :3].astype(float)), T_ld[:3, 3].astype(float))
        back = SE3.log(Tin)
        assert np.abs(back[3:] - xi[3:]).max() <= log_tol(theta), (theta, back[3:] - xi[3:])
        assert np.abs(back[:3] - xi[:3]).max() <= 4 * log_tol(theta) + 1e-15, (theta, back[:3] - xi[:3])
        # J_l: translation of exp is J_l(phi) rho; J_l^-1 is its inverse
        J, Ji = SO3.left_jacobian(xi[3:]), SO3.inv_left_jacobian(xi[3:])
        assert np.abs(J.dot(RHO) - T_ld[:3, 3].astype(float)).max() <= jl_tol(theta)
        assert np.abs(J.dot(Ji) - np.identity(3)).max() <= 1e-15 + 2e-16 / max(np.pi - theta, 1e-3) + jl_tol(theta)


def test_oracle_restatement_against_the_long_double_series():
    for theta, xi in cases():
        T_ld = expm_ld(wedge6(xi))
        R, t = orc.se_exp(xi[None, :], 6)
        assert np.abs(R[0] - T_ld[:3, :3].astype(float)).max() <= 4e-16 and np.abs(t[0] - T_ld[:3, 3].astype(float)).max() <= jl_tol(theta)
        Th = SE3.exp(xi)                                   # and entry for entry what the host module computes
        assert np.abs(R[0] - Th.rot.as_matrix()).max() <= 4e-16 and np.abs(t[0] - np.asarray(Th.trans)).max() <= 4e-16
        back = orc.se_log(T_ld[None, :3, :3].astype(float), T_ld[None, :3, 3].astype(float), 6)[0]
        assert np.abs(back[3:] - xi[3:]).max() <= log_tol(theta)
        assert np.abs(back[:3] - xi[:3]).max() <= 4 * log_tol(theta) + 1e-15


def test_host_module_and_oracle_agree_entry_for_entry():
    """Same formula, same branch, at every pinned angle -- including theta = pi exactly, where the upstream formula
    divides by sin(theta) ~ 1e-16 and has no meaningful answer: the two restatements must still take the same path."""
    for theta, xi in list(cases()) + [(np.pi, np.concatenate([RHO, np.pi * AXES[1]]))]:
        T = SE3.exp(xi)
        a = SE3.log(T)
        b = orc.se_log(T.rot.as_matrix()[None], np.asarray(T.trans)[None], 6)[0]
        assert np.all(np.isfinite(a) == np.isfinite(b))
        fin = np.isfinite(a)
        assert np.abs(a[fin] - b[fin]).max() <= 1e-12 * max(1., np.abs(a[fin]).max())


def test_se2_at_the_branch_boundary():
    for th in [1e-9, 1e-8, 1.01e-8, 1e-3, 3.0, np.pi - 1e-9]:
        xi = np.array([0.4, -0.7, th])
        W = np.zeros((3, 3), dtype=LD)
        W[0, 1], W[1, 0], W[0, 2], W[1, 2] = -LD(th), LD(th), LD(xi[0]), LD(xi[1])
        T_ld = expm_ld(W).astype(float)
        assert np.abs(SE2.exp(xi).as_matrix() - T_ld).max() <= jl_tol(th)
        R, t = orc.se_exp(xi[None, :], 3)
        assert np.abs(R[0] - T_ld[:2, :2]).max() <= 4e-16 and np.abs(t[0] - T_ld[:2, 2]).max() <= jl_tol(th)
        assert np.abs(SE2.exp(xi).as_matrix()[:2, 2] - t[0]).max() <= 4e-16
        from liegroups import SO2
        back = SE2.log(SE2(SO2(T_ld[:2, :2]), T_ld[:2, 2]))
        assert np.abs(back - xi).max() <= 2e-15
        assert np.abs(orc.se_log(T_ld[None, :2, :2], T_ld[None, :2, 2], 3)[0] - xi).max() <= 2e-15


# ---------------------------------------------------------------------------
# the device (csrc/ps_math.h: se3_log / se3_exp / se2_*), through a one-factor pose graph: with a unit stiffness and
# the L2 loss a prior on ONE pose gives H = I, b = -log(T T_obs^-1), so the Gauss-Newton step IS -log(E) for the
# 4 x 4 matrix E handed in as T_obs^-1 (T = identity), and the update applies exp(dx).
# ---------------------------------------------------------------------------
def one_prior_problem(E, dof):
    from pyslam_amd.lowering import LoweredProblem, pack_pose_matrices
    n = 4 if dof == 6 else 3
    return LoweredProblem(dof=dof, poses=pack_pose_matrices(np.identity(n)[None]), pose_rid=[0],
                          u_i=[0], u_Tobs_inv=pack_pose_matrices(np.asarray(E, dtype=float)[None]), u_grp=[0],
                          stiffd=np.identity(dof).reshape(1, dof * dof), edge_groups=np.array([[0., 0., 0.]]),
                          pose_keys=['T']).finalize()


@pytest.mark.gpu
def test_device_log_and_exp_against_the_long_double_series():
    from pyslam_amd.device import DeviceProblem
    from pyslam_amd.lowering import pose_rows_to_matrices
    for theta, xi in cases():
        T_ld = expm_ld(wedge6(xi))
        dev = DeviceProblem(one_prior_problem(T_ld.astype(float), 6))
        dev.linearize(0.)
        dev.solve_reduced(1e-15, 50)
        dev.backsub()
        dx = dev.get_dx()[0].ravel()
        assert np.abs(-dx[3:] - xi[3:]).max() <= log_tol(theta), (theta, -dx - xi)          # device log
        assert np.abs(-dx[:3] - xi[:3]).max() <= 4 * log_tol(theta) + 1e-15
        dev.apply_update(1.0)                                                            # T <- exp(dx) I
        got = pose_rows_to_matrices(dev.get_params()[0], 6)[0]
        want = expm_ld(wedge6(dx)).astype(float)                                         # the series at the device's own dx
        assert np.abs(got[:3, :3] - want[:3, :3]).max() <= 4e-16, (theta, np.abs(got - want).max())
        assert np.abs(got[:3, 3] - want[:3, 3]).max() <= jl_tol(np.linalg.norm(dx[3:])), (theta, np.abs(got - want).max())
        assert np.abs(got - SE3.exp(dx).as_matrix()).max() <= 3e-16                        # = the host module, entry for entry
        dev.close()
    for th in [1e-9, 1e-8, 1.01e-8, 1e-3, 3.0]:
        xi = np.array([0.4, -0.7, th])
        W = np.zeros((3, 3), dtype=LD)
        W[0, 1], W[1, 0], W[0, 2], W[1, 2] = -LD(th), LD(th), LD(xi[0]), LD(xi[1])
        dev = DeviceProblem(one_prior_problem(expm_ld(W).astype(float), 3))
        dev.linearize(0.)
        dev.solve_reduced(1e-15, 50)
        dev.backsub()
        dx = dev.get_dx()[0].ravel()
        assert np.abs(-dx - xi).max() <= 2e-15
        dev.apply_update(1.0)
        got = pose_rows_to_matrices(dev.get_params()[0], 3)[0]
        W2 = np.zeros((3, 3), dtype=LD)
        W2[0, 1], W2[1, 0], W2[0, 2], W2[1, 2] = -LD(dx[2]), LD(dx[2]), LD(dx[0]), LD(dx[1])
        assert np.abs(got - expm_ld(W2).astype(float)).max() <= jl_tol(abs(dx[2]))
        assert np.abs(got - SE2.exp(dx).as_matrix()).max() <= 3e-16
        dev.close()
