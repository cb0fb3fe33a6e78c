"""Pose residual blocks on SE(2)/SE(3): unary prior and binary relative pose.

Protocol and constructor signatures follow reference
pyslam/residuals/pose_residual.py:4-27 and pose_to_pose_residual.py:4-32.
Both use the reference's *approximate* Jacobians (identity / -Ad), which drop
the J_l^-1(e) factor -- per-iteration parity needs the
same approximation, so do not "fix" it.

Each class carries ``KIND``: the tag ``pyslam_amd.lowering`` dispatches on to
place the block in the device edge tables (csrc/ps_posegraph.hip).
"""
import numpy as np


def _jac_list(params):
    return [None] * len(params)


class PoseResidual:
    """r = S log(T_est T_obs^-1);  dr/dT ~= S."""
    KIND = "pose_prior"

    def __init__(self, T_obs, stiffness):
        self.T_obs = T_obs
        self.stiffness = stiffness
        self.obstype = type(T_obs)

    def evaluate(self, params, compute_jacobians=None):
        T_est = params[0]
        err = self.obstype.log(T_est.dot(self.T_obs.inv()))
        residual = np.dot(self.stiffness, err)
        if not compute_jacobians:
            return residual

        jacobians = _jac_list(params)
        if compute_jacobians[0]:
            jacobians[0] = np.dot(self.stiffness, np.identity(self.obstype.dof))
        return residual, jacobians


class PoseToPoseResidual:
    """r = S log(T_2 (T_1^-1 T_21_obs^-1));  J_1 ~= -S Ad(T_2 T_1^-1), J_2 ~= S."""
    KIND = "pose_pose"

    def __init__(self, T_2_1_obs, stiffness):
        self.T_2_1_obs = T_2_1_obs
        self.stiffness = stiffness
        self.obstype = type(T_2_1_obs)

    def evaluate(self, params, compute_jacobians=None):
        T_1_0, T_2_0 = params[0], params[1]
        T_1_0_inv = T_1_0.inv()
        err = self.obstype.log(T_2_0.dot(T_1_0_inv.dot(self.T_2_1_obs.inv())))
        residual = np.dot(self.stiffness, err)
        if not compute_jacobians:
            return residual

        jacobians = _jac_list(params)
        if compute_jacobians[0]:
            jacobians[0] = np.dot(self.stiffness, -T_2_0.dot(T_1_0_inv).adjoint())
        if compute_jacobians[1]:
            jacobians[1] = np.dot(self.stiffness, np.identity(self.obstype.dof))
        return residual, jacobians


class PoseToPoseOrientationResidual:
    """Rotation-only relative residual on SE(3) poses (3 rows, 3x6 Jacobians).

    Reference pyslam/residuals/pose_to_pose_orientation_residual.py:4-38
    (host-side this round).
    """
    KIND = "pose_pose_orientation"

    def __init__(self, C_2_1_obs, stiffness):
        self.C_2_1_obs = C_2_1_obs
        self.stiffness = stiffness
        self.obstype = type(C_2_1_obs)

    def evaluate(self, params, compute_jacobians=None):
        T_1_0, T_2_0 = params[0], params[1]
        C_21 = T_2_0.dot(T_1_0.inv()).rot
        residual = np.dot(self.stiffness,
                          self.obstype.log(C_21.dot(self.C_2_1_obs.inv())))
        if not compute_jacobians:
            return residual

        jacobians = _jac_list(params)
        if compute_jacobians[0]:
            sel = np.zeros((3, 6))
            sel[:, 3:] = C_21.as_matrix()
            jacobians[0] = np.dot(self.stiffness, -sel)
        if compute_jacobians[1]:
            sel = np.zeros((3, 6))
            sel[:, 3:] = np.identity(3)
            jacobians[1] = np.dot(self.stiffness, sel)
        return residual, jacobians
