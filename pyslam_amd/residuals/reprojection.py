"""Reprojection residual blocks (stereo: 3 rows u,v,d per observation).

Constructor signatures / attributes follow reference
pyslam/residuals/reprojection_residual.py:6-37 and
reprojection_motion_only_residual.py:35-113.  The device restatement is
csrc/ps_reproj.h; ``KIND`` is the lowering tag.
"""
import numpy as np

from pyslam_amd.liegroups import SE3


def _jac_list(params):
    return [None] * len(params)



SE3_ODOT_SHAPE = np.empty(6)
"""Dummy second argument of ``fast_se3_odot`` (reference reprojection_motion_only_residual.py:9, 96): the reference's
numba gufunc needs an input carrying the output's second core dimension; kept so that callers' code runs unchanged."""


def fast_se3_odot(vec, junk=SE3_ODOT_SHAPE, out=None):
    """``SE3.odot`` of a stack of points: (..., 3) -> (..., 3, 6), ``[I | -p^]`` per point -- the reference's
    ``fast_se3_odot`` gufunc, layout '(n),(m)->(n,m)' (reprojection_motion_only_residual.py:12-32, also
    photometric_residual.py:15-35).  ``junk`` only has to have 6 entries (the gufunc's way of naming the output width)."""
    vec = np.asarray(vec, dtype=float)
    if vec.shape[-1] != 3 or np.shape(junk)[-1] != 6:
        raise ValueError('fast_se3_odot: vec must be (..., 3) and junk (6,)')
    if out is None:
        out = np.empty(vec.shape[:-1] + (3, 6))
    out[..., :, :3] = np.identity(3)
    out[..., 0, 3] = 0.
    out[..., 0, 4] = vec[..., 2]
    out[..., 0, 5] = -vec[..., 1]
    out[..., 1, 3] = -vec[..., 2]
    out[..., 1, 4] = 0.
    out[..., 1, 5] = vec[..., 0]
    out[..., 2, 3] = vec[..., 1]
    out[..., 2, 4] = -vec[..., 0]
    out[..., 2, 5] = 0.
    return out


class ReprojectionResidual:
    """r = S (project(T_cam_w p_w) - obs);  J_T = S Jc [I | -p_c^],  J_p = S Jc R."""
    KIND = "reproj"

    def __init__(self, camera, obs, stiffness):
        self.camera = camera
        self.obs = obs
        self.stiffness = stiffness

    def evaluate(self, params, compute_jacobians=None):
        T_cam_w, pt_w = params[0], params[1]
        pt_cam = T_cam_w.dot(pt_w)
        if not compute_jacobians:
            return np.dot(self.stiffness, self.camera.project(pt_cam) - self.obs)

        predicted, cam_jac = self.camera.project(pt_cam, compute_jacobians=True)
        residual = np.dot(self.stiffness, predicted - self.obs)
        jacobians = _jac_list(params)
        if compute_jacobians[0]:
            jacobians[0] = np.dot(self.stiffness, cam_jac.dot(SE3.odot(pt_cam)))
        if compute_jacobians[1]:
            jacobians[1] = np.dot(self.stiffness, cam_jac.dot(T_cam_w.rot.as_matrix()))
        return residual, jacobians


class ReprojectionMotionOnlyResidual:
    """Frame-to-frame reprojection with the point fixed at triangulate(obs_1)."""
    KIND = "reproj_motion_only"

    def __init__(self, camera, obs_1, obs_2, stiffness):
        self.camera = camera
        self.obs_1 = obs_1
        self.obs_2 = obs_2
        self.stiffness = stiffness
        self.pt_1 = self.camera.triangulate(self.obs_1)

    def evaluate(self, params, compute_jacobians=None):
        T_2_1 = params[0]
        pt_2 = T_2_1.dot(self.pt_1)
        if not compute_jacobians:
            return np.dot(self.stiffness, self.camera.project(pt_2) - self.obs_2)

        predicted, cam_jac = self.camera.project(pt_2, compute_jacobians=True)
        residual = np.dot(self.stiffness, predicted - self.obs_2)
        jacobians = _jac_list(params)
        if compute_jacobians[0]:
            jacobians[0] = np.dot(self.stiffness, cam_jac.dot(SE3.odot(pt_2)))
        return residual, jacobians


# The reference example stereo_ba_frame_to_frame.py:6 imports this name, which
# the reference package never defines.
ReprojectionResidualFrameToFrame = ReprojectionMotionOnlyResidual


class ReprojectionMotionOnlyBatchResidual:
    """N frame-to-frame reprojections in ONE block: residual (3N,), Jacobian (3N,6)."""
    KIND = "reproj_motion_only_batch"

    def __init__(self, camera, obs_1, obs_2, stiffness):
        self.camera = camera
        self.obs_1 = obs_1
        self.obs_2 = obs_2
        self.stiffness = stiffness
        self.pts_1 = self.camera.triangulate(self.obs_1)
        self.num_pts = self.pts_1.shape[0]

    def evaluate(self, params, compute_jacobians=None):
        T_2_1 = params[0]
        pts_2 = T_2_1.dot(self.pts_1)
        S = np.asarray(self.stiffness)
        if not compute_jacobians:
            err = self.camera.project(pts_2) - self.obs_2
            return err.dot(S.T).reshape(3 * self.num_pts)

        predicted, cam_jac = self.camera.project(pts_2, compute_jacobians=True)
        residual = (predicted - self.obs_2).dot(S.T).reshape(3 * self.num_pts)
        jacobians = _jac_list(params)
        if compute_jacobians[0]:
            per_point = np.matmul(S, np.matmul(cam_jac, SE3.odot(pts_2)))
            jacobians[0] = per_point.reshape(3 * self.num_pts, 6)
        return residual, jacobians
