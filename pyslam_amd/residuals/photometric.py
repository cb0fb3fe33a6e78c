"""Dense photometric residual.

Constructor signature, attributes, ``evaluate`` protocol (one SE3 parameter or a (SO3, translation)
pair) and arithmetic follow reference pyslam/residuals/photometric_residual.py:38-161.  The device
restatement is csrc/ps_photo.h: a Problem whose only block is a PhotometricResidualSE3 runs its whole
Gauss-Newton iteration on the MI355X (pyslam_amd.device.PhotometricDevice); ``evaluate`` below is the
block protocol for everything else (mixed problems, inspection) and what the parity tests compare with.
"""
import numpy as np

from pyslam_amd.liegroups import SE3
from pyslam_amd.utils import bilinear_interpolate


class PhotometricResidualSE3:
    """Full SE3 photometric residual for greyscale images.  The pre-computed REFERENCE image gradient
    stands in for the tracking image's, assuming small camera motion (reference :39-41).
    ``depth_ref`` / ``depth_stiffness`` may equally be disparity (stereo camera)."""
    KIND = "photometric"

    def __init__(self, camera, im_ref, depth_ref, im_track, im_jac,
                 intensity_stiffness, depth_stiffness, min_grad=0.):
        self.camera = camera
        self.im_ref = np.asarray(im_ref, dtype=float).ravel()
        self.uvd_ref = np.vstack([camera.u_grid.ravel(), camera.v_grid.ravel(),
                                  np.asarray(depth_ref, dtype=float).ravel()]).T
        self.im_jac = np.vstack([np.asarray(im_jac[0], dtype=float).ravel(),
                                 np.asarray(im_jac[1], dtype=float).ravel()]).T
        self.im_track = im_track
        self.intensity_stiffness = intensity_stiffness
        self.depth_stiffness = depth_stiffness
        self.intensity_covar = intensity_stiffness ** -2
        self.depth_covar = depth_stiffness ** -2
        self.min_grad = min_grad

        # drop invalid pixels (NaN / non-positive depth), then pixels with weak gradients (:64-76)
        with np.errstate(invalid='ignore'):
            keep = np.asarray(camera.is_valid_measurement(self.uvd_ref), dtype=bool)
        self.uvd_ref = self.uvd_ref[keep]
        self.im_ref = self.im_ref[keep]
        self.im_jac = self.im_jac[keep]
        strong = np.linalg.norm(self.im_jac, axis=1) >= self.min_grad
        self.uvd_ref = self.uvd_ref[strong]
        self.im_ref = self.im_ref[strong]
        self.im_jac = self.im_jac[strong]

        pt, jac = camera.triangulate(self.uvd_ref, compute_jacobians=True)
        self.pt_ref = np.atleast_2d(pt)
        self.triang_jac = np.asarray(jac).reshape(-1, 3, 3)

    @staticmethod
    def _transform(params):
        if len(params) == 1:
            return params[0]
        if len(params) == 2:
            return SE3(params[0], params[1])
        raise ValueError('In PhotometricResidual.evaluate() params must have length 1 or 2')

    def evaluate(self, params, compute_jacobians=None):
        T = self._transform(params)
        R = T.rot.as_matrix()
        pt_track = self.pt_ref @ R.T + np.asarray(T.trans, dtype=float)
        with np.errstate(divide='ignore', invalid='ignore'):
            uvd_track, project_jac = self.camera.project(pt_track, compute_jacobians=True)
            uvd_track = np.atleast_2d(uvd_track)
            project_jac = np.asarray(project_jac).reshape(-1, 3, 3)
            valid = np.asarray(self.camera.is_valid_measurement(uvd_track), dtype=bool)
            im_ref_est = np.atleast_1d(bilinear_interpolate(self.im_track, uvd_track[:, 0], uvd_track[:, 1]))
            residual = (im_ref_est - self.im_ref)[valid]
            # 1 x 3 image-times-projection Jacobian and the residual's sensitivity to the reference depth (:110-121)
            im_proj_jac = np.einsum('nk,nkj->nj', self.im_jac, project_jac[:, 0:2, :])
            im_depth_jac = np.einsum('nj,nj->n', im_proj_jac @ R, self.triang_jac[:, :, 2])[valid]
        stiffness = 1. / np.sqrt(self.intensity_covar + self.depth_covar * im_depth_jac ** 2)
        residual = stiffness * residual

        if compute_jacobians:
            jac = None
            if any(compute_jacobians):
                g, p = im_proj_jac[valid], pt_track[valid]
                jac = np.empty((g.shape[0], 6))
                jac[:, 0:3] = g                                     # g [I | -p^]
                jac[:, 3] = -g[:, 1] * p[:, 2] + g[:, 2] * p[:, 1]
                jac[:, 4] = g[:, 0] * p[:, 2] - g[:, 2] * p[:, 0]
                jac[:, 5] = -g[:, 0] * p[:, 1] + g[:, 1] * p[:, 0]
                jac *= stiffness[:, None]
            if len(params) == 1:
                jacobians = [jac if compute_jacobians[0] else None]
            else:                                                   # (SO3, translation): rotation part, translation part
                jacobians = [jac[:, 3:6] if compute_jacobians[0] else None,
                             jac[:, 0:3] if compute_jacobians[1] else None]
            return residual, jacobians
        return residual

    # ---- what the device path uploads (include/pyslam_hip.h: ps_photo_desc) ----------------------
    def device_tables(self):
        cam = self.camera
        cam_type = 0 if hasattr(cam, 'b') else 1
        im = np.ascontiguousarray(np.asarray(self.im_track, dtype=float))
        if im.ndim != 2:
            raise ValueError('the device path takes a single-channel tracking image')
        return dict(pt_ref=np.ascontiguousarray(self.pt_ref, dtype=float),
                    im_ref=np.ascontiguousarray(self.im_ref, dtype=float),
                    im_jac=np.ascontiguousarray(self.im_jac, dtype=float),
                    tri_jac_d=np.ascontiguousarray(self.triang_jac[:, :, 2], dtype=float),
                    im_track=im, cam=np.array([cam.cu, cam.cv, cam.fu, cam.fv, getattr(cam, 'b', 0.)], dtype=float),
                    cam_type=cam_type, cam_w=int(cam.w), cam_h=int(cam.h),
                    intensity_covar=float(self.intensity_covar), depth_covar=float(self.depth_covar))
