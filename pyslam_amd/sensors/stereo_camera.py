"""Pinhole stereo camera, origin in the left camera: (x,y,z) <-> (u,v,d).

Same public surface as reference pyslam/sensors/stereo_camera.py:7-87; the
five numba gufuncs there (:90-174) are restated as whole-array numpy
expressions.  The device restatement of project / project-Jacobian used by
the reprojection kernels is csrc/ps_camera.h.
"""
import numpy as np


def _rows3(a, what):
    a = np.atleast_2d(np.asarray(a, dtype=float))
    if a.shape[1] != 3:
        raise ValueError("{} must have shape (3,) or (N,3)".format(what))
    return a


class StereoCamera:
    CAMERA_ID = 0  # shared with csrc/ps_camera.h

    def __init__(self, cu, cv, fu, fv, b, w, h):
        self.cu = float(cu)
        self.cv = float(cv)
        self.fu = float(fu)
        self.fv = float(fv)
        self.b = float(b)
        self.w = int(w)
        self.h = int(h)

    def intrinsics(self):
        """(cu, cv, fu, fv, b) as lowered into the device camera table."""
        return np.array([self.cu, self.cv, self.fu, self.fv, self.b])

    def clone(self):
        return self.__class__(self.cu, self.cv, self.fu, self.fv, self.b, self.w, self.h)

    def compute_pixel_grid(self):
        u, v = np.meshgrid(np.arange(self.w), np.arange(self.h), indexing='xy')
        self.u_grid = u.astype(float)
        self.v_grid = v.astype(float)

    def is_valid_measurement(self, uvd):
        """Boolean mask: 0<d<w, 0<v<h, 0<u<w (one bool for a single (3,) input)."""
        uvd = _rows3(uvd, "uvd")
        ok = ((uvd[:, 2] > 0.) & (uvd[:, 2] < self.w)
              & (uvd[:, 1] > 0.) & (uvd[:, 1] < self.h)
              & (uvd[:, 0] > 0.) & (uvd[:, 0] < self.w))
        return ok

    def project(self, pt_c, compute_jacobians=None):
        """3D point(s) in the sensor frame -> (u,v,d) [, 3x3 Jacobian(s)]."""
        pt_c = _rows3(pt_c, "pt_c")
        inv_z = 1. / pt_c[:, 2]
        uvd = np.empty_like(pt_c)
        uvd[:, 0] = self.fu * pt_c[:, 0] * inv_z + self.cu
        uvd[:, 1] = self.fv * pt_c[:, 1] * inv_z + self.cv
        uvd[:, 2] = self.fu * self.b * inv_z
        if not compute_jacobians:
            return np.squeeze(uvd)

        inv_z2 = inv_z * inv_z
        jac = np.zeros((pt_c.shape[0], 3, 3))
        jac[:, 0, 0] = self.fu * inv_z
        jac[:, 0, 2] = -self.fu * pt_c[:, 0] * inv_z2
        jac[:, 1, 1] = self.fv * inv_z
        jac[:, 1, 2] = -self.fv * pt_c[:, 1] * inv_z2
        jac[:, 2, 2] = -self.fu * self.b * inv_z2
        return np.squeeze(uvd), np.squeeze(jac)

    def triangulate(self, uvd, compute_jacobians=None):
        """(u,v,d) -> 3D point(s) in the sensor frame [, 3x3 Jacobian(s)]."""
        uvd = _rows3(uvd, "uvd")
        b_over_d = self.b / uvd[:, 2]
        fu_over_fv = self.fu / self.fv
        pt = np.empty_like(uvd)
        pt[:, 0] = (uvd[:, 0] - self.cu) * b_over_d
        pt[:, 1] = (uvd[:, 1] - self.cv) * b_over_d * fu_over_fv
        pt[:, 2] = self.fu * b_over_d
        if not compute_jacobians:
            return np.squeeze(pt)

        b_over_d2 = b_over_d / uvd[:, 2]
        jac = np.zeros((uvd.shape[0], 3, 3))
        jac[:, 0, 0] = b_over_d
        jac[:, 0, 2] = (self.cu - uvd[:, 0]) * b_over_d2
        jac[:, 1, 1] = b_over_d * fu_over_fv
        jac[:, 1, 2] = (self.cv - uvd[:, 1]) * b_over_d2 * fu_over_fv
        jac[:, 2, 2] = -self.fu * b_over_d2
        return np.squeeze(pt), np.squeeze(jac)

    def __repr__(self):
        return ("{}:\n cu: {:f}\n cv: {:f}\n fu: {:f}\n fv: {:f}\n"
                "  b: {:f}\n  w: {:d}\n  h: {:d}\n").format(
                    self.__class__.__name__, self.cu, self.cv, self.fu, self.fv,
                    self.b, self.w, self.h)
