"""Pinhole RGB-D camera: (x,y,z) <-> (u,v,z).

Public surface of reference pyslam/sensors/rgbd_camera.py:7-86 (SURVEY.md
section 8f rank 2); the device restatement shares the stereo reprojection
kernels (csrc/ps_math.h, cam_type = 1).
"""
import numpy as np

from .stereo_camera import _rows3


class RGBDCamera:
    CAMERA_ID = 1

    def __init__(self, cu, cv, fu, fv, w, h):
        self.cu = float(cu)
        self.cv = float(cv)
        self.fu = float(fu)
        self.fv = float(fv)
        self.w = int(w)
        self.h = int(h)

    def intrinsics(self):
        """(cu, cv, fu, fv, -1): the device camera table marks RGB-D rows with b = -1."""
        return np.array([self.cu, self.cv, self.fu, self.fv, -1.0])

    def clone(self):
        return self.__class__(self.cu, self.cv, self.fu, self.fv, self.w, self.h)

    def compute_pixel_grid(self):
        u, v = np.meshgrid(np.arange(self.w), np.arange(self.h), indexing='xy')
        self.u_grid = u.astype(float)
        self.v_grid = v.astype(float)

    def is_valid_measurement(self, uvz):
        uvz = _rows3(uvz, "uvz")
        return ((uvz[:, 2] > 0.)
                & (uvz[:, 1] > 0.) & (uvz[:, 1] < self.h)
                & (uvz[:, 0] > 0.) & (uvz[:, 0] < self.w))

    def project(self, pt_c, compute_jacobians=None):
        pt_c = _rows3(pt_c, "pt_c")
        inv_z = 1. / pt_c[:, 2]
        uvz = np.empty_like(pt_c)
        uvz[:, 0] = self.fu * pt_c[:, 0] * inv_z + self.cu
        uvz[:, 1] = self.fv * pt_c[:, 1] * inv_z + self.cv
        uvz[:, 2] = pt_c[:, 2]
        if not compute_jacobians:
            return np.squeeze(uvz)
        inv_z2 = inv_z * inv_z
        jac = np.zeros((pt_c.shape[0], 3, 3))
        jac[:, 0, 0] = self.fu * inv_z
        jac[:, 0, 2] = -self.fu * pt_c[:, 0] * inv_z2
        jac[:, 1, 1] = self.fv * inv_z
        jac[:, 1, 2] = -self.fv * pt_c[:, 1] * inv_z2
        jac[:, 2, 2] = 1.
        return np.squeeze(uvz), np.squeeze(jac)

    def triangulate(self, uvz, compute_jacobians=None):
        uvz = _rows3(uvz, "uvz")
        z = uvz[:, 2]
        pt = np.empty_like(uvz)
        pt[:, 0] = (uvz[:, 0] - self.cu) * z / self.fu
        pt[:, 1] = (uvz[:, 1] - self.cv) * z / self.fv
        pt[:, 2] = z
        if not compute_jacobians:
            return np.squeeze(pt)
        jac = np.zeros((uvz.shape[0], 3, 3))
        inv_fu, inv_fv = 1. / self.fu, 1. / self.fv
        jac[:, 0, 0] = z * inv_fu
        jac[:, 0, 2] = (uvz[:, 0] - self.cu) * inv_fu
        jac[:, 1, 1] = z * inv_fv
        jac[:, 1, 2] = (uvz[:, 1] - self.cv) * inv_fv
        jac[:, 2, 2] = 1.
        return np.squeeze(pt), np.squeeze(jac)

    def __repr__(self):
        return ("{}:\n cu: {:f}\n cv: {:f}\n fu: {:f}\n fv: {:f}\n"
                "  w: {:d}\n  h: {:d}\n").format(
                    self.__class__.__name__, self.cu, self.cv, self.fu, self.fv,
                    self.w, self.h)
