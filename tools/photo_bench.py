"""Dense photometric alignment at 640 x 480: one Gauss-Newton iteration on the MI355X (ps_photometric_iteration,
host round trip included) vs the numpy product-class evaluate + normal equations on the host."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from pyslam_amd import synthetic
from pyslam_amd.device import PhotometricDevice
from pyslam_amd.liegroups import SE3
from pyslam_amd.losses import HuberLoss
from pyslam_amd.residuals import PhotometricResidualSE3
from pyslam_amd.sensors import StereoCamera

sc = synthetic.photometric_scene(h=480, w=640, seed=9, xi_true=(0.02, -0.01, 0.03, 0.004, -0.006, 0.008), noise=0.2)
cu, cv, fu, fv, b, w, h = sc['cam']
cam = StereoCamera(cu, cv, fu, fv, b, int(w), int(h)); cam.compute_pixel_grid()
blk = PhotometricResidualSE3(cam, sc['im_ref'], sc['depth_ref'], sc['im_track'], sc['im_jac'], 1.0, 2.0, min_grad=0.02)
loss = HuberLoss(10.0)
t = time.perf_counter(); dev = PhotometricDevice(blk, loss, False); t_create = time.perf_counter() - t
dev.set_pose(np.eye(3), np.zeros(3))
for _ in range(3):
    dev.step(False)
n = 50
t = time.perf_counter()
for _ in range(n):
    dev.step(False)
gpu = (time.perf_counter() - t) / n
t = time.perf_counter()
for _ in range(n):
    dev.normal_equations()
gpu_ne = (time.perf_counter() - t) / n
T = SE3.identity()
t = time.perf_counter()
for _ in range(3):
    r, J = blk.evaluate([T], [True])
    s = np.sqrt(loss.weight(r))
    Jw = J[0] * s[:, None]
    dx = np.linalg.solve(Jw.T @ Jw, -Jw.T @ (s * r))
cpu = (time.perf_counter() - t) / 3
px = dev.num_pixels
print('pixels %d  create %.1f ms  GPU iteration %.3f ms (normal equations only %.3f ms; %.1f GB/s of pixel tables)  '
      'host numpy iteration %.1f ms  (%.0fx)' % (px, t_create * 1e3, gpu * 1e3, gpu_ne * 1e3, px * 72 / gpu_ne / 1e9, cpu * 1e3, cpu / gpu))
