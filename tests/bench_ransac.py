"""Frame-to-frame RANSAC timing: device call (host buffers in/out, so PCIe + allocation inclusive)
vs the numpy oracle (a port of the reference loop, pyslam/pipelines/ransac.py:113-165) on the host."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from liegroups import SE3
from pyslam.sensors import StereoCamera
from pyslam_amd.pipelines.ransac import FrameToFrameRANSAC
from pyslam_amd import synthetic
from oracle import ransac_oracle as orc

for N in (256, 2048, 16384):
    rng = np.random.default_rng(3)
    cam = StereoCamera(*synthetic.STEREO_BA_CAMERA)
    T = SE3.exp(np.array([0.3, -0.05, 0.1, 0.02, -0.04, 0.03]))
    p1 = np.stack([rng.uniform(-6, 6, N), rng.uniform(-3, 3, N), rng.uniform(5, 30, N)], axis=1)
    o1 = cam.project(p1) + 0.2 * rng.standard_normal((N, 3))
    o2 = cam.project(T.dot(p1)) + 0.2 * rng.standard_normal((N, 3))
    bad = rng.choice(N, N // 5, replace=False)
    o2[bad, :2] += rng.uniform(20, 60, (bad.size, 2))
    r = FrameToFrameRANSAC(cam)
    r.set_obs(o1, o2)
    idx = rng.integers(0, N, size=(400, 3))
    for _ in range(3):
        out = r._device_ransac(idx)
    t0 = time.perf_counter()
    for _ in range(20):
        out = r._device_ransac(idx)
    t_dev = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter()
    T_all, counts, best, mask = orc.perform_ransac(r.pts_1, r.pts_2, r.obs_2, idx, cam.intrinsics(), 5.)
    t_cpu = time.perf_counter() - t0
    assert best == out[2] and np.array_equal(mask, out[1])
    print('N = {:6d}, 400 hypotheses: device call {:.3f} ms (host buffers in/out), numpy oracle {:.1f} ms, inliers {}'.format(
        N, t_dev * 1e3, t_cpu * 1e3, out[3]))
