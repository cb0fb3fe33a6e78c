"""Frame-to-frame RANSAC on the device -- same names, arguments and results as the reference's
pyslam/pipelines/ransac.py (compute_transform_fast :13-67, FrameToFrameRANSAC :97-165).

The random minimal sets are drawn on the host with ``np.random.randint`` exactly as the reference
does (:118-119), so a seeded run picks the same hypotheses; everything else -- the 3-point rigid
alignments, the scoring of every hypothesis over every point, the arg-max and the inlier mask --
is one call into the HIP core (ps_ransac_frame_to_frame).  There is no CPU path."""
import ctypes as C

import numpy as np

from pyslam_amd import _native as nat

SE3_SHAPE = np.empty(4)


def _cam5(camera):
    if hasattr(camera, 'intrinsics'):
        return np.ascontiguousarray(camera.intrinsics(), dtype=np.float64)
    return np.array([camera.cu, camera.cv, camera.fu, camera.fv, camera.b], dtype=np.float64)


def compute_transform_fast(pts_1, pts_2, dummy=SE3_SHAPE):
    """SE(3) alignment p_2 ~ T_21 p_1 of point sets by SVD; broadcasts over leading dimensions:
    (..., n, 3), (..., n, 3) -> (..., 4, 4)   (reference ransac.py:13-67)."""
    nat.require_gpu()
    a = np.ascontiguousarray(pts_1, dtype=np.float64)
    b = np.ascontiguousarray(pts_2, dtype=np.float64)
    if a.shape != b.shape or a.ndim < 2 or a.shape[-1] != 3:
        raise ValueError("pts_1 and pts_2 must both have shape (..., n, 3)")
    lead, n = a.shape[:-2], a.shape[-2]
    batch = int(np.prod(lead)) if lead else 1
    out = np.zeros((batch, 4, 4))
    nat.check(nat.load().ps_ransac_transforms(nat.f64p(a), nat.f64p(b), batch, n, nat.f64p(out)))
    return out.reshape(lead + (4, 4))


class FrameToFrameRANSAC:
    def __init__(self, camera):
        self.camera = camera
        self.ransac_iters = 400
        self.ransac_thresh = 5  # (1**2 + 1**2 + 1**2)
        self.num_min_set_pts = 3

    def set_obs(self, obs_1, obs_2):
        self.obs_1 = np.atleast_2d(obs_1)
        self.obs_2 = np.atleast_2d(obs_2)
        self.pts_1 = np.atleast_2d(self.camera.triangulate(self.obs_1))
        self.pts_2 = np.atleast_2d(self.camera.triangulate(self.obs_2))
        self.num_pts = self.pts_1.shape[0]

    def perform_ransac(self):
        """(T_21_best, obs_1_inliers, obs_2_inliers, inlier_indices_best); ValueError below 5 inliers."""
        from liegroups import SE3
        nat.require_gpu()
        rand_idx = np.random.randint(self.num_pts, size=(self.ransac_iters, self.num_min_set_pts))
        T_best, mask, best, count = self._device_ransac(rand_idx)[:4]
        if count < 5:
            raise ValueError(
                " RANSAC failed to find more than 5 inliers. Try adjusting the thresholds.")
        inlier_indices_best = np.where(mask)[0]
        return (SE3.from_matrix(T_best), self.obs_1[inlier_indices_best],
                self.obs_2[inlier_indices_best], inlier_indices_best)

    def _device_ransac(self, rand_idx, want_all=False):
        idx = np.ascontiguousarray(rand_idx, dtype=np.int32)
        H, k = idx.shape
        p1 = np.ascontiguousarray(self.pts_1, dtype=np.float64)
        p2 = np.ascontiguousarray(self.pts_2, dtype=np.float64)
        o2 = np.ascontiguousarray(self.obs_2, dtype=np.float64)
        T_best = np.zeros((4, 4))
        mask = np.zeros(self.num_pts, dtype=np.uint8)
        T_all = np.zeros((H, 4, 4)) if want_all else None
        counts = np.zeros(H, dtype=np.int32) if want_all else None
        best, count = C.c_int32(), C.c_int32()
        nat.check(nat.load().ps_ransac_frame_to_frame(
            nat.f64p(p1), nat.f64p(p2), nat.f64p(o2), self.num_pts, nat.i32p(idx), H, k, nat.f64p(_cam5(self.camera)),
            float(self.ransac_thresh), nat.f64p(T_all), nat.i32p(counts), C.byref(best), C.byref(count),
            nat.f64p(T_best), mask.ctypes.data_as(nat.c_u8p)))
        return T_best, mask.astype(bool), best.value, count.value, T_all, counts

    def compute_ransac_cost(self, T_21_stacked, pts_1, obs_2, camera, inlier_thresh):
        """Boolean inlier mask (num_transforms, num_pts)   (reference ransac.py:153-165)."""
        nat.require_gpu()
        T = np.ascontiguousarray(T_21_stacked, dtype=np.float64).reshape(-1, 4, 4)
        p1 = np.ascontiguousarray(np.atleast_2d(pts_1), dtype=np.float64)
        o2 = np.ascontiguousarray(np.atleast_2d(obs_2), dtype=np.float64)
        masks = np.zeros((T.shape[0], p1.shape[0]), dtype=np.uint8)
        nat.check(nat.load().ps_ransac_cost(nat.f64p(T), T.shape[0], nat.f64p(p1), nat.f64p(o2), p1.shape[0],
                                            nat.f64p(_cam5(camera)), float(inlier_thresh),
                                            masks.ctypes.data_as(nat.c_u8p), None))
        return masks.astype(bool)
