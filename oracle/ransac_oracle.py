"""CPU restatement of the reference's frame-to-frame RANSAC (pyslam/pipelines/ransac.py).

TEST INFRASTRUCTURE ONLY: imported by tests/ (including the tests/bench_*.py timing scripts) and
__graft_entry__.smoke() as the checker / CPU baseline, never by the product path (pyslam_amd has no CPU path).
Pinned against tests/golden/ransac.npz, which oracle/gen_golden.py produced by running the
verbatim reference (FrameToFrameRANSAC.perform_ransac under a fixed numpy seed).
"""
import numpy as np


def compute_transform(pts_1, pts_2):
    """(..., n, 3) x2 -> (..., 4, 4): reference ransac.py:13-67 (Barfoot's SVD alignment)."""
    pts_1 = np.asarray(pts_1, dtype=float)
    pts_2 = np.asarray(pts_2, dtype=float)
    lead = pts_1.shape[:-2]
    a = pts_1.reshape((-1,) + pts_1.shape[-2:])
    b = pts_2.reshape((-1,) + pts_2.shape[-2:])
    out = np.zeros((a.shape[0], 4, 4))
    for k in range(a.shape[0]):
        c1, c2 = a[k].mean(axis=0), b[k].mean(axis=0)            # :22-25
        W = (1.0 / a.shape[1]) * np.dot((b[k] - c2).T, a[k] - c1)  # :27-30
        U, _, V = np.linalg.svd(W)                               # :32  (V is V^T)
        S = np.identity(3)
        S[2, 2] = np.linalg.det(U) * np.linalg.det(V)            # :33-34
        C = U.dot(S).dot(V)                                      # :36
        out[k, :3, :3] = C
        out[k, :3, 3] = c2 - C.dot(c1)                           # :37
        out[k, 3, 3] = 1.
    return out.reshape(lead + (4, 4))


def project(pts, cam5):
    """Stereo (u, v, d) projection, reference sensors/stereo_camera.py:100-108; b < 0: RGB-D (u, v, z)."""
    cu, cv, fu, fv, b = cam5
    with np.errstate(divide='ignore', invalid='ignore'):
        iz = 1. / pts[:, 2]
        third = pts[:, 2] if b < 0 else fu * b * iz
        return np.stack([fu * pts[:, 0] * iz + cu, fv * pts[:, 1] * iz + cv, third], axis=1)


def ransac_cost(T_stacked, pts_1, obs_2, cam5, thresh):
    """Boolean (H, N) inlier masks, reference ransac.py:153-165."""
    out = np.zeros((len(T_stacked), len(pts_1)), dtype=bool)
    for h, T in enumerate(T_stacked):
        pred = project(T[:3, :3].dot(pts_1.T).T + T[:3, 3], cam5)
        with np.errstate(invalid='ignore'):
            out[h] = ((pred - obs_2) ** 2).sum(axis=1) < thresh
    return out


def perform_ransac(pts_1, pts_2, obs_2, rand_idx, cam5, thresh):
    """reference ransac.py:113-151 for given minimal sets -> (T_all, counts, best index, best mask)."""
    T_all = compute_transform(pts_1[rand_idx], pts_2[rand_idx])
    masks = ransac_cost(T_all, pts_1, obs_2, cam5, thresh)
    counts = masks.sum(axis=1)
    best = int(np.argmax(counts))
    return T_all, counts, best, masks[best]


def sample_conditioning(pts_1, pts_2, rand_idx):
    """sigma_2 / sigma_1 of every hypothesis' 3x3 cross-covariance: ~0 means the rotation about one
    axis is undetermined (collinear / repeated sample points) and SVD implementations may differ."""
    a, b = pts_1[rand_idx], pts_2[rand_idx]
    W = np.einsum('hni,hnj->hij', b - b.mean(axis=1, keepdims=True), a - a.mean(axis=1, keepdims=True))
    s = np.linalg.svd(W, compute_uv=False)
    return s[:, 1] / np.maximum(s[:, 0], 1e-300)
