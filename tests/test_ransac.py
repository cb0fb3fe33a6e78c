"""Frame-to-frame RANSAC (reference pyslam/pipelines/ransac.py).

CPU: the numpy oracle against the golden vectors the verbatim reference produced.
GPU: the HIP path (through the C ABI and the reference-named Python classes) against both."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import ransac_oracle as orc

WELL = 1e-3          # sigma_2 / sigma_1 above which a 3-point hypothesis is well determined


def well_conditioned(g):
    return orc.sample_conditioning(g['pts_1'], g['pts_2'], g['rand_idx']) > WELL


def test_oracle_matches_reference_golden():
    g = load_golden('ransac')
    T_all, counts, best, mask = orc.perform_ransac(g['pts_1'], g['pts_2'], g['obs_2'], g['rand_idx'], g['cam'][:5],
                                                   float(g['thresh']))
    ok = well_conditioned(g)
    assert ok.sum() > 350
    assert np.abs(T_all[ok] - g['T_stacked'][ok]).max() < 1e-9
    assert np.array_equal(counts[ok], g['inlier_counts'][ok])
    assert np.allclose(T_all[best], g['T_best'], atol=1e-12)
    assert np.array_equal(np.where(mask)[0], g['inlier_indices'])
    # the winner is a real solution: close to the generating motion, and it rejects the planted outliers
    assert np.abs(g['T_best'] - g['T_true']).max() < 0.05
    assert not np.intersect1d(g['inlier_indices'], g['outliers']).size


def test_oracle_degenerate_samples_do_not_crash():
    g = load_golden('ransac')
    idx = np.array([[5, 5, 5], [1, 1, 7], [0, 1, 2]])
    T = orc.compute_transform(g['pts_1'][idx], g['pts_2'][idx])
    assert np.isfinite(T).all()
    assert np.allclose(T[0, :3, :3], np.identity(3))           # W = 0: LAPACK returns U = V = I


@pytest.mark.gpu
def test_device_hypotheses_match_reference_golden():
    from pyslam_amd.pipelines.ransac import FrameToFrameRANSAC, compute_transform_fast
    from pyslam.sensors import StereoCamera
    g = load_golden('ransac')
    cam = StereoCamera(*g['cam'][:5], int(g['cam'][5]), int(g['cam'][6]))
    r = FrameToFrameRANSAC(cam)
    r.set_obs(g['obs_1'], g['obs_2'])
    assert np.allclose(r.pts_1, g['pts_1'], rtol=1e-14) and np.allclose(r.pts_2, g['pts_2'], rtol=1e-14)
    T_best, mask, best, count, T_all, counts = r._device_ransac(g['rand_idx'], want_all=True)
    ok = well_conditioned(g)
    assert np.abs(T_all[ok] - g['T_stacked'][ok]).max() < 1e-9       # fp64 tolerance: different SVD algorithm
    assert np.array_equal(counts[ok], g['inlier_counts'][ok])
    assert best == int(np.argmax(g['inlier_counts'])) and count == len(g['inlier_indices'])
    assert np.abs(T_best - g['T_best']).max() < 1e-10
    assert np.array_equal(np.where(mask)[0], g['inlier_indices'])
    # the batch alignment entry point, broadcasting over leading dimensions like the guvectorize original
    T2 = compute_transform_fast(g['pts_1'][g['rand_idx']].reshape(20, 20, 3, 3), g['pts_2'][g['rand_idx']].reshape(20, 20, 3, 3))
    assert T2.shape == (20, 20, 4, 4) and np.array_equal(T2.reshape(-1, 4, 4), T_all)
    # scoring of caller-provided transforms
    masks = r.compute_ransac_cost(g['T_stacked'], r.pts_1, r.obs_2, cam, r.ransac_thresh)
    assert masks.shape == (400, 300) and np.array_equal(masks.sum(axis=1), g['inlier_counts'])


@pytest.mark.gpu
def test_perform_ransac_is_a_drop_in_under_the_same_seed():
    from pyslam.pipelines.ransac import FrameToFrameRANSAC
    from pyslam.sensors import StereoCamera
    g = load_golden('ransac')
    cam = StereoCamera(*g['cam'][:5], int(g['cam'][5]), int(g['cam'][6]))
    r = FrameToFrameRANSAC(cam)
    r.set_obs(g['obs_1'], g['obs_2'])
    np.random.seed(int(g['seed']))
    T_21, o1, o2, inl = r.perform_ransac()
    assert np.abs(T_21.as_matrix() - g['T_best']).max() < 1e-10
    assert np.array_equal(inl, g['inlier_indices'])
    assert np.array_equal(o1, g['obs_1_inliers']) and np.array_equal(o2, g['obs_2_inliers'])
    r.ransac_thresh = 1e-9
    with pytest.raises(ValueError):
        r.perform_ransac()


@pytest.mark.gpu
def test_device_many_points_alignment_and_degenerate_sets():
    """n-point alignment (n = 50) against the oracle; repeated / collinear minimal sets stay finite."""
    from pyslam_amd.pipelines.ransac import compute_transform_fast
    rng = np.random.default_rng(5)
    a = rng.standard_normal((7, 50, 3)) * 3.
    from liegroups import SE3
    b = np.stack([SE3.exp(0.3 * rng.standard_normal(6)).dot(a[k]) + 0.01 * rng.standard_normal((50, 3)) for k in range(7)])
    T = compute_transform_fast(a, b)
    assert np.abs(T - orc.compute_transform(a, b)).max() < 1e-12
    g = load_golden('ransac')
    idx = np.array([[5, 5, 5], [1, 1, 7], [0, 1, 2]])
    Td = compute_transform_fast(g['pts_1'][idx], g['pts_2'][idx])
    assert np.isfinite(Td).all()
    assert np.allclose(Td[0, :3, :3], np.identity(3))
    for k in range(3):                                           # always a proper rotation
        C = Td[k, :3, :3]
        assert np.allclose(C.dot(C.T), np.identity(3), atol=1e-12) and abs(np.linalg.det(C) - 1.) < 1e-12
    # a reflection-prone case (noisy, nearly planar sets): the det(U) det(V) correction
    flat = rng.standard_normal((4, 6, 3)) * np.array([1., 1., 1e-3])
    other = rng.standard_normal((4, 6, 3)) * np.array([1., 1., 1e-3])
    assert np.abs(compute_transform_fast(flat, other) - orc.compute_transform(flat, other)).max() < 1e-9
