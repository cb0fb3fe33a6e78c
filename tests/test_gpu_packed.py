"""Landmark pass and back-substitution with the wave's lanes packed by observation (csrc/ps_k_packed.h, option lm_packed,
default) against the 16-lanes-per-landmark kernels they replace (lm_packed 0) and against the oracle: landmark factors, the
reduced system, the step and whole iterations.  Counterpart of reference pyslam/problem.py:338-360 (per-block J~, r~) +
:186 (the landmark part of the solve)."""
import numpy as np
import pytest

from oracle import gn_oracle as orc
from pyslam_amd import synthetic, losses

pytestmark = pytest.mark.gpu


def ragged_ba(seed, num_kf=30, num_lm=700, max_obs=16, const_point_fraction=0.05, loss=None):
    """A stereo BA whose landmarks have 2 .. max_obs observations each (the synthetic generator gives every landmark the same
    number: rows are dropped at random), so that runs of landmarks start and end anywhere inside a wave."""
    lp, _ = synthetic.stereo_ba(num_kf=num_kf, num_lm=num_lm, obs_per_lm=max_obs, half_window=max(8, max_obs // 2 + 1), seed=seed,
                                const_point_fraction=const_point_fraction, loss=loss)
    rng = np.random.default_rng(seed + 100)
    keep_n = rng.integers(2, max_obs + 1, size=lp.num_points)
    order = np.argsort(lp.obs_point, kind='stable')
    rank_in_lm = np.empty(lp.num_obs, dtype=np.int64)
    pts = lp.obs_point[order]
    starts = np.r_[0, np.flatnonzero(np.diff(pts)) + 1]
    lens = np.diff(np.r_[starts, pts.size])
    rank_in_lm[order] = np.arange(pts.size) - np.repeat(starts, lens)
    keep = rank_in_lm < keep_n[lp.obs_point]
    out = lp.copy()
    out.obs_pose, out.obs_point = lp.obs_pose[keep], lp.obs_point[keep]
    out.obs_uvd, out.obs_grp = lp.obs_uvd[keep], lp.obs_grp[keep]
    return out.finalize()


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


CASES = [
    ('uniform10', lambda: synthetic.stereo_ba(num_kf=40, num_lm=3000, obs_per_lm=10, half_window=8, seed=2)[0]),
    ('ragged16', lambda: ragged_ba(5)),
    ('ragged7_huber', lambda: ragged_ba(6, max_obs=7, loss=losses.HuberLoss(1.5))),
    ('two_obs', lambda: synthetic.stereo_ba(num_kf=12, num_lm=500, obs_per_lm=2, half_window=3, seed=9)[0]),
    ('sixteen', lambda: synthetic.stereo_ba(num_kf=40, num_lm=300, obs_per_lm=16, half_window=10, seed=4, const_point_fraction=0.1)[0]),
]


@pytest.mark.parametrize('name,make', CASES, ids=[c[0] for c in CASES])
def test_packed_kernels_equal_the_16_lane_kernels_and_the_oracle(name, make):
    from pyslam_amd.device import DeviceProblem
    lp = make()
    out = {}
    for packed in (1, 0):
        dev = DeviceProblem(lp)
        dev.set_option('lm_packed', packed)
        dev.set_option('lagged_inverse', 0)
        dev.linearize(0.0)
        S, g = dev.reduced_dense()
        cinv, c = dev.landmark_factors()
        its, relres = dev.solve_reduced(1e-13, 4000)
        dev.backsub()
        xp, xl = dev.get_dx()
        trace = [dev.gn_iteration(0.0, 1e-12, 4000, True) for _ in range(3)]
        out[packed] = (S, g, cinv, c, xp, xl, trace, dev.get_params())
        dev.close()
    a, b = out[1], out[0]
    assert rel(a[0], b[0]) <= 1e-13 and rel(a[1], b[1]) <= 1e-12                     # reduced system
    assert rel(a[2], b[2]) <= 1e-13 and rel(a[3], b[3]) <= 1e-11                     # C^-1, c (sums in another order)
    assert rel(a[4], b[4]) <= 1e-9 and rel(a[5], b[5]) <= 1e-9                       # step (through a CG to 1e-13)
    for ta, tb in zip(a[6], b[6]):
        assert abs(ta[0] - tb[0]) <= 1e-10 * abs(tb[0]) + 1e-18
    assert np.abs(a[7][0] - b[7][0]).max() <= 1e-9 and np.abs(a[7][1] - b[7][1]).max() <= 1e-8
    # ... and the oracle's Schur complement / step (SURVEY section 8d tolerances)
    So, go = orc.reduced_system(lp) if hasattr(orc, 'reduced_system') else (None, None)
    if So is not None:
        assert rel(a[0], So) <= 1e-12 and rel(a[1], go) <= 1e-12
    dxo, _ = orc.gauss_newton_step(lp, points_first=False)
    assert rel(np.concatenate([a[4].ravel(), a[5].ravel()]), dxo) <= 1e-8


def test_long_tracks_keep_the_16_lane_kernels():
    """A landmark with more than 16 observations: no run table is built (the 16-lane kernels' second sweep handles any length);
    the option is a no-op and the step still matches the oracle."""
    from pyslam_amd.device import DeviceProblem
    lp, _ = synthetic.stereo_ba(num_kf=40, num_lm=200, obs_per_lm=24, half_window=14, seed=3)
    dev = DeviceProblem(lp)
    dev.set_option('lm_packed', 1)
    dev.linearize(0.0)
    dev.solve_reduced(1e-13, 4000)
    dev.backsub()
    xp, xl = dev.get_dx()
    dxo, _ = orc.gauss_newton_step(lp, points_first=False)
    assert rel(np.concatenate([xp.ravel(), xl.ravel()]), dxo) <= 1e-8
    dev.close()


def test_packed_kernels_with_the_device_structure_build(monkeypatch):
    """PS_CREATE_DEVICE=2: the run table comes from the device kernels (k_lmw_maxobs, k_lmw_items) instead of the host loop."""
    from pyslam_amd.device import DeviceProblem
    lp = ragged_ba(11, num_kf=40, num_lm=2500, max_obs=12)
    res = {}
    for mode in ('0', '2'):
        monkeypatch.setenv('PS_CREATE_DEVICE', mode)
        dev = DeviceProblem(lp)
        dev.linearize(0.0)
        res[mode] = dev.reduced_dense() + dev.landmark_factors()
        dev.close()
    for x, y in zip(res['0'], res['2']):
        assert np.array_equal(x, y)
