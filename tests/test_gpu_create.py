"""ps_problem_create's structure build on the GPU (csrc/ps_host_build.h) against the host builder (PS_CREATE_DEVICE=0, the
test oracle of the device build): every structure table bit for bit (ps_debug_table_checksums), and the iteration on top of
them.  Counterpart of the bookkeeping the reference redoes in every iteration, pyslam/problem.py:294-329."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TABLES = ['point slots', 'lobs', 'lorig', 'lm_ptr', 'lm_point', 'pose_of_rid', 'pitems', 'pitem_ptr', 'pobs', 'pairs',
          'pair work items', 'combine items', 'combine tasks', 'row_ptr', 'col_idx', 'diag_slot']


def _checksums(dev):
    from pyslam_amd import _native
    lib = _native.load()
    out = (ctypes.c_uint64 * 16)()
    n = ctypes.c_int(0)
    assert lib.ps_debug_table_checksums(dev._h, out, 16, ctypes.byref(n)) == 0, lib.ps_last_error()
    assert n.value == 16
    return list(out)


def _build(lp, mode, **env):
    from pyslam_amd.device import DeviceProblem
    old = {k: os.environ.get(k) for k in list(env) + ['PS_CREATE_DEVICE']}
    os.environ['PS_CREATE_DEVICE'] = str(mode)
    for k, v in env.items():
        os.environ[k] = str(v)
    try:
        return DeviceProblem(lp)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _compare(lp, **env):
    host = _build(lp, 0, **env)
    devb = _build(lp, 2, **env)
    try:
        a, b = _checksums(host), _checksums(devb)
        bad = [TABLES[i] for i in range(16) if a[i] != b[i]]
        assert not bad, bad
        ia, ib = host.info, devb.info
        assert (ia['num_pairs'], ia['reduced_nnzb']) == (ib['num_pairs'], ib['reduced_nnzb'])
        ra, rb = host.gn_iteration(), devb.gn_iteration()
        assert ra == rb, (ra, rb)
        return a
    finally:
        host.close(); devb.close()


@pytest.mark.parametrize('kf,lm,obs,hw,seed', [(6, 64, 4, 3, 1), (40, 3000, 6, 8, 2), (120, 20000, 10, 20, 3), (33, 777, 12, 16, 4)])
def test_device_built_structure_is_bit_identical_to_the_host_builders(kf, lm, obs, hw, seed):
    from pyslam_amd import synthetic
    lp, _ = synthetic.stereo_ba(num_kf=kf, num_lm=lm, obs_per_lm=obs, half_window=hw, seed=seed)
    _compare(lp)


def test_device_build_with_landmark_tiles_and_the_untiled_fallback():
    """Tiles forced small (PS_SCHUR_TILE_MIN_MB): the tiled list, its combine lists, and -- on a problem whose tasks come out
    too short -- the rule that generates the list again without tiles, decided identically by both builders."""
    from pyslam_amd import synthetic
    lp, _ = synthetic.stereo_ba(num_kf=60, num_lm=30000, obs_per_lm=10, half_window=10, seed=5)
    sums = _compare(lp, PS_SCHUR_TILE_MIN_MB=1)
    assert sums[11] != 0 and sums[12] != 0            # tiled: there are combine lists
    lp2, _ = synthetic.stereo_ba(num_kf=200, num_lm=9000, obs_per_lm=10, half_window=20, seed=6)
    assert _compare(lp2, PS_SCHUR_TILE_MIN_MB=1)[11] == 0        # the blocks alone fill the chip: untiled
    lp3, _ = synthetic.stereo_ba(num_kf=100, num_lm=3000, obs_per_lm=10, half_window=20, seed=8)
    assert _compare(lp3, PS_SCHUR_TILE_MIN_MB=1)[11] == 0        # tiled tasks too short: generated again without tiles


def test_device_build_with_constant_poses_constant_points_and_repeated_observations():
    """Rows of constant poses take no part in pairs; observations of constant points have no Z row; a landmark seen twice by
    one pose makes a diagonal-block task."""
    from pyslam_amd import synthetic
    lp, _ = synthetic.stereo_ba(num_kf=30, num_lm=2000, obs_per_lm=8, half_window=6, seed=7)
    rid = lp.pose_rid.copy()
    keep = rid >= 0
    const = np.zeros(rid.size, bool)
    const[[0, 5, 11]] = True
    rid[const] = -1
    rid[~const] = np.arange((~const).sum())
    lp.pose_rid = rid.astype(np.int32)
    vid = lp.point_vid.copy()
    cpt = np.zeros(vid.size, bool)
    cpt[::7] = True
    vid[cpt] = -1
    vid[~cpt] = np.arange((~cpt).sum())
    lp.point_vid = vid.astype(np.int32)
    # a second observation of the same landmark from the same pose, for the first 50 observations
    n = lp.obs_pose.size
    dup = np.arange(50)
    for name in ('obs_pose', 'obs_point', 'obs_grp'):
        setattr(lp, name, np.concatenate([getattr(lp, name), getattr(lp, name)[dup]]))
    lp.obs_uvd = np.concatenate([lp.obs_uvd, lp.obs_uvd[dup] + 0.25])
    assert keep.any() and lp.obs_pose.size == n + 50
    _compare(lp)


def test_device_build_reads_resident_observation_columns_in_place():
    """PS_DESC_DEVICE_TABLES with the device build: the observation columns are read where the caller holds them (no copy to
    the host first); same tables as from host columns, whichever builder."""
    from pyslam_amd import synthetic
    from pyslam_amd.device import resident_tables
    lp, truth = synthetic.stereo_ba(num_kf=25, num_lm=4000, obs_per_lm=7, half_window=6, seed=9)
    lp = synthetic.with_pose_edges(lp, 5, seed=3, truth_poses=truth['poses'])
    host = _build(lp, 0)
    dev_host_cols = _build(lp, 2)
    dev_res_cols = _build(resident_tables(lp, params=True, tables=True), 2)
    host_res_cols = _build(resident_tables(lp, params=False, tables=True), 0)
    try:
        ref = _checksums(host)
        for other in (dev_host_cols, dev_res_cols, host_res_cols):
            got = _checksums(other)
            assert [TABLES[i] for i in range(16) if ref[i] != got[i]] == []
        a = host.gn_iteration()
        assert dev_host_cols.gn_iteration() == a and dev_res_cols.gn_iteration() == a and host_res_cols.gn_iteration() == a
    finally:
        for d in (host, dev_host_cols, dev_res_cols, host_res_cols):
            d.close()


def test_device_build_with_64_bit_pair_keys():
    """Problems beyond tiles x poses^2 = 2^32 sort their pairs by 64-bit (tile | row | column) keys; none that fits a test gets
    there by size, so the switch PS_CREATE_KEYS64 puts an ordinary problem on that path: same tables, tiled and untiled."""
    from pyslam_amd import synthetic
    lp, _ = synthetic.stereo_ba(num_kf=60, num_lm=30000, obs_per_lm=10, half_window=10, seed=5)
    assert _compare(lp, PS_SCHUR_TILE_MIN_MB=1, PS_CREATE_KEYS64=1)[11] != 0        # tiled
    lp2, _ = synthetic.stereo_ba(num_kf=33, num_lm=777, obs_per_lm=12, half_window=16, seed=4)
    assert _compare(lp2, PS_CREATE_KEYS64=1)[11] == 0


@pytest.mark.parametrize('kf,lm', [(200, 50000), (2000, 500000)])
def test_device_build_at_the_baseline_sizes_is_the_host_builders(kf, lm):
    """C3 (untiled, 2.2 M pairs) and C4 (72 landmark tiles, 22.5 M pairs, 29-bit packed keys): every structure table of the default
    (device) build against the host builder's, bit for bit."""
    from pyslam_amd import synthetic
    lp, _ = synthetic.stereo_ba(num_kf=kf, num_lm=lm, obs_per_lm=10, half_window=20, seed=0 if kf == 200 else 1)
    host = _build(lp, 0)
    dev = _build(lp, 1)
    try:
        a, b = _checksums(host), _checksums(dev)
        assert [TABLES[i] for i in range(16) if a[i] != b[i]] == []
        assert (a[11] != 0) == (kf == 2000)
    finally:
        host.close(); dev.close()
