"""The C walk over runs of reprojection blocks (pyslam_amd/cext/lower_fast.c) against the Python loop of
pyslam_amd/lowering.py: the same LoweredProblem, table by table, on problems that mix block kinds, key containers and
observation containers, and the same exceptions.  Host logic; no GPU.  (What the walk replaces: the reference's per-iteration
walk over its block objects, pyslam/problem.py:279-360, done once per solve here.)"""
import types

import numpy as np
import pytest

import liegroups as G
import pyslam.losses as Ls
import pyslam.problem as P
import pyslam.residuals as R
import pyslam.sensors as S
from pyslam_amd import lowering, synthetic

NS = types.SimpleNamespace(Problem=P.Problem, Options=P.Options, StereoCamera=S.StereoCamera, PoseResidual=R.PoseResidual,
                           PoseToPoseResidual=R.PoseToPoseResidual, PoseToPoseOrientationResidual=R.PoseToPoseOrientationResidual,
                           ReprojectionResidual=R.ReprojectionResidual, L2Loss=Ls.L2Loss, L1Loss=Ls.L1Loss, CauchyLoss=Ls.CauchyLoss,
                           HuberLoss=Ls.HuberLoss, TukeyLoss=Ls.TukeyLoss, TDistributionLoss=Ls.TDistributionLoss,
                           SE3=G.SE3, SO3=G.SO3, SE2=G.SE2, SO2=G.SO2)
TABLES = ['poses', 'pose_rid', 'points', 'point_vid', 'obs_pose', 'obs_point', 'obs_uvd', 'obs_grp', 'cams', 'stiff3', 'obs_groups',
          'e_i', 'e_j', 'e_Tobs_inv', 'e_grp', 'u_i', 'u_Tobs_inv', 'u_grp', 'stiffd', 'edge_groups']


def _walk_available():
    lowering._FAST[:] = [False, None]
    return lowering._fast_walk() is not None


def _both(problem):
    """(tables with the C walk, tables with the Python loop)"""
    assert _walk_available(), 'pyslam_amd/lib/_lower_fast<EXT_SUFFIX> could not be built or loaded'
    fast = problem._lower()
    lowering._FAST[:] = [True, None]
    try:
        slow = problem._lower()
    finally:
        lowering._FAST[:] = [False, None]
    return fast, slow


def _same(a, b):
    for name in TABLES:
        x, y = getattr(a, name), getattr(b, name)
        assert np.asarray(x).dtype == np.asarray(y).dtype and np.array_equal(np.asarray(x), np.asarray(y)), name
    assert a.pose_keys == b.pose_keys and a.point_keys == b.point_keys


def test_bundle_adjustment_tables_are_the_python_loops():
    lp, _ = synthetic.stereo_ba(num_kf=12, num_lm=700, obs_per_lm=5, half_window=4, seed=3)
    fast, slow = _both(synthetic.to_objects(lp, NS))
    _same(fast, slow)
    assert fast.num_obs == lp.num_obs and np.array_equal(fast.obs_uvd, lp.obs_uvd)


def test_mixed_kinds_key_containers_and_observation_containers():
    """Pose-graph blocks between the reprojection blocks (the run ends and starts again), keys as tuples, observations as
    lists / float32 / non-contiguous views (handed back to the Python loop block by block), two cameras, three losses."""
    lp, _ = synthetic.stereo_ba(num_kf=8, num_lm=120, obs_per_lm=4, half_window=3, seed=4)
    problem = synthetic.to_objects(lp, NS)
    cam2 = S.StereoCamera(300., 200., 500., 510., 0.3, 1280, 960)
    losses = [Ls.HuberLoss(1.5), Ls.CauchyLoss(2.0), problem.block_loss_functions[0]]
    wide = np.zeros((7, 6))
    for i, (block, keys) in enumerate(zip(problem.residual_blocks, problem.block_param_keys)):
        if getattr(block, 'KIND', '') != 'reproj':
            continue
        if i % 5 == 0:
            problem.block_param_keys[i] = tuple(keys)
        if i % 7 == 0:
            block.obs = list(block.obs)
        elif i % 7 == 1:
            block.obs = block.obs.astype(np.float32)
        elif i % 7 == 2:
            wide[i % 7, ::2] = block.obs
            block.obs = wide[i % 7, ::2].copy()[::1]
        elif i % 7 == 3:
            block.obs = np.stack([block.obs, block.obs])[:, :][0:2][1]        # a row view: still contiguous
        elif i % 7 == 4:
            block.obs = np.asfortranarray(np.stack([block.obs, 2 * block.obs]).T)[:, 0]    # strided view
        if i % 11 == 0:
            block.camera = cam2
        problem.block_loss_functions[i] = losses[i % 3]
    # pose-graph blocks in the middle of the list
    T = G.SE3.exp(0.01 * np.arange(6))
    mid = len(problem.residual_blocks) // 2
    problem.residual_blocks.insert(mid, R.PoseToPoseResidual(T, np.eye(6)))
    problem.block_param_keys.insert(mid, ['T_cam1_w', 'T_cam2_w'])
    problem.block_loss_functions.insert(mid, Ls.L2Loss())
    problem.residual_blocks.insert(3, R.PoseResidual(T, 2 * np.eye(6)))
    problem.block_param_keys.insert(3, ['T_cam3_w'])
    problem.block_loss_functions.insert(3, Ls.L2Loss())
    fast, slow = _both(problem)
    _same(fast, slow)
    assert fast.num_edges == 1 and fast.num_priors == 1 and len(fast.cams) == 2 and len(fast.obs_groups) >= 6


def test_motion_only_batch_blocks_between_single_observations():
    lp, _ = synthetic.stereo_ba(num_kf=5, num_lm=60, obs_per_lm=3, half_window=2, seed=5)
    problem = synthetic.to_objects(lp, NS)
    cam = problem.residual_blocks[-1].camera
    obs1 = np.array([[640., 480., 12.], [600., 470., 9.], [500., 400., 20.]])
    batch = R.ReprojectionMotionOnlyBatchResidual(cam, obs1, obs1 + 0.5, np.eye(3))
    problem.residual_blocks.insert(10, batch)
    problem.block_param_keys.insert(10, ['T_cam1_w'])
    problem.block_loss_functions.insert(10, Ls.L2Loss())
    fast, slow = _both(problem)
    _same(fast, slow)
    assert fast.num_points == lp.num_points + 3


@pytest.mark.parametrize('what', ['unknown_key', 'camera', 'landmark_is_a_pose', 'stiffness'])
def test_errors_are_the_python_loops(what):
    lp, _ = synthetic.stereo_ba(num_kf=5, num_lm=40, obs_per_lm=3, half_window=2, seed=6)
    problem = synthetic.to_objects(lp, NS)
    k = 17
    if what == 'unknown_key':
        problem.block_param_keys[k] = [problem.block_param_keys[k][0], 'no such landmark']
        exc = KeyError
    elif what == 'camera':
        class OtherCamera:
            pass
        problem.residual_blocks[k].camera = OtherCamera()
        exc = lowering.NotLowerable
    elif what == 'landmark_is_a_pose':
        problem.block_param_keys[k] = ['T_cam1_w', 'T_cam2_w']
        exc = lowering.NotLowerable
    else:
        problem.residual_blocks[k].stiffness = np.eye(2)
        exc = lowering.NotLowerable
    assert _walk_available()
    with pytest.raises(exc) as e_fast:
        problem._lower()
    lowering._FAST[:] = [True, None]
    try:
        with pytest.raises(exc) as e_slow:
            problem._lower()
    finally:
        lowering._FAST[:] = [False, None]
    assert str(e_fast.value) == str(e_slow.value)


def test_parameter_split_gather_and_write_back_with_unusual_landmark_arrays():
    """Landmarks that are float32 arrays, strided views or read-only arrays go through the Python statements one by one (the C
    helpers hand their indices back); a parameter that is neither a pose nor a 3-vector raises the same error either way."""
    lp, _ = synthetic.stereo_ba(num_kf=5, num_lm=50, obs_per_lm=3, half_window=2, seed=7)
    problem = synthetic.to_objects(lp, NS)
    keys = list(lp.point_keys)
    big = np.zeros((4, 6))
    big[1, ::2] = problem.param_dict[keys[3]]
    problem.param_dict[keys[3]] = big[1, ::2]                        # strided view (writable)
    problem.param_dict[keys[5]] = problem.param_dict[keys[5]].astype(np.float32)
    ro = problem.param_dict[keys[8]].copy(); ro.setflags(write=False)
    problem.param_dict[keys[8]] = ro
    fast, slow = _both(problem)
    _same(fast, slow)
    assert np.array_equal(fast.points[3], big[1, ::2]) and fast.points.dtype == np.float64

    class Dev:                                                       # what _write_back needs of a DeviceProblem
        def __init__(self, lp):
            self.lp = lp
        def get_params(self):
            return self.lp.poses.copy(), self.lp.points + 1.0
    problem.param_dict[keys[8]] = ro.copy()                          # (a read-only landmark cannot be written back by either path)
    for use_c in (True, False):
        lowering._FAST[:] = [False, None] if use_c else [True, None]
        try:
            lpw = problem._lower()
            problem._write_back(Dev(lpw))
            got = np.array([np.asarray(problem.param_dict[k], dtype=np.float64) for k in lpw.point_keys])
            assert np.allclose(got, lpw.points + 1.0, rtol=0, atol=1e-6), use_c     # (float32 landmark: rounded)
            assert np.array_equal(big[1, ::2], lpw.points[lpw.point_keys.index(keys[3])] + 1.0)
            for k in lpw.point_keys:                                  # back, for the second pass
                v = problem.param_dict[k]
                v[...] = np.asarray(v, dtype=np.float64) - 1.0
        finally:
            lowering._FAST[:] = [False, None]
    problem.param_dict['odd'] = [1.0, 2.0, 3.0]
    for use_c in (True, False):
        lowering._FAST[:] = [False, None] if use_c else [True, None]
        try:
            with pytest.raises(lowering.NotLowerable, match="'odd'"):
                problem._lower()
        finally:
            lowering._FAST[:] = [False, None]


def test_the_walk_on_several_threads_gives_the_serial_walks_tables(monkeypatch):
    """Round 6: long runs of reprojection blocks are walked by read-only worker threads under the caller's GIL
    (cext/lower_fast.c: walk_chunk; PYSLAM_AMD_LOWER_THREADS).  Same tables as the serial walk and as the Python loop -- also when
    irregular blocks sit inside the workers' chunks: an observation that is a list, a float32 one, a key that is not an exact str
    (handed back to the Python loop one by one, the run starts again behind each), a second loss and a block of another kind."""
    assert _walk_available()
    lp, _ = synthetic.stereo_ba(num_kf=40, num_lm=18000, obs_per_lm=5, half_window=8, seed=6)      # 90 000 blocks: beyond the 65 536 a run needs
    problem = synthetic.to_objects(lp, NS)
    n = len(problem.residual_blocks)
    assert n >= 70000

    class KeyStr(str):
        pass
    blocks, keys, losses = problem.residual_blocks, problem.block_param_keys, problem.block_loss_functions
    blocks[30000].obs = [float(v) for v in blocks[30000].obs]                     # a list
    blocks[61234].obs = np.asarray(blocks[61234].obs, dtype=np.float32)           # not float64
    keys[45000] = [KeyStr(keys[45000][0]), keys[45000][1]]                        # equal to the key, not an exact str
    losses[52000] = Ls.HuberLoss(2.0)                                             # a group the warm-up has not seen
    losses[52001] = losses[52000]
    out = {}
    for threads in ('1', '3', '8'):
        monkeypatch.setenv('PYSLAM_AMD_LOWER_THREADS', threads)
        out[threads] = problem._lower()
    lowering._FAST[:] = [True, None]
    try:
        slow = problem._lower()
    finally:
        lowering._FAST[:] = [False, None]
    for threads in out:
        _same(out[threads], slow)
    assert out['8'].num_obs == n and len(out['8'].obs_groups) == 2
