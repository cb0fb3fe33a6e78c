"""Seeded synthetic problems as LoweredProblem tables.

The generators are shared by bench.py, the parity tests and
oracle/gen_golden.py, so the HIP path, the numpy oracle and the verbatim
reference all see byte-identical inputs.  ``to_objects`` rebuilds the
object-graph form (one residual object per block) for any package that
offers the reference's class names -- the reference itself in the authoring
container, or this build's ``pyslam`` shim.
"""
import numpy as np

from pyslam_amd.liegroups import SE2, SE3, SO2, SO3
from pyslam_amd.lowering import (LoweredProblem, pack_pose_matrices,
                                 pose_rows_to_matrices)
from pyslam_amd.utils import invsqrt
from pyslam_amd import losses

STEREO_BA_CAMERA = (640., 480., 1000., 1000., 0.25, 1280, 960)  # reference examples/stereo_ba.py:29


# ---------------------------------------------------------------------------
# batched SE(3) helpers (generation only)
# ---------------------------------------------------------------------------
def _exp_many(xis, group=SE3):
    return np.stack([group.exp(x).as_matrix() for x in np.atleast_2d(xis)])


def _inv_many(Ts):
    n = Ts.shape[1] - 1
    out = np.tile(np.identity(n + 1), (Ts.shape[0], 1, 1))
    Rt = np.transpose(Ts[:, :n, :n], (0, 2, 1))
    out[:, :n, :n] = Rt
    out[:, :n, n] = -np.einsum('nij,nj->ni', Rt, Ts[:, :n, n])
    return out


def _project(cam5, pts_c):
    cu, cv, fu, fv, b = cam5
    iz = 1. / pts_c[:, 2]
    return np.stack([fu * pts_c[:, 0] * iz + cu, fv * pts_c[:, 1] * iz + cv, fu * b * iz], axis=1)


def _triangulate(cam5, uvd):
    cu, cv, fu, fv, b = cam5
    bd = b / uvd[:, 2]
    return np.stack([(uvd[:, 0] - cu) * bd, (uvd[:, 1] - cv) * bd * (fu / fv), fu * bd], axis=1)


def _loss_row(loss):
    return float(loss.LOSS_ID), float(getattr(loss, 'k', 0.))


# ---------------------------------------------------------------------------
# C3 / C4: stereo bundle adjustment
# ---------------------------------------------------------------------------
def stereo_ba(num_kf=200, num_lm=50000, obs_per_lm=10, half_window=20, seed=0,
              loss=None, pose_noise=0.01, point_noise=0.05, const_first_pose=True,
              const_point_fraction=0.0, lm_offset=0, lm_total=None):
    """Stereo BA in the shape of reference examples/stereo_ba.py, scaled up.

    ``lm_offset`` / ``lm_total`` select a contiguous landmark shard of a larger
    problem generated with the same seed (multi-GPU: every rank derives the
    same keyframes and its own landmarks).
    Returns (LoweredProblem, truth dict).
    """
    loss = loss or losses.L2Loss()
    cam5 = np.array(STEREO_BA_CAMERA[:5])
    stiff = invsqrt(np.diagflat([1., 1., 2.]))
    rng = np.random.default_rng(seed)

    k = np.arange(num_kf, dtype=float)
    xi = np.zeros((num_kf, 6))
    xi[:, 0] = 0.05 * k
    xi[:, 5] = 0.002 * k
    T_true = _exp_many(xi)
    T_init = np.einsum('nij,njk->nik', _exp_many(pose_noise * rng.standard_normal((num_kf, 6))), T_true)
    if const_first_pose:
        T_init[0] = T_true[0]

    total = lm_total if lm_total is not None else num_lm
    # per-landmark streams are derived from (seed, landmark id) blocks so a
    # shard reproduces exactly the rows of the full problem
    lrng = np.random.default_rng([seed, 1])
    kc_all = lrng.integers(0, num_kf, size=total)
    pc_all = np.stack([lrng.uniform(-8., 8., total), lrng.uniform(-3., 3., total),
                       lrng.uniform(6., 30., total)], axis=1)
    keys_all = lrng.random((total, 2 * half_window + 1))
    noise_pt_all = point_noise * lrng.standard_normal((total, 3))
    sl = slice(lm_offset, lm_offset + num_lm)
    kc, pc, sel_keys, noise_pt = kc_all[sl], pc_all[sl], keys_all[sl], noise_pt_all[sl]

    T_inv = _inv_many(T_true)
    pts_true = np.einsum('nij,nj->ni', T_inv[kc, :3, :3], pc) + T_inv[kc, :3, 3]
    pts_init = pts_true + noise_pt

    # choose obs_per_lm distinct keyframes in [kc-hw, kc+hw] clipped to [0, K)
    offs = np.arange(-half_window, half_window + 1)
    cand = kc[:, None] + offs[None, :]
    valid = (cand >= 0) & (cand < num_kf)
    sel_keys = np.where(valid, sel_keys, 2.)          # invalid candidates sort last
    n_obs = min(obs_per_lm, int(valid.sum(axis=1).min()))
    pick = np.sort(np.argsort(sel_keys, axis=1)[:, :n_obs], axis=1)
    obs_pose = np.take_along_axis(cand, pick, axis=1).reshape(-1)
    obs_point = np.repeat(np.arange(num_lm), n_obs)

    p_cam = (np.einsum('nij,nj->ni', T_true[obs_pose, :3, :3], pts_true[obs_point])
             + T_true[obs_pose, :3, 3])
    orng = np.random.default_rng([seed, 2, lm_offset])
    uvd = _project(cam5, p_cam) + orng.standard_normal((obs_pose.size, 3)) * np.sqrt([1., 1., 2.])

    rid = np.arange(num_kf, dtype=np.int32)
    if const_first_pose:
        rid = rid - 1          # pose 0 -> -1 (held constant)
    vid = np.arange(num_lm, dtype=np.int32)
    if const_point_fraction > 0.:
        fixed = np.random.default_rng([seed, 3]).random(num_lm) < const_point_fraction
        vid = np.where(fixed, -1, np.cumsum(~fixed) - 1).astype(np.int32)
        pts_init[fixed] = pts_true[fixed]
        # the reference cannot solve with a block whose parameters are ALL constant
        # (problem.py:346-348 leaves e_blocks[ridx] = None and np.bmat raises)
        keep = ~((rid[obs_pose] < 0) & (vid[obs_point] < 0))
        obs_pose, obs_point, uvd = obs_pose[keep], obs_point[keep], uvd[keep]

    lp = LoweredProblem(
        dof=6, poses=pack_pose_matrices(T_init), pose_rid=rid,
        points=pts_init, point_vid=vid,
        obs_pose=obs_pose, obs_point=obs_point, obs_uvd=uvd,
        cams=cam5[None, :], stiff3=stiff.reshape(1, 9),
        obs_groups=np.array([[0., 0., *_loss_row(loss)]]),
        pose_keys=['T_cam{}_w'.format(i) for i in range(num_kf)],
        point_keys=['pt{}_w'.format(j + lm_offset) for j in range(num_lm)]).finalize()
    return lp, {'poses': T_true, 'points': pts_true}


# ---------------------------------------------------------------------------
# C2: SE(3) pose graph;  C1-style: SE(2) pose graph
# ---------------------------------------------------------------------------
def pose_graph(num_poses=10000, num_loops=40001, dof=6, seed=2, loss=None,
               prior_first=True, const_first=False, init_noise=0.02, meas_noise=0.01,
               orientation_loops=False):
    """Noisy helix (SE3) / arc (SE2) with odometry + short-range loop closures.
    orientation_loops (SE3): the loop closures measure the relative ROTATION only
    (reference residuals/pose_to_pose_orientation_residual.py), lowered as lowering.py does."""
    loss = loss if loss is not None else losses.HuberLoss(1.0)
    group = SE3 if dof == 6 else SE2
    rng = np.random.default_rng(seed)
    step = np.array([0.5, 0, 0, 0, 0, 0.05]) if dof == 6 else np.array([0.5, 0, 0.05])
    incs = _exp_many(step + 0.01 * rng.standard_normal((num_poses - 1, dof)), group)
    T_true = [np.identity(dof // 3 + 2)]
    for M in incs:
        T_true.append(M.dot(T_true[-1]))
    T_true = np.stack(T_true)
    T_init = np.einsum('nij,njk->nik',
                       _exp_many(init_noise * rng.standard_normal((num_poses, dof)), group), T_true)

    ei = list(range(num_poses - 1))
    ej = list(range(1, num_poses))
    if num_loops > 0 and num_poses > 3:
        li = rng.integers(0, num_poses - 2, size=num_loops)
        lj = np.minimum(li + rng.integers(2, 31, size=num_loops), num_poses - 1)
        ei += li.tolist()
        ej += lj.tolist()
    ei, ej = np.array(ei), np.array(ej)
    rel = np.einsum('nij,njk->nik', T_true[ej], _inv_many(T_true[ei]))
    meas = np.einsum('nij,njk->nik', _exp_many(meas_noise * rng.standard_normal((ei.size, dof)), group), rel)

    odom = invsqrt(1e-3 * np.identity(dof))
    prior = invsqrt(1e-12 * np.identity(dof))
    rid = np.arange(num_poses, dtype=np.int32)
    if const_first:
        rid = rid - 1
        T_init[0] = T_true[0]
    e_grp = np.zeros(ei.size)
    stiffd = [odom.ravel(), prior.ravel()]
    edge_groups = [[0., *_loss_row(loss)], [1., *_loss_row(losses.L2Loss())]]
    if orientation_loops:
        assert dof == 6
        meas[num_poses - 1:, :3, 3] = 0.           # T_obs = (C_obs, 0)
        S6 = np.zeros((6, 6))
        S6[3:, 3:] = invsqrt(1e-3 * np.identity(3))
        stiffd.append(S6.ravel())
        edge_groups.append([2., *_loss_row(loss)])
        e_grp[num_poses - 1:] = 2
    lp = LoweredProblem(
        dof=dof, poses=pack_pose_matrices(T_init), pose_rid=rid,
        e_i=ei, e_j=ej, e_Tobs_inv=pack_pose_matrices(_inv_many(meas)),
        e_grp=e_grp,
        stiffd=np.stack(stiffd),
        edge_groups=np.array(edge_groups),
        pose_keys=['T_{}_0'.format(i) for i in range(num_poses)])
    if prior_first and not const_first:
        lp.u_i = [0]
        lp.u_Tobs_inv = pack_pose_matrices(_inv_many(T_true[0:1]))
        lp.u_grp = [1]
    return lp.finalize(), {'poses': T_true}


# ---------------------------------------------------------------------------
# C5: sliding-window stereo VO (one pose, N fixed points, robust loss)
# ---------------------------------------------------------------------------
def with_pose_edges(lp, num_loops, seed, loss=None, orientation_loops=False, truth_poses=None):
    """A stereo-BA problem plus the odometry / loop-closure edges and the first-pose prior of a pose graph over the same
    keyframes (mixed visual + relative-pose constraints, as sliding-window VO with odometry would pose them).  Without
    ``truth_poses`` the edge measurements come from ``pose_graph``'s own trajectory and disagree with the visual
    constraints (a valid, if unhappy, problem for single-step parity); with the BA's true poses they are consistent."""
    pg, _ = pose_graph(num_poses=lp.num_poses, num_loops=num_loops, dof=6, seed=seed, loss=loss,
                       orientation_loops=orientation_loops)
    out = lp.copy()
    for name in ('e_i', 'e_j', 'e_Tobs_inv', 'e_grp', 'u_i', 'u_Tobs_inv', 'u_grp', 'stiffd', 'edge_groups'):
        setattr(out, name, getattr(pg, name).copy())
    if truth_poses is not None:
        # measurements of the BA's own true trajectory (+ 1 % noise): visual and relative-pose constraints agree
        T = np.asarray(truth_poses)
        rng = np.random.default_rng([seed, 9])
        rel = np.einsum('nij,njk->nik', T[out.e_j], _inv_many(T[out.e_i]))
        meas = np.einsum('nij,njk->nik', _exp_many(0.01 * rng.standard_normal((out.e_i.size, 6))), rel)
        meas[out.e_grp == 2, :3, 3] = 0.               # rotation-only loop closures: T_obs = (C_obs, 0)
        out.e_Tobs_inv = pack_pose_matrices(_inv_many(meas)) if out.e_i.size else out.e_Tobs_inv
        if out.u_i.size:
            out.u_Tobs_inv = pack_pose_matrices(_inv_many(T[out.u_i]))
    out.validate()
    return out


def motion_only(num_pts=256, seed=3, loss=None, outlier_fraction=0.2):
    loss = loss if loss is not None else losses.CauchyLoss(3.0)
    cam5 = np.array(STEREO_BA_CAMERA[:5])
    stiff = invsqrt(np.diagflat([1., 1., 2.]))
    rng = np.random.default_rng(seed)
    T21 = SE3.exp(np.array([0.3, -0.05, 0.1, 0.01, -0.02, 0.03])).as_matrix()
    p1 = np.stack([rng.uniform(-8, 8, num_pts), rng.uniform(-3, 3, num_pts),
                   rng.uniform(6, 30, num_pts)], axis=1)
    sig = np.sqrt([1., 1., 2.])
    obs1 = _project(cam5, p1) + 0.1 * rng.standard_normal((num_pts, 3)) * sig
    p2 = p1.dot(T21[:3, :3].T) + T21[:3, 3]
    obs2 = _project(cam5, p2) + rng.standard_normal((num_pts, 3)) * sig
    bad = rng.random(num_pts) < outlier_fraction
    obs2[bad, :2] += rng.uniform(20, 60, (int(bad.sum()), 2)) * rng.choice([-1., 1.], (int(bad.sum()), 2))
    lp = LoweredProblem(
        dof=6, poses=pack_pose_matrices(np.identity(4)[None]), pose_rid=[0],
        points=_triangulate(cam5, obs1), point_vid=-np.ones(num_pts),
        obs_pose=np.zeros(num_pts), obs_point=np.arange(num_pts), obs_uvd=obs2,
        cams=cam5[None, :], stiff3=stiff.reshape(1, 9),
        obs_groups=np.array([[0., 0., *_loss_row(loss)]]),
        pose_keys=['T_2_1'], point_keys=[]).finalize()
    return lp, {'poses': T21[None], 'obs_1': obs1, 'obs_2': obs2}


# ---------------------------------------------------------------------------
# LoweredProblem -> object graph (reference-style API)
# ---------------------------------------------------------------------------
_LOSS_NAMES = ['L2Loss', 'L1Loss', 'CauchyLoss', 'HuberLoss', 'TukeyLoss', 'TDistributionLoss']


def make_loss(ns, loss_id, k):
    cls = getattr(ns, _LOSS_NAMES[int(loss_id)])
    return cls() if int(loss_id) < 2 else cls(k)


def pose_objects(rows, dof, ns):
    out = []
    for M in pose_rows_to_matrices(rows, dof):
        if dof == 6:
            out.append(ns.SE3(ns.SO3(M[:3, :3].copy()), M[:3, 3].copy()))
        else:
            out.append(ns.SE2(ns.SO2(M[:2, :2].copy()), M[:2, 2].copy()))
    return out


def to_objects(lp, ns, options=None, points_first=True):
    """Build ``ns.Problem`` from tables.  ``ns`` exposes the reference's public
    names: Problem, Options, StereoCamera, the residual and loss classes and
    the liegroups types.  Parameters are inserted landmarks-first by default,
    like reference examples/stereo_ba.py:53-62."""
    problem = ns.Problem(options if options is not None else ns.Options())
    poses = pose_objects(lp.poses, lp.dof, ns)
    pkeys = lp.pose_keys or ['T{}'.format(i) for i in range(lp.num_poses)]
    lkeys = list(lp.point_keys) + ['_fixed_pt{}'.format(i)
                                   for i in range(lp.num_points - len(lp.point_keys))]

    cams = [ns.StereoCamera(*row, 1280, 960) for row in lp.cams]
    st3 = [row.reshape(3, 3) for row in lp.stiff3]
    std = [row.reshape(lp.dof, lp.dof) for row in lp.stiffd]
    og = [(cams[int(g[0])], st3[int(g[1])], make_loss(ns, g[2], g[3])) for g in lp.obs_groups]
    eg = [(std[int(g[0])], make_loss(ns, g[1], g[2])) for g in lp.edge_groups]

    for i, Tinv, g in zip(lp.u_i, pose_objects(lp.u_Tobs_inv, lp.dof, ns), lp.u_grp):
        problem.add_residual_block(ns.PoseResidual(Tinv.inv(), eg[g][0]), [pkeys[i]], eg[g][1])
    for i, j, Tinv, g in zip(lp.e_i, lp.e_j, pose_objects(lp.e_Tobs_inv, lp.dof, ns), lp.e_grp):
        S = eg[g][0]
        if lp.dof == 6 and not S[:3, :].any() and not S[:, :3].any():
            # rotation-only edge (lowering.py: 3x3 stiffness embedded in the rotational corner)
            block = ns.PoseToPoseOrientationResidual(Tinv.inv().rot, S[3:, 3:].copy())
        else:
            block = ns.PoseToPoseResidual(Tinv.inv(), S)
        problem.add_residual_block(block, [pkeys[i], pkeys[j]], eg[g][1])
    for i, j, uvd, g in zip(lp.obs_pose, lp.obs_point, lp.obs_uvd, lp.obs_grp):
        cam, S, loss = og[g]
        problem.add_residual_block(ns.ReprojectionResidual(cam, uvd.copy(), S),
                                   [pkeys[i], lkeys[j]], loss)

    params = {}
    pt_items = [(k, p.copy()) for k, p in zip(lkeys, lp.points)]
    pose_items = list(zip(pkeys, poses))
    for k, v in (pt_items + pose_items) if points_first else (pose_items + pt_items):
        params[k] = v
    problem.initialize_params(params)
    const = [k for k, r in zip(pkeys, lp.pose_rid) if r < 0] + \
            [k for k, v in zip(lkeys, lp.point_vid) if v < 0]
    if const:
        problem.set_parameters_constant(const)
    return problem


def photometric_scene(h=48, w=64, seed=5, xi_true=(0.03, -0.02, 0.04, 0.01, -0.015, 0.02), rgbd=False,
                      noise=0.0, fu=None):
    """A textured plane seen from two poses, rendered EXACTLY (no warping): the intensity is a smooth function of
    the 3-D point, the plane n.P = c is given in the reference frame, and every tracking-image pixel is the
    intensity at the intersection of its ray with that plane.

    Returns camera parameters ``cam`` = (cu, cv, fu, fv, b, w, h) (b = 0 and depth instead of disparity when
    ``rgbd``), ``im_ref`` (h, w), ``depth_ref`` (disparity or depth, (h, w)), ``im_track`` (h, w), ``im_jac``
    (2, h, w) central-difference gradient of im_ref, and ``T_true`` (4 x 4, track <- ref).
    The inputs of the reference's PhotometricResidualSE3 (photometric_residual.py:44-46)."""
    rng = np.random.default_rng(seed)
    fu = float(fu if fu is not None else 0.9 * w)
    cu, cv, fv, b = 0.5 * w - 0.5, 0.5 * h - 0.5, fu, 0.25
    n = np.array([0.15, -0.1, 1.0]); n /= np.linalg.norm(n)
    c = 4.0
    freq = rng.uniform(0.6, 2.2, size=(6, 3)) * rng.choice([-1., 1.], size=(6, 3))
    phase = rng.uniform(0, 2 * np.pi, 6)
    amp = rng.uniform(10., 30., 6)

    def intensity(P):
        return 100. + np.sum(amp * np.sin(P @ freq.T + phase), axis=-1)

    u, v = np.meshgrid(np.arange(w, dtype=float), np.arange(h, dtype=float), indexing='xy')
    rays = np.stack([(u - cu) / fu, (v - cv) / fv, np.ones_like(u)], axis=-1)
    # reference view: P = s d with n.P = c
    s_ref = c / (rays @ n)
    P_ref = rays * s_ref[..., None]
    im_ref = intensity(P_ref)
    z_ref = P_ref[..., 2]
    depth_ref = z_ref if rgbd else fu * b / z_ref
    # tracking view: P_ref = R^T (s d' - t)
    T = SE3.exp(np.asarray(xi_true, dtype=float))
    R, t = T.rot.as_matrix(), np.asarray(T.trans, dtype=float)
    Rt_n = R @ n                                   # n . R^T x = (R n) . x
    s_trk = (c + Rt_n @ t) / (rays @ Rt_n)
    P_in_ref = (rays * s_trk[..., None] - t) @ R    # rows: R^T (s d' - t)
    im_track = intensity(P_in_ref)
    if noise:
        im_ref = im_ref + noise * rng.standard_normal(im_ref.shape)
        im_track = im_track + noise * rng.standard_normal(im_track.shape)
    gy, gx = np.gradient(im_ref)
    cam = (cu, cv, fu, fv, 0. if rgbd else b, w, h)
    return dict(cam=cam, im_ref=im_ref, depth_ref=depth_ref, im_track=im_track, im_jac=np.stack([gx, gy]),
                T_true=T.as_matrix(), rgbd=rgbd)
