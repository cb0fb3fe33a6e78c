"""Lower a ``Problem`` object graph into flat SoA tables for the HIP core.

The reference keeps one Python object per residual block and per parameter
(reference pyslam/problem.py:43-108) and walks them every iteration
(problem.py:279-360).  Here the walk happens ONCE, at ``solve()`` time: blocks
are dispatched on their ``KIND`` tag into typed tables that are uploaded to
HBM and stay resident for the whole Gauss-Newton loop (SURVEY.md section 7,
step 2).  ``LoweredProblem`` is also what the synthetic generators emit
directly (pyslam_amd/synthetic.py), so the benchmark never builds 500 k Python
objects.

Table layout (all fp64 / int32, C-contiguous):

* ``poses``      (P, 12) SE(3): R row-major (9) | t (3);  (P, 6) SE(2): R (4) | t (2)
* ``pose_rid``   (P,)  index into the reduced (Schur) system, -1 = held constant
* ``points``     (L, 3) landmarks;  ``point_vid`` (L,) variable index, -1 = constant
* ``obs_*``      reprojection observations: pose idx, point idx, (u,v,d), group idx
* ``obs_groups`` (G, 4) [camera idx, stiffness idx, loss id, loss k]
* ``e_*``        binary pose-pose edges: i, j, T_obs^-1 (same packing as poses), group idx
* ``u_*``        unary pose priors:      i,    T_obs^-1,                        group idx
* ``edge_groups``(G, 3) [stiffness idx, loss id, loss k]
"""
from dataclasses import dataclass, field

import numpy as np

from pyslam_amd.liegroups import SE2, SE3
from pyslam_amd import losses as _losses

F64 = np.float64
I32 = np.int32


def _f(a, shape):
    return np.ascontiguousarray(np.asarray(a, dtype=F64).reshape(shape))


def _i(a):
    return np.ascontiguousarray(np.asarray(a, dtype=I32).reshape(-1))


@dataclass
class LoweredProblem:
    dof: int = 6
    poses: np.ndarray = None
    pose_rid: np.ndarray = None
    points: np.ndarray = None
    point_vid: np.ndarray = None
    obs_pose: np.ndarray = None
    obs_point: np.ndarray = None
    obs_uvd: np.ndarray = None
    obs_grp: np.ndarray = None
    cams: np.ndarray = None
    stiff3: np.ndarray = None
    obs_groups: np.ndarray = None
    e_i: np.ndarray = None
    e_j: np.ndarray = None
    e_Tobs_inv: np.ndarray = None
    e_grp: np.ndarray = None
    u_i: np.ndarray = None
    u_Tobs_inv: np.ndarray = None
    u_grp: np.ndarray = None
    stiffd: np.ndarray = None
    edge_groups: np.ndarray = None
    pose_keys: list = field(default_factory=list)
    point_keys: list = field(default_factory=list)

    # ---- derived sizes -------------------------------------------------
    @property
    def pose_width(self):
        return 12 if self.dof == 6 else 6

    @property
    def num_poses(self):
        return 0 if self.poses is None else self.poses.shape[0]

    @property
    def num_points(self):
        return 0 if self.points is None else self.points.shape[0]

    @property
    def num_obs(self):
        return 0 if self.obs_pose is None else self.obs_pose.shape[0]

    @property
    def num_edges(self):
        return 0 if self.e_i is None else self.e_i.shape[0]

    @property
    def num_priors(self):
        return 0 if self.u_i is None else self.u_i.shape[0]

    @property
    def num_reduced(self):
        return int((self.pose_rid >= 0).sum()) if self.num_poses else 0

    @property
    def num_var_points(self):
        return int((self.point_vid >= 0).sum()) if self.num_points else 0

    def finalize(self):
        """Fill absent tables with empty arrays of the right width / dtype."""
        d, pw = self.dof, self.pose_width
        self.poses = _f(self.poses if self.poses is not None else [], (-1, pw))
        self.pose_rid = _i(self.pose_rid if self.pose_rid is not None else [])
        self.points = _f(self.points if self.points is not None else [], (-1, 3))
        self.point_vid = _i(self.point_vid if self.point_vid is not None else [])
        self.obs_pose = _i(self.obs_pose if self.obs_pose is not None else [])
        self.obs_point = _i(self.obs_point if self.obs_point is not None else [])
        self.obs_uvd = _f(self.obs_uvd if self.obs_uvd is not None else [], (-1, 3))
        self.obs_grp = _i(self.obs_grp if self.obs_grp is not None else
                          np.zeros(self.obs_pose.shape[0]))
        self.cams = _f(self.cams if self.cams is not None else [], (-1, 5))
        self.stiff3 = _f(self.stiff3 if self.stiff3 is not None else [], (-1, 9))
        self.obs_groups = _f(self.obs_groups if self.obs_groups is not None else [], (-1, 4))
        self.e_i = _i(self.e_i if self.e_i is not None else [])
        self.e_j = _i(self.e_j if self.e_j is not None else [])
        self.e_Tobs_inv = _f(self.e_Tobs_inv if self.e_Tobs_inv is not None else [], (-1, pw))
        self.e_grp = _i(self.e_grp if self.e_grp is not None else np.zeros(self.e_i.shape[0]))
        self.u_i = _i(self.u_i if self.u_i is not None else [])
        self.u_Tobs_inv = _f(self.u_Tobs_inv if self.u_Tobs_inv is not None else [], (-1, pw))
        self.u_grp = _i(self.u_grp if self.u_grp is not None else np.zeros(self.u_i.shape[0]))
        self.stiffd = _f(self.stiffd if self.stiffd is not None else [], (-1, d * d))
        self.edge_groups = _f(self.edge_groups if self.edge_groups is not None else [], (-1, 3))
        self.validate()
        return self

    def validate(self):
        P, L = self.num_poses, self.num_points
        assert self.dof in (3, 6)
        assert self.pose_rid.shape == (P,) and self.point_vid.shape == (L,)
        for idx, hi in ((self.obs_pose, P), (self.e_i, P), (self.e_j, P), (self.u_i, P),
                        (self.obs_point, L)):
            if idx.size:
                assert idx.min() >= 0 and idx.max() < hi, "index out of range"
        if self.num_obs:
            assert self.dof == 6, "reprojection residuals need SE(3) poses"
            assert self.obs_grp.max() < self.obs_groups.shape[0]
        if self.num_edges:
            assert self.e_grp.max() < self.edge_groups.shape[0]
        if self.num_priors:
            assert self.u_grp.max() < self.edge_groups.shape[0]

    # everything except the parameter VALUES (poses, points): measurements, stiffness / loss / camera
    # groups, connectivity, constant masks, key order
    STRUCTURE_FIELDS = ('pose_rid', 'point_vid', 'obs_pose', 'obs_point', 'obs_uvd', 'obs_grp', 'cams', 'stiff3',
                        'obs_groups', 'e_i', 'e_j', 'e_Tobs_inv', 'e_grp', 'u_i', 'u_Tobs_inv', 'u_grp', 'stiffd',
                        'edge_groups')

    def same_tables(self, other):
        """True when `other` differs from this problem at most in the parameter values: the tables
        resident in HBM (pyslam_amd/device.py) are then still valid and only poses / points need a refresh."""
        if self.dof != other.dof or self.pose_keys != other.pose_keys or self.point_keys != other.point_keys:
            return False
        if self.poses.shape != other.poses.shape or self.points.shape != other.points.shape:
            return False
        return all(np.array_equal(getattr(self, f), getattr(other, f)) for f in self.STRUCTURE_FIELDS)

    def copy(self):
        out = LoweredProblem()
        for k, v in self.__dict__.items():
            setattr(out, k, v.copy() if isinstance(v, np.ndarray) else
                    (list(v) if isinstance(v, list) else v))
        return out


# ---------------------------------------------------------------------------
# pose packing
# ---------------------------------------------------------------------------
def pack_pose(T):
    """liegroups SE2/SE3 -> flat row [R row-major | t]."""
    return np.concatenate([np.asarray(T.rot.mat, dtype=F64).ravel(),
                           np.asarray(T.trans, dtype=F64).ravel()])


def pack_pose_matrix(M):
    """(d+1,d+1) homogeneous matrix -> flat row."""
    n = M.shape[0] - 1
    return np.concatenate([M[:n, :n].ravel(), M[:n, n]])


def pack_pose_matrices(Ms):
    Ms = np.asarray(Ms, dtype=F64)
    n = Ms.shape[1] - 1
    return np.concatenate([Ms[:, :n, :n].reshape(len(Ms), n * n), Ms[:, :n, n]], axis=1)


def unpack_pose(row, dof):
    """flat row -> (R, t) numpy views (copied)."""
    n = 3 if dof == 6 else 2
    return np.array(row[:n * n]).reshape(n, n), np.array(row[n * n:n * n + n])


def pose_rows_to_matrices(rows, dof):
    rows = np.asarray(rows, dtype=F64)
    n = 3 if dof == 6 else 2
    out = np.tile(np.identity(n + 1), (rows.shape[0], 1, 1))
    out[:, :n, :n] = rows[:, :n * n].reshape(-1, n, n)
    out[:, :n, n] = rows[:, n * n:]
    return out


# ---------------------------------------------------------------------------
# object graph -> LoweredProblem
# ---------------------------------------------------------------------------
class NotLowerable(Exception):
    """Raised when a problem contains blocks/parameters with no typed device
    kernel; Problem then takes the host-evaluated generic path."""


def refresh_params(param_dict, lp):
    """Parameter VALUES only, in the order of an existing LoweredProblem (Options.static_blocks: the blocks are declared
    unchanged, so the walk over them is skipped): (poses (P, pose_width), points (L, 3)); rows of `points` beyond the
    keyed landmarks (the constant points of motion-only blocks) are kept."""
    poses = np.stack([pack_pose(param_dict[k]) for k in lp.pose_keys]) if lp.pose_keys else lp.poses.copy()
    points = lp.points.copy()
    if lp.point_keys:
        points[:len(lp.point_keys)] = np.array([param_dict[k] for k in lp.point_keys], dtype=F64).reshape(-1, 3)
    return _f(poses, (-1, lp.pose_width)), _f(points, (-1, 3))


class _Interner:
    """Small table of distinct rows keyed by the identity / bytes of the source."""

    def __init__(self):
        self.rows, self._by_key = [], {}

    def add(self, row, key=None):
        row = np.ascontiguousarray(np.asarray(row, dtype=F64).ravel())
        key = row.tobytes() if key is None else key
        idx = self._by_key.get(key)
        if idx is None:
            idx = len(self.rows)
            self.rows.append(row)
            self._by_key[key] = idx
        return idx

    def table(self, width):
        return (np.stack(self.rows) if self.rows else np.zeros((0, width))).reshape(-1, width)


_DEVICE_LOSSES = (_losses.L2Loss, _losses.L1Loss, _losses.CauchyLoss, _losses.HuberLoss,
                  _losses.TukeyLoss, _losses.TDistributionLoss)

# the observation record carries 8 bits of group: up to 255 (camera, stiffness, loss) rows are device groups as they are;
# beyond that (one stiffness per observation, reference reprojection_residual.py:8-11) the device groups are the
# distinct (camera, loss) CLASSES -- at most 255 of those -- and the stiffness travels as a per-observation index
# (csrc/ps_math.h: ObsWide; ps_problem_create splits the rows)
MAX_OBS_GROUPS = 255


def _loss_id_k(loss):
    """(device loss id, k) of a loss object.  Only the built-in classes THEMSELVES have a device
    restatement (csrc/ps_math.h: ps_loss_rho / ps_loss_weight): a user subclass may override loss() /
    weight(), which the reference would call (problem.py:351-360), so it takes the host-evaluated path."""
    if type(loss) not in _DEVICE_LOSSES:
        raise NotLowerable("loss {} has no device restatement".format(type(loss).__name__))
    return float(loss.LOSS_ID), float(getattr(loss, 'k', 0.))


_FAST = [False, None]        # [looked for, the C walk or None]


def _walk_threads():
    """Threads of the C walk over long runs of reprojection blocks (cext/lower_fast.c: read-only workers under the caller's GIL):
    PYSLAM_AMD_LOWER_THREADS, default min(16, cores); 1 = the serial walk."""
    import os
    try:
        n = int(os.environ.get('PYSLAM_AMD_LOWER_THREADS', '0'))
    except ValueError:
        n = 0
    return n if n > 0 else max(1, min(16, os.cpu_count() or 1))


def _fast_walk():
    """pyslam_amd/cext/lower_fast.c (built by __graft_entry__.build() into pyslam_amd/lib/_lower_fast<EXT_SUFFIX>, or here on
    first use when a C compiler is at hand): the walk over runs of reprojection blocks.  None -- every block through the
    Python loop, same tables -- when it cannot be had (said once on stderr) or PYSLAM_AMD_LOWER_FAST=0."""
    if not _FAST[0]:
        _FAST[0] = True
        import os
        if os.environ.get('PYSLAM_AMD_LOWER_FAST', '1') != '0':
            try:
                _FAST[1] = _load_fast_walk()
            except Exception as e:       # noqa: BLE001 (no compiler, no headers: the Python loop does the same job)
                import sys
                sys.stderr.write('pyslam_amd: the C lowering walk is unavailable ({}: {}); using the Python loop\n'.format(
                    type(e).__name__, e))
                _FAST[1] = None
    return _FAST[1]


def _load_fast_walk(rebuild=False):
    """Build (if the source's hash differs from the one the library was built from) and import the C walk.  The library's name
    carries the interpreter's ABI tag; it is compiled to a temporary file and renamed into place, so that ranks starting
    together (torchrun) never import a half-written file (round-4 ADVICE)."""
    import hashlib
    import importlib.machinery
    import importlib.util
    import os
    import subprocess
    import sysconfig
    import tempfile
    here = os.path.dirname(os.path.abspath(__file__))
    src = os.path.join(here, 'cext', 'lower_fast.c')
    suffix = sysconfig.get_config_var('EXT_SUFFIX') or '.so'
    out = os.path.join(here, 'lib', '_lower_fast' + suffix)
    with open(src, 'rb') as fh:
        sha = hashlib.sha256(fh.read()).hexdigest()[:16]
    stamp = out + '.sha'
    built = None
    if os.path.exists(out) and os.path.exists(stamp):
        with open(stamp) as fh:
            built = fh.read().strip()
    if rebuild or built != sha:
        os.makedirs(os.path.dirname(out), exist_ok=True)
        fd, tmp = tempfile.mkstemp(suffix=suffix, dir=os.path.dirname(out))
        os.close(fd)
        try:
            # (numpy's headers for the direct read of block.obs; without them the buffer protocol does the same job, slower)
            try:
                np_inc = ['-I' + np.get_include()] if os.path.exists(os.path.join(np.get_include(), 'numpy', 'arrayobject.h')) else ['-DPS_LOWER_NO_NUMPY']
            except Exception:       # noqa: BLE001
                np_inc = ['-DPS_LOWER_NO_NUMPY']
            subprocess.run(['gcc', '-O2', '-shared', '-fPIC', '-pthread', '-I' + sysconfig.get_paths()['include']] + np_inc + [src, '-o', tmp], check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            os.replace(tmp, out)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
        fd, tmp = tempfile.mkstemp(dir=os.path.dirname(out))
        with os.fdopen(fd, 'w') as fh:
            fh.write(sha)
        os.replace(tmp, stamp)
    spec = importlib.util.spec_from_file_location('_lower_fast', out, loader=importlib.machinery.ExtensionFileLoader('_lower_fast', out))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def lower(param_dict, residual_blocks, block_param_keys, block_loss_functions,
          constant_param_keys):
    """Build a LoweredProblem from the registries of a Problem."""
    const = set(constant_param_keys)
    pose_keys, point_keys, pose_ix, point_ix = [], [], {}, {}
    dof = None
    fast = _fast_walk()
    if fast is not None and type(param_dict) is dict:
        # the same split in C (pyslam_amd/cext/lower_fast.c: classify): the keys in the dictionary's order
        pose_keys, point_keys, bad, pose_ix, point_ix = fast.classify(param_dict, (SE3, SE2), np.ndarray)
        if bad is not None:
            raise NotLowerable("parameter {!r} is neither an SE2/SE3 pose nor a 3-vector".format(bad))
        for key in pose_keys:
            d = param_dict[key].dof
            if dof is None:
                dof = d
            elif dof != d:
                raise NotLowerable("mixed SE(2)/SE(3) poses")
    else:
        for key, val in param_dict.items():
            if isinstance(val, (SE3, SE2)):
                d = val.dof
                if dof is None:
                    dof = d
                elif dof != d:
                    raise NotLowerable("mixed SE(2)/SE(3) poses")
                pose_ix[key] = len(pose_keys)
                pose_keys.append(key)
            elif isinstance(val, np.ndarray) and val.shape == (3,):
                point_ix[key] = len(point_keys)
                point_keys.append(key)
            else:
                raise NotLowerable("parameter {!r} is neither an SE2/SE3 pose nor a 3-vector".format(key))
    if dof is None:
        dof = 6

    lp = LoweredProblem(dof=dof, pose_keys=pose_keys, point_keys=point_keys)
    pw = lp.pose_width
    poses = np.zeros((len(pose_keys), pw))
    for k, i in pose_ix.items():
        poses[i] = pack_pose(param_dict[k])
    points = np.empty((len(point_keys), 3), dtype=F64)
    rest = fast.gather3(param_dict, point_keys, points) if fast is not None and type(param_dict) is dict else range(len(point_keys))
    for i in rest:
        points[i] = np.asarray(param_dict[point_keys[i]], dtype=F64)
    fixed_points, n_fixed = [], 0   # motion-only points appended after the landmarks ((m, 3) chunks)

    rid, n = np.full(len(pose_keys), -1, dtype=I32), 0
    for k in pose_keys:
        if k not in const:
            rid[pose_ix[k]] = n
            n += 1
    # (variable landmarks numbered in key order; constants -1 -- vectorised: a loop over 50 000 keys was 3 ms of a C3 lowering)
    is_const = np.zeros(len(point_keys), dtype=bool)
    if const:
        for k in const:
            j = point_ix.get(k)
            if j is not None:
                is_const[j] = True
    vid = np.where(is_const, -1, np.cumsum(~is_const) - 1).astype(I32)

    cams, st3, std = _Interner(), _Interner(), _Interner()
    ogrp, egrp = _Interner(), _Interner()
    # single-observation blocks, one entry each: columns of capacity len(residual_blocks), `cnt` filled
    n_blocks = min(len(residual_blocks), len(block_param_keys), len(block_loss_functions))
    o_pose, o_pt, o_g = np.empty(n_blocks, dtype=I32), np.empty(n_blocks, dtype=I32), np.empty(n_blocks, dtype=I32)
    o_uvd = np.empty((n_blocks, 3), dtype=F64)
    cnt = 0
    c_pose, c_pt, c_uvd, c_g = [], [], [], []      # batch blocks, one array each (appended after the single ones)
    e_i, e_j, e_T, e_g = [], [], [], []
    u_i, u_T, u_g = [], [], []
    L0 = len(point_keys)
    ogrp_cache, egrp_cache = {}, {}   # keyed by object identity: O(1) per block

    def obs_group(cam, block, loss):
        gkey = (id(cam), id(block.stiffness), id(loss))
        g = ogrp_cache.get(gkey)
        if g is None:
            if np.size(block.stiffness) != 9:
                raise NotLowerable("reprojection stiffness must be 3x3")
            lid, lk = _loss_id_k(loss)
            g = ogrp.add([cams.add(cam.intrinsics()), st3.add(block.stiffness), lid, lk])
            ogrp_cache[gkey] = g
        return g

    walk = fast.walk if fast is not None and dof == 6 and isinstance(residual_blocks, list) else None
    i = -1
    while i + 1 < n_blocks:
        i += 1
        block, keys, loss = residual_blocks[i], block_param_keys[i], block_loss_functions[i]
        kind = getattr(block, 'KIND', 'generic')
        if kind == 'reproj' and walk is not None:
            # a run of consecutive reprojection blocks in C (pyslam_amd/cext/lower_fast.c); it stops at the first block it
            # does not take, which then goes through the general path below
            nxt, cnt = walk(residual_blocks, block_param_keys, block_loss_functions, i, param_dict, pose_ix, point_ix, obs_group,
                            o_pose, o_pt, o_uvd, o_g, cnt, _walk_threads())
            if nxt > i:
                i = nxt - 1
                continue
        for k in keys:
            if k not in param_dict:
                raise KeyError(k)
        if kind in ('reproj', 'reproj_motion_only', 'reproj_motion_only_batch'):
            cam = block.camera
            if getattr(cam, 'CAMERA_ID', None) not in (0, 1):
                raise NotLowerable("camera {} has no device restatement".format(type(cam).__name__))
            if dof != 6 or keys[0] not in pose_ix:
                raise NotLowerable("reprojection block needs an SE(3) pose first")
            g = obs_group(cam, block, loss)
            if kind == 'reproj':
                if keys[1] not in point_ix:
                    raise NotLowerable("reprojection block needs a 3-vector landmark second")
                o_pose[cnt], o_pt[cnt], o_g[cnt] = pose_ix[keys[0]], point_ix[keys[1]], g
                o_uvd[cnt] = np.asarray(block.obs, dtype=F64).reshape(3)
                cnt += 1
            else:
                # whole block at once (the per-frame Problem of pipelines/sparse.py:153-161 is ONE batch block of
                # ~10^3 points: a Python loop over them was 0.9 ms of a 2 ms solve)
                pts = np.atleast_2d(np.asarray(block.pt_1 if kind == 'reproj_motion_only' else block.pts_1, dtype=F64))
                obs2 = np.atleast_2d(np.asarray(block.obs_2, dtype=F64))
                m = min(len(pts), len(obs2))
                # (whole columns as arrays: Python lists of 2 048 ints cost 0.1 ms per frame to convert)
                c_pose.append(np.full(m, pose_ix[keys[0]], dtype=I32))
                c_pt.append(np.arange(L0 + n_fixed, L0 + n_fixed + m, dtype=I32))
                fixed_points.append(pts[:m])
                n_fixed += m
                c_uvd.append(obs2[:m].reshape(-1, 3))
                c_g.append(np.full(m, g, dtype=I32))
        elif kind in ('pose_pose', 'pose_prior'):
            if any(k not in pose_ix for k in keys):
                raise NotLowerable("pose block on a non-pose parameter")
            if block.obstype.dof != dof:
                raise NotLowerable("observation group differs from the pose group")
            obs = block.T_2_1_obs if kind == 'pose_pose' else block.T_obs
            gkey = (id(block.stiffness), id(loss))
            g = egrp_cache.get(gkey)
            if g is None:
                if np.size(block.stiffness) != dof * dof:
                    raise NotLowerable("pose stiffness must be dof x dof")
                lid, lk = _loss_id_k(loss)
                g = egrp.add([std.add(block.stiffness), lid, lk])
                egrp_cache[gkey] = g
            if kind == 'pose_pose':
                e_i.append(pose_ix[keys[0]])
                e_j.append(pose_ix[keys[1]])
                e_T.append(pack_pose(obs.inv()))
                e_g.append(g)
            else:
                u_i.append(pose_ix[keys[0]])
                u_T.append(pack_pose(obs.inv()))
                u_g.append(g)
        elif kind == 'pose_pose_orientation':
            # Rotation-only relative measurement (reference pose_to_pose_orientation_residual.py:4-38)
            # as a pose-pose edge: T_obs = (C_obs, 0) and the 3x3 stiffness embedded in the rotational
            # corner of a 6x6 one.  Then S6 log(T_2 T_1^-1 T_obs^-1) = [0; S3 log(C_2 C_1^T C_obs^T)],
            # -S6 Ad(T_2 T_1^-1) = [0 0; 0 -S3 C_21] and S6 = [0 0; 0 S3]: exactly the reference's
            # residual and 3x6 Jacobians, padded with three absent (all-zero) rows the kernels skip.
            if dof != 6 or any(k not in pose_ix for k in keys):
                raise NotLowerable("orientation block needs two SE(3) poses")
            if getattr(block.obstype, 'dof', None) != 3 or np.size(block.stiffness) != 9:
                raise NotLowerable("orientation block needs an SO(3) observation and a 3x3 stiffness")
            gkey = (id(block.stiffness), id(loss), 'orientation')
            g = egrp_cache.get(gkey)
            if g is None:
                S6 = np.zeros((6, 6))
                S6[3:, 3:] = np.asarray(block.stiffness, dtype=F64).reshape(3, 3)
                lid, lk = _loss_id_k(loss)
                g = egrp.add([std.add(S6), lid, lk])
                egrp_cache[gkey] = g
            Tinv = np.zeros(12)
            Tinv[:9] = np.asarray(block.C_2_1_obs.inv().as_matrix(), dtype=F64).reshape(-1)
            e_i.append(pose_ix[keys[0]])
            e_j.append(pose_ix[keys[1]])
            e_T.append(Tinv)
            e_g.append(g)
        else:
            raise NotLowerable("block {} has no typed device kernel".format(type(block).__name__))
    if len(ogrp.rows) > MAX_OBS_GROUPS:
        classes = {(r[0], r[2], r[3]) for r in ogrp.rows}
        if len(classes) > MAX_OBS_GROUPS:
            raise NotLowerable("{} distinct (camera, loss) classes among the observation groups; the typed device "
                               "tables hold at most {}".format(len(classes), MAX_OBS_GROUPS))

    lp.poses, lp.pose_rid = poses, rid
    lp.points = np.concatenate([points] + fixed_points) if fixed_points else points
    lp.point_vid = np.concatenate([vid, np.full(n_fixed, -1, dtype=I32)]) if n_fixed else vid
    o_pose, o_pt, o_g, o_uvd = o_pose[:cnt], o_pt[:cnt], o_g[:cnt], o_uvd[:cnt]
    if c_pose:
        one = lambda col: [col] if cnt else []
        lp.obs_pose, lp.obs_point = np.concatenate(one(o_pose) + c_pose), np.concatenate(one(o_pt) + c_pt)
        lp.obs_grp = np.concatenate(one(o_g) + c_g)
        lp.obs_uvd = np.concatenate(one(o_uvd) + c_uvd)
    else:
        lp.obs_pose, lp.obs_point, lp.obs_grp, lp.obs_uvd = o_pose, o_pt, o_g, o_uvd
    lp.cams, lp.stiff3, lp.obs_groups = cams.table(5), st3.table(9), ogrp.table(4)
    lp.e_i, lp.e_j, lp.e_grp = e_i, e_j, e_g
    lp.e_Tobs_inv = np.array(e_T).reshape(-1, pw)
    lp.u_i, lp.u_grp = u_i, u_g
    lp.u_Tobs_inv = np.array(u_T).reshape(-1, pw)
    lp.stiffd, lp.edge_groups = std.table(dof * dof), egrp.table(3)
    return lp.finalize()
