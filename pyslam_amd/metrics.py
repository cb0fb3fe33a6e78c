"""Trajectory error metrics and their ``.mat`` exchange format.

Mirror of the reference's ``pyslam.metrics.TrajectoryMetrics`` (pyslam/metrics.py:7-300): same
constructor, attributes, method names, argument meaning, units and ``savemat`` keys, so result
files written by either side load on the other.  The reference walks Python lists of liegroups
objects pose by pose; here the poses are stacked once into (N, d, d) arrays and every metric is a
batched array expression (a 10 000-pose trajectory -- config C2 -- takes milliseconds, not seconds).

Behaviour kept on purpose (the reference's tests and result files depend on it):
  * ``traj_errors`` expresses BOTH trajectories relative to the ground-truth first pose of the
    segment (metrics.py:206-209), ``rel_errors`` is ``gt^-1 est`` while the other errors are
    ``est^-1 gt`` (metrics.py:233 vs 154, 211);
  * ``mean_err`` / ``cum_err`` ignore ``delta`` (metrics.py:258-281);
  * ``segment_errors`` uses ``searchsorted(..., side='right')`` and drops segments that run off
    the end (metrics.py:176-187);
  * trajectories of unequal length are truncated with a printed warning (metrics.py:25-31).
Difference: SE(2) trajectories work (the reference's ``_compute_distances`` writes 2-vectors into
3-wide rows and raises, metrics.py:50-52).
"""
import numpy as np
import scipy.io

from .liegroups import SE2, SE3

_METERS = {'m': 1., 'dm': 10., 'cm': 100., 'mm': 1000.}
_RADIANS = {'rad': 1, 'deg': 180. / np.pi}


def _stack(poses):
    """list of SE2/SE3 -> rotations (N, d, d), translations (N, d)"""
    R = np.array([np.asarray(T.rot.as_matrix(), dtype=float) for T in poses])
    t = np.array([np.asarray(T.trans, dtype=float).ravel() for T in poses])
    return R, t


def _inv(R, t):
    Rt = np.swapaxes(R, -1, -2)
    return Rt, -np.einsum('...ij,...j->...i', Rt, t)


def _mul(Ra, ta, Rb, tb):
    return Ra @ Rb, np.einsum('...ij,...j->...i', Ra, tb) + ta


def _rot_log(R):
    """Batched SO(2)/SO(3) logarithm with the liegroups small-angle switch (``np.isclose(angle, 0)``)."""
    if R.shape[-1] == 2:
        return np.arctan2(R[..., 1, 0], R[..., 0, 0])[..., None]
    tr = np.trace(R, axis1=-2, axis2=-1)
    angle = np.arccos(np.clip(0.5 * tr - 0.5, -1., 1.))
    small = np.isclose(angle, 0.)
    A = np.where(small[..., None, None], R - np.eye(3), R - np.swapaxes(R, -1, -2))
    with np.errstate(divide='ignore', invalid='ignore'):
        scale = np.where(small, 1., 0.5 * angle / np.sin(angle))
    return scale[..., None] * np.stack([A[..., 2, 1], A[..., 0, 2], A[..., 1, 0]], axis=-1)


class TrajectoryMetrics:
    """Metrics on SE2/SE3 trajectories (reference: pyslam/metrics.py:7).

        convention='Twv' -- poses are vehicle-to-world transforms
        convention='Tvw' -- poses are world-to-vehicle transforms (converted to Twv internally)
    """

    def __init__(self, poses_gt, poses_est, convention='Twv'):
        if not isinstance(convention, str):                 # (loadmat hands strings back as arrays)
            convention = str(np.asarray(convention).ravel()[0])
        to_world = {'Twv': lambda T: T, 'Tvw': lambda T: T.inv()}.get(convention)
        if to_world is None:
            raise ValueError('convention must be \'Tvw\' or \'Twv\'')
        n_gt, n_est = len(poses_gt), len(poses_est)
        n = min(n_gt, n_est)
        if n_gt != n_est:                                   # the reference truncates to the common prefix and says so
            print('WARNING: poses_gt has length {} but poses_est has length {}. Truncating to {}.'.format(n_gt, n_est, n))

        self.convention = convention
        # vehicle-to-world poses, as the reference's public attributes (lists of liegroups objects) ...
        self.Twv_gt = [to_world(T) for T in poses_gt[:n]]
        self.Twv_est = [to_world(T) for T in poses_est[:n]]
        self.pose_type = type(self.Twv_gt[0])
        self.num_poses = n
        # ... and stacked once for the batched metrics
        self._Rg, self._tg = _stack(self.Twv_gt)
        self._Re, self._te = _stack(self.Twv_est)
        self.rel_dists, self.cum_dists = self._compute_distances()

    # ---- helpers -------------------------------------------------------------------------------
    def _compute_distances(self):
        """Relative and cumulative distance travelled at each ground-truth pose (metrics.py:47-57)."""
        rel = np.linalg.norm(np.diff(self._tg, axis=0), axis=1)
        rel = np.append([0.], rel)
        return rel, np.cumsum(rel)

    def _convert_meters(self, meters, unit):
        return _METERS[unit] * meters

    def _convert_radians(self, radians, unit):
        return _RADIANS[unit] * radians

    @staticmethod
    def _indices(segment_range, n):
        return np.arange(n) if segment_range is None else np.asarray(list(segment_range), dtype=int)

    # ---- .mat exchange format ------------------------------------------------------------------
    def savemat(self, filename, extras=None):
        """Same keys and layout as the reference (metrics.py:80-109): poses as d x d x N in the
        caller's original convention, 'convention', 'pose_type', 'num_poses', 'rel_dists', 'cum_dists'."""
        def mats(R, t):
            if self.convention == 'Tvw':
                R, t = _inv(R, t)
            n, d = t.shape
            M = np.zeros((n, d + 1, d + 1))
            M[:, :d, :d] = R
            M[:, :d, d] = t
            M[:, d, d] = 1.
            return np.transpose(M, [1, 2, 0])
        mdict = {'poses_gt': mats(self._Rg, self._tg),
                 'poses_est': mats(self._Re, self._te),
                 'convention': self.convention,
                 'pose_type': self.pose_type.__name__,
                 'num_poses': self.num_poses,
                 'rel_dists': self.rel_dists,
                 'cum_dists': self.cum_dists}
        if extras is not None:
            mdict.update(extras)
        scipy.io.savemat(filename, mdict, do_compression=True)

    @classmethod
    def loadmat(cls, filename):
        """Load a file written by ``savemat`` (of this class or of the reference, metrics.py:111-145)."""
        mdict = scipy.io.loadmat(filename, verify_compressed_data_integrity=True)
        name = str(np.asarray(mdict['pose_type']).ravel()[0])
        if name == 'SE2':
            pose_type = SE2
        elif name == 'SE3':
            pose_type = SE3
        else:
            raise ValueError('Got invalid pose type: {}'.format(mdict['pose_type']))
        num_poses = int(np.asarray(mdict['num_poses']).ravel()[0])
        poses_gt = [pose_type.from_matrix(mdict['poses_gt'][:, :, i], normalize=True) for i in range(num_poses)]
        poses_est = [pose_type.from_matrix(mdict['poses_est'][:, :, i], normalize=True) for i in range(num_poses)]
        tm = cls(poses_gt, poses_est, convention=str(np.asarray(mdict['convention']).ravel()[0]))
        tm.mdict = mdict
        return tm

    # ---- errors --------------------------------------------------------------------------------
    def endpoint_error(self, segment_range=None, trans_unit='m', rot_unit='rad'):
        """Translational and rotational error at the end of a segment (metrics.py:147-161)."""
        idx = self._indices(segment_range, self.num_poses)
        trans, rot = self._endpoint_errors(idx[:1], idx[-1:])
        return self._convert_meters(trans[0], trans_unit), self._convert_radians(rot[0], rot_unit)

    def _endpoint_errors(self, start, stop):
        """Batched endpoint errors of the segments start[k] .. stop[k]."""
        Rg, tg = _mul(*_inv(self._Rg[start], self._tg[start]), self._Rg[stop], self._tg[stop])
        Re, te = _mul(*_inv(self._Re[start], self._te[start]), self._Re[stop], self._te[stop])
        Rerr, terr = _mul(*_inv(Re, te), Rg, tg)
        return np.linalg.norm(terr, axis=1), np.linalg.norm(_rot_log(Rerr), axis=1)

    def segment_errors(self, segment_lengths, trans_unit='m', rot_unit='rad'):
        """Endpoint errors of all segments of the given lengths and their averages (metrics.py:163-198).

            Output rows: length | proportional trans err | proportional rot err
        """
        errs = []
        starts = np.arange(self.num_poses)
        for length in segment_lengths:
            length = self._convert_meters(length, trans_unit)
            # first pose whose distance from the segment start exceeds `length`
            stops = np.searchsorted(self.cum_dists, self.cum_dists + length, side='right')
            # the reference searches cum_dists - cum_dists[start] (metrics.py:178-180); make the vectorised search
            # agree with that exact arithmetic where rounding puts a pose on the other side of `length`
            n, cd = self.num_poses, self.cum_dists
            for _ in range(3):
                right = (stops < n) & (cd[np.minimum(stops, n - 1)] - cd[starts] <= length)
                left = (stops > 0) & (cd[np.maximum(stops - 1, 0)] - cd[starts] > length)
                if not (np.any(right) or np.any(left)):
                    break
                stops = stops + right.astype(int) - (left & ~right).astype(int)
            ok = stops < self.num_poses
            if np.any(ok):
                trans, rot = self._endpoint_errors(starts[ok], stops[ok])
                trans = self._convert_meters(trans, trans_unit)
                rot = self._convert_radians(rot, rot_unit)
                errs.append(np.stack([np.full(trans.shape, float(length)), trans / length, rot / length], axis=1))
        errs = np.concatenate(errs, axis=0) if errs else np.array([])
        avg_errs = []
        for length in segment_lengths:
            length = self._convert_meters(length, trans_unit)
            avg_errs.append(np.mean(errs[errs[:, 0] == length], axis=0))
        return errs, np.array(avg_errs)

    def traj_errors(self, segment_range=None, trans_unit='m', rot_unit='rad'):
        """Per-pose translational / rotational errors in all degrees of freedom (metrics.py:200-219)."""
        idx = self._indices(segment_range, self.num_poses)
        R0, t0 = _inv(self._Rg[idx[0]], self._tg[idx[0]])
        Rg, tg = _mul(R0, t0, self._Rg[idx], self._tg[idx])
        Re, te = _mul(R0, t0, self._Re[idx], self._te[idx])        # (sic: relative to the GROUND-TRUTH start)
        Rerr, terr = _mul(*_inv(Re, te), Rg, tg)
        return self._convert_meters(terr, trans_unit), self._convert_radians(_rot_log(Rerr), rot_unit)

    def rel_errors(self, segment_range=None, trans_unit='m', rot_unit='rad', delta=1):
        """Relative pose errors over `delta` frames, Sturm et al. eq. (1) (metrics.py:221-243)."""
        idx = self._indices(segment_range, self.num_poses)[:-delta]
        Rg, tg = _mul(*_inv(self._Rg[idx], self._tg[idx]), self._Rg[idx + delta], self._tg[idx + delta])
        Re, te = _mul(*_inv(self._Re[idx], self._te[idx]), self._Re[idx + delta], self._te[idx + delta])
        Rerr, terr = _mul(*_inv(Rg, tg), Re, te)
        return self._convert_meters(terr, trans_unit), self._convert_radians(_rot_log(Rerr), rot_unit)

    def error_norms(self, segment_range=None, trans_unit='m', rot_unit='rad', error_type='traj', delta=1):
        if error_type == 'traj':
            trans_errs, rot_errs = self.traj_errors(segment_range, trans_unit, rot_unit)
        elif error_type == 'rel':
            trans_errs, rot_errs = self.rel_errors(segment_range, trans_unit, rot_unit, delta)
        else:
            raise ValueError('error_type must be either `traj` or `rel`.')
        return np.sqrt(np.sum(trans_errs**2, axis=1)), np.sqrt(np.sum(rot_errs**2, axis=1))

    def mean_err(self, segment_range=None, trans_unit='m', rot_unit='rad', error_type='traj'):
        trans_norms, rot_norms = self.error_norms(segment_range, trans_unit, rot_unit, error_type)
        return np.mean(trans_norms), np.mean(rot_norms)

    def cum_err(self, segment_range=None, trans_unit='m', rot_unit='rad', error_type='traj'):
        trans_norms, rot_norms = self.error_norms(segment_range, trans_unit, rot_unit, error_type)
        return np.cumsum(trans_norms), np.cumsum(rot_norms)

    def rms_err(self, segment_range=None, trans_unit='m', rot_unit='rad', error_type='traj', delta=1):
        trans_norms, rot_norms = self.error_norms(segment_range, trans_unit, rot_unit, error_type, delta)
        return np.sqrt(np.mean(trans_norms**2)), np.sqrt(np.mean(rot_norms**2))
