"""world_size-2 gloo test of the landmark-sharded iteration (CPU).

The HIP core cannot run here, so the per-shard device is replaced by a numpy
stand-in built on the ORACLE (tests may use it): it produces the shard's partial
reduced system exactly as the device does, and the test checks that sharding,
pattern union, the [S|g|cost] all-reduce and the scalar reductions reproduce
the single-process Gauss-Newton step of the oracle.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import gn_oracle as orc
from pyslam_amd import synthetic
from pyslam_amd.distributed import shard_landmarks, pose_pair_keys, ShardedDeviceProblem


class OracleShardDevice:
    """numpy stand-in for DeviceProblem: dense reduced system over ALL reduced poses."""

    def __init__(self, lp, extra_pairs):
        self.lp = lp
        d, nr = lp.dof, lp.num_reduced
        self.n = d * nr
        self.reduce_tensor = torch.zeros(self.n * self.n + self.n + 1, dtype=torch.float64)
        self.info = {'num_obs': lp.num_obs, 'num_reduced': nr}
        self.extra_pairs = extra_pairs

    def _split(self):
        P, b, cost = orc.normal_equations(self.lp, points_first=False)
        P = P.toarray()
        n = self.n
        return P[:n, :n], P[:n, n:], P[n:, n:], b[:n], b[n:], cost

    def eval_cost(self, include_all=True):
        return orc.eval_cost(self.lp, include_all)

    def linearize(self, lam):
        Hpp, Hpl, Hll, bp, bl, cost = self._split()
        self.Hinv = np.linalg.inv(Hll) if Hll.size else Hll
        self.Hpl, self.bl = Hpl, bl
        S = Hpp - Hpl @ self.Hinv @ Hpl.T
        g = bp - Hpl @ self.Hinv @ bl
        self.reduce_tensor[:] = torch.from_numpy(np.concatenate([S.ravel(), g, [cost]]))

    def solve_reduced(self, tol, max_iters):
        buf = self.reduce_tensor.numpy()
        n = self.n
        self.dxp = np.linalg.solve(buf[:n * n].reshape(n, n), buf[n * n:n * n + n])
        return 1, 0.0

    def gn_solve_finish(self, tol, max_iters, linesearch):
        its, rel = self.solve_reduced(tol, max_iters)
        return self.gn_finish(linesearch) + (its, rel)

    def gn_finish(self, linesearch):
        dxl = self.Hinv @ (self.bl - self.Hpl.T @ self.dxp)
        dx = np.concatenate([self.dxp, dxl])
        lin_cost = orc.eval_cost(self.lp, False)
        self.lp = orc.apply_update(self.lp, dx, points_first=False)
        cost = orc.eval_cost(self.lp, True) if linesearch else lin_cost
        return cost, float(self.dxp @ self.dxp), float(dxl @ dxl)


    def segment_layout(self):
        """(indices(blocks, poses), tail) of this stand-in's buffer [dense S | g | cost] for SegmentExchange."""
        n, d = self.n, self.lp.dof

        def indices(blocks, poses):
            ri, rj = (blocks >> 32).astype(np.int64), (blocks & 0xFFFFFFFF).astype(np.int64)
            e = np.arange(d)
            rows = (ri[:, None, None] * d + e[None, :, None]) * n + rj[:, None, None] * d + e[None, None, :]
            rowsT = (rj[:, None, None] * d + e[None, :, None]) * n + ri[:, None, None] * d + e[None, None, :]
            grad = n * n + (poses[:, None] * d + e[None, :]).ravel()
            return np.unique(np.concatenate([rows.ravel(), rowsT.ravel(), grad])).astype(np.int64)
        return indices, np.array([n * n + n], dtype=np.int64)

    # parameter tables in and out (what Problem.solve drives through ShardedProblemView)
    def get_params(self):
        return self.lp.poses.copy(), self.lp.points.copy()

    def set_params(self, poses=None, points=None):
        self.lp = self.lp.copy()
        if poses is not None:
            self.lp.poses = np.array(poses, dtype=float)
        if points is not None:
            self.lp.points = np.array(points, dtype=float).reshape(-1, 3)

    def snapshot(self):
        self._snap = self.get_params()

    def restore(self):
        self.set_params(*self._snap)

    def close(self):
        pass


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out, exchange='allreduce', kf=8, lm=90):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    lp, _ = synthetic.stereo_ba(num_kf=kf, num_lm=lm, obs_per_lm=4, half_window=3, seed=11)
    # an odometry edge + prior so factors exist (rank 0 only after sharding)
    shard = shard_landmarks(lp, rank, world)
    sp = ShardedDeviceProblem(shard, dist, device_factory=lambda l, e: OracleShardDevice(l, e), exchange=exchange)
    if exchange == 'segments':
        assert sp.segments is not None and sp.segments.bytes_sent < sp.segments.bytes_allreduce
    c0 = sp.eval_cost(True)
    cost, nrm, _, _ = sp.gn_iteration(0., 1e-12, 100, True)
    poses, _ = sp.dev.lp.poses, None
    if rank == 0:
        out.put((c0, cost, nrm, sp.pattern_keys, sp.dev.lp.poses.copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_iteration_matches_single_process():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    c0, cost, nrm, keys, poses = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    lp, _ = synthetic.stereo_ba(num_kf=8, num_lm=90, obs_per_lm=4, half_window=3, seed=11)
    assert abs(c0 - orc.eval_cost(lp)) <= 1e-12 * c0
    dx, _ = orc.gauss_newton_step(lp, points_first=False)
    new = orc.apply_update(lp, dx, points_first=False)
    assert abs(cost - orc.eval_cost(new)) <= 1e-9 * cost
    assert abs(nrm - np.linalg.norm(dx)) <= 1e-9 * nrm
    assert np.abs(poses - new.poses).max() < 1e-9
    assert np.array_equal(keys, pose_pair_keys(lp))          # union of shard patterns == global pattern


@pytest.mark.parametrize('world', [2, 3])
def test_segment_all_gather_gives_the_all_reduced_system(world):
    """Round 5: landmarks sharded by first observing pose + the exchange as an all-gather of band segments summed in a fixed order
    (pyslam_amd/distributed.py: SegmentExchange) -- the same Gauss-Newton step as the single-process oracle, on a chain long
    enough (24 keyframes, windows of 7) that a rank's segment is a proper part of the system."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, 'segments', 24, 300)) for r in range(world)]
    for p in procs:
        p.start()
    c0, cost, nrm, keys, poses = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    lp, _ = synthetic.stereo_ba(num_kf=24, num_lm=300, obs_per_lm=4, half_window=3, seed=11)
    assert abs(c0 - orc.eval_cost(lp)) <= 1e-12 * c0
    dx, _ = orc.gauss_newton_step(lp, points_first=False)
    new = orc.apply_update(lp, dx, points_first=False)
    assert abs(cost - orc.eval_cost(new)) <= 1e-9 * cost
    assert abs(nrm - np.linalg.norm(dx)) <= 1e-9 * nrm
    assert np.abs(poses - new.poses).max() < 1e-9


def test_locality_shards_touch_band_segments():
    """landmark_owner_lists(order='first_pose'): a rank's landmarks see a contiguous stretch of the trajectory, so its blocks
    of S are a fraction of the pattern; the caller-index split of rounds 1-4 touches (nearly) all of it on every rank."""
    from pyslam_amd.distributed import landmark_owner_lists, shard_touch
    lp, _ = synthetic.stereo_ba(num_kf=80, num_lm=2000, obs_per_lm=5, half_window=4, seed=5)
    total = pose_pair_keys(lp).size + lp.num_reduced
    for order, bound in (('first_pose', 0.45), ('index', 1.01)):
        owners = landmark_owner_lists(lp, 4, order)
        assert np.array_equal(np.sort(np.concatenate(owners)), np.arange(lp.num_points))
        frac = [shard_touch(shard_landmarks(lp, r, 4, owners=owners))[0].size / total for r in range(4)]
        assert max(frac) <= bound, (order, frac)
        if order == 'index':
            assert min(frac) > 0.9, frac
        obs = [shard_landmarks(lp, r, 4, owners=owners).num_obs for r in range(4)]
        assert max(obs) - min(obs) <= 0.1 * lp.num_obs / 4 + 10


def test_owner_lists_and_packed_layout_edge_cases():
    """landmark_owner_lists: a partition for any world size (also more ranks than landmarks with observations), unobserved
    landmarks sort last; packed_layout: positions inside [upper(S) | g | cost (2) | flag] in ascending key order, diagonals
    included, a block outside the pattern refused."""
    from pyslam_amd.distributed import landmark_owner_lists, packed_layout, shard_touch
    lp, _ = synthetic.stereo_ba(num_kf=10, num_lm=30, obs_per_lm=3, half_window=3, seed=1)
    # two landmarks nobody observes (appended): they must end up with the LAST rank
    lp2 = lp.copy()
    lp2.points = np.vstack([lp.points, np.zeros((2, 3))])
    lp2.point_vid = np.concatenate([lp.point_vid, [-1, -1]]).astype(lp.point_vid.dtype)
    lp2.point_keys = list(lp.point_keys) + ['x0', 'x1']
    lp2 = lp2.finalize()
    for world in (1, 2, 3, 8, 40):
        owners = landmark_owner_lists(lp2, world)
        assert len(owners) == world
        assert np.array_equal(np.sort(np.concatenate(owners)), np.arange(lp2.num_points))
        if world > 1:
            assert {30, 31} <= set(owners[-1].tolist())
        for r in range(world):
            sh = shard_landmarks(lp2, r, world, owners=owners)
            assert sh.num_points == owners[r].size and sh.num_obs == int(np.isin(lp2.obs_point, owners[r]).sum())
    # packed_layout
    keys = pose_pair_keys(lp)
    nr, d = lp.num_reduced, lp.dof
    indices, tail = packed_layout(keys, nr, d)
    allk = np.unique(np.concatenate([keys, (np.arange(nr, dtype=np.int64) << 32) | np.arange(nr, dtype=np.int64)]))
    nup = allk.size
    assert np.array_equal(tail, nup * d * d + nr * d + np.arange(3))
    blocks, poses = shard_touch(lp)
    idx = indices(blocks, poses)
    assert np.array_equal(np.sort(idx), np.arange(nup * d * d + nr * d))      # the whole problem touches everything but the tail
    one = indices(allk[5:6], np.array([2], dtype=np.int64))
    assert np.array_equal(one, np.concatenate([5 * d * d + np.arange(d * d), nup * d * d + 2 * d + np.arange(d)]))
    with pytest.raises(ValueError):
        indices(np.array([(np.int64(nr + 3) << 32) | np.int64(nr + 4)]), np.zeros(0, np.int64))


def test_shards_partition_the_problem():
    lp, _ = synthetic.stereo_ba(num_kf=10, num_lm=200, obs_per_lm=5, half_window=4, seed=3,
                                const_point_fraction=0.1)
    shards = [shard_landmarks(lp, r, 4) for r in range(4)]
    assert sum(s.num_obs for s in shards) == lp.num_obs
    assert sum(s.num_points for s in shards) == lp.num_points
    assert sum(s.num_var_points for s in shards) == lp.num_var_points
    counts = [s.num_obs for s in shards]
    assert max(counts) - min(counts) <= 0.2 * lp.num_obs / 4 + 10     # balanced by observations
    for s in shards:
        assert np.array_equal(s.poses, lp.poses) and np.array_equal(s.pose_rid, lp.pose_rid)
        assert set(s.point_vid[s.point_vid >= 0]) == set(range(s.num_var_points))
    # cost is additive over shards
    assert abs(sum(orc.eval_cost(s) for s in shards) - orc.eval_cost(lp)) <= 1e-10 * orc.eval_cost(lp)
    # the union of the shards' block patterns is the global pattern
    union = np.unique(np.concatenate([pose_pair_keys(s) for s in shards]))
    assert np.array_equal(union, pose_pair_keys(lp))


def test_pose_graph_factors_live_on_rank0_only():
    lp, _ = synthetic.pose_graph(num_poses=30, num_loops=20, dof=6, seed=1)
    s0, s1 = shard_landmarks(lp, 0, 2), shard_landmarks(lp, 1, 2)
    assert s0.num_edges == lp.num_edges and s0.num_priors == lp.num_priors
    assert s1.num_edges == 0 and s1.num_priors == 0


def _solve_worker(rank, world, port, out):
    """Problem.solve() through the public API with Options.devices = 'auto': every rank runs the same script on the
    same Problem; the tables are sharded by landmark behind it."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_host_api import build_namespace
    import pyslam_amd.distributed as D
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    D.DEVICE_FACTORY = lambda l, e: OracleShardDevice(l, e)
    lp, _ = synthetic.stereo_ba(num_kf=8, num_lm=90, obs_per_lm=4, half_window=3, seed=11)
    ns = build_namespace()
    opts = ns.Options()
    opts.devices = 'auto'
    opts.allow_nondecreasing_steps = True
    opts.max_nondecreasing_steps = 3
    problem = synthetic.to_objects(lp, ns, options=opts)
    assert problem._route() == 'sharded'
    params = problem.solve()
    poses = np.stack([params[k].as_matrix() for k in lp.pose_keys])
    points = np.stack([params[k] for k in lp.point_keys])
    if rank == 0:
        out.put((list(problem._cost_history), poses, points, type(problem._device).__name__))
    dist.barrier()
    dist.destroy_process_group()


def test_problem_solve_routes_to_the_sharded_driver():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_solve_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    hist, poses, points, devname = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert devname == 'ShardedProblemView'
    lp, _ = synthetic.stereo_ba(num_kf=8, num_lm=90, obs_per_lm=4, half_window=3, seed=11)
    final, trace = orc.solve(lp, dict(allow_nondecreasing_steps=True, max_nondecreasing_steps=3), points_first=True)
    ref = trace['cost_history']
    assert len(hist) == len(ref) and np.allclose(hist, ref, rtol=1e-7, atol=1e-18)
    from pyslam_amd.lowering import pose_rows_to_matrices
    assert np.abs(poses - pose_rows_to_matrices(final.poses, 6)).max() < 1e-8
    assert np.abs(points - final.points).max() < 1e-8


def test_segment_plan_sums_like_a_dense_all_reduce_in_rank_order():
    """The plan the core's segment exchange runs (include/pyslam_hip.h: ps_set_segment_exchange; pyslam_amd.distributed.segment_plan),
    emulated with numpy: k_seg_pack on every rank, the all-gather, k_seg_sum -- against the dense sum of the ranks' packed buffers,
    with every destination's contributions taken in ascending rank order (what makes the sum the same number on every rank)."""
    from pyslam_amd.distributed import segment_plan
    rng = np.random.default_rng(5)
    for world in (1, 2, 3, 8):
        n, T = 5000, 3                                        # packed buffer: n body elements + 3 tail words
        idx = []
        for r in range(world):                                # overlapping windows, as landmark shards of a trajectory give
            lo = int(n * r / world * 0.9); hi = min(n, lo + int(n / world * 1.6) + 1)
            idx.append(np.sort(rng.choice(np.arange(lo, hi), size=(hi - lo) * 3 // 4, replace=False)).astype(np.int64))
        maxlen, dst, src_ptr, src_off = segment_plan(idx, T)
        assert maxlen == T + max(i.size for i in idx) and src_ptr[0] == 0 and src_ptr[-1] == src_off.size == sum(i.size for i in idx)
        assert np.array_equal(dst, np.unique(np.concatenate(idx)))
        packs = []
        for r in range(world):
            p = np.zeros(n + T); p[idx[r]] = rng.standard_normal(idx[r].size) * 10.0 ** rng.integers(-6, 6, idx[r].size)
            p[n:] = rng.standard_normal(T); packs.append(p)
        seg_all = np.zeros(world * maxlen)
        for r in range(world):                                # k_seg_pack + all-gather
            seg_all[r * maxlen:r * maxlen + T] = packs[r][n:]
            seg_all[r * maxlen + T:r * maxlen + T + idx[r].size] = packs[r][idx[r]]
        for me in range(world):                               # k_seg_sum on rank `me`: into ITS packed buffer
            out = packs[me].copy()
            for k in range(dst.size):
                v = 0.0
                ranks = []
                for q in range(src_ptr[k], src_ptr[k + 1]):
                    v += seg_all[src_off[q]]; ranks.append(src_off[q] // maxlen)
                assert ranks == sorted(ranks) and len(set(ranks)) == len(ranks)
                out[dst[k]] = v
            for w in range(T):
                out[n + w] = sum(seg_all[r * maxlen + w] for r in range(world))
            ref = np.zeros(n + T)
            for r in range(world):                            # the dense sum, in rank order
                ref += packs[r]
            assert np.array_equal(out, ref), (world, me)
    with pytest.raises(ValueError):
        segment_plan([np.array([1, 1, 2], dtype=np.int64)], 3)
