"""Landmark-sharded multi-GPU Gauss-Newton.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).
Landmarks -- and therefore observations -- are partitioned across ranks; the
pose table is replicated.  Per iteration:

    ps_linearize            shard-local: residuals, Jacobians, H_ll, Z, partial S, g
    ONE sum all-reduce      [upper block triangle of S | g | cost | failure flag], packed by the core
                            (ps_shard_pack / ps_shard_unpack): S is symmetric, one triangle travels
    ps_gn_solve_finish      replicated, deterministic reduced solve (identical dx_pose on every
                            rank) + shard-local back-substitution, update, cost; one host sync
    scalar all-reduce       (cost, ||dx_point||^2)

The reduced system's block pattern must be identical on every rank for the
element-wise all-reduce, so ranks exchange their co-visibility pose pairs once
at construction and pass the union to ps_problem_create as ``extra_pairs``.
Pose factors (edges / priors) must live on exactly one rank (``shard_landmarks``
gives them to rank 0): summing replicated factors would count them N times.
"""
import numpy as np

from pyslam_amd.lowering import LoweredProblem


def landmark_owner_lists(lp, world, order='first_pose'):
    """Which landmarks (the caller's numbering, ascending) every rank owns.

    ``order='first_pose'`` (default, round 5): landmarks are put in the order of the LOWEST POSE INDEX THAT OBSERVES them --
    the order the device gives its landmark slots (csrc/ps_abi_problem.h "internal landmark order") -- and that order is
    cut into ``world`` contiguous runs balanced by observation count.  On a trajectory a landmark is seen from a window of
    poses, so a rank's landmarks touch one SEGMENT of the pose chain: its partial reduced system is a band segment of S
    (plus a halo of one co-visibility window on either side), its Schur task list is 1/world of the whole one, and what it
    has to contribute to the exchange is that segment only (``ShardedDeviceProblem``: segment all-gather).
    ``order='index'`` (rounds 1-4): contiguous ranges of the caller's landmark index -- with landmarks numbered in no
    particular order every shard touches every block of S."""
    L = lp.num_points
    if world == 1:
        return [np.arange(L, dtype=np.int64)]
    counts = np.bincount(lp.obs_point, minlength=L).astype(np.int64)
    if order == 'first_pose':
        first = np.full(L, lp.num_poses, dtype=np.int64)          # (a landmark nobody observes sorts last)
        if lp.num_obs:                                            # lowest observing pose per landmark: sort + reduceat (np.minimum.at
            o = np.argsort(lp.obs_point, kind='stable')           # walks 5 M observations one by one at C4 on older numpy: ADVICE)
            pts = np.asarray(lp.obs_point)[o]
            starts = np.flatnonzero(np.concatenate([[True], pts[1:] != pts[:-1]]))
            first[pts[starts]] = np.minimum.reduceat(np.asarray(lp.obs_pose, dtype=np.int64)[o], starts)
        perm = np.lexsort((np.arange(L), first))
    elif order == 'index':
        perm = np.arange(L, dtype=np.int64)
    else:
        raise ValueError("order must be 'first_pose' or 'index'")
    cum = np.concatenate([[0], np.cumsum(counts[perm])])
    total = cum[-1]
    cuts = [int(np.searchsorted(cum, total * r / world, side='left')) for r in range(world)] + [L]
    return [np.sort(perm[cuts[r]:cuts[r + 1]]) for r in range(world)]


def shard_landmarks(lp, rank, world, order='first_pose', owners=None):
    """This rank's landmark shard (``landmark_owner_lists``), balanced by observation count; factors on rank 0."""
    if world == 1:
        return lp
    idx = (owners if owners is not None else landmark_owner_lists(lp, world, order))[rank]
    L = lp.num_points
    remap = np.full(L, -1, dtype=np.int64)
    remap[idx] = np.arange(idx.size)
    new_point = remap[lp.obs_point]
    keep = new_point >= 0
    out = lp.copy()
    out.points = lp.points[idx]
    vid = lp.point_vid[idx].copy()
    var = vid >= 0
    vid[var] = np.arange(int(var.sum()))
    out.point_vid = vid
    out.point_keys = [lp.point_keys[i] for i in idx]
    out.obs_pose, out.obs_point = lp.obs_pose[keep], new_point[keep].astype(lp.obs_point.dtype)
    out.obs_uvd, out.obs_grp = lp.obs_uvd[keep], lp.obs_grp[keep]
    if rank != 0:
        pw = lp.pose_width
        out.e_i = out.e_j = out.e_grp = out.u_i = out.u_grp = np.zeros(0, np.int32)
        out.e_Tobs_inv = out.u_Tobs_inv = np.zeros((0, pw))
    return out.finalize()


def pose_pair_keys(lp):
    """Unique upper-triangle reduced pose pairs (ri < rj) this shard couples:
    co-visibility through variable landmarks + pose-pose edges.  int64 ri<<32|rj."""
    keys = []
    if lp.num_obs:
        rid = lp.pose_rid[lp.obs_pose]
        sel = (rid >= 0) & (lp.point_vid[lp.obs_point] >= 0)
        pt, rid = lp.obs_point[sel], rid[sel].astype(np.int64)
        order = np.argsort(pt, kind='stable')
        pt, rid = pt[order], rid[order]
        shift = 1
        while shift < pt.size:
            same = pt[shift:] == pt[:-shift]
            if not same.any():
                break
            a, b = rid[:-shift][same], rid[shift:][same]
            ne = a != b
            keys.append(np.unique((np.minimum(a, b)[ne] << 32) | np.maximum(a, b)[ne]))
            shift += 1
    if lp.num_edges:
        a, b = lp.pose_rid[lp.e_i].astype(np.int64), lp.pose_rid[lp.e_j].astype(np.int64)
        ok = (a >= 0) & (b >= 0) & (a != b)
        keys.append(np.unique((np.minimum(a, b)[ok] << 32) | np.maximum(a, b)[ok]))
    return np.unique(np.concatenate(keys)) if keys else np.zeros(0, np.int64)


def shard_touch(lp):
    """(upper block keys incl. diagonals, reduced poses) this shard's partial reduced system can be non-zero in: blocks
    (ri<<32|rj, ri <= rj) coupled through its landmarks / factors, the diagonal block and gradient rows of every variable
    pose it observes or has a factor on."""
    rid = lp.pose_rid.astype(np.int64)
    touched = [rid[lp.obs_pose]] if lp.num_obs else []
    if lp.num_edges:
        touched += [rid[lp.e_i], rid[lp.e_j]]
    if getattr(lp, 'num_priors', 0):
        touched.append(rid[lp.u_i])
    poses = np.unique(np.concatenate(touched)) if touched else np.zeros(0, np.int64)
    poses = poses[poses >= 0]
    blocks = np.unique(np.concatenate([pose_pair_keys(lp), (poses << 32) | poses]))
    return blocks, poses


def packed_layout(pattern_keys, nr, dof):
    """Element indices inside the core's exchange buffer [upper(S) | g | cost (2) | flag] (csrc/ps_k_tail.h: k_shard_pack; the
    upper blocks in (row, column) order = ascending key order, diagonals included).  -> (indices(blocks, poses), tail indices)."""
    all_keys = np.unique(np.concatenate([pattern_keys, (np.arange(nr, dtype=np.int64) << 32) | np.arange(nr, dtype=np.int64)]))
    dd, nup = dof * dof, all_keys.size

    def indices(blocks, poses):
        slot = np.searchsorted(all_keys, blocks)
        if blocks.size and (slot.max() >= nup or not np.array_equal(all_keys[slot], blocks)):
            raise ValueError('a shard touches a block outside the common pattern')
        blk = (slot[:, None] * dd + np.arange(dd)[None, :]).ravel()
        grad = nup * dd + (poses[:, None] * dof + np.arange(dof)[None, :]).ravel()
        return np.concatenate([blk, grad]).astype(np.int64)
    tail = nup * dd + nr * dof + np.arange(3, dtype=np.int64)
    return indices, tail


def segment_plan(idx_lists, tail_len=3):
    """The summation plan of the segment exchange as the core runs it (include/pyslam_hip.h: ps_set_segment_exchange).
    idx_lists[r] = the element positions rank r sends (in the packed buffer [upper(S) | g | tail]); every rank's buffer in the
    gathered array is [tail words | its elements | padding] of `maxlen` doubles.  -> (maxlen, dst, src_ptr, src_off): every
    position some rank touches, and for each the offsets r * maxlen + tail_len + k of its contributions in ASCENDING RANK order --
    the order every rank adds them up in, hence bit-identical sums everywhere.  Pure numpy: the CPU tests hold it against a
    dense sum (tests/test_distributed_cpu.py)."""
    lens = [int(i.size) for i in idx_lists]
    maxlen = tail_len + (max(lens) if lens else 0)
    if not lens or sum(lens) == 0:
        return maxlen, np.zeros(0, np.int64), np.zeros(1, np.int64), np.zeros(0, np.int64)
    pos = np.concatenate([np.asarray(i, dtype=np.int64) for i in idx_lists])
    off = np.concatenate([r * maxlen + tail_len + np.arange(n, dtype=np.int64) for r, n in enumerate(lens)])
    for r, i in enumerate(idx_lists):
        if np.unique(i).size != i.size:
            raise ValueError('segment_plan: rank {} lists an element twice'.format(r))
    order = np.argsort(pos, kind='stable')               # stable: equal positions stay in rank order (the lists are concatenated by rank)
    pos, off = pos[order], off[order]
    first = np.flatnonzero(np.concatenate([[True], pos[1:] != pos[:-1]]))
    return maxlen, pos[first], np.concatenate([first, [pos.size]]).astype(np.int64), off


class SegmentExchange:
    """Round 5: the exchange of the partial reduced systems as an ALL-GATHER OF SEGMENTS instead of a sum all-reduce of
    the whole buffer.  With landmarks sharded by first observing pose (landmark_owner_lists) a rank's partial system is
    non-zero on one band segment of S (+ a halo of one co-visibility window) -- C4 on 8 ranks: 3.4 MB of the 23.5 MB
    buffer -- so every rank sends its segment once, receives the others', and adds them up itself in a FIXED order (the
    same on every rank: the replicated reduced solve stays bit-identical across ranks).  A ring all-reduce moves
    2 (N-1)/N x 23.5 MB per link, the segment all-gather (N-1) x 3.4 MB; with neighbours overlapping only pairwise the
    sum is two scatter-adds (even ranks, then odd ranks: no index twice inside one of them, hence no atomics race).
    Buffer per rank: [tail words (cost, flag: summed over ranks) | its elements], padded to the longest.
    Unmeasured on hardware (one GPU per lease): opt-in (ShardedDeviceProblem(exchange='segments') /
    PYSLAM_AMD_EXCHANGE=segments); the sum all-reduce stays the default."""

    def __init__(self, dist, torch, reduce_tensor, idx_lists, tail_idx):
        self.dist, self.torch = dist, torch
        self.buf = reduce_tensor
        dev = reduce_tensor.device
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.T = int(tail_idx.size)
        # the element lists come from packed_layout, a HOST restatement of k_shard_pack's layout: hold it against the real buffer
        # (ps_reduce_buffer's count) -- a pattern the device packs differently would scatter into the wrong slots, silently (ADVICE)
        if int(tail_idx.max()) + 1 != int(reduce_tensor.numel()):
            raise ValueError('SegmentExchange: the packed layout ends at element {} but the core\'s exchange buffer holds {} '
                             '(the device pattern differs from the one the shards agreed on)'.format(int(tail_idx.max()) + 1,
                                                                                                   int(reduce_tensor.numel())))
        for i in idx_lists:
            if i.size and (int(i.min()) < 0 or int(i.max()) >= int(tail_idx.min())):
                raise ValueError('SegmentExchange: a segment element lies outside [upper(S) | g]')
        self.lens = [int(i.size) for i in idx_lists]
        self.maxlen = self.T + max(self.lens)
        self.tail = torch.as_tensor(tail_idx, device=dev)
        self.mine = torch.as_tensor(idx_lists[self.rank], device=dev)
        self.seg_in = torch.zeros(self.maxlen, dtype=reduce_tensor.dtype, device=dev)
        self.seg_all = torch.zeros(self.world, self.maxlen, dtype=reduce_tensor.dtype, device=dev)
        # passes: ranks whose index sets are pairwise disjoint are added in ONE scatter-add (greedy colouring in rank
        # order; a chain of windows gives two passes, anything gives at most `world`)
        passes, sets = [], []
        for r, idx in enumerate(idx_lists):
            for k, members in enumerate(passes):
                if not np.intersect1d(sets[k], idx, assume_unique=True).size:
                    members.append(r); sets[k] = np.union1d(sets[k], idx)
                    break
            else:
                passes.append([r]); sets.append(idx)
        self.passes = []
        for members in passes:
            dst = np.concatenate([idx_lists[r] for r in members])
            src = np.concatenate([r * self.maxlen + self.T + np.arange(self.lens[r], dtype=np.int64) for r in members])
            self.passes.append((torch.as_tensor(dst, device=dev), torch.as_tensor(src, device=dev)))
        self.bytes_sent = 8 * (self.T + self.lens[self.rank])
        self.bytes_allreduce = 8 * int(reduce_tensor.numel())

    def run(self):
        """reduce_tensor (this rank's partial system, packed) -> the sum over ranks, in place."""
        t = self.torch
        n = self.lens[self.rank]
        self.seg_in[:self.T] = self.buf[self.tail]
        t.index_select(self.buf, 0, self.mine, out=self.seg_in[self.T:self.T + n])
        if self.dist.get_backend() == 'nccl':
            self.dist.all_gather_into_tensor(self.seg_all, self.seg_in)
        else:
            self.dist.all_gather(list(self.seg_all.unbind(0)), self.seg_in)
        self.buf.zero_()
        flat = self.seg_all.view(-1)
        for dst, src in self.passes:
            self.buf.index_add_(0, dst, flat.index_select(0, src))
        self.buf[self.tail] = self.seg_all[:, :self.T].sum(0)


class NativeRccl:
    """An ncclComm_t of our own (one rank per process) created through ctypes on the RCCL that
    PyTorch already loaded; the unique id travels over the existing torch.distributed group.
    The HIP core calls ncclAllReduce itself, on the solver's stream (ps_set_collective)."""

    class _UniqueId(__import__('ctypes').Structure):
        _fields_ = [('internal', __import__('ctypes').c_ubyte * 128)]   # c_char would truncate at NUL

    def __init__(self, dist):
        import ctypes as C
        import os
        import torch
        path = os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so')
        self.lib = C.CDLL(path)
        uid = self._UniqueId()
        if dist.get_rank() == 0:
            rc = self.lib.ncclGetUniqueId(C.byref(uid))
            if rc != 0:
                raise RuntimeError('ncclGetUniqueId failed: {}'.format(rc))
        box = [bytes(bytearray(uid.internal)) if dist.get_rank() == 0 else None]
        dist.broadcast_object_list(box, src=0)
        C.memmove(C.byref(uid), box[0], 128)
        self.comm = C.c_void_p()
        self.lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, self._UniqueId, C.c_int]
        rc = self.lib.ncclCommInitRank(C.byref(self.comm), dist.get_world_size(), uid, dist.get_rank())
        if rc != 0:
            raise RuntimeError('ncclCommInitRank failed: {}'.format(rc))
        self.allreduce_ptr = C.cast(self.lib.ncclAllReduce, C.c_void_p).value
        self.allgather_ptr = C.cast(self.lib.ncclAllGather, C.c_void_p).value
        try:
            self.self_check(dist)
        except Exception:
            # a communicator that failed its check must not stay alive beside the torch.distributed fallback
            # (self_check aborts it itself on a timeout and clears self.comm)
            if getattr(self, 'comm', None):
                try:
                    self.lib.ncclCommAbort.argtypes = [C.c_void_p]
                    self.lib.ncclCommAbort(self.comm)
                finally:
                    self.comm = None
            raise

    def self_check(self, dist, timeout_s=30.0):
        """One 1-element sum over the new communicator, on a stream of its own, with a deadline: this second
        communicator (beside PyTorch's) has to prove it works with all ranks before the solver's stream depends on it.
        A wrong sum, an error code or no completion within `timeout_s` raises; the caller falls back to the
        torch.distributed collectives on every rank (the choice is agreed with a MIN all-reduce there)."""
        import ctypes as C
        import time
        import torch
        world = dist.get_world_size()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            t = torch.ones(1, dtype=torch.float64, device='cuda')
            self.lib.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
            self.lib.ncclAllReduce.restype = C.c_int
            rc = self.lib.ncclAllReduce(C.c_void_p(t.data_ptr()), C.c_void_p(t.data_ptr()), 1, 8, 0, self.comm,
                                        C.c_void_p(side.cuda_stream))      # ncclFloat64 = 8, ncclSum = 0
            if rc != 0:
                raise RuntimeError('ncclAllReduce (self-check) returned {}'.format(rc))
            ev = torch.cuda.Event()
            ev.record(side)
        t0 = time.time()
        while not ev.query():
            if time.time() - t0 > timeout_s:
                try:
                    self.lib.ncclCommAbort.argtypes = [C.c_void_p]
                    self.lib.ncclCommAbort(self.comm)
                finally:
                    self.comm = None
                raise RuntimeError('self-check all-reduce over {} ranks did not complete in {:.0f} s'.format(world, timeout_s))
            time.sleep(0.0005)
        got = float(t.item())
        if got != float(world):
            raise RuntimeError('self-check all-reduce returned {} instead of {}'.format(got, world))

    def close(self):
        if getattr(self, 'comm', None):
            self.lib.ncclCommDestroy.argtypes = [__import__('ctypes').c_void_p]
            self.lib.ncclCommDestroy(self.comm)
            self.comm = None


class _RawDeviceArray:
    """Zero-copy view of a device pointer for torch.as_tensor (CUDA array interface)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {'shape': (n,), 'typestr': '<f8', 'data': (ptr, False),
                                         'version': 2, 'strides': None}


# tests (no GPU) may install a stand-in here: callable (lp_shard, extra_pairs) -> device; None = the HIP core
DEVICE_FACTORY = None


def _default_device_factory(lp, extra_pairs):
    import torch
    from pyslam_amd.device import DeviceProblem
    dev = DeviceProblem(lp, stream=torch.cuda.current_stream().cuda_stream, extra_pairs=extra_pairs)
    ptr, n = dev.reduce_buffer()
    dev.reduce_tensor = torch.as_tensor(_RawDeviceArray(ptr, n), device='cuda')
    dev.shard_tensor = torch.as_tensor(_RawDeviceArray(dev.shard_buffer(), 2), device='cuda')
    return dev


class ShardedDeviceProblem:
    """Same surface as DeviceProblem for the pieces Problem.solve / bench.py use."""

    def __init__(self, lp_shard, dist, device_factory=None, native_rccl=True, exchange=None, pattern_keys=None):
        """exchange: 'allreduce' (default: one sum all-reduce of the whole packed system, driven by the core itself when the
        native RCCL communicator is up) or 'segments' (SegmentExchange; torch.distributed collectives).  pattern_keys: more
        block keys for the common pattern (bench.py --emulate-shard: one rank of N on one GPU with all N ranks' pattern)."""
        import os
        import torch
        self._torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.exchange = exchange or os.environ.get('PYSLAM_AMD_EXCHANGE', 'allreduce')
        if self.exchange not in ('allreduce', 'segments'):
            raise ValueError("exchange must be 'allreduce' or 'segments'")
        mine = pose_pair_keys(lp_shard)
        gathered = [None] * self.world
        dist.all_gather_object(gathered, mine)
        union = np.unique(np.concatenate(gathered)) if gathered else mine
        if pattern_keys is not None:
            union = np.union1d(union, np.asarray(pattern_keys, dtype=np.int64))
        extra = np.setdiff1d(union, mine)
        pairs = ((extra >> 32).astype(np.int32), (extra & 0xFFFFFFFF).astype(np.int32))
        self.pattern_keys = union
        device_factory = device_factory or DEVICE_FACTORY
        self.dev = (device_factory or _default_device_factory)(lp_shard, pairs)
        self.lp = lp_shard
        self.info = dict(self.dev.info)
        # GPU: let the HIP core drive RCCL itself (one ABI call + one sync per iteration)
        self.native = None
        self.native_reason = 'not requested'
        import os
        if device_factory is None and native_rccl and os.environ.get('PYSLAM_AMD_NATIVE_RCCL', '1') != '0':
            try:
                self.native = NativeRccl(dist)
                self.dev.set_collective(self.native.allreduce_ptr, self.native.comm.value)
            except Exception as e:                       # fall back to torch.distributed collectives
                print('pyslam_amd[rank {}]: native RCCL unavailable ({}); using torch.distributed collectives'.format(self.rank, e),
                      flush=True)
                self.native = None
                self.native_reason = str(e)
            # the choice must be the same on every rank (a rank that fell back alone would wait in a different
            # collective than its peers): native only if it came up everywhere
            flag = torch.tensor([1 if self.native is not None else 0], dtype=torch.int32,
                                device=self.dev.reduce_tensor.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0 and self.native is not None:
                print('pyslam_amd[rank {}]: another rank has no native RCCL; using torch.distributed collectives'.format(self.rank),
                      flush=True)
                self.dev.set_collective(0, 0)
                self.native.close()
                self.native = None
                self.native_reason = 'a peer rank fell back'
        # (Round 2 first set "coarse_refresh_every" = 2 from 4 ranks on: the dense side-stream factorisation of C4's coarse
        # level, 1.6 ms, would have outlasted a ~1.1 ms sharded iteration.  The banded factorisation takes 0.6 ms and
        # finishes beside the CG and the next shard-local kernels, so every rank count refreshes every iteration; the
        # option stays for problems whose coarse matrix is not banded.)
        self._scal = torch.zeros(2, dtype=torch.float64, device=self.dev.reduce_tensor.device)
        self._prof_level, self._pending_events, self._host_stage = 0, [], {}
        self.segments = None
        self.core_segments = False
        if self.exchange == 'segments':
            touch = [None] * self.world
            dist.all_gather_object(touch, shard_touch(lp_shard))
            if hasattr(self.dev, 'segment_layout'):          # (the CPU stand-in of the tests has its own buffer layout)
                indices, tail = self.dev.segment_layout()
            else:
                indices, tail = packed_layout(union, lp_shard.num_reduced, lp_shard.dof)
            idx_lists = [indices(b, p) for b, p in touch]
            self._seg_idx_lists, self._seg_tail = idx_lists, tail
            if self.native is not None and hasattr(self.dev, 'set_segment_exchange'):
                self.enable_core_segments(self.native.allgather_ptr)
            else:
                self.segments = SegmentExchange(dist, torch, self.dev.reduce_tensor, idx_lists, tail)
                self.segment_bytes = (self.segments.bytes_sent, self.segments.bytes_allreduce)

    def enable_core_segments(self, allgather_fn_ptr):
        """Round 6: the core's one-call iteration does the segment exchange itself (ncclAllGather -- or a stand-in with its
        signature -- on the solver's stream + the fixed-order sum: include/pyslam_hip.h ps_set_segment_exchange); no torch launches
        in the iteration.  Needs exchange='segments' and the core's collective path (ps_set_collective) set."""
        idx_lists, tail = self._seg_idx_lists, self._seg_tail
        if int(tail.max()) + 1 != int(self.dev.reduce_tensor.numel()):
            raise ValueError('the packed layout ends at element {} but the core\'s exchange buffer holds {}'.format(
                int(tail.max()) + 1, int(self.dev.reduce_tensor.numel())))
        maxlen, dst, src_ptr, src_off = segment_plan(idx_lists, int(tail.size))
        self.dev.set_segment_exchange(allgather_fn_ptr, self.world, self.rank, maxlen, idx_lists[self.rank], dst, src_ptr, src_off)
        self.core_segments, self.segments = True, None
        self.segment_bytes = (8 * (int(tail.size) + int(idx_lists[self.rank].size)), 8 * int(self.dev.reduce_tensor.numel()))

    # ---- iteration -----------------------------------------------------
    def eval_cost(self, include_all_constant=True):
        self._scal[0] = self.dev.eval_cost(include_all_constant)
        self._scal[1] = 0.
        self.dist.all_reduce(self._scal)
        return float(self._scal[0])

    def gn_iteration(self, lm_lambda=0., pcg_tol=1e-12, pcg_max_iters=1000, linesearch=True):
        if self.native is not None:
            return self.dev.gn_iteration(lm_lambda, pcg_tol, pcg_max_iters, linesearch)
        self.dev.linearize(lm_lambda)
        with self._timed('pack_unpack'):
            if hasattr(self.dev, 'shard_pack'):
                self.dev.shard_pack()                         # [upper(S) | g | cost | failure flag]
        with self._timed('allreduce'):
            if self.segments is not None:
                self.segments.run()                           # all-gather of the ranks' band segments + fixed-order sum
            else:
                self.dist.all_reduce(self.dev.reduce_tensor)  # RCCL sum over xGMI, on the solver's stream
        with self._timed('pack_unpack'):
            if hasattr(self.dev, 'shard_unpack'):
                self.dev.shard_unpack()                       # mirrored back into S; a shard's failure reaches every rank
        if hasattr(self.dev, 'shard_tensor'):
            # fully asynchronous second half: the shard's {cost, ||dx_l||^2} are all-reduced on the
            # device; ONE synchronisation per iteration (gn_result)
            first = True
            while True:
                last = self.dev.gn_solve_finish_enqueue(pcg_tol, pcg_max_iters, linesearch, first)
                with self._timed('allreduce'):
                    self.dist.all_reduce(self.dev.shard_tensor)
                done, cost, dxl2, dxp2, its, rel = self.dev.gn_result()
                if done or last:
                    return cost, float(np.sqrt(dxp2 + dxl2)), its, rel
                first = False
        cost, dxp2, dxl2, its, rel = self.dev.gn_solve_finish(pcg_tol, pcg_max_iters, linesearch)
        self._scal.copy_(self._torch.tensor([cost, dxl2], dtype=self._torch.float64), non_blocking=True)
        self.dist.all_reduce(self._scal)
        s = self._scal.tolist()
        return s[0], float(np.sqrt(dxp2 + s[1])), its, rel

    # ---- passthrough ---------------------------------------------------
    def snapshot(self):
        self.dev.snapshot()

    def restore(self):
        self.dev.restore()

    def reset_solver_state(self):
        """Every rank forgets its lagged solver state at the same call (the replicated solve stays identical)."""
        if hasattr(self.dev, 'reset_solver_state'):
            self.dev.reset_solver_state()

    def set_solve_horizon(self, n):
        if hasattr(self.dev, 'set_solve_horizon'):
            self.dev.set_solve_horizon(n)

    def set_expect_next(self, flag):
        """What the caller's loop knows about its next call (pyslam_amd.problem.device_solve): the core's own sharded iteration
        then runs the successor's landmark pass in its tail and sums the cost on its way, as on one GPU (round 6)."""
        if self.native is not None and hasattr(self.dev, 'set_expect_next'):
            self.dev.set_expect_next(flag)

    def get_params(self):
        return self.dev.get_params()

    def set_profiling(self, level=2):
        self._prof_level = level
        self.dev.set_profiling(level)

    def _timed(self, stage):
        """Event pair on the solver's stream around a collective / pack step of the torch.distributed fallback path (the
        native path times the same stages inside the core: ps_get_stage_times 'allreduce' / 'pack_unpack')."""
        outer = self

        class _T:
            def __enter__(self_t):
                self_t.on = getattr(outer, '_prof_level', 0) >= 2 and outer.native is None and outer._torch.cuda.is_available() \
                    and getattr(outer.dev, 'reduce_tensor', None) is not None and outer.dev.reduce_tensor.is_cuda
                if self_t.on:
                    self_t.a = outer._torch.cuda.Event(enable_timing=True); self_t.b = outer._torch.cuda.Event(enable_timing=True)
                    self_t.a.record()

            def __exit__(self_t, *exc):
                if self_t.on:
                    self_t.b.record()
                    outer._pending_events.append((stage, self_t.a, self_t.b))
                return False
        return _T()

    def stage_times(self, reset=False):
        st = self.dev.stage_times(reset)
        if self._pending_events or self._host_stage:
            for stage, a, b in self._pending_events:
                b.synchronize()
                acc = self._host_stage.setdefault(stage, [0., 0])
                acc[0] += a.elapsed_time(b); acc[1] += 1
            self._pending_events = []
            # (several events per iteration add up to one entry per iteration: divide by the iterations the core counted)
            n_it = max([1] + [int(v[1]) for k, v in st.items() if k in ('iteration_total', 'landmark_pass', 'pose_pass')])
            for stage, (ms, n) in self._host_stage.items():
                st[stage] = (ms, n_it)
            if reset:
                self._host_stage = {}
        return st

    def close(self):
        self.dev.close()
        if self.native is not None:
            self.native.close()


def landmark_bounds(lp, world):
    """The contiguous landmark ranges ``order='index'`` cuts (len world + 1) -- rounds 1-4's split, kept for comparison."""
    L = lp.num_points
    if world == 1:
        return [0, L]
    counts = np.bincount(lp.obs_point, minlength=L).astype(np.int64)
    cum = np.concatenate([[0], np.cumsum(counts)])
    return [int(np.searchsorted(cum, cum[-1] * r / world, side='left')) for r in range(world)] + [L]


class ShardedProblemView:
    """What ``Problem.solve()`` drives when ``Options.devices`` routes it to the multi-GPU path: it is handed the
    FULL lowered problem on every rank (every process runs the same script on the same Problem), keeps this rank's
    landmark shard in HBM and presents the interface of DeviceProblem -- parameters in and out are the full tables
    (poses are replicated; landmark rows are gathered from their owners), the iteration is ShardedDeviceProblem's.
    Covariance columns are not sharded: they are solved on a replica of the whole problem on this rank's GPU."""

    def __init__(self, lp, dist, device_factory=None, native_rccl=True):
        self.lp, self.dist = lp, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.owners = landmark_owner_lists(lp, self.world)   # every rank computes every rank's list: no exchange needed
        self.sharded = ShardedDeviceProblem(shard_landmarks(lp, self.rank, self.world, owners=self.owners), dist,
                                            device_factory=device_factory, native_rccl=native_rccl)
        self.dof = lp.dof
        self.info = dict(self.sharded.info)
        self._replica = None

    # ---- parameters: full tables <-> this rank's shard ---------------------
    def set_params(self, poses=None, points=None):
        self.sharded.dev.set_params(poses, None if points is None else np.ascontiguousarray(np.asarray(points)[self.owners[self.rank]]))

    def get_params(self):
        poses, mine = self.sharded.get_params()
        parts = [None] * self.world
        self.dist.all_gather_object(parts, mine)
        points = np.empty((self.lp.num_points, 3))
        for r in range(self.world):                          # every landmark row back to the caller's numbering
            points[self.owners[r]] = np.asarray(parts[r]).reshape(-1, 3)
        return poses, points

    def get_dx(self):
        """(dx_pose, dx_point) of the last iteration in the FULL problem's device order.  After a staged solve
        (solve_reduced) the step gathered there is returned: no second collective."""
        if getattr(self, '_staged_dx', None) is not None:
            return self._staged_dx
        return self._gather_dx()

    def _gather_dx(self):
        xp, xl = self.sharded.dev.get_dx()
        parts = [None] * self.world
        self.dist.all_gather_object(parts, xl)
        full = np.empty((self.lp.num_var_points, 3))
        for r in range(self.world):                          # a shard's variable landmarks keep their relative order: its vid k
            vids = self.lp.point_vid[self.owners[r]]         # is the k-th variable landmark of its (ascending) index list
            full[vids[vids >= 0]] = np.asarray(parts[r]).reshape(-1, 3)
        return xp, full

    def snapshot(self):
        self.sharded.snapshot()

    def restore(self):
        self.sharded.restore()

    def reset_solver_state(self):
        self.sharded.reset_solver_state()

    def set_solve_horizon(self, n):
        self.sharded.set_solve_horizon(n)

    def set_expect_next(self, flag):
        self.sharded.set_expect_next(flag)

    # ---- hot path ----------------------------------------------------------
    def eval_cost(self, include_all_constant=True):
        return self.sharded.eval_cost(include_all_constant)

    def gn_iteration(self, lm_lambda=0., pcg_tol=1e-12, pcg_max_iters=1000, linesearch=True):
        self._staged_dx = None
        return self.sharded.gn_iteration(lm_lambda, pcg_tol, pcg_max_iters, linesearch)

    # ---- staged calls (Problem.solve_one_iter): one sharded iteration, then the parameters go back -----------
    def linearize(self, lm_lambda=0.):
        self._staged_lambda = lm_lambda

    def solve_reduced(self, tol=1e-12, max_iters=1000):
        self.sharded.snapshot()
        self._staged = self.sharded.gn_iteration(self._staged_lambda, tol, max_iters, True)
        self._staged_dx = self._gather_dx()                # ONE gather of the step; get_dx() hands it out
        self.sharded.restore()
        return self._staged[2], self._staged[3]

    def backsub(self):
        pass                                               # (part of the sharded iteration above)

    def apply_update(self, step=1.0):
        """Only the full step of the reference's degenerate line search (problem.py:362-398) is ever applied."""
        self.sharded.dev.apply_update(step)

    # ---- covariance: a replica of the whole problem on this rank's GPU --------------------------------------
    # (memory: the replica holds the UNSHARDED tables -- a problem sharded because it does not fit one GPU cannot have its
    #  covariance computed this way; C4 needs 1.6 GB, far from that limit)
    def covariance_begin(self):
        from pyslam_amd.device import DeviceProblem
        poses, points = self.get_params()
        if self._replica is None:
            self._replica = DeviceProblem(self.lp)
        self._replica.set_params(poses, points)
        self._replica.covariance_begin()

    def covariance_column(self, kind, index, comp, tol=1e-13, max_iters=4000):
        return self._replica.covariance_column(kind, index, comp, tol, max_iters)

    def close(self):
        self.sharded.close()
        if self._replica is not None:
            self._replica.close()
            self._replica = None
