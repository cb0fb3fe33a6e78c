"""The landmark-sharded (multi-GPU) driver on ONE GPU: a world of size 1 runs the same code path as
N ranks -- RCCL all-reduce of [S | g | cost] issued by the HIP core on the solver's stream, replicated
reduced solve, gated shard-local tail, all-reduce of the shard scalars -- and must reproduce the
unsharded iteration bit for bit (a sum over one rank is the identity).  The 2-rank arithmetic is
covered on CPU by tests/test_distributed_cpu.py (gloo)."""
import os

import numpy as np
import pytest

from pyslam_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dist1():
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', RANK='0', WORLD_SIZE='1')
    dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize('native', [True, False])
@pytest.mark.parametrize('explicit', [False, True])
def test_one_rank_sharded_iteration_equals_the_unsharded_one(dist1, native, explicit):
    """explicit: the large-system solver (explicitly applied two-level preconditioner, side-stream coarse inverse)
    forced onto the small problem -- the sharded protocol must drive it too (C4-sized shards use it by default)."""
    import torch
    from pyslam_amd.device import DeviceProblem
    from pyslam_amd.distributed import ShardedDeviceProblem
    lp, _ = synthetic.stereo_ba(num_kf=40, num_lm=4000, obs_per_lm=6, half_window=8, seed=3)
    ref = DeviceProblem(lp, stream=torch.cuda.current_stream().cuda_stream)
    sh = ShardedDeviceProblem(lp, dist1, native_rccl=native)
    assert (sh.native is not None) == native
    # (the sharded iteration never has the lagged dense inverse -- its reduced solve is replicated -- so the unsharded
    #  reference must not use it either if the comparison is to be bit for bit: with a seed usable one call later the
    #  third iteration below would be preconditioned differently)
    ref.set_option('lagged_inverse', 0)
    if not native:       # (round 6) a caller that drives the collectives itself cannot repeat a solve on ONE rank, so it never gets
        ref.set_option('cg_persist', 0); ref.set_option('xcg_persist', 0)      # the one-launch solvers, which may time out: same kernels
    if explicit:
        for d in (ref, sh.dev):
            d.set_option('cg_explicit_min_rows', 0)
            d.set_option('cg_split_min_rows', 0)
    # Bit for bit, both variants.  (Round 3 had to allow the explicit variant two ulps: now and then its side-stream
    # factorisation -- then on a LOW-PRIORITY stream -- returned a different inverse for the same A_c when it ran beside
    # normal-priority kernels; tools/hunt_explicit_flake.py, DESIGN.md section 4.  The side stream is an ordinary stream now.)
    for _ in range(3):
        a = ref.gn_iteration(0., 1e-12, 1000, True)
        b = sh.gn_iteration(0., 1e-12, 1000, True)
        assert a[0] == b[0] and a[2] == b[2]                            # cost and CG iterations: identical
        assert abs(a[1] - b[1]) <= 1e-15 * a[1]                         # ||dx||: summed in a different grouping
    pa, la = ref.get_params()
    pb, lb = sh.get_params()
    assert np.array_equal(pa, pb) and np.array_equal(la, lb)
    assert sh.eval_cost(True) == ref.eval_cost(True)
    sh.close()
    ref.close()


def test_one_rank_segment_exchange_is_the_identity(dist1):
    """exchange='segments' (round 5) with one rank: the element list built on the host (packed_layout: the upper blocks in key
    order, gradient rows, tail words) must cover everything the core's pack kernel leaves non-zero -- gather, all-gather over
    one rank and the scatter-add then reproduce the buffer, and the iterations are the unsharded ones bit for bit."""
    import torch
    from pyslam_amd.device import DeviceProblem
    from pyslam_amd.distributed import ShardedDeviceProblem
    lp, _ = synthetic.stereo_ba(num_kf=40, num_lm=4000, obs_per_lm=6, half_window=8, seed=3)
    ref = DeviceProblem(lp, stream=torch.cuda.current_stream().cuda_stream)
    ref.set_option('lagged_inverse', 0)
    ref.set_option('cg_persist', 0); ref.set_option('xcg_persist', 0)      # (as the torch-driven sharded iteration: see above)
    sh = ShardedDeviceProblem(lp, dist1, exchange='segments', native_rccl=False)       # (the torch-driven exchange)
    assert sh.native is None and sh.segments is not None
    for _ in range(3):
        a = ref.gn_iteration(0., 1e-12, 1000, True)
        b = sh.gn_iteration(0., 1e-12, 1000, True)
        assert a[0] == b[0] and a[2] == b[2]
    pa, la = ref.get_params()
    pb, lb = sh.get_params()
    assert np.array_equal(pa, pb) and np.array_equal(la, lb)
    sh.close()
    ref.close()


def test_one_rank_core_segment_exchange_is_the_identity(dist1):
    """Round 6: exchange='segments' inside the core's own sharded iteration (ps_set_segment_exchange: k_seg_pack, ncclAllGather over
    the native communicator, k_seg_sum in rank order) with one rank: the iterations are the unsharded ones bit for bit, and the
    cost rides on the successor's landmark pass as on one GPU (option expect_next)."""
    import torch
    from pyslam_amd.device import DeviceProblem
    from pyslam_amd.distributed import ShardedDeviceProblem
    lp, _ = synthetic.stereo_ba(num_kf=40, num_lm=4000, obs_per_lm=6, half_window=8, seed=3)
    for expect in (False, True):
        ref = DeviceProblem(lp, stream=torch.cuda.current_stream().cuda_stream)
        ref.set_option('lagged_inverse', 0)
        sh = ShardedDeviceProblem(lp, dist1, exchange='segments')
        assert sh.native is not None and sh.core_segments and sh.segments is None
        for d in (ref, sh):
            d.set_expect_next(expect)
        for _ in range(3):
            a = ref.gn_iteration(0., 1e-12, 1000, True)
            b = sh.gn_iteration(0., 1e-12, 1000, True)
            assert a[0] == b[0] and a[2] == b[2]
        pa, la = ref.get_params()
        pb, lb = sh.get_params()
        assert np.array_equal(pa, pb) and np.array_equal(la, lb)
        if expect:
            assert sh.dev.get_info()['landmark_passes_taken_over'] >= 2          # the cost rode on the next landmark pass on the shard too
        sh.close()
        ref.close()


def _two_rank_worker(rank, world, port, out, native, exchange='allreduce'):
    """One of two processes sharing cuda:0: the REAL device code on a landmark shard, the collectives over
    gloo (RCCL refuses two ranks on one device; the arithmetic of the exchange is the same)."""
    import torch
    import torch.distributed as dist
    from pyslam_amd.distributed import ShardedDeviceProblem, shard_landmarks
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    import datetime
    dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=90))   # (a dead peer fails the
    lp, _ = synthetic.stereo_ba(num_kf=40, num_lm=4000, obs_per_lm=6, half_window=8, seed=3)                 #  collective, not the suite)
    sp = ShardedDeviceProblem(shard_landmarks(lp, rank, world), dist, native_rccl=False, exchange=exchange)
    if exchange == 'segments':
        assert sp.segment_bytes[0] < 0.8 * sp.segment_bytes[1]                  # a band segment, not the whole system
    if native:
        # drive the core's OWN collective path (ps_set_collective: ps_gn_iteration issues both all-reduces
        # itself) with a stand-in for ncclAllReduce that sums over gloo -- same signature, in place
        import ctypes as C
        from pyslam_amd.distributed import _RawDeviceArray

        def all_reduce(send, recv, count, dtype, op, comm, stream):
            dist.all_reduce(torch.as_tensor(_RawDeviceArray(recv, count), device='cuda'))
            return 0
        cb = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p)(all_reduce)
        sp.dev.set_collective(C.cast(cb, C.c_void_p).value, 1)
        sp.native = type('Standin', (), {'close': lambda self: None, 'keep': cb})()
        if exchange == 'segments':
            # ... and the core's own SEGMENT exchange (round 6, ps_set_segment_exchange) with a stand-in for ncclAllGather
            def all_gather(send, recv, count, dtype, comm, stream):
                src = torch.as_tensor(_RawDeviceArray(send, count), device='cuda')
                dst = torch.as_tensor(_RawDeviceArray(recv, count * world), device='cuda')
                dist.all_gather(list(dst.view(world, count).unbind(0)), src)
                return 0
            cg = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p)(all_gather)
            sp.native.keep2 = cg
            sp.enable_core_segments(C.cast(cg, C.c_void_p).value)
            assert sp.core_segments and sp.segments is None
    trace = [sp.eval_cost(True)]
    for _ in range(3):
        trace.append(sp.gn_iteration(0., 1e-12, 1000, True))
    poses, _ = sp.get_params()
    if rank == 0:
        out.put((trace, poses))
    dist.barrier()
    sp.close()
    dist.destroy_process_group()


@pytest.mark.parametrize('native,exchange,world', [(False, 'allreduce', 2), (True, 'allreduce', 2), (False, 'segments', 2), (True, 'segments', 2),
                                                   (True, 'segments', 3)])
def test_two_ranks_on_one_gpu_reproduce_the_unsharded_iterations(native, exchange, world):
    """world_size 2 (3: the core's segment sum with three contributors to a block) with the real HIP core in every rank (one GPU,
    gloo collectives): sharded linearisation, exchange of [S | g | cost], replicated reduced solve, shard-local tail, all-reduce of
    the shard scalars."""
    import socket
    import torch
    import torch.multiprocessing as mp
    from pyslam_amd.device import DeviceProblem
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, world, port, q, native, exchange)) for r in range(world)]
    for p in procs:
        p.start()
    trace, poses = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    lp, _ = synthetic.stereo_ba(num_kf=40, num_lm=4000, obs_per_lm=6, half_window=8, seed=3)
    ref = DeviceProblem(lp, stream=torch.cuda.current_stream().cuda_stream)
    c0 = ref.eval_cost(True)
    assert abs(trace[0] - c0) <= 1e-12 * c0
    for k in range(3):
        cost, nrm, its, rel = ref.gn_iteration(0., 1e-12, 1000, True)
        assert abs(trace[k + 1][0] - cost) <= 1e-10 * cost + 1e-14           # the shard sums group differently
        assert abs(trace[k + 1][1] - nrm) <= 1e-9 * nrm
    assert np.abs(poses - ref.get_params()[0]).max() < 1e-9
    ref.close()


def test_problem_solve_with_devices_option_on_a_one_rank_group(dist1):
    """Options.devices = 'all' routes Problem.solve() / eval_cost() / solve_one_iter() / compute_covariance() through
    ShardedProblemView (landmark shard + native RCCL) -- with one rank the results are those of the single-GPU route."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from conftest import load_golden, golden_lp
    from test_host_api import build_namespace
    ns = build_namespace()
    lp = golden_lp(load_golden('ba_small'))
    out = {}
    for devices in (None, 'all'):
        opts = ns.Options()
        opts.allow_nondecreasing_steps = True
        opts.max_nondecreasing_steps = 3
        opts.devices = devices
        problem = synthetic.to_objects(lp, ns, options=opts)
        c0 = problem.eval_cost()
        dx, c1 = problem.solve_one_iter()
        assert problem.eval_cost() == c0                        # solve_one_iter does not move the parameters
        problem.solve()
        problem.compute_covariance()
        key = [k for k in problem.param_dict if k.startswith('T')][3]
        out[devices] = (c0, dx, c1, list(problem._cost_history), problem._lower(), problem.get_covariance_block(key, key),
                        type(problem._device).__name__)
    a, b = out[None], out['all']
    assert b[6] == 'ShardedProblemView' and a[6] == 'DeviceProblem'
    assert a[0] == b[0] and abs(a[2] - b[2]) <= 1e-12 * a[2]
    assert np.linalg.norm(a[1] - b[1]) <= 1e-9 * np.linalg.norm(a[1])
    assert len(a[3]) == len(b[3]) and np.allclose(a[3], b[3], rtol=1e-9)
    assert np.abs(a[4].poses - b[4].poses).max() < 1e-9 and np.abs(a[4].points - b[4].points).max() < 1e-8
    assert np.abs(a[5] - b[5]).max() <= 1e-9 * np.abs(a[5]).max()
