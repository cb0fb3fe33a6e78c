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
def test_one_rank_sharded_iteration_equals_the_unsharded_one(dist1, native):
    import torch
    from pyslam_amd.device import DeviceProblem
    from pyslam_amd.distributed import ShardedDeviceProblem
    lp, _ = synthetic.stereo_ba(num_kf=40, num_lm=4000, obs_per_lm=6, half_window=8, seed=3)
    ref = DeviceProblem(lp, stream=torch.cuda.current_stream().cuda_stream)
    sh = ShardedDeviceProblem(lp, dist1, native_rccl=native)
    assert (sh.native is not None) == native
    for _ in range(3):
        a = ref.gn_iteration(0., 1e-12, 1000, True)
        b = sh.gn_iteration(0., 1e-12, 1000, True)
        assert a[0] == b[0] and a[2] == b[2]                    # cost and CG iterations: identical
        assert abs(a[1] - b[1]) <= 1e-15 * a[1]                 # ||dx||: summed in a different grouping
    pa, la = ref.get_params()
    pb, lb = sh.get_params()
    assert np.array_equal(pa, pb) and np.array_equal(la, lb)
    assert abs(sh.eval_cost(True) - ref.eval_cost(True)) == 0.
    sh.close()
    ref.close()
