"""Shard-local stages of ONE rank of N on the C4 problem, on one GPU: the rank's landmark shard (by first observing pose, or
by the caller's index as in rounds 1-4) with the block pattern of all N ranks, `ps_linearize` (landmark pass, pose pass, Schur
pair + combine) timed with event pairs, the pack kernel, and what the rank would send in either exchange.
(The reduced solve is not run: a locality shard alone leaves the poses outside its segment without a diagonal block -- the
other ranks' contributions fill them in a real run.)
    python tools/shard_stage_probe.py [N [rank ...]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyslam_amd import synthetic
from pyslam_amd.device import DeviceProblem
from pyslam_amd.distributed import landmark_owner_lists, shard_landmarks, pose_pair_keys, shard_touch, packed_layout

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ranks = [int(a) for a in sys.argv[2:]] or [0, N // 2]
kf, lm = (int(os.environ.get('KF', 2000)), int(os.environ.get('LM', 500000)))
lp_full, _ = synthetic.stereo_ba(num_kf=kf, num_lm=lm, obs_per_lm=10, half_window=20, seed=1 if kf != 200 else 0)
union = pose_pair_keys(lp_full)
indices, tail = packed_layout(union, lp_full.num_reduced, lp_full.dof)
full_bytes = 8 * (int(tail[-1]) + 1)
for order in ('first_pose', 'index'):
    owners = landmark_owner_lists(lp_full, N, order)
    for r in ranks:
        lp = shard_landmarks(lp_full, r, N, owners=owners)
        mine = pose_pair_keys(lp)
        extra = np.setdiff1d(union, mine)
        dev = DeviceProblem(lp, stream=torch.cuda.current_stream().cuda_stream,
                            extra_pairs=((extra >> 32).astype(np.int32), (extra & 0xFFFFFFFF).astype(np.int32)))
        blocks, poses = shard_touch(lp)
        seg_bytes = 8 * (indices(blocks, poses).size + tail.size)
        for _ in range(3):
            dev.linearize(0.0)
        torch.cuda.synchronize()
        dev.set_profiling(2); dev.stage_times(reset=True)
        for _ in range(10):
            dev.linearize(0.0)
        torch.cuda.synchronize()
        st = dev.stage_times(reset=True); dev.set_profiling(0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dev.shard_pack(); torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            dev.shard_pack()
        e1.record(); torch.cuda.synchronize(); pack_ms = e0.elapsed_time(e1) / 20
        print('N %d rank %d order %-10s: obs %7d, blocks touched %6d of %6d, ' % (N, r, order, lp.num_obs, blocks.size, union.size + lp_full.num_reduced) +
              ' '.join('%s %.4f' % (k, v[0] / max(v[1], 1)) for k, v in st.items() if v[1] and k in ('landmark_pass', 'pose_pass', 'schur_pairs')) +
              ' pack %.4f ms | exchange: all-reduce buffer %.2f MB, this rank\'s segment %.2f MB' % (pack_ms, full_bytes / 1e6, seg_bytes / 1e6))
        dev.close()
