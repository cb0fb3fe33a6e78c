"""Hunt for the rare run-to-run difference of the explicit two-level PCG (tests/fuzz_shards.py case 860043, seen once in two
runs): three handles on the same problem -- two unsharded, one one-rank sharded -- iterate side by side; on any difference
the solver counters of all three are printed.   usage: python tools/hunt_explicit_flake.py [rounds] [seed0]   (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from pyslam_amd import synthetic, losses
from pyslam_amd.device import DeviceProblem
from pyslam_amd.distributed import ShardedDeviceProblem

os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29549', RANK='0', WORLD_SIZE='1')
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
KEYS = ('cg_restarts', 'cg_kernel_launches', 'xcg_fused_solves', 'xcg_fused_fallbacks', 'ldi_solves', 'ldi_fallbacks', 'ldi_seeds')
bad = errors = 0
for rnd in range(rounds):
    rng = np.random.default_rng(seed0 + rnd)
    kf = int(rng.choice([40, 90, 90, 150]))
    obs = int(rng.integers(2, 6))
    lp, truth = synthetic.stereo_ba(num_kf=kf, num_lm=int(rng.integers(30 * kf // obs + 8, 60 * kf // obs + 40)), obs_per_lm=obs,
                                half_window=int(rng.integers(obs, 3 * obs + 2)), seed=seed0 + rnd, loss=losses.HuberLoss(1.5))
    if rng.integers(2):
        lp = synthetic.with_pose_edges(lp, int(rng.integers(0, 2 * kf)), rnd + 1, loss=losses.HuberLoss(1.5), truth_poses=truth['poses'])
    s = torch.cuda.current_stream().cuda_stream
    a, b = DeviceProblem(lp, stream=s), DeviceProblem(lp, stream=s)
    sh = ShardedDeviceProblem(lp, dist, native_rccl=True)
    for d in (a, b, sh.dev):
        d.set_option('cg_explicit_min_rows', 0); d.set_option('cg_split_min_rows', 0)
        for kv in os.environ.get('HUNT_OPTS', '').split(','):
            if kv:
                d.set_option(kv.split('=')[0], float(kv.split('=')[1]))
    sync = (lambda: torch.cuda.synchronize()) if os.environ.get('HUNT_SYNC') else (lambda: None)
    for it in range(4):
        try:
            ra = a.gn_iteration(0., 1e-12, 2000, True); sync()
            rb = b.gn_iteration(0., 1e-12, 2000, True); sync()
            rs = sh.gn_iteration(0., 1e-12, 2000, True); sync()
        except Exception as e:          # noqa: BLE001  (a spurious failure of a side-stream factorisation: counted, the hunt goes on)
            errors += 1
            print('ERROR round %d iteration %d: %s' % (rnd, it, str(e)[:90]), flush=True)
            break
        if not (ra == rb and ra[0] == rs[0] and ra[2] == rs[2]):
            bad += 1
            print('DIFF round %d kf %d obs %d edges %d iteration %d' % (rnd, kf, lp.num_obs, lp.num_edges, it))
            for name, d, r in (('a ', a, ra), ('b ', b, rb), ('sh', sh.dev, rs)):
                print('  ', name, r, flush=True)
            break
    sh.close(); a.close(); b.close()
    if rnd % 50 == 0:
        print('round', rnd, 'differences so far', bad, flush=True)
print('rounds', rounds, 'differences', bad, 'errors', errors)
