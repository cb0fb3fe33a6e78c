import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from pyslam_amd import synthetic
from pyslam_amd.device import DeviceProblem
from pyslam_amd.distributed import ShardedDeviceProblem
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', RANK='0', WORLD_SIZE='1')
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
fused = int(sys.argv[1]) if len(sys.argv) > 1 else 1
lp, _ = synthetic.stereo_ba(num_kf=40, num_lm=4000, obs_per_lm=6, half_window=8, seed=3)
bad = 0
for rep in range(30):
    junk = torch.full((64 * 1024 * 1024,), float('nan'), dtype=torch.float64, device='cuda'); torch.cuda.synchronize(); del junk; torch.cuda.empty_cache()
    lp2, _ = synthetic.stereo_ba(num_kf=60, num_lm=3000, obs_per_lm=5, half_window=6, seed=rep)
    d2 = DeviceProblem(lp2); [d2.gn_iteration(0., 1e-12, 500, True) for _ in range(4)]; d2.close()
    ref = DeviceProblem(lp, stream=torch.cuda.current_stream().cuda_stream)
    sh = ShardedDeviceProblem(lp, dist, native_rccl=True)
    for d in (ref, sh.dev):
        d.set_option('cg_explicit_min_rows', 0); d.set_option('cg_split_min_rows', 0); d.set_option('xcg_fused', fused)
    for it in range(3):
        a = ref.gn_iteration(0., 1e-12, 1000, True); b = sh.gn_iteration(0., 1e-12, 1000, True)
        if a[0] != b[0] or a[2] != b[2]:
            bad += 1; print('rep', rep, 'it', it, a, b); break
    sh.close(); ref.close()
print('fused', fused, 'mismatches', bad, 'of 30')
dist.destroy_process_group()
