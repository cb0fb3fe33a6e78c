import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, torch.distributed as dist, numpy as np
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29519', RANK='0', WORLD_SIZE='1')
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
from pyslam_amd import synthetic
from pyslam_amd.distributed import ShardedDeviceProblem
lp,_ = synthetic.stereo_ba(200, 50000, 10, 20, seed=0)
sp = ShardedDeviceProblem(lp, dist); sp.snapshot()
dev = sp.dev
T = {k:0.0 for k in ('restore','linearize','allreduce','solve_finish','scal_h2d','scal_allreduce','scal_read')}
for it in range(25):
    t=[time.perf_counter()]
    dev.restore(); t.append(time.perf_counter())
    dev.linearize(0.); t.append(time.perf_counter())
    dist.all_reduce(dev.reduce_tensor); t.append(time.perf_counter())
    cost, a, b, its, rel = dev.gn_solve_finish(1e-12, 1000, True); t.append(time.perf_counter())
    sp._scal.copy_(torch.tensor([cost, b], dtype=torch.float64), non_blocking=True); t.append(time.perf_counter())
    dist.all_reduce(sp._scal); t.append(time.perf_counter())
    s = sp._scal.tolist(); t.append(time.perf_counter())
    if it >= 5:
        for k,d in zip(T, np.diff(t)): T[k]+=d
print({k: round(v/20*1e3,4) for k,v in T.items()}, 'ms', 'sum', round(sum(T.values())/20*1e3,4))
