import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, torch.distributed as dist
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29521', RANK='0', WORLD_SIZE='1')
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
for n in (2, 527_000, 5_782_040):
    t = torch.zeros(n, dtype=torch.float64, device='cuda')
    for _ in range(5): dist.all_reduce(t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): dist.all_reduce(t)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    print('all_reduce of %d doubles (%.2f MB): %.1f us' % (n, n * 8 / 1e6, dt * 1e6))
dist.destroy_process_group()
