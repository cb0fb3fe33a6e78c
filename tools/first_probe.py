import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from pyslam_amd import synthetic
from pyslam_amd.device import DeviceProblem
lp, _ = synthetic.pose_graph(num_poses=10000, num_loops=40001, dof=6, seed=2)
torch.cuda.synchronize()
t = time.time(); dev = DeviceProblem(lp); torch.cuda.synchronize(); print('create %.1f ms' % ((time.time() - t) * 1e3))
for rep in range(2):
    t = time.time(); dev.linearize(0.); torch.cuda.synchronize(); print('linearize %.2f ms' % ((time.time() - t) * 1e3))
    t = time.time(); r = dev.solve_reduced(1e-12, 4000); torch.cuda.synchronize(); print('solve %.2f ms' % ((time.time() - t) * 1e3), r)
    t = time.time(); dev.backsub(); torch.cuda.synchronize(); print('backsub %.2f ms' % ((time.time() - t) * 1e3))
