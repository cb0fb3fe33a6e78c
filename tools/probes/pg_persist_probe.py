import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from pyslam_amd import synthetic
from pyslam_amd.device import DeviceProblem
for poses, loops, dof in ((100, 60, 3), (400, 300, 3), (200, 150, 6), (330, 300, 3)):
    lp, _ = synthetic.pose_graph(num_poses=poses, num_loops=loops, dof=dof, seed=2)
    for persist in (1, 0):
        dev = DeviceProblem(lp, stream=torch.cuda.current_stream().cuda_stream)
        dev.set_option('cg_persist', persist); dev.set_option('lagged_inverse', 0); dev.set_option('cg_explicit_min_rows', 100000)
        ts, its = [], []
        for k in range(6):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = dev.gn_iteration(0., 1e-12, 4000, True)
            ts.append((time.perf_counter() - t0) * 1e3); its.append(out[2])
        print('dof %d poses %d persist %d: ms %s its %s counts %s' % (dof, poses, persist, ['%.3f' % t for t in ts], its, dev.cg_persist_counts()))
        dev.close()
