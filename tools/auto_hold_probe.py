"""coarse_auto_hold on / off: per-iteration time and CG iterations of real solves (C2 pose graph, C4 BA), final parameters."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from pyslam_amd import synthetic
from pyslam_amd.device import DeviceProblem


def run(lp, tag, iters):
    res = {}
    for hold in (0, 1, 0, 1):
        dev = DeviceProblem(lp)
        dev.set_option('coarse_auto_hold', hold)
        ts, its = [], []
        for it in range(iters):
            t = time.time(); out = dev.gn_iteration(0., 1e-12, 4000, True); ts.append((time.time() - t) * 1e3); its.append(out[2])
        res[hold] = dev.get_params()
        print(tag, 'auto hold', hold, 'ms', ' '.join('%.2f' % x for x in ts), 'pcg', its, 'cost %.12e' % out[0])
        dev.close()
    print(tag, 'max parameter difference %.2e' % max(np.abs(a - b).max() for a, b in zip(res[0], res[1]) if a.size))


which = sys.argv[1:] or ['c2', 'c4']
if 'c2' in which:
    run(synthetic.pose_graph(num_poses=10000, num_loops=40001, dof=6, seed=2)[0], 'C2', 10)
if 'c4' in which:
    run(synthetic.stereo_ba(2000, 500000, 10, 20, seed=1)[0], 'C4', 8)
