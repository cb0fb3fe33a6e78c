"""Explicit two-level PCG: banded factorisation / inverse of the coarse matrix against the dense one (option band_chol)."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from pyslam_amd import synthetic
from pyslam_amd.device import DeviceProblem


def run(lp, tag, iters=5):
    res = {}
    for band in (0, 1, 0, 1):
        dev = DeviceProblem(lp)
        dev.set_option('band_chol', band)
        ts, its = [], []
        for it in range(iters):
            t = time.time(); out = dev.gn_iteration(0., 1e-12, 4000, True); ts.append((time.time() - t) * 1e3); its.append(out[2])
        res[band] = dev.get_params()
        print(tag, 'band', band, 'ms', ' '.join('%.3f' % x for x in ts), 'pcg', its, 'cost %.12e' % out[0])
        dev.close()
    d = max(np.abs(a - b).max() for a, b in zip(res[0], res[1]) if a.size)
    print(tag, 'max parameter difference after %d iterations %.2e' % (iters, d))


which = sys.argv[1:] or ['c4', 'c2']
if 'c4' in which:
    run(synthetic.stereo_ba(2000, 500000, 10, 20, seed=1)[0], 'C4')
if 'c2' in which:
    run(synthetic.pose_graph(num_poses=10000, num_loops=40001, dof=6, seed=2)[0], 'C2', 7)
if 'mid' in which:
    run(synthetic.stereo_ba(700, 100000, 10, 20, seed=3)[0], 'BA700')
    run(synthetic.pose_graph(num_poses=1500, num_loops=6001, dof=3, seed=4)[0], 'PG2D-1500')
    run(synthetic.pose_graph(num_poses=1500, num_loops=6001, dof=6, seed=5)[0], 'PG3D-1500')
