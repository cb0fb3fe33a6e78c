"""Explicit two-level PCG, three launches per iteration against four: C4 (BA, 2000 x 500k) and C2 (10k-pose graph).
Same iterate, both forms: time per iteration, CG iterations, difference of the steps."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from pyslam_amd import synthetic
from pyslam_amd.device import DeviceProblem


def run(dev, lp, tag, warm):
    res = {}
    for rt in (0, 1, 0, 1):
        dev.set_option('xcg_restrict_fused', rt)
        dev.set_params(lp.poses, lp.points)
        dev.snapshot()
        for it in range(warm):
            dev.restore(); dev.set_profiling(2); dev.stage_times(reset=True)
            t = time.time(); out = dev.gn_iteration(0., 1e-12, 4000, True); dt = time.time() - t
            st = {k: round(v[0], 3) for k, v in dev.stage_times(reset=True).items() if v[1]}
        dxp, dxl = dev.get_dx()
        res[rt] = dxp.copy()
        print(tag, 'fused restrict', rt, 'iter %.3f ms' % (dt * 1e3), 'pcg', out[2], 'relres %.1e' % out[3], 'cost %.9e' % out[0], st)
    d = np.linalg.norm(res[0] - res[1]) / np.linalg.norm(res[0])
    print(tag, 'relative difference of the pose steps %.2e' % d)


which = sys.argv[1:] or ['c4', 'c2']
if 'c4' in which:
    lp, _ = synthetic.stereo_ba(2000, 500000, 10, 20, seed=1)
    run(DeviceProblem(lp), lp, 'C4', 4)
if 'c2' in which:
    lp, _ = synthetic.pose_graph(num_poses=10000, num_loops=40001, dof=6, seed=2)
    run(DeviceProblem(lp), lp, 'C2', 5)
if 'mid' in which:
    lp, _ = synthetic.stereo_ba(700, 100000, 10, 20, seed=3)
    run(DeviceProblem(lp), lp, 'BA700', 4)
