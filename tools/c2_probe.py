"""C2 probe: SE(3) pose graph, 10k poses / 50k edges, Huber, prior on pose 0."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from pyslam_amd import synthetic
from pyslam_amd.device import DeviceProblem
P = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
t = time.time(); lp, _ = synthetic.pose_graph(num_poses=P, num_loops=4 * P + 1, dof=6, seed=2); print('generate %.1fs' % (time.time() - t))
t = time.time(); dev = DeviceProblem(lp); print('create %.2fs' % (time.time() - t), dev.info)
for G in ([int(a) for a in sys.argv[2:]] or [-1, 0]):
    dev.set_option('coarse_groups', G)
    if 'C2_SPLIT_MIN' in os.environ: dev.set_option('cg_split_min_rows', float(os.environ['C2_SPLIT_MIN']))
    if 'C2_EXPLICIT' in os.environ: dev.set_option('cg_explicit', float(os.environ['C2_EXPLICIT']))
    if 'C2_REFRESH' in os.environ: dev.set_option('coarse_refresh_every', int(os.environ['C2_REFRESH']))
    dev.set_params(lp.poses, lp.points)
    hist = [dev.eval_cost(True)]
    for it in range(int(os.environ.get('C2_ITERS', '6'))):
        dev.set_profiling(2); dev.stage_times(reset=True)
        t = time.time(); out = dev.gn_iteration(0., 1e-12, int(os.environ.get("C2_MAXIT", "4000")), True); dt = time.time() - t
        st = {k: round(v[0], 3) for k, v in dev.stage_times(reset=True).items() if v[1]}
        hist.append(out[0])
        print('G', G, 'it', it, 'cost %.6e' % out[0], '%.3f ms' % (dt * 1e3), 'pcg', out[2], 'relres %.1e' % out[3], st)
