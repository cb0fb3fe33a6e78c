"""C4-scale probe: 2000 keyframes x 500k landmarks x 10 obs (5 M blocks) on ONE GPU."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from pyslam_amd import synthetic
from pyslam_amd.device import DeviceProblem
kf, lm = int(sys.argv[1]) if len(sys.argv) > 1 else 2000, int(sys.argv[2]) if len(sys.argv) > 2 else 500000
t = time.time(); lp, _ = synthetic.stereo_ba(kf, lm, 10, 20, seed=1); print('generate %.1fs' % (time.time() - t))
t = time.time(); dev = DeviceProblem(lp); print('create %.1fs' % (time.time() - t), dev.info)
for G in ([int(a) for a in sys.argv[3:]] or [-1]):
    dev.set_option('coarse_groups', G)
    dev.set_option('coarse_lag', float(os.environ.get('C4_LAG', '1')))
    if 'C4_SPLIT_MIN' in os.environ: dev.set_option('cg_split_min_rows', float(os.environ['C4_SPLIT_MIN']))
    dev.set_option('cg_explicit', float(os.environ.get('C4_EXPLICIT', '1')))      # 0: folded split mode
    dev.set_params(lp.poses, lp.points)
    dev.snapshot()
    c0 = dev.eval_cost(True)
    for it in range(4):
        dev.restore(); dev.set_profiling(2); dev.stage_times(reset=True)
        t = time.time(); out = dev.gn_iteration(0., 1e-12, 3000, True); dt = time.time() - t
        st = {k: round(v[0], 3) for k, v in dev.stage_times(reset=True).items() if v[1]}
    print('G', G, 'cost %.6e -> %.6e' % (c0, out[0]), 'iter %.3f ms' % (dt * 1e3), 'pcg', out[2], 'relres %.1e' % out[3], st)
