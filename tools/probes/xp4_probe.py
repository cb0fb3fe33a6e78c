"""k_xcg_persist4 against k_xcg_persist (option xcg_persist4): same iteration counts, costs and parameters bit for bit; time per call.
    python tools/probes/xp4_probe.py [kf lm]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pyslam_amd import synthetic
from pyslam_amd.device import DeviceProblem

kf, lm = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1000, 60000)
lp, _ = synthetic.stereo_ba(num_kf=kf, num_lm=lm, obs_per_lm=10, half_window=20, seed=3)
out = {}
for four in (1, 0):
    dev = DeviceProblem(lp)
    dev.set_option('xcg_persist4', four)
    res = []
    for rep in range(3):
        dev.reset_solver_state(); dev.set_params(lp.poses.copy(), lp.points.copy())
        t0 = time.perf_counter()
        tr = [dev.gn_iteration(0.0, 1e-12, 2000, True) for _ in range(4)]
        res.append((time.perf_counter() - t0) * 1e3 / 4)
    info = dev.get_info()
    out[four] = (tr, dev.get_params(), info)
    print('xcg_persist4', four, 'ms per call', ['%.4f' % r for r in res], 'trace', [(float(t[0]), int(t[2])) for t in tr],
          'persist solves', info['cg_persist_solves'], 'four-wave', info['xcg_persist4_solves'], 'failures', info['cg_persist_failures'])
    dev.close()
a, b = out[1], out[0]
print('same costs and iteration counts:', all(x[0] == y[0] and x[2] == y[2] for x, y in zip(a[0], b[0])),
      '| poses bit-identical:', np.array_equal(a[1][0], b[1][0]), '| points bit-identical:', np.array_equal(a[1][1], b[1][1]))
