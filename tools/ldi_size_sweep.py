"""Lagged dense inverse on / off over bundle-adjustment sizes: eight whole-iteration calls from the perturbed start (a real
solve: both phases) and the settled iteration -- the check that the default helps (or at least does not hurt) at every size
it is eligible for."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.getcwd()))
import numpy as np, torch
from pyslam_amd import synthetic
from pyslam_amd.device import DeviceProblem
for kf, lm in ((16, 1500), (24, 3000), (40, 6000), (64, 12000), (100, 25000), (150, 40000), (200, 50000), (250, 60000), (300, 75000), (340, 85000)):
    lp = synthetic.stereo_ba(num_kf=kf, num_lm=lm, obs_per_lm=8, half_window=12, seed=kf)[0]
    for on in (1, 0):
        dev = DeviceProblem(lp)
        dev.set_option('lagged_inverse', on)
        dev.eval_cost(True); dev.snapshot()
        best = None
        for rep in range(3):
            dev.restore(); torch.cuda.synchronize(); rows = []
            for _ in range(8):
                t0 = time.perf_counter(); out = dev.gn_iteration(0., 1e-12, 2000, True); rows.append(((time.perf_counter() - t0) * 1e3, out[2]))
            tot = sum(r[0] for r in rows)
            if best is None or tot < best[0]: best = (tot, rows)
        i = dev.get_info()
        print('kf %3d lm %5d n %4d lagged_inverse=%d total %.3f ms settled %.3f  ms %s its %s solves %d fallbacks %d seeds %d' % (
            kf, lm, dev.nr * dev.dof, on, best[0], np.mean([r[0] for r in best[1][5:]]), [round(r[0], 3) for r in best[1]], [r[1] for r in best[1]],
            i['ldi_solves'], i['ldi_fallbacks'], i['ldi_seeds']), flush=True)
        dev.close()
