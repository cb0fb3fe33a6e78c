"""C3: cost of one CG iteration (slope of the solve time between a loose and a tight tolerance) per CG variant."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyslam_amd import synthetic
from pyslam_amd.device import DeviceProblem
lp, _ = synthetic.stereo_ba(num_kf=200, num_lm=50000, obs_per_lm=10, half_window=20, seed=0)
for lds in (1.0, 0.0):
    dev = DeviceProblem(lp, stream=torch.cuda.current_stream().cuda_stream)
    dev.set_option('cg_lds', lds)
    dev.linearize(0.0)
    res = {}
    for tol in (1e-3, 1e-13):
        for _ in range(3):
            dev.solve_reduced(tol, 1000)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            out = dev.solve_reduced(tol, 1000)
        torch.cuda.synchronize(); res[tol] = ((time.perf_counter() - t0) / 20, out[0])
    (ta, na), (tb, nb) = res[1e-3], res[1e-13]
    print('cg_lds', lds, 'solve(%d its) %.1f us  solve(%d its) %.1f us  per iteration %.2f us' % (
        na, ta * 1e6, nb, tb * 1e6, (tb - ta) / (nb - na) * 1e6))
