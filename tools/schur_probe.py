"""Schur stage (pair kernel + combine) of the pose-stationary kernel against the gather kernels on the same handle.
    PS_SCHUR_MODE=2 python tools/schur_probe.py [kf lm]"""
import os, sys, time
os.environ.setdefault('PS_SCHUR_MODE', '2')
if os.environ.get('ABLATE'):
    os.environ.setdefault('PYSLAM_AMD_MEASURE', '1')      # the ablation options exist in the measurement build only
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
if os.environ.get('PYSLAM_AMD_MEASURE') == '1':
    import __graft_entry__ as _ge
    if not os.path.exists(_ge.OUT_MEASURE) or os.path.getmtime(_ge.OUT_MEASURE) < os.path.getmtime(_ge.SRC):
        _ge.build_measure()
from pyslam_amd import synthetic
from pyslam_amd.device import DeviceProblem
kf, lm = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) >= 3 else (200, 50000)
lp, _ = synthetic.stereo_ba(num_kf=kf, num_lm=lm, obs_per_lm=10, half_window=20, seed=0 if kf == 200 else 1)
t = time.perf_counter(); dev = DeviceProblem(lp); torch.cuda.synchronize()
print('create (both list sets) %.1f ms' % ((time.perf_counter() - t) * 1e3))
dev.eval_cost(True); dev.snapshot()
res = {}
for mode in (1, 0, 1, 0):
    dev.set_option('schur_mode', mode)
    for _ in range(3):
        dev.restore(); dev.gn_iteration(0., 1e-12, 2000, True)
    dev.set_profiling(2); dev.stage_times(reset=True)
    for _ in range(10):
        dev.restore(); dev.gn_iteration(0., 1e-12, 2000, True)
    st = dev.stage_times(reset=True); dev.set_profiling(0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        dev.restore(); out = dev.gn_iteration(0., 1e-12, 2000, True)
    torch.cuda.synchronize(); it_ms = (time.perf_counter() - t0) * 1e3 / 20
    print('schur_mode %d (%s): Schur stage %.4f ms, iteration (same point) %.4f ms, pcg its %d, cost %.9e' % (
        mode, 'pose-stationary' if mode else 'gather', st['schur_pairs'][0] / max(st['schur_pairs'][1], 1), it_ms, out[2], out[0]))
if os.environ.get('ABLATE'):
    dev.set_option('schur_mode', 1)
    for ab in (0, 1, 2, 3, 4, 8, 15):
        dev.set_option('schur_ablate', ab)
        dev.set_profiling(2); dev.stage_times(reset=True)
        for _ in range(10):
            dev.restore()
            try:
                dev.gn_iteration(0., 1e-12, 50, True)
            except Exception:
                pass
        st = dev.stage_times(reset=True); dev.set_profiling(0)
        print('pose-stationary, ablate %2d (1 no products, 2 no partner fetch, 4 no row fill, 8 no partial stores): Schur stage %.4f ms' % (ab, st['schur_pairs'][0] / max(st['schur_pairs'][1], 1)))
dev.close()
