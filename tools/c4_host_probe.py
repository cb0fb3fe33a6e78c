"""C4 on one GPU: where the wall time of a step goes (restore, enqueue, wait) -- run with PS_HOST_TIMING=1."""
# (measurement build: the PS_* switches / ablation options used here exist only in lib/libpyslam_hip_measure.so)
import os as _os, sys as _sys
_os.environ.setdefault('PYSLAM_AMD_MEASURE', '1')
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import __graft_entry__ as _ge
if not _os.path.exists(_ge.OUT_MEASURE) or _os.path.getmtime(_ge.OUT_MEASURE) < _os.path.getmtime(_ge.SRC):
    _ge.build_measure()
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from pyslam_amd import synthetic
from pyslam_amd.device import DeviceProblem
lp, _ = synthetic.stereo_ba(2000, 500000, 10, 20, seed=1)
dev = DeviceProblem(lp, stream=torch.cuda.current_stream().cuda_stream)
if 'C4_REFRESH' in os.environ: dev.set_option('coarse_refresh_every', int(os.environ['C4_REFRESH']))
dev.snapshot()
for _ in range(6):
    dev.restore(); dev.gn_iteration(0., 1e-12, 3000, True)
torch.cuda.synchronize()
tr = tg = 0.
N = 20
t00 = time.perf_counter()
for _ in range(N):
    t0 = time.perf_counter(); dev.restore(); t1 = time.perf_counter()
    out = dev.gn_iteration(0., 1e-12, 3000, True); t2 = time.perf_counter()
    tr += t1 - t0; tg += t2 - t1
torch.cuda.synchronize()
tot = time.perf_counter() - t00
print('per step: restore call %.1f us, gn_iteration call %.1f us, loop wall %.1f us, pcg iters %d' % (tr / N * 1e6, tg / N * 1e6, tot / N * 1e6, out[2]))
dev.set_profiling(2)
for _ in range(3):
    dev.restore(); dev.gn_iteration(0., 1e-12, 3000, True)
print({k: round(v[0] / v[1], 4) for k, v in dev.stage_times(reset=True).items() if v[1] > 0})
dev.close()
