"""Per-stage GPU time (event pairs) of whole-iteration calls at a repeated point, stereo BA.
    [PS_...=.. PYSLAM_AMD_MEASURE=1] python tools/stage_probe.py [kf lm] [--opt=name:value ...] [--label=text]
Prints one line: label, every stage's mean ms, the iteration's wall ms at the same point."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyslam_amd import synthetic
from pyslam_amd.device import DeviceProblem

args = [a for a in sys.argv[1:] if not a.startswith('--')]
kf, lm = (int(args[0]), int(args[1])) if len(args) >= 2 else (200, 50000)
label = ([a[8:] for a in sys.argv[1:] if a.startswith('--label=')] or [''])[0]
lp, _ = synthetic.stereo_ba(num_kf=kf, num_lm=lm, obs_per_lm=10, half_window=20, seed=0 if kf == 200 else 1)
dev = DeviceProblem(lp, stream=torch.cuda.current_stream().cuda_stream)
for o in [a for a in sys.argv[1:] if a.startswith('--opt=')]:
    k, v = o[6:].split(':'); dev.set_option(k, float(v))
dev.eval_cost(True); dev.snapshot()
for _ in range(3):
    dev.restore(); dev.gn_iteration(0., 1e-12, 2000, True)
dev.set_profiling(2); dev.stage_times(reset=True)
for _ in range(10):
    dev.restore(); out = dev.gn_iteration(0., 1e-12, 2000, True)
st = dev.stage_times(reset=True); dev.set_profiling(0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    dev.restore(); out = dev.gn_iteration(0., 1e-12, 2000, True)
torch.cuda.synchronize(); it_ms = (time.perf_counter() - t0) * 1e3 / 20
print('%s kf %d lm %d: ' % (label, kf, lm) + ' '.join('%s %.4f' % (n, v[0] / max(v[1], 1)) for n, v in st.items() if v[1]) +
      ' | iteration (same point) %.4f ms, pcg %d, cost %.12e' % (it_ms, out[2], out[0]))
dev.close()
