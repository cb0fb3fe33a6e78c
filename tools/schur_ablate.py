"""Ablation timing of k_schur_pairs at C3 (bits: 1 = no LDS reads / FMAs, 2 = no Z fetch, 4 = all
rows from the first 64 Z rows).  Results are wrong under ablation; timing only."""
# (measurement build: the PS_* switches / ablation options used here exist only in lib/libpyslam_hip_measure.so)
import os as _os, sys as _sys
_os.environ.setdefault('PYSLAM_AMD_MEASURE', '1')
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import __graft_entry__ as _ge
if not _os.path.exists(_ge.OUT_MEASURE) or _os.path.getmtime(_ge.OUT_MEASURE) < _os.path.getmtime(_ge.SRC):
    _ge.build_measure()
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyslam_amd import synthetic
from pyslam_amd.device import DeviceProblem
lp, _ = synthetic.stereo_ba(num_kf=200, num_lm=50000, obs_per_lm=10, half_window=20, seed=0)
dev = DeviceProblem(lp, stream=torch.cuda.current_stream().cuda_stream)
dev.snapshot()
if os.environ.get('SCHUR_PIPELINE') is not None:
    dev.set_option('schur_pipeline', int(os.environ['SCHUR_PIPELINE']))
import sys
OPT = sys.argv[1] if len(sys.argv) > 1 else 'schur_ablate'
for ab in ([0, 1, 2, 3, 4, 5] if OPT == 'schur_ablate' else [0, 1]):
    dev.set_option(OPT, ab)
    for _ in range(3):
        dev.restore(); dev.linearize(0.0)
    dev.set_profiling(2); dev.stage_times(reset=True)
    for _ in range(20):
        dev.restore(); dev.linearize(0.0)
    torch.cuda.synchronize()
    st = dev.stage_times(reset=True)
    dev.set_profiling(0)
    print('ablate', ab, {k: round(v[0] / max(v[1], 1), 4) for k, v in st.items() if v[1] > 0})
