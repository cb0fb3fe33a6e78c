
# (measurement build: the PS_* switches / ablation options used here exist only in lib/libpyslam_hip_measure.so)
import os as _os, sys as _sys
_os.environ.setdefault('PYSLAM_AMD_MEASURE', '1')
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import __graft_entry__ as _ge
if not _os.path.exists(_ge.OUT_MEASURE) or _os.path.getmtime(_ge.OUT_MEASURE) < _os.path.getmtime(_ge.SRC):
    _ge.build_measure()
import os, sys, time, csv
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from pyslam_amd import synthetic
from pyslam_amd.device import DeviceProblem
lp,_ = synthetic.stereo_ba(200, 50000, 10, 20, seed=0)
dev = DeviceProblem(lp); dev.snapshot()
for ab in (0, 1, 2):
    dev.set_option('cg_ablate', ab)
    for it in range(6):
        dev.restore(); dev.linearize(0.)
        try:
            dev.solve_reduced(1e-12, 60)
        except Exception as e:
            pass
