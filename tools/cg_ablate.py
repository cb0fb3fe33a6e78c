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
