"""C3 on the host: the reference-faithful linear solve (scipy spsolve of the full normal equations, reference
problem.py:186) next to the CPU Schur complement the bench's cpu_baseline uses.  Minutes of CPU time (SuperLU fill)."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
import scipy.sparse.linalg as spla
from oracle import gn_oracle as orc
from pyslam_amd import synthetic
lp, _ = synthetic.stereo_ba(num_kf=200, num_lm=50000, obs_per_lm=10, half_window=20, seed=0)
for pf in (True, False):
    t = time.time(); P, b, cost = orc.normal_equations(lp, points_first=pf); t1 = time.time() - t
    t = time.time(); dx = spla.spsolve(P, b); t2 = time.time() - t
    t = time.time(); dx2 = orc.schur_solve(lp, P, b, points_first=pf); t3 = time.time() - t
    print('points_first', pf, 'normal equations %.2f s, spsolve (reference-faithful) %.2f s, CPU Schur %.2f s, |diff|/|dx| %.1e' % (
        t1, t2, t3, np.linalg.norm(dx - dx2) / np.linalg.norm(dx)), flush=True)
