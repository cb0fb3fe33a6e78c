"""BASELINE configuration 1 and neighbours: small pose graphs, ten whole-iteration calls from the perturbed start, with and
without the lagged dense inverse (pose graphs: direct seed on a stream of its own, ps_host_ldi.h)."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.getcwd()))
import torch
from pyslam_amd import synthetic, losses
from pyslam_amd.device import DeviceProblem

CASES = (('C1 SE2 100 poses', dict(num_poses=100, num_loops=150, dof=3, seed=4, loss=losses.HuberLoss(1.0))),
         ('SE2 100 poses dense loops', dict(num_poses=100, num_loops=400, dof=3, seed=5)),
         ('SE3 50 poses', dict(num_poses=50, num_loops=120, dof=6, seed=6)),
         ('SE3 100 poses', dict(num_poses=100, num_loops=300, dof=6, seed=7)),
         ('SE3 200 poses', dict(num_poses=200, num_loops=800, dof=6, seed=2, loss=losses.HuberLoss(1.0))),
         ('SE2 400 poses', dict(num_poses=400, num_loops=1600, dof=3, seed=4, loss=losses.HuberLoss(1.0))))
for name, kw in CASES:
    lp = synthetic.pose_graph(**kw)[0]
    for opts in ('', 'lagged_inverse=0'):
        dev = DeviceProblem(lp)
        for kv in opts.split(','):
            if kv: dev.set_option(kv.split('=')[0], float(kv.split('=')[1]))
        dev.eval_cost(True)
        ms, its = [], []
        for _ in range(10):
            torch.cuda.synchronize(); t0 = time.perf_counter(); out = dev.gn_iteration(0., 1e-12, 2000, True)
            ms.append((time.perf_counter() - t0) * 1e3); its.append(out[2])
        i = dev.get_info()
        print('%-26s %-17s n %4d total %5.2f ms  ms %s its %s  solves %d fallbacks %d seeds %d' % (
            name, opts or '(defaults)', dev.nr * dev.dof, sum(ms), [round(m, 3) for m in ms], its, i['ldi_solves'], i['ldi_fallbacks'], i['ldi_seeds']), flush=True)
        dev.close()
