"""Cold, reference-terminated solves (bench.py: cold_solves) of a stereo BA on one GPU: per-call wall clock, iteration counts.
    python tools/cold_probe.py [kf lm [solves]] [--python-loop] [--pg] [--opt=name:value ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from pyslam_amd import synthetic
from pyslam_amd.device import DeviceProblem
from pyslam_amd.problem import device_solve

args = [a for a in sys.argv[1:] if not a.startswith('--')]
kf, lm = (int(args[0]), int(args[1])) if len(args) >= 2 else (200, 50000)
solves = int(args[2]) if len(args) >= 3 else 6
core_loop = '--python-loop' not in sys.argv
opts = [a for a in sys.argv[1:] if a.startswith('--opt=')]
if '--pg' in sys.argv:       # kf = poses, lm = loop closures, SE(3), Huber (BASELINE configuration 2 at 10000 40001)
    lp, _ = synthetic.pose_graph(num_poses=kf, num_loops=lm, dof=6, seed=2)
else:
    lp, _ = synthetic.stereo_ba(num_kf=kf, num_lm=lm, obs_per_lm=10, half_window=20, seed=0 if kf == 200 else 1)
dev = DeviceProblem(lp, stream=torch.cuda.current_stream().cuda_stream)
for o in opts:
    k, v = o[6:].split(':'); dev.set_option(k, float(v))
if '--prof' in sys.argv:      # as bench.py's timed region: one event pair around the Schur kernel on every 4th linearisation
    dev.set_option('profile_every', 4); dev.set_profiling(1)
if '--stages' in sys.argv:    # an event pair around every stage: per solve, the GPU time of each stage summed over its calls
    dev.set_profiling(2)
start = (lp.poses.copy(), lp.points.copy())
opt = bench.example_options()
tot, its = 0.0, 0
all_ms, all_dt = [], []
for s in range(solves + 1):
    dev.reset_solver_state(); dev.set_params(*start); torch.cuda.synchronize()
    ms = []
    t0 = time.perf_counter()
    hist, stats = device_solve(dev, opt, use_core_loop=core_loop, call_ms=ms)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    if s:
        tot += dt; its += len(ms); all_ms.append(ms); all_dt.append(dt)
    print('solve %d: %.4f ms, calls %s, pcg %s, outside the calls %.4f ms' % (s, dt, ['%.4f' % m for m in ms], [a for a, _ in stats], dt - sum(ms)))
    if '--stages' in sys.argv:
        st = dev.stage_times(reset=True)
        print('   stages (ms summed over the solve, launches): ' + ', '.join('%s %.4f/%d' % (k, v[0], v[1]) for k, v in st.items() if v[1]))
print('kf %d lm %d %s loop: %.4f ms per iteration over %d solves (first excluded); cost history %s' % (
    kf, lm, 'core' if core_loop else 'python', tot / its, solves, ['%.6e' % c for c in hist]))
n_calls = min(len(m) for m in all_ms)
print('%s %d %d%s %s loop, median over %d solves: solve %.4f ms = %.4f ms per iteration, calls %s, pcg %s' % (
    'graph' if '--pg' in sys.argv else 'BA', kf, lm, (' [' + ' '.join(o[6:] for o in opts) + ']') if opts else '', 'core' if core_loop else 'python',
    solves, float(np.median(all_dt)), float(np.median(all_dt)) / n_calls, ['%.4f' % float(np.median([m[k] for m in all_ms])) for k in range(n_calls)],
    [a for a, _ in stats]))
dev.close()
