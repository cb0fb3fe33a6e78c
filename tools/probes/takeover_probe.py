"""How many linearisations of a cold reference-terminated solve find their landmark pass done (ps_problem_info.landmark_passes_taken_over):
the start cost and every tail that expects a successor run the NEXT point's landmark pass and sum the cost on the way.
    python tools/probes/takeover_probe.py [kf lm]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from pyslam_amd import synthetic
from pyslam_amd.device import DeviceProblem
from pyslam_amd.problem import device_solve

kf, lm = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (200, 50000)
lp, _ = synthetic.stereo_ba(num_kf=kf, num_lm=lm, obs_per_lm=10, half_window=20, seed=0 if kf == 200 else 1)
dev = DeviceProblem(lp, stream=torch.cuda.current_stream().cuda_stream)
opt = bench.example_options()
start = (lp.poses.copy(), lp.points.copy())
prev = 0
for s in range(3):
    dev.reset_solver_state(); dev.set_params(*start)
    ms = []
    hist, stats = device_solve(dev, opt, use_core_loop=True, call_ms=ms)
    info = dev.get_info()
    print('solve', s, 'iterations', len(ms), 'taken over', info['landmark_passes_taken_over'] - prev, 'cost history', ['%.6e' % c for c in hist])
    prev = info['landmark_passes_taken_over']
dev.close()
