"""Latency of one GN iteration on the small configurations (C5 motion-only, C1-like pose graphs, small BA)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyslam_amd import synthetic
from pyslam_amd.device import DeviceProblem
cases = [('C5 motion-only 256 pts', synthetic.motion_only(num_pts=256, seed=3)[0]),
         ('C5 motion-only 512 pts', synthetic.motion_only(num_pts=512, seed=3)[0]),
         ('C5 motion-only 1024 pts', synthetic.motion_only(num_pts=1024, seed=3)[0]),
         ('C5 motion-only 2048 pts', synthetic.motion_only(num_pts=2048, seed=3)[0]),
         ('pose graph 6 poses (C1 shape, SE3)', synthetic.pose_graph(num_poses=6, num_loops=2, dof=6, seed=1)[0]),
         ('pose graph 200 poses', synthetic.pose_graph(num_poses=200, num_loops=801, dof=6, seed=2)[0]),
         ('BA 20 kf x 400 lm', synthetic.stereo_ba(20, 400, 5, 6, seed=0)[0]),
         # reduced systems of 30 / 54 / 90 unknowns: the one-launch direct solve (k_direct_solve)
         ('BA 6 kf x 300 lm', synthetic.stereo_ba(num_kf=6, num_lm=300, obs_per_lm=5, half_window=12, seed=6)[0]),
         ('BA 10 kf x 800 lm', synthetic.stereo_ba(num_kf=10, num_lm=800, obs_per_lm=8, half_window=12, seed=10)[0]),
         ('BA 16 kf x 1500 lm', synthetic.stereo_ba(num_kf=16, num_lm=1500, obs_per_lm=8, half_window=12, seed=16)[0])]
FUSED = float(sys.argv[1]) if len(sys.argv) > 1 else 1.
for name, lp in cases:
    dev = DeviceProblem(lp, stream=torch.cuda.current_stream().cuda_stream)
    dev.set_option('fused_motion_only', FUSED)
    dev.snapshot()
    for _ in range(5):
        dev.restore(); out = dev.gn_iteration(0.0, 1e-12, 1000, True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100):
        dev.restore(); out = dev.gn_iteration(0.0, 1e-12, 1000, True)
    torch.cuda.synchronize()
    print('%-36s %.1f us / iteration (incl. restore)  cg iters %d  reduced poses %d' % (name, (time.perf_counter() - t0) / 100 * 1e6, out[2], dev.nr))
    dev.close()
