"""The in-launch exchange of the one-launch solves under UNEVEN load (the HIP guide's advice for every hand-off): cold solves while a
second stream of the same process keeps the memory system busy (large device-to-device copies and a reduction, enqueued in bursts) --
results must be the bit-identical ones of the idle chip, and a time-out (answered by the launch-per-iteration kernels, same result to
the solver's tolerance) must stay the exception.   python tools/probes/persist_under_load.py"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from pyslam_amd import synthetic
from pyslam_amd.device import DeviceProblem
from pyslam_amd.problem import device_solve

stop = False
def load():
    s = torch.cuda.Stream()
    a = torch.empty(256 << 20, dtype=torch.uint8, device='cuda'); b = torch.empty_like(a)
    c = torch.randn(64 << 20, device='cuda')
    with torch.cuda.stream(s):
        while not stop:
            for _ in range(4):
                b.copy_(a); a.copy_(b)
            c.sum()
            s.synchronize()
            time.sleep(0.0005)          # bursts: the chip alternates between loaded and idle

for name, kw, reps in (('C3 200 x 50 000', dict(num_kf=200, num_lm=50000, obs_per_lm=10, half_window=20, seed=0), 60),
                       ('BA 1 000 x 60 000', dict(num_kf=1000, num_lm=60000, obs_per_lm=10, half_window=20, seed=1), 30),
                       ('C4 2 000 x 500 000', dict(num_kf=2000, num_lm=500000, obs_per_lm=10, half_window=20, seed=1), 10)):
    lp, _ = synthetic.stereo_ba(**kw)
    start = (lp.poses.copy(), lp.points.copy())
    dev = DeviceProblem(lp, stream=torch.cuda.current_stream().cuda_stream)
    def solve():
        dev.reset_solver_state(); dev.set_params(*start)
        hist, stats = device_solve(dev, bench.example_options())
        p = dev.get_params()
        return hist, [s[0] for s in stats], p[0].tobytes(), p[1].tobytes()
    ref = solve()
    stop = False
    th = threading.Thread(target=load); th.start()
    time.sleep(0.05)
    same, close, t0 = 0, 0, time.perf_counter()
    for r in range(reps):
        got = solve()
        same += got == ref
        close += len(got[0]) == len(ref[0]) and np.allclose(got[0], ref[0], rtol=1e-9)
    dt = time.perf_counter() - t0
    stop = True; th.join()
    print('%-20s %d solves beside the load (%.1f ms each): %d bit-identical to the idle solve, %d with the same history to 1e-9; one-launch solves / time-outs: %s'
          % (name, reps, dt / reps * 1e3, same, close, dev.cg_persist_counts()))
    dev.close()
