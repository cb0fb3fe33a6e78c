mkdir -p gpurun_out/r02c
(time python -m pytest tests -m gpu -x -q 2>&1 | tail -30) > gpurun_out/r02c/tests.log 2>&1
python bench.py --no-cpu-baseline --no-c4 > gpurun_out/r02c/bench.json 2> gpurun_out/r02c/bench.err
python - <<'PY' > gpurun_out/r02c/lagx.log 2>&1
import sys, time, json
sys.path.insert(0, '.')
import numpy as np, torch
from pyslam_amd import synthetic
from pyslam_amd.device import DeviceProblem
lp, _ = synthetic.stereo_ba(200, 50000, 10, 20, seed=0)
for lagx in (1, 0):
    dev = DeviceProblem(lp)
    dev.set_option('coarse_lag_x', lagx)
    dev.snapshot()
    for _ in range(5):
        dev.restore(); out = dev.gn_iteration(0., 1e-12, 1000, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        dev.restore(); out = dev.gn_iteration(0., 1e-12, 1000, True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 50 * 1e3
    dev.set_profiling(2)
    for _ in range(5):
        dev.restore(); dev.gn_iteration(0., 1e-12, 1000, True)
    st = dev.stage_times(reset=True)
    dev.set_profiling(0)
    # trajectory
    dev.restore()
    traj = [dev.gn_iteration(0., 1e-12, 1000, True) for _ in range(5)]
    print('lagx', lagx, 'ms/iter %.4f' % ms, 'out', out, {k: round(v[0] / v[1], 4) for k, v in st.items() if v[1]})
    print('   trajectory', [(round(c, 6), n) for c, _, n, _ in traj])
    dev.close()
PY
tail -4 gpurun_out/r02c/tests.log; cat gpurun_out/r02c/lagx.log
