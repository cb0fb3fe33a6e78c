# kernel timeline of ONE trajectory (6 iterations from the restored start) with the lagged dense inverse live
export TMPDIR=/tmp
REPO=$PWD
cat > /tmp/traj_run.py <<'PY'
import sys, time, torch
sys.path.insert(0, sys.argv[1])
from pyslam_amd import synthetic
from pyslam_amd.device import DeviceProblem
lp, _ = synthetic.stereo_ba(num_kf=200, num_lm=50000, obs_per_lm=10, half_window=20, seed=0)
dev = DeviceProblem(lp, stream=torch.cuda.current_stream().cuda_stream)
dev.eval_cost(True); dev.snapshot()
for _ in range(8):
    dev.restore(); dev.gn_iteration(0., 1e-12, 1000, True)
for rep in range(2):
    dev.restore(); torch.cuda.synchronize()
    for _ in range(6):
        t0 = time.perf_counter(); out = dev.gn_iteration(0., 1e-12, 1000, True)
        print('iter %.4f ms its %d' % ((time.perf_counter() - t0) * 1e3, out[2]))
PY
(cd /tmp && rm -rf /tmp/gt && rocprofv3 --kernel-trace --output-format csv -d /tmp/gt -o kt -- python /tmp/traj_run.py $REPO 2>&1 | grep -v amdgpu.ids | tail -20)
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/gt/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_landmark_pass' in r['Kernel_Name']]
a = idx[-6]
t0 = int(rows[a]['Start_Timestamp'])
mainq = rows[a]['Queue_Id']
prev = None
for r in rows[a:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')[:30]
    main = r['Queue_Id'] == mainq
    gap = (s - prev) / 1e3 if (main and prev is not None) else 0.0
    if main: prev = e
    if 'cg_fused' in name and gap < 3: continue
    print('%s start %8.1f dur %7.1f gap %6.1f  %s' % ('main' if main else '   side', (s - t0) / 1e3, (e - s) / 1e3, gap, name))
PY
