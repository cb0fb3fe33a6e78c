# timeline of one iteration of the C2 pose graph (rocprofv3 kernel trace): solver-queue gaps, side-stream kernels, CG pace
export TMPDIR=/tmp
cat > /tmp/xt.py <<PY
import sys, os
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
from pyslam_amd import synthetic
from pyslam_amd.device import DeviceProblem
lp, _ = synthetic.pose_graph(num_poses=int(os.environ.get('PG_POSES', '10000')), num_loops=4 * int(os.environ.get('PG_POSES', '10000')) + 1, dof=6, seed=2)
dev = DeviceProblem(lp)
for _ in range(7):
    out = dev.gn_iteration(0., 1e-12, 4000, True)
print('iters', out[2])
PY
(cd /tmp && rm -rf /tmp/xtp && rocprofv3 --kernel-trace --output-format csv -d /tmp/xtp -o kt -- python /tmp/xt.py 2>/dev/null | grep iters)
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/xtp/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f))); rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_factor_pass' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]['Start_Timestamp']); mainq = rows[a]['Queue_Id']; prev = None; ncg = 0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp']); name = r['Kernel_Name'].split('(')[0].replace('void ', '')[:30]
    if r['Queue_Id'] != mainq:
        if e - s > 20000: print('      side start %8.1f dur %7.1f %s' % ((s - t0) / 1e3, (e - s) / 1e3, name))
        continue
    gap = (s - prev) / 1e3 if prev else 0.0; prev = e
    if 'k_xcg_spmv' in name: ncg += 1
    if gap > 8 or (e - s) > 30000 or ('spmv' in name and ncg % 10 == 1):
        print('main start %8.1f gap %6.1f dur %7.1f %s (cg %d)' % ((s - t0) / 1e3, gap, (e - s) / 1e3, name, ncg))
print('span %.1f' % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3))
PY
