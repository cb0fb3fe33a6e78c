# per-kernel averages of the explicit PCG, four-launch against three-launch form (C2 by default)
export TMPDIR=/tmp
WHICH=${1:-c2}
cat > /tmp/xt.py <<PY
import sys, os
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
from pyslam_amd import synthetic
from pyslam_amd.device import DeviceProblem
if '$WHICH' == 'c2': lp, _ = synthetic.pose_graph(num_poses=10000, num_loops=40001, dof=6, seed=2)
else: lp, _ = synthetic.stereo_ba(2000, 500000, 10, 20, seed=1)
dev = DeviceProblem(lp)
dev.set_option('xcg_restrict_fused', int(sys.argv[1]))
dev.snapshot()
for _ in range(4):
    dev.restore(); out = dev.gn_iteration(0., 1e-12, 4000, True)
print('rt', sys.argv[1], 'iters', out[2])
PY
for rt in 0 1; do
(cd /tmp && rm -rf /tmp/xtp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/xtp -o kt -- python /tmp/xt.py $rt 2>/dev/null | grep iters)
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/xtp/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'k_xcg' in r['Name']: print('   %-40s calls %6s avg us %7.2f' % (r['Name'].split('(')[0][:40], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
