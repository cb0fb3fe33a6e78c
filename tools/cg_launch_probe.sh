# average duration of one fused-CG launch at C3 with and without the coarse rows (rocprofv3 kernel trace)
export TMPDIR=/tmp
cat > /tmp/cgp.py <<'PY'
import sys, os
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
from pyslam_amd import synthetic
from pyslam_amd.device import DeviceProblem
lp, _ = synthetic.stereo_ba(200, 50000, 10, 20, seed=0)
dev = DeviceProblem(lp)
dev.set_option('coarse_groups', int(sys.argv[1]))
dev.snapshot()
for _ in range(12):
    dev.restore(); out = dev.gn_iteration(0., 1e-12, 2000, True)
print('groups', sys.argv[1], 'iters', out[2])
PY
for g in -1 0; do
(cd /tmp && rm -rf /tmp/ktp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktp -o kt -- python /tmp/cgp.py $g 2>/dev/null | grep groups)
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/ktp/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'k_cg_fused' in r['Name']: print('   ', r['Name'][:28], 'calls', r['Calls'], 'avg us %.2f' % (float(r['AverageNs']) / 1e3), 'min', r['MinNs'], 'max', r['MaxNs'])
PY
done
