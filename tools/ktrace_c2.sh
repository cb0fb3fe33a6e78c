# usage (ON THE GPU BOX): bash tools/ktrace_c2.sh <tag>  -> gpurun_out/<tag>/c2_kernel_stats.csv
# rocprofv3 --kernel-trace --stats of eight whole-iteration calls on C2 (10 000 SE(3) poses, 50 001 edges, Huber)
TAG=${1:-r03}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
R=$PWD
cat > /tmp/c2_run.py <<PY
import os, sys, time
sys.path.insert(0, "$R")
import torch
from pyslam_amd import synthetic, losses
from pyslam_amd.device import DeviceProblem
lp = synthetic.pose_graph(num_poses=10000, num_loops=40001, dof=6, seed=2, loss=losses.HuberLoss(1.0))[0]
dev = DeviceProblem(lp); dev.eval_cost(True)
for k in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter(); out = dev.gn_iteration(0., 1e-12, 4000, True)
    print('iteration', k, round((time.perf_counter() - t0) * 1e3, 3), 'ms  pcg', out[2], 'relres %.1e' % out[3], flush=True)
print(dev.get_info())
PY
(cd /tmp && rm -rf /tmp/kt2 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -o kt -- python /tmp/c2_run.py > "$OUT/c2_run.txt" 2> /dev/null)
find /tmp/kt2 -name '*kernel_stats.csv' -exec cp {} "$OUT/c2_kernel_stats.csv" \;
grep -v amdgpu "$OUT/c2_run.txt" | head -12
python - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/c2_kernel_stats.csv")))
for r in rows[:12]:
    print('%-44s calls %5s avg %8.2f us total %9.1f us' % (r['Name'].split('(')[0][:44], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e3))
PY
