# usage: bash tools/run_quick.sh <tag> : gpu tests (stop at first failure), then kernel trace of the C3 bench
TAG=${1:-q}
mkdir -p gpurun_out/$TAG
(time python -m pytest tests -m gpu -x -q 2>&1 | tail -30) > gpurun_out/$TAG/tests.log 2>&1
grep -n "passed\|failed" gpurun_out/$TAG/tests.log
python bench.py --no-cpu-baseline --no-c4 2> gpurun_out/$TAG/bench.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], d['stage_ms'], d['trajectory_ms_per_iter']['per_iter'], d['trajectory_ms_per_iter']['pcg_iters'])
"
bash tools/ktrace.sh $TAG > /dev/null 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/$TAG/kernel_stats.csv')))
for r in rows[:22]:
    print('%-40s calls %5s avg %9.1f us  %5s%%' % (r['Name'].split('(')[0][:40], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
