export TMPDIR=/tmp
BENCH="python $PWD/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-c4 --kf 2000 --lm 500000"
(cd /tmp && rm -rf /tmp/kt4 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt4 -o kt -- $BENCH > /dev/null 2>&1)
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/kt4/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
n = float([r for r in rows if 'k_landmark_pass' in r['Name']][0]['Calls'])
for r in rows[:14]:
    print('%-36s calls %5s avg %8.1f us per-iter %8.1f' % (r['Name'].split('(')[0][:36], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e3/n))
PY
