# timeline of one steady-state iteration (rocprofv3 kernel trace timestamps): solver-queue gaps and side-stream kernels
# usage: bash tools/gap_trace.sh [kf lm]      (default: the C3 bench workload)
export TMPDIR=/tmp
EXTRA=""
if [ -n "$1" ]; then EXTRA="--kf $1 --lm $2"; fi
BENCH="python $PWD/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-c4 $EXTRA"
(cd /tmp && rm -rf /tmp/gt && rocprofv3 --kernel-trace --output-format csv -d /tmp/gt -o kt -- $BENCH > /dev/null 2>&1)
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/gt/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_landmark_pass' in r['Kernel_Name']]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]['Start_Timestamp'])
mainq = rows[a]['Queue_Id']
prev_end = None
agg = collections.OrderedDict()
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')[:34]
    if r['Queue_Id'] != mainq:
        print('      side  start %8.1f  dur %7.1f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, name)) if (e - s) > 20000 else None
        continue
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    prev_end = e
    if gap > 2.0 or (e - s) > 30000:
        print('main  start %8.1f  gap %6.1f  dur %7.1f  %s' % ((s - t0) / 1e3, gap, (e - s) / 1e3, name))
print('iteration span %.1f us' % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3))
PY
