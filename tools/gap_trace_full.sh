export TMPDIR=/tmp
BENCH="python $PWD/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-c4"
(cd /tmp && rm -rf /tmp/gt && rocprofv3 --kernel-trace --output-format csv -d /tmp/gt -o kt -- $BENCH > /dev/null 2>&1)
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/gt/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_landmark_pass' in r['Kernel_Name']]
a, b = idx[-4], idx[-2]
t0 = int(rows[a]['Start_Timestamp'])
mainq = rows[a]['Queue_Id']
prev_end = None
for r in rows[a:b+1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')[:40]
    if r['Queue_Id'] != mainq:
        print('      side  start %8.1f  dur %7.1f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, name)); continue
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    prev_end = e
    print('main  start %8.1f  gap %6.1f  dur %7.1f  %s' % ((s - t0) / 1e3, gap, (e - s) / 1e3, name))
PY
