# timeline of one steady-state C3 iteration with the lagged dense inverse (kernel-trace timestamps, both queues)
export TMPDIR=/tmp
BENCH="python $PWD/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-c4"
(cd /tmp && rm -rf /tmp/gt && rocprofv3 --kernel-trace --output-format csv -d /tmp/gt -o kt -- $BENCH > /dev/null 2>&1)
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/gt/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_landmark_pass' in r['Kernel_Name']]
ini = [i for i, r in enumerate(rows) if 'k_ldi_init' in r['Kernel_Name']]
if ini:                      # the last two iterations that solved with the lagged dense inverse
    last = max(j for j, i in enumerate(idx) if i < ini[-1])
    a, b = idx[max(0, last - 1)], (idx[last + 1] if last + 1 < len(idx) else len(rows) - 1)
else:
    a, b = idx[-4], idx[-2]
t0 = int(rows[a]['Start_Timestamp'])
mainq = rows[a]['Queue_Id']
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')[:30]
    print('%s start %8.1f end %8.1f dur %7.1f  %s' % ('main' if r['Queue_Id'] == mainq else '   side', (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, name))
PY
