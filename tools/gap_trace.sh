# gaps between consecutive kernels of one steady-state C3 iteration (rocprofv3 kernel trace timestamps)
export TMPDIR=/tmp
BENCH="python $PWD/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-c4"
(cd /tmp && rm -rf /tmp/gt && rocprofv3 --kernel-trace --output-format csv -d /tmp/gt -o kt -- $BENCH > /dev/null 2>&1)
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/gt/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# find the last but one k_landmark_pass and print until the next
idx = [i for i, r in enumerate(rows) if 'k_landmark_pass' in r['Kernel_Name']]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]['Start_Timestamp'])
prev_end = None
tot_gap = 0
for r in rows[a - 2:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) if prev_end is not None else 0
    q = r.get('Queue_Id', '?')
    print('%8.1f us  +%6.1f gap  dur %7.1f  q %s  %s' % ((s - t0) / 1e3, gap / 1e3, (e - s) / 1e3, q, r['Kernel_Name'].split('(')[0][:40]))
    prev_end = max(prev_end or 0, e) if gap < -0.0 else e
PY
