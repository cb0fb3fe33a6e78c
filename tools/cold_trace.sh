# timeline of the last cold solve of tools/cold_probe.py (rocprofv3 kernel-trace timestamps): per iteration the span, the
# kernel time on the solver queue and the gaps in it; side-stream kernels longer than 20 us
# usage: bash tools/cold_trace.sh [kf lm]
export TMPDIR=/tmp
CMD="python $PWD/tools/cold_probe.py ${1:-200} ${2:-50000} 4 $3"
(cd /tmp && rm -rf /tmp/ct && rocprofv3 --kernel-trace --output-format csv -d /tmp/ct -o kt -- $CMD 2> /dev/null | tail -3)
python - <<'PY'
import csv, glob, os
f = glob.glob('/tmp/ct/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
lm = [i for i, r in enumerate(rows) if 'k_landmark_pass' in r['Kernel_Name']]
mainq = rows[lm[-1]]['Queue_Id']
main = [r for r in rows if r['Queue_Id'] == mainq]
lmm = [i for i, r in enumerate(main) if 'k_landmark_pass' in r['Kernel_Name']]
# the last solve = the last 4 linearisations (C3 / C4 under the example options)
first = lmm[-4]
# back up to the start-cost pass in front of it
j = first
while j > 0 and 'k_cost' not in main[j]['Kernel_Name']:
    j -= 1
t0 = int(main[j]['Start_Timestamp'])
bounds = [j] + lmm[-3:] + [len(main)]
for k in range(4):
    seg = main[bounds[k]:bounds[k + 1]]
    s0, e1 = int(seg[0]['Start_Timestamp']), int(seg[-1]['End_Timestamp'])
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg) / 1e3
    nxt = int(main[bounds[k + 1]]['Start_Timestamp']) if bounds[k + 1] < len(main) else e1
    print('iteration %d: start %8.1f us  first->last kernel %7.1f us  kernel time %7.1f us  %3d launches  idle before the next iteration %6.1f us'
          % (k + 1, (s0 - t0) / 1e3, (e1 - s0) / 1e3, busy, len(seg), (nxt - e1) / 1e3))
    prev = None
    for r in seg:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        gap = (s - prev) / 1e3 if prev else 0.0
        prev = e
        name = r['Kernel_Name'].split('(')[0].replace('void ', '')[:36]
        if gap > 3.0 or (e - s) > 15000 or (k == 0 and os.environ.get('COLD_TRACE_ALL')):
            print('      +%8.1f  gap %5.1f  dur %6.1f  %s' % ((s - s0) / 1e3, gap, (e - s) / 1e3, name))
PY
