# timeline of the last cold solve of a pose graph (tools/cold_probe.py --pg) under rocprofv3 --kernel-trace: per iteration (cut at
# k_factor_pass on the solver queue) span, kernel time, launches, long kernels and gaps; kernels of the other queues (side-stream
# factorisations) longer than 50 us with their start relative to the solve
# usage: bash tools/cold_trace_pg.sh [poses loops iterations_per_solve]
export TMPDIR=/tmp
NIT=${3:-5}
CMD="python $PWD/tools/cold_probe.py ${1:-10000} ${2:-40001} 3 --pg"
(cd /tmp && rm -rf /tmp/ctp && rocprofv3 --kernel-trace --output-format csv -d /tmp/ctp -o kt -- $CMD 2> /dev/null | tail -2)
NIT=$NIT python - <<'PY'
import csv, glob, os
nit = int(os.environ['NIT'])
f = glob.glob('/tmp/ctp/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
fp = [r for r in rows if 'k_factor_pass' in r['Kernel_Name']]
mainq = fp[-1]['Queue_Id']
main = [r for r in rows if r['Queue_Id'] == mainq]
cut = [i for i, r in enumerate(main) if 'k_factor_pass' in r['Kernel_Name']]
# a solve = start cost (one k_factor_pass-free cost pass) + nit linearisations; the post-step cost passes do not run k_factor_pass
starts = cut[-nit:]
t0 = int(main[starts[0]]['Start_Timestamp'])
tend = int(main[-1]['End_Timestamp'])
bounds = starts + [len(main)]
for k in range(nit):
    seg = main[bounds[k]:bounds[k + 1]]
    s0, e1 = int(seg[0]['Start_Timestamp']), int(seg[-1]['End_Timestamp'])
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg) / 1e3
    print('iteration %d: start %8.1f us  span %7.1f us  kernel time %7.1f us  %3d launches' % (k + 1, (s0 - t0) / 1e3, (e1 - s0) / 1e3, busy, len(seg)))
    prev = None
    for r in seg:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        gap = (s - prev) / 1e3 if prev else 0.0
        prev = e
        if gap > 20.0 or (e - s) > 100000:
            print('      +%8.1f  gap %6.1f  dur %7.1f  %s' % ((s - s0) / 1e3, gap, (e - s) / 1e3, r['Kernel_Name'].split('(')[0].replace('void ', '')[:40]))
print('other queues:')
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if r['Queue_Id'] != mainq and s >= t0 and s <= tend and e - s > 50000:
        print('      +%8.1f  dur %7.1f  queue %s  %s' % ((s - t0) / 1e3, (e - s) / 1e3, r['Queue_Id'], r['Kernel_Name'].split('(')[0].replace('void ', '')[:40]))
PY
