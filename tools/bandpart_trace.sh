#!/bin/bash
# per-kernel durations of the partitioned band factorisation at C2's / C4's coarse sizes (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
OUT=${GRAFT_REPO_ROOT:-/root/repo}/gpurun_out/bp_trace
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bp -- python ${GRAFT_REPO_ROOT:-/root/repo}/tools/bandpart_probe.py > $OUT/run.txt 2>&1
python - <<'P'
import csv, glob, os
out = os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'gpurun_out/bp_trace')
f = glob.glob(out + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = None
for r in rows:
    n = r['Kernel_Name'].split('(')[0][-40:]
    if not any(k in n for k in ('k_band', 'k_bp_')): continue
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%-42s grid %6s  dur %8.1f us  gap %6.1f' % (n, r.get('Grid_Size_X', r.get('Grid_Size', '?')), (e - s) / 1e3, 0 if t0 is None else (s - t0) / 1e3))
    t0 = e
P
