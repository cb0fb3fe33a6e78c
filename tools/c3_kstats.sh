#!/bin/bash
# kernel stats (rocprofv3) of cold C3 solves: bash tools/c3_kstats.sh <tag> [cold_probe args]
TAG=${1:-c3k}; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/$TAG -o k -- python $ROOT/tools/cold_probe.py 200 50000 6 "$@" > $OUT/run.txt 2>&1
find /tmp/$TAG -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
python - "$OUT/kernel_stats.csv" <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r['Percentage']) > 0.6: print('%-50s calls %5s avg %8.1f us  %5.1f%%' % (r['Name'].split('(')[0][:50], r['Calls'], float(r['AverageNs']) / 1e3, float(r['Percentage'])))
P
tail -1 $OUT/run.txt
