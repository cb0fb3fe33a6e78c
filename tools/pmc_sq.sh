#!/bin/bash
# SQ counters of the observation passes (VALU-bound or waiting?): bash tools/pmc_sq.sh <tag> [kf lm]
TAG=${1:-sq}; KF=${2:-200}; LM=${3:-50000}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  (cd /tmp && rm -rf /tmp/sq$i && rocprofv3 --pmc $SET --kernel-trace --output-format csv -d /tmp/sq$i -o p -- python $ROOT/tools/stage_probe.py $KF $LM > /dev/null 2>&1)
  find /tmp/sq$i -name '*counter_collection.csv' -exec cp {} $OUT/set$i.csv \;
done
python - $OUT <<'P'
import csv, sys, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in sorted(glob.glob(sys.argv[1] + '/set*.csv')):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].split('(')[0].split('<')[0].replace('void ', '')
        if n not in ('k_landmark_pass', 'k_pose_pass', 'k_backsub', 'k_cost_reproj', 'k_schur_pairs_db'): continue
        a = acc[n][r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for n, d in acc.items():
    print(n)
    for c, (v, k) in sorted(d.items()): print('    %-28s %14.1f  (avg of %d)' % (c, v / k, k))
P
