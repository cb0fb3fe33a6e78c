#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root:  bash tools/collect_profiles.sh <tag>
# Writes everything the committed profiles/ summaries are derived from into gpurun_out/<tag>/:
#   bench.json                     default bench run (CPU baseline included)
#   ktrace_kernel_stats.csv        rocprofv3 --kernel-trace --stats of the same bench command
#   pmc_<set>.csv                  one rocprofv3 --pmc pass per counter set (no other trace domains)
#   bench_sharded_1rank.json       the multi-GPU driver (native RCCL) forced on with one rank
# tools/summarize_profiles.py <tag> then turns them into profiles/ (run locally).
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
python bench.py --force-sharded --no-cpu-baseline 2> /dev/null | grep '^{' > "$OUT/bench_sharded_1rank.json"
BENCH="python $PWD/bench.py --steps 20 --warmup 3 --no-cpu-baseline"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- $BENCH > "$OUT/bench_under_rocprof.json" 2> /dev/null)
find /tmp/kt -name '*kernel_stats.csv' -exec cp {} "$OUT/ktrace_kernel_stats.csv" \;
PMCB="python $PWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
    i=$((i + 1))
    (cd /tmp && rocprofv3 --pmc $SET --kernel-trace --output-format csv -d /tmp/pmc$i -o p -- $PMCB > /dev/null 2>&1)
    find /tmp/pmc$i -name '*counter_collection.csv' -exec cp {} "$OUT/pmc_$(echo $SET | tr ' ' '_').csv" \;
done
ls -la "$OUT"
