#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root:  bash tools/collect_profiles.sh <tag>
# Writes everything the committed profiles/ summaries are derived from into gpurun_out/<tag>/:
#   bench.json                     default bench run (CPU baseline, trajectory, C4 single GPU included)
#   ktrace_kernel_stats.csv        rocprofv3 --kernel-trace --stats of the C3 bench command
#   c4_kernel_stats.csv            the same for the C4 problem on one GPU (bench.py --kf 2000 --lm 500000)
#   pmc_<set>.csv, pmc4_<set>.csv  one rocprofv3 --pmc pass per counter set (no other trace domains), C3 and C4
#   bench_sharded_1rank.json       the multi-GPU driver (native RCCL) forced on with one rank, C3 and C4
#   source_sha.txt                 hash of pyslam_amd/csrc/* (bench.py: kernel_source_sha) the passes were taken on
# tools/summarize_profiles.py <tag> <round> then turns them into profiles/ (run locally).
set -u
TAG=${1:-r02}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python -c "import bench; print(bench.kernel_source_sha())" > "$OUT/source_sha.txt"
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
python bench.py --force-sharded --no-cpu-baseline --no-c4 --no-wall 2> /dev/null | grep '^{' > "$OUT/bench_sharded_1rank.json"
python bench.py --force-sharded --no-cpu-baseline --no-c4 --no-wall --kf 2000 --lm 500000 --steps 10 2> /dev/null | grep '^{' > "$OUT/bench_sharded_1rank_c4.json"
BENCH="python $PWD/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-c4 --no-wall"
(cd /tmp && rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- $BENCH > "$OUT/bench_under_rocprof.json" 2> /dev/null)
find /tmp/kt -name '*kernel_stats.csv' -exec cp {} "$OUT/ktrace_kernel_stats.csv" \;
(cd /tmp && rm -rf /tmp/kt4 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt4 -o kt -- $BENCH --kf 2000 --lm 500000 --steps 8 > "$OUT/bench_c4_under_rocprof.json" 2> /dev/null)
find /tmp/kt4 -name '*kernel_stats.csv' -exec cp {} "$OUT/c4_kernel_stats.csv" \;
PMCB="python $PWD/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-c4 --no-wall"
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
    i=$((i + 1))
    (cd /tmp && rm -rf /tmp/pmc$i && rocprofv3 --pmc $SET --kernel-trace --output-format csv -d /tmp/pmc$i -o p -- $PMCB > /dev/null 2>&1)
    find /tmp/pmc$i -name '*counter_collection.csv' -exec cp {} "$OUT/pmc_$(echo $SET | tr ' ' '_').csv" \;
done
# the same counters at the north star's size (C4: Z = 640 MB no longer fits the 256 MiB Infinity Cache, so FETCH_SIZE is HBM traffic there)
PMC4="python $PWD/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-c4 --no-wall --kf 2000 --lm 500000"
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
    i=$((i + 1))
    (cd /tmp && rm -rf /tmp/pmc4_$i && rocprofv3 --pmc $SET --kernel-trace --output-format csv -d /tmp/pmc4_$i -o p -- $PMC4 > /dev/null 2>&1)
    find /tmp/pmc4_$i -name '*counter_collection.csv' -exec cp {} "$OUT/pmc4_$(echo $SET | tr ' ' '_').csv" \;
done
ls -la "$OUT"
