# usage: bash tools/ktrace.sh <tag>   -> gpurun_out/<tag>/kernel_stats.csv (rocprofv3 --kernel-trace --stats of the C3 bench)
TAG=${1:-kt}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python $PWD/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-c4"
(cd /tmp && rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- $BENCH > "$OUT/bench_under_rocprof.json" 2> /dev/null)
find /tmp/kt -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
head -40 "$OUT/kernel_stats.csv" | cut -c1-150
