# usage: bash tools/ldi_trace.sh <tag> [probe args]  -> gpurun_out/<tag>/ldi_kernel_stats.csv (+ the probe's own output)
TAG=${1:-lt}; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python $PWD/tools/ldi_probe.py $*"
(cd /tmp && rm -rf /tmp/lt && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lt -o lt -- $CMD > "$OUT/probe_under_rocprof.log" 2> /dev/null)
find /tmp/lt -name '*kernel_stats.csv' -exec cp {} "$OUT/ldi_kernel_stats.csv" \;
head -30 "$OUT/ldi_kernel_stats.csv" | cut -c1-160
