# Randomised sweeps on fresh seeds, logs kept: bash tools/fuzz_campaign.sh <tag> [scale]   (run ON THE GPU BOX)
TAG=${1:-fuzz}
K=${2:-1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python -c "import bench; print('kernel sources', bench.kernel_source_sha())" > $OUT/summary.txt
run() { # name script cases seed0
  ( timeout 1500 python tests/$2 $3 $4 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" ) > $OUT/$1.txt
  echo "$1: $(tail -1 $OUT/$1.txt)   [seeds $4 .. $(( $4 + $3 - 1 )); FAIL lines: $(grep -c '^FAIL' $OUT/$1.txt)]" >> $OUT/summary.txt
}
run parity fuzz_parity.py $((1500 * K)) 1520000
run solve fuzz_solve.py $((600 * K)) 1530000
run api fuzz_api.py $((400 * K)) 1540000
run cov fuzz_cov.py $((300 * K)) 1550000
run shards fuzz_shards.py $((120 * K)) 1560000
run small fuzz_small.py $((800 * K)) 1570000
cat $OUT/summary.txt
