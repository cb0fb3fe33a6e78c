#!/bin/bash
# Run ON THE GPU BOX after tools/collect_profiles.sh <tag>: the round-4 measurements that are not bench lines.
# Output: gpurun_out/<tag>/r04_extra.txt (copied to profiles/r04_other_configs.txt)
TAG=${1:-r04}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
{
echo "# cold, reference-terminated solves (tools/cold_probe.py: ps_solve under the options of reference examples/stereo_ba.py:38-40)"
python tools/cold_probe.py 200 50000 6 2>&1 | tail -3
python tools/cold_probe.py 200 50000 6 --python-loop 2>&1 | tail -1
python tools/cold_probe.py 2000 500000 3 2>&1 | tail -2
python tools/cold_probe.py 2000 500000 3 --opt=coarse_adaptive_hold:0 2>&1 | tail -1
python tools/cold_probe.py 600 150000 3 2>&1 | tail -1
python tools/cold_probe.py 1000 60000 3 2>&1 | tail -1
echo
echo "# C2 (10 000 SE(3) poses, 50 001 edges, Huber) and a 1 500-pose graph, cold solves"
python tools/cold_probe.py 10000 40001 2 --pg 2>&1 | tail -2
python tools/cold_probe.py 10000 40001 2 --pg --opt=coarse_adaptive_hold:0 2>&1 | tail -1
python tools/cold_probe.py 10000 40001 2 --pg --opt=coarse_groups:400 2>&1 | tail -1
python tools/cold_probe.py 1500 6000 2 --pg 2>&1 | tail -1
python tools/cold_probe.py 5000 20000 2 --pg 2>&1 | tail -1
echo "# its timeline under rocprofv3 --kernel-trace (tools/cold_trace_pg.sh)"
bash tools/cold_trace_pg.sh 2>&1 | grep -v "^solve\|^kf"
echo
echo "# the first whole-iteration call of a fresh process (tools/first_call_probe.py)"
python tools/first_call_probe.py 2>&1 | tail -1
echo
echo "# ps_problem_create stages (tools/create_time.py; the measurement build prints the laps)"
python -c "import __graft_entry__ as g; g.build_measure()" > /dev/null 2>&1
echo "## structure built on the device (default from 200 000 observations; csrc/ps_host_build.h)"
PYSLAM_AMD_MEASURE=1 PS_CREATE_TIMING=1 python tools/create_time.py 2>&1 | grep -v "amdgpu.ids\| 0.0 ms\|build_coarse"
echo "## the host builder (PS_CREATE_DEVICE=0: the device build's test oracle)"
PYSLAM_AMD_MEASURE=1 PS_CREATE_TIMING=1 PS_CREATE_DEVICE=0 python tools/create_time.py 2>&1 | grep "again\|synthetic"
echo
echo "# pose-stationary Schur kernel against the pipelined gather kernel (tools/schur_probe.py, PS_SCHUR_MODE=2) + its ablation"
ABLATE=1 python tools/schur_probe.py 2>&1 | grep -v amdgpu.ids | tail -12
python tools/schur_probe.py 2000 500000 2>&1 | tail -2
echo
echo "# timeline of one cold C3 solve under rocprofv3 --kernel-trace (tools/cold_trace.sh; the profiler stretches the launch gaps)"
bash tools/cold_trace.sh 2>&1 | grep "iteration [1-4]:\|ms per iteration"
echo
echo "# the sharded protocol at shard size on ONE GPU, cold solves (bench.py --force-sharded --kf 2000 --lm L: what rank 0 of an N-GPU run executes, 1-rank RCCL)"
for L in 500000 250000 125000 62500; do
  python bench.py --force-sharded --kf 2000 --lm $L --steps 8 --no-cpu-baseline --no-c4 --no-wall 2> /dev/null | grep '^{' | python -c "
import json, sys
b = json.loads(sys.stdin.read())
print('landmarks', $L, 'ms per iteration (cold solves)', b['value'], 'per call', b['cold_solve']['per_call_ms'], 'stage_ms', b.get('stage_ms'), 'native_rccl', b.get('native_rccl'))"
done
} > "$OUT/r04_extra.txt" 2>&1
tail -5 "$OUT/r04_extra.txt"
