#!/bin/bash
# (PS_* measurement switches exist only in the measurement build: python -c "import __graft_entry__ as g; g.build_measure()" first)
export PYSLAM_AMD_MEASURE=1
# Run ON THE GPU BOX: the non-bench configurations of SURVEY 8d (C4, C2, C5 / small problems) and the RANSAC step.
# Output: gpurun_out/<tag>/other_configs.txt  (copied to profiles/<round>_other_configs.txt)
TAG=${1:-r01}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
{
echo "# C4: 2 000 keyframes x 500 000 landmarks x 10 observations (tools/c4_probe.py)"
python tools/c4_probe.py 2>&1 | grep -v amdgpu.ids
echo
echo "# C2: 10 000 SE(3) poses, 50 001 edges, Huber (tools/c2_probe.py)"
python tools/c2_probe.py 2>&1 | grep -v amdgpu.ids
echo
echo "# small problems (tools/small_probe.py)"
python tools/small_probe.py 2>&1 | grep -v amdgpu.ids
echo
echo "# frame-to-frame RANSAC (tests/bench_ransac.py)"
python tests/bench_ransac.py 2>&1 | grep -v amdgpu.ids
echo
echo "# dense photometric alignment, 640 x 480 (tools/photo_bench.py)"
python tools/photo_bench.py 2>&1 | grep -v amdgpu.ids
echo
echo "# lagged dense inverse on / off: steady state, two trajectories, counters (tools/ldi_probe.py)"
python tools/ldi_probe.py c3 kf100 kf250 pg200 2>&1 | grep -v amdgpu.ids
echo
echo "# one-launch explicit two-level PCG on / off (tools/xf_probe.py)"
python tools/xf_probe.py mid c4 c2 2>&1 | grep -v amdgpu.ids
echo
echo "# explicit two-level PCG, three / one (auto) / two launches per iteration over problem sizes (tools/xf_forms_probe.py)"
python tools/xf_forms_probe.py 2>&1 | grep -v amdgpu.ids
echo
echo "# small pose graphs (BASELINE configuration 1): direct seed of the lagged inverse on / off (tools/c1_probe.py)"
python tools/c1_probe.py 2>&1 | grep -v amdgpu.ids
echo
echo "# lagged dense inverse on / off over BA sizes, eight-call trajectories (tools/ldi_size_sweep.py)"
python tools/ldi_size_sweep.py 2>&1 | grep -v amdgpu.ids
echo
echo "# folded CG / explicit PCG / lagged inverse around the sizes where the default switches (tools/path_threshold_probe.py)"
python tools/path_threshold_probe.py 2>&1 | grep -v amdgpu.ids
echo
echo "# per-frame motion-only Problem through the public API (tools/c5_frame_probe.py)"
python tools/c5_frame_probe.py 2>&1 | grep -v amdgpu.ids | tail -1
echo
echo "# Schur pair kernel ablation at C3 (tools/schur_ablate.py: 1 no compute, 2 no fetch, 3 neither, 4 no a-rows)"
python tools/schur_ablate.py 2>&1 | grep ablate
echo
echo "# the sharded protocol at shard size on ONE GPU (bench.py --force-sharded --kf 2000 --lm L: what rank 0 of an N-GPU run executes, 1-rank RCCL; DESIGN.md section 6)"
for L in 500000 125000 62500; do
  python bench.py --force-sharded --kf 2000 --lm $L --steps 10 --no-cpu-baseline --no-c4 2> /dev/null | grep '^{' | python -c "
import json, sys
b = json.loads(sys.stdin.read())
print('landmarks', $L, 'iteration ms', b['value'], 'stage_ms', b.get('stage_ms'), 'native_rccl', b.get('native_rccl'))"
done
echo
echo "# ps_problem_create stages (tools/create_time.py, PS_CREATE_TIMING=1)"
PS_CREATE_TIMING=1 python tools/create_time.py 2>&1 | grep -v amdgpu.ids
} > "$OUT/other_configs.txt"
tail -5 "$OUT/other_configs.txt"
