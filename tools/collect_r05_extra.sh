#!/bin/bash
# Run ON THE GPU BOX after tools/collect_profiles.sh <tag>: the round-5 measurements that are not bench lines.
# Output: gpurun_out/<tag>/r05_extra.txt (copied to profiles/r05_other_configs.txt)
TAG=${1:-r05}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
F="grep -v amdgpu.ids"
{
echo "# cold, reference-terminated solves (tools/cold_probe.py: ps_solve under the options of reference examples/stereo_ba.py:38-40)"
python tools/cold_probe.py 200 50000 6 2>&1 | tail -3
python tools/cold_probe.py 200 50000 6 --python-loop 2>&1 | tail -1
echo "## C3 with the 16-lane landmark pass / back-substitution (lm_packed 0)"
python tools/cold_probe.py 200 50000 6 --opt=lm_packed:0 2>&1 | tail -1
echo "## C4 on one GPU; then lm_packed 0; then band_part 0 (serial band factorisation of the coarse matrix)"
python tools/cold_probe.py 2000 500000 3 2>&1 | tail -2
python tools/cold_probe.py 2000 500000 3 --opt=lm_packed:0 2>&1 | tail -1
python tools/cold_probe.py 2000 500000 3 --opt=band_part:0 2>&1 | tail -2
python tools/cold_probe.py 600 150000 3 2>&1 | tail -1
python tools/cold_probe.py 1000 60000 3 2>&1 | tail -1
echo
echo "# C2 (10 000 SE(3) poses, 50 001 edges, Huber): cold solves; then band_part 0; a 1 500- and a 5 000-pose graph"
python tools/cold_probe.py 10000 40001 3 --pg 2>&1 | tail -2
python tools/cold_probe.py 10000 40001 3 --pg --opt=band_part:0 2>&1 | tail -2
python tools/cold_probe.py 1500 6000 2 --pg 2>&1 | tail -1
python tools/cold_probe.py 5000 20000 2 --pg 2>&1 | tail -1
echo "## C2 kernel stats (rocprofv3, tools/c2_kstats.sh)"
bash tools/c2_kstats.sh ${TAG}_c2k 2>&1 | grep -v "simple_timer"
echo
echo "# coarse-level band factorisation + inverse on its own: serial walk against the partitioned form (tools/bandpart_probe.py)"
python tools/bandpart_probe.py 2>&1 | $F
echo "## per-kernel durations (tools/bandpart_trace.sh: rocprofv3 kernel trace; C4's and C2's automatic chunk size)"
bash tools/bandpart_trace.sh 2>&1 | $F | awk '/grid   1280/ && ++a==3 {p=1} p && ++n<=9' 
echo
echo "# per-stage GPU time at a repeated point (tools/stage_probe.py): packed / 16-lane, C3 and C4"
python tools/stage_probe.py 200 50000 --label=packed 2>&1 | grep "kf "
python tools/stage_probe.py 200 50000 --label=lanes16 --opt=lm_packed:0 2>&1 | grep "kf "
python tools/stage_probe.py 2000 500000 --label=packed 2>&1 | grep "kf "
python tools/stage_probe.py 2000 500000 --label=lanes16 --opt=lm_packed:0 2>&1 | grep "kf "
echo
echo "# C3: CG iterations against the number of coarse hat intervals (cold_probe --opt=coarse_groups:G)"
for g in -1 16 24 32; do python tools/cold_probe.py 200 50000 3 --opt=coarse_groups:$g 2>&1 | tail -2 | head -1 | cut -c1-160; done
echo
echo "# one rank of N on the C4 problem, shard-local stages, first-pose split against the caller-index split (tools/shard_stage_probe.py)"
python tools/shard_stage_probe.py 8 0 3 2>&1 | grep "^N "
python tools/shard_stage_probe.py 4 1 2>&1 | grep "^N "
python tools/shard_stage_probe.py 2 1 2>&1 | grep "^N "
echo
echo "# SQ counters of the observation passes and the Schur kernel, C3 (tools/pmc_sq.sh; one --pmc pass per set)"
bash tools/pmc_sq.sh ${TAG}_sq 2>&1 | $F
echo
echo "# second half of round 5 -- the cost summed by the next landmark pass (fuse_cost), the folded CG in one launch (cg_persist), the explicit PCG of bundle adjustments in one launch (xcg_persist): medians over cold solves"
for o in "" "--opt=fuse_cost:0" "--opt=cg_persist:0" "--opt=fuse_cost:0 --opt=cg_persist:0"; do echo "C3 [$o] $(python tools/cold_probe.py 200 50000 30 $o 2>&1 | grep "median over")"; done
for o in "" "--opt=fuse_cost:0"; do echo "C4 [$o] $(python tools/cold_probe.py 2000 500000 8 $o 2>&1 | grep "median over")"; done
for cfg in "2000 500000 6" "1500 400000 4" "1000 60000 6" "600 150000 6"; do for o in "" "--opt=xcg_persist:0"; do echo "BA $cfg [$o] $(python tools/cold_probe.py $cfg $o 2>&1 | grep "median over")"; done; done
for cfg in "100 60" "200 150" "240 300"; do for o in "" "--opt=cg_persist:0"; do echo "SE(3) graph $cfg [$o] $(python tools/cold_probe.py $cfg 12 --pg $o 2>&1 | grep "median over")"; done; done
echo "## phase clocks of the one-launch CG's first workgroup, C3 (measurement build, PS_CP_CLOCKS; the clock reads themselves cost ~0.1 us each)"
python -c "import __graft_entry__ as g; g.build_measure()" > /dev/null 2>&1
PYSLAM_AMD_MEASURE=1 PS_CP_CLOCKS=1 python tools/cold_probe.py 200 50000 12 2>&1 | grep "k_cg_persist"
echo "## the exchange on its own (tools/probes/allgather_probe.hip: nwg, threads, doubles, iterations, stand-in work, one-XCD flag)"
hipcc --offload-arch=gfx950 -O3 -o /tmp/agp tools/probes/allgather_probe.hip 2> /dev/null
for cfg in "32 512 1280 40 0 0" "64 512 1280 40 0 0" "32 512 128 40 0 0" "32 512 4096 40 0 0" "32 512 1280 40 0 1"; do echo "[$cfg] $(timeout 60 /tmp/agp $cfg 2>&1 | tail -2 | head -1)"; done
echo
echo "# small pose graphs (BASELINE configuration 1 and neighbours; tools/c1_probe.py: ten whole-iteration calls, lagged inverse on / off)"
python tools/c1_probe.py 2>&1 | $F
echo
echo "# batch entry: C3 from tables end to end (tests/test_gpu_batch_entry.py)"
python -m pytest tests/test_gpu_batch_entry.py -q -s -k c3 2>&1 | grep "solve_tables\|passed\|failed"
echo
echo "# Schur pair kernel ablation at C3 (measurement build; tools/schur_ablate.py) and block-row order against longest-first (PS_SCHUR_NO_LPT)"
python -c "import __graft_entry__ as g; g.build_measure()" > /dev/null 2>&1
PYSLAM_AMD_MEASURE=1 python tools/schur_ablate.py 2>&1 | grep ablate
PYSLAM_AMD_MEASURE=1 python tools/stage_probe.py 200 50000 --label=longest-first 2>&1 | grep "kf " | cut -c1-120
PYSLAM_AMD_MEASURE=1 PS_SCHUR_NO_LPT=1 python tools/stage_probe.py 200 50000 --label=block-order 2>&1 | grep "kf " | cut -c1-120
echo
echo "# ps_problem_create stages (tools/create_time.py; the measurement build prints the laps)"
PYSLAM_AMD_MEASURE=1 PS_CREATE_TIMING=1 python tools/create_time.py 2>&1 | grep -v "amdgpu.ids\| 0.0 ms\|build_coarse" | tail -30
echo
echo "# the first whole-iteration call of a fresh process (tools/first_call_probe.py)"
python tools/first_call_probe.py 2>&1 | tail -1
} > "$OUT/r05_extra.txt" 2>&1
tail -5 "$OUT/r05_extra.txt"
