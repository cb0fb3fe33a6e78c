#!/bin/bash
# Run ON THE GPU BOX: the side-stream corruption of HISTORY.md section 3 with the lowest-priority side stream back (PS_SIDE_LOWPRIO,
# measurement build), under one switch at a time -- which switch makes it go away says what it is.  tools/hunt_explicit_flake.py
# counts handles that disagree ("differences") and spurious failures ("errors") over N problems.
N=${1:-40}
export PYSLAM_AMD_MEASURE=1
run() { echo "== $1"; shift; env "$@" timeout 400 python tools/hunt_explicit_flake.py $N 0 2>&1 | grep -v "^RCCL\|^HIP \|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|^round \|^   \|^DIFF" | tail -5; }
run "ordinary side stream" X=1
run "lowest-priority side stream" PS_SIDE_LOWPRIO=1
run "lowest priority + AMD_SERIALIZE_KERNEL=3" PS_SIDE_LOWPRIO=1 AMD_SERIALIZE_KERNEL=3
run "lowest priority + A_c assembled on the solver stream" PS_SIDE_LOWPRIO=1 PS_XCG_AC_MAIN=1
run "lowest priority + solver stream waits for the side assembly at once (PS_XCG_AC_WAIT=1)" PS_SIDE_LOWPRIO=1 PS_XCG_AC_WAIT=1
run "lowest priority + solver stream waits for the whole side job (PS_XCG_AC_WAIT=2)" PS_SIDE_LOWPRIO=1 PS_XCG_AC_WAIT=2
run "lowest priority + serial band walk (band_part 0)" PS_SIDE_LOWPRIO=1 HUNT_OPTS=band_part=0
run "lowest priority + dense factorisation (band_chol 0)" PS_SIDE_LOWPRIO=1 HUNT_OPTS=band_chol=0
run "lowest priority + default event flags" PS_SIDE_LOWPRIO=1 PS_EVENT_FLAGS=0
run "lowest priority + host sync after every call" PS_SIDE_LOWPRIO=1 HUNT_SYNC=1
run "lowest priority + one-launch PCG off (xcg_persist 0)" PS_SIDE_LOWPRIO=1 HUNT_OPTS=xcg_persist=0
