mkdir -p gpurun_out/r02h
(time python -m pytest tests -m gpu -x -q 2>&1 | tail -30) > gpurun_out/r02h/tests.log 2>&1
grep -n "passed\|failed" gpurun_out/r02h/tests.log
python tools/c2_probe.py 2>&1 | grep -v amdgpu.ids | tail -12
