# everything the driver runs at round end, on the GPU box: gpu tests, smoke, bench
mkdir -p gpurun_out/final
(time python -m pytest tests -x -q -m gpu 2>&1 | tail -14) > gpurun_out/final/tests.log 2>&1
grep -n "passed\|failed" gpurun_out/final/tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
(time python bench.py) > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
tail -3 gpurun_out/final/bench.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/final/bench.json') if l.startswith('{')][-1])
print(d['value'], d['stage_ms'], d['roofline']['frac'], d['roofline']['traffic_stale'], d['c4_single_gpu_ms'], d['c4_single_gpu']['stage_ms'], d['cpu_baseline']['value'], d['cpu_baseline']['spsolve_full_s'])
PY
