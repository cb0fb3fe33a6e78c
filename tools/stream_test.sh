mkdir -p gpurun_out/st
PS_SCHUR_STREAM=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_edges.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -12
for m in 0 1; do
PS_SCHUR_STREAM=$m PS_CREATE_TIMING=1 python bench.py --no-cpu-baseline 2> gpurun_out/st/err$m.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('stream $m', d['value'], d['stage_ms'], 'c4', d['c4_single_gpu']['ms'], d['c4_single_gpu']['stage_ms'])
"
grep "streaming\|pair generation\|streaming Schur lists" gpurun_out/st/err$m.log | head -6
done
