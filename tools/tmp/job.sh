mkdir -p gpurun_out/x15
timeout 1200 python -m pytest tests/test_gpu_edges.py -x -q -k "ended_by_the_last" 2>&1 | grep -E "passed|failed|Error|assert" | tail -8 > gpurun_out/x15/t.txt; cat gpurun_out/x15/t.txt
python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys
b=json.loads(sys.stdin.read()); print('C3', b['value'], b['stage_ms'], b.get('trajectory_ms_per_iter',{}).get('per_iter')); c=b['c4_single_gpu']; print('C4', c['ms'], c['stage_ms'], c['trajectory_ms_per_iter']); print(b.get('c5_solve_wall_ms'))" > gpurun_out/x15/bench.txt; cat gpurun_out/x15/bench.txt
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -3 > gpurun_out/x15/suite.txt; cat gpurun_out/x15/suite.txt
