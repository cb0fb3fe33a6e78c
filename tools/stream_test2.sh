mkdir -p gpurun_out/st
for m in 0 1; do
PS_SCHUR_STREAM=$m PS_CREATE_TIMING=1 python bench.py --no-cpu-baseline 2> gpurun_out/st/err$m.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('stream $m', d['value'], d['stage_ms'], 'c4', d['c4_single_gpu']['ms'], d['c4_single_gpu']['stage_ms'])
"
grep "streaming Schur" gpurun_out/st/err$m.log | head -6
done
for t in 128 199 512; do
PS_ST_TILES=$t PS_SCHUR_STREAM=1 PS_CREATE_TIMING=1 python bench.py --no-cpu-baseline --no-c4 2> gpurun_out/st/errt.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('tiles $t', d['value'], d['stage_ms']['schur_pairs'])
"
grep "streaming Schur:" gpurun_out/st/errt.log | head -2
done
bash tools/stream_ablate.sh | grep "^ablate"
