mkdir -p gpurun_out/r02b
(time python -m pytest tests -m gpu -x -q -s 2>&1 | tail -60) > gpurun_out/r02b/tests.log 2>&1
for kb in 9216 4096 3072 2048; do
  echo "tile_kb $kb" >> gpurun_out/r02b/tiles.log
  PS_SCHUR_TILE_KB=$kb python bench.py --no-cpu-baseline --no-c4 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], d['stage_ms'], d['roofline']['frac'])
" >> gpurun_out/r02b/tiles.log
done
python bench.py --no-cpu-baseline > gpurun_out/r02b/bench.json 2> gpurun_out/r02b/bench.err
tail -5 gpurun_out/r02b/tests.log; cat gpurun_out/r02b/tiles.log
