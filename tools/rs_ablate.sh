for a in 0 1 2 4 8 15; do
  export PS_RS_ABLATE=$a
  bash tools/ktrace.sh rsab > /dev/null 2>&1
  echo "ablate $a: $(grep k_rows_setup gpurun_out/rsab/kernel_stats.csv | cut -d, -f14-17 | tail -c 60)"
  python - <<PY
import csv
for r in csv.DictReader(open('gpurun_out/rsab/kernel_stats.csv')):
    if 'k_rows_setup' in r['Name'] or 'mreduce' in r['Name']: print('   ', r['Name'][:30], r['Calls'], r['AverageNs'])
PY
done
