"""gpurun_out/<tag>/ (tools/collect_profiles.sh) -> the committed summaries under profiles/:
    <round>_c3_bench.json, <round>_c3_bench_under_rocprof.json, <round>_c3_bench_sharded_1rank_rccl.json,
    <round>_c3_kernel_stats.csv, <round>_c3_pmc_summary.txt, pmc_traffic.json (read by bench.py).
usage: python tools/summarize_profiles.py <tag> [round-prefix, default r01]"""
import csv, json, os, shutil, sys, collections

tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else 'r01'
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, 'gpurun_out', tag), os.path.join(root, 'profiles')
os.makedirs(dst, exist_ok=True)
for a, b in (('bench.json', '_c3_bench.json'), ('bench_under_rocprof.json', '_c3_bench_under_rocprof.json'),
             ('bench_sharded_1rank.json', '_c3_bench_sharded_1rank_rccl.json'), ('ktrace_kernel_stats.csv', '_c3_kernel_stats.csv'),
             ('bench_sharded_1rank_c4.json', '_c4_bench_sharded_1rank_rccl.json'), ('c4_kernel_stats.csv', '_c4_kernel_stats.csv')):
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, rnd + b))

def short(name):
    return name.split('(')[0].strip()

def pmc_table(prefix, title, outname):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(os.listdir(src)):
        if not (f.startswith(prefix) and f.endswith('.csv')):
            continue
        for row in csv.DictReader(open(os.path.join(src, f))):
            if row['Kernel_Name'].startswith('__amd'):
                continue
            acc[short(row['Kernel_Name'])][row['Counter_Name']].append(float(row['Counter_Value']))
    lines = ['# PMC summary, ' + title,
             '# one `rocprofv3 --pmc <set> --kernel-trace` pass per counter set (no other trace domains); per-launch averages.',
             '# FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide loads',
             '# (MI355X_MICROARCH.md, section HBM): the corrected read traffic is 2x FETCH_SIZE.',
             '# TCC_HIT/MISS: L2 requests; TCP_TOTAL_CACHE_ACCESSES: L1 accesses; TCP_TCC_READ_REQ: L1->L2 read requests', '']
    traffic = {}
    for k in sorted(acc):
        for c in sorted(acc[k]):
            v = acc[k][c]
            lines.append('{:36s} {:30s} avg {:14.1f}  launches {}'.format(k[:36], c, sum(v) / len(v), len(v)))
        if 'FETCH_SIZE' in acc[k] and 'WRITE_SIZE' in acc[k]:
            fk = sum(acc[k]['FETCH_SIZE']) / len(acc[k]['FETCH_SIZE'])
            wk = sum(acc[k]['WRITE_SIZE']) / len(acc[k]['WRITE_SIZE'])
            traffic[k] = {'fetch_kb': round(fk, 1), 'write_kb': round(wk, 1), 'hbm_bytes_corrected': int(round((2 * fk + wk) * 1024))}
    if acc:
        open(os.path.join(dst, rnd + outname), 'w').write('\n'.join(lines) + '\n')
    return traffic


traffic = pmc_table('pmc_', 'C3 bench (python bench.py --steps 5 --warmup 2 --no-cpu-baseline)', '_c3_pmc_summary.txt')
traffic_c4 = pmc_table('pmc4_', 'C4 on one GPU (python bench.py --steps 3 --warmup 1 --no-cpu-baseline --kf 2000 --lm 500000)',
                       '_c4_pmc_summary.txt')
if traffic_c4:
    traffic['C4'] = traffic_c4
# the build the passes were taken on: bench.py compares source_sha with the sources it runs and flags a stale table
import subprocess
sha = open(os.path.join(src, 'source_sha.txt')).read().strip() if os.path.exists(os.path.join(src, 'source_sha.txt')) else None
try:
    head = subprocess.check_output(['git', '-C', root, 'rev-parse', '--short=12', 'HEAD']).decode().strip()
except Exception:
    head = None
traffic['_meta'] = {'source_sha': sha, 'git_head': head, 'round': rnd, 'tag': tag,
                    'note': 'per-launch averages of separate rocprofv3 --pmc passes of `bench.py --steps 5 --warmup 2`; read traffic = 2 x FETCH_SIZE (gfx950)'}
json.dump(traffic, open(os.path.join(dst, 'pmc_traffic.json'), 'w'), indent=1, sort_keys=True)
b = json.load(open(os.path.join(src, 'bench.json')))
print('bench', b['value'], b['stage_ms'], b['roofline'], b.get('cpu_baseline', {}).get('value'))
for k in ('void k_schur_pairs_db<0>', 'k_schur_pairs', 'k_schur_combine', 'k_landmark_pass', 'k_pose_pass', 'k_backsub'):
    if k in traffic and k != '_meta':
        print(k, traffic[k])
