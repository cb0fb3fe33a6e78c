"""bench.py's bookkeeping, without a GPU: the byte model of SURVEY.md section 8d behind `roofline`, the options of the timed solves
(reference examples/stereo_ba.py:38-40), the PMC table the `traffic` field is read from, and the tie between the line and the
binary (ps_build_sha == hash of the sources on disk).  Host logic only."""
import json
import os

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_byte_model_is_section_8d():
    import bench
    info = dict(num_obs=500000, num_var_points=50000, num_reduced=199, reduced_nnzb=14479)
    b_iter, b_schur, b_spmv = bench.algorithmic_bytes(info, 20)
    assert b_schur == 144 * 500000 + 288 * (14479 - 199) == 76112640          # the figure the roofline line carries for C3
    assert b_spmv == 288 * 14479 + 3 * 48 * 199
    assert b_iter == 496 * 500000 + 264 * 50000 + 960 * 199 + 288 * 14479 * 21


def test_timed_solves_run_under_the_reference_examples_options():
    import bench
    opt = bench.example_options()
    assert opt.allow_nondecreasing_steps is True and opt.max_nondecreasing_steps == 3       # examples/stereo_ba.py:38-40
    assert opt.max_iters == 100 and opt.min_cost_decrease == 0.9                            # reference defaults otherwise
    assert bench.C3 == dict(num_kf=200, num_lm=50000, seed=0) and bench.C4['num_kf'] == 2000 and bench.C4['num_lm'] == 500000


def test_line_is_tied_to_the_binary_and_the_pmc_table_to_the_sources():
    """The library on disk carries the hash of the sources it was built from; the committed PMC table names the sources its counter
    passes ran on -- when that is not the current hash the bench line says `traffic_stale` instead of pretending."""
    import bench
    import __graft_entry__ as ge
    assert bench.kernel_source_sha() == ge.source_sha()
    with open(os.path.join(REPO, 'profiles', 'pmc_traffic.json')) as f:
        table = json.load(f)
    total, sha, head = bench.pmc_traffic('k_schur_pairs_db')
    assert total is not None and total > 76112640 and sha == table['_meta']['source_sha']
    total4, _, _ = bench.pmc_traffic('k_schur_pairs_db', 'C4')
    assert total4 is not None and total4 > 10 * total * 0.5
    assert bench.pmc_traffic('no such kernel') == (None, None, None)


def test_committed_bench_line_has_the_contract_fields():
    with open(os.path.join(REPO, 'profiles', 'r04_c3_bench.json')) as f:
        line = json.load(f)
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in line, key
    r = line['roofline']
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3
    assert abs(r['achieved'] - r['algorithmic_bytes_per_launch'] / (r['avg_launch_ms'] * 1e-3) / 1e9) < 1.0
    assert line['cpu_baseline']['kind'] == 'port' and line['cpu_baseline']['cores'] == 1
    assert line['higher_is_better'] is False and line['dtype'] == 'f64' and 'workload' in line['config']
    cs = line['cold_solve']
    assert cs['iterations'] == line['steps'] and abs(np.sum(cs['per_solve_ms']) / cs['iterations'] - line['value']) < 2e-3


def test_roofline_names_the_candidate_with_the_largest_time():
    """`roofline.kernel` is chosen by measurement (round-5 verdict): the event pair with the largest time per iteration."""
    import bench
    c3 = {'schur_pairs': 0.047, 'cg_kernel': 0.115, 'landmark_pass': 0.030, 'pose_pass': 0.027, 'backsub': 0.022, 'pcg': 0.167}
    assert bench.pick_dominant(c3) == ('cg_kernel', 0.115)
    c4 = {'schur_pairs': 0.529, 'cg_kernel': 0.354, 'landmark_pass': 0.249, 'pose_pass': 0.184, 'backsub': 0.173}
    assert bench.pick_dominant(c4)[0] == 'schur_pairs'
    assert bench.pick_dominant({'backsub': 0.5, 'schur_pairs': 0.1})[0] == 'backsub'
    assert bench.cg_kernel_name(True, False) == 'k_cg_persist' and bench.cg_kernel_name(True, True) == 'k_xcg_persist'
    assert bench.cg_kernel_name(False, True) == 'k_xcg_fused1' and bench.cg_kernel_name(False, False) == 'k_cg_fused_lds'
    info = dict(num_obs=500000, num_var_points=50000, num_reduced=199, reduced_nnzb=14479)
    r = bench.roofline_objects(info, c3, None, 19, True, False)
    assert r['roofline']['kernel'] == 'k_cg_persist' and r['roofline']['bound'] == 'latency'
    assert r['roofline']['algorithmic_bytes_per_launch'] == 288 * 14479 * 19          # SURVEY 8d's PCG term
    assert abs(r['roofline']['frac'] - r['roofline']['achieved'] / 8000.) < 1e-3
    assert abs(r['roofline']['achieved'] - 288 * 14479 * 19 / 0.115e-3 / 1e9) < 1.0
    assert r['roofline']['latency_ceiling']['floor_ms'] == round(19 * 2.0e-3, 5)
    assert r['roofline_schur']['kernel'] == 'k_schur_pairs_db' and r['roofline_schur']['algorithmic_bytes_per_launch'] == 76112640
    assert bench.roofline_objects(info, c4, 'C4', 20, True, True)['roofline']['kernel'] == 'k_schur_pairs_db'


ITERATION_KERNELS = {'k_schur_pairs_db': 'k_schur_pairs_db', 'k_schur_combine': 'k_schur_pairs_db', 'k_cg_persist': 'k_cg_persist',
                     'k_xcg_persist': 'k_xcg_persist', 'k_cg_fused_lds': 'k_cg_fused_lds', 'k_xcg_fused1': 'k_xcg_fused1',
                     'k_landmark_pass_packed': 'k_landmark_pass_packed', 'k_pose_pass': 'k_pose_pass', 'k_backsub_packed': 'k_backsub_packed'}


def _largest_iteration_kernel(csv_path):
    """From a `rocprofv3 --kernel-trace --stats` summary of the bench command: the kernel of the Gauss-Newton iteration with the
    largest TOTAL time (per-iteration kernels of the CG summed over their launches; pair + combine kernel together, as the bench
    line times them).  Kernels of the other legs of the command (C5 frames, create-time sorts) are not in the table above."""
    import csv
    total = {}
    with open(csv_path) as f:
        for row in csv.DictReader(f):
            name = row['Name'].replace('void ', '')
            for prefix, group in ITERATION_KERNELS.items():
                if name.startswith(prefix):
                    total[group] = total.get(group, 0.0) + float(row['TotalDurationNs'])
                    break
    return max(total, key=total.get), total


def test_committed_line_names_the_top_kernel_of_the_committed_kernel_stats():
    """The round-6 line's `roofline.kernel` (chosen live from event pairs) against the rocprofv3 summary committed beside it:
    they must name the same kernel, at C3 (the bench line) and at C4 (its c4_single_gpu leg, stats of its own)."""
    import pytest
    path = os.path.join(REPO, 'profiles', 'r06_c3_bench.json')
    if not os.path.exists(path):
        pytest.skip('round-6 profiles not collected yet')
    with open(path) as f:
        line = json.load(f)
    top, total = _largest_iteration_kernel(os.path.join(REPO, 'profiles', 'r06_c3_kernel_stats.csv'))
    assert line['roofline']['kernel'] == top, (line['roofline']['kernel'], total)
    assert line['roofline_candidates_ms'] and line['roofline_schur']['kernel'] == 'k_schur_pairs_db'
    assert 'value_median' in line and line['value_median']['ms_per_iteration_median_of_solves'] > 0
    top4, total4 = _largest_iteration_kernel(os.path.join(REPO, 'profiles', 'r06_c4_kernel_stats.csv'))
    assert line['c4_single_gpu']['roofline']['kernel'] == top4, (line['c4_single_gpu']['roofline']['kernel'], total4)
