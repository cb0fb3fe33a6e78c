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
