import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A test that hangs must end as ONE failed test, not as a suite that runs into the box's limit: every test gets a
    deadline when pytest-timeout is there (it is in this image), unless the command line set one."""
    if not config.pluginmanager.hasplugin('timeout') or getattr(config.option, 'timeout', None):
        return
    for item in items:
        if item.get_closest_marker('timeout') is None:
            item.add_marker(pytest.mark.timeout(600))


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False))


def golden_lp(g):
    """Rebuild the LoweredProblem stored in a golden file."""
    from pyslam_amd.lowering import LoweredProblem
    lp = LoweredProblem(dof=int(g['lp_dof']))
    for k, v in g.items():
        if k.startswith('lp_') and k != 'lp_dof':
            setattr(lp, k[3:], np.array(v))
    return lp.finalize()


def golden_options(g):
    out = {}
    for k, v in g.items():
        if k.startswith('opt_'):
            out[k[4:]] = float(v)
    for k in ('max_iters', 'linesearch_max_iters', 'max_nondecreasing_steps', 'num_threads'):
        if k in out:
            out[k] = int(out[k])
    if 'allow_nondecreasing_steps' in out:
        out['allow_nondecreasing_steps'] = bool(out['allow_nondecreasing_steps'])
    return out


SOLVE_CASES = ['stereo_ba_example', 'posegraph_2d_example', 'posegraph_3d_example',
               'ba_tiny_huber', 'ba_tiny_nolinesearch', 'ba_small', 'ba_8k', 'pg_small_huber',
               'pg2d_small_huber', 'motion_only_cauchy', 'pg_orientation_huber']


def rel_err(a, b, floor=0.):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return np.linalg.norm((a - b).ravel()) / (np.linalg.norm(b.ravel()) + floor)
