"""The folded two-level CG in ONE launch (csrc/ps_k_cg_persist.h, option cg_persist, default on where the augmented system fits
its layout): matrix in registers, vectors replicated per workgroup, one in-launch exchange of tagged granules per iteration.
Held against the launch-per-iteration kernels it stands in for (cg_persist 0: k_cg_fused_lds / k_cg_fused) -- same recurrences, so
the same iteration counts (a sum taken in another order may move a count by one), the same step to the solver's tolerance --
against the oracle's sparse direct solve (reference pyslam/problem.py:186), and through its failure path: an exchange that
times out is a breakdown the host answers with the other kernels."""
import numpy as np
import pytest

from oracle import gn_oracle as orc
from pyslam_amd import synthetic, losses

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


CASES = [
    ('ba40', lambda: synthetic.stereo_ba(num_kf=40, num_lm=3000, obs_per_lm=10, half_window=8, seed=2)[0]),
    ('ba120_huber', lambda: synthetic.stereo_ba(num_kf=120, num_lm=9000, obs_per_lm=8, half_window=14, seed=5, loss=losses.HuberLoss(2.0))[0]),
    ('ba200', lambda: synthetic.stereo_ba(num_kf=200, num_lm=12000, obs_per_lm=10, half_window=20, seed=0)[0]),
    ('ba_with_edges', lambda: synthetic.with_pose_edges(synthetic.stereo_ba(num_kf=60, num_lm=4000, obs_per_lm=6, half_window=9, seed=8)[0], 40, 9)),
    # pose graphs: short rows, dense coarse rows cut into many tasks (300 poses: 517 tasks -> the instantiation with 12 sums per thread)
    ('se3_graph_120', lambda: synthetic.pose_graph(num_poses=120, num_loops=90, dof=6, seed=3)[0]),
    ('se3_graph_300', lambda: synthetic.pose_graph(num_poses=300, num_loops=200, dof=6, seed=2)[0]),
    ('se2_graph_200', lambda: synthetic.pose_graph(num_poses=200, num_loops=150, dof=3, seed=4)[0]),
    ('ba260', lambda: synthetic.stereo_ba(num_kf=260, num_lm=8000, obs_per_lm=10, half_window=20, seed=1)[0]),
]


@pytest.mark.parametrize('name,make', CASES, ids=[c[0] for c in CASES])
def test_one_launch_cg_equals_the_launch_per_iteration_kernels(name, make):
    from pyslam_amd.device import DeviceProblem
    lp = make()
    out = {}
    for persist in (1, 0):
        dev = DeviceProblem(lp)
        dev.set_option('cg_persist', persist)
        dev.set_option('lagged_inverse', 0)
        if lp.num_reduced > 250:                                     # (without the lagged inverse the explicit PCG takes over from 250 poses)
            dev.set_option('cg_explicit_min_rows', 100000)
        dev.linearize(0.0)
        its, relres = dev.solve_reduced(1e-13, 4000)
        dev.backsub()
        xp, xl = dev.get_dx()
        trace = [dev.gn_iteration(0.0, 1e-12, 4000, True) for _ in range(3)]
        if lp.num_obs == 0:
            xl = np.zeros((0, 3))
        out[persist] = (its, relres, xp, xl, trace, dev.get_params(), dev.cg_persist_counts())
        dev.close()
    a, b = out[1], out[0]
    assert a[6][0] >= 4 and a[6][1] == 0 and b[6][0] == 0            # the one-launch form ran (staged solve + three iterations), never failed
    assert abs(a[0] - b[0]) <= max(1, a[0] // 50) and a[1] <= 1e-13 * 1.001
    assert rel(a[2], b[2]) <= 1e-9 and (a[3].size == 0 or rel(a[3], b[3]) <= 1e-9)
    for ta, tb in zip(a[4], b[4]):
        assert abs(ta[0] - tb[0]) <= 1e-10 * abs(tb[0]) + 1e-18 and abs(ta[2] - tb[2]) <= max(1, tb[2] // 50)
    assert np.abs(a[5][0] - b[5][0]).max() <= 1e-9 and (a[5][1].size == 0 or np.abs(a[5][1] - b[5][1]).max() <= 1e-8)
    if lp.num_reduced <= 60:                                         # (the sparse direct solve of the larger ones takes minutes on the host)
        dxo, _ = orc.gauss_newton_step(lp, points_first=False)
        assert rel(np.concatenate([a[2].ravel(), a[3].ravel()]), dxo) <= 1e-8


def test_one_launch_cg_is_deterministic_and_survives_many_solves():
    """Tags are (launch counter, iteration): nothing of an earlier solve may pass for the current one.  Forty solves on one
    handle, the same answer bit for bit every time."""
    from pyslam_amd.device import DeviceProblem
    lp, _ = synthetic.stereo_ba(num_kf=60, num_lm=4000, obs_per_lm=8, half_window=10, seed=4)
    start = (lp.poses.copy(), lp.points.copy())
    dev = DeviceProblem(lp)
    dev.set_option('lagged_inverse', 0)
    ref = None
    for rep in range(40):
        dev.reset_solver_state(); dev.set_params(*start)
        got = dev.gn_iteration(0.0, 1e-12, 2000, True)
        if ref is None:
            ref = (got, dev.get_params())
        else:
            assert got == ref[0]
            p = dev.get_params()
            assert np.array_equal(p[0], ref[1][0]) and np.array_equal(p[1], ref[1][1])
    assert dev.cg_persist_counts() == (40, 0)
    dev.close()


def test_a_timed_out_exchange_is_answered_by_the_other_kernels():
    """cg_persist_spin 0: every workgroup gives up on its first unsuccessful pass over the exchange -- the call must still
    return the right step (solved again by the launch-per-iteration kernels), count the failure, and stop using the
    one-launch form on the handle."""
    from pyslam_amd.device import DeviceProblem
    lp, _ = synthetic.stereo_ba(num_kf=60, num_lm=4000, obs_per_lm=8, half_window=10, seed=4)
    ref = DeviceProblem(lp)
    ref.set_option('cg_persist', 0); ref.set_option('lagged_inverse', 0)
    want = [ref.gn_iteration(0.0, 1e-12, 2000, True) for _ in range(3)]
    want_p = ref.get_params()
    ref.close()
    dev = DeviceProblem(lp)
    dev.set_option('lagged_inverse', 0)
    dev.set_option('cg_persist_spin', 0)
    got = [dev.gn_iteration(0.0, 1e-12, 2000, True) for _ in range(3)]
    solves, failures = dev.cg_persist_counts()
    p = dev.get_params()
    dev.close()
    for a, b in zip(got, want):
        assert abs(a[0] - b[0]) <= 1e-10 * abs(b[0])
    assert np.abs(p[0] - want_p[0]).max() <= 1e-9 and np.abs(p[1] - want_p[1]).max() <= 1e-8
    # (a pass may succeed at once when every workgroup happens to have published already: then nothing times out)
    assert failures <= 1 and (failures == 0 or solves == 1)


def test_covariance_columns_through_the_one_launch_cg():
    from pyslam_amd.device import DeviceProblem
    lp, _ = synthetic.stereo_ba(num_kf=30, num_lm=600, obs_per_lm=5, half_window=6, seed=3)
    cols = {}
    for persist in (1, 0):
        dev = DeviceProblem(lp)
        dev.set_option('cg_persist', persist)
        dev.covariance_begin()
        cols[persist] = [dev.covariance_column(kind, idx, comp) for kind, idx, comp in ((0, 0, 0), (0, 7, 4), (1, 33, 2))]
        dev.close()
    for a, b in zip(cols[1], cols[0]):
        assert rel(a[0], b[0]) <= 1e-8 and rel(a[1], b[1]) <= 1e-8


XCASES = [
    ('ba600', lambda: synthetic.stereo_ba(num_kf=600, num_lm=9000, obs_per_lm=10, half_window=20, seed=3)[0]),
    ('ba1000_huber', lambda: synthetic.stereo_ba(num_kf=1000, num_lm=12000, obs_per_lm=8, half_window=14, seed=4, loss=losses.HuberLoss(2.0))[0]),
]


@pytest.mark.parametrize('name,make', XCASES, ids=[c[0] for c in XCASES])
def test_one_launch_explicit_pcg_equals_the_launch_per_iteration_form(name, make):
    """The explicit two-level PCG of larger bundle adjustments (csrc/ps_k_xcg_persist.h, option xcg_persist): every iteration of
    a solve in one launch, the matrix in registers / LDS, one in-launch exchange per iteration -- against k_xcg_fused1 launched
    once per iteration (xcg_persist 0): same recurrences, same order of the sums, so the same iteration counts and the same step
    to the solver's tolerance; then with the exchange made to time out: repeated launch by launch, counted, not used again."""
    from pyslam_amd.device import DeviceProblem
    lp = make()
    out = {}
    for mode in ('persist', 'per_iteration', 'timeout'):
        dev = DeviceProblem(lp)
        dev.set_option('xcg_persist', 0 if mode == 'per_iteration' else 1)
        if mode == 'timeout':
            dev.set_option('cg_persist_spin', 0)
        trace = [dev.gn_iteration(0.0, 1e-12, 2000, True) for _ in range(3)]
        out[mode] = (trace, dev.get_params(), dev.cg_persist_counts())
        dev.close()
    a, b, c = out['persist'], out['per_iteration'], out['timeout']
    assert a[2][0] == 3 and a[2][1] == 0 and b[2] == (0, 0)
    assert c[2][1] <= 1 and (c[2][1] == 0 or c[2][0] == 1)   # (a pass over the exchange may succeed at once: then nothing times out)
    for ta, tb, tc in zip(a[0], b[0], c[0]):
        assert abs(ta[0] - tb[0]) <= 1e-10 * abs(tb[0]) and abs(ta[2] - tb[2]) <= 1
        assert abs(tc[0] - tb[0]) <= 1e-10 * abs(tb[0])
    for other in (b, c):
        assert np.abs(a[1][0] - other[1][0]).max() <= 1e-9 and np.abs(a[1][1] - other[1][1]).max() <= 1e-8


def test_four_wave_form_of_the_one_launch_explicit_pcg_gives_the_same_bits():
    """Round 6, option xcg_persist4 (off by default: measured slower, HISTORY round 6): k_xcg_persist as four waves per workgroup --
    one wave per SIMD, 512 registers per lane, the workgroup's rows of the fp32 coarse inverse resident in registers
    (csrc/ps_k_xcg_persist4.h).  Same recurrences, same order of every sum: costs, iteration counts and parameters are those of
    the eight-wave kernel bit for bit."""
    from pyslam_amd.device import DeviceProblem
    lp = synthetic.stereo_ba(num_kf=1000, num_lm=30000, obs_per_lm=10, half_window=20, seed=3)[0]
    out = {}
    for four in (1, 0):
        dev = DeviceProblem(lp)
        dev.set_option('xcg_persist4', four)
        trace = [dev.gn_iteration(0.0, 1e-12, 2000, True) for _ in range(3)]
        out[four] = (trace, dev.get_params(), dev.get_info())
        dev.close()
    a, b = out[1], out[0]
    assert a[2]['xcg_persist4_solves'] == 3 and a[2]['cg_persist_failures'] == 0
    assert b[2]['xcg_persist4_solves'] == 0 and b[2]['cg_persist_solves'] == 3
    for ta, tb in zip(a[0], b[0]):
        assert ta[0] == tb[0] and ta[2] == tb[2]
    assert np.array_equal(a[1][0], b[1][0]) and np.array_equal(a[1][1], b[1][1])


def test_pipelined_recurrences_give_the_same_solve_on_a_well_conditioned_system():
    """Round 6, option cg_pipelined (off by default: DESIGN.md section 5): the one-launch CG with the pipelined recurrences -- products of
    w_k published first, the dot products formed while the exchange is in flight -- is the same Krylov iteration: on a bundle
    adjustment (preconditioned condition number ~3) the same iteration counts, the step to 1e-9, the trajectory to 1e-10."""
    from pyslam_amd.device import DeviceProblem
    lp, _ = synthetic.stereo_ba(num_kf=200, num_lm=12000, obs_per_lm=10, half_window=20, seed=0)
    out = {}
    for pipe in (2, 0):
        dev = DeviceProblem(lp)
        dev.set_option('cg_pipelined', pipe)
        dev.set_option('lagged_inverse', 0)
        dev.linearize(0.0)
        its, relres = dev.solve_reduced(1e-13, 4000)
        dev.backsub()
        xp, xl = dev.get_dx()
        trace = [dev.gn_iteration(0.0, 1e-12, 4000, True) for _ in range(3)]
        out[pipe] = (its, relres, xp, xl, trace, dev.cg_persist_counts())
        dev.close()
    a, b = out[2], out[0]
    assert a[5][0] >= 4 and a[5][1] == 0
    assert abs(a[0] - b[0]) <= 1 and a[1] <= 1e-13 * 1.001
    assert rel(a[2], b[2]) <= 1e-9 and rel(a[3], b[3]) <= 1e-9
    for ta, tb in zip(a[4], b[4]):
        assert abs(ta[0] - tb[0]) <= 1e-10 * abs(tb[0]) and abs(ta[2] - tb[2]) <= 1
