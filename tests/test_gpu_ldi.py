"""GPU: the lagged dense inverse of the reduced system as CG preconditioner (option "lagged_inverse", csrc/ps_k_ldi.h,
ps_host_ldi.h).  It only PRECONDITIONS -- every solve still ends at pcg_tol on the current system -- so the tests are:
the same Gauss-Newton trajectory with and without it, steps against the oracle's direct solve while it is active,
and every way out of it (iteration cap, rejected seeds, a problem that jumps) landing on the standard solver."""
import ctypes as C

import numpy as np
import pytest

from oracle import gn_oracle as orc

pytestmark = pytest.mark.gpu


def info(dev):
    from pyslam_amd import _native as nat
    i = nat.ProblemInfo()
    nat.check(dev._lib.ps_get_info(dev._h, C.byref(i)))
    return i.ldi_solves, i.ldi_fallbacks, i.ldi_seeds


def make(lp, on, **opts):
    from pyslam_amd.device import DeviceProblem
    dev = DeviceProblem(lp)
    dev.set_option('lagged_inverse', 1 if on else 0)
    for k, v in opts.items():
        dev.set_option(k, v)
    dev.eval_cost(True)          # as Problem.solve() does: the core knows the cost it starts from
    return dev


def device_dx(dev, lp):
    xp, xl = dev.get_dx()
    return np.concatenate([xp.ravel(), xl.ravel()])


def ba(kf, lm, seed, **kw):
    from pyslam_amd import synthetic
    return synthetic.stereo_ba(num_kf=kf, num_lm=lm, obs_per_lm=6, half_window=8, seed=seed, **kw)[0]


@pytest.mark.parametrize('lam', [0.0, 1e-3])
def test_same_trajectory_with_and_without_the_lagged_inverse(lam):
    """60 keyframes (354 reduced unknowns), 10 Gauss-Newton iterations: cost after every step to 1e-10, final parameters
    to 1e-9; the inverse takes over once the solve has settled and needs a third of the CG iterations."""
    lp = ba(60, 6000, 21)
    a, b = make(lp, True), make(lp, False)
    its_a, its_b = [], []
    for _ in range(10):
        ca, na, ia, ra = a.gn_iteration(lam, 1e-12, 500, True)
        cb, nb, ib, rb = b.gn_iteration(lam, 1e-12, 500, True)
        assert abs(ca - cb) <= 1e-10 * abs(cb)
        assert ra <= 1e-11 and rb <= 1e-11
        its_a.append(ia); its_b.append(ib)
    solves, fallbacks, seeds = info(a)
    assert solves >= 4 and seeds >= 1, (solves, fallbacks, seeds, its_a)
    assert info(b) == (0, 0, 0)
    assert min(its_a) * 2 <= min(its_b), (its_a, its_b)
    pa, pb = a.get_params(), b.get_params()
    assert np.abs(pa[0] - pb[0]).max() < 1e-9 and np.abs(pa[1] - pb[1]).max() < 1e-9


def test_steps_match_the_oracle_while_the_inverse_is_active():
    """Every step of a solve -- the standard ones, the ones beside a seed, the ones preconditioned with the inverse --
    against the oracle's sparse direct solve at the same linearisation point (||dx - dx_ref|| / ||dx_ref|| <= 1e-8)."""
    from pyslam_amd.lowering import LoweredProblem
    lp = ba(30, 1500, 5)
    dev = make(lp, True)
    cur = lp.copy()
    active, first = 0, None
    for it in range(8):
        before = info(dev)[0]
        dx_ref, _ = orc.gauss_newton_step(cur, points_first=False)
        cost, nrm, its, rel = dev.gn_iteration(0., 1e-12, 500, True)
        dx = device_dx(dev, cur)
        first = np.linalg.norm(dx_ref) if first is None else first
        # (relative to the step while it is a step; once the solve has converged -- steps 1e-6 of the first one -- both
        #  solvers resolve it to cond(S) * their residual tolerance, which is what the floor stands for)
        assert np.linalg.norm(dx - dx_ref) <= 1e-8 * max(np.linalg.norm(dx_ref), 1e-4 * first), it
        active += info(dev)[0] - before
        poses, points = dev.get_params()
        cur = cur.copy(); cur.poses, cur.points = poses, points
    assert active >= 3


def test_repeated_linearisation_point():
    """The steady-state bench's pattern (and a damping retry's): the same point linearised again and again.  The inverse is
    seeded from the second call on, takes over two calls later, and the step stays the standard solver's."""
    lp = ba(80, 6000, 3)
    dev = make(lp, True)
    dev.snapshot()
    ref = make(lp, False)
    ref.gn_iteration(0., 1e-12, 500, True)
    dx_ref = device_dx(ref, lp)
    its = []
    for _ in range(8):
        dev.restore()
        out = dev.gn_iteration(0., 1e-12, 500, True)
        its.append(out[2])
        assert np.linalg.norm(device_dx(dev, lp) - dx_ref) <= 1e-9 * np.linalg.norm(dx_ref)
    assert info(dev)[0] >= 4 and its[-1] <= 8 < its[0], its


def test_iteration_cap_falls_back_to_the_standard_solver():
    """ldi_cap = 1: no solve with the inverse can finish, every attempt is given up -- nothing applied -- and the standard
    path solves the same system; the trajectory is the standard one."""
    lp = ba(40, 3000, 8)
    a, b = make(lp, True, ldi_cap=1), make(lp, False)
    for _ in range(9):
        ca = a.gn_iteration(0., 1e-12, 500, True)
        cb = b.gn_iteration(0., 1e-12, 500, True)
        assert abs(ca[0] - cb[0]) <= 1e-10 * abs(cb[0])
    solves, fallbacks, seeds = info(a)
    assert solves == 0 and fallbacks >= 1 and seeds >= 2
    assert np.abs(a.get_params()[0] - b.get_params()[0]).max() < 1e-9


def test_a_jump_of_the_parameters_is_survived():
    """Parameters replaced from outside while an inverse is valid (its cost tag no longer matches, or is unknown): the next
    solve is either the standard one or a capped attempt followed by it -- and correct either way."""
    lp = ba(40, 3000, 9)
    dev = make(lp, True)
    for _ in range(7):
        dev.gn_iteration(0., 1e-12, 500, True)
    assert info(dev)[0] >= 2
    dev.set_params(lp.poses, lp.points)                      # back to the perturbed start: far from where the inverse was built
    dx_ref, _ = orc.gauss_newton_step(lp, points_first=False)
    dev.gn_iteration(0., 1e-12, 500, True)
    assert np.linalg.norm(device_dx(dev, lp) - dx_ref) <= 1e-8 * np.linalg.norm(dx_ref)
    dev.set_params(lp.poses, lp.points)
    dev.eval_cost(True)                                      # ... and with the start cost known: no attempt at all
    before = info(dev)
    dev.gn_iteration(0., 1e-12, 500, True)
    assert np.linalg.norm(device_dx(dev, lp) - dx_ref) <= 1e-8 * np.linalg.norm(dx_ref)
    after = info(dev)
    assert after[0] == before[0] and after[1] == before[1]


@pytest.mark.parametrize('dof', [3, 6])
def test_pose_graphs_keep_their_trajectory(dof):
    """Pose graphs of 120 poses (SE(2): 357 unknowns, SE(3): 714) with Huber loss: long CG solves whose two-level operator
    is a poor seed (seeds may be rejected and backed off) -- whatever the inverse does, the trajectory is the standard one."""
    from pyslam_amd import synthetic, losses
    lp, _ = synthetic.pose_graph(num_poses=120, num_loops=400, dof=dof, seed=4, loss=losses.HuberLoss(1.0))
    a, b = make(lp, True), make(lp, False)
    for _ in range(8):
        ca = a.gn_iteration(0., 1e-13, 2000, True)
        cb = b.gn_iteration(0., 1e-13, 2000, True)
        assert abs(ca[0] - cb[0]) <= 1e-9 * abs(cb[0])
    assert np.abs(a.get_params()[0] - b.get_params()[0]).max() < 1e-8


@pytest.mark.parametrize('dof,poses,loops', [(3, 100, 150), (3, 100, 400), (6, 50, 120), (6, 100, 300), (6, 200, 800)])
def test_direct_seed_takes_pose_graphs_to_a_handful_of_cg_iterations(dof, poses, loops):
    """Pose graphs get the DIRECT seed (fp64 blocked Cholesky + triangular inverse of S on its own stream, ps_host_ldi.h:
    ldi_direct_enqueue): once it is in (a fixed number of calls after the seed, by size) the CG takes <= 8 iterations where
    the two-level operator alone took 50-80, and the trajectory is the one of the solver without any of this."""
    from pyslam_amd import synthetic, losses
    lp, _ = synthetic.pose_graph(num_poses=poses, num_loops=loops, dof=dof, seed=4, loss=losses.HuberLoss(1.0))
    a, b = make(lp, True), make(lp, False)
    its_a, its_b = [], []
    for _ in range(12):
        ca = a.gn_iteration(0., 1e-13, 2000, True)
        cb = b.gn_iteration(0., 1e-13, 2000, True)
        its_a.append(ca[2]); its_b.append(cb[2])
        assert abs(ca[0] - cb[0]) <= 1e-9 * abs(cb[0])
    assert np.abs(a.get_params()[0] - b.get_params()[0]).max() < 1e-8
    solves, fallbacks, seeds = info(a)
    assert seeds >= 1 and solves >= 2 and fallbacks == 0
    assert max(its_a[-2:]) <= 8 and min(its_b[-2:]) >= 5 * max(its_a[-2:]), (its_a, its_b)


def test_direct_seed_is_deterministic_and_switchable():
    """Same problem twice: the same CG iteration counts and bit-identical parameters (the inverse enters at a fixed call,
    not when an event happens to have fired).  ldi_direct = 0 keeps pose graphs on the Newton-Schulz seed / standard solver."""
    from pyslam_amd import synthetic, losses
    lp, _ = synthetic.pose_graph(num_poses=100, num_loops=300, dof=6, seed=9, loss=losses.HuberLoss(1.0))
    runs = []
    for _ in range(2):
        d = make(lp, True)
        its = [d.gn_iteration(0., 1e-12, 2000, True)[2] for _ in range(10)]
        runs.append((its, d.get_params()[0].copy()))
        d.close()
    assert runs[0][0] == runs[1][0]
    assert np.array_equal(runs[0][1], runs[1][1])
    off = make(lp, True, ldi_direct=0)
    ref = make(lp, False)
    for _ in range(10):
        c0 = off.gn_iteration(0., 1e-12, 2000, True); c1 = ref.gn_iteration(0., 1e-12, 2000, True)
        assert abs(c0[0] - c1[0]) <= 1e-9 * abs(c1[0])


def test_direct_seed_on_a_bundle_adjustment_problem():
    """ldi_direct = 1 on a stereo BA problem (whose Newton-Schulz seed works too): steps still match the oracle's direct
    solve of the full system while the directly formed inverse preconditions."""
    lp = ba(40, 3000, 6)
    dev = make(lp, True, ldi_direct=1)
    dev.snapshot()
    ref, _ = orc.gauss_newton_step(lp, points_first=False)
    for k in range(8):
        dev.restore()
        out = dev.gn_iteration(0., 1e-13, 500, True)
        dx = device_dx(dev, lp)
        assert np.linalg.norm(dx - ref) <= 1e-8 * np.linalg.norm(ref), k
    assert info(dev)[0] >= 2 and out[2] <= 6


@pytest.mark.parametrize('kf,limit', [(300, None), (400, 3328)])
def test_lagged_inverse_on_the_largest_systems_it_takes(kf, limit):
    """1 794 unknowns (inside the default limit of 2 048) and 2 394 (option raised to the hard limit, where the refresh is
    off): the same trajectory as the standard solver, and the inverse does engage."""
    from pyslam_amd import synthetic
    lp = synthetic.stereo_ba(num_kf=kf, num_lm=100 * kf, obs_per_lm=8, half_window=12, seed=kf)[0]
    a, b = make(lp, True), make(lp, False)
    if limit:
        a.set_option('ldi_max_unknowns', limit)
    its = []
    for _ in range(8):
        ca = a.gn_iteration(0., 1e-12, 2000, True)
        cb = b.gn_iteration(0., 1e-12, 2000, True)
        assert abs(ca[0] - cb[0]) <= 1e-10 * abs(cb[0]) and ca[3] <= 1e-12
        its.append((ca[2], cb[2]))
    pa, pb = a.get_params(), b.get_params()
    assert np.abs(pa[0] - pb[0]).max() < 1e-9 and np.abs(pa[1] - pb[1]).max() < 1e-9
    solves, fallbacks, seeds = info(a)
    assert solves >= 3 and seeds >= 1, (solves, fallbacks, seeds, its)
    assert its[-1][0] <= 10 and its[-1][1] >= 2 * its[-1][0], its


def test_option_off_and_rebuilt_coarse_level():
    """Switching the option off drops the inverse; changing the coarse level re-lays it out (sizes depend on it)."""
    lp = ba(50, 4000, 2)
    dev = make(lp, True)
    dev.snapshot()
    for _ in range(6):
        dev.restore(); dev.gn_iteration(0., 1e-12, 500, True)
    s0 = info(dev)[0]
    assert s0 >= 1
    dev.set_option('lagged_inverse', 0)
    dev.restore(); dev.gn_iteration(0., 1e-12, 500, True)
    assert info(dev)[0] == s0
    dev.set_option('lagged_inverse', 1)
    dev.set_option('coarse_groups', 5)
    ref = make(lp, False)
    ref.gn_iteration(0., 1e-12, 500, True)
    for _ in range(6):
        dev.restore(); dev.gn_iteration(0., 1e-12, 500, True)
        assert np.linalg.norm(device_dx(dev, lp) - device_dx(ref, lp)) <= 1e-9 * np.linalg.norm(device_dx(ref, lp))
    assert info(dev)[0] > s0


# ---------------------------------------------------------------------------------------------------------------------
# one-launch explicit two-level PCG (option "xcg_fused", csrc/ps_k_xcg.h: k_xcg_fused1)
# ---------------------------------------------------------------------------------------------------------------------
def xf_info(dev):
    from pyslam_amd import _native as nat
    i = nat.ProblemInfo()
    nat.check(dev._lib.ps_get_info(dev._h, C.byref(i)))
    return i.xcg_fused_solves, i.xcg_fused_fallbacks


@pytest.mark.parametrize('mode', [1, 2])
@pytest.mark.parametrize('kind', ['pose_graph', 'ba'])
def test_one_launch_explicit_pcg_matches_the_three_launch_form(kind, mode):
    """(mode 2: the TWO-launch form -- scalars, t and y = A_c^-1 t once in k_xcg_f2_coarse -- that coarse levels beyond 2 048
    unknowns get.)  The single-reduction (Chronopoulos-Gear) arrangement of the explicit two-level PCG in ONE launch per iteration against
    the three-launch form: same preconditioner, same iteration counts (+-1), the same Gauss-Newton trajectory (cost 1e-10,
    parameters 1e-9), and the first step against the oracle's direct solve."""
    from pyslam_amd import synthetic, losses
    from pyslam_amd.device import DeviceProblem
    if kind == 'pose_graph':
        lp, _ = synthetic.pose_graph(num_poses=600, num_loops=2401, dof=6, seed=2, loss=losses.HuberLoss(1.0))
    else:
        lp, _ = synthetic.stereo_ba(num_kf=700, num_lm=30000, obs_per_lm=6, half_window=8, seed=12)
    a, b = DeviceProblem(lp), DeviceProblem(lp)
    a.set_option('xcg_fused', mode)
    b.set_option('xcg_fused', 0)
    dx_ref, _ = orc.gauss_newton_step(lp, points_first=False)
    for it in range(5):
        ra = a.gn_iteration(0., 1e-13, 3000, True)
        rb = b.gn_iteration(0., 1e-13, 3000, True)
        if it == 0:
            for d in (a, b):
                assert np.linalg.norm(device_dx(d, lp) - dx_ref) <= 1e-8 * np.linalg.norm(dx_ref)
        assert abs(ra[0] - rb[0]) <= 1e-10 * abs(rb[0])
        assert abs(ra[2] - rb[2]) <= 2 and ra[3] <= 1e-12
    used, fell = xf_info(a)
    assert used >= 5 and fell == 0 and xf_info(b) == (0, 0)
    pa, pb = a.get_params(), b.get_params()
    assert np.abs(pa[0] - pb[0]).max() < 1e-9
    if pa[1].size:
        assert np.abs(pa[1] - pb[1]).max() < 1e-9


def test_one_launch_explicit_pcg_through_the_staged_api_and_covariance():
    """ps_solve_reduced (synchronous driver) and a covariance column (unit right-hand side) through the one-launch form."""
    from pyslam_amd import synthetic, losses
    from pyslam_amd.device import DeviceProblem
    lp, _ = synthetic.pose_graph(num_poses=600, num_loops=2401, dof=6, seed=5, loss=losses.HuberLoss(1.0))
    a, b = DeviceProblem(lp), DeviceProblem(lp)
    b.set_option('xcg_fused', 0)
    for d in (a, b):
        d.linearize(0.)
        d.solve_reduced(1e-13, 3000)
        d.backsub()
    xa, xb = a.get_dx()[0], b.get_dx()[0]
    assert np.linalg.norm(xa - xb) <= 1e-9 * np.linalg.norm(xb)
    assert xf_info(a)[0] >= 1


def test_two_launch_explicit_pcg_on_a_coarse_level_too_wide_for_one_launch():
    """3 000 SE(3) poses with 400 coarse intervals asked for: 2 406 coarse unknowns, beyond the one-launch kernel's 2 048 --
    the default picks the two-launch form by itself; trajectory against the three-launch form."""
    from pyslam_amd import synthetic
    from pyslam_amd.device import DeviceProblem
    lp, _ = synthetic.pose_graph(num_poses=3000, num_loops=12001, dof=6, seed=3)
    a, b = DeviceProblem(lp), DeviceProblem(lp)
    for d in (a, b):
        d.set_option('coarse_groups', 400)
    b.set_option('xcg_fused', 0)
    for it in range(4):
        ra = a.gn_iteration(0., 1e-12, 4000, True)
        rb = b.gn_iteration(0., 1e-12, 4000, True)
        assert abs(ra[0] - rb[0]) <= 1e-10 * abs(rb[0])
        assert abs(ra[2] - rb[2]) <= 3 and ra[3] <= 1e-11
    used, fell = xf_info(a)
    assert used >= 4 and fell == 0 and xf_info(b) == (0, 0)
    assert np.abs(a.get_params()[0] - b.get_params()[0]).max() < 1e-9


def test_coarse_inverse_is_held_at_a_repeated_linearisation_point():
    """Explicit two-level PCG: a call that linearises where the inverse in use was formed (restore + iterate, a damping
    retry with another lambda) holds it -- no factorisation beside the solve, same preconditioner.  Against the same calls
    with the hold switched off (coarse_auto_hold = 0) and, for the first step, the oracle's Schur solve."""
    lp = ba(600, 15000, 12)
    from pyslam_amd.device import DeviceProblem
    a, b = DeviceProblem(lp), DeviceProblem(lp)
    b.set_option('coarse_auto_hold', 0)
    for d in (a, b):
        d.eval_cost(True); d.snapshot()
    dx_ref, _ = orc.gauss_newton_step(lp, points_first=False, linear_solver='schur')
    its = []
    for k, lam in enumerate([0., 0., 0., 1e-3, 1e-3, 0.]):
        outs = []
        for d in (a, b):
            d.restore()
            outs.append(d.gn_iteration(lam, 1e-13, 3000, False))
        xa, xb = device_dx(a, lp), device_dx(b, lp)
        assert np.linalg.norm(xa - xb) <= 1e-9 * np.linalg.norm(xb), (k, lam)
        assert outs[0][3] <= 1e-12 and abs(outs[0][2] - outs[1][2]) <= 2
        if k == 0:
            assert np.linalg.norm(xa - dx_ref) <= 1e-8 * np.linalg.norm(dx_ref)
        its.append(outs[0][2])
    assert max(its[:3]) - min(its[:3]) <= 1, its
