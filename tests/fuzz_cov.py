"""Randomised sweep of the covariance columns: seeded random small problems under random solver
modes; ps_covariance_begin + ps_covariance_column (Schur-form right-hand side, reduced solve, back-substitution) against
the oracle's normal matrix: column k must solve P x = e_k.  usage: python tests/fuzz_cov.py [cases] [seed0]"""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scipy.sparse.linalg as spla
from oracle import gn_oracle as orc
from pyslam_amd import synthetic, losses
from pyslam_amd.device import DeviceProblem

LOSSES = [lambda: losses.L2Loss(), lambda: losses.HuberLoss(1.5), lambda: losses.CauchyLoss(3.0)]


def run(n_cases, seed0=0, verbose=True):
    bad = 0
    t0 = time.time()
    for case in range(seed0, seed0 + n_cases):
        rng = np.random.default_rng(9000 + case)
        loss = LOSSES[rng.integers(len(LOSSES))]()
        kind = rng.choice(['ba', 'ba', 'pg3', 'pg2'])
        if kind == 'ba':
            kf, obs = int(rng.choice([3, 6, 12, 20, 40])), int(rng.integers(2, 5))
            lp, truth = synthetic.stereo_ba(num_kf=kf, num_lm=int(rng.integers(30 * kf // obs, 30 * kf // obs + 60)), obs_per_lm=min(obs, kf),
                                        half_window=int(rng.integers(obs, 2 * obs + 3)), seed=case, loss=loss,
                                        const_point_fraction=float(rng.choice([0., 0.2])))
            if kf >= 3 and rng.integers(3) == 0:        # + pose-pose edges and a prior over the same keyframes
                lp = synthetic.with_pose_edges(lp, int(rng.integers(0, 2 * kf)), case + 1, loss=loss, truth_poses=truth['poses'])
        else:
            P = int(rng.choice([5, 14, 17, 40, 120, 300]))
            lp, _ = synthetic.pose_graph(num_poses=P, num_loops=int(rng.integers(1, 3 * P)), dof=6 if kind == 'pg3' else 3,
                                         seed=case, loss=loss)
        mode = rng.choice(['auto', 'auto', 'explicit', 'nodirect', 'G'])
        dev = DeviceProblem(lp)
        if mode == 'explicit':
            dev.set_option('cg_explicit_min_rows', 0); dev.set_option('cg_split_min_rows', 0)
        elif mode == 'nodirect':
            dev.set_option('direct_max_unknowns', 0)
        elif mode == 'G':
            dev.set_option('coarse_groups', int(rng.integers(1, 16)))
        try:
            if rng.integers(2):                     # a state reached by an iteration (lagged factor present), else the initial one
                dev.gn_iteration(0., 1e-12, 2000, True)
                poses, points = dev.get_params()
                cur = lp.copy(); cur.poses[...] = poses; cur.points[...] = points
            else:
                cur = lp
            Pm, _, _ = orc.normal_equations(cur, points_first=False)
            Pm = Pm.tocsc()
            dev.covariance_begin()
            nr, nv, d = dev.nr, dev.nv, lp.dof
            worst = 0.
            for _ in range(3):
                if nv > 0 and rng.integers(2):
                    k, idx, comp = 1, int(rng.integers(nv)), int(rng.integers(3))
                    col = nr * d + idx * 3 + comp
                else:
                    k, idx, comp = 0, int(rng.integers(nr)), int(rng.integers(d))
                    col = idx * d + comp
                xp, xl = dev.covariance_column(k, idx, comp)
                x = np.concatenate([xp.ravel(), xl.ravel()])
                e = np.zeros(Pm.shape[0]); e[col] = 1.
                ref = spla.spsolve(Pm, e)
                res = np.linalg.norm(Pm @ x - e)                  # the column must satisfy the oracle's system
                worst = max(worst, min(np.linalg.norm(x - ref) / np.linalg.norm(ref), res * 1e3))
            ok = worst < 1e-6
            msg = 'worst %.1e' % worst
            if not ok and Pm.shape[0] <= 3000:
                # accuracy of a CG solve to a relative residual of 1e-13 degrades with the condition number
                w = np.linalg.eigvalsh(Pm.toarray())
                cond = w[-1] / max(w[0], 1e-300)
                ok = worst < 1e-13 * cond
                msg += ' (condition number %.1e)' % cond
        except Exception as ex:     # noqa: BLE001
            ok, msg = False, 'EXCEPTION %r' % (ex,)
        bad += not ok
        if verbose and (not ok or case % 25 == 0):
            print('%s case %d %s poses %d obs %d edges %d %s %s' % ('ok  ' if ok else 'FAIL', case, kind, lp.num_poses, lp.num_obs, lp.num_edges, mode, msg), flush=True)
        dev.close()
    if verbose:
        print('%d cases, %d failures, %.0f s' % (n_cases, bad, time.time() - t0))
    return bad


if __name__ == '__main__':
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
