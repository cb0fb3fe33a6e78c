"""Randomised sweep of the whole solve(): seeded random small problems are turned into Problem objects
(synthetic.to_objects -> add_residual_block / initialize_params / set_parameters_constant), solved through the
public API on the device, and compared with the oracle's solve (reference control flow + sparse direct solves):
same number of iterations, same cost history, same final parameters.  usage: python tests/fuzz_solve.py [cases] [seed0]"""
import os, sys, time
root = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [root, os.path.join(root, 'tests')]
import numpy as np
from oracle import gn_oracle as orc
from pyslam_amd import synthetic, losses
from pyslam_amd.lowering import pack_pose
from test_host_api import build_namespace

LOSSES = [lambda: losses.L2Loss(), lambda: losses.HuberLoss(1.5), lambda: losses.CauchyLoss(3.0), lambda: losses.TDistributionLoss(5.0)]


def run(n_cases, seed0=0, verbose=True):
    bad = 0
    ns = build_namespace()
    t0 = time.time()
    for case in range(seed0, seed0 + n_cases):
        rng = np.random.default_rng(5000 + case)
        loss = LOSSES[rng.integers(len(LOSSES))]()
        kind = rng.choice(['ba', 'pg3', 'pg2'])
        if kind == 'ba':
            kf, obs = int(rng.choice([3, 4, 6, 10, 18])), int(rng.integers(2, 4))
            lp, truth = synthetic.stereo_ba(num_kf=kf, num_lm=int(rng.integers(30 * kf // obs, 60 * kf // obs)), obs_per_lm=obs,
                                        half_window=int(rng.integers(obs, 2 * obs + 2)), seed=case, loss=loss,
                                        const_point_fraction=float(rng.choice([0., 0.2])))
            if kf >= 3 and rng.integers(3) == 0:        # + pose-pose edges and a prior over the same keyframes
                lp = synthetic.with_pose_edges(lp, int(rng.integers(0, 2 * kf)), case + 1, loss=loss, truth_poses=truth['poses'])
        else:
            P = int(rng.choice([4, 6, 12, 17, 40]))
            lp, _ = synthetic.pose_graph(num_poses=P, num_loops=int(rng.integers(1, 3 * P)), dof=6 if kind == 'pg3' else 3,
                                         seed=case, loss=loss)
        opts = dict(allow_nondecreasing_steps=bool(rng.integers(2)), max_nondecreasing_steps=int(rng.integers(2, 5)),
                    linesearch_max_iters=int(rng.choice([0, 10])), max_iters=int(rng.choice([4, 30])),
                    min_cost_decrease=float(rng.choice([0.9, 0.99])))
        pf = bool(rng.integers(2))
        opt = ns.Options()
        for k, v in opts.items():
            setattr(opt, k, v)
        try:
            problem = synthetic.to_objects(lp, ns, opt, points_first=pf)
            final = problem.solve()
            ref_lp, ref = orc.solve(lp, opts, points_first=pf)
            hist, want = np.array(problem._cost_history), ref['cost_history']
            ok = len(hist) == len(want)
            if ok:
                big = want > 1e-9 * want[0]
                ok = np.allclose(hist[big], want[big], rtol=1e-6)
                got = np.stack([pack_pose(final[k]) for k in problem._device.lp.pose_keys])
                e_p = np.abs(got - ref_lp.poses).max()
                ok = ok and e_p < 1e-6
                msg = 'iterations %d, pose error %.1e' % (len(hist) - 1, e_p)
                if ok and case % 4 == 0 and ref_lp.num_reduced * ref_lp.dof + 3 * ref_lp.num_var_points <= 1200:
                    # compute_covariance / get_covariance_block at the solution, in the reference's unknown order
                    Pm, _, _ = orc.normal_equations(ref_lp, points_first=pf)
                    cov = np.linalg.inv(Pm.toarray())
                    problem.compute_covariance()
                    part = problem._update_partition_dict
                    keys = list(part.keys())
                    k0, k1 = keys[int(rng.integers(len(keys)))], keys[int(rng.integers(len(keys)))]
                    got_c = np.atleast_2d(problem.get_covariance_block(k0, k1))
                    r0, r1 = part[k0], part[k1]
                    want_c = np.atleast_2d(cov[r0.start:r0.stop, r1.start:r1.stop])
                    e_c = np.abs(got_c - want_c).max() / max(np.abs(cov).max(), 1e-300)
                    ok = e_c < 1e-6
                    msg += ', covariance block error %.1e' % e_c
            else:
                msg = 'iterations %d vs %d: %s | %s' % (len(hist) - 1, len(want) - 1, hist[-3:], want[-3:])
        except Exception as e:      # noqa: BLE001
            ok, msg = False, 'EXCEPTION %r' % (e,)
        bad += not ok
        if verbose and (not ok or case % 10 == 0):
            print('%s case %d %s poses %d obs %d edges %d %s %s  %s' % ('ok  ' if ok else 'FAIL', case, kind, lp.num_poses, lp.num_obs,
                                                                   lp.num_edges, type(loss).__name__, opts, msg), flush=True)
    if verbose:
        print('%d cases, %d failures, %.0f s' % (n_cases, bad, time.time() - t0))
    return bad


if __name__ == '__main__':
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
