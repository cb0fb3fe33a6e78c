"""Randomised sweep of Problem-API SEQUENCES: one Problem object lives through several solves while parameters are
frozen / released (set_parameters_constant / _variable), with eval_cost, solve_one_iter and compute_covariance calls in
between -- the device handle is reused or rebuilt as the structure changes (pyslam_amd/problem.py:_get_device).  After
every change the current object graph is lowered again and the oracle solves that table; cost histories and final
parameters must agree.  usage: python tests/fuzz_api.py [cases] [seed0]"""
import os, sys, time
root = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [root, os.path.join(root, 'tests')]
import numpy as np
from oracle import gn_oracle as orc
from pyslam_amd import synthetic, losses
from test_host_api import build_namespace

LOSSES = [lambda: losses.L2Loss(), lambda: losses.HuberLoss(1.5), lambda: losses.CauchyLoss(3.0)]


def run(n_cases, seed0=0, verbose=True):
    bad = 0
    ns = build_namespace()
    t0 = time.time()
    for case in range(seed0, seed0 + n_cases):
        rng = np.random.default_rng(64000 + case)
        loss = LOSSES[rng.integers(len(LOSSES))]()
        kind = rng.choice(['ba', 'mix', 'pg3', 'pg2'])
        if kind in ('ba', 'mix'):
            kf, obs = int(rng.choice([4, 6, 10, 18])), int(rng.integers(2, 4))
            lp, truth = synthetic.stereo_ba(num_kf=kf, num_lm=int(rng.integers(40 * kf // obs, 70 * kf // obs)), obs_per_lm=obs,
                                            half_window=int(rng.integers(obs, 2 * obs + 2)), seed=case, loss=loss)
            if kind == 'mix':
                lp = synthetic.with_pose_edges(lp, int(rng.integers(0, 2 * kf)), case + 1, loss=loss, truth_poses=truth['poses'])
        else:
            P = int(rng.choice([5, 9, 17, 40]))
            lp, _ = synthetic.pose_graph(num_poses=P, num_loops=int(rng.integers(P, 3 * P)), dof=6 if kind == 'pg3' else 3, seed=case, loss=loss)
        opts = dict(allow_nondecreasing_steps=True, max_nondecreasing_steps=3, linesearch_max_iters=int(rng.choice([0, 10])),
                    max_iters=int(rng.choice([3, 20])), min_cost_decrease=0.99)
        pf = bool(rng.integers(2))
        opt = ns.Options()
        for k, v in opts.items():
            setattr(opt, k, v)
        log = []
        try:
            problem = synthetic.to_objects(lp, ns, opt, points_first=pf)
            pose_keys = [k for k in problem.param_dict if k.startswith('T')]
            point_keys = [k for k in problem.param_dict if k.startswith('pt')]
            ok, msg = True, ''
            for step in range(int(rng.integers(2, 5))):
                op = rng.choice(['freeze_poses', 'release', 'freeze_points', 'perturb', 'nothing'])
                if op == 'freeze_poses' and len(pose_keys) > 3:
                    problem.set_parameters_constant([str(k) for k in rng.choice(pose_keys[1:], size=min(2, len(pose_keys) - 2), replace=False)])
                elif op == 'release':
                    frozen = [k for k in problem.constant_param_keys if k not in (pose_keys[:1])]
                    if frozen:
                        problem.set_parameters_variable(frozen[: int(rng.integers(1, len(frozen) + 1))])
                elif op == 'freeze_points' and point_keys:
                    problem.set_parameters_constant([str(k) for k in rng.choice(point_keys, size=min(5, len(point_keys)), replace=False)])
                elif op == 'perturb':
                    for k in pose_keys[1:3]:
                        if k not in problem.constant_param_keys:
                            problem.param_dict[k].perturb(0.01 * rng.standard_normal(problem.param_dict[k].dof))
                log.append(op)
                cur = problem._lower()
                if cur.num_reduced == 0 and cur.num_var_points == 0:
                    continue
                c_dev = problem.eval_cost()
                c_orc = orc.eval_cost(cur, True)
                if rng.integers(3) == 0:
                    dx, _ = problem.solve_one_iter()              # must not move the parameters
                    if abs(problem.eval_cost() - c_dev) > 1e-12 * abs(c_dev):
                        ok, msg = False, 'solve_one_iter moved the parameters'
                        break
                ref_lp, ref = orc.solve(cur, opts, points_first=pf)
                problem.solve()
                hist, want = np.array(problem._cost_history), ref['cost_history']
                after = problem._lower()
                e_p = np.abs(after.poses - ref_lp.poses).max() if after.num_poses else 0.
                e_l = np.abs(after.points - ref_lp.points).max() if after.num_points else 0.
                big = want > 1e-9 * want[0]
                if not (abs(c_dev - c_orc) <= 1e-10 * abs(c_orc) and len(hist) == len(want) and np.allclose(hist[big], want[big], rtol=1e-6)
                        and e_p < 1e-6 and e_l < 1e-4):
                    ok, msg = False, 'step %d (%s): cost %.3e vs %.3e (rel %.1e), iterations %d vs %d, history rel err %.1e, pose %.1e point %.1e' % (
                        step, op, c_dev, c_orc, abs(c_dev - c_orc) / abs(c_orc), len(hist) - 1, len(want) - 1,
                        float(np.max(np.abs(hist[:min(len(hist), len(want))] - want[:min(len(hist), len(want))]) / want[:min(len(hist), len(want))])), e_p, e_l)
                    msg += ' | device %s | oracle %s | opts %s' % (np.array2string(hist, precision=9), np.array2string(want, precision=9), opts)
                    break
                if rng.integers(4) == 0:
                    problem.compute_covariance()
                    part = problem._update_partition_dict
                    if part:
                        k0 = list(part.keys())[int(rng.integers(len(part)))]
                        Pm, _, _ = orc.normal_equations(after, points_first=pf)
                        if Pm.shape[0] <= 1200:
                            cov = np.linalg.inv(Pm.toarray())
                            r0 = part[k0]
                            e_c = np.abs(np.atleast_2d(problem.get_covariance_block(k0, k0)) - cov[r0.start:r0.stop, r0.start:r0.stop]).max() / np.abs(cov).max()
                            if e_c > 1e-6:
                                ok, msg = False, 'covariance block %s error %.1e' % (k0, e_c)
                                break
            msg = msg or 'ops %s' % log
        except Exception as e:      # noqa: BLE001
            import traceback
            ok, msg = False, 'EXCEPTION %r after %s; oracle cost history %s' % (e, log, locals().get('ref', {}).get('cost_history') if isinstance(locals().get('ref'), dict) else None)
        if not ok and 'landmark block' in msg and locals().get('cur') is not None and not os.environ.get('FUZZ_API_NOSKIP'):
            # a landmark whose 3 x 3 block is singular to rounding (rays nearly parallel after an update): the device
            # refuses it, the reference's LU silently produces a huge step; nothing to compare
            try:
                # (round 5, case 940269: the reference's own trajectory diverges -- cost 2e3 -> 1e11 -> NaN -- and the device stops at
                #  the first landmark block that is no longer positive definite, which is the documented difference in error behaviour,
                #  a NaN history is nothing to compare against)
                if not np.all(np.isfinite(np.asarray(ref['cost_history'], dtype=float))):
                    ok, msg = True, 'the reference trajectory itself ends in NaN, skipped'
                    raise StopIteration
                Pc, _, _ = orc.normal_equations(ref_lp, points_first=False)
                nn = ref_lp.num_reduced * ref_lp.dof
                Hd = Pc[nn:, nn:].toarray()
                ev = np.array([np.linalg.eigvalsh(Hd[3 * i:3 * i + 3, 3 * i:3 * i + 3]) for i in range(ref_lp.num_var_points)])
                if (ev[:, 0] / ev[:, 2]).min() < 1e-9:
                    ok, msg = True, 'degenerate landmark in the reference trajectory, skipped'
            except Exception:       # noqa: BLE001
                pass
        if not ok and 'history rel err' in msg and locals().get('cur') is not None and not os.environ.get('FUZZ_API_NOSKIP'):
            # Is the reference's OWN result determined by its inputs to the tolerance of the comparison?  Re-run the oracle
            # with the landmarks (or, without landmarks, the poses) perturbed by 1e-13 relative: undamped Gauss-Newton with
            # the reference's degenerate line search can amplify a rounding-level difference by 1e3 per iteration around a
            # minimum it oscillates about (tools/diag_fuzz_api.py, case 701050: 1e-13 -> 1e-11 -> 2e-10 -> 3e-3).  Only if
            # the oracle's two histories separate by more than the test's own tolerance is the case set aside.
            try:
                pert = cur.copy()
                g2 = np.random.default_rng(1)
                if pert.num_var_points:
                    pert.points = pert.points * (1. + 1e-13 * g2.standard_normal(pert.points.shape))
                else:
                    pert.poses = pert.poses * (1. + 1e-13 * g2.standard_normal(pert.poses.shape))
                ref2_lp, ref2 = orc.solve(pert, opts, points_first=pf)
                h1, h2 = np.asarray(ref['cost_history']), np.asarray(ref2['cost_history'])
                n = min(len(h1), len(h2))
                sens = float(np.max(np.abs(h1[:n] - h2[:n]) / np.abs(h1[:n]))) if len(h1) == len(h2) else np.inf
                # (round 6, case 1040022: histories agree to 1e-11 and ONE landmark of the result differs by 1.6e-3 -- a landmark
                #  seen along two nearly parallel rays, block condition 9e11: the oracle's own final landmarks move by 2e-4 .. 3e-3
                #  under 1e-15 .. 1e-13 perturbations of its inputs.  The same rule for the parameters as for the history.)
                sens_l = float(np.abs(ref2_lp.points - ref_lp.points).max()) if ref_lp.num_points else 0.
                sens_p = float(np.abs(ref2_lp.poses - ref_lp.poses).max()) if ref_lp.num_poses else 0.
                if sens > 1e-6:
                    ok, msg = True, 'reference not determined by its inputs (1e-13 perturbation -> %.1e in its own history), skipped: %s' % (sens, msg[:60])
                elif sens_l > 1e-4 or sens_p > 1e-6:
                    ok, msg = True, 'reference parameters not determined by its inputs (1e-13 perturbation -> landmarks %.1e, poses %.1e in its own result), skipped: %s' % (sens_l, sens_p, msg[:60])
            except Exception:       # noqa: BLE001
                pass
        if not ok and os.environ.get('FUZZ_API_DUMP') and locals().get('cur') is not None:
            # the tables the failing solve started from (diagnosis offline, tools/diag_fuzz_api.py)
            os.makedirs(os.environ['FUZZ_API_DUMP'], exist_ok=True)
            np.savez(os.path.join(os.environ['FUZZ_API_DUMP'], 'fuzz_api_case_%d.npz' % case), lp_dof=cur.dof, pf=pf,
                     device_history=np.asarray(problem._cost_history), opts_keys=np.array(list(opts.keys())),
                     opts_vals=np.array([float(v) for v in opts.values()]),
                     **{'lp_' + k: getattr(cur, k) for k in cur.STRUCTURE_FIELDS + ('poses', 'points')})
        bad += not ok
        if verbose and (not ok or case % 20 == 0):
            print('%s case %d %s poses %d obs %d edges %d  %s' % ('ok  ' if ok else 'FAIL', case, kind, lp.num_poses, lp.num_obs, lp.num_edges, msg), flush=True)
    if verbose:
        print('%d cases, %d failures, %.0f s' % (n_cases, bad, time.time() - t0))
    return bad


if __name__ == '__main__':
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
