"""Randomised sweep of what the multi-GPU all-reduce relies on, on ONE GPU: random problems (stereo BA, with and without
pose-pose edges / a prior, constant landmarks, damping) cut into 2-6 landmark shards; every shard is built with the union
block pattern, and the shards' reduced systems, gradients and costs must add up to the unsharded ones.
usage: python tests/fuzz_shards.py [cases] [seed0]"""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pyslam_amd import synthetic, losses
from pyslam_amd.device import DeviceProblem
from pyslam_amd.distributed import shard_landmarks, pose_pair_keys

LOSSES = [lambda: losses.L2Loss(), lambda: losses.HuberLoss(1.5), lambda: losses.CauchyLoss(3.0)]


def run(n_cases, seed0=0, verbose=True):
    bad = 0
    t0 = time.time()
    for case in range(seed0, seed0 + n_cases):
        rng = np.random.default_rng(31000 + case)
        loss = LOSSES[rng.integers(len(LOSSES))]()
        kf, obs = int(rng.choice([3, 6, 15, 40, 90])), int(rng.integers(2, 6))
        lp, _ = synthetic.stereo_ba(num_kf=kf, num_lm=int(rng.integers(20 * kf // min(obs, kf) + 8, 60 * kf // min(obs, kf) + 40)),
                                    obs_per_lm=min(obs, kf), half_window=int(rng.integers(obs, 3 * obs + 2)), seed=case, loss=loss,
                                    const_point_fraction=float(rng.choice([0., 0.25])))
        if kf >= 3 and rng.integers(2):
            lp = synthetic.with_pose_edges(lp, int(rng.integers(0, 2 * kf)), case + 1, loss=loss)
        world = int(rng.integers(2, 7))
        lam = float(rng.choice([0., 0., 0.05]))
        try:
            full = DeviceProblem(lp)
            full.linearize(lam)
            rp, ci, vals, g = full.reduced_system()
            union = pose_pair_keys(lp)
            acc_v, acc_g, cost, nobs = np.zeros_like(vals), np.zeros_like(g), 0., 0
            same_pattern = True
            for r in range(world):
                sh = shard_landmarks(lp, r, world)
                extra = np.setdiff1d(union, pose_pair_keys(sh))
                dev = DeviceProblem(sh, extra_pairs=((extra >> 32).astype(np.int32), (extra & 0xFFFFFFFF).astype(np.int32)))
                dev.linearize(lam)
                rp2, ci2, v2, g2 = dev.reduced_system()
                same_pattern = same_pattern and np.array_equal(rp, rp2) and np.array_equal(ci, ci2)
                acc_v += v2; acc_g += g2
                cost += dev.eval_cost(True); nobs += sh.num_obs
                dev.close()
            e_v = np.abs(acc_v - vals).max() / np.abs(vals).max()
            e_g = np.abs(acc_g - g).max() / max(np.abs(g).max(), 1e-300)
            e_c = abs(cost - full.eval_cost(True)) / cost
            ok = same_pattern and nobs == lp.num_obs and e_v < 1e-10 and e_g < 1e-10 and e_c < 1e-12
            msg = 'S %.1e g %.1e cost %.1e pattern %s' % (e_v, e_g, e_c, same_pattern)
            full.close()
        except Exception as e:      # noqa: BLE001
            ok, msg = False, 'EXCEPTION %r' % (e,)
        bad += not ok
        if verbose and (not ok or case % 20 == 0):
            print('%s case %d kf %d obs %d edges %d world %d lambda %g  %s' % ('ok  ' if ok else 'FAIL', case, kf, lp.num_obs, lp.num_edges, world, lam, msg), flush=True)
    if verbose:
        print('%d cases, %d failures, %.0f s' % (n_cases, bad, time.time() - t0))
    return bad


if __name__ == '__main__':
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
