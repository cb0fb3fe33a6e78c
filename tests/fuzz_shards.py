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
        lp, truth = synthetic.stereo_ba(num_kf=kf, num_lm=int(rng.integers(20 * kf // min(obs, kf) + 8, 60 * kf // min(obs, kf) + 40)),
                                    obs_per_lm=min(obs, kf), half_window=int(rng.integers(obs, 3 * obs + 2)), seed=case, loss=loss,
                                    const_point_fraction=float(rng.choice([0., 0.25])))
        if kf >= 3 and rng.integers(2):
            lp = synthetic.with_pose_edges(lp, int(rng.integers(0, 2 * kf)), case + 1, loss=loss, truth_poses=truth['poses'])
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




def run_one_rank(n_cases, seed0=0, verbose=True):
    """The sharded driver (native RCCL path and torch.distributed fallback) with ONE rank against the unsharded
    iteration on random problems and solver modes: a sum over one rank is the identity, so costs and CG iteration
    counts must be identical and ||dx|| equal to rounding."""
    import torch
    import torch.distributed as dist
    from pyslam_amd.distributed import ShardedDeviceProblem
    own = not dist.is_initialized()
    if own:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29547', RANK='0', WORLD_SIZE='1')
        dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
    bad = 0
    t0 = time.time()
    for case in range(seed0, seed0 + n_cases):
        rng = np.random.default_rng(47000 + case)
        loss = LOSSES[rng.integers(len(LOSSES))]()
        kf, obs = int(rng.choice([3, 6, 15, 40, 90])), int(rng.integers(2, 6))
        lp, truth = synthetic.stereo_ba(num_kf=kf, num_lm=int(rng.integers(30 * kf // min(obs, kf) + 8, 60 * kf // min(obs, kf) + 40)),
                                    obs_per_lm=min(obs, kf), half_window=int(rng.integers(obs, 3 * obs + 2)), seed=case, loss=loss)
        if rng.integers(2):
            lp = synthetic.with_pose_edges(lp, int(rng.integers(0, 2 * kf)), case + 1, loss=loss, truth_poses=truth['poses'])
        native = bool(rng.integers(2))
        mode = rng.choice(['auto', 'explicit', 'G'])
        ls = bool(rng.integers(2))
        try:
            ref = DeviceProblem(lp, stream=torch.cuda.current_stream().cuda_stream)
            # (the sharded solve never has the lagged dense inverse: a reference that seeds it in its second call -- a start so
            #  close to the optimum that the first step changes the cost by less than 5 %, four of the 120 seeds 960000.. -- solves
            #  its third call in 7 iterations where the sharded one takes 29: same step to rounding, not the same launches;
            #  round 3's library does the same on those seeds)
            ref.set_option('lagged_inverse', 0)
            sh = ShardedDeviceProblem(lp, dist, native_rccl=native)
            if sh.native is None:       # a caller that drives the collectives itself cannot repeat a solve on one rank: it never gets
                ref.set_option('cg_persist', 0); ref.set_option('xcg_persist', 0)      # the one-launch solvers (round 6): same kernels
            for d in (ref, sh.dev):
                if mode == 'explicit':
                    d.set_option('cg_explicit_min_rows', 0); d.set_option('cg_split_min_rows', 0)
                elif mode == 'G':
                    d.set_option('coarse_groups', int(1 + case % 9))
            ok, msg, same_solver = True, '', True
            for it in range(3):
                a = ref.gn_iteration(0., 1e-12, 2000, ls)
                b = sh.gn_iteration(0., 1e-12, 2000, ls)
                # (systems of <= 90 unknowns: the unsharded call solves directly, the enqueue-only sharded protocol by CG)
                same_solver = a[2] > 0
                tol_c, tol_n = (0., 1e-13) if same_solver else (1e-9, 1e-8)
                if not (abs(a[0] - b[0]) <= tol_c * abs(a[0]) and (a[2] == b[2] or not same_solver) and abs(a[1] - b[1]) <= tol_n * max(a[1], 1e-300)):
                    ok, msg = False, 'iteration %d: %r vs %r' % (it, a, b)
                    break
            pa, la = ref.get_params(); pb, lb = sh.get_params()
            ok = ok and (np.array_equal(pa, pb) and np.array_equal(la, lb) if same_solver else
                         np.abs(pa - pb).max() < 1e-7 and np.abs(la - lb).max() < 1e-6)
            sh.close(); ref.close()
        except Exception as e:      # noqa: BLE001
            ok, msg = False, 'EXCEPTION %r' % (e,)
        bad += not ok
        if verbose and (not ok or case % 20 == 0):
            print('%s case %d kf %d obs %d edges %d native %d %s ls %d  %s' % ('ok  ' if ok else 'FAIL', case, kf, lp.num_obs, lp.num_edges, native, mode, ls, msg), flush=True)
    if verbose:
        print('%d one-rank cases, %d failures, %.0f s' % (n_cases, bad, time.time() - t0))
    if own:
        dist.destroy_process_group()
    return bad


if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    sys.exit(1 if (run(n, s0) + run_one_rank(n, s0)) else 0)
