#!/usr/bin/env python
"""bench.py -- ms per Gauss-Newton/LM iteration on the BASELINE stereo-BA workload.

    python bench.py --gpus N --steps K --warmup W

One "step" = one full iteration of the hot path on the synthetic stereo bundle
adjustment C3 of SURVEY.md section 8d (200 keyframes, 50 000 landmarks,
500 000 reprojection blocks = 1.5 M residual rows): residuals + Jacobians +
IRLS weights, J^T J assembly, Schur elimination of the landmarks, block-PCG on
the reduced pose system (relative tolerance 1e-12, the parity setting),
back-substitution, retraction and the post-step cost pass -- i.e. exactly what
the reference's ``Problem.solve_one_iter`` + update does (pyslam/problem.py:145-156).
Every step starts from the same linearisation point (a device-side restore of
the initial parameters), so all K steps do identical work.

N > 1 (launched by torch.distributed.run, one rank per GPU): weak scaling --
every rank owns its own 50 000 landmarks / 500 000 observations over the SAME
200 keyframes; the reduced pose system [S | g | cost] is summed with one RCCL
all-reduce per iteration (issued by the HIP core itself on the solver's stream)
and then solved redundantly on every rank; a second 2-double all-reduce sums
the shards' cost and landmark step norm.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0           # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PCG_TOL = 1e-12
PCG_MAX = 1000

KF, LM_PER_GPU, OBS_PER_LM, HALF_WINDOW = 200, 50000, 10, 20


def algorithmic_bytes(info, n_pcg, dof=6):
    """SURVEY.md section 8d byte model, per iteration and for the Schur kernel."""
    N, L, P, nnzb = info['num_obs'], info['num_var_points'], info['num_reduced'], info['reduced_nnzb']
    blk = 8 * dof * dof
    b_iter = 496 * N + 264 * L + 960 * P + blk * nnzb * (1 + n_pcg)
    # k_schur_pairs: every Z row read once (144 B / observation) + every off-diagonal
    # block of S written once (both triangles)
    b_schur = 144 * N + blk * (nnzb - P)
    b_spmv = blk * nnzb + 3 * 8 * dof * P          # k_pcg_spmv: S once, z/p_old in, p_new/q out
    return b_iter, b_schur, b_spmv


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/pmc_traffic.json:
    separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this same command, read
    traffic corrected x2 as MI355X_MICROARCH.md prescribes for gfx950).  None if not collected."""
    try:
        with open(os.path.join(REPO, 'profiles', 'pmc_traffic.json')) as f:
            table = json.load(f)
        total = table[kernel]['hbm_bytes_corrected']
        if kernel == 'k_schur_pairs' and 'k_schur_combine' in table:    # tiled mode: the pair kernel's partials are
            total += table['k_schur_combine']['hbm_bytes_corrected']    # summed by a second, small kernel (same timer)
        return total
    except Exception:
        return None


def cpu_baseline(lp, repeats=5):
    """The numpy/scipy oracle (a port of the reference's algebra) on the host cores: `repeats` whole
    iterations of the same workload (about 10 s of CPU work), mean time per iteration."""
    from oracle import gn_oracle as orc
    ta = tb = tc = 0.
    for _ in range(repeats):
        t0 = time.perf_counter()
        P, b, _ = orc.normal_equations(lp, True)                 # residuals, Jacobians, J^T J, -J^T e
        t1 = time.perf_counter()
        dx = orc.schur_solve(lp, P, b, True)                     # landmark elimination + sparse reduced solve
        t2 = time.perf_counter()
        new = orc.apply_update(lp, dx, True)
        orc.eval_cost(new); orc.eval_cost(new)                   # the reference's line search: 2 cost passes
        t3 = time.perf_counter()
        ta += t1 - t0; tb += t2 - t1; tc += t3 - t2
    ta, tb, tc = ta / repeats, tb / repeats, tc / repeats
    return {'value': round((ta + tb + tc) * 1e3, 1), 'unit': 'ms/LM-iter', 'cores': 1, 'kind': 'port',
            'sample': '{} iterations of the same workload (500k blocks), mean per iteration: vectorised numpy '
                      'residual/Jacobian + scipy CSR J^T J {:.2f} s, CPU Schur + scipy spsolve of the reduced system '
                      '{:.2f} s, update + 2 cost passes {:.2f} s; numpy/scipy single-threaded'.format(repeats, ta, tb, tc),
            'host_cpus': os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--kf', type=int, default=KF)
    ap.add_argument('--lm', type=int, default=LM_PER_GPU)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--force-sharded', action='store_true',
                    help='use the multi-GPU driver (RCCL all-reduce) even with one rank (testing)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus and not (world == 1 and args.gpus == 1):
        raise SystemExit('launch with torch.distributed.run --nproc-per-node {} (WORLD_SIZE={})'.format(args.gpus, world))

    import torch
    from pyslam_amd import synthetic
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_sharded:
        import torch.distributed as dist
        if 'MASTER_ADDR' not in os.environ:
            os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29517', RANK='0', WORLD_SIZE='1')
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))

    lp, _ = synthetic.stereo_ba(num_kf=args.kf, num_lm=args.lm, obs_per_lm=OBS_PER_LM,
                                half_window=HALF_WINDOW, seed=0,
                                lm_offset=rank * args.lm, lm_total=world * args.lm)

    if dist is not None:
        from pyslam_amd.distributed import ShardedDeviceProblem
        dev = ShardedDeviceProblem(lp, dist)
    else:
        from pyslam_amd.device import DeviceProblem
        dev = DeviceProblem(lp, stream=torch.cuda.current_stream().cuda_stream)
    info = dev.info
    dev.snapshot()                                           # the common linearisation point

    def step():
        dev.restore()
        return dev.gn_iteration(0.0, PCG_TOL, PCG_MAX, True)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    # timed region: ONE hipEvent pair around the dominant (Schur) kernel, on every 4th iteration
    # (an event pair costs ~8 us of pipeline bubbles: the kernel is timed on every 4th step of the timed region)
    (dev.dev if hasattr(dev, 'dev') else dev).set_option('profile_every', 4)
    dev.set_profiling(1)
    dev.stage_times(reset=True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    elapsed = time.perf_counter() - t0
    stages = dev.stage_times(reset=True)
    # untimed: a few more steps with an event pair around every stage, for the breakdown only
    (dev.dev if hasattr(dev, 'dev') else dev).set_option('profile_every', 1)
    dev.set_profiling(2)
    for _ in range(5):
        step()
    detail = dev.stage_times(reset=True)
    dev.set_profiling(0)
    for k, v in detail.items():
        if k != 'schur_pairs' and v[1] > 0:
            stages[k] = v
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    cost, dx_norm, n_pcg, relres = out
    ms_per_step = elapsed * 1e3 / args.steps

    if rank == 0:
        b_iter, b_schur, b_spmv = algorithmic_bytes(info, n_pcg)
        stage_ms = {k: (v[0] / max(v[1], 1)) for k, v in stages.items() if v[1] > 0}
        # dominant kernel: the Schur pair kernel is ONE launch per iteration; PCG is 2*n_pcg launches
        sch = stage_ms.get('schur_pairs', 0.0)
        pcg_per_iter = stage_ms.get('pcg', 0.0) / max(n_pcg, 1)
        if sch >= pcg_per_iter * 1.0 and sch > 0:
            kern, dur_ms, nbytes = 'k_schur_pairs', sch, b_schur
        else:
            kern, dur_ms, nbytes = 'void k_cg_fused<6>', pcg_per_iter, b_spmv
        achieved = nbytes / (dur_ms * 1e-3) / 1e9 if dur_ms > 0 else 0.0
        line = {
            'metric': 'ms/LM-iter (Jac build + J^T J + Schur solve), stereo BA @ 500k residuals',
            'value': round(ms_per_step, 4), 'unit': 'ms/LM-iter', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 4),
            'higher_is_better': False, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64',
            'data': 'synthetic',
            'config': {'workload': 'C3 stereo BA: {} keyframes x {} landmarks/GPU x {} obs/landmark = {} '
                                   'reprojection blocks/GPU (3 rows each), L2 loss, pose 0 constant'.format(
                                       args.kf, args.lm, OBS_PER_LM, info['num_obs']),
                       'parallelism': 'landmark-sharded x{} + RCCL all-reduce of the reduced pose system'.format(world)
                       if world > 1 else 'single GPU',
                       'pcg_tol': PCG_TOL, 'pcg_iters': n_pcg, 'pcg_relres': relres,
                       'reduced_blocks': info['reduced_nnzb'], 'schur_pairs': info['num_pairs'],
                       'cost_after_step': cost, 'step_norm': dx_norm},
            'residual_blocks_per_s': round(info['num_obs'] * world / (ms_per_step * 1e-3), 1),
            'stage_ms': {k: round(v, 4) for k, v in stage_ms.items()},
            'stage_ms_note': 'schur_pairs: hipEvent pair inside the timed region, on every 4th step; the other stages and iteration_total '
                             '(GPU time of one iteration) from 5 extra untimed steps with an event pair around every stage',
            'iteration_algorithmic_GBps': round(b_iter / (ms_per_step * 1e-3) / 1e9, 2),
            'roofline': {'bound': 'hbm', 'kernel': kern, 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4),
                         'traffic': pmc_traffic(kern),
                         'algorithmic_bytes_per_launch': int(nbytes), 'avg_launch_ms': round(dur_ms, 5)},
        }
        if not args.no_cpu_baseline and world == 1:
            line['cpu_baseline'] = cpu_baseline(lp)
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dev.close()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
