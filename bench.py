#!/usr/bin/env python
"""bench.py -- ms per Gauss-Newton/LM iteration on the BASELINE stereo-BA workloads.

    python bench.py --gpus N --steps K --warmup W

One "step" = one full iteration of the hot path: residuals + Jacobians + IRLS
weights, J^T J assembly, Schur elimination of the landmarks, block-PCG on the
reduced pose system (relative tolerance 1e-12, the parity setting),
back-substitution, retraction and the post-step cost pass -- exactly what the
reference's ``Problem.solve_one_iter`` + update does (pyslam/problem.py:145-156).

N = 1: workload C3 of SURVEY.md section 8d (200 keyframes, 50 000 landmarks,
500 000 reprojection blocks = 1.5 M residual rows), the configuration
BASELINE.json's metric is quoted on.  ``value`` is the steady-state figure (every
timed step starts from the same linearisation point: a device-side restore of the
initial parameters, so all K steps do identical work); next to it the line carries
``trajectory_ms_per_iter`` (5 consecutive iterations of a real solve, no restore:
the lagged coarse factor and the CG launch-count prediction are live) and
``c4_single_gpu_ms`` (the north star's 2 000-keyframe / 500 000-landmark problem on
this one GPU -- the strong-scaling base).

N > 1 (launched by torch.distributed.run, one rank per GPU): STRONG scaling of
the fixed C4 problem (2 000 keyframes, 500 000 landmarks, 5 M blocks; BASELINE.json
configs[3]): every rank generates the same problem and keeps landmark shard
``rank`` of N (``shard_landmarks``); the reduced pose system [upper(S) | g | cost | flag]
is summed with one RCCL all-reduce per iteration (issued by the HIP core itself on
the solver's stream) and then solved redundantly on every rank; a second small
all-reduce sums the shards' cost and landmark step norm.  Rank 0 also times the
unsharded C4 problem on its own GPU after the timed region (``c4_single_gpu_ms``).

Prints ONE JSON line (rank 0).
"""
import argparse
import hashlib
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

C3 = dict(num_kf=200, num_lm=50000, seed=0)
C4 = dict(num_kf=2000, num_lm=500000, seed=1)
OBS_PER_LM, HALF_WINDOW = 10, 20
SPSOLVE_FULL_C3_RECORDED_S = 288.0      # tests/bench_cpu_solvers.py on an MI355X box's host (round 1)


def algorithmic_bytes(info, n_pcg, dof=6):
    """SURVEY.md section 8d byte model, per iteration and for the Schur kernel."""
    N, L, P, nnzb = info['num_obs'], info['num_var_points'], info['num_reduced'], info['reduced_nnzb']
    blk = 8 * dof * dof
    b_iter = 496 * N + 264 * L + 960 * P + blk * nnzb * (1 + n_pcg)
    # k_schur_pairs: every Z row read once (144 B / observation in the survey's model) + every
    # off-diagonal block of S written once (both triangles)
    b_schur = 144 * N + blk * (nnzb - P)
    b_spmv = blk * nnzb + 3 * 8 * dof * P          # one CG launch: S once, three vectors
    return b_iter, b_schur, b_spmv


def kernel_source_sha():
    """Hash of the HIP sources: ties a committed PMC pass to the build it was taken on."""
    h = hashlib.sha256()
    d = os.path.join(REPO, 'pyslam_amd', 'csrc')
    for f in sorted(os.listdir(d)):
        with open(os.path.join(d, f), 'rb') as fh:
            h.update(f.encode()); h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel, config=None):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/pmc_traffic.json:
    separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this same command, read
    traffic corrected x2 as MI355X_MICROARCH.md prescribes for gfx950) + the source hash of the
    build the passes were taken on.  (None, ...) if not collected."""
    try:
        with open(os.path.join(REPO, 'profiles', 'pmc_traffic.json')) as f:
            table = json.load(f)
        meta = table.get('_meta', {})
        if config is not None:
            table = table[config]
        schur = kernel.startswith('k_schur_pairs')
        if schur:       # rocprof's name of the pipelined pair kernel (a template), else the round-2 kernel
            kernel = next(k for k in ('void k_schur_pairs_db<0>', 'k_schur_pairs') if k in table)
        total = table[kernel]['hbm_bytes_corrected']
        if schur and 'k_schur_combine' in table:                        # tiled mode: the pair kernel's partials are
            total += table['k_schur_combine']['hbm_bytes_corrected']    # summed by a second, small kernel (same timer)
        return total, meta.get('source_sha'), meta.get('git_head')
    except Exception:
        return None, None, None


def cpu_baseline(lp, repeats=5):
    """The numpy/scipy oracle (a port of the reference's algebra) on the host cores: `repeats` whole
    iterations of the same workload (about 10 s of CPU work), mean time per iteration; plus the
    reference-faithful linear solve (scipy spsolve of the FULL normal equations, pyslam/problem.py:186) on
    a bounded sample, next to the CPU Schur complement on the same sample (SURVEY.md section 8d ii)."""
    import scipy.sparse.linalg as spla
    from oracle import gn_oracle as orc
    from pyslam_amd import synthetic
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
    # reference-faithful: SuperLU on the full normal equations, landmarks-first (the reference's parameter order)
    small = dict(num_kf=100, num_lm=5000)
    lps, _ = synthetic.stereo_ba(obs_per_lm=OBS_PER_LM, half_window=HALF_WINDOW, seed=0, **small)
    Ps, bs, _ = orc.normal_equations(lps, True)
    t0 = time.perf_counter(); dxf = spla.spsolve(Ps, bs); t_full = time.perf_counter() - t0
    t0 = time.perf_counter(); dxs = orc.schur_solve(lps, Ps, bs, True); t_schur = time.perf_counter() - t0
    return {'value': round((ta + tb + tc) * 1e3, 1), 'unit': 'ms/LM-iter', 'cores': 1, 'kind': 'port',
            'sample': '{} iterations of the same workload (500k blocks), mean per iteration: vectorised numpy '
                      'residual/Jacobian + scipy CSR J^T J {:.2f} s, CPU Schur + scipy spsolve of the reduced system '
                      '{:.2f} s, update + 2 cost passes {:.2f} s; numpy/scipy single-threaded'.format(repeats, ta, tb, tc),
            'host_cpus': os.cpu_count(),
            'spsolve_full_s': round(t_full, 2),
            'spsolve_full_sample': 'scipy.sparse.linalg.spsolve on the FULL normal equations (reference problem.py:186) of a '
                                   '{num_kf}-keyframe x {num_lm}-landmark x 10 stereo BA ({n} blocks, {u} unknowns): {f:.2f} s; '
                                   'CPU Schur + spsolve of the reduced system on the same matrix: {s:.2f} s; |dx_full - dx_schur| / |dx| = '
                                   '{e:.1e}'.format(n=lps.num_obs, u=Ps.shape[0], f=t_full, s=t_schur,
                                                    e=float(np.linalg.norm(dxf - dxs) / np.linalg.norm(dxf)), **small),
            'spsolve_full_c3_recorded_s': SPSOLVE_FULL_C3_RECORDED_S,
            'spsolve_full_c3_note': 'recorded (tests/bench_cpu_solvers.py, round 1, same host type): one spsolve call on the '
                                    'full 151 194-unknown C3 system; not re-run here (exceeds the time bound)'}


def time_steps(dev, steps, warmup, fence):
    """W untimed + K timed steady-state steps (restore + iteration); -> (seconds, last result)."""
    def step():
        dev.restore()
        return dev.gn_iteration(0.0, PCG_TOL, PCG_MAX, True)
    out = None
    for _ in range(warmup):
        out = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    fence()
    return time.perf_counter() - t0, out


def trajectory(dev, iters=5, reps=3):
    """`iters` consecutive Gauss-Newton iterations from the perturbed start WITHOUT restoring the
    linearisation point: what a real solve() pays (stale lagged coarse factor, CG launch-count misses).
    A single call is 0.2-0.4 ms of host wall clock, so the trajectory is run `reps` times (each time after a few
    steady-state steps at the start point, which is the state the first run starts from) and the per-call MEDIAN reported."""
    import torch
    runs, its, costs = [], [], []
    for rep in range(reps):
        for _ in range(0 if rep == 0 else 8):
            dev.restore(); dev.gn_iteration(0.0, PCG_TOL, PCG_MAX, True)
        dev.restore()
        torch.cuda.synchronize()
        per, its, costs = [], [], []
        for _ in range(iters):
            t0 = time.perf_counter()
            cost, _, n, _ = dev.gn_iteration(0.0, PCG_TOL, PCG_MAX, True)     # synchronises (results are read back)
            per.append((time.perf_counter() - t0) * 1e3)
            its.append(n); costs.append(cost)
        runs.append(per)
    dev.restore()
    per = [float(np.median([r[k] for r in runs])) for k in range(iters)]
    return {'mean': round(float(np.mean(per)), 4), 'per_iter': [round(p, 4) for p in per], 'pcg_iters': its,
            'cost': costs, 'runs_mean': [round(float(np.mean(r)), 4) for r in runs],
            'note': '{} consecutive iterations from the perturbed start, no restore, host wall clock per ps_gn_iteration call '
                    '(each call ends with its one synchronisation); per-call median of {} runs'.format(iters, reps)}


def problem_info(dev):
    """ps_problem_info of a DeviceProblem as a dict (the lagged-inverse counters live there)."""
    import ctypes as C
    from pyslam_amd import _native as nat
    i = nat.ProblemInfo()
    nat.check(dev._lib.ps_get_info(dev._h, C.byref(i)))
    return {k: getattr(i, k) for k, _ in nat.ProblemInfo._fields_}


def c5_frames():
    """Wall time of ONE per-frame motion-only Problem (BASELINE config 5, the way the reference's sparse VO pipeline runs it,
    pyslam/pipelines/sparse.py:153-161): Problem(); add_residual_block(ReprojectionMotionOnlyBatchResidual, CauchyLoss);
    initialize_params(); solve() -- split into lowering / ps_problem_create / iterations / write-back (tools/c5_frame_probe.py)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('c5_frame_probe', os.path.join(REPO, 'tools', 'c5_frame_probe.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = {}
    for n in (256, 2048):
        r = mod.frame(n, reps=20)
        out[str(n)] = {k: round(v, 4) if isinstance(v, float) else v for k, v in r.items()}
    return out


def c4_single_gpu(stream, steps=10, warmup=5):
    """The unsharded C4 problem on the current GPU: steady-state ms per iteration + stage breakdown."""
    import torch
    from pyslam_amd import synthetic
    from pyslam_amd.device import DeviceProblem
    lp, _ = synthetic.stereo_ba(obs_per_lm=OBS_PER_LM, half_window=HALF_WINDOW, **C4)
    dev = DeviceProblem(lp, stream=stream)
    dev.eval_cost(True)                                      # (as Problem.solve does: the core then knows the start cost)
    dev.snapshot()
    sec, out = time_steps(dev, steps, warmup, torch.cuda.synchronize)
    dev.set_profiling(2)
    for _ in range(3):
        dev.restore(); dev.gn_iteration(0.0, PCG_TOL, PCG_MAX, True)
    st = dev.stage_times(reset=True)
    dev.set_profiling(0)
    traj = trajectory(dev, 4)
    stage = {k: v[0] / v[1] for k, v in st.items() if v[1] > 0}
    _, b_schur, _ = algorithmic_bytes(dev.info, out[2])
    sch_ms = stage.get('schur_pairs', 0.0)
    traffic, tsha, _ = pmc_traffic('k_schur_pairs_db', 'C4')
    ach = b_schur / (sch_ms * 1e-3) / 1e9 if sch_ms > 0 else 0.0
    roof = {'bound': 'hbm', 'kernel': 'k_schur_pairs_db', 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': round(ach / HBM_PEAK_GBS, 4), 'traffic': traffic, 'traffic_source_sha': tsha,
            'traffic_stale': bool(traffic is not None and tsha != kernel_source_sha()),
            'algorithmic_bytes_per_launch': int(b_schur), 'avg_launch_ms': round(sch_ms, 5),
            'note': 'pair + combine kernel, hipEvent pair around both on 3 untimed steps (profiling level 2)'}
    res = {'ms': round(sec * 1e3 / steps, 4), 'roofline': roof, 'pcg_iters': out[2], 'blocks': dev.info['num_obs'],
           'reduced_blocks': dev.info['reduced_nnzb'], 'device_bytes': dev.info['device_bytes'],
           'stage_ms': {k: round(v[0] / v[1], 4) for k, v in st.items() if v[1] > 0},
           'trajectory_ms_per_iter': traj['per_iter'], 'trajectory_pcg_iters': traj['pcg_iters']}
    dev.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--kf', type=int, default=None, help='override the keyframe count of the workload')
    ap.add_argument('--lm', type=int, default=None, help='override the landmark count of the workload')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-c4', action='store_true', help='skip the single-GPU C4 leg')
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
    # test hooks (a one-GPU box): PYSLAM_BENCH_ONE_GPU=1 puts every rank on cuda:0, PYSLAM_BENCH_BACKEND=gloo carries the
    # collectives over the host (RCCL refuses two ranks on one device) -- the multi-rank code path end to end, minus RCCL
    if os.environ.get('PYSLAM_BENCH_ONE_GPU'):
        local_rank = 0
    backend = os.environ.get('PYSLAM_BENCH_BACKEND', 'nccl')
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_sharded:
        import torch.distributed as dist
        if 'MASTER_ADDR' not in os.environ:
            os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29517', RANK='0', WORLD_SIZE='1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)

    cfg = dict(C4 if world > 1 else C3)
    name = 'C4' if world > 1 else 'C3'
    if args.kf: cfg['num_kf'] = args.kf
    if args.lm: cfg['num_lm'] = args.lm
    lp_full, _ = synthetic.stereo_ba(obs_per_lm=OBS_PER_LM, half_window=HALF_WINDOW, **cfg)

    stream = torch.cuda.current_stream().cuda_stream
    if dist is not None:
        from pyslam_amd.distributed import ShardedDeviceProblem, shard_landmarks
        lp = shard_landmarks(lp_full, rank, world)               # the FIXED problem, split N ways (strong scaling)
        dev = ShardedDeviceProblem(lp, dist)
    else:
        from pyslam_amd.device import DeviceProblem
        lp = lp_full
        dev = DeviceProblem(lp, stream=stream)
    info = dev.info
    dev.eval_cost(True)                                      # Problem.solve() evaluates the start cost first (reference problem.py:133)
    dev.snapshot()                                           # the common linearisation point

    def step():
        dev.restore()
        return dev.gn_iteration(0.0, PCG_TOL, PCG_MAX, True)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    core = dev.dev if hasattr(dev, 'dev') else dev
    for _ in range(args.warmup):
        out = step()
    # timed region: ONE hipEvent pair around the dominant (Schur) kernel, on every 4th iteration
    # (an event pair costs ~8 us of pipeline bubbles: the kernel is timed on every 4th step of the timed region)
    core.set_option('profile_every', 4)
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
    core.set_option('profile_every', 1)
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
    # reduced-solve launches per iteration (counter in ps_problem_info) and the lagged-inverse statistics
    i0 = problem_info(core)
    for _ in range(4):
        step()
    i1 = problem_info(core)
    traj = trajectory(dev) if dist is None else None         # (sharded: every rank would have to follow; single GPU only)
    info_after = {'cg_kernel_launches_per_iter': (i1['cg_kernel_launches'] - i0['cg_kernel_launches']) / 4.0,
                  'ldi': {k: i1[k] for k in ('ldi_solves', 'ldi_fallbacks', 'ldi_seeds')},
                  'solver': 'lagged dense inverse PCG, 2 launches per iteration' if i1['ldi_solves'] > i0['ldi_solves']
                  else 'two-level CG'}
    first_iter_ms = first_iter_its = None
    if dist is None:                                         # the moving phase of a solve: the same steps without the inverse
        core.set_option('lagged_inverse', 0)
        sec0, out0 = time_steps(dev, 10, 3, fence)
        core.set_option('lagged_inverse', 1)
        first_iter_ms, first_iter_its = sec0 * 1e3 / 10, out0[2]
    ranks_stage = None
    if dist is not None:                                     # every rank's stage times (incl. allreduce, pack_unpack) on rank 0
        mine = {k: round(v[0] / max(v[1], 1), 4) for k, v in stages.items() if v[1] > 0}
        ranks_stage = [None] * world
        dist.all_gather_object(ranks_stage, mine)

    if rank == 0:
        b_iter, b_schur, b_spmv = algorithmic_bytes(info, n_pcg)
        stage_ms = {k: (v[0] / max(v[1], 1)) for k, v in stages.items() if v[1] > 0}
        # dominant kernel: the Schur pair kernel is ONE launch per iteration; the CG is n_pcg + 2 launches
        sch = stage_ms.get('schur_pairs', 0.0)
        pcg_per_iter = stage_ms.get('pcg', 0.0) / max(n_pcg, 1)
        if sch >= pcg_per_iter * 1.0 and sch > 0:
            kern, dur_ms, nbytes = 'k_schur_pairs_db', sch, b_schur
        else:
            kern, dur_ms, nbytes = 'void k_cg_fused<6>', pcg_per_iter, b_spmv
        achieved = nbytes / (dur_ms * 1e-3) / 1e9 if dur_ms > 0 else 0.0
        traffic, traffic_sha, traffic_head = pmc_traffic(kern)
        sha = kernel_source_sha()
        if traffic is not None and traffic_sha != sha:
            print('bench.py: WARNING profiles/pmc_traffic.json was collected on kernel sources {} but this build is {}: '
                  'roofline.traffic is stale (re-run tools/collect_profiles.sh)'.format(traffic_sha, sha), file=sys.stderr)
        total_blocks = lp_full.num_obs
        line = {
            'metric': 'ms/LM-iter (Jac build + J^T J + Schur solve), stereo BA @ 500k residuals',
            'value': round(ms_per_step, 4), 'unit': 'ms/LM-iter', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 4),
            'higher_is_better': False, 'scaling': 'strong' if world > 1 else 'none', 'vs_baseline': None, 'dtype': 'f64',
            'data': 'synthetic',
            'config': {'workload': '{} stereo BA: {} keyframes x {} landmarks x {} obs/landmark = {} reprojection blocks '
                                   '(3 rows each), L2 loss, pose 0 constant{}'.format(
                                       name, cfg['num_kf'], cfg['num_lm'], OBS_PER_LM, total_blocks,
                                       '; FIXED problem, landmark shard 1/{} per GPU ({} blocks on rank 0)'.format(world, info['num_obs'])
                                       if world > 1 else ''),
                       'parallelism': 'landmark-sharded x{} + RCCL all-reduce of the reduced pose system (upper triangle)'.format(world)
                       if world > 1 else 'single GPU',
                       'pcg_tol': PCG_TOL, 'pcg_iters': n_pcg, 'pcg_relres': relres,
                       'reduced_blocks': info['reduced_nnzb'], 'schur_pairs': info['num_pairs'],
                       'cost_after_step': cost, 'step_norm': dx_norm},
            'residual_blocks_per_s': round(total_blocks / (ms_per_step * 1e-3), 1),
            'stage_ms': {k: round(v, 4) for k, v in stage_ms.items()},
            'stage_ms_note': 'schur_pairs: hipEvent pair inside the timed region, on every 4th step; the other stages and iteration_total '
                             '(GPU time of one iteration) from 5 extra untimed steps with an event pair around every stage',
            'value_note': 'steady state: every timed step restores the same linearisation point; with the lagged dense inverse '
                          '(round 3) that is the SETTLED phase of a solve -- the inverse of this very S preconditions the CG; '
                          'first_iteration_ms is the same step with the standard two-level CG (the moving phase), '
                          'trajectory_ms_per_iter a real solve from the perturbed start (both phases and the switch between them)',
            'iteration_algorithmic_GBps': round(b_iter / (ms_per_step * 1e-3) / 1e9, 2),
            'roofline': {'bound': 'hbm', 'kernel': kern, 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4),
                         'traffic': traffic, 'traffic_source_sha': traffic_sha, 'traffic_git_head': traffic_head,
                         'build_source_sha': sha, 'traffic_stale': bool(traffic is not None and traffic_sha != sha),
                         'algorithmic_bytes_per_launch': int(nbytes), 'avg_launch_ms': round(dur_ms, 5)},
        }
        # the kernel with the largest TOTAL time per iteration next to the largest single launch: the reduced solve's launches
        n_launch = info_after['cg_kernel_launches_per_iter']
        pcg_ms = stage_ms.get('pcg', 0.0)
        line['roofline_aggregate'] = {
            'kernels': 'reduced solve (' + info_after['solver'] + ')', 'launches_per_iteration': round(n_launch, 2),
            'total_ms_per_iteration': round(pcg_ms, 5), 'avg_launch_us': round(1e3 * pcg_ms / max(n_launch, 1), 3),
            'algorithmic_bytes_per_launch': int(b_spmv),
            'achieved_GBps': round(b_spmv * n_launch / max(pcg_ms * 1e-3, 1e-12) / 1e9, 1), 'peak': HBM_PEAK_GBS,
            'frac': round(b_spmv * n_launch / max(pcg_ms * 1e-3, 1e-12) / 1e9 / HBM_PEAK_GBS, 4),
            'bound': 'latency (dependent launches of ~5 us; DESIGN.md section 5), not bandwidth',
            'note': 'bytes per launch = one pass over S (288 B per block) + three vectors; the launches of the lagged-inverse '
                    'PCG alternate between that SpMV and a dense fp32 mat-vec of n^2 * 4 B (5.7 MB at C3)'}
        line['lagged_inverse'] = info_after['ldi']
        if first_iter_ms is not None:
            line['first_iteration_ms'] = round(first_iter_ms, 4)
            line['first_iteration_note'] = ('the same steady-state measurement with the lagged dense inverse switched off: what an '
                                            'iteration costs while a solve is still moving (the standard two-level CG, {} iterations); '
                                            '`value` is the settled phase, where the inverse preconditions ({} iterations)'.format(
                                                first_iter_its, n_pcg))
        if traj is not None:
            line['trajectory_ms_per_iter'] = traj
        if not args.no_c4 and not args.kf and not args.lm:
            if dist is None:
                dev.close()                                      # free the C3 tables first
            c4 = c4_single_gpu(stream)
            line['c4_single_gpu_ms'] = c4['ms']
            line['c4_single_gpu'] = c4
            if world > 1:                                    # same problem, same run, one GPU of this node: the strong-scaling base
                line['speedup_vs_c4_single_gpu'] = round(c4['ms'] / ms_per_step, 3)
        if ranks_stage is not None:
            line['stage_ms_per_rank'] = ranks_stage
            line['native_rccl'] = bool(getattr(dev, 'native', None) is not None)
            line['native_rccl_reason'] = None if line['native_rccl'] else getattr(dev, 'native_reason', None)
        if world == 1 and not args.kf and not args.lm:
            line['c5_solve_wall_ms'] = c5_frames()
        if not args.no_cpu_baseline and world == 1:
            line['cpu_baseline'] = cpu_baseline(lp)
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dev.close()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
