#!/usr/bin/env python
"""bench.py -- ms per Gauss-Newton/LM iteration on the BASELINE stereo-BA workloads.

    python bench.py --gpus N --steps K --warmup W

One "step" = one full iteration of the hot path: residuals + Jacobians + IRLS
weights, J^T J assembly, Schur elimination of the landmarks, block-PCG on the
reduced pose system (relative tolerance 1e-12, the parity setting),
back-substitution, retraction and the post-step cost pass -- exactly what the
reference's ``Problem.solve_one_iter`` + update does (pyslam/problem.py:145-156).

``value`` is the per-iteration cost of COLD, REFERENCE-TERMINATED SOLVES: the timed
region is whole solves from the perturbed start, each running the loop of
``Problem.solve`` (pyslam_amd/problem.py: device_solve -- the very function Problem.solve
calls; reference pyslam/problem.py:130-178) under the options of the reference's own
example (examples/stereo_ba.py:38-40: allow_nondecreasing_steps, max_nondecreasing_steps
= 3): start cost, then iterations until the reference's stopping rule ends the solve (4 at
C3).  Before every solve the solver state is cleared (ps_reset_solver_state: no lagged
coarse factor, no lagged inverse, no held operator, no launch-count prediction, no cost
history) and the start parameters are uploaded again -- both outside the clock; everything
a solve itself does (start cost, iterations, best-parameter snapshots, the final restore)
is inside.  value = sum of solve times / sum of iterations, over exactly K iterations (the
last solve is cut at the K-th).  The W warm-up steps are whole untimed solves (at least one).
Extra keys keep the figures of earlier rounds: ``steady_same_point_ms`` (every step restores
the same linearisation point; with the lagged inverse that is the settled phase no
reference-terminated solve reaches), ``moving_same_point_ms`` (the same with the inverse off).

N = 1: workload C3 of SURVEY.md section 8d (200 keyframes, 50 000 landmarks,
500 000 reprojection blocks = 1.5 M residual rows), the configuration
BASELINE.json's metric is quoted on; ``c4_single_gpu`` carries the same cold-solve
measurement of the north star's 2 000-keyframe / 500 000-landmark problem on this one
GPU (the strong-scaling base), ``cold_solve_wall_ms`` the wall clock of ``Problem.solve()``
through the public API (objects -> lowering -> ps_problem_create -> iterations).

N > 1 (launched by torch.distributed.run, one rank per GPU): STRONG scaling of
the fixed C4 problem (2 000 keyframes, 500 000 landmarks, 5 M blocks; BASELINE.json
configs[3]), the same cold solves: every rank generates the same problem and keeps landmark
shard ``rank`` of N (``shard_landmarks``); the reduced pose system [upper(S) | g | cost | flag]
is exchanged once per iteration by the HIP core itself on the solver's stream -- N > 1: an RCCL
all-gather of the ranks' band segments, summed by every rank in rank order (``--exchange allreduce``:
one sum all-reduce of the whole packed system) -- and then solved redundantly on every rank; a
second small all-reduce sums the shards' cost and landmark step norm.  Rank 0 also times the
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
    """The source hash COMPILED INTO the loaded library (ps_build_sha, -DPS_BUILD_SHA of __graft_entry__.build()), after
    checking that it equals the hash of the sources on disk: a bench line is tied to the binary that produced it."""
    import __graft_entry__ as ge
    from pyslam_amd import _native as nat
    built = nat.load().ps_build_sha().decode()
    disk = ge.source_sha()
    if built != disk:
        raise SystemExit('bench.py: the loaded library was built from sources {} but the files on disk hash to {}: '
                         'rebuild (python -c "import __graft_entry__ as g; g.build()")'.format(built, disk))
    return built


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
        if kernel not in table:                                         # (rocprof names templates 'void k<6, 8>(...)')
            kernel = next(k for k in table if kernel in k)
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


def example_options():
    """Options of the reference's stereo-BA example (examples/stereo_ba.py:38-40) + the parity tolerance of the reduced solve."""
    from pyslam_amd.problem import Options
    opt = Options()
    opt.allow_nondecreasing_steps = True
    opt.max_nondecreasing_steps = 3
    opt.pcg_tol, opt.pcg_max_iters = PCG_TOL, PCG_MAX
    return opt


def cold_solves(dev, start, total_iters, fence, warm_solves=1):
    """Whole solves from the perturbed start (`start` = (poses, points) host tables), solver state cleared before each, the
    loop of Problem.solve (pyslam_amd.problem.device_solve: ps_solve on one GPU, the Python loop over ps_gn_iteration on the
    sharded driver) under the reference example's options, until exactly `total_iters` iterations have run (the last
    solve is cut there).  Timed per solve: start cost + iterations + snapshots + final restore + the synchronisation that
    ends it; NOT timed: ps_reset_solver_state and the upload of the start parameters.
    -> dict(seconds, iterations, solves, per_call_ms (median by position in the solve), pcg_iters, cost_history)."""
    from pyslam_amd.problem import device_solve
    core = dev.dev if hasattr(dev, 'dev') else dev

    def one(max_its):
        opt = example_options()
        if max_its is not None:
            opt.max_iters = max_its - 1                   # (the loop stops once its counter exceeds max_iters)
        core.reset_solver_state()
        core.set_params(*start)
        fence()
        call_ms = []
        t0 = time.perf_counter()
        hist, stats = device_solve(dev, opt, call_ms=call_ms)
        fence()                                           # (the final restore of the best parameters is enqueue-only)
        return time.perf_counter() - t0, hist, [(ms, st[0]) for ms, st in zip(call_ms, stats)]

    for _ in range(max(1, warm_solves)):
        one(None)
    sec, its, solves, calls_by_pos, hist0, pcg0, per_solve, per_solve_its = 0.0, 0, 0, [], None, None, [], []
    import gc
    gc_was = gc.isenabled()
    gc.disable()              # (as timeit does: a collection of the interpreter inside a 1.3 ms solve is not the solver's time)
    while its < total_iters:
        dt, hist, calls = one(total_iters - its)
        sec += dt; its += len(calls); solves += 1
        per_solve.append(round(dt * 1e3, 4)); per_solve_its.append(len(calls))
        if hist0 is None:
            hist0, pcg0 = hist, [c[1] for c in calls]
        for k, c in enumerate(calls):
            if k >= len(calls_by_pos):
                calls_by_pos.append([])
            calls_by_pos[k].append(c[0])
    if gc_was:
        gc.enable()
    return {'seconds': sec, 'iterations': its, 'solves': solves, 'per_solve_ms': per_solve, 'per_solve_iterations': per_solve_its,
            # SURVEY 8d asks for a median: ms per iteration of every solve, median over the solves (`value` stays the mean over
            # exactly K iterations, as the bench contract wants it; one slow solve moves the mean, not this)
            'ms_median_of_solves': round(float(np.median([m / max(n, 1) for m, n in zip(per_solve, per_solve_its)])), 4),
            'per_call_ms': [round(float(np.median(c)), 4) for c in calls_by_pos],      # median over the solves, by call index
            'pcg_iters': pcg0, 'cost_history': hist0}


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


def stage_breakdown(dev, start, fence):
    """One more (untimed) cold solve with an event pair around every stage: mean ms per stage and iteration."""
    core = dev.dev if hasattr(dev, 'dev') else dev
    core.set_option('profile_every', 1)
    dev.set_profiling(2)
    dev.stage_times(reset=True)
    cold_solves(dev, start, 4, fence, warm_solves=0)
    st = dev.stage_times(reset=True)
    dev.set_profiling(0)
    return st


STAGE_KEYS = ('landmark_pass', 'pose_pass', 'schur_pairs', 'pose_factors', 'pcg', 'backsub', 'update', 'cost', 'allreduce', 'pack_unpack')   # ('cg_kernel' lies inside 'pcg')


def name_stage_totals(stage):
    """The event pair 'iteration_total' of the core spans one ps_gn_iteration CALL.  Inside ps_solve the next iteration's
    linearisation is enqueued behind a converged tail while the host waits (csrc/ps_host_cg.h: wait_published), i.e. AFTER the
    call's pair has been closed and BEFORE the next call's pair opens: from the second iteration of a solve on the pair holds the
    reduced solve and the tail only (C4: 1.2 ms beside stages that sum to 1.9 -- round-4 verdict).  So the line says which is
    which: `stages_sum` = the GPU time of an iteration as the sum of its stages' own event pairs, `call_event_pair` = the core's
    pair (excludes a speculatively enqueued linearisation)."""
    out = dict(stage)
    if 'iteration_total' in out:
        out['call_event_pair'] = out.pop('iteration_total')
    # (the sharded driver's reduced solve has no event pair of its own: its stages do not add up to an iteration, so the sum is named
    #  for what it is)
    out['stages_sum' if 'pcg' in stage or 'schur_pairs' not in stage else 'stages_sum_without_reduced_solve'] = \
        sum(v for k, v in stage.items() if k in STAGE_KEYS)
    return out


def pmc_config(cfg):
    """Which table of profiles/pmc_traffic.json belongs to a workload: None = the top level (C3), 'C4', or no table at all."""
    size = (cfg['num_kf'], cfg['num_lm'])
    return None if size == (C3['num_kf'], C3['num_lm']) else ('C4' if size == (C4['num_kf'], C4['num_lm']) else 'not collected')


# ---- the roofline object: the kernel that MEASURABLY takes the largest share of an iteration -------------------------------------
# (round-5 verdict: the line used to hard-code the Schur pair kernel, which at C3 is not the largest.)  Every candidate has an event
# pair of its own in the core (include/pyslam_hip.h: PS_ST_*): 'schur_pairs' and 'cg_kernel' are sampled INSIDE the timed region
# (profiling level 1, every 4th linearisation), the three passes over the observations come from the untimed solve with a pair
# around every stage.  ms per iteration each; the largest is `roofline`, the Schur and CG figures stay under keys of their own.
CANDIDATE_STAGES = ('schur_pairs', 'cg_kernel', 'landmark_pass', 'pose_pass', 'backsub')
# per-iteration exchange of the one-launch solvers, measured on its own (tools/probes/allgather_probe.hip):
# one write-through store + one agent-scope load across the fabric
EXCHANGE_US = (2.0, 2.7)


def cg_kernel_name(persist, explicit):
    """rocprof's name (prefix) of the kernel(s) the 'cg_kernel' pair spans."""
    if explicit:
        return 'k_xcg_persist' if persist else 'k_xcg_fused1'
    return 'k_cg_persist' if persist else 'k_cg_fused_lds'


def pick_dominant(stage_ms):
    """-> (stage key, ms per iteration) of the candidate with the largest time per iteration (ties: the first in CANDIDATE_STAGES)."""
    best = max(CANDIDATE_STAGES, key=lambda k: (stage_ms.get(k, 0.0), -CANDIDATE_STAGES.index(k)))
    return best, stage_ms.get(best, 0.0)


def _traffic_fields(kernel, config):
    traffic, tsha, thead = pmc_traffic(kernel, config)
    sha = kernel_source_sha()
    if traffic is not None and tsha != sha:
        print('bench.py: WARNING profiles/pmc_traffic.json ({}) was collected on kernel sources {} but this build is {}: '
              'roofline.traffic is stale (re-run tools/collect_profiles.sh)'.format(config, tsha, sha), file=sys.stderr)
    return {'traffic': traffic, 'traffic_source_sha': tsha, 'traffic_git_head': thead, 'build_source_sha': sha,
            'traffic_stale': bool(traffic is not None and tsha != sha)}


def schur_roofline(info, sch_ms, config, n_pcg=0, note=None):
    """roofline object of the Schur pair kernel (+ its combine launch in tiled mode): SURVEY 8d bytes 144 N + 288 (nnzb - P)."""
    _, b_schur, _ = algorithmic_bytes(info, n_pcg)
    ach = b_schur / (sch_ms * 1e-3) / 1e9 if sch_ms > 0 else 0.0
    roof = {'bound': 'hbm', 'kernel': 'k_schur_pairs_db', 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': round(ach / HBM_PEAK_GBS, 4)}
    roof.update(_traffic_fields('k_schur_pairs_db', config))
    roof.update({'algorithmic_bytes_per_launch': int(b_schur), 'avg_launch_ms': round(sch_ms, 5)})
    if note:
        roof['note'] = note
    return roof


def cg_roofline(info, cg_ms, config, n_pcg, persist, explicit, n_launch=1.0, dof=6):
    """roofline object of the kernel(s) that run the CG iterations of the reduced solve (event pair 'cg_kernel': those launches
    alone, without the set-up kernels and the recovery).  Bytes: SURVEY 8d's PCG term, 288 B x nnzb(S) x CG iterations -- what a
    CG that streamed S once per iteration would move; the one-launch forms read S ONCE into registers and are bound by one
    exchange over the fabric per CG iteration, so `bound` says 'latency' and the object carries that ceiling beside the HBM figures."""
    name = cg_kernel_name(persist, explicit)
    b = 8 * dof * dof * info['reduced_nnzb'] * max(n_pcg, 0)
    ach = b / (cg_ms * 1e-3) / 1e9 if cg_ms > 0 else 0.0
    roof = {'bound': 'latency', 'kernel': name, 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': round(ach / HBM_PEAK_GBS, 4)}
    roof.update(_traffic_fields(name, config))
    roof.update({'algorithmic_bytes_per_launch': int(b), 'avg_launch_ms': round(cg_ms / max(n_launch, 1.0), 5),
                 'launches_per_iteration': round(n_launch, 2), 'cg_iterations_per_gn_iteration': n_pcg,
                 'ms_per_gn_iteration': round(cg_ms, 5)})
    if persist:
        lo, hi = EXCHANGE_US
        roof['latency_ceiling'] = {
            'what': 'one exchange of the products over the fabric per CG iteration (write-through store + agent-scope load), measured '
                    'on its own: tools/probes/allgather_probe.hip',
            'exchange_us': [lo, hi], 'floor_ms': round(n_pcg * lo * 1e-3, 5),
            'frac_of_floor': round(n_pcg * lo * 1e-3 / cg_ms, 4) if cg_ms > 0 else 0.0}
        roof['note'] = ('S is read ONCE into registers (288 B x nnzb = {} bytes; `traffic` is what the counters see per launch); `achieved` '
                        'prices the launch on the bytes a streaming CG would move and is not a bandwidth claim'.format(8 * dof * dof * info['reduced_nnzb']))
    else:
        roof['bound'] = 'latency'
        roof['note'] = 'one dependent memory round trip per launch (DESIGN.md section 5), not bandwidth'
    return roof


def pass_roofline(key, info, ms, config):
    """roofline object of one of the three passes over the observations (bytes per unit: DESIGN.md section 3's kernel table)."""
    N, L, P = info['num_obs'], info['num_var_points'], info['num_reduced']
    name, b = {'landmark_pass': ('k_landmark_pass_packed', 160 * N + 72 * L),
               'pose_pass': ('k_pose_pass', 32 * N + 336 * P),
               'backsub': ('k_backsub_packed', 128 * N + 24 * L + 48 * P)}[key]
    ach = b / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    roof = {'bound': 'hbm', 'kernel': name, 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(ach / HBM_PEAK_GBS, 4)}
    roof.update(_traffic_fields(name, config))
    roof.update({'algorithmic_bytes_per_launch': int(b), 'avg_launch_ms': round(ms, 5)})
    return roof


def roofline_objects(info, stage_ms, config, n_pcg, persist, explicit, n_launch=1.0, schur_note=None):
    """-> dict(roofline = the dominant candidate's object, roofline_schur, roofline_cg, roofline_candidates_ms)."""
    cand = {k: round(stage_ms.get(k, 0.0), 5) for k in CANDIDATE_STAGES}
    key, _ = pick_dominant(stage_ms)
    schur = schur_roofline(info, stage_ms.get('schur_pairs', 0.0), config, n_pcg, note=schur_note)
    cg = cg_roofline(info, stage_ms.get('cg_kernel', 0.0), config, n_pcg, persist, explicit, n_launch)
    dom = schur if key == 'schur_pairs' else cg if key == 'cg_kernel' else pass_roofline(key, info, stage_ms.get(key, 0.0), config)
    dom = dict(dom)
    dom['selected_by'] = ('largest time per iteration among the event pairs {} (ms per iteration: {})'.format(', '.join(CANDIDATE_STAGES), cand))
    return {'roofline': dom, 'roofline_schur': schur, 'roofline_cg': cg, 'roofline_candidates_ms': cand}


def c4_single_gpu(stream, iters=8):
    """The unsharded C4 problem on the current GPU: the same cold, reference-terminated solves + stage breakdown."""
    import torch
    from pyslam_amd import synthetic
    from pyslam_amd.device import DeviceProblem
    lp, _ = synthetic.stereo_ba(obs_per_lm=OBS_PER_LM, half_window=HALF_WINDOW, **C4)
    dev = DeviceProblem(lp, stream=stream)
    start = (lp.poses.copy(), lp.points.copy())
    i0 = problem_info(dev)
    cold = cold_solves(dev, start, iters, torch.cuda.synchronize)
    i1 = problem_info(dev)
    st = stage_breakdown(dev, start, torch.cuda.synchronize)
    n_its = st.get('iteration_total', (0.0, 0))[1]
    # per ITERATION of the untimed solve (a stage that does not run in every iteration is spread over them, as in the C3 leg)
    stage = {k: v[0] / (n_its if (n_its > 0 and k != 'iteration_total') else v[1]) for k, v in st.items() if v[1] > 0}
    n_pcg = int(round(float(np.mean(cold['pcg_iters'])))) if cold['pcg_iters'] else 0
    roofs = roofline_objects(dev.info, stage, 'C4', n_pcg, persist=i1['cg_persist_solves'] > i0['cg_persist_solves'],
                             explicit=i1['xcg_fused_solves'] > i0['xcg_fused_solves'],
                             schur_note='pair + combine kernel, hipEvent pair around both on the iterations of one untimed cold solve (profiling level 2)')
    # the figures of earlier rounds: the same linearisation point restored before every step (coarse inverse held)
    dev.reset_solver_state(); dev.set_params(*start)
    dev.eval_cost(True); dev.snapshot()
    sec, out = time_steps(dev, 10, 5, torch.cuda.synchronize)
    res = {'ms': round(cold['seconds'] * 1e3 / cold['iterations'], 4), 'iterations': cold['iterations'], 'solves': cold['solves'],
           'per_call_ms': cold['per_call_ms'], 'pcg_iters': cold['pcg_iters'], 'cost_history': cold['cost_history'],
           'steady_same_point_ms': round(sec * 1e3 / 10, 4), 'steady_same_point_pcg_iters': out[2],
           'roofline': roofs['roofline'], 'roofline_schur': roofs['roofline_schur'], 'roofline_cg': roofs['roofline_cg'],
           'roofline_candidates_ms': roofs['roofline_candidates_ms'],
           'ms_median_of_solves': cold['ms_median_of_solves'], 'per_solve_ms': cold['per_solve_ms'],
           'blocks': dev.info['num_obs'], 'reduced_blocks': dev.info['reduced_nnzb'], 'device_bytes': dev.info['device_bytes'],
           'stage_ms': {k: round(v, 4) for k, v in name_stage_totals(stage).items()}}
    dev.close()
    return res


def cold_solve_wall(cfg, reps=2):
    """Wall clock of Problem.solve() through the public API on the workload `cfg`, fresh Problem, fresh handle: residual-block
    objects -> lowering (the walk over the blocks) -> ps_problem_create (structure build + upload) -> start cost + iterations
    -> write-back into the parameter objects; the reference's example options.  Run twice: the second run shows what the
    process-wide one-off costs (code-object load, first allocations) were."""
    import types
    import torch
    import pyslam.problem as P
    import pyslam.residuals as R
    import pyslam.losses as Ls
    import pyslam.sensors as S
    import liegroups as G
    from pyslam_amd import synthetic
    import pyslam_amd.problem as core_problem
    import pyslam_amd.device as core_device
    ns = types.SimpleNamespace(Problem=P.Problem, Options=P.Options, StereoCamera=S.StereoCamera, PoseResidual=R.PoseResidual,
                               PoseToPoseResidual=R.PoseToPoseResidual, PoseToPoseOrientationResidual=R.PoseToPoseOrientationResidual,
                               ReprojectionResidual=R.ReprojectionResidual, L2Loss=Ls.L2Loss, L1Loss=Ls.L1Loss, CauchyLoss=Ls.CauchyLoss,
                               HuberLoss=Ls.HuberLoss, TukeyLoss=Ls.TukeyLoss, TDistributionLoss=Ls.TDistributionLoss,
                               SE3=G.SE3, SO3=G.SO3, SE2=G.SE2, SO2=G.SO2)
    lp, _ = synthetic.stereo_ba(obs_per_lm=OBS_PER_LM, half_window=HALF_WINDOW, **cfg)
    runs = []
    for rep in range(reps):
        t0 = time.perf_counter()
        problem = synthetic.to_objects(lp, ns, example_options())
        t_objects = time.perf_counter() - t0
        marks = {}
        # split the solve by wrapping the three phases (lowering, create, device loop) with clocks
        orig_lower, orig_make, orig_loop = problem._lower, problem._make_device, core_problem.device_solve

        def lower(*a, **k):
            t = time.perf_counter(); out = orig_lower(*a, **k); marks['lowering'] = time.perf_counter() - t; return out

        def make(*a, **k):
            t = time.perf_counter(); out = orig_make(*a, **k); torch.cuda.synchronize()
            marks['create'] = time.perf_counter() - t; return out

        def loop(dev, opt):
            t = time.perf_counter(); ms = []; out = orig_loop(dev, opt, call_ms=ms); torch.cuda.synchronize()
            marks['device_loop'] = time.perf_counter() - t; marks['calls'] = [c * 1e-3 for c in ms]; return out
        problem._lower, problem._make_device, core_problem.device_solve = lower, make, loop
        try:
            t0 = time.perf_counter()
            problem.solve()
            t_solve = time.perf_counter() - t0
        finally:
            core_problem.device_solve = orig_loop
        calls = marks.get('calls', [])
        runs.append({'objects_s': round(t_objects, 3), 'solve_wall_ms': round(t_solve * 1e3, 2),
                     'lowering_ms': round(marks.get('lowering', 0.) * 1e3, 2), 'create_ms': round(marks.get('create', 0.) * 1e3, 2),
                     'device_loop_ms': round(marks.get('device_loop', 0.) * 1e3, 3),
                     'first_call_ms': round(calls[0] * 1e3, 3) if calls else None,
                     'other_calls_ms': [round(c * 1e3, 3) for c in calls[1:]],
                     'write_back_and_rest_ms': round((t_solve - marks.get('lowering', 0.) - marks.get('create', 0.) - marks.get('device_loop', 0.)) * 1e3, 2),
                     'iterations': len(calls)})
        if problem._device is not None:
            problem._device.close()
        del problem
    return {'first_in_process': runs[0], 'second': runs[-1] if reps > 1 else None,
            'note': 'Problem.solve() on a fresh Problem and a fresh device handle (public API: block objects -> lowering -> '
                    'ps_problem_create -> start cost + iterations -> write-back); objects_s = building the Python block objects, '
                    'not part of solve()'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--kf', type=int, default=None, help='override the keyframe count of the workload')
    ap.add_argument('--lm', type=int, default=None, help='override the landmark count of the workload')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-c4', action='store_true', help='skip the single-GPU C4 leg')
    ap.add_argument('--no-wall', action='store_true', help='skip the Problem.solve() wall-clock leg (500 000 Python block objects)')
    ap.add_argument('--force-sharded', action='store_true',
                    help='use the multi-GPU driver (RCCL all-reduce) even with one rank (testing)')
    ap.add_argument('--shard-order', default='first_pose', choices=['first_pose', 'index'],
                    help="how landmarks are cut into shards (pyslam_amd/distributed.py: landmark_owner_lists)")
    ap.add_argument('--exchange', default=None, choices=['allreduce', 'segments'])
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
    sha = kernel_source_sha()                                # (refuses a library that does not match the sources on disk)
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
        lp = shard_landmarks(lp_full, rank, world, order=args.shard_order)       # the FIXED problem, split N ways (strong scaling)
        # (N > 1: the segment exchange inside the core -- a rank sends the 3.2 MB band segment its shard touches instead of joining a
        #  23.5 MB all-reduce; --exchange allreduce / PYSLAM_AMD_EXCHANGE choose otherwise)
        dev = ShardedDeviceProblem(lp, dist, exchange=args.exchange or os.environ.get('PYSLAM_AMD_EXCHANGE') or ('segments' if world > 1 else None))
    else:
        from pyslam_amd.device import DeviceProblem
        lp = lp_full
        dev = DeviceProblem(lp, stream=stream)
    info = dev.info
    start = (lp.poses.copy(), lp.points.copy())              # the perturbed start every cold solve begins at (this rank's shard)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    core = dev.dev if hasattr(dev, 'dev') else dev
    # timed region: cold solves (module docstring); ONE hipEvent pair per four linearisations (a pair costs ~8 us of pipeline
    # bubbles): around the Schur pair kernel on every 8th, around the CG launch on the 8th's in between
    core.set_option('profile_every', 8)
    dev.set_profiling(1)
    dev.stage_times(reset=True)
    warm_solves = max(1, (args.warmup + 3) // 4)
    # (the warm-up solves run inside cold_solves, before its clock; their Schur timings are discarded below)
    cold = cold_solves(dev, start, args.steps, fence, warm_solves=warm_solves)
    stages = dev.stage_times(reset=True)
    dev.set_profiling(0)
    elapsed = cold['seconds']
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed * 1e3 / cold['iterations']
    n_pcg_list = cold['pcg_iters']
    n_pcg = int(round(float(np.mean(n_pcg_list)))) if n_pcg_list else 0
    # untimed: one more cold solve with an event pair around every stage, for the breakdown only
    # (per ITERATION of that solve: a stage that does not run in every iteration -- the cost-only pass since round 5: the cost
    #  after a step is summed by the next iteration's landmark pass, run in the tail -- is spread over the solve's iterations)
    detail = stage_breakdown(dev, start, fence)
    n_detail_its = detail.get('iteration_total', (0.0, 0))[1]
    for k, v in detail.items():
        if k not in ('schur_pairs', 'cg_kernel') and v[1] > 0:
            stages[k] = (v[0], n_detail_its if (n_detail_its > 0 and k != 'iteration_total') else v[1])
    i0 = problem_info(core)
    counted = cold_solves(dev, start, 4, fence, warm_solves=1)
    i1 = problem_info(core)
    # launches of the reduced solve per iteration, spare (gated, no-op) launches past convergence included.  cold_solves always runs
    # one untimed solve in front of the counted one: both are between the two counter reads.  (Rounds 3-4 divided by the counted
    # solve's iterations only and reported twice the launches -- 47.5 for 23.8 -- hence half the time per launch.)
    n_launch = (i1['cg_kernel_launches'] - i0['cg_kernel_launches']) / float(2 * max(counted['iterations'], 1))

    # the figures of rounds 1-3, kept as extra keys: the same linearisation point restored before every step
    steady = moving = None
    if dist is None:
        core.reset_solver_state(); core.set_params(*start)
        dev.eval_cost(True); dev.snapshot()
        sec1, out1 = time_steps(dev, 20, 8, fence)
        core.set_option('lagged_inverse', 0)
        core.reset_solver_state(); core.set_params(*start)
        dev.eval_cost(True); dev.snapshot()
        sec0, out0 = time_steps(dev, 10, 3, fence)
        core.set_option('lagged_inverse', 1)
        steady = (sec1 * 1e3 / 20, out1[2]); moving = (sec0 * 1e3 / 10, out0[2])
    ranks_stage = None
    if dist is not None:                                     # every rank's stage times (incl. allreduce, pack_unpack) on rank 0
        mine = {k: round(v[0] / max(v[1], 1), 4) for k, v in stages.items() if v[1] > 0}
        ranks_stage = [None] * world
        dist.all_gather_object(ranks_stage, mine)

    if rank == 0:
        b_iter, b_schur, b_spmv = algorithmic_bytes(info, n_pcg)
        stage_ms = {k: (v[0] / max(v[1], 1)) for k, v in stages.items() if v[1] > 0}
        sch = stage_ms.get('schur_pairs', 0.0)
        total_blocks = lp_full.num_obs
        hist = cold['cost_history']
        line = {
            'metric': 'ms/LM-iter (Jac build + J^T J + Schur solve), stereo BA @ 500k residuals',
            'value': round(ms_per_step, 4), 'unit': 'ms/LM-iter', 'n_gpus': world,
            'steps': cold['iterations'], 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 4),
            'higher_is_better': False, 'scaling': 'strong' if world > 1 else 'none', 'vs_baseline': None, 'dtype': 'f64',
            'data': 'synthetic',
            'config': {'workload': '{} stereo BA: {} keyframes x {} landmarks x {} obs/landmark = {} reprojection blocks '
                                   '(3 rows each), L2 loss, pose 0 constant{}'.format(
                                       name, cfg['num_kf'], cfg['num_lm'], OBS_PER_LM, total_blocks,
                                       '; FIXED problem, landmark shard 1/{} per GPU ({} blocks on rank 0)'.format(world, info['num_obs'])
                                       if world > 1 else ''),
                       'parallelism': ('landmark-sharded x{} (landmarks cut by first observing pose) + '.format(world) +
                                       ('all-gather of the ranks\' band segments of the reduced pose system, summed in a fixed order'
                                        if (getattr(dev, 'segments', None) is not None or getattr(dev, 'core_segments', False)) else
                                        'RCCL all-reduce of the reduced pose system (upper triangle)'))
                       if world > 1 else 'single GPU',
                       'timed_region': '{} cold solves from the perturbed start = {} iterations: solver state cleared before each solve '
                                       '(ps_reset_solver_state), the loop of Problem.solve under the options of reference '
                                       'examples/stereo_ba.py:38-40 (allow_nondecreasing_steps, max_nondecreasing_steps = 3); '
                                       '{} untimed warm-up solve(s)'.format(cold['solves'], cold['iterations'], warm_solves),
                       'pcg_tol': PCG_TOL, 'pcg_iters_per_call': n_pcg_list, 'cost_history': hist,
                       'reduced_blocks': info['reduced_nnzb'], 'schur_pairs': info['num_pairs']},
            'residual_blocks_per_s': round(total_blocks / (ms_per_step * 1e-3), 1),
            'cold_solve': {'solves': cold['solves'], 'iterations': cold['iterations'],
                           'ms_per_solve': round(elapsed * 1e3 / cold['solves'], 4), 'per_solve_ms': cold.get('per_solve_ms'),
                           'per_call_ms': cold['per_call_ms'],
                           'per_call_note': 'host wall clock of each ps_gn_iteration call by its position in the solve (median over the '
                                            'solves); the solve time also holds the start-cost pass, the best-parameter snapshots and '
                                            'the final restore'},
            'stage_ms': {k: round(v, 4) for k, v in name_stage_totals(stage_ms).items()},
            'stage_ms_note': 'schur_pairs, cg_kernel (the CG launch alone): hipEvent pairs inside the timed region, each on every 8th linearisation (alternating: one pair per four iterations); the other stages from one '
                             'more untimed cold solve with an event pair around every stage, per iteration of that solve. landmark_pass: '
                             'since round 5 the pass of the NEXT iteration runs in the tail and sums the cost after the step on its way '
                             '(one evaluation of every observation per iteration); cost = what is left of the cost-only pass (the last '
                             'iteration of a solve), spread over the iterations. stages_sum = GPU time of an iteration (sum of '
                             'the stage pairs); call_event_pair = the pair around one ps_gn_iteration call, which EXCLUDES a linearisation '
                             'enqueued speculatively behind the previous call (every iteration of a solve but the first)',
            'metric_definition': {'version': 2, 'since_round': 4,
                                  'text': 'ms per iteration of COLD, reference-terminated solves (rounds 1-3: a steady-state step at a '
                                          'restored linearisation point, kept as steady_same_point_ms); ps_reset_solver_state and the '
                                          'parameter upload of each solve are outside the clock, everything else of the solve inside'},
            'iteration_algorithmic_GBps': round(b_iter / (ms_per_step * 1e-3) / 1e9, 2),
        }
        # `roofline` = the kernel with the largest measured time per iteration (roofline_objects above); the Schur pair kernel and
        # the CG launch keep objects of their own whichever of them that is
        persist = i1.get('cg_persist_solves', 0) > i0.get('cg_persist_solves', 0)
        explicit = i1.get('xcg_fused_solves', 0) > i0.get('xcg_fused_solves', 0)
        line.update(roofline_objects(info, stage_ms, pmc_config(cfg), n_pcg, persist, explicit, n_launch))
        pcg_ms = stage_ms.get('pcg', 0.0)
        line['reduced_solve_stage'] = {
            'ms_per_iteration': round(pcg_ms, 5), 'cg_kernel_ms': round(stage_ms.get('cg_kernel', 0.0), 5),
            'launches_of_the_cg_per_iteration': round(n_launch, 2), 'cg_iterations': n_pcg,
            'note': 'event pair around the whole reduced solve (set-up kernels + CG launch(es) + recovery) on one untimed cold solve; '
                    'cg_kernel_ms: the pair around the CG launch(es) alone, sampled inside the timed region'}
        line['value_median'] = {
            'ms_per_iteration_median_of_solves': cold['ms_median_of_solves'],
            'sum_of_per_call_medians_over_calls': round(float(np.sum(cold['per_call_ms'])) / max(len(cold['per_call_ms']), 1), 4),
            'note': 'SURVEY 8d words the metric as a median: per-solve ms / iterations, median over the {} timed solves; and the per-call '
                    'medians (ps_gn_iteration calls only: no start cost, snapshots, restore) averaged over a solve. `value` is the MEAN over '
                    'exactly {} iterations (bench contract)'.format(cold['solves'], cold['iterations'])}
        line['lagged_inverse'] = {k: i1[k] for k in ('ldi_solves', 'ldi_fallbacks', 'ldi_seeds')}
        if steady is not None:
            line['steady_same_point_ms'] = round(steady[0], 4)
            line['steady_same_point_note'] = ('rounds 1-3 `value`: every step restores the same linearisation point, so the lagged dense '
                                              'inverse of this very S preconditions ({} CG iterations) -- the settled phase, which a '
                                              'reference-terminated solve of this workload never reaches'.format(steady[1]))
            line['moving_same_point_ms'] = round(moving[0], 4)
            line['moving_same_point_note'] = 'the same with the lagged dense inverse off ({} CG iterations; round 3: first_iteration_ms)'.format(moving[1])
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
            if not args.no_wall:
                line['cold_solve_wall_ms'] = {'C3': cold_solve_wall(C3)}
                if not args.no_c4:      # once: building the 5 M block objects (not part of solve()) takes ~12 s
                    w4 = cold_solve_wall(C4, reps=1)
                    line['cold_solve_wall_ms']['C4'] = {'run': w4['first_in_process'], 'note': w4['note']}
        if not args.no_cpu_baseline and world == 1:
            line['cpu_baseline'] = cpu_baseline(lp)
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dev.close()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
