"""Problem / Options: the reference's user-facing API over the HIP core.

Same public surface as reference pyslam/problem.py:11-409 (Options fields,
``add_residual_block``, ``initialize_params``, ``set_parameters_constant`` /
``_variable``, ``eval_cost``, ``solve``, ``solve_one_iter``,
``compute_covariance``, ``get_covariance_block``, ``summary`` and the public
attributes the reference's tests read).  What differs is where the work runs:

* recognised blocks (reprojection, pose-pose, pose prior -- ``KIND`` tags) are
  lowered ONCE into HBM tables (pyslam_amd/lowering.py) and the whole
  Gauss-Newton iteration -- residuals, Jacobians, IRLS weights, J^T J, landmark
  elimination, block-PCG, back-substitution, retraction, post-step cost -- runs
  on the MI355X (include/pyslam_hip.h: ps_gn_iteration); the host sees two
  scalars per iteration and runs the reference's termination logic on them;
* problems with user-defined blocks / parameters are evaluated through the
  block protocol on the host (that is user Python code) and only the normal
  equations are formed and solved on the device (ps_dense_normal_solve; beyond 2048
  unknowns J goes up as CSR and the device runs CG on J^T J: ps_sparse_normal_solve).

There is no CPU solver: without the HIP library or a GPU, solve() raises.
"""
import copy
import time

import numpy as np

from pyslam_amd.losses import L2Loss
from pyslam_amd import lowering
from pyslam_amd.lowering import NotLowerable


class Options:
    """Optimisation options (reference pyslam/problem.py:11-37) plus the knobs
    of the device solver, whose defaults reproduce the reference (lambda = 0)."""

    def __init__(self):
        self.max_iters = 100
        self.min_update_norm = 1e-6
        self.min_cost = 1e-12
        self.min_cost_decrease = 0.9

        self.linesearch_alpha = 0.8
        self.linesearch_max_iters = 10
        self.linesearch_min_cost_decrease = 0.9

        self.allow_nondecreasing_steps = False
        self.max_nondecreasing_steps = 3

        self.num_threads = 1            # kept for API compatibility; the device path ignores it

        # build-only
        self.lm_lambda = 0.             # Marquardt damping lambda * diag(J^T J); 0 = Gauss-Newton
        self.pcg_tol = None             # relative (preconditioned) residual of the reduced solve; None: the core's default -- 1e-12
                                        # for bundle adjustments, 1e-14 for pose graphs (include/pyslam_hip.h: ps_solve_reduced)
        self.pcg_max_iters = 2000
        # None / 1: this process's GPU.  'auto': landmark-sharded over the torch.distributed process group when one
        # with more than one rank exists (every rank runs the same script on the same Problem: one process per GPU).
        # 'all' or the world size: sharded, an error without a process group.
        self.devices = None
        # True: the caller promises that residual blocks, their measurements, stiffnesses and losses do not change between
        # calls on this Problem; eval_cost / solve / solve_one_iter / compute_covariance then re-read only the parameter
        # VALUES instead of walking every block again (0.29 s at 500 000 blocks).  Default False = the reference's
        # semantics: everything is re-read on every call (pyslam/problem.py:338-360).
        self.static_blocks = False
        # solve() of a one-pose motion-only Problem (the per-frame Problem of the reference's sparse pipeline) runs its
        # whole loop -- iterations, stopping rules, best-parameter bookkeeping -- in one device launch.  False: one call
        # per iteration, as every other problem shape.  Same decisions, same cost history either way.
        self.fused_solve_loop = True


def solve_horizon(opt, iteration, nondecreasing_steps_taken):
    """Iterations the stopping rules of Problem.solve (reference pyslam/problem.py:159-178) still allow AFTER iteration
    number `iteration` (1-based) if its step turns out non-decreasing (cost >= min_cost_decrease * previous cost): none
    without allow_nondecreasing_steps, else what max_nondecreasing_steps and max_iters leave.  The device core does not
    start side work that needs several more iterations to pay back when fewer are to come (ps_set_option "solve_horizon")."""
    if not opt.allow_nondecreasing_steps:
        return 0
    left_nd = opt.max_nondecreasing_steps - (nondecreasing_steps_taken + 1)
    left_it = opt.max_iters + 1 - iteration       # the loop stops once optimization_iters > max_iters
    return max(0, min(left_nd, left_it))


def device_solve(dev, opt, use_core_loop=True, call_ms=None):
    """The loop of Problem.solve (reference pyslam/problem.py:130-178) on a problem resident on the device: start cost,
    then whole iterations (one ps_gn_iteration call each) until the reference's stopping rules fire, with the best
    parameters kept on the device (snapshot / restore = best_params).  -> (cost history, [(pcg iterations, relative
    residual)] per iteration).  Problem.solve() runs exactly this; bench.py times exactly this.

    `dev`: DeviceProblem, ShardedProblemView / ShardedDeviceProblem or PhotometricDevice.  A DeviceProblem runs the loop in
    the core (ps_solve: the same statements in C, no interpreter between two iterations, the start cost's pass enqueued in
    front of the first iteration); tests/test_gpu_solve_loop.py holds the two against each other.  `call_ms`, if a list, receives
    the wall clock of every iteration call in ms."""
    counts = getattr(dev, 'cg_persist_counts', None)
    failures_before = counts()[1] if counts is not None else 0
    try:
        return _device_solve_loop(dev, opt, use_core_loop, call_ms)
    finally:
        if counts is not None:
            try:
                failed = counts()[1] - failures_before
            except Exception:       # noqa: BLE001  (a closed handle: nothing to report)
                failed = 0
            if failed > 0:
                import warnings
                warnings.warn('pyslam_amd: {} reduced solve(s) of this handle ran the one-launch CG into its time-out (its workgroups were '
                              'not resident together: another process on the device, a CU mask below HIP?) and were solved again launch '
                              'by launch; the handle keeps the launch-per-iteration kernels from here on (slower, same results) -- '
                              'ps_problem_info.cg_persist_failures'.format(failed), RuntimeWarning, stacklevel=3)


def _device_solve_loop(dev, opt, use_core_loop, call_ms):
    """device_solve's body (the wrapper reports a one-launch CG that timed out)."""
    loop = getattr(dev, 'solve_loop', None) if use_core_loop else None
    if loop is not None:
        out = loop(opt)
        if out is not None:
            if call_ms is not None:
                call_ms.extend(out[2])
            return out[0], out[1]
    if hasattr(dev, 'reset_solver_state'):
        dev.reset_solver_state()          # a solve is a function of (parameters, options), not of the handle's history
    horizon = getattr(dev, 'set_solve_horizon', None)
    expect = getattr(dev, 'set_expect_next', None)
    linesearch = opt.linesearch_max_iters > 0
    lam = getattr(opt, 'lm_lambda', 0.)
    pcg_tol, pcg_max = (getattr(opt, 'pcg_tol', None) or 0.), getattr(opt, 'pcg_max_iters', 2000)      # (0: the core's default)
    cost = dev.eval_cost(True)
    try:
        return _reference_loop(dev, opt, expect, horizon, cost, lam, pcg_tol, pcg_max, linesearch, call_ms)
    finally:
        if expect is not None:
            expect(False)        # (also when an iteration raises: the handle must not keep "a successor follows" for later callers)


def _reference_loop(dev, opt, expect, horizon, cost, lam, pcg_tol, pcg_max, linesearch, call_ms):
    """The statements of reference pyslam/problem.py:141-178 on device calls."""
    history, stats = [cost], []
    optimization_iters = 0
    nondecreasing_steps_taken = 0
    done_optimization = False
    last_ratio = 1.
    while not done_optimization:
        optimization_iters += 1
        prev_cost = cost
        if horizon is not None:
            horizon(solve_horizon(opt, optimization_iters, nondecreasing_steps_taken))
        if expect is not None:            # the rule ps_solve applies (csrc/ps_abi_solver.h): will another iteration follow?
            expect(optimization_iters <= opt.max_iters and
                   (solve_horizon(opt, optimization_iters, nondecreasing_steps_taken) >= 1 if opt.allow_nondecreasing_steps
                    else (optimization_iters >= 2 and last_ratio < 0.5)))
        # one device call: linearise, solve, update, post-step cost
        t0 = time.perf_counter()
        cost, dx_norm, its, rel = dev.gn_iteration(lam, pcg_tol, pcg_max, linesearch)
        if call_ms is not None:
            call_ms.append((time.perf_counter() - t0) * 1e3)
        stats.append((its, rel))
        history.append(cost)
        last_ratio = cost / prev_cost if prev_cost > 0. else 1.

        done_optimization = optimization_iters > opt.max_iters or \
            dx_norm < opt.min_update_norm or cost < opt.min_cost

        if opt.allow_nondecreasing_steps:
            if nondecreasing_steps_taken == 0:
                dev.snapshot()
            if cost >= opt.min_cost_decrease * prev_cost:
                nondecreasing_steps_taken += 1
            else:
                nondecreasing_steps_taken = 0
            if nondecreasing_steps_taken >= opt.max_nondecreasing_steps:
                done_optimization = True
                dev.restore()
        else:
            done_optimization = done_optimization or cost >= opt.min_cost_decrease * prev_cost
    return history, stats


def solve_tables(lp, options=None, stream=None, device=None):
    """BATCH ENTRY for large problems: the solve of ``Problem.solve()`` from TABLES instead of one Python object per residual
    block (INTEGRATION.md "Batch entry").  ``lp``: a ``pyslam_amd.lowering.LoweredProblem`` -- pose rows (R | t), points,
    observation columns (pose index, point index, (u, v, d), group), camera / stiffness / loss group tables, pose-pose edges and
    priors -- i.e. exactly what ``Problem.solve()`` lowers its registries to (``Problem._lower()`` returns it; ``pyslam_amd.
    synthetic`` builds it directly).  Same options, same loop (device_solve: reference pyslam/problem.py:130-178), same cost
    history bit for bit; what is skipped is the walk over 500 000 block objects (58 of the 68 ms of a C3 solve through the
    object API).  -> (cost history, poses (P, 12 | 6), points (L, 3), [(pcg iterations, relative residual)]).
    ``device``: a DeviceProblem of the same tables kept from an earlier call (its parameters are reset to ``lp``'s)."""
    from pyslam_amd.device import DeviceProblem
    options = options if options is not None else Options()
    dev = device
    if dev is None:
        dev = DeviceProblem(lp, stream=stream)
    else:
        dev.set_params(lp.poses, lp.points)
    try:
        history, stats = device_solve(dev, options)
        poses, points = dev.get_params()
    finally:
        if device is None:
            dev.close()
    return history, poses, points, stats


class Problem:
    def __init__(self, options=Options()):
        self.options = options
        self.param_dict = dict()
        self.residual_blocks = []
        self.block_param_keys = []
        self.block_loss_functions = []
        self.constant_param_keys = []

        self._update_partition_dict = {}
        self._covariance_matrix = None
        self._cost_history = []

        self._device = None
        self._device_sig = None
        self._device_route = None
        self._static_sig = None
        self.solver_stats = []          # per iteration: (pcg iterations, pcg relative residual)

    # ------------------------------------------------------------------
    # registry (reference problem.py:72-108)
    # ------------------------------------------------------------------
    def add_residual_block(self, block, param_keys, loss=L2Loss()):
        if isinstance(param_keys, str):
            param_keys = [param_keys]
        self.residual_blocks.append(block)
        self.block_param_keys.append(param_keys)
        self.block_loss_functions.append(loss)

    def initialize_params(self, param_dict):
        self.param_dict.update(copy.deepcopy(param_dict))

    def set_parameters_constant(self, param_keys):
        if isinstance(param_keys, str):
            param_keys = [param_keys]
        for key in param_keys:
            if key not in self.constant_param_keys:
                self.constant_param_keys.append(key)

    def set_parameters_variable(self, param_keys):
        if isinstance(param_keys, str):
            param_keys = [param_keys]
        for key in param_keys:
            if key in self.constant_param_keys:
                self.constant_param_keys.remove(key)

    # ------------------------------------------------------------------
    # lowering / device handle
    # ------------------------------------------------------------------
    def _lower(self, param_dict=None):
        pd = self.param_dict if param_dict is None else param_dict
        return lowering.lower(pd, self.residual_blocks, self.block_param_keys,
                              self.block_loss_functions, self.constant_param_keys)

    def _photometric_form(self):
        """'se3' / 'split' when the problem is ONE PhotometricResidualSE3 block on variable parameters (the dense VO
        pipeline's per-pyramid-level problem, reference pipelines/dense.py:174-186), else None."""
        if len(self.residual_blocks) != 1 or getattr(self.residual_blocks[0], 'KIND', None) != 'photometric':
            return None
        keys = self.block_param_keys[0]
        if any(k in self.constant_param_keys for k in keys) or set(keys) != set(self.param_dict.keys()):
            return None
        return {1: 'se3', 2: 'split'}.get(len(keys))

    def _get_photometric_device(self, param_dict=None):
        from pyslam_amd.device import PhotometricDevice
        form = self._photometric_form()
        block, loss, keys = self.residual_blocks[0], self.block_loss_functions[0], self.block_param_keys[0]
        sig = ('photometric', form, id(block), id(loss))
        if self._device is None or self._device_sig != sig:
            if self._device is not None:
                self._device.close()
            self._device = PhotometricDevice(block, loss, form == 'split')
            self._device_sig = sig
        pd = self.param_dict if param_dict is None else param_dict
        if form == 'se3':
            T = pd[keys[0]]
            self._device.set_pose(T.rot.as_matrix(), T.trans)
        else:
            self._device.set_pose(pd[keys[0]].as_matrix(), pd[keys[1]])
        return self._device

    def _get_device(self, param_dict=None):
        """DeviceProblem for the current structure; parameters refreshed from param_dict."""
        from pyslam_amd.device import DeviceProblem
        if self._photometric_form():
            return self._get_photometric_device(param_dict)
        dev = self._device
        if getattr(self.options, 'static_blocks', False) and dev is not None and self._device_sig == 'tables' and \
                self._device_route == self._route() and self._static_sig == self._cheap_sig():
            # the caller has declared the blocks unchanged since the tables were built: only the values are re-read
            pd = self.param_dict if param_dict is None else param_dict
            poses, points = lowering.refresh_params(pd, dev.lp)
            dev.set_params(poses, points)
            dev.lp.poses, dev.lp.points = poses, points
            return dev
        lp = self._lower(param_dict)
        # The resident tables are reused only if EVERYTHING but the parameter values is unchanged: measurements,
        # stiffness / loss / camera groups, connectivity, constant masks (the reference re-reads all of it on every
        # iteration, problem.py:338-360, so an edited block.obs, loss.k or a swapped block must take effect).
        dev = self._device
        if dev is not None and self._device_sig == 'tables' and dev.lp.same_tables(lp) and \
                self._device_route == self._route():
            dev.set_params(lp.poses, lp.points)
            dev.lp = lp
        else:
            if dev is not None:
                dev.close()
            self._device = self._make_device(lp)
            self._device_sig = 'tables'
            self._device_route = self._route()
        self._static_sig = self._cheap_sig()
        return self._device

    def _cheap_sig(self):
        """What Options.static_blocks checks instead of walking the blocks: counts and the constant-key list."""
        return (len(self.residual_blocks), len(self.block_param_keys), len(self.block_loss_functions),
                len(self.param_dict), tuple(self.constant_param_keys))

    def _route(self):
        """'sharded' when Options.devices asks for the landmark-sharded multi-GPU driver and a
        torch.distributed process group exists (one process per GPU), else 'single'."""
        want = getattr(self.options, 'devices', None)
        if want in (None, 1, 'single'):
            return 'single'
        try:
            import torch.distributed as dist
        except ImportError:
            dist = None
        up = dist is not None and dist.is_available() and dist.is_initialized()
        if want == 'auto':
            return 'sharded' if up and dist.get_world_size() > 1 else 'single'
        if not up:
            raise RuntimeError("Options.devices = {!r} needs an initialised torch.distributed process group "
                               "(one process per GPU, backend 'nccl')".format(want))
        if isinstance(want, int) and want != dist.get_world_size():
            raise RuntimeError("Options.devices = {} but the process group has {} ranks".format(want, dist.get_world_size()))
        return 'sharded'

    def _make_device(self, lp):
        """Upload the tables: one GPU, or this rank's landmark shard of them behind the same interface."""
        from pyslam_amd.device import DeviceProblem
        if self._route() == 'single':
            return DeviceProblem(lp)
        import torch.distributed as dist
        from pyslam_amd.distributed import ShardedProblemView
        return ShardedProblemView(lp, dist)

    def _write_back(self, dev, poses=None):
        """Copy the device parameter tables into the live param_dict objects (`poses`: rows the caller already holds, for
        a problem without named landmarks)."""
        form = self._photometric_form()
        if form:
            R, t = dev.get_pose()
            keys = self.block_param_keys[0]
            if form == 'se3':
                self.param_dict[keys[0]].rot.mat = R
                self.param_dict[keys[0]].trans = t
            else:
                self.param_dict[keys[0]].mat = R
                val = self.param_dict[keys[1]]
                if isinstance(val, np.ndarray):
                    val[...] = t
                else:
                    self.param_dict[keys[1]] = t
            return
        lp = dev.lp
        if poses is None or len(lp.point_keys):
            poses, points = dev.get_params()
        else:
            points = ()
        for key, row in zip(lp.pose_keys, poses):
            R, t = lowering.unpack_pose(row, lp.dof)
            T = self.param_dict[key]
            T.rot.mat = R
            T.trans = t
        fast = lowering._fast_walk()
        if fast is not None and len(lp.point_keys) and type(self.param_dict) is dict and isinstance(lp.point_keys, list):
            # landmarks that are plain arrays of three doubles are written in place in C (pyslam_amd/cext/lower_fast.c: scatter3)
            todo = fast.scatter3(self.param_dict, lp.point_keys, np.ascontiguousarray(points[:len(lp.point_keys)], dtype=np.float64))
            todo = [(lp.point_keys[i], points[i]) for i in todo]
        else:
            todo = zip(lp.point_keys, points)
        for key, p in todo:
            val = self.param_dict[key]
            if isinstance(val, np.ndarray):
                val[...] = p
            else:
                self.param_dict[key] = p.copy()

    def _dx_in_reference_order(self, dev):
        xp, xl = dev.get_dx()
        lp = dev.lp
        n = max([r.stop for r in self._update_partition_dict.values()] + [0])
        dx = np.zeros(n)
        for key, rid in zip(lp.pose_keys, lp.pose_rid):
            if rid >= 0:
                dx[self._update_partition_dict[key]] = xp[rid]
        for key, vid in zip(lp.point_keys, lp.point_vid):
            if vid >= 0:
                dx[self._update_partition_dict[key]] = xl[vid]
        return dx

    # ------------------------------------------------------------------
    # cost (reference problem.py:110-128)
    # ------------------------------------------------------------------
    def eval_cost(self, param_dict=None):
        try:
            dev = self._get_device(param_dict)
        except NotLowerable:
            return self._eval_cost_host(param_dict)
        return dev.eval_cost(include_all_constant=True)

    def _eval_cost_host(self, param_dict=None):
        pd = self.param_dict if param_dict is None else param_dict
        cost = 0.
        for block, keys, loss in zip(self.residual_blocks, self.block_param_keys,
                                     self.block_loss_functions):
            try:
                params = [pd[key] for key in keys]
            except KeyError as e:
                print("Parameter {} has not been initialized".format(e.args[0]))
            cost += np.sum(loss.loss(block.evaluate(params)))
        return cost

    # ------------------------------------------------------------------
    # solve (reference problem.py:130-180)
    # ------------------------------------------------------------------
    def solve(self):
        # (reference problem.py:132 rebuilds the partition here; the device path below never reads it -- it is rebuilt on first
        #  use instead: one Python statement per parameter, 0.15 s for the 500 000 landmarks of C4)
        self._partition = None
        try:
            dev = self._get_device()
        except NotLowerable:
            dev = None
        opt = self.options
        self.solver_stats = []

        # a single pose against constant landmarks (the per-frame Problem of pipelines/sparse.py): this whole loop runs on
        # the device in one launch; identical decisions, identical cost history
        fused = getattr(dev, 'motion_only_solve', None) if dev is not None and opt.fused_solve_loop else None
        if fused is not None:
            out = fused(opt, opt.linesearch_max_iters > 0)
            if out is not None:
                self._cost_history, iters, _, pose = out
                self.solver_stats = [(0, 0.0)] * iters
                self._write_back(dev, poses=pose.reshape(1, 12))      # (the call returned the pose: no second copy)
                return self.param_dict

        if dev is not None:
            self._cost_history, self.solver_stats = device_solve(dev, opt)
            self._write_back(dev)
            return self.param_dict

        cost = self._eval_cost_host()
        dx_norm = 100.
        optimization_iters = 0
        nondecreasing_steps_taken = 0
        self._cost_history = [cost]
        best_params = None
        done_optimization = False

        while not done_optimization:
            optimization_iters += 1
            prev_cost = cost

            dx, cost = self.solve_one_iter()
            dx_norm = np.linalg.norm(dx)
            for k, r in self._update_partition_dict.items():
                self._perturb_by_key(k, dx[r])
            self._cost_history.append(cost)

            done_optimization = optimization_iters > opt.max_iters or \
                dx_norm < opt.min_update_norm or cost < opt.min_cost

            if opt.allow_nondecreasing_steps:
                if nondecreasing_steps_taken == 0:
                    best_params = copy.deepcopy(self.param_dict)
                if cost >= opt.min_cost_decrease * prev_cost:
                    nondecreasing_steps_taken += 1
                else:
                    nondecreasing_steps_taken = 0
                if nondecreasing_steps_taken >= opt.max_nondecreasing_steps:
                    done_optimization = True
                    self.param_dict.update(best_params)
            else:
                done_optimization = done_optimization or cost >= opt.min_cost_decrease * prev_cost

        return self.param_dict

    def solve_one_iter(self):
        """One Gauss-Newton step: (step * dx, cost); parameters are NOT updated
        (reference problem.py:182-194)."""
        # the reference orders dx by the CURRENT non-constant parameters (block_cidx_dict is rebuilt on every call,
        # problem.py:294-303): recompute, a partition left by an earlier solve() may be stale after freeze / release
        self._update_partition_dict = self._get_update_partition_dict()
        opt = self.options
        try:
            dev = self._get_device()
        except NotLowerable:
            return self._solve_one_iter_host()
        if self._photometric_form():
            saved = dev.get_pose()
            dxd, cost = dev.step(opt.linesearch_max_iters > 0)
            dev.set_pose(*saved)
            dx = np.zeros(6)
            keys = self.block_param_keys[0]
            if len(keys) == 1:
                dx[self._update_partition_dict[keys[0]]] = dxd
            else:
                dx[self._update_partition_dict[keys[0]]] = dxd[3:6]
                dx[self._update_partition_dict[keys[1]]] = dxd[0:3]
            self.solver_stats.append((0, 0.0))
            return dx, cost
        saved = dev.get_params()
        dev.linearize(opt.lm_lambda)
        its, rel = dev.solve_reduced(opt.pcg_tol or 0., opt.pcg_max_iters)
        dev.backsub()
        dx = self._dx_in_reference_order(dev)
        if opt.linesearch_max_iters > 0:
            dev.apply_update(1.0)                     # the reference's search always ends at step 1
            cost = dev.eval_cost(True)
            dev.set_params(*saved)
        else:
            cost = dev.eval_cost(False)
        self.solver_stats.append((its, rel))
        return dx, cost

    # ---- generic (host-evaluated) path ---------------------------------
    # Unknowns up to which the host-evaluated path forms J dense and solves by Cholesky on the device
    # (ps_dense_normal_solve); up to 8 192 J travels as CSR and the device forms J^T J dense and factors it with the blocked
    # multi-workgroup Cholesky (ps_sparse_normal_direct, round 5); beyond that it runs CG on the normal equations
    # (ps_sparse_normal_solve) -- the reference's sparse LU has no such limits (pyslam/problem.py:186).
    DENSE_GENERIC_LIMIT = 2048

    def _host_jacobian(self):
        """IRLS-scaled J~ (scipy CSR, block-sparse: one dense (rows x dof) piece per block and parameter), e~ and cost
        by walking the blocks (reference problem.py:338-360); used only for blocks without a device kernel."""
        import scipy.sparse as sp
        part = self._update_partition_dict
        n = max([r.stop for r in part.values()] + [0])
        rows_i, cols_i, vals, es, cost, m = [], [], [], [], 0., 0
        for block, keys, loss in zip(self.residual_blocks, self.block_param_keys,
                                     self.block_loss_functions):
            params = [self.param_dict[key] for key in keys]
            want = [key not in self.constant_param_keys for key in keys]
            if not any(want):
                continue
            residual, jacobians = block.evaluate(params, want)
            residual = np.atleast_1d(residual).reshape(-1)
            s = np.sqrt(np.asarray(loss.weight(residual), dtype=float)).reshape(-1)
            nres = residual.size
            for key, jac in zip(keys, jacobians):
                if jac is not None:
                    jac = s[:, None] * np.asarray(jac, dtype=float).reshape(nres, -1)
                    c0 = part[key].start
                    rows_i.append(np.repeat(np.arange(m, m + nres), jac.shape[1]))
                    cols_i.append(np.tile(np.arange(c0, c0 + jac.shape[1]), nres))
                    vals.append(jac.ravel())
            es.append(s * residual)
            cost += np.sum(loss.loss(residual))
            m += nres
        J = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows_i), np.concatenate(cols_i))), shape=(m, n))
        J.sum_duplicates()                       # a parameter listed twice in one block: its Jacobian pieces add up
        return J, np.concatenate(es), cost

    def _generic_step(self, J, e):
        from pyslam_amd.device import dense_normal_solve, sparse_normal_solve, sparse_normal_direct, DIRECT_GENERIC_LIMIT
        if J.shape[1] <= self.DENSE_GENERIC_LIMIT:
            return dense_normal_solve(J.toarray(), e)
        if J.shape[1] <= DIRECT_GENERIC_LIMIT:
            # a direct solve, as the reference's (pyslam/problem.py:186): dense blocked Cholesky on the device + refinement.
            # Where it cannot answer -- a rank-deficient but consistent J^T J (gauge freedom, an unobserved parameter: "not positive
            # definite" / NotConverged), or 2 n^2 doubles that the device cannot allocate -- the CG on the normal equations below,
            # which converges on consistent singular systems, takes over (still on the device; round-5 ADVICE)
            from pyslam_amd.device import NotConverged
            from pyslam_amd._native import NativeError
            try:
                dx, its, rel = sparse_normal_direct(J, r=e)
                self.solver_stats.append((its, rel))
                return dx
            except (NotConverged, NativeError):
                pass
        dx, its, rel = sparse_normal_solve(J, r=e, tol=self.options.pcg_tol or 1e-12, max_iters=max(self.options.pcg_max_iters, 10 * J.shape[1]))
        self.solver_stats.append((its, rel))
        return dx

    def _solve_one_iter_host(self):
        J, e, cost = self._host_jacobian()
        dx = self._generic_step(J, e)
        if self.options.linesearch_max_iters > 0:
            test = copy.deepcopy(self.param_dict)
            for k, r in self._update_partition_dict.items():
                self._perturb_by_key(k, dx[r], test)
            cost = self._eval_cost_host(test)
        return dx, cost

    # ------------------------------------------------------------------
    # covariance (reference problem.py:196-216)
    # ------------------------------------------------------------------
    #
    # Typed problems: column k of the covariance is the solution of (J^T W J) x = e_k, computed on the
    # device by the iteration's own Schur elimination + CG + back-substitution (one call per scalar
    # unknown of the requested parameter), so blocks are available at any problem size.  The full
    # dense ``_covariance_matrix`` the reference exposes is materialised only while it is small
    # (<= DENSE_COVARIANCE_LIMIT unknowns); beyond that ``get_covariance_block`` computes columns on demand.
    DENSE_COVARIANCE_LIMIT = 1500

    def compute_covariance(self):
        try:
            self._update_partition_dict = self._get_update_partition_dict()      # (never stale: see solve_one_iter)
            self._covariance_matrix = None
            self._cov_sparse_J = None
            self._cov_device = None
            self._cov_columns = {}
            try:
                dev = self._get_device()
            except NotLowerable:
                dev = None
            if dev is None:
                from pyslam_amd.device import dense_normal_solve
                J, e, _ = self._host_jacobian()
                if J.shape[1] <= self.DENSE_GENERIC_LIMIT:
                    _, self._covariance_matrix = dense_normal_solve(J.toarray(), e, want_covariance=True)
                else:
                    self._cov_sparse_J = J           # columns on demand (get_covariance_block): CG with a unit right-hand side
                return
            if self._photometric_form():
                # 6 x 6: the device forms J~^T J~ over all pixels; its inverse in the reference's unknown order
                Hm = dev.normal_equations()[0]
                keys = self.block_param_keys[0]
                order = np.empty(6, dtype=int)
                if len(keys) == 1:
                    order[self._update_partition_dict[keys[0]]] = np.arange(6)
                else:
                    order[self._update_partition_dict[keys[0]]] = np.arange(3, 6)
                    order[self._update_partition_dict[keys[1]]] = np.arange(0, 3)
                self._covariance_matrix = np.linalg.inv(Hm[np.ix_(order, order)])
                return
            dev.covariance_begin()
            self._cov_device = dev
            # scatter maps device order -> reference ordering (vectorised: C3 has 50 000 landmarks)
            lp, part = dev.lp, self._update_partition_dict
            self._cov_where = {}
            sel = [(k, int(rid)) for k, rid in zip(lp.pose_keys, lp.pose_rid) if rid >= 0]
            for k, rid in sel:
                self._cov_where[k] = (0, rid, lp.dof)
            self._cov_pose_src = np.array([rid for _, rid in sel], dtype=np.int64)
            self._cov_pose_dst = (np.array([part[k].start for k, _ in sel], dtype=np.int64)[:, None]
                                  + np.arange(lp.dof)[None, :]).reshape(-1)
            sel = [(k, int(vid)) for k, vid in zip(lp.point_keys, lp.point_vid) if vid >= 0]
            for k, vid in sel:
                self._cov_where[k] = (1, vid, 3)
            self._cov_point_src = np.array([vid for _, vid in sel], dtype=np.int64)
            self._cov_point_dst = (np.array([part[k].start for k, _ in sel], dtype=np.int64)[:, None]
                                   + np.arange(3)[None, :]).reshape(-1)
            n = max([r.stop for r in part.values()] + [0])
            if n <= self.DENSE_COVARIANCE_LIMIT:
                cov = np.zeros((n, n))
                for key, r in part.items():
                    cov[:, r] = self._covariance_columns(key)
                self._covariance_matrix = 0.5 * (cov + cov.T)
        except Exception as e:
            print('Covariance computation failed!\n{}'.format(e))

    def _covariance_columns(self, key):
        """(n, dof_key) column block of the covariance for parameter `key`, reference ordering."""
        if key in self._cov_columns:
            return self._cov_columns[key]
        kind, index, width = self._cov_where[key]          # KeyError: constant / unknown parameter
        n = max([r.stop for r in self._update_partition_dict.values()] + [0])
        cols = np.zeros((n, width))
        for c in range(width):
            xp, xl = self._cov_device.covariance_column(kind, index, c)
            if self._cov_pose_src.size:
                cols[self._cov_pose_dst, c] = xp[self._cov_pose_src].reshape(-1)
            if self._cov_point_src.size:
                cols[self._cov_point_dst, c] = xl[self._cov_point_src].reshape(-1)
        if len(self._cov_columns) < 64:
            self._cov_columns[key] = cols
        return cols

    def get_covariance_block(self, param0, param1):
        try:
            r0 = self._update_partition_dict[param0]
            r1 = self._update_partition_dict[param1]
            if self._covariance_matrix is None and getattr(self, '_cov_sparse_J', None) is not None:
                from pyslam_amd.device import sparse_normal_solve
                J = self._cov_sparse_J
                cols = np.zeros((r0.stop - r0.start, r1.stop - r1.start))
                for c, k in enumerate(range(r1.start, r1.stop)):
                    rhs = np.zeros(J.shape[1]); rhs[k] = 1.
                    x, _, _ = sparse_normal_solve(J, rhs=rhs, tol=1e-13, max_iters=20 * J.shape[1], accept=1e-8)
                    cols[:, c] = x[r0.start:r0.stop]
                return np.squeeze(cols)
            if self._covariance_matrix is None and getattr(self, '_cov_device', None) is not None:
                return np.squeeze(self._covariance_columns(param1)[r0.start:r0.stop, :])
            return np.squeeze(self._covariance_matrix[r0.start:r0.stop, r1.start:r1.stop])
        except KeyError as e:
            print('Cannot compute covariance for constant parameter {}'.format(e.args[0]))
        return None

    # ------------------------------------------------------------------
    # reporting (reference problem.py:218-250; text format kept byte-identical)
    # ------------------------------------------------------------------
    def summary(self, format='brief'):
        if not self._cost_history:
            raise ValueError('solve has not yet been called')
        if format == 'brief':
            return 'Iterations: {:3} | Cost: {:12e} --> {:12e}'.format(
                len(self._cost_history), self._cost_history[0], self._cost_history[-1])
        if format == 'full':
            header = '{:>5s} | {:>12s} --> {:>12s} | {:>10s}\n'.format(
                'Iter', 'Initial cost', 'Final cost', 'Rel change')
            lines = [header, '-' * len(header) + '\n']
            for i, (ic, fc) in enumerate(zip(self._cost_history[:-1], self._cost_history[1:])):
                lines.append('{:5} | {:12e} --> {:12e} | {:+10f}\n'.format(i, ic, fc, (fc - ic) / ic))
            return ''.join(lines)
        raise ValueError('Invalid summary format \'{}\'.'.format(format) +
                         'Valid formats are \'brief\' and \'full\'')

    # ------------------------------------------------------------------
    # helpers (reference problem.py:252-277, 400-409)
    # ------------------------------------------------------------------
    @property
    def _update_partition_dict(self):
        """Parameter key -> range in dx (reference problem.py:60, 132, 252-262): the reference's attribute, built when first read
        after solve() invalidated it."""
        if self._partition is None:
            self._partition = self._get_update_partition_dict()
        return self._partition

    @_update_partition_dict.setter
    def _update_partition_dict(self, value):
        self._partition = value

    def _get_update_partition_dict(self):
        out, offset = {}, 0
        for key, param in self.param_dict.items():
            if key in self.constant_param_keys:
                continue
            if hasattr(param, 'dof'):
                dof = param.dof
            elif hasattr(param, '__len__'):
                dof = len(param)
            else:
                dof = 1
            out[key] = range(offset, offset + dof)
            offset += dof
        return out

    def _perturb_by_key(self, key, dx, param_dict=None):
        pd = self.param_dict if param_dict is None else param_dict
        try:
            pd[key].perturb(dx)
        except AttributeError:
            pd[key] += dx
