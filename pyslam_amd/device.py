"""DeviceProblem: a LoweredProblem resident in HBM behind the C ABI.

Thin, stateless-on-the-host wrapper: every method maps to one ``ps_*`` entry
point of include/pyslam_hip.h.  ``Problem.solve`` (pyslam_amd/problem.py) and
bench.py drive it; multi-GPU sharding lives in pyslam_amd/distributed.py.
"""
import ctypes as C

import numpy as np

from pyslam_amd import _native as nat


class DeviceProblem:
    def __init__(self, lp, stream=None, extra_pairs=None):
        lib = nat.require_gpu()
        self._lib = lib
        self.lp = lp
        self.dof = lp.dof
        d = nat.ProblemDesc()
        keep = []       # host arrays must outlive the create call only
        # tables a torch caller already holds in HBM (`resident_tables(lp)` below, or its own tensors) are handed over by
        # address: ps_problem_desc.flags.  All-or-nothing per class, as the C ABI has it.
        params_res = nat.is_resident(lp.poses)
        tables_res = nat.is_resident(lp.pose_rid)
        if params_res != nat.is_resident(lp.points):
            raise TypeError('poses and points must both be host arrays or both device tensors')
        d.flags = (nat.PS_DESC_DEVICE_PARAMS if params_res else 0) | (nat.PS_DESC_DEVICE_TABLES if tables_res else 0)

        def conv(a, np_dtype, want_res, to_ptr):
            if nat.is_resident(a) != want_res:
                raise TypeError('mixed host / device tables (ps_problem_desc.flags is per class: parameters, everything else)')
            if not want_res:
                a = np.ascontiguousarray(a, dtype=np_dtype)
            keep.append(a)
            return to_ptr(a) if a.shape[0] else None

        def F(a, res=None):
            return conv(a, np.float64, tables_res if res is None else res, nat.f64p)

        def I(a):
            return conv(a, np.int32, tables_res, nat.i32p)

        d.dof = lp.dof
        d.num_poses, d.poses, d.pose_rid = lp.num_poses, F(lp.poses, params_res), I(lp.pose_rid)
        d.num_points, d.points, d.point_vid = lp.num_points, F(lp.points, params_res), I(lp.point_vid)
        d.num_obs = lp.num_obs
        d.obs_pose, d.obs_point, d.obs_uvd, d.obs_grp = I(lp.obs_pose), I(lp.obs_point), F(lp.obs_uvd), I(lp.obs_grp)
        d.num_cams, d.cams = lp.cams.shape[0], F(lp.cams)
        d.num_stiff3, d.stiff3 = lp.stiff3.shape[0], F(lp.stiff3)
        d.num_obs_groups, d.obs_groups = lp.obs_groups.shape[0], F(lp.obs_groups)
        d.num_edges = lp.num_edges
        d.e_i, d.e_j, d.e_Tobs_inv, d.e_grp = I(lp.e_i), I(lp.e_j), F(lp.e_Tobs_inv), I(lp.e_grp)
        d.num_priors = lp.num_priors
        d.u_i, d.u_Tobs_inv, d.u_grp = I(lp.u_i), F(lp.u_Tobs_inv), I(lp.u_grp)
        d.num_stiffd, d.stiffd = lp.stiffd.shape[0], F(lp.stiffd)
        d.num_edge_groups, d.edge_groups = lp.edge_groups.shape[0], F(lp.edge_groups)
        if extra_pairs is not None and len(extra_pairs[0]):
            d.num_extra_pairs = len(extra_pairs[0])
            d.extra_pair_i, d.extra_pair_j = I(extra_pairs[0]), I(extra_pairs[1])
        self._h = nat.H()
        nat.check(lib.ps_problem_create(C.byref(d), C.c_void_p(stream or 0), C.byref(self._h)))
        info = nat.ProblemInfo()
        nat.check(lib.ps_get_info(self._h, C.byref(info)))
        self.info = {k: getattr(info, k) for k, _ in nat.ProblemInfo._fields_}
        self.nr, self.nv = info.num_reduced, info.num_var_points

    # ---- lifetime ------------------------------------------------------
    def close(self):
        if getattr(self, '_h', None):
            self._lib.ps_problem_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- hot path ------------------------------------------------------
    def eval_cost(self, include_all_constant=True):
        c = C.c_double()
        nat.check(self._lib.ps_eval_cost(self._h, int(include_all_constant), C.byref(c)))
        return c.value

    def gn_iteration(self, lm_lambda=0., pcg_tol=1e-12, pcg_max_iters=1000, linesearch=True):
        """-> (cost, ||dx||, pcg iterations, pcg relative residual); parameters are updated."""
        cost, nrm, rel, it = C.c_double(), C.c_double(), C.c_double(), C.c_int()
        nat.check(self._lib.ps_gn_iteration(self._h, lm_lambda, pcg_tol or 0., pcg_max_iters, int(linesearch),
                                            C.byref(cost), C.byref(nrm), C.byref(it), C.byref(rel)))
        return cost.value, nrm.value, it.value, rel.value

    def motion_only_solve(self, opt, linesearch):
        """Problem.solve's whole loop in one launch (ps_motion_only_solve) for a one-pose motion-only problem.
        -> (cost history, iterations, last ||dx||, final pose row) or None when the problem is not of that kind (iterate
        instead)."""
        o = nat.SolveOptions()
        o.max_iters, o.allow_nondecreasing_steps = int(opt.max_iters), int(bool(opt.allow_nondecreasing_steps))
        o.max_nondecreasing_steps, o.linesearch = int(opt.max_nondecreasing_steps), int(bool(linesearch))
        o.min_update_norm, o.min_cost = float(opt.min_update_norm), float(opt.min_cost)
        o.min_cost_decrease, o.lm_lambda = float(opt.min_cost_decrease), float(getattr(opt, 'lm_lambda', 0.))
        cap = o.max_iters + 2
        if cap < 2 or cap > 238:
            return None
        hist, pose = np.zeros(cap), np.zeros(12)
        n, its, dxn = C.c_int32(), C.c_int32(), C.c_double()
        rc = self._lib.ps_motion_only_solve(self._h, C.byref(o), nat.f64p(hist), cap, C.byref(n), C.byref(its), C.byref(dxn),
                                            nat.f64p(pose))
        if rc == 1:
            return None
        nat.check(rc)
        return hist[:n.value].tolist(), its.value, dxn.value, pose

    def solve_loop(self, opt):
        """Problem.solve's loop in one C call (ps_solve).  -> (cost history, [(pcg iterations, relative residual)], [ms per
        iteration call]) or None when the core does not offer it for this handle (the caller loops itself)."""
        o = nat.SolveOptions()
        o.max_iters, o.allow_nondecreasing_steps = int(opt.max_iters), int(bool(opt.allow_nondecreasing_steps))
        o.max_nondecreasing_steps, o.linesearch = int(opt.max_nondecreasing_steps), int(opt.linesearch_max_iters > 0)
        o.min_update_norm, o.min_cost = float(opt.min_update_norm), float(opt.min_cost)
        o.min_cost_decrease, o.lm_lambda = float(opt.min_cost_decrease), float(getattr(opt, 'lm_lambda', 0.))
        cap = o.max_iters + 2
        if cap < 2 or cap > 100000:
            return None
        buf = getattr(self, '_solve_buf', None)              # (result buffers kept on the object: four allocations less per solve)
        if buf is None or buf[0].size < cap:
            buf = self._solve_buf = (np.zeros(cap), np.zeros(cap), np.zeros(cap), np.zeros(cap, dtype=np.int32))
            self._solve_ptr = (nat.f64p(buf[0]), nat.f64p(buf[1]), nat.f64p(buf[2]), nat.i32p(buf[3]))
        hist, rel, ms, its = buf
        p_hist, p_rel, p_ms, p_its = self._solve_ptr
        n, iters, dxn = C.c_int32(), C.c_int32(), C.c_double()
        rc = self._lib.ps_solve(self._h, C.byref(o), float(getattr(opt, 'pcg_tol', None) or 0.), int(getattr(opt, 'pcg_max_iters', 2000)),
                                p_hist, cap, C.byref(n), C.byref(iters), C.byref(dxn), p_its, p_rel, p_ms)
        if rc == 1:
            return None
        nat.check(rc)
        k = iters.value
        return hist[:n.value].tolist(), [(int(a), float(b)) for a, b in zip(its[:k], rel[:k])], ms[:k].tolist()

    def gn_finish(self, linesearch=True):
        """-> (shard cost, ||dx_pose||^2, ||dx_point||^2); parameters are updated."""
        c, a, b = C.c_double(), C.c_double(), C.c_double()
        nat.check(self._lib.ps_gn_finish(self._h, int(linesearch), C.byref(c), C.byref(a), C.byref(b)))
        return c.value, a.value, b.value

    def gn_solve_finish(self, pcg_tol=1e-12, pcg_max_iters=1000, linesearch=True):
        """Reduced solve + back-substitution + update + cost with one sync.
        -> (shard cost, ||dx_pose||^2, ||dx_point||^2, pcg iterations, relative residual)."""
        c, a, b, rel, it = C.c_double(), C.c_double(), C.c_double(), C.c_double(), C.c_int()
        nat.check(self._lib.ps_gn_solve_finish(self._h, pcg_tol or 0., pcg_max_iters, int(linesearch), C.byref(c),
                                               C.byref(a), C.byref(b), C.byref(it), C.byref(rel)))
        return c.value, a.value, b.value, it.value, rel.value

    def set_collective(self, allreduce_fn_ptr, comm_ptr):
        """Native RCCL: ps_gn_iteration then performs the sharded iteration incl. both all-reduces."""
        nat.check(self._lib.ps_set_collective(self._h, C.c_void_p(allreduce_fn_ptr), C.c_void_p(comm_ptr)))

    def set_segment_exchange(self, allgather_fn_ptr, world, rank, maxlen, mine, dst, src_ptr, src_off):
        """The core's own segment exchange (include/pyslam_hip.h: ps_set_segment_exchange; plan: distributed.segment_plan)."""
        a = [np.ascontiguousarray(x, dtype=np.int64) for x in (mine, dst, src_ptr, src_off)]
        p64 = lambda x: x.ctypes.data_as(C.POINTER(C.c_int64))
        nat.check(self._lib.ps_set_segment_exchange(self._h, C.c_void_p(allgather_fn_ptr), int(world), int(rank), int(maxlen),
                                                    int(a[0].size), p64(a[0]), int(a[1].size), p64(a[1]), p64(a[2]), p64(a[3])))

    def shard_buffer(self):
        ptr = C.c_void_p()
        nat.check(self._lib.ps_shard_buffer(self._h, C.byref(ptr)))
        return ptr.value

    def gn_solve_finish_enqueue(self, pcg_tol, pcg_max_iters, linesearch, first):
        """No host sync.  Returns True on the final (ungated) pass."""
        rc = self._lib.ps_gn_solve_finish_enqueue(self._h, pcg_tol or 0., pcg_max_iters, int(linesearch), int(first))
        if rc < 0:
            nat.check(rc)
        return rc == 1

    def gn_result(self):
        """Sync. -> (done, shard cost sum, shard ||dx_point||^2 sum, ||dx_pose||^2, iterations, relres)."""
        done, it, rel, dxp2 = C.c_int(), C.c_int(), C.c_double(), C.c_double()
        sb = (C.c_double * 2)()
        nat.check(self._lib.ps_gn_result(self._h, C.byref(done), sb, C.byref(dxp2), C.byref(it), C.byref(rel)))
        return done.value != 0, sb[0], sb[1], dxp2.value, it.value, rel.value

    def linearize(self, lm_lambda=0.):
        nat.check(self._lib.ps_linearize(self._h, lm_lambda))

    def solve_reduced(self, tol=1e-12, max_iters=1000):
        rel, it = C.c_double(), C.c_int()
        nat.check(self._lib.ps_solve_reduced(self._h, tol or 0., max_iters, C.byref(it), C.byref(rel)))
        return it.value, rel.value

    def backsub(self):
        nat.check(self._lib.ps_backsub(self._h))

    def apply_update(self, step=1.0):
        nat.check(self._lib.ps_apply_update(self._h, step))

    def step_norm(self):
        n2 = C.c_double()
        nat.check(self._lib.ps_step_norm2(self._h, C.byref(n2)))
        return float(np.sqrt(n2.value))

    def snapshot(self):
        nat.check(self._lib.ps_snapshot_params(self._h))

    def restore(self):
        nat.check(self._lib.ps_restore_params(self._h))

    def reset_solver_state(self):
        """A new solve starts here: nothing the solver carried over from earlier calls is used (ps_reset_solver_state)."""
        nat.check(self._lib.ps_reset_solver_state(self._h))

    def set_solve_horizon(self, n):
        """Further iterations the caller's stopping rule allows if the coming step is non-decreasing (-1: unknown)."""
        nat.check(self._lib.ps_set_option(self._h, b'solve_horizon', float(n)))

    def set_expect_next(self, flag):
        """The caller's loop will call gn_iteration again after the coming call unless a stopping rule on ||dx|| or the cost
        fires (what ps_solve tells the core about its own loop): the coming call's tail then runs the NEXT linearisation's
        landmark pass in place of its cost pass -- one evaluation of every observation instead of two."""
        nat.check(self._lib.ps_set_option(self._h, b'expect_next', 1.0 if flag else 0.0))

    # ---- covariance by columns (reference problem.py:196-216) ------------
    def covariance_begin(self):
        """Linearise at the current parameters and prepare the reduced solver."""
        nat.check(self._lib.ps_covariance_begin(self._h))

    def covariance_column(self, kind, index, comp, tol=1e-13, max_iters=4000):
        """Column of the covariance (J^T W J)^-1 for component `comp` of reduced pose `index`
        (kind 0) or variable landmark `index` (kind 1): (pose part (nr, dof), point part (nv, 3))."""
        rel, it = C.c_double(), C.c_int()
        nat.check(self._lib.ps_covariance_column(self._h, int(kind), int(index), int(comp), tol, max_iters,
                                                 C.byref(it), C.byref(rel)))
        return self.get_dx()

    # ---- data movement -------------------------------------------------
    def get_dx(self):
        """(dx_pose (nr, dof), dx_point (nv, 3)) in device order."""
        xp = np.zeros((self.nr, self.dof))
        xl = np.zeros((self.nv, 3))
        nat.check(self._lib.ps_get_dx(self._h, nat.f64p(xp), nat.f64p(xl)))
        return xp, xl

    def get_params(self, poses_out=None, points_out=None):
        """Without arguments: fresh host arrays.  With torch tensors resident in HBM: filled in place, device to device."""
        if poses_out is not None or points_out is not None:
            nat.check(self._lib.ps_get_params(self._h, nat.f64p(poses_out), nat.f64p(points_out)))
            return poses_out, points_out
        poses = np.zeros((self.lp.num_poses, self.lp.pose_width))
        points = np.zeros((self.lp.num_points, 3))
        nat.check(self._lib.ps_get_params(self._h, nat.f64p(poses), nat.f64p(points)))
        return poses, points

    def set_params(self, poses=None, points=None):
        """Host arrays, or float64 torch tensors resident in HBM (copied device to device)."""
        p = poses if poses is None or nat.is_resident(poses) else np.ascontiguousarray(poses, dtype=np.float64)
        q = points if points is None or nat.is_resident(points) else np.ascontiguousarray(points, dtype=np.float64)
        nat.check(self._lib.ps_set_params(self._h, nat.f64p(p), nat.f64p(q)))

    def reduce_buffer(self):
        """(device pointer, number of doubles) of the exchange buffer [upper(S) | g | cost | flag] the
        multi-GPU driver all-reduces between shard_pack() and shard_unpack()."""
        ptr, n = C.c_void_p(), C.c_int64()
        nat.check(self._lib.ps_reduce_buffer(self._h, C.byref(ptr), C.byref(n)))
        return ptr.value, n.value

    def shard_pack(self):
        nat.check(self._lib.ps_shard_pack(self._h))

    def shard_unpack(self):
        nat.check(self._lib.ps_shard_unpack(self._h))

    # ---- parity taps ---------------------------------------------------
    def reduced_system(self):
        """BSR (row_ptr, col_idx, vals (nnzb, d, d), g (nr*d,))."""
        nnzb, d = self.info['reduced_nnzb'], self.dof
        rp = np.zeros(self.nr + 1, dtype=np.int32)
        ci = np.zeros(max(nnzb, 1), dtype=np.int32)
        vals = np.zeros((max(nnzb, 1), d, d))
        g = np.zeros(max(self.nr * d, 1))
        nat.check(self._lib.ps_get_reduced_system(self._h, nat.i32p(rp), nat.i32p(ci), nat.f64p(vals), nat.f64p(g)))
        return rp, ci[:nnzb], vals[:nnzb], g[:self.nr * d]

    def reduced_dense(self):
        rp, ci, vals, g = self.reduced_system()
        d, n = self.dof, self.nr * self.dof
        S = np.zeros((n, n))
        for r in range(self.nr):
            for k in range(rp[r], rp[r + 1]):
                S[r * d:(r + 1) * d, ci[k] * d:(ci[k] + 1) * d] = vals[k]
        return S, g

    def landmark_factors(self):
        cinv, c = np.zeros((self.nv, 6)), np.zeros((self.nv, 3))
        nat.check(self._lib.ps_get_landmark_factors(self._h, nat.f64p(cinv), nat.f64p(c)))
        return cinv, c

    def debug_reproj_blocks(self):
        n = self.lp.num_obs
        r, jp, jl = np.zeros((n, 3)), np.zeros((n, 3, 6)), np.zeros((n, 3, 3))
        nat.check(self._lib.ps_debug_reproj_blocks(self._h, nat.f64p(r), nat.f64p(jp), nat.f64p(jl)))
        return r, jp, jl

    def debug_factor_blocks(self):
        """(r~, J~_1, J~_2) of every pose-pose edge, then every prior, from the production factor kernel."""
        f, d = self.lp.num_edges + self.lp.num_priors, self.lp.dof
        r, j1, j2 = np.zeros((f, d)), np.zeros((f, d, d)), np.zeros((f, d, d))
        nat.check(self._lib.ps_debug_factor_blocks(self._h, nat.f64p(r), nat.f64p(j1), nat.f64p(j2)))
        return r, j1, j2

    def get_info(self):
        """ps_problem_info as a dict, read now (sizes, and the solver counters: restarts, launches, lagged-inverse and
        one-launch-PCG statistics)."""
        info = nat.ProblemInfo()
        nat.check(self._lib.ps_get_info(self._h, C.byref(info)))
        return {k: getattr(info, k) for k, _ in nat.ProblemInfo._fields_}

    def cg_restarts(self):
        """Restarts of the pipelined CG so far (ps_problem_info.cg_restarts)."""
        info = nat.ProblemInfo()
        nat.check(self._lib.ps_get_info(self._h, C.byref(info)))
        return int(info.cg_restarts)

    def cg_kernel_launches(self):
        """Kernels enqueued for iterations of the reduced solve so far (ps_problem_info.cg_kernel_launches)."""
        info = nat.ProblemInfo()
        nat.check(self._lib.ps_get_info(self._h, C.byref(info)))
        return int(info.cg_kernel_launches)

    def cg_persist_counts(self):
        """(folded CG solves run in one launch, of which timed out and were solved again launch by launch) --
        ps_problem_info.cg_persist_solves / cg_persist_failures (csrc/ps_k_cg_persist.h)."""
        info = nat.ProblemInfo()
        nat.check(self._lib.ps_get_info(self._h, C.byref(info)))
        return int(info.cg_persist_solves), int(info.cg_persist_failures)

    def set_option(self, name, value):
        nat.check(self._lib.ps_set_option(self._h, name.encode(), float(value)))

    # ---- tracing -------------------------------------------------------
    def set_profiling(self, level=2):
        """0 off; 1 = hipEvents around the whole iteration and the Schur kernel only (cheap);
        2 = around every stage (adds ~10 event records per iteration)."""
        nat.check(self._lib.ps_set_profiling(self._h, int(level)))

    def stage_times(self, reset=False):
        ms = (C.c_double * nat.PS_NUM_STAGES)()
        cnt = (C.c_int64 * nat.PS_NUM_STAGES)()
        nat.check(self._lib.ps_get_stage_times(self._h, ms, cnt, int(reset)))
        return {n: (ms[i], cnt[i]) for i, n in enumerate(nat.STAGE_NAMES)}


def dense_normal_solve(J, r, want_covariance=False):
    """Host-evaluated generic path: dx = (J^T J)^-1 (-J^T r) on the device."""
    lib = nat.require_gpu()
    J = np.ascontiguousarray(J, dtype=np.float64)
    r = np.ascontiguousarray(r, dtype=np.float64).reshape(-1)
    m, n = J.shape
    dx = np.zeros(n)
    cov = np.zeros((n, n)) if want_covariance else None
    nat.check(lib.ps_dense_normal_solve(nat.f64p(J), nat.f64p(r), m, n, nat.f64p(dx), nat.f64p(cov)))
    return (dx, cov) if want_covariance else dx


def band_inverse(A, ncb, dof, bw, chunk_nodes=0):
    """Inverse (fp32) of a symmetric positive definite block-banded matrix by the coarse level's factorisation kernels
    (include/pyslam_hip.h: ps_debug_band_inverse): chunk_nodes < 0 the serial column walk, 0 / > 0 the partitioned form.
    -> (inverse, GPU microseconds of the launches)."""
    lib = nat.require_gpu()
    A = np.ascontiguousarray(A, dtype=np.float64)
    n = ncb * dof
    assert A.shape == (n, n)
    out = np.zeros((n, n), dtype=np.float32)
    us = C.c_double(0.0)
    nat.check(lib.ps_debug_band_inverse(nat.f64p(A), ncb, dof, bw, chunk_nodes, out.ctypes.data_as(C.POINTER(C.c_float)),
                                        C.cast(C.byref(us), nat.c_f64p)))
    return out, us.value


class NotConverged(nat.NativeError):
    """The CG of the host-evaluated path stopped at max_iters with a residual too large to use."""


def sparse_normal_solve(J, r=None, rhs=None, tol=1e-12, max_iters=10000, accept=1e-6):
    """Host-evaluated generic path beyond the dense solver: (J^T J) dx = -J^T r (or = rhs) by CG on the device with J
    (scipy CSR) resident in HBM.  -> (dx, iterations, relative preconditioned residual).

    The reference solves these systems with a sparse LU (pyslam/problem.py:186), which does not care about the
    condition number; Jacobi-preconditioned CG on J^T J does (it squares it).  So the outcome is checked: a solve
    that ran out of iterations with a relative residual above `accept` raises NotConverged (an unconverged step or
    covariance column must not be used silently -- round-2 ADVICE), one that ended between `tol` and `accept` warns."""
    lib = nat.require_gpu()
    J = J.tocsr()
    J.sort_indices()
    Jt = J.T.tocsr()
    Jt.sort_indices()
    m, n = J.shape
    a = [np.ascontiguousarray(x, dtype=t) for x, t in ((J.indptr, np.int32), (J.indices, np.int32), (J.data, np.float64),
                                                        (Jt.indptr, np.int32), (Jt.indices, np.int32), (Jt.data, np.float64))]
    rr = None if r is None else np.ascontiguousarray(r, dtype=np.float64).reshape(-1)
    bb = None if rhs is None else np.ascontiguousarray(rhs, dtype=np.float64).reshape(-1)
    dx = np.zeros(n)
    its, rel = C.c_int32(), C.c_double()
    nat.check(lib.ps_sparse_normal_solve(m, n, nat.i32p(a[0]), nat.i32p(a[1]), nat.f64p(a[2]), nat.i32p(a[3]), nat.i32p(a[4]),
                                         nat.f64p(a[5]), nat.f64p(rr), nat.f64p(bb), float(tol), int(max_iters), nat.f64p(dx),
                                         C.byref(its), C.byref(rel)))
    relres = float(rel.value)
    if not (relres <= max(accept, tol)):
        raise NotConverged('sparse normal equations ({} unknowns): CG stopped after {} iterations at a relative residual of '
                           '{:.2e} (tolerance {:.1e}); the system is too ill-conditioned for Jacobi-preconditioned CG on '
                           'J^T J -- rescale the parameters / stiffnesses, or hold enough parameters constant for the '
                           'dense path (<= 2048 unknowns)'.format(n, int(its.value), relres, tol))
    if relres > 10. * tol:
        import warnings
        warnings.warn('pyslam_amd: sparse normal equations solved to a relative residual of {:.2e} only (tolerance {:.1e}, '
                      '{} iterations)'.format(relres, tol, int(its.value)), RuntimeWarning)
    return dx, int(its.value), relres


DIRECT_GENERIC_LIMIT = 8192       # csrc/ps_sparse.h: PS_SPD_MAXN


def sparse_normal_direct(J, r=None, rhs=None, refine_steps=3, accept=1e-8):
    """Host-evaluated generic path, 2 049 .. 8 192 unknowns (round 5): (J^T J) dx = -J^T r (or = rhs) by a dense blocked
    Cholesky on the device + refinement on the residual of the original system (include/pyslam_hip.h:
    ps_sparse_normal_direct) -- what the reference's sparse LU (pyslam/problem.py:186) is replaced by where the CG of
    sparse_normal_solve would not converge.  -> (dx, refinement steps allowed, ||rhs - J^T J dx|| / ||rhs||).
    Raises NotConverged if the residual stays above `accept` (a normal matrix singular to rounding)."""
    lib = nat.require_gpu()
    J = J.tocsr(copy=True)
    J.sum_duplicates()                # (the device looks entries up by binary search: one entry per (row, column))
    J.sort_indices()
    Jt = J.T.tocsr()
    Jt.sort_indices()
    m, n = J.shape
    a = [np.ascontiguousarray(x, dtype=t) for x, t in ((J.indptr, np.int32), (J.indices, np.int32), (J.data, np.float64),
                                                        (Jt.indptr, np.int32), (Jt.indices, np.int32), (Jt.data, np.float64))]
    rr = None if r is None else np.ascontiguousarray(r, dtype=np.float64).reshape(-1)
    bb = None if rhs is None else np.ascontiguousarray(rhs, dtype=np.float64).reshape(-1)
    dx = np.zeros(n)
    rel = C.c_double()
    nat.check(lib.ps_sparse_normal_direct(m, n, nat.i32p(a[0]), nat.i32p(a[1]), nat.f64p(a[2]), nat.i32p(a[3]), nat.i32p(a[4]),
                                          nat.f64p(a[5]), nat.f64p(rr), nat.f64p(bb), int(refine_steps), nat.f64p(dx),
                                          C.cast(C.byref(rel), nat.c_f64p)))
    relres = float(rel.value)
    if not (relres <= accept):
        raise NotConverged('dense direct solve of the normal equations ({} unknowns): relative residual {:.2e} after {} refinement '
                           'steps -- J^T J is singular to rounding'.format(n, relres, refine_steps))
    return dx, int(refine_steps), relres


class PhotometricDevice:
    """Device side of a Problem whose only block is a PhotometricResidualSE3 (include/pyslam_hip.h:
    ps_photometric_*): the pixel tables live in HBM, one call runs a whole Gauss-Newton iteration.
    Exposes the subset of DeviceProblem's interface that Problem.solve() drives."""

    def __init__(self, block, loss, split_params, stream=None):
        from pyslam_amd.lowering import _loss_id_k
        lib = nat.require_gpu()
        self._lib = lib
        self.split_params = bool(split_params)
        t = block.device_tables()
        d = nat.PhotoDesc()
        self._keep = t
        d.num_pixels = t['pt_ref'].shape[0]
        d.pt_ref, d.im_ref = nat.f64p(t['pt_ref']), nat.f64p(t['im_ref'])
        d.im_jac, d.tri_jac_d = nat.f64p(t['im_jac']), nat.f64p(t['tri_jac_d'])
        d.height, d.width = t['im_track'].shape
        d.im_track = nat.f64p(t['im_track'])
        for k in range(5):
            d.cam[k] = t['cam'][k]
        d.cam_type, d.cam_w, d.cam_h = t['cam_type'], t['cam_w'], t['cam_h']
        d.intensity_covar, d.depth_covar = t['intensity_covar'], t['depth_covar']
        lid, lk = _loss_id_k(loss)
        d.loss_id, d.loss_k = int(lid), float(lk)
        self.num_pixels = d.num_pixels
        self._h = nat.H()
        nat.check(lib.ps_photometric_create(C.byref(d), C.c_void_p(stream or 0), C.byref(self._h)))
        self._keep = None
        self._saved = None

    def close(self):
        if getattr(self, '_h', None):
            self._lib.ps_photometric_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- pose: (R, t) <-> 12 doubles -----------------------------------
    def set_pose(self, R, t):
        row = np.concatenate([np.asarray(R, dtype=float).reshape(9), np.asarray(t, dtype=float).reshape(3)])
        nat.check(self._lib.ps_photometric_set_pose(self._h, nat.f64p(row)))

    def get_pose(self):
        row = np.zeros(12)
        nat.check(self._lib.ps_photometric_get_pose(self._h, nat.f64p(row)))
        return row[:9].reshape(3, 3).copy(), row[9:].copy()

    def snapshot(self):
        self._saved = self.get_pose()

    def restore(self):
        self.set_pose(*self._saved)

    # ---- evaluation ----------------------------------------------------
    def eval_cost(self, include_all_constant=True):
        cost, n = C.c_double(), C.c_int64()
        nat.check(self._lib.ps_photometric_eval_cost(self._h, C.byref(cost), C.byref(n)))
        self.num_valid = n.value
        return cost.value

    def normal_equations(self):
        Hm, b = np.zeros((6, 6)), np.zeros(6)
        cost, n = C.c_double(), C.c_int64()
        nat.check(self._lib.ps_photometric_normal_equations(self._h, nat.f64p(Hm), nat.f64p(b), C.byref(cost), C.byref(n)))
        return Hm, b, cost.value, n.value

    def step(self, linesearch):
        """One iteration in place: (dx in [translation; rotation] order, cost)."""
        dx, cost = np.zeros(6), C.c_double()
        nat.check(self._lib.ps_photometric_iteration(self._h, int(self.split_params), int(bool(linesearch)),
                                                     nat.f64p(dx), C.byref(cost)))
        return dx, cost.value

    def gn_iteration(self, lam, pcg_tol, pcg_max_iters, linesearch):
        """Same tuple as DeviceProblem.gn_iteration: (cost, |dx|, solver iterations, relative residual)."""
        if lam:
            raise ValueError('the photometric device path has no damping (Options.lm_lambda must be 0)')
        dx, cost = self.step(linesearch)
        return cost, float(np.linalg.norm(dx)), 0, 0.0


_TABLES_F64 = ('obs_uvd', 'cams', 'stiff3', 'obs_groups', 'e_Tobs_inv', 'u_Tobs_inv', 'stiffd', 'edge_groups')
_TABLES_I32 = ('pose_rid', 'point_vid', 'obs_pose', 'obs_point', 'obs_grp', 'e_i', 'e_j', 'e_grp', 'u_i', 'u_grp')


def resident_tables(lp, device='cuda:0', params=True, tables=True):
    """A view of a LoweredProblem whose tables are torch tensors in HBM -- what a torch caller that built its problem on the
    device hands to DeviceProblem (ps_problem_desc.flags = PS_DESC_DEVICE_PARAMS | PS_DESC_DEVICE_TABLES): no host copy of
    the measurement tables is needed on the caller's side."""
    import copy
    import torch
    out = copy.copy(lp)
    if params:
        out.poses = torch.as_tensor(np.ascontiguousarray(lp.poses, dtype=np.float64)).to(device)
        out.points = torch.as_tensor(np.ascontiguousarray(lp.points, dtype=np.float64)).to(device)
    if tables:
        for name in _TABLES_F64:
            setattr(out, name, torch.as_tensor(np.ascontiguousarray(getattr(lp, name), dtype=np.float64)).to(device))
        for name in _TABLES_I32:
            setattr(out, name, torch.as_tensor(np.ascontiguousarray(getattr(lp, name), dtype=np.int32)).to(device))
    return out
