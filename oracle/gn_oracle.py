"""CPU ORACLE -- numpy/scipy restatement of the reference's Gauss-Newton path.

TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's
``cpu_baseline`` leg may import this module; the product (pyslam_amd) never
does and raises if its HIP library is missing.

What it restates (reference file:line, all under /root/reference):

* residual + Jacobian algebra of the hot-path blocks --
  pyslam/residuals/reprojection_residual.py:13-37,
  pose_to_pose_residual.py:12-32, pose_residual.py:12-27,
  sensors/stereo_camera.py:100-134, and the liegroups conventions listed in
  SURVEY.md section 8c (third-party, unpinned: restated from its published
  numpy backend; pinned here by scipy expm/logm tests);
* IRLS scaling and normal equations -- pyslam/problem.py:279-360:
  J~ = diag(sqrt(w(r))) J, e~ = sqrt(w(r)) r, precision = J~^T J~,
  information = -J~^T e~, dx = scipy.sparse.linalg.spsolve (problem.py:186);
* the solve() control flow incl. the degenerate line search --
  pyslam/problem.py:130-194, 362-398.

It is *vectorised* over blocks (the reference loops in Python and cannot run
the BASELINE sizes), but the algebra per block and the
order of unknowns / residual rows are the reference's.

PINNING: tests/test_oracle.py checks this module against tests/golden/*.npz,
which oracle/gen_golden.py produced by running the verbatim reference in the
authoring container (with a decorator-only numba stand-in and this build's
liegroups).  The liegroups arithmetic itself is third-party and absent, so its
element-level parity is pinned independently (scipy) rather than by the
reference.
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

SMALL = 1e-8  # np.isclose(x, 0.)


# ---------------------------------------------------------------------------
# Lie group arithmetic, batched
# ---------------------------------------------------------------------------
def wedge3(v):
    v = np.atleast_2d(v)
    out = np.zeros((v.shape[0], 3, 3))
    out[:, 0, 1], out[:, 0, 2] = -v[:, 2], v[:, 1]
    out[:, 1, 0], out[:, 1, 2] = v[:, 2], -v[:, 0]
    out[:, 2, 0], out[:, 2, 1] = -v[:, 1], v[:, 0]
    return out


def unpack(rows, dof):
    n = 3 if dof == 6 else 2
    rows = np.asarray(rows, dtype=float).reshape(-1, n * n + n)
    return rows[:, :n * n].reshape(-1, n, n), rows[:, n * n:]


def pack(R, t):
    return np.concatenate([R.reshape(R.shape[0], -1), t], axis=1)


def compose(Ra, ta, Rb, tb):
    return np.einsum('nij,njk->nik', Ra, Rb), np.einsum('nij,nj->ni', Ra, tb) + ta


def inverse(R, t):
    Rt = np.transpose(R, (0, 2, 1))
    return Rt, -np.einsum('nij,nj->ni', Rt, t)


def so3_log(R):
    cos_a = np.clip(0.5 * np.trace(R, axis1=1, axis2=2) - 0.5, -1., 1.)
    ang = np.arccos(cos_a)
    small = np.abs(ang) <= SMALL
    A = R - np.transpose(R, (0, 2, 1))
    with np.errstate(divide='ignore', invalid='ignore'):
        f = np.where(small, 0., 0.5 * ang / np.sin(ang))
    big = f[:, None] * np.stack([A[:, 2, 1], A[:, 0, 2], A[:, 1, 0]], axis=1)
    B = R - np.identity(3)
    tiny = np.stack([B[:, 2, 1], B[:, 0, 2], B[:, 1, 0]], axis=1)
    return np.where(small[:, None], tiny, big)


def so3_inv_left_jacobian(phi):
    ang = np.linalg.norm(phi, axis=1)
    small = np.abs(ang) <= SMALL
    W = wedge3(phi)
    I = np.identity(3)
    safe = np.where(small, 1., ang)
    axis = phi / safe[:, None]
    half = 0.5 * safe
    hc = half / np.tan(half)
    big = (hc[:, None, None] * I + (1. - hc)[:, None, None] * axis[:, :, None] * axis[:, None, :]
           - half[:, None, None] * wedge3(axis))
    return np.where(small[:, None, None], I - 0.5 * W, big)


def so3_left_jacobian(phi):
    ang = np.linalg.norm(phi, axis=1)
    small = np.abs(ang) <= SMALL
    I = np.identity(3)
    safe = np.where(small, 1., ang)
    axis = phi / safe[:, None]
    s, c = np.sin(safe), np.cos(safe)
    big = ((s / safe)[:, None, None] * I
           + (1. - s / safe)[:, None, None] * axis[:, :, None] * axis[:, None, :]
           + ((1. - c) / safe)[:, None, None] * wedge3(axis))
    return np.where(small[:, None, None], I + 0.5 * wedge3(phi), big)


def so3_exp(phi):
    ang = np.linalg.norm(phi, axis=1)
    small = np.abs(ang) <= SMALL
    I = np.identity(3)
    safe = np.where(small, 1., ang)
    axis = phi / safe[:, None]
    s, c = np.sin(safe), np.cos(safe)
    big = (c[:, None, None] * I + (1. - c)[:, None, None] * axis[:, :, None] * axis[:, None, :]
           + s[:, None, None] * wedge3(axis))
    return np.where(small[:, None, None], I + wedge3(phi), big)


_J2 = np.array([[0., -1.], [1., 0.]])


def so2_jac(phi, inverse_):
    small = np.abs(phi) <= SMALL
    safe = np.where(small, 1., phi)
    I = np.identity(2)
    if inverse_:
        half = 0.5 * safe
        big = (half / np.tan(half))[:, None, None] * I - half[:, None, None] * _J2
        tiny = I - 0.5 * phi[:, None, None] * _J2
    else:
        big = (np.sin(safe) / safe)[:, None, None] * I + ((1. - np.cos(safe)) / safe)[:, None, None] * _J2
        tiny = I + 0.5 * phi[:, None, None] * _J2
    return np.where(small[:, None, None], tiny, big)


def se_log(R, t, dof):
    if dof == 6:
        phi = so3_log(R)
        rho = np.einsum('nij,nj->ni', so3_inv_left_jacobian(phi), t)
        return np.concatenate([rho, phi], axis=1)
    phi = np.arctan2(R[:, 1, 0], R[:, 0, 0])
    rho = np.einsum('nij,nj->ni', so2_jac(phi, True), t)
    return np.concatenate([rho, phi[:, None]], axis=1)


def se_exp(xi, dof):
    xi = np.atleast_2d(xi)
    if dof == 6:
        rho, phi = xi[:, :3], xi[:, 3:]
        return so3_exp(phi), np.einsum('nij,nj->ni', so3_left_jacobian(phi), rho)
    rho, phi = xi[:, :2], xi[:, 2]
    c, s = np.cos(phi), np.sin(phi)
    R = np.stack([np.stack([c, -s], 1), np.stack([s, c], 1)], 1)
    return R, np.einsum('nij,nj->ni', so2_jac(phi, False), rho)


def se_adjoint(R, t, dof):
    n = R.shape[0]
    if dof == 6:
        out = np.zeros((n, 6, 6))
        out[:, :3, :3] = R
        out[:, :3, 3:] = np.einsum('nij,njk->nik', wedge3(t), R)
        out[:, 3:, 3:] = R
        return out
    out = np.tile(np.identity(3), (n, 1, 1))
    out[:, :2, :2] = R
    out[:, 0, 2] = t[:, 1]
    out[:, 1, 2] = -t[:, 0]
    return out


# ---------------------------------------------------------------------------
# losses (reference pyslam/losses.py), by id
# ---------------------------------------------------------------------------
def loss_rho(lid, k, x):
    lid = int(lid)
    a = np.abs(x)
    if lid == 0:
        return 0.5 * x * x
    if lid == 1:
        return a
    if lid == 2:
        return (0.5 * k ** 2) * np.log(1. + (x / k) ** 2)
    if lid == 3:
        return np.where(a <= k, 0.5 * x * x, k * (a - 0.5 * k))
    if lid == 4:
        c = k ** 2 / 6.
        return np.where(a <= k, c * (1. - (1. - (x / k) ** 2) ** 3), c)
    if lid == 5:
        return 0.5 * (k + 1.) * np.log(1. + x * x / k)
    raise ValueError(lid)


def loss_weight(lid, k, x):
    lid = int(lid)
    a = np.abs(x)
    with np.errstate(divide='ignore', invalid='ignore'):
        if lid == 0:
            return np.ones_like(x)
        if lid == 1:
            return np.where(a <= SMALL, np.nan, 1. / a)
        if lid == 2:
            return 1. / (1. + (x / k) ** 2)
        if lid == 3:
            return np.where(a <= k, 1., k / a)
        if lid == 4:
            return np.where(a <= k, 1. - (x / k) ** 2, 0.)
        if lid == 5:
            return (k + 1.) / (k + x * x)
    raise ValueError(lid)


def _by_group(groups, grp, col_id, col_k, fn, x):
    """Apply fn(loss_id, k, x_rows) group by group (x: (n, m))."""
    out = np.empty_like(x)
    for g in range(groups.shape[0]):
        m = grp == g
        if m.any():
            out[m] = fn(groups[g, col_id], groups[g, col_k], x[m])
    return out


# ---------------------------------------------------------------------------
# residual blocks, batched
# ---------------------------------------------------------------------------
def eval_reproj(lp, jac=True):
    """(r (N,3), J_pose (N,3,6), J_point (N,3,3)) -- stiffness applied, no IRLS."""
    R, t = unpack(lp.poses, 6)
    Ro, to = R[lp.obs_pose], t[lp.obs_pose]
    pc = np.einsum('nij,nj->ni', Ro, lp.points[lp.obs_point]) + to
    g = lp.obs_groups[lp.obs_grp]
    cam = lp.cams[g[:, 0].astype(int)]
    S = lp.stiff3[g[:, 1].astype(int)].reshape(-1, 3, 3)
    cu, cv, fu, fv, b = cam.T
    rgbd = b < 0                       # RGB-D rows (reference sensors/rgbd_camera.py:96-135): third coordinate is z
    iz = 1. / pc[:, 2]
    uvd = np.stack([fu * pc[:, 0] * iz + cu, fv * pc[:, 1] * iz + cv,
                    np.where(rgbd, pc[:, 2], fu * b * iz)], axis=1)
    r = np.einsum('nij,nj->ni', S, uvd - lp.obs_uvd)
    if not jac:
        return r
    iz2 = iz * iz
    Jc = np.zeros((pc.shape[0], 3, 3))
    Jc[:, 0, 0] = fu * iz
    Jc[:, 0, 2] = -fu * pc[:, 0] * iz2
    Jc[:, 1, 1] = fv * iz
    Jc[:, 1, 2] = -fv * pc[:, 1] * iz2
    Jc[:, 2, 2] = np.where(rgbd, 1., -fu * b * iz2)
    odot = np.zeros((pc.shape[0], 3, 6))
    odot[:, :, :3] = np.identity(3)
    odot[:, :, 3:] = wedge3(-pc)
    SJ = np.einsum('nij,njk->nik', S, Jc)
    return r, np.einsum('nij,njk->nik', SJ, odot), np.einsum('nij,njk->nik', SJ, Ro)


def eval_edges(lp, jac=True):
    """Binary pose-pose blocks: (r (E,d), J_1 (E,d,d), J_2 (E,d,d))."""
    d = lp.dof
    R, t = unpack(lp.poses, d)
    R1, t1 = R[lp.e_i], t[lp.e_i]
    R2, t2 = R[lp.e_j], t[lp.e_j]
    Ro, to = unpack(lp.e_Tobs_inv, d)
    R1i, t1i = inverse(R1, t1)
    Rm, tm = compose(R1i, t1i, Ro, to)            # T_1^-1 . T_obs^-1
    Re, te = compose(R2, t2, Rm, tm)              # T_2 . (T_1^-1 . T_obs^-1)
    S = lp.stiffd[lp.edge_groups[lp.e_grp, 0].astype(int)].reshape(-1, d, d)
    r = np.einsum('nij,nj->ni', S, se_log(Re, te, d))
    if not jac:
        return r
    R21, t21 = compose(R2, t2, R1i, t1i)
    J1 = -np.einsum('nij,njk->nik', S, se_adjoint(R21, t21, d))
    return r, J1, S.copy()


def eval_priors(lp, jac=True):
    d = lp.dof
    R, t = unpack(lp.poses, d)
    Ro, to = unpack(lp.u_Tobs_inv, d)
    Re, te = compose(R[lp.u_i], t[lp.u_i], Ro, to)
    S = lp.stiffd[lp.edge_groups[lp.u_grp, 0].astype(int)].reshape(-1, d, d)
    r = np.einsum('nij,nj->ni', S, se_log(Re, te, d))
    return (r, S.copy()) if jac else r


def eval_cost(lp, include_all_constant=True):
    """Sum of rho(r) over every block (reference problem.py:110-128)."""
    cost = 0.
    if lp.num_priors:
        r = eval_priors(lp, jac=False)
        m = np.ones(len(r), bool) if include_all_constant else lp.pose_rid[lp.u_i] >= 0
        cost += _by_group(lp.edge_groups, lp.u_grp, 1, 2, loss_rho, r)[m].sum()
    if lp.num_edges:
        r = eval_edges(lp, jac=False)
        m = np.ones(len(r), bool) if include_all_constant else \
            (lp.pose_rid[lp.e_i] >= 0) | (lp.pose_rid[lp.e_j] >= 0)
        cost += _by_group(lp.edge_groups, lp.e_grp, 1, 2, loss_rho, r)[m].sum()
    if lp.num_obs:
        r = eval_reproj(lp, jac=False)
        m = np.ones(len(r), bool) if include_all_constant else \
            (lp.pose_rid[lp.obs_pose] >= 0) | (lp.point_vid[lp.obs_point] >= 0)
        cost += _by_group(lp.obs_groups, lp.obs_grp, 2, 3, loss_rho, r)[m].sum()
    return float(cost)


# ---------------------------------------------------------------------------
# normal equations in the reference's unknown order
# ---------------------------------------------------------------------------
def unknown_offsets(lp, points_first=True):
    """Column offset of every variable pose / point in dx (reference
    problem.py:252-277: param_dict insertion order minus constants)."""
    d = lp.dof
    nr, nv = lp.num_reduced, lp.num_var_points
    if points_first:
        pt0, pose0 = 0, 3 * nv
    else:
        pose0, pt0 = 0, d * nr
    pose_off = np.where(lp.pose_rid >= 0, pose0 + d * lp.pose_rid, -1)
    point_off = np.where(lp.point_vid >= 0, pt0 + 3 * lp.point_vid, -1)
    return pose_off, point_off, d * nr + 3 * nv


def _coo_block(rows0, cols0, J, keep):
    """COO triplets of dense blocks J[n] placed at (rows0[n], cols0[n])."""
    J = J[keep]
    n, m, k = J.shape
    rr = (rows0[keep][:, None, None] + np.arange(m)[None, :, None]) + np.zeros((1, 1, k), int)
    cc = (cols0[keep][:, None, None] + np.arange(k)[None, None, :]) + np.zeros((1, m, 1), int)
    return rr.ravel(), cc.ravel(), J.ravel()


def linearize(lp, points_first=True):
    """IRLS-scaled sparse Jacobian J~ (CSR), e~ and the cost at the
    linearisation point, rows in residual-block insertion order
    (priors, edges, observations -- the order synthetic.to_objects adds them)."""
    d = lp.dof
    pose_off, point_off, n = unknown_offsets(lp, points_first)
    rows, cols, vals, es = [], [], [], []
    row0, cost = 0, 0.

    def scale(groups, grp, cid, ck, r):
        w = _by_group(groups, grp, cid, ck, loss_weight, r)
        return np.sqrt(w), _by_group(groups, grp, cid, ck, loss_rho, r)

    if lp.num_priors:
        r, J = eval_priors(lp)
        s, rho = scale(lp.edge_groups, lp.u_grp, 1, 2, r)
        act = pose_off[lp.u_i] >= 0
        base = row0 + d * np.arange(len(r))
        a, b, c = _coo_block(base, pose_off[lp.u_i], s[:, :, None] * J, act)
        rows.append(a); cols.append(b); vals.append(c)
        es.append(np.where(act[:, None], s * r, 0.).ravel())
        cost += rho[act].sum()
        row0 += d * len(r)
    if lp.num_edges:
        r, J1, J2 = eval_edges(lp)
        s, rho = scale(lp.edge_groups, lp.e_grp, 1, 2, r)
        a1, a2 = pose_off[lp.e_i] >= 0, pose_off[lp.e_j] >= 0
        act = a1 | a2
        base = row0 + d * np.arange(len(r))
        for J, off, m in ((J1, pose_off[lp.e_i], a1), (J2, pose_off[lp.e_j], a2)):
            a, b, c = _coo_block(base, off, s[:, :, None] * J, m)
            rows.append(a); cols.append(b); vals.append(c)
        es.append(np.where(act[:, None], s * r, 0.).ravel())
        cost += rho[act].sum()
        row0 += d * len(r)
    if lp.num_obs:
        r, Jp, Jl = eval_reproj(lp)
        s, rho = scale(lp.obs_groups, lp.obs_grp, 2, 3, r)
        a1, a2 = pose_off[lp.obs_pose] >= 0, point_off[lp.obs_point] >= 0
        act = a1 | a2
        base = row0 + 3 * np.arange(len(r))
        for J, off, m in ((Jp, pose_off[lp.obs_pose], a1), (Jl, point_off[lp.obs_point], a2)):
            a, b, c = _coo_block(base, off, s[:, :, None] * J, m)
            rows.append(a); cols.append(b); vals.append(c)
        es.append(np.where(act[:, None], s * r, 0.).ravel())
        cost += rho[act].sum()
        row0 += 3 * len(r)

    J = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                      shape=(row0, n))
    return J, np.concatenate(es), float(cost)


def normal_equations(lp, points_first=True, lm_lambda=0.):
    """precision = J~^T J~, information = -J~^T e~ (reference problem.py:332-333).
    lm_lambda > 0 adds Marquardt damping lambda * diag(J~^T J~) (a build-only option; the
    reference is lambda = 0)."""
    J, e, cost = linearize(lp, points_first)
    JT = J.T.tocsr()
    P = JT.dot(J).tocsr()
    if lm_lambda:
        P = (P + lm_lambda * sp.diags(P.diagonal())).tocsr()
    return P, -JT.dot(e), cost


def apply_update(lp, dx, points_first=True):
    """T <- exp(xi) T for poses (left perturbation), p += dp for points."""
    out = lp.copy()
    d = lp.dof
    pose_off, point_off, _ = unknown_offsets(lp, points_first)
    vp = np.nonzero(pose_off >= 0)[0]
    if vp.size:
        xi = dx[pose_off[vp][:, None] + np.arange(d)[None, :]]
        Re, te = se_exp(xi, d)
        R, t = unpack(lp.poses[vp], d)
        Rn, tn = compose(Re, te, R, t)
        out.poses[vp] = pack(Rn, tn)
    vl = np.nonzero(point_off >= 0)[0]
    if vl.size:
        out.points[vl] = lp.points[vl] + dx[point_off[vl][:, None] + np.arange(3)[None, :]]
    return out


def gauss_newton_step(lp, points_first=True, linear_solver='spsolve'):
    """One reference GN step: returns (dx, cost at the linearisation point)."""
    P, b, cost = normal_equations(lp, points_first)
    if linear_solver == 'spsolve':
        dx = spla.spsolve(P, b)
    else:
        dx = schur_solve(lp, P, b, points_first)
    return np.atleast_1d(dx), cost


def schur_solve(lp, P, b, points_first=True):
    """Landmark elimination on the CPU (block-diagonal 3x3 inverse + sparse
    reduced solve): the algebra the HIP path implements, for the
    'cpu_baseline' Schur timing and as a cross-check of spsolve."""
    d = lp.dof
    pose_off, point_off, n = unknown_offsets(lp, points_first)
    nr, nv = lp.num_reduced, lp.num_var_points
    ip = (np.sort(pose_off[pose_off >= 0])[:, None] + np.arange(d)[None, :]).reshape(-1)
    il = (np.sort(point_off[point_off >= 0])[:, None] + np.arange(3)[None, :]).reshape(-1)
    P = P.tocsr()
    if nv == 0:
        return spla.spsolve(P.tocsc(), b)
    Hll = P[il][:, il]
    blocks = np.zeros((nv, 3, 3))
    coo = Hll.tocoo()
    blocks[coo.row // 3, coo.row % 3, coo.col % 3] = coo.data
    inv = np.linalg.inv(blocks)
    Hll_inv = sp.block_diag(list(inv), format='csr') if nv < 4096 else _bdiag(inv)
    dx = np.zeros(n)
    if nr == 0:
        dx[il] = Hll_inv.dot(b[il])
        return dx
    Hpl = P[ip][:, il]
    Hpp = P[ip][:, ip]
    Y = Hpl.dot(Hll_inv)
    S = (Hpp - Y.dot(Hpl.T)).tocsc()
    g = b[ip] - Y.dot(b[il])
    dxp = spla.spsolve(S, g) if S.shape[0] > 1 else g / S.toarray().ravel()
    dx[ip] = dxp
    dx[il] = Hll_inv.dot(b[il] - Hpl.T.dot(dxp))
    return dx


def _bdiag(blocks):
    n = blocks.shape[0]
    r = (3 * np.arange(n)[:, None, None] + np.arange(3)[None, :, None]) + np.zeros((1, 1, 3), int)
    c = (3 * np.arange(n)[:, None, None] + np.arange(3)[None, None, :]) + np.zeros((1, 3, 1), int)
    return sp.csr_matrix((blocks.ravel(), (r.ravel(), c.ravel())), shape=(3 * n, 3 * n))


# ---------------------------------------------------------------------------
# solve(): reference control flow (problem.py:130-194, 362-398)
# ---------------------------------------------------------------------------
DEFAULT_OPTIONS = dict(max_iters=100, min_update_norm=1e-6, min_cost=1e-12,
                       min_cost_decrease=0.9, linesearch_alpha=0.8, linesearch_max_iters=10,
                       linesearch_min_cost_decrease=0.9, allow_nondecreasing_steps=False,
                       max_nondecreasing_steps=3, num_threads=1)


def line_search(lp, dx, opt, points_first):
    """The reference's backtracking search, quirk included: every trial point
    uses best_step_size (problem.py:375), so the step is always 1 and the cost
    of the full step is evaluated twice."""
    step, best_step, best_cost, iters = 1., 1., np.inf, 0
    while True:
        iters += 1
        test_cost = eval_cost(apply_update(lp, best_step * dx, points_first))
        if iters < opt['linesearch_max_iters'] and \
                test_cost < opt['linesearch_min_cost_decrease'] * best_cost:
            best_cost, best_step = test_cost, step
        else:
            if test_cost < best_cost:
                best_cost, best_step = test_cost, step
            break
        step = opt['linesearch_alpha'] * step
    return best_step, best_cost


def solve(lp, options=None, points_first=True, linear_solver='spsolve'):
    """Returns (final LoweredProblem, dict(cost_history, iter_dx, iter_cost))."""
    opt = dict(DEFAULT_OPTIONS)
    opt.update(options or {})
    cur = lp.copy()
    cost = eval_cost(cur)
    dx = np.array([100.])
    iters, nondecreasing = 0, 0
    history, dxs, best = [cost], [], None
    done = False
    while not done:
        iters += 1
        prev = cost
        dx, lin_cost = gauss_newton_step(cur, points_first, linear_solver)
        if opt['linesearch_max_iters'] > 0:
            step, cost = line_search(cur, dx, opt, points_first)
        else:
            step, cost = 1., lin_cost
        dx = step * dx
        history.append(cost)
        dxs.append(dx)
        cur = apply_update(cur, dx, points_first)
        done = iters > opt['max_iters'] or np.linalg.norm(dx) < opt['min_update_norm'] \
            or cost < opt['min_cost']
        if opt['allow_nondecreasing_steps']:
            if nondecreasing == 0:
                best = cur.copy()
            if cost >= opt['min_cost_decrease'] * prev:
                nondecreasing += 1
            else:
                nondecreasing = 0
            if nondecreasing >= opt['max_nondecreasing_steps']:
                done = True
                cur = best
        else:
            done = done or cost >= opt['min_cost_decrease'] * prev
    return cur, {'cost_history': np.array(history), 'iter_dx': dxs,
                 'iter_cost': np.array(history[1:])}
