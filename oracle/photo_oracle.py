"""CPU restatement of the reference's dense photometric alignment step
(pyslam/residuals/photometric_residual.py:38-161 inside pyslam/problem.py:182-194, 279-360).

TEST INFRASTRUCTURE ONLY: imported by tests/ as the checker,
never by the product path.  Works on plain tables (the constructor's outputs), one pixel per row, written
with explicit per-pixel formulas rather than the product class's stacked einsum expressions.
Pinned against tests/golden/photometric.npz, which oracle/gen_golden.py produced by running the reference
class and the reference Problem, and -- the image lookup on its own -- against tests/golden/bilinear.npz: outputs of the
reference's own kernel body with the four spellings that keep it from running repaired (x[1], y[1], out[1] -> [0],
np.int -> int; gen_golden.py: reference_bilinear_body, which the photometric case uses too).
"""
import numpy as np

from oracle import gn_oracle as orc


def tables(camera_params, im_ref, depth_ref, im_jac, min_grad, rgbd):
    """Constructor (:44-81): valid + strong-gradient pixels, triangulated points and d point / d depth."""
    cu, cv, fu, fv, b, w, h = camera_params
    w, h = int(w), int(h)
    u, v = np.meshgrid(np.arange(w, dtype=float), np.arange(h, dtype=float), indexing='xy')
    u, v, d = u.ravel(), v.ravel(), np.asarray(depth_ref, dtype=float).ravel()
    I = np.asarray(im_ref, dtype=float).ravel()
    g = np.stack([np.asarray(im_jac[0], dtype=float).ravel(), np.asarray(im_jac[1], dtype=float).ravel()], axis=1)
    with np.errstate(invalid='ignore'):
        keep = (d > 0) & (v > 0) & (v < h) & (u > 0) & (u < w)             # stereo_camera.py:93-97 / rgbd_camera.py:91-95
        if not rgbd:
            keep &= d < w
    keep &= np.sqrt(g[:, 0] ** 2 + g[:, 1] ** 2) >= min_grad             # :72-76 (NaN depth fails `keep` already)
    u, v, d, I, g = u[keep], v[keep], d[keep], I[keep], g[keep]
    pt = np.empty((u.size, 3)); tj = np.empty((u.size, 3))
    if rgbd:                                                               # rgbd_camera.py:123-168
        pt[:, 0], pt[:, 1], pt[:, 2] = (u - cu) * d / fu, (v - cv) * d / fv, d
        tj[:, 0], tj[:, 1], tj[:, 2] = (u - cu) / fu, (v - cv) / fv, 1.
    else:                                                                  # stereo_camera.py:123-148
        bd = b / d
        pt[:, 0], pt[:, 1], pt[:, 2] = (u - cu) * bd, (v - cv) * bd * fu / fv, fu * bd
        tj[:, 0], tj[:, 1], tj[:, 2] = (cu - u) * bd / d, (cv - v) * bd / d * fu / fv, -fu * bd / d
    return dict(pt_ref=pt, im_ref=I, im_jac=g, tri_jac_d=tj, cam=(cu, cv, fu, fv, b), w=w, h=h, rgbd=bool(rgbd))


def bilinear(im, x, y):
    """utils.py:27-75 with [0] indices: weights from the unclipped corners, corners clamped afterwards."""
    h, w = im.shape
    out = np.empty(len(x))
    for i in range(len(x)):
        x0, y0 = int(x[i]), int(y[i])
        x1, y1 = x0 + 1, y0 + 1
        wa, wb = (x1 - x[i]) * (y1 - y[i]), (x1 - x[i]) * (y[i] - y0)
        wc, wd = (x[i] - x0) * (y1 - y[i]), (x[i] - x0) * (y[i] - y0)
        x0, x1 = min(max(x0, 0), w - 1), min(max(x1, 0), w - 1)
        y0, y1 = min(max(y0, 0), h - 1), min(max(y1, 0), h - 1)
        out[i] = wa * im[y0, x0] + wb * im[y1, x0] + wc * im[y0, x1] + wd * im[y1, x1]
    return out


def evaluate(tb, im_track, var_i, var_d, R, t):
    """evaluate() (:83-143): residual (valid pixels only), its N x 6 Jacobian ([translation | rotation]), valid mask."""
    cu, cv, fu, fv, b = tb['cam']
    p = tb['pt_ref'] @ R.T + t
    with np.errstate(divide='ignore', invalid='ignore'):
        iz = 1. / p[:, 2]
        u, v = fu * p[:, 0] * iz + cu, fv * p[:, 1] * iz + cv
        d = p[:, 2] if tb['rgbd'] else fu * b * iz
        valid = (d > 0) & (v > 0) & (v < tb['h']) & (u > 0) & (u < tb['w'])
        if not tb['rgbd']:
            valid &= d < tb['w']
    p, iz, u, v = p[valid], iz[valid], u[valid], v[valid]
    gu, gv = tb['im_jac'][valid, 0], tb['im_jac'][valid, 1]
    g = np.stack([gu * fu * iz, gv * fv * iz, -(gu * fu * p[:, 0] + gv * fv * p[:, 1]) * iz * iz], axis=1)   # :110-111
    jd = np.sum((g @ R) * tb['tri_jac_d'][valid], axis=1)                                                  # :112-116
    s = 1. / np.sqrt(var_i + var_d * jd ** 2)                                                              # :120-121
    r = s * (bilinear(np.asarray(im_track, dtype=float), u, v) - tb['im_ref'][valid])
    J = np.empty((r.size, 6))
    J[:, :3] = g
    J[:, 3:] = np.cross(p, g)                                   # g (-p^) = p x g                          # :133-137
    return r, s[:, None] * J, valid


def normal_equations(tb, im_track, var_i, var_d, R, t, loss_id=0, loss_k=1.):
    """H = J~^T J~, b = -J~^T r~, cost = sum rho(r) with element-wise IRLS (problem.py:351-360, 329-335)."""
    r, J, valid = evaluate(tb, im_track, var_i, var_d, R, t)
    w = orc.loss_weight(loss_id, loss_k, r)
    H = (J * w[:, None]).T @ J
    return H, -(J * w[:, None]).T @ r, float(np.sum(orc.loss_rho(loss_id, loss_k, r))), int(valid.sum())


def gn_step(tb, im_track, var_i, var_d, R, t, loss_id=0, loss_k=1., split=False):
    """solve_one_iter + update: dx in [translation; rotation] order, the new (R, t), the linearisation cost."""
    H, b, cost, _ = normal_equations(tb, im_track, var_i, var_d, R, t, loss_id, loss_k)
    dx = np.linalg.solve(H, b)
    if split:                                                   # (SO3, translation) parameters: problem.py:405-409
        return dx, orc.so3_exp(dx[None, 3:])[0] @ R, t + dx[:3], cost
    Re, te = orc.se_exp(dx, 6)                                  # T <- exp(dx) T (liegroups perturb)
    return dx, Re[0] @ R, Re[0] @ t + te[0], cost
