"""Why the pipelined (Chronopoulos-Gear) CG of the folded two-level system can break down -- numpy emulation.

The folded system is A~ = V^T S^ V with V = [I, X]: singular (null space {(-X c, c)}), consistent.  In exact arithmetic
every residual lies in range(V^T): its coarse part is X^T times its fine part.  The emulation runs the device's
recurrences (r <- r - alpha s, s <- w + beta s, w = A~ r; alpha from gamma, delta and the previous alpha) in float64,
in float64 with that invariant restored after every update, and in 80-bit long double, on unit right-hand sides of an
SE(2) pose graph with the 1e12 prior, and prints the violation |r_c - X^T r_f| / |r|: in float64 it grows to O(1) by
the time the true residual has dropped ten orders -- rounding in s leaks out of range(V^T), A~ cannot act on that
component, so it is never reduced while gamma = r.r keeps counting it -- four orders less in long double, zero with the
projection.  delta - beta gamma / alpha is then a difference of quantities polluted by the leaked component: when it
turns non-positive the device reports a breakdown and RESTARTS from the true residual V^T (g^ - S^ V x~), which is the
projection (cg_fused_run).  usage: python tools/cg_drift.py [seed]"""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np, scipy.linalg as sl
from oracle import gn_oracle as orc
from pyslam_amd import synthetic, losses

def build(seed, P=300, dof=3):
    rng = np.random.default_rng(9000 + seed)
    lp, _ = synthetic.pose_graph(num_poses=P, num_loops=int(rng.integers(1, 3 * P)), dof=dof, seed=seed, loss=losses.HuberLoss(1.5))
    H, _, _ = orc.normal_equations(lp, points_first=False)
    return lp, H.toarray()

def setup(H, D, G):
    n = H.shape[0]; nr = n // D
    Linv = np.zeros((n, n))
    for i in range(nr):
        L = np.linalg.cholesky(H[i*D:(i+1)*D, i*D:(i+1)*D]); Linv[i*D:(i+1)*D, i*D:(i+1)*D] = np.linalg.inv(L)
    S = Linv @ H @ Linv.T
    ncb = G + 1
    Pm = np.zeros((n, ncb * D))
    for i in range(nr):
        u = i * G / (nr - 1); k = min(G - 1, int(u)); th = u - k
        Pm[i*D:(i+1)*D, k*D:(k+1)*D] = (1 - th) * np.eye(D)
        Pm[i*D:(i+1)*D, (k+1)*D:(k+2)*D] += th * np.eye(D)
    Ac = Pm.T @ S @ Pm
    Lc = np.linalg.cholesky(Ac)
    X = Pm @ np.linalg.inv(Lc).T
    return S, Linv, X

def cg_cgear(S, X, bf, dtype, iters=400, tol=1e-13, fix=None):
    """Chronopoulos-Gear CG on A~ = V^T S V, V = [I, X]; returns per-iteration (gamma, denom, invariant violation, true relres)."""
    S = S.astype(dtype); X = X.astype(dtype); bf = bf.astype(dtype)
    n, nc = X.shape
    def A(v):
        y = S @ (v[:n] + X @ v[n:])
        return np.concatenate([y, X.T @ y])
    b = np.concatenate([bf, X.T @ bf])
    x = np.zeros(n + nc, dtype); r = b.copy(); w = A(r)
    p = np.zeros_like(r); s = np.zeros_like(r)
    g_prev = a_prev = None
    log = []
    g0 = None
    for k in range(iters):
        gamma = r @ r; delta = w @ r
        if g0 is None: g0 = gamma
        if k == 0: beta = dtype(0); denom = delta
        else:
            beta = gamma / g_prev; denom = delta - beta * gamma / a_prev
        viol = np.linalg.norm((r[n:] - X.T @ r[:n]).astype(float)) / np.linalg.norm(r.astype(float))
        xhat = (x[:n] + X @ x[n:])
        true = np.linalg.norm((bf - S @ xhat).astype(float)) / np.linalg.norm(bf.astype(float))
        log.append((float(gamma / g0) ** 0.5, float(denom), viol, true))
        if not (denom > 0) or gamma <= tol * tol * g0:
            break
        alpha = gamma / denom
        p = r + beta * p; s = w + beta * s
        x = x + alpha * p; r = r - alpha * s
        if fix == 'project':                 # restore the invariant r_c = X^T r_f (r in range(V^T))
            r[n:] = X.T @ r[:n]
        w = A(r)
        g_prev, a_prev = gamma, alpha
    return log

if __name__ == '__main__':
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    lp, H = build(seed)
    D = lp.dof
    S, Linv, X = setup(H, D, 30)
    for col in (0, 5 * D + 1, 150 * D, 299 * D + 2):
        e = np.zeros(H.shape[0]); e[col] = 1.
        bf = Linv @ e
        for name, dt, fix in (('f64', np.float64, None), ('f64+project', np.float64, 'project'), ('f80', np.longdouble, None)):
            log = cg_cgear(S, X, bf, dt, fix=fix)
            last = log[-1]
            print('seed %d col %4d %-12s iters %3d  relres(rec) %.1e  denom %.2e  |r_c - X^T r_f|/|r| %.1e  true relres %.1e' % (seed, col, name, len(log), last[0], last[1], last[2], last[3]))
