#!/usr/bin/env python
"""CPU prototype behind DESIGN.md section 3 "C2: what does and does not remove CG iterations" (round 6; round-5 verdict item 1b asked
for <= 35 CG iterations per call on BASELINE configuration C2 by a banded / incomplete factorisation preconditioner).
Test infrastructure: drives oracle/gn_oracle.py with numpy / scipy on the oracle's own normal equations of the 10 000-pose SE(3)
graph (first Gauss-Newton iteration), before any kernel is written.  Not collected by pytest.

    python tests/proto_posegraph_preconditioners.py [poses=10000] [loops=40001]

Measured here (tolerance 1e-14 on the preconditioned residual, block-Jacobi scaling everywhere; `coarse` = the hat-function level in
body-frame twists the device uses, exact inverse):
    block-Jacobi + coarse, one node per 20 poses (what the device runs)        80 iterations
    + exact solve of the block band |i-j| <= 1 / 2 / 4 of S, additive           71 / 70 / 69
    the same band solves multiplicatively (symmetric, two more SpMVs each)      33 / 33 / 34   (three operator applications: no gain)
    + exact inverses of chunks of 4 / 8 / 16 / 32 consecutive poses, additive   72 / 71 / 68 / 65
    + block-Jacobi mid levels at 2000 | 1000 | 2500+1000 | 5000+2500+1250 nodes 82 / 90 / 96 / 100 (additive multilevel: worse)
    coarse level alone made finer: one node per 20 / 10 / 5 / 3 / 2 poses        80 / 60 / 48 / 40 / 34  (A_c half-bandwidth 3 / 4 / 7 / 11 / 16)
Nothing local removes iterations (as on the bundle-adjustment systems, HISTORY.md section 3); what does is a coarse level so fine that
it is the problem itself -- 34 iterations need the exact solve of a 30 000-unknown banded system per CG iteration."""
import sys
import os
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyslam_amd import synthetic
from oracle import gn_oracle as orc

P = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
NL = int(sys.argv[2]) if len(sys.argv) > 2 else 40001
lp, _ = synthetic.pose_graph(num_poses=P, num_loops=NL, dof=6, seed=2)
H, b, _ = orc.normal_equations(lp, points_first=False)
H = H.tocsr(); poses = lp.poses
n, D = H.shape[0], 6
nr = n // D
Hb = H.tobsr(blocksize=(D, D)); diag = np.zeros((nr, D, D))
for i in range(nr):
    for k in range(Hb.indptr[i], Hb.indptr[i + 1]):
        if Hb.indices[k] == i:
            diag[i] = Hb.data[k]
L = np.linalg.cholesky(diag); Lis = sp.block_diag(list(np.linalg.inv(L)), format='csr')
S = (Lis @ H @ Lis.T).tocsr(); g = Lis @ b


def hat(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


Bs = []
for i in range(nr):
    R, t = poses[i, :9].reshape(3, 3), poses[i, 9:12]
    Ad = np.zeros((6, 6)); Ad[:3, :3] = R; Ad[:3, 3:] = hat(t) @ R; Ad[3:, 3:] = R
    Bs.append(L[i].T @ Ad)


def build_P(G):
    rows, cols, vals = [], [], []
    rr, cc = np.meshgrid(np.arange(6), np.arange(6), indexing='ij')
    for i in range(nr):
        u = i * G / (nr - 1); k = min(G - 1, int(u)); th = u - k
        for q, w in ((k, 1 - th), (k + 1, th)):
            if w == 0 and q == k + 1:
                continue
            rows.append((i * 6 + rr).ravel()); cols.append((q * 6 + cc).ravel()); vals.append((w * Bs[i]).ravel())
    return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, (G + 1) * 6))


def pcg(A, g, Minv, tol=1e-14, maxit=600):
    x = np.zeros_like(g); r = g.copy(); z = Minv(r); p = z.copy(); rz = r @ z; rz0 = rz; it = 0
    while it < maxit:
        q = A @ p; a = rz / (p @ q); x += a * p; r -= a * q; z = Minv(r); rzn = r @ z; it += 1
        if np.sqrt(abs(rzn) / rz0) < tol:
            break
        p = z + (rzn / rz) * p; rz = rzn
    return x, it


G0 = max(2, nr // 20)
P0 = build_P(G0); lu0 = spla.splu((P0.T @ S @ P0).tocsc())
coarse = lambda r: P0 @ lu0.solve(P0.T @ r)
print('block-Jacobi + coarse (1 node per 20 poses): %d iterations' % pcg(S, g, lambda r: r + coarse(r))[1], flush=True)
coo = S.tocoo()
for w in (1, 2, 4):
    m = np.abs(coo.row // D - coo.col // D) <= w
    lu = spla.splu(sp.csr_matrix((coo.data[m], (coo.row[m], coo.col[m])), shape=S.shape).tocsc())
    print('band(%d) solve + coarse, additive: %d' % (w, pcg(S, g, lambda r: lu.solve(r) + coarse(r))[1]), flush=True)

    def mult(r):
        z = lu.solve(r); z = z + coarse(r - S @ z); return z + lu.solve(r - S @ z)
    print('band(%d) solve, coarse, band solve (symmetric multiplicative): %d' % (w, pcg(S, g, mult)[1]), flush=True)
for m in (4, 8, 16, 32):
    invs = [np.linalg.inv(S[s0 * D:min(nr, s0 + m) * D, s0 * D:min(nr, s0 + m) * D].toarray()) for s0 in range(0, nr, m)]
    Binv = sp.block_diag(invs, format='csr')
    print('chunks of %d poses (exact inverses) + coarse, additive: %d' % (m, pcg(S, g, lambda r: Binv @ r + coarse(r))[1]), flush=True)
for G in (nr // 20, nr // 10, nr // 5, nr // 3, nr // 2):
    Pk = build_P(G); Ac = (Pk.T @ S @ Pk).tocsc(); lu = spla.splu(Ac); c = Ac.tocoo()
    print('coarse level of %d nodes (A_c block half-bandwidth %d): %d' % (G + 1, int(np.abs(c.row // 6 - c.col // 6).max()),
                                                                          pcg(S, g, lambda r: r + Pk @ lu.solve(Pk.T @ r))[1]), flush=True)
