// ps_sparse.h -- host-evaluated generic path at scale: Jacobi-preconditioned CG on the normal equations J^T J dx = rhs
// with J held as CSR (and its transpose as CSR) in HBM; J^T J is never formed.  Used when user-defined residual blocks,
// losses or parameter types keep a problem off the typed kernels and it has more unknowns than the dense Cholesky path
// takes (ps_dense_normal_solve: n <= 2048) -- the reference solves such problems with scipy's sparse LU
// (pyslam/problem.py:186); here only the block evaluation stays on the host (it is user Python code).
// Part of ps_core.hip (one translation unit).
#pragma once

// y = A x, one thread per row (rows of a block-structured Jacobian are short: a few parameter blocks)
__global__ __launch_bounds__(256) void k_sp_spmv(int nrows, const int32_t* __restrict__ rp, const int32_t* __restrict__ ci,
                                                 const double* __restrict__ v, const double* __restrict__ x,
                                                 double* __restrict__ y, double scale)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows) return;
    double s = 0.0;
    for (int k = rp[i]; k < rp[i + 1]; ++k) s += v[k] * x[ci[k]];
    y[i] = scale * s;
}

// Jacobi preconditioner: Minv_j = 1 / sum_k Jt[j][k]^2 (1 where the column is empty: the unknown is then free and stays 0)
__global__ __launch_bounds__(256) void k_sp_diag(int n, const int32_t* __restrict__ rp, const double* __restrict__ v,
                                                 double* __restrict__ Minv)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    double s = 0.0;
    for (int k = rp[j]; k < rp[j + 1]; ++k) s += v[k] * v[k];
    Minv[j] = s > 0.0 ? 1.0 / s : 1.0;
}

// scalars: [0] rz, [1] rz_prev, [2] rz0, [3] pq, [4] done flag (as double), [5] iterations
enum { SPS_RZ = 0, SPS_RZPREV = 1, SPS_RZ0 = 2, SPS_PQ = 3, SPS_DONE = 4, SPS_ITERS = 5, SPS_N = 8 };

// r = rhs (x = 0), z = Minv r, partials of r.z
__global__ __launch_bounds__(256) void k_sp_init(int n, const double* __restrict__ rhs, const double* __restrict__ Minv,
                                                 double* __restrict__ x, double* __restrict__ r, double* __restrict__ z,
                                                 double* __restrict__ p, double* __restrict__ part)
{
    __shared__ double lds[16];
    double acc = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double ri = rhs[i], zi = Minv[i] * ri;
        x[i] = 0.0; r[i] = ri; z[i] = zi; p[i] = 0.0;
        acc += ri * zi;
    }
    acc = block_sum(acc, lds);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

// one block: fold the partials into the scalars, decide convergence (relative preconditioned residual)
__global__ __launch_bounds__(256) void k_sp_scalars(int nb, const double* __restrict__ part, double* __restrict__ sc,
                                                    int slot, double tol2, int first)
{
    __shared__ double lds[16];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) acc += part[i];
    acc = block_sum(acc, lds);
    if (threadIdx.x != 0) return;
    if (slot == SPS_RZ) {
        if (!first && sc[SPS_DONE] != 0.0) return;                  // (launches past convergence are no-ops)
        sc[SPS_RZPREV] = first ? 0.0 : sc[SPS_RZ];
        sc[SPS_RZ] = acc;
        if (first) { sc[SPS_RZ0] = acc; sc[SPS_DONE] = 0.0; sc[SPS_ITERS] = 0.0; }
        else sc[SPS_ITERS] += 1.0;
        if (!(acc > tol2 * sc[SPS_RZ0])) sc[SPS_DONE] = (acc == acc) ? 1.0 : 2.0;      // NaN: breakdown
    } else {
        sc[SPS_PQ] = acc;
        if (!(acc > 0.0) && sc[SPS_DONE] == 0.0) sc[SPS_DONE] = 2.0;                   // p . J^T J p <= 0: breakdown
    }
}

// p = z + beta p (beta = rz / rz_prev; first iteration: p = z)
__global__ __launch_bounds__(256) void k_sp_dir(int n, const double* __restrict__ z, double* __restrict__ p,
                                                const double* __restrict__ sc)
{
    if (sc[SPS_DONE] != 0.0) return;
    const double beta = sc[SPS_RZPREV] > 0.0 ? sc[SPS_RZ] / sc[SPS_RZPREV] : 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = z[i] + beta * p[i];
}

// partials of p . q
__global__ __launch_bounds__(256) void k_sp_dot(int n, const double* __restrict__ a, const double* __restrict__ b,
                                                double* __restrict__ part)
{
    __shared__ double lds[16];
    double acc = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) acc += a[i] * b[i];
    acc = block_sum(acc, lds);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

// alpha = rz / pq; x += alpha p; r -= alpha q; z = Minv r; partials of r . z
__global__ __launch_bounds__(256) void k_sp_update(int n, const double* __restrict__ p, const double* __restrict__ q,
                                                   const double* __restrict__ Minv, double* __restrict__ x,
                                                   double* __restrict__ r, double* __restrict__ z,
                                                   const double* __restrict__ sc, double* __restrict__ part)
{
    __shared__ double lds[16];
    if (sc[SPS_DONE] != 0.0) { if (threadIdx.x == 0) part[blockIdx.x] = 0.0; return; }
    const double alpha = sc[SPS_RZ] / sc[SPS_PQ];
    double acc = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        x[i] += alpha * p[i];
        const double ri = r[i] - alpha * q[i], zi = Minv[i] * ri;
        r[i] = ri; z[i] = zi;
        acc += ri * zi;
    }
    acc = block_sum(acc, lds);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

// ---------------------------------------------------------------------------
// Round 5: a DIRECT solve for the generic path between the single-workgroup dense Cholesky (n <= 2 048) and the CG above:
// up to PS_SPD_MAXN unknowns the normal matrix H = J^T J is formed DENSE on the device from the two CSR copies of J, factored
// by the blocked multi-workgroup Cholesky of the coarse level (k_bchol_panel / k_bchol_update, csrc/ps_k_coarse.h: 24 columns
// per step, the whole chip on the trailing update) and solved by block substitutions with the steps' inverted diagonal tiles;
// the residual of the ORIGINAL system then drives refinement steps through the same factor.  The reference factors such systems
// with SuperLU (pyslam/problem.py:186); Jacobi-preconditioned CG on an ill-conditioned J^T J (a stiff chain: condition number
// 1e15) ran out of iterations where that LU simply solves.
// ---------------------------------------------------------------------------
#define PS_SPD_MAXN 8192

// H[i][j] = sum_k J[k][i] J[k][j]: the rows k of column i come from J^T's CSR row i, the entry (k, j) by binary search in J's
// (column-sorted) row k -- a fixed summation order, no atomics.  One thread per entry of the lower triangle, mirrored.
__global__ __launch_bounds__(256) void k_spd_normal(int n, const int32_t* __restrict__ jrp, const int32_t* __restrict__ jci,
                                                    const double* __restrict__ jv, const int32_t* __restrict__ trp,
                                                    const int32_t* __restrict__ tci, const double* __restrict__ tv,
                                                    double* __restrict__ H)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)n * n) return;
    const int i = (int)(t / n), j = (int)(t % n);
    if (j > i) return;
    double s = 0.0;
    for (int q = trp[i]; q < trp[i + 1]; ++q) {
        const int k = tci[q];
        int lo = jrp[k], hi = jrp[k + 1];
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (jci[mid] < j) lo = mid + 1; else hi = mid; }
        if (lo < jrp[k + 1] && jci[lo] == j) s += tv[q] * jv[lo];
    }
    H[(size_t)i * n + j] = s;
    H[(size_t)j * n + i] = s;
}

// x <- L^-1 x (trans 0) or L^-T x (trans 1) with L = the blocked factor k_bchol_* left in A (below the 24 x 24 diagonal tiles,
// whose INVERSES are in Tinv, one per step): block substitution, one workgroup of 8 waves, three rows (columns) per wave.
__global__ __launch_bounds__(512) void k_spd_subst(int n, const double* __restrict__ A, const double* __restrict__ Tinv,
                                                   double* __restrict__ x, int trans)
{
    constexpr int W = PS_BC_W;
    __shared__ double st[W], sx[W];
    const int t = threadIdx.x, wv = t >> 6, lane = t & 63;
    const int nsteps = (n + W - 1) / W;
    for (int sidx = 0; sidx < nsteps; ++sidx) {
        const int s2 = trans ? nsteps - 1 - sidx : sidx;
        const int j0 = s2 * W, w = min(W, n - j0);
        // t_J = x_J - (what the blocks already solved contribute)
        for (int rr = wv; rr < w; rr += 8) {
            double acc = 0.0;
            if (!trans) {
                const double* row = A + (size_t)(j0 + rr) * n;        // L[j0 + rr][0 .. j0): contiguous
                for (int k = lane; k < j0; k += 64) acc += row[k] * x[k];
            } else {
                for (int k = j0 + w + lane; k < n; k += 64) acc += A[(size_t)k * n + j0 + rr] * x[k];   // L[k][j0 + rr], k below the tile
            }
            acc = wave_sum(acc);
            if (lane == 0) st[rr] = x[j0 + rr] - acc;
        }
        __syncthreads();
        if (t < w) {                                                  // x_J = Tinv t_J (lower triangular) or Tinv^T t_J
            const double* Ti = Tinv + (size_t)s2 * W * W;
            double v = 0.0;
            if (!trans) for (int k = 0; k <= t; ++k) v += Ti[t * W + k] * st[k];
            else for (int k = t; k < w; ++k) v += Ti[k * W + t] * st[k];
            sx[t] = v;
        }
        __syncthreads();
        if (t < w) x[j0 + t] = sx[t];
        __threadfence_block();
        __syncthreads();
    }
}

// r = b - H x (H dense symmetric, one wave per row); partial sums of r.r and b.b per workgroup (4 rows)
__global__ __launch_bounds__(256) void k_spd_residual(int n, const double* __restrict__ H, const double* __restrict__ b,
                                                      const double* __restrict__ x, double* __restrict__ r,
                                                      double* __restrict__ part /* [2 * gridDim] */)
{
    __shared__ double s1[4], s2[4];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, i = blockIdx.x * 4 + wv;
    double ri = 0.0, bi = 0.0;
    if (i < n) {
        const double* row = H + (size_t)i * n;
        double acc = 0.0;
        for (int k = lane; k < n; k += 64) acc += row[k] * x[k];
        acc = wave_sum(acc);
        bi = b[i]; ri = bi - acc;
        if (lane == 0) r[i] = ri;
    }
    if (lane == 0) { s1[wv] = ri * ri; s2[wv] = bi * bi; }
    __syncthreads();
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = (s1[0] + s1[1]) + (s1[2] + s1[3]); part[2 * blockIdx.x + 1] = (s2[0] + s2[1]) + (s2[2] + s2[3]); }
}

__global__ __launch_bounds__(256) void k_spd_axpy(int n, const double* __restrict__ d, double* __restrict__ x) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] += d[i];
}
