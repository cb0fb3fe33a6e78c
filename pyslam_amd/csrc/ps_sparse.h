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
