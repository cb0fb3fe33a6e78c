// ps_k_xcg.h -- explicit two-level PCG for large reduced systems.
// Part of ps_kernels.h (included from there, in this order; not a stand-alone header).
#pragma once

// ---------------------------------------------------------------------------
// Explicit two-level PCG for long sparse chains (pose graphs with thousands of poses).  Same
// preconditioner as the folded form, M^-1 = I + P A_c^-1 P^T in the scaled coordinates, but APPLIED:
//   k_xcg_spmv      beta, p = z + beta p (on the fly, also for the neighbours), q = S^ p, partials of p.q
//   k_xcg_restrict  alpha, r -= alpha q, x += alpha p (owner node), t_q = sum_i w(i,q) B_i^T r_i
//   k_xcg_coarse    y = A_c^-1 t                      (dense nc x nc matrix-vector product, one wave per row)
//   k_xcg_prolong   z_i = r_i + B_i (w0 y[n] + w1 y[n+1]), partials of r.z
// Four small launches per iteration and 288 B x nnzb of matrix traffic, instead of dragging a dense
// border of ncb blocks through every row (C2: 60 -> 11 blocks per row, 100 -> ~28 us per iteration).
// xstate: [1] threshold, [2] r0.z0, [4 + parity] r.z of iteration k (double-buffered by parity)
// ---------------------------------------------------------------------------
// Three-launch form (default when every SpMV workgroup touches at most PS_XCG_NSLOT coarse nodes): the restriction
// is linear, so t = P^T r follows the recurrence t -= alpha P^T q, and P^T q falls out of the SpMV's epilogue:
//   k_xcg_spmv<.., RT>   ... + per-workgroup pieces of P^T q (one record per (workgroup, node), a node's records contiguous)
//   k_xcg_coarse_rt      alpha, t = t_old - alpha sum(records) (every workgroup, in LDS; workgroup 0 keeps it), y = A_c^-1 t
//   k_xcg_prolong_rt     alpha, r -= alpha q, x += alpha p, z = r + P y, partials of r.z
// t is a recurrence beside r: rounding lets it drift from P^T r by ~1e-16 |t_0| -- that perturbs only what the
// preconditioner is applied TO (r.z and the residual test use the true r and z), not the solution.
#define PS_XCG_ROWS 4                         // rows (waves) per workgroup of the SpMV, four-launch form
#define PS_XCG_ROWS_RT 8                      // ... three-launch form (4 / 8 / 16 measured: C2 4.40 / 4.29 / 4.44 ms, C4 2.20 / 2.15 / 2.22)
#define PS_XCG_NSLOT 4                        // coarse nodes one SpMV workgroup may touch in the three-launch form
#define PS_XCG_CROWS 8                        // rows of A_c^-1 per workgroup of k_xcg_coarse_rt
#ifndef PS_XCG_RBATCH
#define PS_XCG_RBATCH 6                       // records of a node loaded together by k_xcg_coarse_rt
#endif
#define PS_XCG_DROWS 256                      // rows per workgroup of the prolongation
#define PS_XCG_MAXNODES 1024                  // coarse nodes of the explicit form (t of the big coarse kernel in LDS: 48 KB)

PS_DEV double xcg_total(const double* __restrict__ part, int n, double* lds) {
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) v += part[i];
    return block_sum(v, lds);
}

struct XcgRestrictArgs {                      // three-launch form only (RT)
    const double* Bmat; const int32_t* pnode; const double* pw0; const double* pw1;
    const int32_t* wg_out;                    // [workgroup][PS_XCG_NSLOT]: record index of node (first node of the workgroup + slot), -1 none
    double* tq_part;                          // records of D doubles
};

// (Requesting a row's first sixteen blocks and their vector entries ahead of the workgroup-wide sum of r.z was measured:
// 106 VGPRs instead of 58 halve the occupancy and the kernel goes from 11 to 16 us at C2 and C4 -- all 10 000 waves
// resident at once hide the dependent loads better than a shorter chain in fewer waves.)
template <int D, int ROWS, bool RT>
__global__ __launch_bounds__(64 * ROWS) void k_xcg_spmv(
    int nr, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col_idx, int wf,
    const double* __restrict__ S, const double* __restrict__ z, const double* __restrict__ p_old,
    double* __restrict__ p_new, double* __restrict__ q, const double* __restrict__ rz_part, int n_rz,
    double* __restrict__ pq_part, double* __restrict__ xstate, int k, double tol2,
    double* __restrict__ hist, int32_t* __restrict__ status, double* __restrict__ scalars, XcgRestrictArgs ra)
{
    __shared__ double lds[16];
    __shared__ double wpq[ROWS];
    __shared__ double cw[RT ? ROWS : 1][PS_XCG_NSLOT][D];
    constexpr int DD = D * D;
    const int done = status[ST_PCG_DONE];
    // r.z of the previous iteration sits in the slot of the other parity: workgroup 0 of THIS launch writes
    // this iteration's slot while later workgroups may still be starting
    const double rz_prev = xstate[4 + ((k + 1) & 1)], thresh_in = xstate[1];
    double rz = xcg_total(rz_part, n_rz, lds);
    if (done) return;
    const double thresh = (k == 0) ? tol2 * rz : thresh_in;
    const bool first = blockIdx.x == 0 && threadIdx.x == 0;
    if (!(rz > thresh)) {                                 // converged (or rz == 0 / NaN)
        if (first) { status[ST_PCG_DONE] = (rz != rz) ? 2 : 1; scalars[SC_RRFINAL] = rz; if (k == 0) scalars[SC_RR0] = rz; }
        return;
    }
    const double beta = (k == 0) ? 0.0 : rz / rz_prev;
    if (first) {
        xstate[4 + (k & 1)] = rz; hist[k] = rz; status[ST_PCG_ITERS] = k + 1; scalars[SC_RRFINAL] = rz;
        if (k == 0) { xstate[1] = thresh; xstate[2] = rz; scalars[SC_RR0] = rz; }
    }
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * ROWS + w;
    const int kk = lane >> 3, r = lane & 7;
    double pq = 0.0;
    // RT: the row's basis block (one entry per lane), hat weights, nodes -- requested here, first USED after the row's
    // product (a difference of the two node numbers taken here would wait for them before the first matrix load)
    double bl = 0.0, rw0 = 0.0, rw1 = 0.0;
    int prow = 0, pfirst = 0, rout = -1;
    constexpr bool rt = RT;
    if (rt) {
        if (threadIdx.x < PS_XCG_NSLOT * D) rout = ra.wg_out[blockIdx.x * PS_XCG_NSLOT + threadIdx.x / D];
        if (lane < PS_XCG_NSLOT * D) (&cw[w][0][0])[lane] = 0.0;
        if (row < nr) {
            // lane 8 m + a holds B[a][m]: column m of the basis block in one aligned group of eight lanes
            if (r < D && kk < D) bl = ra.Bmat[(size_t)row * DD + r * D + kk];
            const int urow = __builtin_amdgcn_readfirstlane(row);      // wave-uniform: scalar loads, no VGPRs held across the row loop
            prow = ra.pnode[urow]; pfirst = ra.pnode[blockIdx.x * ROWS]; rw0 = ra.pw0[urow]; rw1 = ra.pw1[urow];
        }
    }
    if (row < nr) {
        const int rbeg = wf > 0 ? row * wf : row_ptr[row];
        const int rend = wf > 0 ? rbeg + wf : row_ptr[row + 1];
        double acc = 0.0;
        if (r < D) {
            for (int b = rbeg + kk; b < rend; b += 8) {
                const size_t j = (size_t)col_idx[b] * D;
                const double* sb = S + (size_t)b * DD + r * D;
#pragma unroll
                for (int c = 0; c < D; ++c) acc += sb[c] * (z[j + c] + beta * p_old[j + c]);
            }
        }
        acc += __shfl_xor(acc, 8, 64);
        acc += __shfl_xor(acc, 16, 64);
        acc += __shfl_xor(acc, 32, 64);
        double pn = 0.0;
        if (lane < D) {
            const size_t i = (size_t)row * D + lane;
            pn = z[i] + beta * p_old[i];
            p_new[i] = pn; q[i] = acc;
        }
        pq = wave_sum(lane < D ? pn * acc : 0.0);
        if (rt) {                                           // c = B_i^T q_i, weighted into the row's two nodes
            double v = bl * acc;                            // B[a][m] q[a] (after the xor-shuffles every lane 8 kk + a holds q[a]); 0 where a >= D or m >= D
            v = dpp_shift_add<0x111, 0xf, 0xf>(v);          // row_shr:1
            v = dpp_shift_add<0x112, 0xf, 0xf>(v);          // row_shr:2
            v = dpp_shift_add<0x114, 0xf, 0xa>(v);          // row_shr:4 into banks 1 and 3: lane 8 m + 7 = c[m]
            const int rslot = prow - pfirst;
            if (r == 7 && kk < D) {
                cw[w][rslot][kk] = rw0 * v;
                if (rslot + 1 < PS_XCG_NSLOT) cw[w][rslot + 1][kk] = rw1 * v;
            }
        }
    }
    if (lane == 0) wpq[w] = pq;
    __syncthreads();
    if (threadIdx.x == 0) {
        double v = 0.0;
#pragma unroll
        for (int ww = 0; ww < ROWS; ++ww) v += wpq[ww];
        pq_part[blockIdx.x] = v;
    }
    if (rt && threadIdx.x < PS_XCG_NSLOT * D) {
        const int o = rout;
        if (o >= 0) {
            double v = 0.0;
#pragma unroll
            for (int ww = 0; ww < ROWS; ++ww) v += (&cw[ww][0][0])[threadIdx.x];
            ra.tq_part[(size_t)o * D + threadIdx.x % D] = v;
        }
    }
}

// one workgroup per coarse node q: the rows of its support (two hat intervals)
template <int D>
__global__ __launch_bounds__(256) void k_xcg_restrict(
    int nr, int ncb, const int32_t* __restrict__ slo, const int32_t* __restrict__ shi,
    const int32_t* __restrict__ pnode, const double* __restrict__ pw0, const double* __restrict__ pw1,
    const double* __restrict__ Bmat, const double* __restrict__ r_old, double* __restrict__ r_new,
    const double* __restrict__ qv, const double* __restrict__ p, double* __restrict__ x,
    const double* __restrict__ pq_part, int n_pq, const double* __restrict__ xstate, int k /* < 0: initialisation */,
    double* __restrict__ tvec, const int32_t* __restrict__ status)
{
    __shared__ double lds[16];
    __shared__ double wt[4][8];
    const int done = status[ST_PCG_DONE];
    const int init = k < 0;
    double alpha = 0.0;
    if (!init) {
        const double pq = xcg_total(pq_part, n_pq, lds);
        alpha = xstate[4 + (k & 1)] / pq;
    }
    if (done) return;
    const int qn = blockIdx.x, t = threadIdx.x;
    double acc[D];
#pragma unroll
    for (int c = 0; c < D; ++c) acc[c] = 0.0;
    for (int i = slo[qn] + t; i < shi[qn]; i += 256) {
        const bool owner = pnode[i] == qn;                 // every row has exactly one left node
        const double wgt = (pnode[i] == qn) ? pw0[i] : pw1[i];
        double rn[D];
#pragma unroll
        for (int c = 0; c < D; ++c) {
            const size_t e = (size_t)i * D + c;
            rn[c] = init ? r_old[e] : r_old[e] - alpha * qv[e];
            if (owner) {
                r_new[e] = rn[c];
                if (!init) x[e] += alpha * p[e];
            }
        }
        const double* B = Bmat + (size_t)i * D * D;
#pragma unroll
        for (int c = 0; c < D; ++c) {
            double v = 0.0;
#pragma unroll
            for (int a = 0; a < D; ++a) v += B[a * D + c] * rn[a];
            acc[c] += wgt * v;
        }
    }
    const int wv = t >> 6, lane = t & 63;
#pragma unroll
    for (int c = 0; c < D; ++c) {
        const double v = wave_sum(acc[c]);
        if (lane == 0) wt[wv][c] = v;
    }
    __syncthreads();
    if (t < D) tvec[(size_t)qn * D + t] = ((wt[0][t] + wt[1][t]) + wt[2][t]) + wt[3][t];
}

// A_c^-1 = Lci^T Lci, dense and symmetric, formed once per solve so that the per-iteration coarse solve
// is ONE parallel matrix-vector product (a single workgroup walking two triangular factors with dependent
// L2 loads took ~70 us per iteration).  One workgroup per 64 x 64 tile of the lower triangle (mirrored on
// store), 4 x 4 outputs per thread, rows of Lci staged through LDS 16 at a time.
#define PS_AI_T 64
#define PS_AI_K 16
__global__ __launch_bounds__(256) void k_xcg_ainv(int nc, const double* __restrict__ Lci, float* __restrict__ Ainv,
                                                  int ld /* leading dimension of Ainv (>= nc) */)
{
    __shared__ double As[PS_AI_K][PS_AI_T + 4];
    __shared__ double Bs[PS_AI_K][PS_AI_T + 4];
    // tile (ti >= tj) from the linear index
    int ti = (int)((sqrt(8.0 * blockIdx.x + 1.0) - 1.0) * 0.5);
    while ((ti + 1) * (ti + 2) / 2 <= (int)blockIdx.x) ++ti;
    while (ti * (ti + 1) / 2 > (int)blockIdx.x) --ti;
    const int tj = blockIdx.x - ti * (ti + 1) / 2;
    const int i0 = ti * PS_AI_T, j0 = tj * PS_AI_T;
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
    for (int k0 = i0; k0 < nc; k0 += PS_AI_K) {            // Lci[k][i] = 0 for k < i, and i >= i0 >= j
#pragma unroll
        for (int e = t; e < PS_AI_K * PS_AI_T; e += 256) {
            const int kk = e >> 6, c = e & 63, k = k0 + kk;
            const int ia = i0 + c, jb = j0 + c;
            As[kk][c] = (k < nc && ia < nc && k >= ia) ? Lci[(size_t)k * nc + ia] : 0.0;
            Bs[kk][c] = (k < nc && jb < nc && k >= jb) ? Lci[(size_t)k * nc + jb] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < PS_AI_K; ++kk) {
            double av[4], bv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) { av[a] = As[kk][ty * 4 + a]; bv[a] = Bs[kk][tx * 4 + a]; }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] += av[a] * bv[b];
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int i = i0 + ty * 4 + a, j = j0 + tx * 4 + b;
            if (i < nc && j < nc) {
                const float v = (float)acc[a][b];          // (both triangles get the SAME rounded value: still symmetric)
                if (ti != tj || i >= j) { Ainv[(size_t)i * ld + j] = v; Ainv[(size_t)j * ld + i] = v; }
            }
        }
}

// y = A_c^-1 t : one wave per row.  The inverse is kept in fp32 -- it only preconditions (any symmetric positive
// definite approximation keeps the CG exact), and this product is bound by reading it (19 -> 9.4 MB at nc = 1536).
__global__ __launch_bounds__(256) void k_xcg_coarse(
    int nc, const float* __restrict__ Ainv, const double* __restrict__ tvec,
    double* __restrict__ y, int32_t* __restrict__ status, const int32_t* __restrict__ lag_status)
{
    if (lag_status && blockIdx.x == 0 && threadIdx.x == 0 && lag_status[ST_DIAG_FAIL]) atomicAdd(&status[ST_DIAG_FAIL], 1);
    if (status[ST_PCG_DONE]) return;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= nc) return;
    const float* a = Ainv + (size_t)row * nc;
    double v = 0.0;
    if ((nc & 1) == 0) {
        for (int j = 2 * lane; j < nc; j += 128) {
            const float2 f = *reinterpret_cast<const float2*>(a + j);
            v += (double)f.x * tvec[j] + (double)f.y * tvec[j + 1];
        }
    } else {
        for (int j = lane; j < nc; j += 64) v += (double)a[j] * tvec[j];
    }
    v = wave_sum(v);
    if (lane == 0) y[row] = v;
}

template <int D>
__global__ __launch_bounds__(PS_XCG_DROWS) void k_xcg_prolong(
    int nr, int ncb, const int32_t* __restrict__ pnode, const double* __restrict__ pw0, const double* __restrict__ pw1,
    const double* __restrict__ Bmat, const double* __restrict__ r, const double* __restrict__ y,
    double* __restrict__ z, double* __restrict__ rz_part, const int32_t* __restrict__ status)
{
    __shared__ double lds[16];
    if (status[ST_PCG_DONE]) return;
    const int i = blockIdx.x * PS_XCG_DROWS + threadIdx.x;
    double rz = 0.0;
    if (i < nr) {
        const int n0 = pnode[i];
        const double w0 = pw0[i], w1 = pw1[i];
        double yy[D];
#pragma unroll
        for (int m = 0; m < D; ++m) yy[m] = w0 * y[n0 * D + m] + ((n0 + 1 < ncb) ? w1 * y[(n0 + 1) * D + m] : 0.0);
        const double* B = Bmat + (size_t)i * D * D;
#pragma unroll
        for (int a = 0; a < D; ++a) {
            const size_t e = (size_t)i * D + a;
            double v = r[e];
#pragma unroll
            for (int m = 0; m < D; ++m) v += B[a * D + m] * yy[m];
            z[e] = v;
            rz += r[e] * v;
        }
    }
    rz = block_sum(rz, lds);
    if (threadIdx.x == 0) rz_part[blockIdx.x] = rz;
}

// three-launch form: alpha, t = t_old - alpha P^T q from the SpMV's records (every workgroup forms all of t in LDS --
// a few records per entry, L2-resident), then CROWS rows of y = A_c^-1 t, one wave per row
template <int D>
__global__ __launch_bounds__(64 * PS_XCG_CROWS) void k_xcg_coarse_rt(
    int nc, const float* __restrict__ Ainv, const double* __restrict__ t_old, double* __restrict__ t_new,
    const int32_t* __restrict__ nptr, const double* __restrict__ tq_part, const double* __restrict__ pq_part, int n_pq,
    const double* __restrict__ xstate, int k, double* __restrict__ y, const int32_t* __restrict__ status)
{
    constexpr int NT = 64 * PS_XCG_CROWS, NE = (256 * D + NT - 1) / NT, NA = (256 * D + 127) / 128;
    __shared__ double lds[16];
    __shared__ double tl[256 * D];
    // every load that does not depend on another one is issued up front: two memory round trips, not five
    const int done = status[ST_PCG_DONE];
    const double rz = xstate[4 + (k & 1)];
    double pqv = 0.0;
    for (int i = threadIdx.x; i < n_pq; i += NT) pqv += pq_part[i];
    const int row = blockIdx.x * PS_XCG_CROWS + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const bool even = (nc & 1) == 0;
    const float* a = Ainv + (size_t)(row < nc ? row : 0) * nc;
    float2 av[NA];
    if (even) {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int j = 2 * lane + 128 * u;
            av[u] = (j < nc) ? *reinterpret_cast<const float2*>(a + j) : make_float2(0.f, 0.f);
        }
    }
    int lo[NE], hi[NE];
    double to[NE], sq[NE];
#pragma unroll
    for (int u = 0; u < NE; ++u) {
        const int e = threadIdx.x + u * NT;
        const int n = e / D;
        lo[u] = hi[u] = 0; to[u] = 0.0;
        if (e < nc) { lo[u] = nptr[n]; hi[u] = nptr[n + 1]; to[u] = t_old[e]; }
    }
    // a node's records in batches of PS_XCG_RBATCH independent loads (a plain loop waits out one memory latency per record)
#pragma unroll
    for (int u = 0; u < NE; ++u) sq[u] = 0.0;
    for (int base = 0;; base += PS_XCG_RBATCH) {
        bool any = false;
        double rec[NE][PS_XCG_RBATCH];
#pragma unroll
        for (int u = 0; u < NE; ++u) {
            const int m = (threadIdx.x + u * NT) % D;
#pragma unroll
            for (int c = 0; c < PS_XCG_RBATCH; ++c) {
                const int j = lo[u] + base + c;
                rec[u][c] = (j < hi[u]) ? tq_part[(size_t)j * D + m] : 0.0;
            }
            any = any || (lo[u] + base + PS_XCG_RBATCH < hi[u]);
        }
#pragma unroll
        for (int u = 0; u < NE; ++u)
#pragma unroll
            for (int c = 0; c < PS_XCG_RBATCH; ++c) sq[u] += rec[u][c];
        if (!__syncthreads_or(any)) break;
    }
    const double pq = block_sum(pqv, lds);
    if (done) return;
    const double alpha = rz / pq;
#pragma unroll
    for (int u = 0; u < NE; ++u) {
        const int e = threadIdx.x + u * NT;
        if (e < nc) {
            const double tn = to[u] - alpha * sq[u];
            tl[e] = tn;
            if (blockIdx.x == 0) t_new[e] = tn;
        }
    }
    __syncthreads();
    if (row >= nc) return;
    double v = 0.0;
    if (even) {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int j = 2 * lane + 128 * u;
            if (j < nc) v += (double)av[u].x * tl[j] + (double)av[u].y * tl[j + 1];
        }
    } else {
        for (int j = lane; j < nc; j += 64) v += (double)a[j] * tl[j];
    }
    v = wave_sum(v);
    if (lane == 0) y[row] = v;
}

// the same for coarse levels beyond 256 nodes (pose graphs with thousands of poses: the CG iteration count falls with
// the node count, and the banded factorisation makes a fine coarse level affordable): t in dynamic LDS, 16 rows per
// workgroup, plain loops
#define PS_XCG_CROWS_BIG 16
template <int D>
__global__ __launch_bounds__(64 * PS_XCG_CROWS_BIG) void k_xcg_coarse_rt_big(
    int nc, const float* __restrict__ Ainv, const double* __restrict__ t_old, double* __restrict__ t_new,
    const int32_t* __restrict__ nptr, const double* __restrict__ tq_part, const double* __restrict__ pq_part, int n_pq,
    const double* __restrict__ xstate, int k, double* __restrict__ y, const int32_t* __restrict__ status)
{
    constexpr int NT = 64 * PS_XCG_CROWS_BIG;
    __shared__ double lds[16];
    extern __shared__ double tl_dyn[];
    const int done = status[ST_PCG_DONE];
    const double rz = xstate[4 + (k & 1)];
    double pqv = 0.0;
    for (int i = threadIdx.x; i < n_pq; i += NT) pqv += pq_part[i];
    const double pq = block_sum(pqv, lds);
    if (done) return;
    const double alpha = rz / pq;
    for (int e = threadIdx.x; e < nc; e += NT) {
        const int n = e / D, m = e - n * D;
        const int lo = nptr[n], hi = nptr[n + 1];
        double s = 0.0;
        for (int j0 = lo; j0 < hi; j0 += PS_XCG_RBATCH) {
            double rec[PS_XCG_RBATCH];
#pragma unroll
            for (int c = 0; c < PS_XCG_RBATCH; ++c) rec[c] = (j0 + c < hi) ? tq_part[(size_t)(j0 + c) * D + m] : 0.0;
#pragma unroll
            for (int c = 0; c < PS_XCG_RBATCH; ++c) s += rec[c];
        }
        const double tn = t_old[e] - alpha * s;
        tl_dyn[e] = tn;
        if (blockIdx.x == 0) t_new[e] = tn;
    }
    __syncthreads();
    const int row = blockIdx.x * PS_XCG_CROWS_BIG + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= nc) return;
    const float* a = Ainv + (size_t)row * nc;
    double v = 0.0;
    if ((nc & 1) == 0) {
        for (int j = 2 * lane; j < nc; j += 128) {
            const float2 f = *reinterpret_cast<const float2*>(a + j);
            v += (double)f.x * tl_dyn[j] + (double)f.y * tl_dyn[j + 1];
        }
    } else {
        for (int j = lane; j < nc; j += 64) v += (double)a[j] * tl_dyn[j];
    }
    v = wave_sum(v);
    if (lane == 0) y[row] = v;
}

template <int D>
__global__ __launch_bounds__(PS_XCG_DROWS) void k_xcg_prolong_rt(
    int nr, int ncb, const int32_t* __restrict__ pnode, const double* __restrict__ pw0, const double* __restrict__ pw1,
    const double* __restrict__ Bmat, const double* __restrict__ r_old, double* __restrict__ r_new,
    const double* __restrict__ qv, const double* __restrict__ p, double* __restrict__ x,
    const double* __restrict__ pq_part, int n_pq, const double* __restrict__ xstate, int k,
    const double* __restrict__ y, double* __restrict__ z, double* __restrict__ rz_part, const int32_t* __restrict__ status)
{
    __shared__ double lds[16];
    const int done = status[ST_PCG_DONE];
    const double rzk = xstate[4 + (k & 1)];
    double pqv = 0.0;
    for (int i = threadIdx.x; i < n_pq; i += PS_XCG_DROWS) pqv += pq_part[i];
    const int i = blockIdx.x * PS_XCG_DROWS + threadIdx.x;
    const bool live = i < nr;
    // (all of the row's inputs are loaded before the workgroup-wide sum of p.q, whose barriers would otherwise
    // serialise them behind it)
    double ro[D], qo[D], po[D], xo[D], B[D * D], yy[D];
    if (live) {
        const int n0 = pnode[i];
        const double w0 = pw0[i], w1 = pw1[i];
#pragma unroll
        for (int a = 0; a < D; ++a) {
            const size_t e = (size_t)i * D + a;
            ro[a] = r_old[e]; qo[a] = qv[e]; po[a] = p[e]; xo[a] = x[e];
        }
#pragma unroll
        for (int a = 0; a < D * D; ++a) B[a] = Bmat[(size_t)i * D * D + a];
#pragma unroll
        for (int m = 0; m < D; ++m) yy[m] = w0 * y[n0 * D + m] + ((n0 + 1 < ncb) ? w1 * y[(n0 + 1) * D + m] : 0.0);
    }
    const double pq = block_sum(pqv, lds);
    if (done) return;
    const double alpha = rzk / pq;
    double rz = 0.0;
    if (live) {
        double rn[D];
#pragma unroll
        for (int a = 0; a < D; ++a) {
            const size_t e = (size_t)i * D + a;
            rn[a] = ro[a] - alpha * qo[a];
            r_new[e] = rn[a];
            x[e] = xo[a] + alpha * po[a];
        }
#pragma unroll
        for (int a = 0; a < D; ++a) {
            double v = rn[a];
#pragma unroll
            for (int m = 0; m < D; ++m) v += B[a * D + m] * yy[m];
            z[(size_t)i * D + a] = v;
            rz += rn[a] * v;
        }
    }
    rz = block_sum(rz, lds);
    if (threadIdx.x == 0) rz_part[blockIdx.x] = rz;
}

// x = Linv^T x^
template <int D>
__global__ __launch_bounds__(256) void k_cg_unscale(int nr, const double* __restrict__ Linv,
                                                     const double* __restrict__ xh, double* __restrict__ x,
                                                     const int32_t* __restrict__ gate)
{
    if (gate && gate[ST_PCG_DONE] != 1) return;      // (2 = CG breakdown: the host falls back, nothing is applied)
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nr * D) return;
    const int i = t / D, c = t % D;
    double v = 0.0;
#pragma unroll
    for (int a = 0; a < D; ++a) v += Linv[(size_t)i * D * D + a * D + c] * xh[(size_t)i * D + a];
    x[t] = v;
}

// ---------------------------------------------------------------------------
// ONE launch per iteration (round 3; option "xcg_fused"): the same two-level PCG in the single-reduction
// (Chronopoulos-Gear) arrangement, so that everything between two global reductions fits one kernel:
//     u = M^-1 r,  w = S^ u,  gamma = r.u,  delta = w.u            (this launch, for the NEXT iteration)
//     beta = gamma / gamma_old,  alpha = gamma / (delta - beta gamma / alpha_old)     (from the previous launch's partials)
//     p = u + beta p,  s = w + beta s,  x += alpha p,  r -= alpha s
// Launch k (8 rows of S^ per workgroup, one wave per row):
//   0. totals of the gamma / delta partials of launch k-1 -> alpha_k, beta_k, convergence test on gamma_k = r_k . M^-1 r_k;
//   1. t_{k+1} = P^T r_{k+1} by recurrence, ALL of it in every workgroup's LDS: ts_k = P^T w_k + beta ts_{k-1} (P^T w_k from the
//      records of launch k-1's epilogue), t_{k+1} = t_k - alpha ts_k -- as in the three-launch form, one level deeper;
//   2. y = A_c^-1 t_{k+1} for the coarse nodes this workgroup's columns hang on (a contiguous range; fp32 rows from L2);
//   3. for every DISTINCT column j of the workgroup's rows (host list; a band: 88 at C4): s_j = w_j + beta s_j,
//      r_j = r_j - alpha s_j, u_j = r_j + P_j y into LDS -- once per workgroup, not once per row; the owner of row j
//      also stores r, s, u and updates p, x;
//   4. w_i = sum_j S^_ij u_j from LDS (block index -> LDS slot through a uint16 column), partials of r.u and w.u, and
//      this workgroup's records of P^T w.
// r, w, s are double-buffered by launch parity (neighbours read the old ones while owners write the new), so are t, ts,
// the records and the partials.  Launch -1 is the initialisation (alpha = beta = 0: u_0, w_0, gamma_0, delta_0).
// The recurrences can break down (delta - beta gamma / alpha <= 0) where the classic ones cannot: status 2, and the host
// repeats the solve with the three-launch form.
// 24.4 us in three launches -> one launch per iteration at C4.
// ---------------------------------------------------------------------------
// a failed side-stream factorisation of the lagged coarse matrix reaches the solver's status words (k_xcg_coarse does this in
// the other forms)
__global__ void k_lag_status_check(const int32_t* __restrict__ lag_status, int32_t* __restrict__ status)
{
    if (blockIdx.x == 0 && threadIdx.x == 0 && lag_status[ST_DIAG_FAIL]) atomicAdd(&status[ST_DIAG_FAIL], 1);
}

#define PS_XF_ROWS 8
#define PS_XF_CAP 170                         // distinct columns one workgroup may touch
#define PS_XF_NODES 16                        // coarse nodes (contiguous) one workgroup may need y for
#define PS_XF_NEMAX 4                         // coarse entries per thread: nc <= 4 * 512
#define PS_XF_RB 4                            // records of a node requested together

// y = A_c^-1 t for `nrows_y` consecutive rows of the fp32 inverse (phase 2 of k_xcg_fused1 / k_xcg_persist), t in LDS.
// Round 6: a WAVE per row, the lanes along the row -- lane l takes the columns 2 l + 128 q: every load instruction of the wave reads
// 512 contiguous bytes; no segment sums in LDS, one barrier less.  (Rounds 4-5 gave a THREAD 64 consecutive entries of a row.)  The
// phase is a chain of L2 round trips either way -- C4: 42 rows of 600 entries per interior workgroup, 5.2 us of k_xcg_persist's
// 17.6 us per iteration in both forms; three rows of a wave requested together cost 200 B per lane of scratch in kernels that have
// no register to spare -- which is why k_xcg_persist4 keeps its rows of the inverse in registers (ps_k_xcg_persist4.h).
// A lane sums its columns in ascending order, the lanes by the wave's fixed tree: the same sums in all three kernels, run to run.
template <int RB = 1 /* rows of a wave requested together: the phase is a chain of L2 round trips, RB of them in flight */>
PS_DEV void xcg_coarse_rows(const float* __restrict__ Ainv, int nc, int row_first, int nrows_y, const double* __restrict__ tl /* LDS */,
                            double* __restrict__ yl /* LDS */, int wv, int lane, int nwaves)
{
    if ((nc & 1) == 0) {                                     // (rows of an even nc start 8-byte aligned)
        for (int rr0 = wv; rr0 < nrows_y; rr0 += nwaves * RB) {
            const float* ar[RB];
            double v[RB];
#pragma unroll
            for (int b = 0; b < RB; ++b) { ar[b] = Ainv + (size_t)(row_first + min(rr0 + b * nwaves, nrows_y - 1)) * nc; v[b] = 0.0; }
#pragma unroll (RB >= 4 ? 1 : 2)
            for (int j = 2 * lane; j < nc; j += 128) {
                float2 f[RB];
#pragma unroll
                for (int b = 0; b < RB; ++b) f[b] = *reinterpret_cast<const float2*>(ar[b] + j);
                const double t0 = tl[j], t1 = tl[j + 1];
#pragma unroll
                for (int b = 0; b < RB; ++b) v[b] += (double)f[b].x * t0 + (double)f[b].y * t1;
            }
#pragma unroll
            for (int b = 0; b < RB; ++b) {
                const double y = wave_sum(v[b]);
                if (lane == 0 && rr0 + b * nwaves < nrows_y) yl[rr0 + b * nwaves] = y;
            }
        }
        return;
    }
    for (int rr = wv; rr < nrows_y; rr += nwaves) {
        const float* ar = Ainv + (size_t)(row_first + rr) * nc;
        double v = 0.0;
        for (int j = lane; j < nc; j += 64) v += (double)ar[j] * tl[j];
        v = wave_sum(v);
        if (lane == 0) yl[rr] = v;
    }
}

struct XcgFusedArgs {
    const int32_t* cptr; const int32_t* cols;   // per workgroup: its distinct columns (ascending)
    const uint16_t* lidx;                       // per matrix block: slot of its column in the workgroup's list
    const int32_t* nlo; const int32_t* nhi;     // per workgroup: first / last coarse node needed
    const float* Ainv; int nc, ncb;
    const int32_t* pnode; const double* pw0; const double* pw1; const double* Bmat;
    const int32_t* rec_out; int rmax;           // records of P^T w: node q owns slots [q rmax, (q + 1) rmax) (unused ones stay 0); per (workgroup, node slot) its record
    const double* tq_in; double* tq_out;
    const double* t_in; double* t_out; const double* ts_in; double* ts_out;
    const double* gd_in; double* gd_out; int nwg;
    const double* r_in; double* r_out; const double* w_in; double* w_out; const double* s_in; double* s_out;
    double* u; double* p; double* x;
    double* y;                                  // two-launch form: y = A_c^-1 t_{k+1}, written by k_xcg_f2_coarse
};

// PF: blocks per lane of the row's matrix blocks requested at kernel start (8 PF blocks per row): the matrix stream is in
// flight during the scalar / coarse phases instead of behind them (two waves per SIMD: 256 VGPRs to spend)
template <int D, int PF, bool TWO>
__global__ __launch_bounds__(64 * PS_XF_ROWS) void k_xcg_fused1(
    int nr, const int32_t* __restrict__ row_ptr, int wf, const double* __restrict__ S, XcgFusedArgs a, int k, double tol2,
    double* __restrict__ hist, int cap, int32_t* __restrict__ status, double* __restrict__ scalars, double* __restrict__ xstate)
{
    constexpr int NT = 64 * PS_XF_ROWS, DD = D * D;
    constexpr int NCOL = (PS_XF_CAP * D + NT - 1) / NT;          // (column, component) items per thread
    extern __shared__ __attribute__((aligned(16))) double tl[];   // nc: t_{k+1}
    __shared__ double su[PS_XF_CAP * D];                          // u_{k+1} of the workgroup's columns
    __shared__ double sr[PS_XF_ROWS * D], suo[PS_XF_ROWS * D];    // r_{k+1}, u_{k+1} of its own rows
    __shared__ double yl[PS_XF_NODES * D];
    __shared__ double lds[32];
    __shared__ double wred[PS_XF_ROWS][2];
    __shared__ double cw[PS_XF_ROWS][PS_XCG_NSLOT][D];
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, wg = blockIdx.x;
    const int nc = a.nc;
    // ---- 0. everything that does not depend on the scalars is requested first
    const int done = status[ST_PCG_DONE];
    const double g_prev = hist[k > 0 ? k - 1 : 0], a_prev = hist[cap + (k > 0 ? k - 1 : 0)], thresh_in = xstate[1];
    const int row0 = wg * PS_XF_ROWS, row = row0 + wv;
    const int kk = lane >> 3, r = lane & 7;
    // the row's matrix blocks (row r of block rbeg + kk + 8 i) and their LDS slots
    const int rbeg = row < nr ? (wf > 0 ? row * wf : row_ptr[row]) : 0;
    const int rend = row < nr ? (wf > 0 ? rbeg + wf : row_ptr[row + 1]) : 0;
    double gs = 0.0, ds = 0.0;
    if (!TWO && k >= 0) for (int i = tid; i < a.nwg; i += NT) { gs += a.gd_in[i]; ds += a.gd_in[a.nwg + i]; }
    // two-launch form: alpha_k, beta_k and y are what k_xcg_f2_coarse left for this launch
    const double alpha_in = (TWO && k >= 0) ? xstate[6] : 0.0, beta_in = (TWO && k >= 0) ? xstate[7] : 0.0;
    const int c0 = a.cptr[wg], ncols = a.cptr[wg + 1] - c0;
    const int n_lo = a.nlo[wg], nrows_y = (a.nhi[wg] - n_lo + 1) * D;
    double y_in = 0.0;
    if (TWO && tid < nrows_y) y_in = a.y[n_lo * D + tid];
    double rj[NCOL], wj[NCOL], sj[NCOL], Bj[NCOL][D], cw0[NCOL], cw1[NCOL];
    int jj[NCOL], nj[NCOL];
#pragma unroll
    for (int q = 0; q < NCOL; ++q) {
        const int e = tid + q * NT, c = e / D, m = e - c * D;
        jj[q] = -1; nj[q] = 0; rj[q] = wj[q] = sj[q] = cw0[q] = cw1[q] = 0.0;
#pragma unroll
        for (int mm = 0; mm < D; ++mm) Bj[q][mm] = 0.0;
        if (c < ncols) {
            const int j = a.cols[c0 + c];
            jj[q] = j;
            const size_t o = (size_t)j * D + m;
            rj[q] = a.r_in[o]; wj[q] = a.w_in[o]; sj[q] = a.s_in[o];
            nj[q] = a.pnode[j]; cw0[q] = a.pw0[j]; cw1[q] = a.pw1[j];
            const double* B = a.Bmat + (size_t)j * DD + m * D;
#pragma unroll
            for (int mm = 0; mm < D; ++mm) Bj[q][mm] = B[mm];
        }
    }
    // the own rows' u, p, x (one (row, component) per thread of the first PS_XF_ROWS * D)
    double uo = 0.0, po = 0.0, xo = 0.0;
    const bool own_item = tid < PS_XF_ROWS * D && row0 + tid / D < nr;
    if (own_item) { const size_t o = (size_t)row0 * D + tid; uo = a.u[o]; po = a.p[o]; xo = a.x[o]; }
    // t_k, ts_{k-1} and the first batch of the records of P^T w_k, for this thread's coarse entries
    double to[PS_XF_NEMAX], tso[PS_XF_NEMAX], sq[PS_XF_NEMAX];
#pragma unroll
    for (int u = 0; u < (TWO ? 0 : PS_XF_NEMAX); ++u) {
        const int e = tid + u * NT;
        to[u] = tso[u] = sq[u] = 0.0;
        if (e < nc) {
            const int n = e / D, m = e - n * D;
            to[u] = a.t_in[e]; tso[u] = a.ts_in[e];
            double rec[PS_XF_RB];
#pragma unroll
            for (int c = 0; c < PS_XF_RB; ++c) rec[c] = (c < a.rmax) ? a.tq_in[((size_t)n * a.rmax + c) * D + m] : 0.0;
#pragma unroll
            for (int c = 0; c < PS_XF_RB; ++c) sq[u] += rec[c];
        }
    }
    for (int base = PS_XF_RB; !TWO && base < a.rmax; base += PS_XF_RB) {   // (more than PS_XF_RB records per node: pose graphs)
#pragma unroll
        for (int u = 0; u < PS_XF_NEMAX; ++u) {
            const int e = tid + u * NT;
            if (e < nc) {
                const int n = e / D, m = e - n * D;
                double rec[PS_XF_RB];
#pragma unroll
                for (int c = 0; c < PS_XF_RB; ++c) rec[c] = (base + c < a.rmax) ? a.tq_in[((size_t)n * a.rmax + base + c) * D + m] : 0.0;
#pragma unroll
                for (int c = 0; c < PS_XF_RB; ++c) sq[u] += rec[c];
            }
        }
    }
    // (requested LAST: vmcnt retires in order, so everything above can be consumed while this stream is still in flight)
    double sb[PF > 0 ? PF : 1][D];
    int sl[PF > 0 ? PF : 1];
#pragma unroll
    for (int i = 0; i < PF; ++i) {
        const int b = rbeg + kk + 8 * i;
        sl[i] = 0;
#pragma unroll
        for (int c = 0; c < D; ++c) sb[i][c] = 0.0;
        if (r < D && b < rend) {
            sl[i] = (int)a.lidx[b] * D;
            const double* sp = S + (size_t)b * DD + r * D;
#pragma unroll
            for (int c = 0; c < D; ++c) sb[i][c] = sp[c];
        }
    }
    if (!TWO) block_sum2(gs, ds, lds);
    if (done) return;
    double alpha = alpha_in, beta = beta_in;
    if (!TWO && k >= 0) {
        const double gamma = gs, delta = ds;
        const double thresh = (k == 0) ? tol2 * gamma : thresh_in;
        const bool first = wg == 0 && tid == 0;
        if (!(gamma > thresh)) {
            if (first) { status[ST_PCG_DONE] = (gamma != gamma) ? 2 : 1; scalars[SC_RRFINAL] = gamma; if (k == 0) scalars[SC_RR0] = gamma; }
            return;
        }
        beta = (k == 0) ? 0.0 : gamma / g_prev;
        const double denom = (k == 0) ? delta : delta - beta * gamma / a_prev;
        if (!(denom > 0.0)) { if (first) { status[ST_PCG_DONE] = 2; scalars[SC_RRFINAL] = gamma; } return; }
        alpha = gamma / denom;
        if (first) {
            hist[k] = gamma; hist[cap + k] = alpha; status[ST_PCG_ITERS] = k + 1; scalars[SC_RRFINAL] = gamma;
            if (k == 0) { xstate[1] = thresh; xstate[2] = gamma; scalars[SC_RR0] = gamma; }
        }
    }
    if (TWO) {
        if (tid < nrows_y) yl[tid] = y_in;
    } else {
    // ---- 1. t_{k+1} (all of it) into LDS
#pragma unroll
    for (int u = 0; u < PS_XF_NEMAX; ++u) {
        const int e = tid + u * NT;
        if (e < nc) {
            const double ts = sq[u] + beta * tso[u];
            const double tn = to[u] - alpha * ts;
            tl[e] = tn;
            if (wg == 0) { a.t_out[e] = tn; a.ts_out[e] = ts; }
        }
    }
    __syncthreads();
    // ---- 2. y = A_c^-1 t_{k+1} for the nodes n_lo .. n_hi (a wave per row: xcg_coarse_rows)
    xcg_coarse_rows(a.Ainv, nc, n_lo * D, nrows_y, tl, yl, wv, lane, PS_XF_ROWS);
    }
    __syncthreads();
    // ---- 3. the workgroup's columns: s, r, u (+ the owner's stores and p, x)
#pragma unroll
    for (int q = 0; q < NCOL; ++q) {
        const int j = jj[q];
        if (j >= 0) {
            const int e = tid + q * NT, c = e / D, m = e - c * D;
            const double sn = wj[q] + beta * sj[q];
            const double rn = rj[q] - alpha * sn;
            const int n0 = nj[q] - n_lo;
            const bool two = nj[q] + 1 < a.ncb;
            double un = rn;
#pragma unroll
            for (int mm = 0; mm < D; ++mm) {
                const double yy = cw0[q] * yl[n0 * D + mm] + (two ? cw1[q] * yl[(n0 + 1) * D + mm] : 0.0);
                un += Bj[q][mm] * yy;
            }
            su[c * D + m] = un;
            if (j >= row0 && j < row0 + PS_XF_ROWS) {
                const size_t o = (size_t)j * D + m;
                a.r_out[o] = rn; a.s_out[o] = sn; a.u[o] = un;
                sr[(j - row0) * D + m] = rn; suo[(j - row0) * D + m] = un;
            }
        }
    }
    if (own_item) {                                          // p_k = u_k + beta p_{k-1}, x_{k+1} = x_k + alpha p_k (u_k: what this launch found in a.u)
        const size_t o = (size_t)row0 * D + tid;
        const double pn = uo + beta * po;
        a.p[o] = pn; a.x[o] = xo + alpha * pn;
    }
    __syncthreads();
    // ---- 4. w_{k+1} = S^ u_{k+1} for the own rows, partials, records of P^T w
    double bl = 0.0, rw0 = 0.0, rw1 = 0.0;
    int prow = 0, pfirst = 0, rout = -1;
    if (tid < PS_XCG_NSLOT * D) rout = a.rec_out[wg * PS_XCG_NSLOT + tid / D];
    if (lane < PS_XCG_NSLOT * D) (&cw[wv][0][0])[lane] = 0.0;
    double g2 = 0.0, d2 = 0.0;
    if (row < nr) {
        if (r < D && kk < D) bl = a.Bmat[(size_t)row * DD + r * D + kk];
        const int urow = __builtin_amdgcn_readfirstlane(row);
        prow = a.pnode[urow]; pfirst = a.pnode[row0]; rw0 = a.pw0[urow]; rw1 = a.pw1[urow];
        double acc = 0.0;
        if (r < D) {
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const double* uc = su + sl[i];
#pragma unroll
                for (int c = 0; c < D; ++c) acc += sb[i][c] * uc[c];
            }
            for (int b = rbeg + kk + 8 * PF; b < rend; b += 8) {
                const double* uc = su + (int)a.lidx[b] * D;
                const double* sp = S + (size_t)b * DD + r * D;
#pragma unroll
                for (int c = 0; c < D; ++c) acc += sp[c] * uc[c];
            }
        }
        acc += __shfl_xor(acc, 8, 64);
        acc += __shfl_xor(acc, 16, 64);
        acc += __shfl_xor(acc, 32, 64);
        double ru = 0.0, wu = 0.0;
        if (lane < D) {
            a.w_out[(size_t)row * D + lane] = acc;
            const double un = suo[wv * D + lane];
            ru = sr[wv * D + lane] * un; wu = acc * un;
        }
        g2 = wave_sum(ru); d2 = wave_sum(wu);
        double v = bl * acc;                                  // records: c = B_i^T w_i, weighted into the row's two nodes
        v = dpp_shift_add<0x111, 0xf, 0xf>(v);
        v = dpp_shift_add<0x112, 0xf, 0xf>(v);
        v = dpp_shift_add<0x114, 0xf, 0xa>(v);
        const int rslot = prow - pfirst;
        if (r == 7 && kk < D) {
            cw[wv][rslot][kk] = rw0 * v;
            if (rslot + 1 < PS_XCG_NSLOT) cw[wv][rslot + 1][kk] = rw1 * v;
        }
    }
    if (lane == 0) { wred[wv][0] = g2; wred[wv][1] = d2; }
    __syncthreads();
    if (tid == 0) {
        double g = 0.0, d = 0.0;
#pragma unroll
        for (int ww = 0; ww < PS_XF_ROWS; ++ww) { g += wred[ww][0]; d += wred[ww][1]; }
        a.gd_out[wg] = g; a.gd_out[a.nwg + wg] = d;
    }
    if (tid < PS_XCG_NSLOT * D && rout >= 0) {
        double v = 0.0;
#pragma unroll
        for (int ww = 0; ww < PS_XF_ROWS; ++ww) v += (&cw[ww][0][0])[tid];
        a.tq_out[(size_t)rout * D + tid % D] = v;
    }
}

// ---------------------------------------------------------------------------
// TWO launches per iteration: coarse levels too wide for every workgroup of k_xcg_fused1 to form its own rows of y
// (C2: 2 406 coarse unknowns -- 16 x 6 rows of a 2 406-wide inverse per workgroup, 1 250 workgroups: 1.1 GB per iteration,
// measured 14.5 ms against 3.0).  The scalar phase, t_{k+1} and y = A_c^-1 t_{k+1} move into this kernel, computed ONCE
// (16 rows of the inverse per workgroup, every workgroup forming all of t in LDS as k_xcg_coarse_rt_big does), and
// k_xcg_fused1<.., TWO = true> reads alpha, beta (xstate[6], [7]) and its nodes' y.  Same recurrences, same records.
// ---------------------------------------------------------------------------
#define PS_XF2_NEMAX 6                        // coarse entries per thread: nc <= 6 * 1024 (PS_XCG_MAXNODES nodes of SE(3))
// NE: coarse entries per thread (3: nc <= 3 072 -- C2 -- leaves the registers for 16 row loads in flight per lane; 6: up to 6 144, 8 in flight)
template <int D, int NE>
__global__ __launch_bounds__(64 * PS_XCG_CROWS_BIG) void k_xcg_f2_coarse(
    XcgFusedArgs a, int k, double tol2, double* __restrict__ hist, int cap, int32_t* __restrict__ status,
    double* __restrict__ scalars, double* __restrict__ xstate, int rows_per_wg, int ablate /* measurement build: 1 no rows, 2 no records */)
{
    constexpr int NT = 64 * PS_XCG_CROWS_BIG;
    extern __shared__ __attribute__((aligned(16))) double tl[];   // nc: t_{k+1}
    __shared__ double lds[32];
    const int tid = threadIdx.x, wg = blockIdx.x, nc = a.nc;
    // ---- 0. everything that does not depend on the scalars is requested first
    const int done = status[ST_PCG_DONE];
    const double g_prev = hist[k > 0 ? k - 1 : 0], a_prev = hist[cap + (k > 0 ? k - 1 : 0)], thresh_in = xstate[1];
    double gs = 0.0, ds = 0.0;
    if (k >= 0) for (int i = tid; i < a.nwg; i += NT) { gs += a.gd_in[i]; ds += a.gd_in[a.nwg + i]; }
    double to[NE], tso[NE], sq[NE];
#pragma unroll
    for (int u = 0; u < NE; ++u) {
        const int e = tid + u * NT;
        to[u] = tso[u] = sq[u] = 0.0;
        if (e < nc) {
            const int n = e / D, m = e - n * D;
            to[u] = a.t_in[e]; tso[u] = a.ts_in[e];
            double rec[PS_XF_RB];
#pragma unroll
            for (int c = 0; c < PS_XF_RB; ++c) rec[c] = (c < a.rmax && !(ablate & 2)) ? a.tq_in[((size_t)n * a.rmax + c) * D + m] : 0.0;
#pragma unroll
            for (int c = 0; c < PS_XF_RB; ++c) sq[u] += rec[c];
        }
    }
    for (int base = PS_XF_RB; base < a.rmax && !(ablate & 2); base += PS_XF_RB) {
#pragma unroll
        for (int u = 0; u < NE; ++u) {
            const int e = tid + u * NT;
            if (e < nc) {
                const int n = e / D, m = e - n * D;
                double rec[PS_XF_RB];
#pragma unroll
                for (int c = 0; c < PS_XF_RB; ++c) rec[c] = (base + c < a.rmax) ? a.tq_in[((size_t)n * a.rmax + base + c) * D + m] : 0.0;
#pragma unroll
                for (int c = 0; c < PS_XF_RB; ++c) sq[u] += rec[c];
            }
        }
    }
    block_sum2(gs, ds, lds);
    if (done) return;
    double alpha = 0.0, beta = 0.0;
    if (k >= 0) {
        const double gamma = gs, delta = ds;
        const double thresh = (k == 0) ? tol2 * gamma : thresh_in;
        const bool first = wg == 0 && tid == 0;
        if (!(gamma > thresh)) {
            if (first) { status[ST_PCG_DONE] = (gamma != gamma) ? 2 : 1; scalars[SC_RRFINAL] = gamma; if (k == 0) scalars[SC_RR0] = gamma; }
            return;
        }
        beta = (k == 0) ? 0.0 : gamma / g_prev;
        const double denom = (k == 0) ? delta : delta - beta * gamma / a_prev;
        if (!(denom > 0.0)) { if (first) { status[ST_PCG_DONE] = 2; scalars[SC_RRFINAL] = gamma; } return; }
        alpha = gamma / denom;
        if (first) {
            hist[k] = gamma; hist[cap + k] = alpha; status[ST_PCG_ITERS] = k + 1; scalars[SC_RRFINAL] = gamma;
            xstate[6] = alpha; xstate[7] = beta;
            if (k == 0) { xstate[1] = thresh; xstate[2] = gamma; scalars[SC_RR0] = gamma; }
        }
    }
    // ---- 1. t_{k+1} (all of it) into LDS
#pragma unroll
    for (int u = 0; u < NE; ++u) {
        const int e = tid + u * NT;
        if (e < nc) {
            const double ts = sq[u] + beta * tso[u];
            const double tn = to[u] - alpha * ts;
            tl[e] = tn;
            if (wg == 0) { a.t_out[e] = tn; a.ts_out[e] = ts; }
        }
    }
    __syncthreads();
    // ---- 2. y = A_c^-1 t_{k+1}, one wave per row, rows_per_wg = ceil(nc / 256) rows per workgroup: every compute unit has one
    // workgroup and all of them run in ONE round (the skeleton in front of this point costs 8.4 us per round at C2 -- measured with
    // the rows switched off -- so more, smaller workgroups lose: 8 rows each 31 us).  Round 3 read a row 128 columns at a time with
    // 8-byte loads, 3.7 TB/s out of the Infinity Cache; 16-byte loads, all of a lane's share of the row requested before the first
    // multiply: the rows start on 16 bytes or 8 bytes past (nc even), so a row is [head pair] + float4 body + [tail pair].
    const int wv = tid >> 6, lane = tid & 63;
    const int row = wg * rows_per_wg + wv;
    if (wv >= rows_per_wg || row >= nc || (ablate & 1)) return;
    const float* ar = a.Ainv + (size_t)row * nc;
    double v = 0.0;
    if ((nc & 1) == 0) {
        const int head = (int)(((size_t)row * nc) & 3);          // 0 or 2 floats in front of the first 16-byte boundary
        const int nq = (nc - head) >> 2, tail0 = head + 4 * nq;  // float4 body, then 0 or 2 floats
        const float4* ab = reinterpret_cast<const float4*>(ar + head);
        constexpr int U = 12;                                    // 12 x 64 float4 = 3 072 columns per pass
        float2 hp = make_float2(0.f, 0.f);
        if (lane == 0 && head) hp = *reinterpret_cast<const float2*>(ar);
        if (lane == 1 && tail0 < nc) hp = *reinterpret_cast<const float2*>(ar + tail0);
        for (int q0 = lane; q0 < nq; q0 += 64 * U) {
            float4 f[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { const int q = q0 + 64 * u; f[u] = q < nq ? ab[q] : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = min(q0 + 64 * u, nq - 1);          // (beyond the row: f = 0)
                const double2 t0 = *reinterpret_cast<const double2*>(tl + head + 4 * q);
                const double2 t1 = *reinterpret_cast<const double2*>(tl + head + 4 * q + 2);
                v += ((double)f[u].x * t0.x + (double)f[u].y * t0.y) + ((double)f[u].z * t1.x + (double)f[u].w * t1.y);
            }
        }
        if (lane == 0 && head) v += (double)hp.x * tl[0] + (double)hp.y * tl[1];
        if (lane == 1 && tail0 < nc) v += (double)hp.x * tl[tail0] + (double)hp.y * tl[tail0 + 1];
    } else {
        for (int j = lane; j < nc; j += 64) v += (double)ar[j] * tl[j];
    }
    v = wave_sum(v);
    if (lane == 0) a.y[row] = v;
}
