// ps_k_coarse.h -- coarse level: row sums, coarse matrix, Cholesky (LDS-resident and blocked), triangular inverse, border, right-hand side, recovery.
// Part of ps_kernels.h (included from there, in this order; not a stand-alone header).
#pragma once

// ---------------------------------------------------------------------------
// Two-level (aggregation) preconditioning, folded into the matrix.
//   coarse basis P: hat functions over the reduced-pose index (ncb nodes), nc = ncb * D
//   A_c = P^T S^ P = L_c L_c^T,  B = P (scaled coordinates)
//   additive two-level M^-1 = I + B A_c^-1 B^T = V V^T,  V = [I, B L_c^-T]
// CG on the augmented, consistent semi-definite system  V^T S^ V x~ = V^T g^,
//        [[S^, K], [K^T, I]],   K = S^ P L_c^-T
// is exactly that PCG (Griebel 1994), so k_cg_fused runs unchanged on a larger BSR.
// Low-frequency trajectory modes (lambda_min(M^-1 S) ~ 6e-4 on the C3 workload) are what
// make block-Jacobi CG take ~100 iterations; the coarse space removes them (~25-30).
// ---------------------------------------------------------------------------

// Coarse space: continuous piecewise-linear "hat" functions over the reduced-pose index, one
// per coarse node and tangent dof, expressed in the SCALED coordinates x^ = L^T x (B = P, the
// interpolation matrix with two weights per pose).  A_c = P^T S^ P inherits the unit block
// diagonal of S^ and stays well conditioned even when block scales differ by 1e12 (priors),
// which keeps the augmented matrix numerically positive semi-definite.  Hats need ~35 % fewer
// coarse unknowns than discontinuous constant+linear aggregates for the same iteration count.
//   pnode[i], pw0[i], pw1[i] : pose i interpolates nodes pnode[i] (weight pw0) and pnode[i]+1 (pw1)
//   slo[q], shi[q]           : poses in the support of node q
PS_DEV double coarse_weight(int j, int q, const int32_t* __restrict__ pnode,
                            const double* __restrict__ pw0, const double* __restrict__ pw1) {
    return (pnode[j] == q) ? pw0[j] : pw1[j];
}

// SZ[i][q] (D x D) = sum_j S^_ij B_j w(j,q) over the contiguous run of row i's blocks whose column
// lies in the support of node q (run_lo / run_hi, precomputed), i.e. (S^ P)_iq with the coarse basis
// P_jq = w(j,q) B_j; and BSZ[i][q] = B_i^T SZ[i][q], the summand of A_c = P^T S^ P.  One workgroup per fine row.
template <int D>
__global__ __launch_bounds__(256) void k_coarse_rowsums(
    int nr, int ncb, const int32_t* __restrict__ run_lo, const int32_t* __restrict__ run_hi,
    const int32_t* __restrict__ acol_idx, const int32_t* __restrict__ pnode,
    const double* __restrict__ pw0, const double* __restrict__ pw1,
    const double* __restrict__ SB /* S^_ij B_j per fine block (augmented-matrix slots) */,
    double* __restrict__ SZ, const double* __restrict__ Bmat, double* __restrict__ BSZ /* B_i^T SZ[i][q] */)
{
    constexpr int DD = D * D;
    extern __shared__ double srow[];                     // ncb x DD: this row's SZ blocks, + DD: B_i
    const int i = blockIdx.x, nslot = ncb * DD;
    double* sBi = srow + nslot;
    if (threadIdx.x < DD) sBi[threadIdx.x] = Bmat[(size_t)i * DD + threadIdx.x];
    for (int t = threadIdx.x; t < nslot; t += blockDim.x) {
        const int q = t / DD, e = t % DD;
        const int k0 = run_lo[i * ncb + q], k1 = run_hi[i * ncb + q];
        double acc = 0.0;
#pragma unroll 4
        for (int k = k0; k < k1; ++k)
            acc += SB[(size_t)k * DD + e] * coarse_weight(acol_idx[k], q, pnode, pw0, pw1);
        SZ[(size_t)i * nslot + t] = acc;
        srow[t] = acc;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < nslot; t += blockDim.x) {
        const int q = t / DD, e = t % DD, r = e / D, c = e % D;
        double acc = 0.0;
#pragma unroll
        for (int m = 0; m < D; ++m) acc += sBi[m * D + r] * srow[q * DD + m * D + c];
        BSZ[(size_t)i * nslot + t] = acc;
    }
}

// A_c[q][q'] (D x D block) = sum_{i in supp(q)} w(i,q) SZ[i][q'] ; dense nc x nc, row-major
template <int D>
__global__ __launch_bounds__(256) void k_coarse_matrix(
    int nr, int ncb, const int32_t* __restrict__ slo, const int32_t* __restrict__ shi,
    const int32_t* __restrict__ pnode, const double* __restrict__ pw0, const double* __restrict__ pw1,
    const double* __restrict__ SZ, double* __restrict__ Ac)
{
    constexpr int DD = D * D;
    const int nc = ncb * D;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ncb * ncb * DD) return;
    const int e = t % DD, q2 = (t / DD) % ncb, q = t / (DD * ncb);
    double acc = 0.0;
#pragma unroll 8
    for (int i = slo[q]; i < shi[q]; ++i)
        acc += coarse_weight(i, q, pnode, pw0, pw1) * SZ[((size_t)i * ncb + q2) * DD + e];
    Ac[(size_t)(q * D + e / D) * nc + q2 * D + e % D] = acc;
}

// The same two steps for the explicit PCG (hundreds of coarse nodes, a row touches a handful of them): only the
// non-empty (row, node) runs exist, as ENTRIES listed per row (ent_ptr / ent_q / ent_lo / ent_hi) -- a dense
// nr x ncb array of 6 x 6 blocks is 723 MB at C2 and clearing that allocation alone costs 30 ms.
//   k_xcoarse_rowsums : BSZ[e] = B_i^T sum_{k in run(e)} S^_ik B_k w(k, q_e)      one workgroup per fine row
//   k_xcoarse_matrix  : A_c[q][q'] = sum over the SEGMENT (q, q') of w(i, q) BSZ[e]  one thread per output entry;
//                       a segment lists the entries (i in supp(q), q_e = q') in row order (host-built, fixed order)
template <int D>
__global__ __launch_bounds__(256) void k_xcoarse_rowsums(
    int nr, const int32_t* __restrict__ ent_ptr, const int32_t* __restrict__ ent_q,
    const int32_t* __restrict__ ent_lo, const int32_t* __restrict__ ent_hi,
    const int32_t* __restrict__ acol_idx, const int32_t* __restrict__ pnode,
    const double* __restrict__ pw0, const double* __restrict__ pw1,
    const double* __restrict__ SB, const double* __restrict__ Bmat, double* __restrict__ BSZ)
{
    constexpr int DD = D * D;
    extern __shared__ double srow[];                     // (entries of this row) x DD, + DD: B_i
    const int i = blockIdx.x, e0 = ent_ptr[i], n = (ent_ptr[i + 1] - e0) * DD;
    double* sBi = srow + n;
    if (threadIdx.x < DD) sBi[threadIdx.x] = Bmat[(size_t)i * DD + threadIdx.x];
    for (int t = threadIdx.x; t < n; t += blockDim.x) {
        const int e = e0 + t / DD, el = t % DD, q = ent_q[e];
        double acc = 0.0;
#pragma unroll 4
        for (int k = ent_lo[e]; k < ent_hi[e]; ++k)
            acc += SB[(size_t)k * DD + el] * coarse_weight(acol_idx[k], q, pnode, pw0, pw1);
        srow[t] = acc;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < n; t += blockDim.x) {
        const int el = t % DD, r = el / D, c = el % D, base = t - el;
        double acc = 0.0;
#pragma unroll
        for (int m = 0; m < D; ++m) acc += sBi[m * D + r] * srow[base + m * D + c];
        BSZ[(size_t)e0 * DD + t] = acc;
    }
}

template <int D>
__global__ __launch_bounds__(256) void k_xcoarse_matrix(
    int ncb, const int32_t* __restrict__ seg_ptr /* ncb * ncb + 1 */, const int32_t* __restrict__ seg_ent,
    const int32_t* __restrict__ seg_row, const int32_t* __restrict__ pnode, const double* __restrict__ pw0,
    const double* __restrict__ pw1, const double* __restrict__ BSZ, double* __restrict__ Ac)
{
    constexpr int DD = D * D;
    const int nc = ncb * D;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ncb * ncb * DD) return;
    const int e = t % DD, q2 = (t / DD) % ncb, q = t / (DD * ncb);
    double acc = 0.0;
    for (int s = seg_ptr[q * ncb + q2]; s < seg_ptr[q * ncb + q2 + 1]; ++s)
        acc += coarse_weight(seg_row[s], q, pnode, pw0, pw1) * BSZ[(size_t)seg_ent[s] * DD + e];
    Ac[(size_t)(q * D + e / D) * nc + q2 * D + e % D] = acc;
}

// A_c = L_c L_c^T and Li = L_c^-1 by ONE workgroup, blocked by D x D (ncb block steps instead of
// nc scalar steps), both matrices full row-major in LDS: 2 nc^2 doubles (nc <= 96).
// Outputs Li and its transpose LiT (row-major, global) so later kernels read either coalesced.
template <int D, bool IN_LDS>
__global__ __launch_bounds__(1024) void k_coarse_chol(int ncb, const double* __restrict__ A,
                                                       double* __restrict__ Li, double* __restrict__ LiT,
                                                       int32_t* __restrict__ status, double* gscratch)
{
    constexpr int DD = D * D;
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int nc = ncb * D;
    // both matrices live in LDS when they fit (nc <= 96); larger coarse levels fall back to a
    // global (L2-resident) scratch -- same code, ~10x slower per step, used for big problems only
    // (compile-time choice: with a run-time pointer select the compiler falls back to flat
    // addressing for every access and the LDS path loses ~40 %)
    double* sL = IN_LDS ? sm : gscratch;                    // nc x nc: A, overwritten by L (lower)
    double* sX = sL + nc * nc;                              // nc x nc: L^-1
    __shared__ double sRl[64 * 6];      // reciprocal diagonal of L
    const int t = threadIdx.x, nt = blockDim.x;
    for (int k = t; k < nc * nc; k += nt) { sL[k] = A[k]; sX[k] = 0.0; }
    // Round 5: the serial part of a block step is the D x D Cholesky ALONE (reciprocal square roots, no division, ~150
    // instructions of one thread); the panel below it is solved row by row against L_JJ^T (one thread per row, in place: no
    // staging pass), and the inverses of the diagonal blocks -- which rounds 1-4 formed inside the serial part, 126 more
    // dependent multiply-adds per step -- are formed for all blocks at once after the loop.  Three barriers per step, were four.
    // ... and it runs AHEAD: wave 0 updates and factors the next diagonal block while the other waves do the rest of the trailing
    // update (as k_band_chol does), so a step is two barriers: panel | trailing update + next diagonal factor.
    auto factor_diag = [&](int J) {                         // one thread: L_JJ in place (upper part zeroed), 1 / diag into sRl
        double a[D][D], rl[D];
        bool ok = true;
#pragma unroll
        for (int r = 0; r < D; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) a[r][c] = sL[(J * D + r) * nc + J * D + c];
#pragma unroll
        for (int jj = 0; jj < D; ++jj) {
            double d = a[jj][jj];
#pragma unroll
            for (int k = 0; k < jj; ++k) d -= a[jj][k] * a[jj][k];
            if (!(d > 0.0)) { ok = false; d = 1.0; }
            rl[jj] = ps_rsqrt(d);
            a[jj][jj] = d * rl[jj];
#pragma unroll
            for (int r = jj + 1; r < D; ++r) {
                double v = a[r][jj];
#pragma unroll
                for (int k = 0; k < jj; ++k) v -= a[r][k] * a[jj][k];
                a[r][jj] = v * rl[jj];
            }
        }
        if (!ok) atomicAdd(&status[ST_DIAG_FAIL], 1);
#pragma unroll
        for (int r = 0; r < D; ++r) {
            sRl[J * D + r] = rl[r];
#pragma unroll
            for (int c = 0; c < D; ++c) sL[(J * D + r) * nc + J * D + c] = c <= r ? a[r][c] : 0.0;
        }
    };
    __syncthreads();
    if (t == 0) factor_diag(0);
    for (int J = 0; J < ncb; ++J) {
        __syncthreads();
        // panel: row x of L_IJ solves x L_JJ^T = (row of A_IJ) -- forward substitution, one thread per row, in place
        const int m = ncb - J - 1;
        for (int row = t; row < m * D; row += nt) {
            double* ar = sL + ((J + 1) * D + row) * nc + J * D;
            double x[D];
#pragma unroll
            for (int c = 0; c < D; ++c) {
                double v = ar[c];
#pragma unroll
                for (int k = 0; k < c; ++k) v -= x[k] * sL[(J * D + c) * nc + J * D + k];
                x[c] = v * sRl[J * D + c];
            }
#pragma unroll
            for (int c = 0; c < D; ++c) ar[c] = x[c];
        }
        __syncthreads();
        if (m == 0) break;
        if (t < 64) {                                       // wave 0: the next diagonal block, updated and factored
            if (t < DD) {
                const int a = t / D, b2 = t % D;
                double v = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) v += sL[((J + 1) * D + a) * nc + J * D + k] * sL[((J + 1) * D + b2) * nc + J * D + k];
                sL[((J + 1) * D + a) * nc + (J + 1) * D + b2] -= v;
            }
            __builtin_amdgcn_wave_barrier();
            if (t == 0) factor_diag(J + 1);
        } else {
            // trailing update A_IK -= L_IJ L_KJ^T for J < K <= I, without the block wave 0 has taken
            for (int idx = t - 64; idx < m * m * DD; idx += nt - 64) {
                const int blk = idx / DD, e = idx % DD, a = e / D, b2 = e % D;
                const int I = J + 1 + blk / m, K = J + 1 + blk % m;
                if (K > I || (I == J + 1 && K == J + 1)) continue;
                double v = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) v += sL[(I * D + a) * nc + J * D + k] * sL[(K * D + b2) * nc + J * D + k];
                sL[(I * D + a) * nc + K * D + b2] -= v;
            }
        }
    }
    // X = L^-1 by RECURSIVE DOUBLING (round 5).  The diagonal blocks first -- column c of L_RR^-1 by forward substitution, one
    // thread per (block, column), all blocks at once --, then groups of s = 1, 2, 4, ... block rows are merged pairwise,
    // [[X11, 0], [X21, X22]] with X21 = -X22 (L21 X11): two small matrix products per level, every entry of every pair at once,
    // two barriers per level -- 8 barriers for 13 block rows.  (Rounds 1-4 went block row by block row, four barriers each.)  The
    // intermediate T = L21 X11 of a pair is kept TRANSPOSED in that pair's (still unused) upper-right block of sX; the upper
    // triangle is dropped on the way out.
    __syncthreads();
    for (int k = t; k < ncb * D; k += nt) {
        const int R = k / D, c = k % D;
        double col[D];
#pragma unroll
        for (int r = 0; r < D; ++r) {
            double v = (r == c) ? 1.0 : 0.0;
#pragma unroll
            for (int q = 0; q < r; ++q) v -= (q >= c) ? sL[(R * D + r) * nc + R * D + q] * col[q] : 0.0;
            col[r] = (r >= c) ? v * sRl[R * D + r] : 0.0;
            sX[(R * D + r) * nc + R * D + c] = col[r];
        }
    }
    // (a thread's entries of the nc x nc array are the same on every level: in the LDS variant -- nc <= 96, at most nine entries
    //  per thread -- their row / column indices are worked out once (divisions by the run-time nc) and kept in registers; a level
    //  only tests bits of the block indices, sg being a power of two.  The out-of-L2 variant for bigger matrices recomputes them.)
    constexpr int NEX = IN_LDS ? 9 : 1;                     // ceil(96^2 / 1024)
    short ei[NEX], ej[NEX];
    if (IN_LDS) {
#pragma unroll
        for (int u = 0; u < NEX; ++u) {
            const int idx = t + u * nt;
            ei[u] = idx < nc * nc ? (short)(idx / nc) : (short)-1;
            ej[u] = idx < nc * nc ? (short)(idx % nc) : (short)0;
        }
    }
    auto t_entry = [&](int i, int jj, int sg, int lg) {      // T[i][j] = sum_{k >= j, k in group 1} L[i][k] X11[k][j], kept transposed
        const int bi = i / D, bj = jj / D;
        if (((bi ^ bj) >> lg) != 0 || !(bi & sg) || (bj & sg)) return;
        const int k1 = (((bj >> lg) << lg) + sg) * D;        // end of group 1 (scalar)
        double v = 0.0;
        for (int k = jj; k < k1; ++k) v += sL[i * nc + k] * sX[k * nc + jj];
        sX[jj * nc + i] = v;
    };
    auto x_entry = [&](int i, int jj, int sg, int lg) {      // X21[i][j] = -sum_{k <= i, k in group 2} X22[i][k] T[k][j]
        const int bi = i / D, bj = jj / D;
        if (((bi ^ bj) >> lg) != 0 || !(bi & sg) || (bj & sg)) return;
        const int k0 = (((bj >> lg) << lg) + sg) * D;        // start of group 2
        double v = 0.0;
        for (int k = k0; k <= i; ++k) v -= sX[i * nc + k] * sX[jj * nc + k];
        sX[i * nc + jj] = v;
    };
    for (int sg = 1, lg = 1; sg < ncb; sg *= 2, ++lg) {    // lg = log2(2 sg)
        __syncthreads();
        if (IN_LDS) {
#pragma unroll
            for (int u = 0; u < NEX; ++u) if (ei[u] >= 0) t_entry(ei[u], ej[u], sg, lg);
        } else for (int idx = t; idx < nc * nc; idx += nt) t_entry(idx / nc, idx % nc, sg, lg);
        __syncthreads();
        if (IN_LDS) {
#pragma unroll
            for (int u = 0; u < NEX; ++u) if (ei[u] >= 0) x_entry(ei[u], ej[u], sg, lg);
        } else for (int idx = t; idx < nc * nc; idx += nt) x_entry(idx / nc, idx % nc, sg, lg);
    }
    __syncthreads();
    for (int k = t; k < nc * nc; k += nt) {
        const int r = k / nc, c = k % nc;
        const double v = (c <= r) ? sX[k] : 0.0;
        Li[k] = v;
        LiT[(size_t)c * nc + r] = v;
    }
}

// ---------------------------------------------------------------------------
// Large coarse matrices (nc > 90: beyond one workgroup's LDS): blocked right-looking Cholesky over the
// whole chip, PS_BC_W columns per step -- k_bchol_panel (one workgroup: diagonal tile factor + its
// inverse + the panel below) and k_bchol_update (one workgroup per 32 x 32 tile of the trailing matrix)
// -- then L^-1 by independent column blocks (k_btri_inverse, one workgroup each, its column block of X
// in LDS).  ~2 ceil(nc / 24) + 1 launches, 0.2-0.4 ms at nc = 294 ... 384 instead of 2.4 ... 8 ms for the
// single-workgroup factorisation out of L2.
// ---------------------------------------------------------------------------
#define PS_BC_W 24
__global__ __launch_bounds__(256) void k_bchol_panel(
    int nc, int j0, double* __restrict__ A /* nc x nc row-major: lower triangle in, L (below the tiles) out */,
    double* __restrict__ Tinv /* PS_BC_W x PS_BC_W: inverse of this step's diagonal factor */,
    int32_t* __restrict__ status)
{
    // Every workgroup factors the (tiny) diagonal tile itself -- 24 sequential steps in LDS, cheaper than a
    // launch boundary -- and then owns a slab of 1024 panel entries, so the panel below the tile is spread over
    // the chip.  The tile's factor itself is never needed again (only its inverse, Tinv), so nobody writes the
    // tile back and the redundant readers do not race with a writer.
    __shared__ double sD[PS_BC_W * PS_BC_W], sI[PS_BC_W * PS_BC_W];
    const int t = threadIdx.x, w = min(PS_BC_W, nc - j0);
    for (int k = t; k < PS_BC_W * PS_BC_W; k += 256) {
        const int r = k / PS_BC_W, c = k % PS_BC_W;
        sD[k] = (r < w && c <= r) ? A[(size_t)(j0 + r) * nc + j0 + c] : 0.0;
        sI[k] = 0.0;
    }
    __syncthreads();
    for (int j = 0; j < w; ++j) {                          // unblocked Cholesky of the w x w tile in LDS
        if (t == 0) {
            const double d = sD[j * PS_BC_W + j];
            if (!(d > 0.0) && blockIdx.x == 0) atomicAdd(&status[ST_DIAG_FAIL], 1);
            sD[j * PS_BC_W + j] = sqrt(d);
        }
        __syncthreads();
        const double inv = 1.0 / sD[j * PS_BC_W + j];
        if (t > j && t < w) sD[t * PS_BC_W + j] *= inv;
        __syncthreads();
        for (int k = t; k < w * w; k += 256) {
            const int r = k / w, c = k % w;
            if (c > j && r >= c) sD[r * PS_BC_W + c] -= sD[r * PS_BC_W + j] * sD[c * PS_BC_W + j];
        }
        __syncthreads();
    }
    if (t < w) {                                           // column t of the inverse by forward substitution
        for (int r = t; r < w; ++r) {
            double v = (r == t) ? 1.0 : 0.0;
            for (int k = t; k < r; ++k) v -= sD[r * PS_BC_W + k] * sI[k * PS_BC_W + t];
            sI[r * PS_BC_W + t] = v / sD[r * PS_BC_W + r];
        }
    }
    __syncthreads();
    if (blockIdx.x == 0)
        for (int k = t; k < PS_BC_W * PS_BC_W; k += 256) Tinv[k] = sI[k];
    // this workgroup's slab of the panel below the tile: L_IJ = A_IJ L_JJ^-T.  A row's outputs only read that
    // row's own w entries; they are all computed into registers before anything is overwritten.
    const int total = (nc - j0 - w) * w, base = blockIdx.x * 1024;
    double out[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int idx = base + t + n * 256;
        double v = 0.0;
        if (idx < total) {
            const int i = j0 + w + idx / w, c = idx % w;
            for (int k = 0; k <= c; ++k) v += A[(size_t)i * nc + j0 + k] * sI[c * PS_BC_W + k];
        }
        out[n] = v;
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int idx = base + t + n * 256;
        if (idx < total) A[(size_t)(j0 + w + idx / w) * nc + j0 + idx % w] = out[n];
    }
}

__global__ __launch_bounds__(256) void k_bchol_update(int nc, int j0, int w, double* __restrict__ A)
{
    // trailing update A[i][k] -= sum_c L[i][j0+c] L[k][j0+c] on the lower triangle, 32 x 32 tiles
    __shared__ double sa[32][PS_BC_W + 1], sb[32][PS_BC_W + 1];
    const int base = j0 + w, m = nc - base, nt = (m + 31) / 32;
    // blockIdx.x enumerates tiles (ti, tk) with tk <= ti
    int ti = 0, rem = blockIdx.x;
    while (rem > ti) { rem -= ti + 1; ++ti; }
    const int tk = rem;
    if (ti >= nt) return;
    const int t = threadIdx.x;
    for (int k = t; k < 32 * w; k += 256) {
        const int r = k / w, c = k % w;
        const int i = base + ti * 32 + r, kk = base + tk * 32 + r;
        sa[r][c] = i < nc ? A[(size_t)i * nc + j0 + c] : 0.0;
        sb[r][c] = kk < nc ? A[(size_t)kk * nc + j0 + c] : 0.0;
    }
    __syncthreads();
    for (int e = t; e < 32 * 32; e += 256) {
        const int r = e / 32, c = e % 32;
        const int i = base + ti * 32 + r, k = base + tk * 32 + c;
        if (i >= nc || k > i) continue;
        double v = 0.0;
#pragma unroll 8
        for (int q = 0; q < w; ++q) v += sa[r][q] * sb[c][q];
        A[(size_t)i * nc + k] -= v;
    }
}

// X = L^-1 (lower) and its transpose, in two parts.
// (1) k_btri_inverse: the PS_BI_S0 x PS_BI_S0 diagonal blocks.  Columns of X are independent forward substitutions:
//     one workgroup per PS_BI_CW columns (the whole chip), its column block of X in LDS, walking the 24-row blocks
//     below the diagonal (down to the end of its diagonal block) with the diagonal tiles' inverses.
// (2) k_btri_merge: the blocks below, level by level (s = S0, 2 S0, ...): [[X11, 0], [X21, X22]] with
//     X21 = -X22 (L21 X11) -- two triangular matrix products per level, 64 x 64 tiles over the whole chip,
//     instead of ever longer substitutions whose L traffic grows as nc^3 / 4 out of L2 (1.9 ms at nc = 1536).
//     The intermediate L21 X11 lives in the (zero) strictly lower triangle of XT and is cleared by k_btri_clear.
#define PS_BI_CW 4
#define PS_BI_S0 192
__global__ __launch_bounds__(256) void k_btri_inverse(
    int nc, const double* __restrict__ L, const double* __restrict__ Tinv_all /* one 24 x 24 tile per row block */,
    double* __restrict__ X, double* __restrict__ XT)
{
    extern __shared__ double sX[];                         // S0 x PS_BI_CW, + one 24 x PS_BI_CW tile
    const int j0 = blockIdx.x * PS_BI_CW, w = min(PS_BI_CW, nc - j0), t = threadIdx.x;
    const int b0 = (j0 / PS_BI_S0) * PS_BI_S0, b1 = min(nc, b0 + PS_BI_S0);   // this column block's diagonal block
    double* sT = sX + (size_t)PS_BI_S0 * PS_BI_CW;
    const int ib = (j0 / PS_BC_W) * PS_BC_W;               // first row block that can be non-zero
    for (int e = t; e < (ib - b0) * w; e += 256) sX[(size_t)(e / w) * PS_BI_CW + e % w] = 0.0;
    __syncthreads();
    for (int i0 = ib; i0 < b1; i0 += PS_BC_W) {
        const int wi = min(PS_BC_W, b1 - i0);
        // t = delta - L[I][ib .. i0) X[ib .. i0)][cols]
        for (int e = t; e < wi * w; e += 256) {
            const int r = e / w, c = e % w, i = i0 + r;
            double v = (i == j0 + c) ? 1.0 : 0.0;
#pragma unroll 4
            for (int k = ib; k < i0; ++k) v -= L[(size_t)i * nc + k] * sX[(size_t)(k - b0) * PS_BI_CW + c];
            sT[r * PS_BI_CW + c] = v;
        }
        __syncthreads();
        const double* Ti = Tinv_all + (size_t)(i0 / PS_BC_W) * PS_BC_W * PS_BC_W;
        for (int e = t; e < wi * w; e += 256) {
            const int r = e / w, c = e % w;
            double v = 0.0;
            for (int k = 0; k <= r; ++k) v += Ti[r * PS_BC_W + k] * sT[k * PS_BI_CW + c];
            sX[(size_t)(i0 + r - b0) * PS_BI_CW + c] = v;
        }
        __syncthreads();
    }
    for (int e = t; e < (b1 - b0) * w; e += 256) {         // (everything outside the diagonal blocks was zeroed by the host)
        const int i = b0 + e / w, c = e % w, j = j0 + c;
        const double v = (i >= j) ? sX[(size_t)(i - b0) * PS_BI_CW + c] : 0.0;
        X[(size_t)i * nc + j] = v;
        XT[(size_t)j * nc + i] = v;
    }
}

// one level of the merge.  stage 0: T = L21 X11 (into XT's lower triangle); stage 1: X21 = -X22 T (to X and XT).
// Pair p of the level: rows r0 = (2p+1) s .. r0 + s, columns c0 = 2 p s .. c0 + s.  grid = pairs x tiles x tiles.
#define PS_BM_T 64
#define PS_BM_K 16
__global__ __launch_bounds__(256) void k_btri_merge(
    int nc, int s, int stage, const double* __restrict__ L, double* __restrict__ X, double* __restrict__ XT)
{
    __shared__ double As[PS_BM_K][PS_BM_T + 4];
    __shared__ double Bs[PS_BM_K][PS_BM_T + 4];
    const int nt = (s + PS_BM_T - 1) / PS_BM_T;
    const int pair = blockIdx.x / (nt * nt), tile = blockIdx.x % (nt * nt);
    const int r0 = (2 * pair + 1) * s, c0 = 2 * pair * s;
    if (r0 >= nc) return;
    const int M = min(s, nc - r0);
    const int i0 = (tile / nt) * PS_BM_T, j0 = (tile % nt) * PS_BM_T;
    if (i0 >= M) return;
    // C[i][j] = sum_k A[i][k] B[k][j], i < M, j < s, k < (stage ? M : s)
    //   stage 0: A = L[r0 + i][c0 + k], B = X[c0 + k][c0 + j] (zero for k < j)
    //   stage 1: A = X[r0 + i][r0 + k] (zero for k > i), B = T[r0 + k][c0 + j]
    const double* A = stage ? X + (size_t)r0 * nc + r0 : L + (size_t)r0 * nc + c0;
    const double* B = stage ? XT + (size_t)r0 * nc + c0 : X + (size_t)c0 * nc + c0;
    const int K = stage ? M : s;
    const int kbeg = stage ? 0 : j0, kend = stage ? min(K, i0 + PS_BM_T) : K;
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
    for (int k0 = kbeg; k0 < kend; k0 += PS_BM_K) {
#pragma unroll
        for (int e = t; e < PS_BM_K * PS_BM_T; e += 256) {
            const int ai = e >> 4, ak = e & 15;            // A: 16 consecutive k of one row
            As[ak][ai] = (i0 + ai < M && k0 + ak < kend) ? A[(size_t)(i0 + ai) * nc + k0 + ak] : 0.0;
            const int bk = e >> 6, bj = e & 63;            // B: 64 consecutive j of one k
            Bs[bk][bj] = (k0 + bk < kend && j0 + bj < s) ? B[(size_t)(k0 + bk) * nc + j0 + bj] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < PS_BM_K; ++kk) {
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
            if (i >= M || j >= s) continue;
            if (stage == 0) XT[(size_t)(r0 + i) * nc + c0 + j] = acc[a][b];
            else { X[(size_t)(r0 + i) * nc + c0 + j] = -acc[a][b]; XT[(size_t)(c0 + j) * nc + r0 + i] = -acc[a][b]; }
        }
}

// XT's strictly lower triangle back to zero (it carried the merge intermediates)
__global__ __launch_bounds__(256) void k_btri_clear(int nc, double* __restrict__ XT)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)nc * nc) return;
    const int i = (int)(e / nc), j = (int)(e % nc);
    if (j < i && i / PS_BI_S0 != j / PS_BI_S0) XT[e] = 0.0;
}

struct CoarseRhsArgs {
    const int32_t *slo, *shi, *pnode;
    const double *pw0, *pw1, *LciT;
    double *tvec, *r, *w, *s, *p, *x;
    int with_coarse_rows;
    const int32_t* lag_status;
    int32_t* status;
    const double* bg;
    double* Mc;                        // split mode + lagged factor: where the rows of M go (else NULL)
};

template <int D>
PS_DEV void coarse_rhs_body(
    int nr, int ncb, const int32_t* __restrict__ slo, const int32_t* __restrict__ shi,
    const int32_t* __restrict__ pnode, const double* __restrict__ pw0, const double* __restrict__ pw1,
    const double* __restrict__ LciT, const int32_t* __restrict__ arow_ptr,
    double* __restrict__ Saug, double* __restrict__ tvec,
    double* __restrict__ r, double* __restrict__ w, double* __restrict__ s,
    double* __restrict__ p, double* __restrict__ x,
    int with_coarse_rows, const int32_t* __restrict__ lag_status, int32_t* __restrict__ status, double* stv,
    const double* __restrict__ bg);

// K_i = SZ_i Lci^T, written to both borders of the augmented BSR matrix.
// One workgroup per fine block row i; thread per (r, c) of the D x nc strip.
// Lagged mode (Ac != NULL): Lci is the inverse factor of the PREVIOUS iteration's A_c, so the
// coarse-coarse block of V^T S^ V is M = Lci A_c Lci^T (close to, but not exactly, I); workgroups
// nr .. nr+ncb-1 compute block row q of M the same way: strip = (Lci A_c)_q, then strip * Lci^T.
template <int D>
__global__ __launch_bounds__(256) void k_coarse_border(
    int nr, int ncb, const double* __restrict__ SZ, const double* __restrict__ Lci,
    const int32_t* __restrict__ arow_ptr, const int32_t* __restrict__ fine_nnz, double* __restrict__ Saug,
    int with_coarse_rows, const double* __restrict__ Ac,
    // the LAST workgroup (rhs.r != NULL) runs the coarse right-hand side instead (independent work, one launch less)
    CoarseRhsArgs rhs, int rpw /* fine block rows per workgroup: 1 or 4 */)
{
    constexpr int DD = D * D, RPW = 4, RW = RPW * D;                // fine block rows per workgroup
    extern __shared__ __attribute__((aligned(16))) double sT[];     // RW x nc
    const int nc = ncb * D;
    const int nfw = rpw == 1 ? nr : (nr + RPW - 1) / RPW;           // workgroups of the fine rows
    if (rhs.r && (int)blockIdx.x == (int)gridDim.x - 1) {
        coarse_rhs_body<D>(nr, ncb, rhs.slo, rhs.shi, rhs.pnode, rhs.pw0, rhs.pw1, rhs.LciT, arow_ptr, Saug, rhs.tvec,
                           rhs.r, rhs.w, rhs.s, rhs.p, rhs.x, rhs.with_coarse_rows, rhs.lag_status, rhs.status, sT, rhs.bg);
        return;
    }
    if ((int)blockIdx.x >= nfw) {                                   // lagged mode: row q of M
        const int q = blockIdx.x - nfw;
        for (int t = threadIdx.x; t < D * nc; t += blockDim.x) {
            const int r = t / nc, c = t % nc, rr = q * D + r;
            double v = 0.0;
#pragma unroll 4
            for (int k = 0; k <= rr; ++k) v += Lci[(size_t)rr * nc + k] * Ac[(size_t)k * nc + c];
            sT[t] = v;
        }
        __syncthreads();
        for (int t = threadIdx.x; t < D * nc; t += blockDim.x) {
            const int r = t / nc, c = t % nc, q2 = c / D, cc = c % D;
            double v = 0.0;
#pragma unroll 8
            for (int k = 0; k <= c; ++k) v += sT[r * nc + k] * rhs.LciT[(size_t)k * nc + c];   // = Lci[c][k], coalesced over c
            if (rhs.Mc) rhs.Mc[(size_t)(q * D + r) * nc + c] = v;        // split mode: dense M beside the matrix
            else Saug[(size_t)(arow_ptr[nr + q] + nr + q2) * DD + r * D + cc] = v;
        }
        return;
    }
    if (rpw == 1) {                                                 // small coarse levels: one block row per workgroup,
        const int i = blockIdx.x;                                   // one thread per strip entry (r, c)
        for (int t = threadIdx.x; t < D * nc; t += blockDim.x) {
            const int r = t / nc, c = t % nc, q = c / D, cc = c % D;
            sT[t] = SZ[((size_t)i * ncb + q) * DD + r * D + cc];
        }
        __syncthreads();
        const int row_slot = arow_ptr[i] + fine_nnz[i];             // first coarse column block of row i
        for (int t = threadIdx.x; t < D * nc; t += blockDim.x) {
            const int r = t / nc, c = t % nc, q = c / D, cc = c % D;
            double v = 0.0;
#pragma unroll 8
            for (int k = 0; k <= c; ++k) v += sT[r * nc + k] * rhs.LciT[(size_t)k * nc + c];   // = Lci[c][k], coalesced over c
            Saug[(size_t)(row_slot + q) * DD + r * D + cc] = v;                       // K   (row i, col nr+q)
            if (with_coarse_rows)
                Saug[(size_t)(arow_ptr[nr + q] + i) * DD + cc * D + r] = v;           // K^T (row nr+q, col i)
        }
        return;
    }
    // large coarse levels (nc >= 192): RPW block rows per workgroup share every element of the inverse factor
    // they load -- one thread per coarse column c, RW accumulators, the strip values broadcast from LDS
    const int i0 = blockIdx.x * RPW;
    for (int t = threadIdx.x; t < RW * nc; t += blockDim.x) {
        const int rr = t / nc, c = t % nc, q = c / D, cc = c % D, i = i0 + rr / D;
        sT[t] = i < nr ? SZ[((size_t)i * ncb + q) * DD + (rr % D) * D + cc] : 0.0;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < nc; c += blockDim.x) {
        double acc[RW];
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) acc[rr] = 0.0;
        for (int k = 0; k <= c; ++k) {
            const double l = rhs.LciT[(size_t)k * nc + c];          // = Lci[c][k], coalesced over c
#pragma unroll
            for (int rr = 0; rr < RW; ++rr) acc[rr] += sT[rr * nc + k] * l;
        }
        const int q = c / D, cc = c % D;
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
            const int i = i0 + rr / D, r = rr % D;
            if (i >= nr) continue;
            const int row_slot = arow_ptr[i] + fine_nnz[i];         // first coarse column block of row i
            Saug[(size_t)(row_slot + q) * DD + r * D + cc] = acc[rr];                   // K   (row i, col nr+q)
            if (with_coarse_rows)
                Saug[(size_t)(arow_ptr[nr + q] + i) * DD + cc * D + r] = acc[rr];       // K^T (row nr+q, col i)
        }
    }
}

// coarse rows: diagonal block = I ; rhs b~_c = Lci * (P^T g^) ; zero the CG vectors of the coarse rows
template <int D>
PS_DEV void coarse_rhs_body(
    int nr, int ncb, const int32_t* __restrict__ slo, const int32_t* __restrict__ shi,
    const int32_t* __restrict__ pnode, const double* __restrict__ pw0, const double* __restrict__ pw1,
    const double* __restrict__ LciT, const int32_t* __restrict__ arow_ptr,
    double* __restrict__ Saug, double* __restrict__ tvec /* nc scratch */,
    double* __restrict__ r /* fine part holds g^ */, double* __restrict__ w, double* __restrict__ s,
    double* __restrict__ p, double* __restrict__ x,
    int with_coarse_rows /* 1: write the coarse-coarse rows as identity (exact factor); 2: leave them (lagged) */,
    const int32_t* __restrict__ lag_status, int32_t* __restrict__ status, double* stv /* LDS, >= nc doubles */,
    const double* __restrict__ bg /* B^T g^ per fine row */)
{
    constexpr int DD = D * D;
    const int nc = ncb * D;
    // a lagged factor whose (side-stream) factorisation failed poisons this solve: report it
    if (lag_status && threadIdx.x == 0 && lag_status[ST_DIAG_FAIL]) atomicAdd(&status[ST_DIAG_FAIL], 1);
    // t_q = sum_{i in supp(q)} w(i,q) g^_i : 8 lanes per output, then a 3-step butterfly
    for (int base = 0; base < nc; base += blockDim.x / 8) {
        const int t = base + threadIdx.x / 8, sub = threadIdx.x & 7;
        double v = 0.0;
        if (t < nc) {
            const int q = t / D, c = t % D;
            for (int i = slo[q] + sub; i < shi[q]; i += 8)
                v += coarse_weight(i, q, pnode, pw0, pw1) * bg[(size_t)i * D + c];       // (P^T g^)_q, bg = B^T g^
        }
        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
        if (t < nc && sub == 0) tvec[t] = v;
    }
    if (with_coarse_rows == 1)
        for (int t = threadIdx.x; t < ncb * ncb * DD; t += blockDim.x) {
            const int q = t / (ncb * DD), q2 = (t / DD) % ncb, e = t % DD;
            Saug[(size_t)(arow_ptr[nr + q] + nr + q2) * DD + e] = (q == q2 && e / D == e % D) ? 1.0 : 0.0;
        }
    __syncthreads();
    for (int t = threadIdx.x; t < nc; t += blockDim.x) stv[t] = tvec[t];
    __syncthreads();
    for (int base = 0; base < nc; base += blockDim.x / 8) {         // b~_c[t] = sum_{k<=t} Lci[t][k] t_k
        const int t = base + threadIdx.x / 8, sub = threadIdx.x & 7;
        double v = 0.0;
        if (t < nc) {
#pragma unroll 4
            for (int k = sub; k <= t; k += 8) v += LciT[(size_t)k * nc + t] * stv[k];
        }
        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
        if (t < nc && sub == 0) {
            const size_t o = (size_t)nr * D + t;
            r[o] = v; w[o] = 0.0; s[o] = 0.0; p[o] = 0.0; x[o] = 0.0;
        }
    }
}

template <int D>
__global__ __launch_bounds__(1024) void k_coarse_rhs(
    int nr, int ncb, const int32_t* __restrict__ slo, const int32_t* __restrict__ shi,
    const int32_t* __restrict__ pnode, const double* __restrict__ pw0, const double* __restrict__ pw1,
    const double* __restrict__ LciT, const int32_t* __restrict__ arow_ptr,
    double* __restrict__ Saug, double* __restrict__ tvec,
    double* __restrict__ r, double* __restrict__ w, double* __restrict__ s,
    double* __restrict__ p, double* __restrict__ x, int with_coarse_rows,
    const int32_t* __restrict__ lag_status, int32_t* __restrict__ status, const double* __restrict__ bg)
{
    __shared__ double stv[400];
    coarse_rhs_body<D>(nr, ncb, slo, shi, pnode, pw0, pw1, LciT, arow_ptr, Saug, tvec, r, w, s, p, x,
                       with_coarse_rows, lag_status, status, stv, bg);
}

// x^_i = x~_f,i + pw0_i y[node_i] + pw1_i y[node_i + 1] with y = Lci^T x~_c ;  x_i = Linv_i^T x^_i
// every workgroup first forms y (nc values) in LDS: y_k = sum_{m>=k} Lci[m][k] x~_c[m]
template <int D>
__global__ __launch_bounds__(256) void k_coarse_recover(
    int nr, int ncb, const int32_t* __restrict__ pnode, const double* __restrict__ pw0,
    const double* __restrict__ pw1, const double* __restrict__ Linv, const double* __restrict__ Lci,
    const double* __restrict__ xh, double* __restrict__ x, const int32_t* __restrict__ gate,
    const double* __restrict__ Bmat)
{
    __shared__ double sy[400];                          // nc <= 384 (Gmax = 63 intervals, D = 6)
    if (gate && gate[ST_PCG_DONE] != 1) return;      // (2 = CG breakdown: the host falls back, nothing is applied)
    const int nc = ncb * D;
    const double* xc = xh + (size_t)nr * D;
    for (int base = 0; base < nc; base += blockDim.x / 8) {
        const int k = base + threadIdx.x / 8, sub = threadIdx.x & 7;
        double v = 0.0;
        if (k < nc) {
#pragma unroll 4
            for (int m = k + sub; m < nc; m += 8) v += Lci[(size_t)m * nc + k] * xc[m];
        }
        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
        if (k < nc && sub == 0) sy[k] = v;
    }
    __syncthreads();
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nr * D) return;
    const int i = t / D, c = t % D, q = pnode[i];
    const double w0 = pw0[i], w1 = pw1[i];
    double z[D];                                         // interpolated coarse unknown at pose i
#pragma unroll
    for (int m = 0; m < D; ++m) z[m] = w0 * sy[q * D + m] + ((q + 1 < ncb) ? w1 * sy[(q + 1) * D + m] : 0.0);
    double v = 0.0;
#pragma unroll
    for (int a = 0; a < D; ++a) {
        double xhat = xh[(size_t)i * D + a];
#pragma unroll
        for (int m = 0; m < D; ++m) xhat += Bmat[(size_t)i * D * D + a * D + m] * z[m];
        v += Linv[(size_t)i * D * D + a * D + c] * xhat;
    }
    x[t] = v;
}

// ---------------------------------------------------------------------------
// Lagged two-level setup in THREE launches (whole-iteration calls from the second Gauss-Newton iteration on).
// Any V = [I, X] gives a consistent augmented system V^T S^ V = [[S^, K], [K^T, M]], K = S^ X, M = X^T K (Griebel);
// the exact set-up takes X = P L_c^-T from THIS iteration's coarse matrix (five dependent launches + a 51 us serial
// factorisation).  Here X~ = P~ L~^-T comes ENTIRELY from the previous iteration -- basis blocks B~_i = L~_i^T Ad(T~_i)
// and coarse factor L~ -- so nothing coarse sits between the block-Jacobi factors and the CG:
//   k_block_jacobi_factor   L_i^-1, g^, and the CURRENT basis B_i (for the next iteration's X)
//   k_rows_setup            one workgroup per fine row: S^_ij = L_i^-1 S_ij L_j^-T  -> matrix;  SZ~_i = sum_j S^_ij P~_j;
//                           K_i = SZ~_i L~^-T -> both borders;  this row's part of M and of the coarse right-hand side,
//                           X~_i^T [K_i | g^_i];  and S^_ij B_j with the current basis for the side stream
//   k_coarse_mreduce        M = sum_i X~_i^T K_i (fixed order), coarse right-hand side, coarse CG vectors
// The next X~ (row sums with the current basis, A_c, its factorisation, k_coarse_xbuild) is formed on the low-priority
// side stream at the start of the NEXT call, beside the linearisation kernels.
// ---------------------------------------------------------------------------
#define PS_RS_THREADS 1024
#define PS_RS_MAXROW 96                 // fine blocks per row the LDS layout holds (4 x 96 x 288 B + strips < 160 KB)

template <int D>
__global__ __launch_bounds__(PS_RS_THREADS) void k_rows_setup(
    int nr, int ncb, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col_idx,
    const int32_t* __restrict__ aug_slot, const double* __restrict__ S, const double* __restrict__ Linv,
    const double* __restrict__ Blag, const double* __restrict__ Bcur,
    const int32_t* __restrict__ arow_ptr, const int32_t* __restrict__ fine_nnz,
    const int32_t* __restrict__ run_lo, const int32_t* __restrict__ run_hi,
    const int32_t* __restrict__ pnode, const double* __restrict__ pw0, const double* __restrict__ pw1,
    const double* __restrict__ LciT /* lagged L~^-1, transposed */, const double* __restrict__ X /* lagged X~: nr x D x nc */,
    const double* __restrict__ ghat /* g^ (nr x D) */,
    double* __restrict__ Saug, double* __restrict__ SB, double* __restrict__ Mpart /* nr x nc x (nc + 1) */,
    int lci_in_lds /* the LDS allocation has room for L~^-T (nc x nc) */, int ablate)
{
    constexpr int DD = D * D;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int i = blockIdx.x, t = threadIdx.x, NT = blockDim.x;
    const int nc = ncb * D, nf = row_ptr[i + 1] - row_ptr[i], rp = row_ptr[i], n36 = nf * DD;
    double* sS = lds;                       // nf x DD: S_ij, then L_i^-1 S_ij, then S^_ij
    double* sLj = sS + n36;                 // nf x DD: L_j^-1
    double* sBl = sLj + n36;                // nf x DD: lagged basis B~_j, then S^_ij B~_j
    double* sBc = sBl + n36;                // nf x DD: current basis B_j
    double* sT = sBc + n36;                 // D x nc: SZ~_i
    double* sK = sT + D * nc;               // D x (nc + 1): K_i | g^_i
    double* sX = sK + D * (nc + 1);         // D x nc: X~_i
    double* sLi = sX + D * nc;              // DD: L_i^-1
    double* sW = sLi + DD;                  // nf x 2: hat weights of column j; then nf: its left node (as double)
    double* sL = sW + 3 * PS_RS_MAXROW;     // nc x nc: L~^-T (when it fits)
    int32_t* sI = reinterpret_cast<int32_t*>(sL + (lci_in_lds ? nc * nc : 0));   // index words: one round trip with the data
    int32_t* sSlot = sI;                    // nf: slot of block k in the augmented matrix
    int32_t* sRun = sSlot + PS_RS_MAXROW;   // 2 ncb: run_lo / run_hi of this row, relative to its first block
    int32_t* sCrow = sRun + 2 * 64;         // ncb: first block of coarse row q
    // ---- every input of the row in one memory round trip (the gathers through col_idx: two)
    const int a_rp = arow_ptr[i];
    for (int idx = t; idx < n36; idx += NT) {
        const int k = idx / DD, e = idx - k * DD, b = rp + k, j = col_idx[b];
        sS[idx] = S[(size_t)b * DD + e];
        sLj[idx] = Linv[(size_t)j * DD + e];
        sBl[idx] = Blag[(size_t)j * DD + e];
        sBc[idx] = Bcur[(size_t)j * DD + e];
    }
    for (int k = t; k < nf; k += NT) {
        const int j = col_idx[rp + k];
        sW[2 * k] = pw0[j]; sW[2 * k + 1] = pw1[j]; sW[2 * PS_RS_MAXROW + k] = (double)pnode[j];
        sSlot[k] = aug_slot[rp + k];
    }
    for (int q = t; q < ncb; q += NT) {
        sRun[q] = run_lo[i * ncb + q] - a_rp; sRun[64 + q] = run_hi[i * ncb + q] - a_rp;
        sCrow[q] = arow_ptr[nr + q];
    }
    for (int idx = t; idx < D * nc; idx += NT) sX[idx] = X[(size_t)i * D * nc + idx];
    if (lci_in_lds && !(ablate & 1)) for (int idx = t; idx < nc * nc; idx += NT) sL[idx] = LciT[idx];
    if (t < DD) sLi[t] = Linv[(size_t)i * DD + t];
    if (t < D) sK[t * (nc + 1) + nc] = ghat[(size_t)i * D + t];
    const int row_slot = a_rp + fine_nnz[i];                        // first coarse column block of row i
    const int a0 = pnode[i] * D;
    __syncthreads();
    // ---- per block, by ONE wave (lane = entry (r, c); wave barriers only): S^_ij = L_i^-1 S_ij L_j^-T -> matrix;
    // S^_ij B~_j (lagged basis: this iteration's borders); S^_ij B_j (current basis: the side stream's input)
    {
        // (stage by stage over ALL of the wave's blocks, so that the LDS latencies of independent blocks overlap)
        constexpr int NB = (PS_RS_MAXROW + PS_RS_THREADS / 64 - 1) / (PS_RS_THREADS / 64);
        const int wv = t >> 6, lane = t & 63, nwv = NT >> 6;
        const bool act = lane < DD && !(ablate & 2);
        const int r = act ? lane / D : 0, c = act ? lane - (lane / D) * D : 0;
        double v[NB], v2[NB];
#pragma unroll
        for (int m = 0; m < NB; ++m) {
            const int k = wv + m * nwv;
            v[m] = 0.0;
            if (act && k < nf) {
#pragma unroll
                for (int a = 0; a < D; ++a) v[m] += sLi[r * D + a] * sS[k * DD + a * D + c];
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int m = 0; m < NB; ++m) { const int k = wv + m * nwv; if (act && k < nf) sS[k * DD + lane] = v[m]; }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int m = 0; m < NB; ++m) {
            const int k = wv + m * nwv;
            v[m] = 0.0;
            if (act && k < nf) {
#pragma unroll
                for (int a = 0; a < D; ++a) v[m] += sS[k * DD + r * D + a] * sLj[k * DD + c * D + a];
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int m = 0; m < NB; ++m) {
            const int k = wv + m * nwv;
            if (act && k < nf) { sS[k * DD + lane] = v[m]; Saug[(size_t)sSlot[k] * DD + lane] = v[m]; }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int m = 0; m < NB; ++m) {
            const int k = wv + m * nwv;
            v[m] = v2[m] = 0.0;
            if (act && k < nf) {
#pragma unroll
                for (int a = 0; a < D; ++a) {
                    const double sv = sS[k * DD + r * D + a];
                    v[m] += sv * sBl[k * DD + a * D + c];
                    v2[m] += sv * sBc[k * DD + a * D + c];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int m = 0; m < NB; ++m) {
            const int k = wv + m * nwv;
            if (act && k < nf) { sBl[k * DD + lane] = v[m]; SB[(size_t)sSlot[k] * DD + lane] = v2[m]; }
        }
    }
    __syncthreads();
    // ---- SZ~_i[q] = sum over the run of row i's blocks whose column lies in supp(q) of w(j, q) S^_ij B~_j
    for (int idx = t; idx < ncb * DD; idx += NT) {
        const int q = idx / DD, e = idx - q * DD, r = e / D, c = e - r * D;
        const int k0 = sRun[q], k1 = sRun[64 + q];
        double acc = 0.0;
        for (int k = k0; k < k1; ++k)
            acc += sBl[k * DD + e] * (((int)sW[2 * PS_RS_MAXROW + k] == q) ? sW[2 * k] : sW[2 * k + 1]);
        sT[r * nc + q * D + c] = acc;
    }
    __syncthreads();
    // ---- K_i = SZ~_i L~^-T, to both borders of the augmented matrix
    const double* Lt = lci_in_lds ? sL : LciT;
    for (int idx = t; idx < 2 * D * nc && !(ablate & 4); idx += NT) {      // two lanes per entry: even / odd terms of the sum
        const int o = idx >> 1, par = idx & 1;
        const int r = o / nc, c = o - r * nc, q = c / D, cc = c - q * D;
        double v = 0.0;
#pragma unroll 8
        for (int k = par; k <= c; k += 2) v += sT[r * nc + k] * Lt[(size_t)k * nc + c];     // = L~^-1[c][k]
        v += __shfl_xor(v, 1, 64);
        if (par) continue;
        sK[r * (nc + 1) + c] = v;
        Saug[(size_t)(row_slot + q) * DD + r * D + cc] = v;                            // K   (row i, col nr+q)
        Saug[(size_t)(sCrow[q] + i) * DD + cc * D + r] = v;                            // K^T (row nr+q, col i)
    }
    __syncthreads();
    // ---- this row's part of M = X~^T K and of the coarse right-hand side X~^T g^ (rows below the row's first node are zero)
    const int ncol = nc + 1;
    for (int idx = t; idx < (nc - a0) * ncol && !(ablate & 8); idx += NT) {
        const int a = a0 + idx / ncol, c = idx % ncol;
        double v = 0.0;
#pragma unroll
        for (int r = 0; r < D; ++r) v += sX[r * nc + a] * sK[r * ncol + c];
        Mpart[((size_t)i * nc + a) * ncol + c] = v;
    }
}

// M[a][c] = sum_i Mpart[i][a][c] over the rows that reach coarse row a (i < shi[node(a)]), in row order: 8 lanes per
// output + a 3-step butterfly (fixed order).  Column nc is the coarse right-hand side; the coarse CG vectors are cleared.
template <int D>
__global__ __launch_bounds__(256) void k_coarse_mreduce(
    int nr, int ncb, const int32_t* __restrict__ shi, const double* __restrict__ Mpart,
    const int32_t* __restrict__ pnode, const int32_t* __restrict__ arow_ptr, double* __restrict__ Saug,
    double* __restrict__ r, double* __restrict__ w, double* __restrict__ s, double* __restrict__ p, double* __restrict__ x,
    const int32_t* __restrict__ lag_status, int32_t* __restrict__ status, long long* __restrict__ hsetup, long long setup_seq)
{
    constexpr int DD = D * D;
    const int nc = ncb * D, ncol = nc + 1;
    const int o = (blockIdx.x * blockDim.x + threadIdx.x) / 8, sub = threadIdx.x & 7;
    // everything the side stream's next coarse operator reads (SB, the new basis) was written by the kernels BEFORE this
    // one: tell the host, which is waiting for the iteration anyway and enqueues that work when it sees the stamp
    if (hsetup && blockIdx.x == 0 && threadIdx.x == 0) { *reinterpret_cast<volatile long long*>(hsetup) = setup_seq; __threadfence_system(); }
    // a lagged factor whose (side-stream) factorisation failed poisons this solve: report it
    if (blockIdx.x == 0 && threadIdx.x == 0 && lag_status && lag_status[ST_DIAG_FAIL]) atomicAdd(&status[ST_DIAG_FAIL], 1);
    const bool live = o < nc * ncol;
    const int a = live ? o / ncol : 0, c = live ? o % ncol : 0;
    double v = 0.0;
    if (live) {
        const int iend = shi[a / D];
#pragma unroll 4
        for (int i = sub; i < iend; i += 8) v += Mpart[((size_t)i * nc + a) * ncol + c];
    }
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
    if (!live || sub != 0) return;
    if (c < nc) {
        const int q = a / D, rr = a % D, q2 = c / D, cc = c % D;
        Saug[(size_t)(arow_ptr[nr + q] + nr + q2) * DD + rr * D + cc] = v;
    } else {
        const size_t oo = (size_t)nr * D + a;
        r[oo] = v; w[oo] = 0.0; s[oo] = 0.0; p[oo] = 0.0; x[oo] = 0.0;
    }
}

// X_i = P_i L_c^-T (D x nc per fine row): X[i][r][a] = sum over the row's two nodes q of w(i,q) sum_m B_i[r][m] L_c^-1[a][qD+m]
template <int D>
__global__ __launch_bounds__(256) void k_coarse_xbuild(
    int nr, int ncb, const int32_t* __restrict__ pnode, const double* __restrict__ pw0, const double* __restrict__ pw1,
    const double* __restrict__ Bmat, const double* __restrict__ Lci, double* __restrict__ X)
{
    constexpr int DD = D * D;
    const int nc = ncb * D;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)nr * D * nc) return;
    const int i = (int)(t / (D * nc)), rem = (int)(t % (D * nc)), r = rem / nc, a = rem % nc;
    const int q0 = pnode[i];
    double v = 0.0;
#pragma unroll
    for (int dq = 0; dq < 2; ++dq) {
        const int q = q0 + dq;
        const double wq = dq == 0 ? pw0[i] : pw1[i];
        if (q >= ncb || wq == 0.0 || q * D > a) continue;          // L_c^-1 is lower triangular
        double acc = 0.0;
#pragma unroll
        for (int m = 0; m < D; ++m) acc += Bmat[(size_t)i * DD + r * D + m] * Lci[(size_t)a * nc + q * D + m];
        v += wq * acc;
    }
    X[t] = v;
}
