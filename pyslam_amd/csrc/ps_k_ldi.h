// ps_k_ldi.h -- lagged dense inverse ("LDI") of the reduced system as the CG preconditioner (round 3).
// Part of ps_kernels.h (included from there; not a stand-alone header).
//
// The reduced solve is a chain of dependent launches of ~5 us each whatever they compute, and
// after block-Jacobi scaling + the hat-function coarse level the operator's effective condition number is ~2.5: no
// local preconditioner (bigger blocks, overlapping windows: measured, section 3) removes iterations.  What does is an
// operator that is nearly S^-1 itself.  Between consecutive Gauss-Newton iterations S changes by O(|dx|), so
//     X ~= S_prev^-1   (dense, fp32, n = nr * D <= ~1 500 unknowns: 5.7 MB at C3)
// preconditions the current system with eig(X S) in 1 +- 1e-2 .. 1e-4: 2-6 CG iterations instead of 18.  X is kept
// current WITHOUT any factorisation (a Cholesky's panel chain is exactly the dependent-launch pattern to avoid):
//   * one Newton-Schulz step per Gauss-Newton iteration on the low-priority side stream,
//         R = I - S^_new X,   X <- sym(X + X R)                      (two fp32 MFMA GEMMs, k_ldi_gemm)
//     in the block-Jacobi-scaled coordinates the inverse was seeded in (frozen factors: fp32 needs the scaling, the
//     scaling need not be current);
//   * seeded from the two-level operator the standard solver just used, X_0 = c (I + X~ X~^T), c from the extremal
//     Ritz values of that solve's own CG coefficients, three Newton-Schulz steps (error 0.4 -> 0.17 -> 0.03 -> 1e-3).
// The solver-stream side is the classic two-launch PCG (k_pcg_spmv) with z = X_u r as the preconditioner
// (k_ldi_init / k_ldi_update), X_u = L^-T X L^-1 the unscaled fp32 inverse, double-buffered against the side stream.
// X only PRECONDITIONS: whatever it is, CG converges to the solution of the CURRENT system at the caller's tolerance;
// if it does not within a few iterations the host falls back to the standard path and re-seeds (ps_host_ldi.h).
#pragma once

typedef float ps_f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------
// C = alpha A B + beta Dm + gamma I      fp32, row-major, M and N multiples of 64, K a multiple of 16.
// 256 threads = 2 x 2 waves, one v_mfma_f32_32x32x2_f32 accumulator (32 x 32) per wave, K in chunks of 16
// double-buffered through LDS.  Operand layout of the instruction (cdna4 ISA): A: lane l holds A[l % 32][l / 32],
// B: lane l holds B[l / 32][l % 32]; D: register v of lane l is D[8 (v / 4) + 4 (l / 32) + v % 4][l % 32].
// fro_part (optional): this workgroup's sum of squares of its C tile (fixed order) for a later ||C||_F^2.
// ---------------------------------------------------------------------------
#define PS_GM_BM 32
#define PS_GM_BN 32
#define PS_GM_BK 16
#define PS_GM_KS 4                              // waves per workgroup = K-split groups of one tile
// Tiling for a 1 216^2 product on 256 CUs: one 32 x 32 tile per workgroup (1 444 tiles, 741 when only the upper triangle is
// wanted), each of the workgroup's four waves owning every fourth K chunk of that tile with LDS double buffers of its OWN
// -- a wave stages, reads and multiplies without ever meeting a workgroup barrier, so four waves per SIMD (several
// workgroups per CU) hide each other's latencies -- and the four accumulators are summed through LDS at the end.
// (64 x 64 tiles, one per CU, were bounded by ONE tile's MFMA time: 16 us at peak, 30-58 us measured.)
__global__ __launch_bounds__(64 * PS_GM_KS) void k_ldi_gemm(
    int M, int N, int K, float alpha, const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
    float beta, const float* __restrict__ Dm, int ldd, float gamma, float* __restrict__ C, int ldc,
    double* __restrict__ fro_part,
    const int2* __restrict__ krange /* per 32-row tile: [k_lo, k_hi) outside which A's rows are zero (banded A), or null */,
    int upper_only /* C is symmetric: tiles below the diagonal are left untouched */,
    const float* __restrict__ dev_scale /* optional: alpha and gamma are multiplied by *dev_scale (a device-side scalar) */)
{
    constexpr int KS = PS_GM_KS;
    struct Tiles { float As[2][PS_GM_BK][PS_GM_BM + 4]; float Bs[2][PS_GM_BK][PS_GM_BN + 4]; };
    __shared__ __attribute__((aligned(16))) Tiles tl[KS];
    __shared__ double red[KS];
    if (upper_only && blockIdx.x < blockIdx.y) return;
    const int grp = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int m0 = blockIdx.y * PS_GM_BM, n0 = blockIdx.x * PS_GM_BN;
    int k_lo = 0, k_hi = K;
    if (krange) { const int2 kr = krange[blockIdx.y]; k_lo = kr.x; k_hi = kr.y; }
    // staging by one wave: A tile 32 x 16 = lane (row l / 2, 8 consecutive k), B tile 16 x 32 = lane (row l / 4, 8 consecutive n)
    const int am = l >> 1, ak = (l & 1) * 8;
    const int bk = l >> 2, bn = (l & 3) * 8;
    const int nchunks = (k_hi - k_lo) / PS_GM_BK;            // chunk c belongs to wave c % KS
    const float* Ap = A + (size_t)(m0 + am) * lda + k_lo + ak;
    const float* Bp = B + (size_t)(k_lo + bk) * ldb + n0 + bn;
    Tiles& T = tl[grp];
    ps_f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    const int fi = l & 31, fk = l >> 5;
    float4 ra0, ra1, rb0, rb1;
    int ch = grp;
    if (ch < nchunks) {
        ra0 = *reinterpret_cast<const float4*>(Ap + (size_t)ch * PS_GM_BK); ra1 = *reinterpret_cast<const float4*>(Ap + (size_t)ch * PS_GM_BK + 4);
        rb0 = *reinterpret_cast<const float4*>(Bp + (size_t)ch * PS_GM_BK * ldb); rb1 = *reinterpret_cast<const float4*>(Bp + (size_t)ch * PS_GM_BK * ldb + 4);
    }
    int buf = 0;
    for (; ch < nchunks; ch += KS) {
        // registers -> this wave's LDS buffer (A transposed to [k][m] so that a fragment read is 32 consecutive words)
        T.As[buf][ak + 0][am] = ra0.x; T.As[buf][ak + 1][am] = ra0.y; T.As[buf][ak + 2][am] = ra0.z; T.As[buf][ak + 3][am] = ra0.w;
        T.As[buf][ak + 4][am] = ra1.x; T.As[buf][ak + 5][am] = ra1.y; T.As[buf][ak + 6][am] = ra1.z; T.As[buf][ak + 7][am] = ra1.w;
        *reinterpret_cast<float4*>(&T.Bs[buf][bk][bn]) = rb0;
        *reinterpret_cast<float4*>(&T.Bs[buf][bk][bn + 4]) = rb1;
        const int nx = ch + KS;
        if (nx < nchunks) {                                   // the next chunk's global loads fly during this chunk's MFMAs
            ra0 = *reinterpret_cast<const float4*>(Ap + (size_t)nx * PS_GM_BK); ra1 = *reinterpret_cast<const float4*>(Ap + (size_t)nx * PS_GM_BK + 4);
            rb0 = *reinterpret_cast<const float4*>(Bp + (size_t)nx * PS_GM_BK * ldb); rb1 = *reinterpret_cast<const float4*>(Bp + (size_t)nx * PS_GM_BK * ldb + 4);
        }
        __builtin_amdgcn_wave_barrier();                     // (one wave: LDS operations complete in order; no workgroup barrier)
#pragma unroll
        for (int kk = 0; kk < PS_GM_BK; kk += 2) {
            const float a = T.As[buf][kk + fk][fi];
            const float b = T.Bs[buf][kk + fk][fi];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        buf ^= 1;
    }
    // sum the waves' accumulators through LDS (each wave parks into its own, now idle, tile buffers: 16 x 64 floats = 4 KB)
    float* park = reinterpret_cast<float*>(&tl[grp]);
    if (grp > 0) {
#pragma unroll
        for (int v = 0; v < 16; ++v) park[v * 64 + l] = acc[v];
    }
    __syncthreads();
    double sq = 0.0;
    if (grp == 0) {
#pragma unroll
        for (int g = 1; g < KS; ++g) {
            const float* pg = reinterpret_cast<const float*>(&tl[g]);
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[v] += pg[v * 64 + l];
        }
        if (dev_scale) { const float sc = *dev_scale; alpha *= sc; gamma *= sc; }
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int i = m0 + 8 * (v >> 2) + 4 * fk + (v & 3), j = n0 + fi;
            float c = alpha * acc[v];
            if (beta != 0.f) c += beta * Dm[(size_t)i * ldd + j];
            if (i == j) c += gamma;
            C[(size_t)i * ldc + j] = c;
            sq += (double)c * (double)c;
        }
        if (fro_part) {
            sq = wave_sum(sq);
            if (l == 0) fro_part[blockIdx.y * gridDim.x + blockIdx.x] = sq;
        }
    }
}

// Seed scale c = 1.9 / (lambda_min + lambda_max) of the preconditioned operator of the CG that has just converged, from its
// own coefficients (Lanczos matrix T_m: T_kk = 1/alpha_k + beta_{k-1}/alpha_{k-1}, T_k,k+1 = sqrt(beta_k)/alpha_k,
// beta_{k-1} = gamma_k / gamma_{k-1}); extremal eigenvalues by Sturm-sequence counts, 64 trial points per round (one wave).
// out[0] = c (0: no usable estimate -- the seed then produces a matrix the host rejects), out[1], out[2] = the two Ritz values.
__global__ __launch_bounds__(64) void k_ldi_ritz(const double* __restrict__ hist, int cap, const int32_t* __restrict__ status,
                                                float* __restrict__ out)
{
    constexpr int MAXM = 64;
    __shared__ double d[MAXM], e[MAXM];
    __shared__ int bad;
    const int lane = threadIdx.x;
    int m = status[ST_PCG_ITERS];
    if (m > MAXM) m = MAXM;
    if (lane == 0) { out[0] = 0.f; out[1] = 0.f; out[2] = 0.f; bad = 0; }
    __syncthreads();
    if (m < 2) return;
    if (lane < m) {
        const int k = lane;
        const double a = hist[cap + k], g = hist[k];
        if (!(a > 0.0) || !(g > 0.0)) bad = 1;
        d[k] = 1.0 / a + (k > 0 ? (g / hist[k - 1]) / hist[cap + k - 1] : 0.0);
        e[k] = (k + 1 < m) ? sqrt(hist[k + 1] / g) / a : 0.0;
    }
    __syncthreads();
    if (bad) return;
    double lo = d[0], hi = d[0];
    for (int k = 0; k < m; ++k) {                              // Gershgorin bracket (every lane the same)
        const double rad = (k > 0 ? fabs(e[k - 1]) : 0.0) + (k + 1 < m ? fabs(e[k]) : 0.0);
        lo = fmin(lo, d[k] - rad); hi = fmax(hi, d[k] + rad);
    }
    double ext[2];
    for (int which = 0; which < 2; ++which) {
        const int kk = which == 0 ? 0 : m - 1;                 // index of the wanted eigenvalue
        double a = lo, b = hi;
        for (int round = 0; round < 6; ++round) {              // 64-way multisection: the bracket shrinks 65-fold per round
            const double x = a + (b - a) * (double)(lane + 1) / 65.0;
            int c = 0; double q = d[0] - x;
            if (q < 0.0) ++c;
            for (int k = 1; k < m; ++k) {
                if (q == 0.0) q = 1e-300;
                q = d[k] - x - e[k - 1] * e[k - 1] / q;
                if (q < 0.0) ++c;
            }
            const unsigned long long above = __ballot(c > kk);          // lanes whose trial point lies above the eigenvalue
            const int first = above ? __ffsll((long long)above) - 1 : 64;
            const double na = first == 0 ? a : a + (b - a) * (double)first / 65.0;
            const double nb = first == 64 ? b : a + (b - a) * (double)(first + 1) / 65.0;
            a = na; b = nb;
        }
        ext[which] = 0.5 * (a + b);
    }
    if (lane != 0) return;
    if (!(ext[0] > 0.0) || !(ext[1] >= ext[0]) || !(ext[1] < 1e300)) return;
    out[0] = (float)(1.9 / (ext[0] + ext[1])); out[1] = (float)ext[0]; out[2] = (float)ext[1];
}

// S32 (np x np, fp32, row-major) <- L_i^-1 S_ij L_j^-T for every block of the BSR pattern, with the FROZEN factors Linv.
// (Entries outside the pattern are zero from the one-time clear; the padding diagonal is one.)
template <int D>
__global__ __launch_bounds__(64) void k_ldi_scaled_dense(
    const int32_t* __restrict__ brow_of, const int32_t* __restrict__ col_idx, const double* __restrict__ S,
    const double* __restrict__ Linv, float* __restrict__ S32, int np)
{
    constexpr int DD = D * D;
    __shared__ double sS[36], sT[36], sLi[36], sLj[36];
    const int b = blockIdx.x, t = threadIdx.x;
    const int i = brow_of[b], j = col_idx[b];
    if (t < DD) {
        sS[t] = S[(size_t)b * DD + t];
        sLi[t] = Linv[(size_t)i * DD + t];
        sLj[t] = Linv[(size_t)j * DD + t];
    }
    __syncthreads();
    const int r = t / D, c = t % D;
    if (t < DD) {
        double v = 0.0;
#pragma unroll
        for (int a = 0; a < D; ++a) v += sLi[r * D + a] * sS[a * D + c];
        sT[t] = v;
    }
    __syncthreads();
    if (t < DD) {
        double v = 0.0;
#pragma unroll
        for (int a = 0; a < D; ++a) v += sT[r * D + a] * sLj[c * D + a];
        S32[(size_t)(i * D + r) * np + j * D + c] = (float)v;
    }
}

// identity on the padding (rows / columns n .. np-1) of an np x np matrix whose other entries were cleared
__global__ void k_ldi_pad_identity(int n, int np, float* __restrict__ A, float value)
{
    const int i = n + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < np) A[(size_t)i * np + i] = value;
}

// seed operands: Xt (np x kp) = fp32 of the two-level prolongation X~ (n x nc, fp64), XtT (kp x np) its transpose, zero padded
__global__ __launch_bounds__(256) void k_ldi_seed_prep(int n, int nc, int np, int kp, const double* __restrict__ X,
                                                      float* __restrict__ Xt, float* __restrict__ XtT)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)np * kp) return;
    const int i = (int)(e / kp), a = (int)(e % kp);
    const float v = (i < n && a < nc) ? (float)X[(size_t)i * nc + a] : 0.f;
    Xt[e] = v;
    XtT[(size_t)a * np + i] = v;
}

// X32 <- sym(T) on the real blocks (+ the padding diagonal), and Xu <- L_i^-T X_ij L_j^-1: the unscaled fp32 inverse the
// solver stream applies.  One wave per block (bi, bj).
template <int D>
__global__ __launch_bounds__(64) void k_ldi_sym_unscale(
    int nr, int np, const float* __restrict__ T, const double* __restrict__ Linv, float* __restrict__ X32,
    float* __restrict__ Xu /* may be null: symmetrise only */)
{
    constexpr int DD = D * D;
    __shared__ double sX[36], sT[36], sLi[36], sLj[36];
    const int t = threadIdx.x;
    const int n = nr * D;
    if ((int)blockIdx.x >= nr * nr) {                        // tail workgroups: padding diagonal
        const int i = n + ((int)blockIdx.x - nr * nr) * 64 + t;
        if (i < np) X32[(size_t)i * np + i] = T[(size_t)i * np + i];
        return;
    }
    const int bi = blockIdx.x / nr, bj = blockIdx.x % nr;
    const int r = t / D, c = t % D;
    if (t < DD) {
        // T was computed for tiles on and above the diagonal only (PS_GM_BM x PS_GM_BN tiles): an entry below it is its mirror image's
        const int gi = bi * D + r, gj = bj * D + c, ti = gi / PS_GM_BM, tj = gj / PS_GM_BN;
        const double a = T[(size_t)gi * np + gj], b = T[(size_t)gj * np + gi];
        const double s = ti < tj ? a : (ti > tj ? b : 0.5 * (a + b));
        sX[t] = s;
        X32[(size_t)(bi * D + r) * np + bj * D + c] = (float)s;
        if (Xu) { sLi[t] = Linv[(size_t)bi * DD + t]; sLj[t] = Linv[(size_t)bj * DD + t]; }
    }
    if (!Xu) return;
    __syncthreads();
    if (t < DD) {                                            // sT = L_i^-T X  : sT[r][c] = sum_a Li[a][r] X[a][c]
        double v = 0.0;
#pragma unroll
        for (int a = 0; a < D; ++a) v += sLi[a * D + r] * sX[a * D + c];
        sT[t] = v;
    }
    __syncthreads();
    if (t < DD) {                                            // (sT L_j^-1)[r][c] = sum_a sT[r][a] Lj[a][c]
        double v = 0.0;
#pragma unroll
        for (int a = 0; a < D; ++a) v += sT[r * D + a] * sLj[a * D + c];
        Xu[(size_t)(bi * D + r) * np + bj * D + c] = (float)v;
    }
}

// X32 <- T mirrored: T holds the tiles on and above the diagonal (32 x 32 tiles); tile (I, J), I < J, is copied as it is and
// transposed into (J, I) through LDS; diagonal tiles are symmetrised.  One 256-thread workgroup per upper tile.
__global__ __launch_bounds__(256) void k_ldi_mirror(int np, const float* __restrict__ T, float* __restrict__ X32)
{
    __shared__ float tile[32][33];
    if (blockIdx.x < blockIdx.y) return;
    const int I = blockIdx.y * 32, J = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // ty: 0..7
#pragma unroll
    for (int r = ty; r < 32; r += 8) tile[r][tx] = T[(size_t)(I + r) * np + J + tx];
    __syncthreads();
    if (I == J) {
#pragma unroll
        for (int r = ty; r < 32; r += 8) X32[(size_t)(I + r) * np + J + tx] = 0.5f * (tile[r][tx] + tile[tx][r]);
    } else {
#pragma unroll
        for (int r = ty; r < 32; r += 8) {
            X32[(size_t)(I + r) * np + J + tx] = tile[r][tx];
            X32[(size_t)(J + r) * np + I + tx] = tile[tx][r];
        }
    }
}

// sum of the per-workgroup squares of the last GEMM (fixed order) -> one double the host reads after the side stream's event
__global__ __launch_bounds__(256) void k_ldi_fro_total(int nparts, const double* __restrict__ parts, double* __restrict__ out)
{
    __shared__ double lds[16];
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) s += parts[i];
    s = block_sum(s, lds);
    if (threadIdx.x == 0) *out = s;
}

// ---------------------------------------------------------------------------
// solver stream: classic PCG with the dense preconditioner.  PS_LDI_ROWS rows of X_u per workgroup (one per wave: with
// two, C3's 1 194 rows made 150 workgroups for 256 CUs -- reduced-solve stage 56.0 -> 54.3 us with 299),
// the whole residual vector in LDS.
// ---------------------------------------------------------------------------
#define PS_LDI_ROWS 4
#define PS_LDI_MAXN 3328
#define PS_LDI_RPW (PS_LDI_ROWS / 4)            // rows of X_u per wave
#define PS_LDI_NF4 (PS_LDI_MAXN / 256)          // float4 pieces of a row per lane

// A wave's rows of X_u into registers (float4 per lane and 256 columns): issued BEFORE anything the kernel has to wait
// for, so the matrix is in flight during the reductions that yield alpha (one exposed memory latency per launch, not two)
struct LdiRows { float4 x[PS_LDI_RPW][PS_LDI_NF4]; };

PS_DEV void ldi_load_rows(LdiRows& R, const float* __restrict__ Xu, int row0, int n, int np, int lane)
{
#pragma unroll
    for (int q = 0; q < PS_LDI_RPW; ++q) {
        const int row = row0 + q;
#pragma unroll
        for (int u = 0; u < PS_LDI_NF4; ++u) {
            const int j = lane * 4 + 256 * u;
            R.x[q][u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < n && j < np) R.x[q][u] = *reinterpret_cast<const float4*>(Xu + (size_t)row * np + j);
        }
    }
}

PS_DEV double ldi_dot_row(const LdiRows& R, int q, const double* __restrict__ rv, int np, int lane)
{
    double s = 0.0;
#pragma unroll
    for (int u = 0; u < PS_LDI_NF4; ++u) {
        const int j = lane * 4 + 256 * u;
        if (j < np) {
            const float4 x = R.x[q][u];
            s += (double)x.x * rv[j] + (double)x.y * rv[j + 1] + (double)x.z * rv[j + 2] + (double)x.w * rv[j + 3];
        }
    }
    return wave_sum(s);
}

// x = 0, r = g, z = X_u g, partials of r.z; stamps the set-up word (S is final: the side stream's update may start)
__global__ __launch_bounds__(256) void k_ldi_init(
    int n, int np, const float* __restrict__ Xu, const double* __restrict__ g, double* __restrict__ x,
    double* __restrict__ r, double* __restrict__ z, double* __restrict__ part, int32_t* __restrict__ status,
    long long* __restrict__ hsetup, long long setup_seq)
{
    extern __shared__ __attribute__((aligned(16))) double rv[];      // np doubles
    __shared__ double wpart[4];
    const int t = threadIdx.x, w = t >> 6, lane = t & 63;
    const int row0 = blockIdx.x * PS_LDI_ROWS + w * PS_LDI_RPW;
    LdiRows R;
    ldi_load_rows(R, Xu, row0, n, np, lane);
    if (blockIdx.x == 0 && t == 0) {
        if (hsetup) { __threadfence_system(); *reinterpret_cast<volatile long long*>(hsetup) = setup_seq; }
        status[ST_PCG_DONE] = 0; status[ST_PCG_ITERS] = 0;
    }
    for (int j = t; j < np; j += 256) rv[j] = j < n ? g[j] : 0.0;
    __syncthreads();
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < PS_LDI_RPW; ++q) {
        const int row = row0 + q;
        if (row < n) {                                       // (uniform per wave)
            const double zi = ldi_dot_row(R, q, rv, np, lane);
            if (lane == 0) { x[row] = 0.0; r[row] = rv[row]; z[row] = zi; }
            acc += zi * rv[row];
        }
    }
    if (lane == 0) wpart[w] = acc;
    __syncthreads();
    if (t == 0) part[blockIdx.x] = ((wpart[0] + wpart[1]) + wpart[2]) + wpart[3];
}

// alpha = rz / p.q ; x += alpha p ; r_new = r_old - alpha q (every workgroup forms ALL of it in LDS, the owner stores its
// rows: r is double-buffered by iteration parity because other workgroups still read r_old) ; z = X_u r_new ; partials r.z
__global__ __launch_bounds__(256) void k_ldi_update(
    int n, int np, const float* __restrict__ Xu, const double* __restrict__ p, const double* __restrict__ q,
    double* __restrict__ x, const double* __restrict__ r_old, double* __restrict__ r_new, double* __restrict__ z,
    const double* __restrict__ pq_part, int npartA, const double* __restrict__ hist, int k,
    double* __restrict__ part, const int32_t* __restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) double rv[];
    __shared__ double lds[16];
    __shared__ double wpart[4];
    const int t = threadIdx.x, w = t >> 6, lane = t & 63;
    const int row0 = blockIdx.x * PS_LDI_ROWS + w * PS_LDI_RPW;
    // every load first: none depends on alpha
    const int done = status[ST_PCG_DONE];
    const double rzk = hist[k];
    double pq = 0.0;
    for (int i = t; i < npartA; i += 256) pq += pq_part[i];
    double ro[PS_LDI_NF4], qo[PS_LDI_NF4];
#pragma unroll
    for (int u = 0; u < PS_LDI_NF4; ++u) {
        const int j = t + 256 * u;
        ro[u] = 0.0; qo[u] = 0.0;
        if (j < n) { ro[u] = r_old[j]; qo[u] = q[j]; }
    }
    double pr[PS_LDI_RPW], xr[PS_LDI_RPW];
#pragma unroll
    for (int qq = 0; qq < PS_LDI_RPW; ++qq) {
        pr[qq] = 0.0; xr[qq] = 0.0;
        if (lane == 0 && row0 + qq < n) { pr[qq] = p[row0 + qq]; xr[qq] = x[row0 + qq]; }
    }
    LdiRows R;
    ldi_load_rows(R, Xu, row0, n, np, lane);
    if (done) return;
    pq = block_sum(pq, lds);
    const double alpha = rzk / pq;
#pragma unroll
    for (int u = 0; u < PS_LDI_NF4; ++u) {
        const int j = t + 256 * u;
        if (j < np) rv[j] = ro[u] - alpha * qo[u];
    }
    __syncthreads();
    double acc = 0.0;
#pragma unroll
    for (int qq = 0; qq < PS_LDI_RPW; ++qq) {
        const int row = row0 + qq;
        if (row < n) {
            const double zi = ldi_dot_row(R, qq, rv, np, lane);
            if (lane == 0) { x[row] = xr[qq] + alpha * pr[qq]; r_new[row] = rv[row]; z[row] = zi; }
            acc += zi * rv[row];
        }
    }
    if (lane == 0) wpart[w] = acc;
    __syncthreads();
    if (t == 0) part[blockIdx.x] = ((wpart[0] + wpart[1]) + wpart[2]) + wpart[3];
}


// ---------------------------------------------------------------------------
// DIRECT seed (pose graphs): S as a dense fp64 matrix for the blocked Cholesky + triangular inverse of ps_k_coarse.h
// (k_bchol_*, k_btri_*) and k_xcg_ainv, which leave S^-1 in fp32 where the Newton-Schulz seed leaves its X_u.
// ---------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(64) void k_ldi_dense64(
    const int32_t* __restrict__ brow_of, const int32_t* __restrict__ col_idx, const double* __restrict__ S, double* __restrict__ A, int n)
{
    constexpr int DD = D * D;
    const int b = blockIdx.x, t = threadIdx.x;
    if (t < DD) A[(size_t)(brow_of[b] * D + t / D) * n + col_idx[b] * D + t % D] = S[(size_t)b * DD + t];
}

// the word ldi_decide reads as "rms of the residual": 0 = take the inverse, huge = the factorisation met a non-positive pivot
__global__ void k_ldi_direct_done(const int32_t* __restrict__ stat, double* __restrict__ hfro)
{
    if (threadIdx.x == 0) { *hfro = stat[ST_DIAG_FAIL] ? 1e300 : 0.0; __threadfence_system(); }
}
