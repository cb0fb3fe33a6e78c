// ps_k_bandpart.h -- PARTITIONED factorisation and inverse of the block-banded coarse matrix (round 5).
// Part of ps_kernels.h (included from there, after ps_k_band.h; not a stand-alone header).
#pragma once

// ---------------------------------------------------------------------------
// k_band_chol / k_band_inverse_rl (ps_k_band.h) walk the block columns of A_c in ONE workgroup: a chain of ncb dependent
// block steps (C2: 501 nodes, 0.95 ms) and, per column of the inverse, 2 nc dependent substitution steps (0.52 ms) -- in
// call 1 of every cold solve and beside the CG of the calls after it.  A banded matrix with B block off-diagonals decouples
// into independent pieces once B consecutive nodes are taken out between them:
//
//     nodes:   [ chunk 0 | sep 0 | chunk 1 | sep 1 | ... | chunk p-1 ]        chunk: m nodes, separator: B nodes
//     A = [[A_II, A_IS], [A_SI, A_SS]],   A_II = blockdiag(A_00, A_11, ...)   (chunks are more than B nodes apart)
//
//     G_a = A_aa^-1                           p independent banded factorisations + inverses   (batched k_band_chol,
//     V_a = G_a A_{a,S}   (m D x 2 s)         k_band_inverse_rl<true>: one workgroup / one wave per column and chunk)
//     T   = A_SS - sum_a A_{S,a} V_a          the separator system: (p - 1) s unknowns, block-tridiagonal in s x s
//                                             blocks (s = B D), i.e. banded with 2 B - 1 block off-diagonals
//     T^-1                                    the same two kernels on T (one "chunk")
//     W_a = V_a T^-1[sep(a), :]
//     A^-1 = [[G + V T^-1 V^T, -W], [-W^T, T^-1]]        dense fp32, symmetric by construction (mirrored stores)
//
// The chain of dependent block steps falls from ncb to m + (p - 1) B with p ~ ncb / (m + B): minimal at m + B ~ sqrt(B ncb)
// (C2: 501 -> 36 + 36, C4: 101 -> 16 + 16); everything else is small dense work over the whole chip.
// Needs 2 B - 1 <= PS_BAND_MAXB (the separator system goes through the same band kernels): B <= 4 -- C2 has 3, C4 has 4.
// Exact up to rounding (the same inverse as the serial kernels': tests/test_gpu_bandpart.py against numpy).
// ---------------------------------------------------------------------------
struct BandPartDev {
    int nc, lda, s, p, nS, ldv /* 2 s */;
    const int32_t* row_seg;     // [nc]  chunk a >= 0 for an interior row, -1 - x for a row of separator x
    const int32_t* row_loc;     // [nc]  scalar index within the chunk's interior / within the separator
    const int32_t* ch_row0;     // [p]   first scalar row of chunk a
    const int32_t* ch_n;        // [p]   scalar size of its interior
    const int64_t* ch_goff;     // [p]   offset of G_a (n x n, row-major) in G
    const int32_t* sep_row0;    // [p-1] first scalar row of separator x
};

// (i, j) of the symmetric A through its LOWER triangle (what k_band_chol reads)
PS_DEV double bp_alow(const double* __restrict__ A, int lda, int i, int j) {
    return i >= j ? A[(size_t)i * lda + j] : A[(size_t)j * lda + i];
}

// V[r][k]: k < s the left separator (a - 1), k >= s the right one (a); only the first / last s rows of the chunk couple to them
__global__ __launch_bounds__(256) void k_bp_v(BandPartDev bp, const double* __restrict__ A, const double* __restrict__ G,
                                              double* __restrict__ V)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)bp.nc * bp.ldv) return;
    const int r = (int)(t / bp.ldv), k = (int)(t % bp.ldv), s = bp.s;
    const int a = bp.row_seg[r];
    double v = 0.0;
    if (a >= 0) {
        const int l = bp.row_loc[r], n = bp.ch_n[a], r0 = bp.ch_row0[a];
        const double* Ga = G + bp.ch_goff[a] + (size_t)l * n;
        if (k < s) {
            if (a > 0) {
                const int c = bp.sep_row0[a - 1] + k;
                const int qn = min(s, n);
#pragma unroll 6
                for (int q = 0; q < qn; ++q) v += Ga[q] * A[(size_t)(r0 + q) * bp.lda + c];          // (row > column: lower triangle)
            }
        } else if (a < bp.p - 1) {
            const int c = bp.sep_row0[a] + (k - s);
#pragma unroll 6
            for (int q = max(0, n - s); q < n; ++q) v += Ga[q] * A[(size_t)c * bp.lda + r0 + q];        // (the separator's row, contiguous)
        }
    }
    V[t] = v;
}

// T = A_SS - sum_a A_{S,a} V_a, entry by entry (block-tridiagonal; everything else is zero)
__global__ __launch_bounds__(256) void k_bp_t(BandPartDev bp, const double* __restrict__ A, const double* __restrict__ V,
                                              double* __restrict__ T)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)bp.nS * bp.nS) return;
    const int s = bp.s, sg = (int)(t / bp.nS), ta = (int)(t % bp.nS);
    const int x = sg / s, sl = sg % s, y = ta / s, tl = ta % s;
    double v = 0.0;
    if (x == y) {
        const int c0 = bp.sep_row0[x];
        v = bp_alow(A, bp.lda, c0 + sl, c0 + tl);
        {   // chunk x: this is its right separator
            const int r0 = bp.ch_row0[x], n = bp.ch_n[x];
#pragma unroll 6
            for (int i = r0 + max(0, n - s); i < r0 + n; ++i) v -= A[(size_t)(c0 + sl) * bp.lda + i] * V[(size_t)i * bp.ldv + s + tl];
        }
        {   // chunk x + 1: its left separator
            const int r0 = bp.ch_row0[x + 1], n = bp.ch_n[x + 1];
            const int ie = r0 + min(s, n);
#pragma unroll 6
            for (int i = r0; i < ie; ++i) v -= A[(size_t)i * bp.lda + c0 + sl] * V[(size_t)i * bp.ldv + tl];
        }
    } else if (y == x + 1 || x == y + 1) {
        // chunk max(x, y) has separator min(x, y) on its left and max(x, y) on its right: A_{left,a} V_a[:, right] (and its transpose)
        const int lo = min(x, y), ll = x < y ? sl : tl, rl = x < y ? tl : sl;
        const int c0 = bp.sep_row0[lo], r0 = bp.ch_row0[lo + 1], n = bp.ch_n[lo + 1];
        const int ie = r0 + min(s, n);
#pragma unroll 6
        for (int i = r0; i < ie; ++i) v -= A[(size_t)i * bp.lda + c0 + ll] * V[(size_t)i * bp.ldv + s + rl];
    }
    T[t] = v;
}

// W[r][tau] = sum_k V[r][k] T^-1[sep(a) + k][tau]   (interior rows; the two separators of a chunk are adjacent in the separator order)
__global__ __launch_bounds__(256) void k_bp_w(BandPartDev bp, const double* __restrict__ V, const double* __restrict__ Tinv,
                                              double* __restrict__ Wm)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)bp.nc * bp.nS) return;
    const int r = (int)(t / bp.nS), ta = (int)(t % bp.nS);
    const int a = bp.row_seg[r];
    double v = 0.0;
    if (a >= 0) {
        const int lo = (a - 1) * bp.s;
        const double* vr = V + (size_t)r * bp.ldv;          // (the same for the consecutive tau of a wave: one broadcast load)
        // T^-1[q][tau]: consecutive lanes read consecutive columns (its row tau would be the same numbers -- T^-1 is exactly
        // symmetric -- at a stride of nS doubles per lane: 78 us at C2 that way, first version)
        const int k0 = max(0, -lo), k1 = min(bp.ldv, bp.nS - lo);
#pragma unroll 6
        for (int k = k0; k < k1; ++k) v += vr[k] * Tinv[(size_t)(lo + k) * bp.nS + ta];
    }
    Wm[t] = v;
}

// the separator rows and columns of A^-1: -W, -W^T and T^-1
__global__ __launch_bounds__(256) void k_bp_dense_sep(BandPartDev bp, const double* __restrict__ Wm, const double* __restrict__ Tinv,
                                                      float* __restrict__ Ainv, int ldo)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)bp.nc * bp.nS) return;
    const int r = (int)(t / bp.nS), ta = (int)(t % bp.nS);
    const int col = bp.sep_row0[ta / bp.s] + ta % bp.s;
    const int a = bp.row_seg[r];
    if (a >= 0) {
        const float f = (float)(-Wm[t]);
        Ainv[(size_t)r * ldo + col] = f; Ainv[(size_t)col * ldo + r] = f;
    } else {
        const int sg = (-1 - a) * bp.s + bp.row_loc[r];
        // (T^-1 comes out of k_band_inverse_rl mirrored: exactly symmetric)
        Ainv[(size_t)r * ldo + col] = (float)Tinv[(size_t)sg * bp.nS + ta];
    }
}

// interior x interior: A^-1[i][j] = [a == b] G_a[i][j] + sum_{k < 2 s} W[i][sep(b) + k] V[j][k].  One workgroup per 64 x 64 tile
// of a (chunk a, chunk b <= a) block, W and V strips through LDS, 4 x 4 outputs per thread, mirrored on store.
#define PS_BP_T 64
#define PS_BP_KMAX 48                         // 2 s = 2 B D with B <= 4 (see above), D <= 6
struct BandPartTile { int32_t a, b, i0, j0; };
__global__ __launch_bounds__(256) void k_bp_dense(BandPartDev bp, const BandPartTile* __restrict__ tiles,
                                                  const double* __restrict__ G, const double* __restrict__ V,
                                                  const double* __restrict__ Wm, float* __restrict__ Ainv, int ldo)
{
    constexpr int T = PS_BP_T;
    __shared__ double Wt[T][PS_BP_KMAX + 1], Vt[T][PS_BP_KMAX + 1];
    const BandPartTile tl = tiles[blockIdx.x];
    const int K = bp.ldv, t = threadIdx.x;
    const int ra = bp.ch_row0[tl.a], na = bp.ch_n[tl.a], rb = bp.ch_row0[tl.b], nb = bp.ch_n[tl.b];
    const int lo = (tl.b - 1) * bp.s;
    for (int e = t; e < T * K; e += 256) {
        const int i = e / K, k = e % K, q = lo + k;
        const int gi = tl.i0 + i, gj = tl.j0 + i;
        Wt[i][k] = (gi < na && q >= 0 && q < bp.nS) ? Wm[(size_t)(ra + gi) * bp.nS + q] : 0.0;
        Vt[i][k] = (gj < nb) ? V[(size_t)(rb + gj) * bp.ldv + k] : 0.0;
    }
    __syncthreads();
    const int ty = t >> 4, tx = t & 15;
    double acc[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] = 0.0;
    for (int k = 0; k < K; ++k) {
        double wa[4], vb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { wa[u] = Wt[ty + 16 * u][k]; vb[u] = Vt[tx + 16 * u][k]; }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[u][v] += wa[u] * vb[v];
    }
    const bool same = tl.a == tl.b;
    const double* Ga = G + bp.ch_goff[tl.a];
    // the tile goes out twice -- rows of chunk a and, mirrored, rows of chunk b -- both as runs of consecutive floats: the mirror
    // through LDS (a direct mirrored store writes 4 bytes per lane at a stride of a whole row)
    __syncthreads();
    float* Ot = reinterpret_cast<float*>(&Wt[0][0]);         // [64][65] floats (16.6 KB of the 25 KB strip)
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int i = ty + 16 * u, j = tx + 16 * v, li = tl.i0 + i, lj = tl.j0 + j;
            double val = acc[u][v];
            if (same && li < na && lj < nb) val += Ga[(size_t)li * na + lj];
            Ot[i * 65 + j] = (float)val;
        }
    __syncthreads();
    const bool diag_tile = same && tl.i0 == tl.j0;
    for (int e = t; e < T * T; e += 256) {
        const int i = e >> 6, j = e & 63, li = tl.i0 + i, lj = tl.j0 + j;
        if (li < na && lj < nb) {
            // a diagonal tile: both triangles from its lower one (exact symmetry, whatever the two sums rounded to)
            const float f = (diag_tile && j > i) ? Ot[j * 65 + i] : Ot[i * 65 + j];
            Ainv[(size_t)(ra + li) * ldo + rb + lj] = f;
        }
    }
    if (!diag_tile)
        for (int e = t; e < T * T; e += 256) {
            const int j = e >> 6, i = e & 63, li = tl.i0 + i, lj = tl.j0 + j;
            if (li < na && lj < nb) Ainv[(size_t)(rb + lj) * ldo + ra + li] = Ot[i * 65 + j];
        }
}
