// ps_ransac.h -- frame-to-frame RANSAC on the device (gfx950, fp64).
//
// The step right before the motion-only solve in sliding-window VO (reference
// pyslam/pipelines/sparse.py:148-150): pyslam/pipelines/ransac.py
//   compute_transform_fast (:13-67)   rigid alignment of two small point sets by SVD
//   compute_ransac_cost   (:153-165)  inlier mask of every hypothesis over all points
//   perform_ransac        (:113-151)  argmax of the inlier counts (first maximum)
// One workgroup per hypothesis: lane 0 aligns the minimal set, then all 256 threads score the
// points (one coalesced pass over pts_1 / obs_2 per hypothesis, both L2-resident: 48 B per point).
#pragma once
#include "ps_math.h"

// Rotation C and translation r of  p_2 ~ C p_1 + r  for n point pairs (reference :17-42):
//   W = 1/n sum (p_2 - c_2)(p_1 - c_1)^T = U S V^T,  C = U diag(1, 1, det U det V) V^T,  r = c_2 - C c_1.
// With (u_i, v_i) the singular pairs of the two largest singular values the reference's formula
// equals  u_1 v_1^T + u_2 v_2^T + (u_1 x u_2)(v_1 x v_2)^T  whatever signs LAPACK picks for the
// third pair, so only those two pairs are needed.  They come from a one-sided (Hestenes) Jacobi
// SVD of the 3 x 3 matrix: columns of W V are rotated until orthogonal (relative accuracy
// ~eps sigma_1 / sigma_2, no squaring of the condition number as with W^T W).
PS_DEV void ransac_align(int n, const int32_t* __restrict__ idx /* or NULL: points 0..n-1 */,
                         const double* __restrict__ pts_1, const double* __restrict__ pts_2,
                         double* __restrict__ T /* 12: rows of [C | r] */)
{
    double c1[3] = {0, 0, 0}, c2[3] = {0, 0, 0};
    for (int k = 0; k < n; ++k) {
        const size_t i = idx ? (size_t)idx[k] : (size_t)k;
#pragma unroll
        for (int a = 0; a < 3; ++a) { c1[a] += pts_1[3 * i + a]; c2[a] += pts_2[3 * i + a]; }
    }
    const double inv_n = 1.0 / (double)n;
#pragma unroll
    for (int a = 0; a < 3; ++a) { c1[a] *= inv_n; c2[a] *= inv_n; }
    double A[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};      // A[col][row]: columns of W
    for (int k = 0; k < n; ++k) {
        const size_t i = idx ? (size_t)idx[k] : (size_t)k;
#pragma unroll
        for (int col = 0; col < 3; ++col) {
            const double q = pts_1[3 * i + col] - c1[col];
#pragma unroll
            for (int row = 0; row < 3; ++row) A[col][row] += (pts_2[3 * i + row] - c2[row]) * q;
        }
    }
#pragma unroll
    for (int col = 0; col < 3; ++col)
#pragma unroll
        for (int row = 0; row < 3; ++row) A[col][row] *= inv_n;
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};      // V[col][row]
    for (int sweep = 0; sweep < 30; ++sweep) {
        bool rotated = false;
#pragma unroll
        for (int pq = 0; pq < 3; ++pq) {
            const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
            const double al = A[p][0] * A[p][0] + A[p][1] * A[p][1] + A[p][2] * A[p][2];
            const double be = A[q][0] * A[q][0] + A[q][1] * A[q][1] + A[q][2] * A[q][2];
            const double ga = A[p][0] * A[q][0] + A[p][1] * A[q][1] + A[p][2] * A[q][2];
            if (ga == 0.0 || fabs(ga) <= 1.2e-16 * sqrt(al * be)) continue;
            rotated = true;
            const double zeta = (be - al) / (2.0 * ga);
            const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
            const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const double ap = A[p][r], aq = A[q][r], vp = V[p][r], vq = V[q][r];
                A[p][r] = c * ap - s * aq; A[q][r] = s * ap + c * aq;
                V[p][r] = c * vp - s * vq; V[q][r] = s * vp + c * vq;
            }
        }
        if (!rotated) break;
    }
    double sg[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) sg[k] = sqrt(A[k][0] * A[k][0] + A[k][1] * A[k][1] + A[k][2] * A[k][2]);
    int i1 = 0;
    if (sg[1] > sg[i1]) i1 = 1;
    if (sg[2] > sg[i1]) i1 = 2;
    int i2 = -1;
#pragma unroll
    for (int k = 0; k < 3; ++k) if (k != i1 && (i2 < 0 || sg[k] > sg[i2])) i2 = k;
    double u1[3], u2[3], v1[3], v2[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) { v1[r] = V[i1][r]; v2[r] = V[i2][r]; }
    if (sg[i1] > 0.0) {
#pragma unroll
        for (int r = 0; r < 3; ++r) u1[r] = A[i1][r] / sg[i1];
    } else {                                  // W = 0 (all sample points identical): LAPACK returns U = V = I
#pragma unroll
        for (int r = 0; r < 3; ++r) u1[r] = v1[r];
    }
    if (sg[i2] > 1e-300 && sg[i2] > 1e-15 * sg[i1]) {
#pragma unroll
        for (int r = 0; r < 3; ++r) u2[r] = A[i2][r] / sg[i2];
    } else {
        // rank <= 1 (collinear or repeated sample points): the rotation about u_1 is undetermined, LAPACK
        // returns an arbitrary completion; take the one that maps v_2 as close to itself as possible
        double w[3];
        const double d = v2[0] * u1[0] + v2[1] * u1[1] + v2[2] * u1[2];
#pragma unroll
        for (int r = 0; r < 3; ++r) w[r] = v2[r] - d * u1[r];
        double nw = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
        if (nw < 1e-8) {                      // v_2 parallel to u_1: any unit vector orthogonal to u_1
            const int m = fabs(u1[0]) <= fabs(u1[1]) ? (fabs(u1[0]) <= fabs(u1[2]) ? 0 : 2) : (fabs(u1[1]) <= fabs(u1[2]) ? 1 : 2);
            double e[3] = {0, 0, 0};
            e[m] = 1.0;
            const double de = u1[m];
#pragma unroll
            for (int r = 0; r < 3; ++r) w[r] = e[r] - de * u1[r];
            nw = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) u2[r] = w[r] / nw;
    }
    const double u3[3] = {u1[1] * u2[2] - u1[2] * u2[1], u1[2] * u2[0] - u1[0] * u2[2], u1[0] * u2[1] - u1[1] * u2[0]};
    const double v3[3] = {v1[1] * v2[2] - v1[2] * v2[1], v1[2] * v2[0] - v1[0] * v2[2], v1[0] * v2[1] - v1[1] * v2[0]};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        double tr = c2[r];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double cv = u1[r] * v1[c] + u2[r] * v2[c] + u3[r] * v3[c];
            T[4 * r + c] = cv;
            tr -= cv * c1[c];
        }
        T[4 * r + 3] = tr;
    }
}

PS_DEV void ransac_store_T(const double* __restrict__ T12, double* __restrict__ out16) {
#pragma unroll
    for (int k = 0; k < 12; ++k) out16[k] = T12[k];
    out16[12] = 0.0; out16[13] = 0.0; out16[14] = 0.0; out16[15] = 1.0;
}

// compute_transform_fast over a batch: one thread per point set (n points each, contiguous)
__global__ __launch_bounds__(64) void k_ransac_transforms(
    int batch, int n, const double* __restrict__ pts_1, const double* __restrict__ pts_2, double* __restrict__ T_out)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    double T[12];
    ransac_align(n, nullptr, pts_1 + (size_t)b * n * 3, pts_2 + (size_t)b * n * 3, T);
    ransac_store_T(T, T_out + (size_t)b * 16);
}

// inlier test of reference :157-163: squared (u, v, d) reprojection error of T p_1 against obs_2
PS_DEV bool ransac_inlier(const double* __restrict__ T, const double* __restrict__ p, const double* __restrict__ o,
                          double cu, double cv, double fu, double fv, double b, double thresh)
{
    const double x = T[0] * p[0] + T[1] * p[1] + T[2] * p[2] + T[3];
    const double y = T[4] * p[0] + T[5] * p[1] + T[6] * p[2] + T[7];
    const double z = T[8] * p[0] + T[9] * p[1] + T[10] * p[2] + T[11];
    const double iz = 1.0 / z;
    const double du = fu * x * iz + cu - o[0], dv = fv * y * iz + cv - o[1];
    const double dd = (b < 0.0 ? z : fu * b * iz) - o[2];        // b = -1: RGB-D camera, third coordinate is z
    return (du * du + dv * dv + dd * dd) < thresh;          // NaN / inf compare false, as in numpy
}

// One workgroup per hypothesis.  sample_idx == NULL: the transforms are given (compute_ransac_cost).
__global__ __launch_bounds__(256) void k_ransac_hypotheses(
    int num_pts, int set_size, const int32_t* __restrict__ sample_idx, const double* __restrict__ pts_1,
    const double* __restrict__ pts_2, const double* __restrict__ obs_2, const double* __restrict__ cam, double thresh,
    double* __restrict__ T_all /* [H][16], input when sample_idx == NULL */, int32_t* __restrict__ counts /* [H] */,
    uint8_t* __restrict__ masks /* [H][num_pts] */)
{
    __shared__ double sT[12];
    __shared__ int32_t scount[4];
    const int h = blockIdx.x, t = threadIdx.x;
    if (sample_idx) {
        if (t == 0) {
            double T[12];
            ransac_align(set_size, sample_idx + (size_t)h * set_size, pts_1, pts_2, T);
#pragma unroll
            for (int k = 0; k < 12; ++k) sT[k] = T[k];
            ransac_store_T(T, T_all + (size_t)h * 16);
        }
    } else if (t < 12) {
        sT[t] = T_all[(size_t)h * 16 + t];
    }
    __syncthreads();
    const double cu = cam[0], cv = cam[1], fu = cam[2], fv = cam[3], b = cam[4];
    int cnt = 0;
    for (int i = t; i < num_pts; i += 256) {
        const bool in = ransac_inlier(sT, pts_1 + 3 * (size_t)i, obs_2 + 3 * (size_t)i, cu, cv, fu, fv, b, thresh);
        masks[(size_t)h * num_pts + i] = in ? 1 : 0;
        cnt += in ? 1 : 0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    if ((t & 63) == 0) scount[t >> 6] = cnt;
    __syncthreads();
    if (t == 0) counts[h] = scount[0] + scount[1] + scount[2] + scount[3];
}

// np.argmax(inlier_nums): the FIRST hypothesis with the maximal count; copies its transform and mask out
__global__ __launch_bounds__(256) void k_ransac_best(
    int H, int num_pts, const int32_t* __restrict__ counts, const double* __restrict__ T_all,
    const uint8_t* __restrict__ masks, int32_t* __restrict__ best /* [2]: index, count */,
    double* __restrict__ T_best, uint8_t* __restrict__ best_mask)
{
    __shared__ int32_t sc[256], si[256];
    const int t = threadIdx.x;
    int bc = -1, bi = 0x7fffffff;
    for (int h = t; h < H; h += 256) {
        const int c = counts[h];
        if (c > bc) { bc = c; bi = h; }            // ascending h: keeps the first maximum of this thread
    }
    sc[t] = bc; si[t] = bi;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (t < off) {
            const int c2 = sc[t + off], i2 = si[t + off];
            if (c2 > sc[t] || (c2 == sc[t] && i2 < si[t])) { sc[t] = c2; si[t] = i2; }
        }
        __syncthreads();
    }
    const int hb = si[0];
    if (t == 0) { best[0] = hb; best[1] = sc[0]; }
    if (t < 16) T_best[t] = T_all[(size_t)hb * 16 + t];
    for (int i = t; i < num_pts; i += 256) best_mask[i] = masks[(size_t)hb * num_pts + i];
}
