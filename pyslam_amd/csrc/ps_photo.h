// ps_photo.h -- dense photometric alignment: the reference's PhotometricResidualSE3
// (pyslam/residuals/photometric_residual.py:38-161) inside a one-pose Gauss-Newton iteration.
//
// One residual per reference pixel: r = s (I_track(project(T p)) - I_ref), s = 1 / sqrt(var_I + var_d J_d^2),
// row Jacobian s g [I | -(T p)^] with g = grad(I_ref) * d project / d p (the reference-image gradient stands in
// for the tracking image's, :40-41), J_d = g R (d p / d depth).  Pixels whose reprojection leaves the image are
// dropped (:99, :106).  Element-wise IRLS as in Problem (pyslam/problem.py:351-360).
//
//   k_photo_pass    thread / pixel (strided)  : residual, weight, 1 x 6 row -> 21 + 6 + 2 sums per workgroup
//   k_photo_finish  one workgroup             : partials in fixed order, 6 x 6 Cholesky solve, pose update
// HBM-bound: 72 B of tables per pixel + 4 gathered image samples (L2-resident image); every reduction has a
// fixed order (bitwise reproducible).
#pragma once
#include "ps_kernels.h"

#define PS_PHOTO_NACC 32        // 21 (upper triangle of H) + 6 (b) + cost + valid count, padded
#define PS_PHOTO_PPT 4          // pixels per thread

struct PhotoArgs {
    int n;
    const double* pt_ref;       // n x 3
    const double* im_ref;       // n
    const double* im_jac;       // n x 2  (dI/du, dI/dv of the reference image)
    const double* tri_jac_d;    // n x 3  (d point / d depth-or-disparity)
    const double* image;        // h x w tracking image
    int h, w;
    double cu, cv, fu, fv, b;
    int cam_type;               // 0 stereo (u, v, disparity), 1 RGB-D (u, v, depth)
    double cam_w, cam_h;        // validity bounds of the camera model (stereo_camera.py:93-97, rgbd_camera.py:91-95)
    double var_i, var_d;
    int loss_id; double loss_k;
};

// reference pyslam/utils.py:27-75 with x = x[0], y = y[0] (the committed body indexes [1], out of bounds):
// weights from the unclipped corner coordinates, THEN the corners are clamped to the image
PS_DEV double photo_bilinear(const double* __restrict__ im, int h, int w, double x, double y) {
    int x0 = (int)x, y0 = (int)y;                       // truncation, like np.int
    int x1 = x0 + 1, y1 = y0 + 1;
    const double wa = (x1 - x) * (y1 - y), wb = (x1 - x) * (y - y0);
    const double wc = (x - x0) * (y1 - y), wd = (x - x0) * (y - y0);
    x0 = min(max(x0, 0), w - 1); x1 = min(max(x1, 0), w - 1);
    y0 = min(max(y0, 0), h - 1); y1 = min(max(y1, 0), h - 1);
    return wa * im[(size_t)y0 * w + x0] + wb * im[(size_t)y1 * w + x0] +
           wc * im[(size_t)y0 * w + x1] + wd * im[(size_t)y1 * w + x1];
}

// residual (+ 1 x 6 row) of pixel i; false when the reprojection is not a valid measurement
template <bool WITH_J>
PS_DEV bool photo_eval(const PhotoArgs& a, const Se3& T, int i, double& r, double* __restrict__ J) {
    double p[3];
    se3_apply(T, a.pt_ref + (size_t)3 * i, p);
    const double iz = 1.0 / p[2];
    const double u = a.fu * p[0] * iz + a.cu, v = a.fv * p[1] * iz + a.cv;
    const double d = a.cam_type == 1 ? p[2] : a.fu * a.b * iz;
    bool ok = (d > 0.0) && (v > 0.0) && (v < a.cam_h) && (u > 0.0) && (u < a.cam_w);
    if (a.cam_type == 0) ok = ok && (d < a.cam_w);
    if (!ok) return false;
    const double gu = a.im_jac[(size_t)2 * i], gv = a.im_jac[(size_t)2 * i + 1];
    const double iz2 = iz * iz;
    const double g0 = gu * a.fu * iz, g1 = gv * a.fv * iz;
    const double g2 = -(gu * a.fu * p[0] + gv * a.fv * p[1]) * iz2;
    const double* tj = a.tri_jac_d + (size_t)3 * i;
    double jd = 0.0;
#pragma unroll
    for (int j = 0; j < 3; ++j) jd += (g0 * T.R[j] + g1 * T.R[3 + j] + g2 * T.R[6 + j]) * tj[j];
    const double s = 1.0 / sqrt(a.var_i + a.var_d * jd * jd);
    r = s * (photo_bilinear(a.image, a.h, a.w, u, v) - a.im_ref[i]);
    if (WITH_J) {
        J[0] = s * g0; J[1] = s * g1; J[2] = s * g2;
        J[3] = s * (-g1 * p[2] + g2 * p[1]);
        J[4] = s * (g0 * p[2] - g2 * p[0]);
        J[5] = s * (-g0 * p[1] + g1 * p[0]);
    }
    return true;
}

// mode 0: cost only (acc[27], acc[28]);  mode 1: + H (upper triangle, row-major) and b = -J^T W r
__global__ __launch_bounds__(256) void k_photo_pass(PhotoArgs a, const double* __restrict__ pose, int with_normal,
                                                     double* __restrict__ partials /* gridDim.x x PS_PHOTO_NACC */)
{
    __shared__ double lds[4][PS_PHOTO_NACC];
    const Se3 T = se3_load(pose);
    double acc[PS_PHOTO_NACC];
#pragma unroll
    for (int k = 0; k < PS_PHOTO_NACC; ++k) acc[k] = 0.0;
    const int base = blockIdx.x * (256 * PS_PHOTO_PPT) + threadIdx.x;
#pragma unroll
    for (int q = 0; q < PS_PHOTO_PPT; ++q) {
        const int i = base + q * 256;
        if (i >= a.n) continue;
        double r, J[6];
        bool ok;
        if (with_normal) ok = photo_eval<true>(a, T, i, r, J); else ok = photo_eval<false>(a, T, i, r, J);
        if (!ok) continue;
        acc[27] += ps_loss_rho(a.loss_id, a.loss_k, r);
        acc[28] += 1.0;
        if (with_normal) {
            const double wgt = ps_loss_weight(a.loss_id, a.loss_k, r);
            int k = 0;
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const double wj = wgt * J[c];
#pragma unroll
                for (int c2 = c; c2 < 6; ++c2) acc[k++] += wj * J[c2];
                acc[21 + c] -= wj * r;
            }
        }
    }
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 29; ++k) {
        if (!with_normal && k < 27) continue;
        const double v = wave_sum(acc[k]);
        if (lane == 0) lds[wv][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < PS_PHOTO_NACC) {
        const int k = threadIdx.x;
        const bool used = k < 29 && (with_normal || k >= 27);
        partials[(size_t)blockIdx.x * PS_PHOTO_NACC + k] = used ? ((lds[0][k] + lds[1][k]) + lds[2][k]) + lds[3][k] : 0.0;
    }
}

// out: [0..35] H (full, row-major) | [36..41] b | [42] cost | [43] valid pixels | [44..49] dx | [50] status
// (0 ok, 1 H not positive definite) | [51..62] updated pose.  update: 0 none, 1 T <- exp(dx) T (one SE3
// parameter), 2 R <- exp(dx[3:6]) R, t += dx[0:3] (the reference pipeline's separate (SO3, translation) parameters,
// pyslam/pipelines/dense.py:185-186)
#define PS_PHOTO_NOUT 64
__global__ __launch_bounds__(256) void k_photo_finish(int nparts, const double* __restrict__ partials, int with_normal,
                                                       int update, double* __restrict__ pose, double* __restrict__ out)
{
    __shared__ double sp[8][PS_PHOTO_NACC];
    __shared__ double tot[PS_PHOTO_NACC];
    const int k = threadIdx.x & 31, g = threadIdx.x >> 5;          // 8 groups of 32 columns
    double v = 0.0;
    for (int p = g; p < nparts; p += 8) v += partials[(size_t)p * PS_PHOTO_NACC + k];
    sp[g][k] = v;
    __syncthreads();
    if (threadIdx.x < PS_PHOTO_NACC) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += sp[q][k];
        tot[k] = t;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    out[42] = tot[27]; out[43] = tot[28];
    if (!with_normal) return;
    double H[6][6], b[6], L[6][6], dx[6];
    int idx = 0;
    for (int c = 0; c < 6; ++c)
        for (int c2 = c; c2 < 6; ++c2) { H[c][c2] = tot[idx]; H[c2][c] = tot[idx]; ++idx; }
    for (int c = 0; c < 6; ++c) {
        b[c] = tot[21 + c];
        out[36 + c] = b[c];
        for (int c2 = 0; c2 < 6; ++c2) out[6 * c + c2] = H[c][c2];
    }
    bool ok = true;
    for (int j = 0; j < 6; ++j) {                                   // H = L L^T
        double d = H[j][j];
        for (int q = 0; q < j; ++q) d -= L[j][q] * L[j][q];
        if (!(d > 0.0)) { ok = false; break; }
        L[j][j] = sqrt(d);
        for (int i = j + 1; i < 6; ++i) {
            double s = H[i][j];
            for (int q = 0; q < j; ++q) s -= L[i][q] * L[j][q];
            L[i][j] = s / L[j][j];
        }
    }
    out[50] = ok ? 0.0 : 1.0;
    if (!ok) { for (int c = 0; c < 6; ++c) out[44 + c] = 0.0; return; }
    for (int i = 0; i < 6; ++i) {                                   // L y = b
        double s = b[i];
        for (int q = 0; q < i; ++q) s -= L[i][q] * dx[q];
        dx[i] = s / L[i][i];
    }
    for (int i = 5; i >= 0; --i) {                                  // L^T x = y
        double s = dx[i];
        for (int q = i + 1; q < 6; ++q) s -= L[q][i] * dx[q];
        dx[i] = s / L[i][i];
    }
    for (int c = 0; c < 6; ++c) out[44 + c] = dx[c];
    if (update) {
        Se3 T = se3_load(pose);
        if (update == 1) {
            T = se3_mul(se3_exp(dx), T);
        } else {
            const double rot[6] = {0.0, 0.0, 0.0, dx[3], dx[4], dx[5]};
            const Se3 E = se3_exp(rot);                              // pure rotation: E.t = 0
            Se3 Rn = T;
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j)
                    Rn.R[3 * i + j] = E.R[3 * i] * T.R[j] + E.R[3 * i + 1] * T.R[3 + j] + E.R[3 * i + 2] * T.R[6 + j];
            for (int i = 0; i < 3; ++i) Rn.t[i] = T.t[i] + dx[i];
            T = Rn;
        }
        se3_store(pose, T);
    }
    for (int c = 0; c < 12; ++c) out[51 + c] = pose[c];
}
