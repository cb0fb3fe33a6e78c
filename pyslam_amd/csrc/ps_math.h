// ps_math.h -- fp64 device math for the residual kernels: SE(2)/SE(3)
// log / exp / adjoint / compose, the stereo camera model and the robust losses.
//
// Restates, per element, what the reference computes through liegroups
// (third-party; conventions in SURVEY.md section 8c), pyslam/sensors/stereo_camera.py:100-134
// and pyslam/losses.py:8-214.  The CPU oracle of the same algebra is
// oracle/gn_oracle.py; tests compare the two at 1e-12.
#pragma once
#include <hip/hip_runtime.h>

#define PS_DEV __device__ __forceinline__

// 1 / sqrt(x) to double precision without a division: the hardware estimate (v_rsq_f64, ~2^-26 relative) and two
// Newton steps y <- y (1.5 - 0.5 x y^2).  IEEE sqrt + IEEE division cost ~50 instructions on this target; the landmark
// pass factors a 3 x 3 block per landmark with three roots and six quotients, a third of its instruction stream.
__device__ __forceinline__ double ps_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = y * (1.5 - hx * y * y);
    y = y * (1.5 - hx * y * y);
    return y;
}
#define PS_SMALL_ANGLE 1e-8   // np.isclose(angle, 0.)

// ---------------------------------------------------------------------------
// robust losses (ids shared with pyslam_amd/losses.py)
// ---------------------------------------------------------------------------
enum { PS_LOSS_L2 = 0, PS_LOSS_L1 = 1, PS_LOSS_CAUCHY = 2, PS_LOSS_HUBER = 3,
       PS_LOSS_TUKEY = 4, PS_LOSS_TDIST = 5 };

// (no contraction into fused multiply-adds here or in the residual chain of reproj_eval_s: the cost of an observation is then a
//  function of its inputs alone, whichever kernel the compiler inlines it into -- the landmark pass that sums the cost as it
//  goes and the cost-only pass give the same bits, ps_k_packed.h)
PS_DEV double ps_loss_rho(int id, double k, double x) {
#pragma clang fp contract(off)
    const double a = fabs(x);
    switch (id) {
    case PS_LOSS_L2: return 0.5 * x * x;
    case PS_LOSS_L1: return a;
    case PS_LOSS_CAUCHY: { const double q = x / k; return (0.5 * k * k) * log(1.0 + q * q); }
    case PS_LOSS_HUBER: return (a <= k) ? 0.5 * x * x : k * (a - 0.5 * k);
    case PS_LOSS_TUKEY: {
        const double c = k * k / 6.0;
        if (a > k) return c;
        const double q = x / k, u = 1.0 - q * q;
        return c * (1.0 - u * u * u);
    }
    default: return 0.5 * (k + 1.0) * log(1.0 + x * x / k);
    }
}

PS_DEV double ps_loss_weight(int id, double k, double x) {
    const double a = fabs(x);
    switch (id) {
    case PS_LOSS_L2: return 1.0;
    case PS_LOSS_L1: return (a <= PS_SMALL_ANGLE) ? __builtin_nan("") : 1.0 / a;  // losses.py:30-33
    case PS_LOSS_CAUCHY: { const double q = x / k; return 1.0 / (1.0 + q * q); }
    case PS_LOSS_HUBER: return (a <= k) ? 1.0 : k / a;
    case PS_LOSS_TUKEY: { const double q = x / k; return (a <= k) ? 1.0 - q * q : 0.0; }
    default: return (k + 1.0) / (k + x * x);
    }
}

// sqrt(w(x)): the IRLS scale of a residual component (reference pyslam/problem.py:351-360).  The loss id is wave-uniform (scalar
// branch), and the L2 case -- every observation of the BASELINE stereo BA -- returns its 1 without a double-precision square
// root of a run-time 1.0 (three of them per observation and pass: sqrt(1.0) == 1.0 exactly, so the results are the same bits)
PS_DEV double ps_loss_sqrt_weight(int id, double k, double x) {
    if (id == PS_LOSS_L2) return 1.0;
    return sqrt(ps_loss_weight(id, k, x));
}

// ---------------------------------------------------------------------------
// SE(3): R row-major (9) | t (3)
// ---------------------------------------------------------------------------
struct Se3 { double R[9]; double t[3]; };

PS_DEV Se3 se3_load(const double* __restrict__ p) {
    Se3 T;
#pragma unroll
    for (int i = 0; i < 9; ++i) T.R[i] = p[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) T.t[i] = p[9 + i];
    return T;
}

PS_DEV void se3_store(double* __restrict__ p, const Se3& T) {
#pragma unroll
    for (int i = 0; i < 9; ++i) p[i] = T.R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) p[9 + i] = T.t[i];
}

PS_DEV Se3 se3_inv(const Se3& T) {
    Se3 o;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) o.R[3 * i + j] = T.R[3 * j + i];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        o.t[i] = -(o.R[3 * i] * T.t[0] + o.R[3 * i + 1] * T.t[1] + o.R[3 * i + 2] * T.t[2]);
    return o;
}

PS_DEV Se3 se3_mul(const Se3& A, const Se3& B) {
    Se3 o;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
            o.R[3 * i + j] = A.R[3 * i] * B.R[j] + A.R[3 * i + 1] * B.R[3 + j] + A.R[3 * i + 2] * B.R[6 + j];
        o.t[i] = A.R[3 * i] * B.t[0] + A.R[3 * i + 1] * B.t[1] + A.R[3 * i + 2] * B.t[2] + A.t[i];
    }
    return o;
}

PS_DEV void se3_apply(const Se3& T, const double* __restrict__ p, double* __restrict__ out) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
        out[i] = T.R[3 * i] * p[0] + T.R[3 * i + 1] * p[1] + T.R[3 * i + 2] * p[2] + T.t[i];
}

// xi = [rho; phi] = log(T): phi = SO3.log(R), rho = J_l^-1(phi) t
PS_DEV void se3_log(const Se3& T, double* __restrict__ xi) {
    const double* R = T.R;
    double ca = 0.5 * (R[0] + R[4] + R[8]) - 0.5;
    ca = fmin(1.0, fmax(-1.0, ca));
    const double ang = acos(ca);
    double phi[3];
    if (fabs(ang) <= PS_SMALL_ANGLE) {
        phi[0] = R[7]; phi[1] = R[2]; phi[2] = R[3];          // vee(R - I)
    } else {
        const double f = 0.5 * ang / sin(ang);
        phi[0] = f * (R[7] - R[5]); phi[1] = f * (R[2] - R[6]); phi[2] = f * (R[3] - R[1]);
    }
    const double a = sqrt(phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2]);
    const double* t = T.t;
    if (fabs(a) <= PS_SMALL_ANGLE) {                          // I - 0.5 phi^
        xi[0] = t[0] - 0.5 * (-phi[2] * t[1] + phi[1] * t[2]);
        xi[1] = t[1] - 0.5 * (phi[2] * t[0] - phi[0] * t[2]);
        xi[2] = t[2] - 0.5 * (-phi[1] * t[0] + phi[0] * t[1]);
    } else {
        const double ax[3] = {phi[0] / a, phi[1] / a, phi[2] / a};
        const double half = 0.5 * a;
        const double hc = half / tan(half);
        const double at = ax[0] * t[0] + ax[1] * t[1] + ax[2] * t[2];
        // hc*t + (1-hc)*a a^T t - half * (a x t)
        xi[0] = hc * t[0] + (1.0 - hc) * ax[0] * at - half * (ax[1] * t[2] - ax[2] * t[1]);
        xi[1] = hc * t[1] + (1.0 - hc) * ax[1] * at - half * (ax[2] * t[0] - ax[0] * t[2]);
        xi[2] = hc * t[2] + (1.0 - hc) * ax[2] * at - half * (ax[0] * t[1] - ax[1] * t[0]);
    }
    xi[3] = phi[0]; xi[4] = phi[1]; xi[5] = phi[2];
}

// T = exp(xi): R = SO3.exp(phi), t = J_l(phi) rho
PS_DEV Se3 se3_exp(const double* __restrict__ xi) {
    Se3 T;
    const double* rho = xi;
    const double* phi = xi + 3;
    const double a = sqrt(phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2]);
    if (fabs(a) <= PS_SMALL_ANGLE) {
        T.R[0] = 1.0;     T.R[1] = -phi[2]; T.R[2] = phi[1];
        T.R[3] = phi[2];  T.R[4] = 1.0;     T.R[5] = -phi[0];
        T.R[6] = -phi[1]; T.R[7] = phi[0];  T.R[8] = 1.0;
        T.t[0] = rho[0] + 0.5 * (-phi[2] * rho[1] + phi[1] * rho[2]);
        T.t[1] = rho[1] + 0.5 * (phi[2] * rho[0] - phi[0] * rho[2]);
        T.t[2] = rho[2] + 0.5 * (-phi[1] * rho[0] + phi[0] * rho[1]);
        return T;
    }
    const double ax[3] = {phi[0] / a, phi[1] / a, phi[2] / a};
    const double s = sin(a), c = cos(a);
    const double omc = 1.0 - c;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            T.R[3 * i + j] = (i == j ? c : 0.0) + omc * ax[i] * ax[j];
    T.R[1] += -s * ax[2]; T.R[2] += s * ax[1];
    T.R[3] += s * ax[2];  T.R[5] += -s * ax[0];
    T.R[6] += -s * ax[1]; T.R[7] += s * ax[0];
    const double sa = s / a, ca = omc / a;
    const double ar = ax[0] * rho[0] + ax[1] * rho[1] + ax[2] * rho[2];
    T.t[0] = sa * rho[0] + (1.0 - sa) * ax[0] * ar + ca * (ax[1] * rho[2] - ax[2] * rho[1]);
    T.t[1] = sa * rho[1] + (1.0 - sa) * ax[1] * ar + ca * (ax[2] * rho[0] - ax[0] * rho[2]);
    T.t[2] = sa * rho[2] + (1.0 - sa) * ax[2] * ar + ca * (ax[0] * rho[1] - ax[1] * rho[0]);
    return T;
}

// Ad(T)[r][c], 6x6: [[C, t^ C], [0, C]]
// the same entry from the packed pose in MEMORY (LDS or global: R row-major | t).  For a column index that differs from lane to
// lane: indexing the register copy with it put the whole pose into scratch (the only scratch users of the library were the two
// kernels that do this: k_factor_pass, k_block_jacobi_factor)
PS_DEV double se3_adjoint_entry_mem(const double* __restrict__ p, int r, int c) {
    if (r >= 3) return (c >= 3) ? p[3 * (r - 3) + (c - 3)] : 0.0;
    if (c < 3) return p[3 * r + c];
    const int j = c - 3;
    const double* t = p + 9;
    if (r == 0) return -t[2] * p[3 + j] + t[1] * p[6 + j];
    if (r == 1) return t[2] * p[j] - t[0] * p[6 + j];
    return -t[1] * p[j] + t[0] * p[3 + j];
}

PS_DEV double se3_adjoint_entry(const Se3& T, int r, int c) {
    if (r >= 3) return (c >= 3) ? T.R[3 * (r - 3) + (c - 3)] : 0.0;
    if (c < 3) return T.R[3 * r + c];
    const int j = c - 3;
    const double* t = T.t;
    // row r of t^ times column j of C
    if (r == 0) return -t[2] * T.R[3 + j] + t[1] * T.R[6 + j];
    if (r == 1) return t[2] * T.R[j] - t[0] * T.R[6 + j];
    return -t[1] * T.R[j] + t[0] * T.R[3 + j];
}

// ---------------------------------------------------------------------------
// SE(2): R row-major (4) | t (2), xi = [rho (2); phi]
// ---------------------------------------------------------------------------
struct Se2 { double R[4]; double t[2]; };

PS_DEV Se2 se2_load(const double* __restrict__ p) {
    Se2 T;
#pragma unroll
    for (int i = 0; i < 4; ++i) T.R[i] = p[i];
    T.t[0] = p[4]; T.t[1] = p[5];
    return T;
}

PS_DEV void se2_store(double* __restrict__ p, const Se2& T) {
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = T.R[i];
    p[4] = T.t[0]; p[5] = T.t[1];
}

PS_DEV Se2 se2_inv(const Se2& T) {
    Se2 o;
    o.R[0] = T.R[0]; o.R[1] = T.R[2]; o.R[2] = T.R[1]; o.R[3] = T.R[3];
    o.t[0] = -(o.R[0] * T.t[0] + o.R[1] * T.t[1]);
    o.t[1] = -(o.R[2] * T.t[0] + o.R[3] * T.t[1]);
    return o;
}

PS_DEV Se2 se2_mul(const Se2& A, const Se2& B) {
    Se2 o;
    o.R[0] = A.R[0] * B.R[0] + A.R[1] * B.R[2]; o.R[1] = A.R[0] * B.R[1] + A.R[1] * B.R[3];
    o.R[2] = A.R[2] * B.R[0] + A.R[3] * B.R[2]; o.R[3] = A.R[2] * B.R[1] + A.R[3] * B.R[3];
    o.t[0] = A.R[0] * B.t[0] + A.R[1] * B.t[1] + A.t[0];
    o.t[1] = A.R[2] * B.t[0] + A.R[3] * B.t[1] + A.t[1];
    return o;
}

PS_DEV void se2_log(const Se2& T, double* __restrict__ xi) {
    const double phi = atan2(T.R[2], T.R[0]);
    double a, b;                     // J_l^-1 = a I - b [[0,-1],[1,0]]
    if (fabs(phi) <= PS_SMALL_ANGLE) { a = 1.0; b = 0.5 * phi; }
    else { const double half = 0.5 * phi; a = half / tan(half); b = half; }
    xi[0] = a * T.t[0] + b * T.t[1];
    xi[1] = -b * T.t[0] + a * T.t[1];
    xi[2] = phi;
}

PS_DEV Se2 se2_exp(const double* __restrict__ xi) {
    Se2 T;
    const double phi = xi[2];
    const double c = cos(phi), s = sin(phi);
    T.R[0] = c; T.R[1] = -s; T.R[2] = s; T.R[3] = c;
    double a, b;                     // J_l = a I + b [[0,-1],[1,0]]
    if (fabs(phi) <= PS_SMALL_ANGLE) { a = 1.0; b = 0.5 * phi; }
    else { a = s / phi; b = (1.0 - c) / phi; }
    T.t[0] = a * xi[0] - b * xi[1];
    T.t[1] = b * xi[0] + a * xi[1];
    return T;
}

// Ad(T)[r][c], 3x3: [[C, (y, -x)^T], [0, 1]]
PS_DEV double se2_adjoint_entry_mem(const double* __restrict__ p, int r, int c) {
    if (r == 2) return (c == 2) ? 1.0 : 0.0;
    if (c < 2) return p[2 * r + c];
    return (r == 0) ? p[5] : -p[4];
}

PS_DEV double se2_adjoint_entry(const Se2& T, int r, int c) {
    if (r == 2) return (c == 2) ? 1.0 : 0.0;
    if (c < 2) return T.R[2 * r + c];
    return (r == 0) ? T.t[1] : -T.t[0];
}

// Group traits so the pose-graph kernels are written once.
template <int D> struct PoseOps;
template <> struct PoseOps<6> {
    typedef Se3 T;
    static constexpr int W = 12;
    static PS_DEV T load(const double* p) { return se3_load(p); }
    static PS_DEV void store(double* p, const T& x) { se3_store(p, x); }
    static PS_DEV T inv(const T& x) { return se3_inv(x); }
    static PS_DEV T mul(const T& a, const T& b) { return se3_mul(a, b); }
    static PS_DEV void log(const T& x, double* xi) { se3_log(x, xi); }
    static PS_DEV T exp(const double* xi) { return se3_exp(xi); }
    static PS_DEV double adj(const T& x, int r, int c) { return se3_adjoint_entry(x, r, c); }
    static PS_DEV double adj_mem(const double* p, int r, int c) { return se3_adjoint_entry_mem(p, r, c); }
};
template <> struct PoseOps<3> {
    typedef Se2 T;
    static constexpr int W = 6;
    static PS_DEV T load(const double* p) { return se2_load(p); }
    static PS_DEV void store(double* p, const T& x) { se2_store(p, x); }
    static PS_DEV T inv(const T& x) { return se2_inv(x); }
    static PS_DEV T mul(const T& a, const T& b) { return se2_mul(a, b); }
    static PS_DEV void log(const T& x, double* xi) { se2_log(x, xi); }
    static PS_DEV T exp(const double* xi) { return se2_exp(xi); }
    static PS_DEV double adj(const T& x, int r, int c) { return se2_adjoint_entry(x, r, c); }
    static PS_DEV double adj_mem(const double* p, int r, int c) { return se2_adjoint_entry_mem(p, r, c); }
};

// ---------------------------------------------------------------------------
// stereo reprojection block:  r = S (project(R p + t) - obs)
//   J_pose = S Jc [I | -pc^]  (3x6),  J_point = S Jc R  (3x3)
// followed by element-wise IRLS scaling (pyslam/problem.py:351-360).
// ---------------------------------------------------------------------------
// cam_type: 0 = stereo (u, v, d = fu b / z), 1 = RGB-D (u, v, z)  -- reference
// pyslam/sensors/stereo_camera.py:100-134 and rgbd_camera.py:96-135
struct ObsGroup { double cu, cv, fu, fv, b; double S[9]; int loss_id; int cam_type; double loss_k; };

struct ReprojEval {
    double r[3];        // sqrt(w) * r
    double Jp[18];      // sqrt(w) * J_pose, row-major 3x6
    double Jl[9];       // sqrt(w) * J_point, row-major 3x3
    double cost;        // sum rho(r_k)
    double pc[3];       // the point in the camera frame, T p (J_pose = S Jc [I | -pc^])
};

// `Sv`: the 3 x 3 stiffness (row-major).  Normally the group's own (g.S: scalar registers, shared by the wave); problems
// with one stiffness per observation (reference reprojection_residual.py:8-11 takes any) pass a per-lane copy instead.
template <bool WITH_JP, bool WITH_JL>
PS_DEV void reproj_eval_s(const Se3& T, const double* __restrict__ pw, const double* __restrict__ uvd,
                          const ObsGroup& g, const double* __restrict__ Sv, ReprojEval& o) {
    double pc[3], rr[3], iz;
    const bool rgbd = g.cam_type == 1;
    {   // the chain the cost depends on: explicit fused multiply-adds, no contraction left to the compiler (see ps_loss_rho)
#pragma clang fp contract(off)
#pragma unroll
        for (int i = 0; i < 3; ++i)
            pc[i] = __builtin_fma(T.R[3 * i + 2], pw[2], __builtin_fma(T.R[3 * i + 1], pw[1], __builtin_fma(T.R[3 * i], pw[0], T.t[i])));
        iz = 1.0 / pc[2];
        const double e0 = __builtin_fma(g.fu * pc[0], iz, g.cu - uvd[0]);
        const double e1 = __builtin_fma(g.fv * pc[1], iz, g.cv - uvd[1]);
        const double e2 = rgbd ? pc[2] - uvd[2] : __builtin_fma(g.fu * g.b, iz, -uvd[2]);
#pragma unroll
        for (int i = 0; i < 3; ++i) rr[i] = __builtin_fma(Sv[3 * i + 2], e2, __builtin_fma(Sv[3 * i + 1], e1, Sv[3 * i] * e0));
        o.cost = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) o.cost += ps_loss_rho(g.loss_id, g.loss_k, rr[i]);
    }
    o.pc[0] = pc[0]; o.pc[1] = pc[1]; o.pc[2] = pc[2];
    double s[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double ri = rr[i];
        s[i] = ps_loss_sqrt_weight(g.loss_id, g.loss_k, ri);
        o.r[i] = s[i] * ri;
    }
    if (!WITH_JP && !WITH_JL) return;
    const double iz2 = iz * iz;
    const double j00 = g.fu * iz, j02 = -g.fu * pc[0] * iz2;
    const double j11 = g.fv * iz, j12 = -g.fv * pc[1] * iz2;
    const double j22 = rgbd ? 1.0 : -g.fu * g.b * iz2;
    double SJ[9];                       // diag(s) * S * Jc
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        SJ[3 * i] = s[i] * (Sv[3 * i] * j00);
        SJ[3 * i + 1] = s[i] * (Sv[3 * i + 1] * j11);
        SJ[3 * i + 2] = s[i] * (Sv[3 * i] * j02 + Sv[3 * i + 1] * j12 + Sv[3 * i + 2] * j22);
    }
    if (WITH_JP) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double a = SJ[3 * i], b = SJ[3 * i + 1], c = SJ[3 * i + 2];
            o.Jp[6 * i] = a; o.Jp[6 * i + 1] = b; o.Jp[6 * i + 2] = c;
            o.Jp[6 * i + 3] = -b * pc[2] + c * pc[1];      // SJ * (-pc)^
            o.Jp[6 * i + 4] = a * pc[2] - c * pc[0];
            o.Jp[6 * i + 5] = -a * pc[1] + b * pc[0];
        }
    }
    if (WITH_JL) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                o.Jl[3 * i + j] = SJ[3 * i] * T.R[j] + SJ[3 * i + 1] * T.R[3 + j] + SJ[3 * i + 2] * T.R[6 + j];
    }
}

template <bool WITH_JP, bool WITH_JL>
PS_DEV void reproj_eval(const Se3& T, const double* __restrict__ pw, const double* __restrict__ uvd,
                        const ObsGroup& g, ReprojEval& o) {
    reproj_eval_s<WITH_JP, WITH_JL>(T, pw, uvd, g, g.S, o);
}

// The (camera, stiffness, loss) group of an observation is almost always the same for a whole wave: it is read through
// a wave-uniform index -- scalar loads into SGPRs instead of 17 per-lane loads of the same 128 bytes into 34 VGPRs.
// Mixed waves go round a waterfall loop, one pass per distinct group (the lanes of the first active lane's group
// evaluate and leave), so there is ONE copy of the evaluation and it never holds a group in vector registers.
template <bool WITH_JP, bool WITH_JL>
PS_DEV void reproj_eval_grp(const Se3& T, const double* __restrict__ pw, const double* __restrict__ uvd,
                            const ObsGroup* __restrict__ groups, int grp, ReprojEval& o) {
    for (;;) {
        const int g0 = __builtin_amdgcn_readfirstlane(grp);
        if (grp == g0) { reproj_eval<WITH_JP, WITH_JL>(T, pw, uvd, groups[g0], o); break; }
    }
}

// "Wide" problems: more than 255 distinct (camera, stiffness, loss) rows -- typically one stiffness per observation.  The
// 8-bit group field then names the (camera, loss) CLASS (still wave-uniform, scalar registers) and the stiffness comes
// from a table through a per-observation index (a parallel int32 column): nine per-lane loads, nine more registers --
// in instantiations of their own (template flag WIDE), so the common path keeps its register budget.
struct ObsWide { const int32_t* sidx; const double* stiff; };

template <bool WITH_JP, bool WITH_JL, bool WIDE>
PS_DEV void reproj_eval_obs(const Se3& T, const double* __restrict__ pw, const double* __restrict__ uvd,
                            const ObsGroup* __restrict__ groups, int grp, const ObsWide& wide, long i, ReprojEval& o) {
    if (!WIDE) { reproj_eval_grp<WITH_JP, WITH_JL>(T, pw, uvd, groups, grp, o); return; }
    double Sl[9];
    const double* sp = wide.stiff + 9 * (size_t)wide.sidx[i];
#pragma unroll
    for (int k = 0; k < 9; ++k) Sl[k] = sp[k];
    for (;;) {
        const int g0 = __builtin_amdgcn_readfirstlane(grp);
        if (grp == g0) { reproj_eval_s<WITH_JP, WITH_JL>(T, pw, uvd, groups[g0], Sl, o); break; }
    }
}
