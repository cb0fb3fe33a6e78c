// ps_k_linearize.h -- landmark pass, pose pass, Schur pair / combine kernels, pose-pose and prior factors.
// Part of ps_kernels.h (included from there, in this order; not a stand-alone header).
#pragma once

// ---------------------------------------------------------------------------
// landmark pass: 16 lanes cooperate on one landmark (4 landmarks per wave), one observation
// per lane: loads of the 32-byte records and stores of the 128-byte Z rows are contiguous
// across lanes, residual + both Jacobians are evaluated ONCE, and H_ll / b_l are reduced with
// a 4-step xor butterfly inside the 16-lane group (fixed order => deterministic).
// Landmarks with more than 16 observations loop (lane j takes observations j, j+16, ...) and
// re-evaluate in a second sweep to emit Z.
// ---------------------------------------------------------------------------
#define PS_LM_GROUP 16

PS_DEV double group16_sum(double v) {          // a 16-lane group is exactly one DPP row
    v = dpp_shift_add<0x111, 0xf, 0xf>(v);
    v = dpp_shift_add<0x112, 0xf, 0xf>(v);
    v = dpp_shift_add<0x114, 0xf, 0xe>(v);
    v = dpp_shift_add<0x118, 0xf, 0xc>(v);      // lane 15 of the row holds the group total
    return __shfl(v, (int)(threadIdx.x & 63) | 15, 64);
}

// ---------------------------------------------------------------------------
// The eliminated-landmark row of an observation, Z_i = W_i C^-T (6 x 3, W_i = J~p^T J~l, H_ll = C C^T), in 12 numbers:
// J~p = A [I | -pc^] with A = diag(sqrt w) S Jc (3 x 3) and pc = T p the point in the camera frame, so
//     J~p^T = [I; pc^] A^T     and     Z_i = [I; pc^] (A^T J~l C^-T) = [M; pc^ M],   M = A^T J~l C^-T  (3 x 3).
// HBM holds (M, pc) per observation, padded to ONE aligned 128-byte line ("Z row": M row-major (9) | pc (3) | reduced
// index of the pose as a double (-1: constant pose) | 3 unused) instead of the 18 entries of Z (144 B, two lines per
// gather): the Schur pair kernel fetches one line per row, the back-substitution reads 128 B per observation and no
// observation record.  Every consumer expands the lower half pc^ M with the same three cross products (zrow_expand),
// so the diagonal blocks (pose pass), the off-diagonal blocks (pair kernel) and the back-substitution see one Z.
// ---------------------------------------------------------------------------
#define PS_ZROW 16

// M = (top three rows of J~p^T) J~l C^-T; (M00 .. M22) = C^-1 (lower triangular)
PS_DEV void lm_emit_m(const ReprojEval& ev, double M00, double M10, double M11, double M20, double M21,
                      double M22, double* __restrict__ m) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double w0 = ev.Jp[a] * ev.Jl[0] + ev.Jp[6 + a] * ev.Jl[3] + ev.Jp[12 + a] * ev.Jl[6];
        const double w1 = ev.Jp[a] * ev.Jl[1] + ev.Jp[6 + a] * ev.Jl[4] + ev.Jp[12 + a] * ev.Jl[7];
        const double w2 = ev.Jp[a] * ev.Jl[2] + ev.Jp[6 + a] * ev.Jl[5] + ev.Jp[12 + a] * ev.Jl[8];
        m[3 * a] = w0 * M00;
        m[3 * a + 1] = w0 * M10 + w1 * M11;
        m[3 * a + 2] = w0 * M20 + w1 * M21 + w2 * M22;
    }
}

// l = pc^ m (3 x 3, row-major): column k of l is pc x (column k of m)
PS_DEV void zrow_cross(const double* __restrict__ m, const double* __restrict__ pc, double* __restrict__ l) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        l[k] = pc[1] * m[6 + k] - pc[2] * m[3 + k];
        l[3 + k] = pc[2] * m[k] - pc[0] * m[6 + k];
        l[6 + k] = pc[0] * m[3 + k] - pc[1] * m[k];
    }
}

// z (6 x 3, row-major) = [m; pc^ m]
PS_DEV void zrow_expand(const double* __restrict__ m, const double* __restrict__ pc, double* __restrict__ z) {
#pragma unroll
    for (int k = 0; k < 9; ++k) z[k] = m[k];
    zrow_cross(m, pc, z + 9);
}

#ifndef PS_LM_WAVES
#define PS_LM_WAVES 3
#endif
template <bool WIDE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PS_LM_WAVES, 8))) void k_landmark_pass(
    int nv, const int32_t* __restrict__ lm_ptr, const int32_t* __restrict__ lm_point,
    const LObs* __restrict__ lobs, const double* __restrict__ poses,
    const double* __restrict__ points, const int32_t* __restrict__ pose_rid,
    const ObsGroup* __restrict__ groups, double lambda,
    double* __restrict__ Z, double* __restrict__ Cinv, double* __restrict__ cvec,
    int32_t* __restrict__ status, int ablate, ObsWide wide)
{
    const int v = blockIdx.x * (blockDim.x / PS_LM_GROUP) + threadIdx.x / PS_LM_GROUP;
    const int sub = threadIdx.x & (PS_LM_GROUP - 1);
    const bool live = v < nv;                       // whole 16-lane groups are live or not
    int b = 0, e = 0;
    double pw[3] = {0.0, 0.0, 0.0};
    if (live) {
        b = lm_ptr[v]; e = lm_ptr[v + 1];
        const int pt = lm_point[v];
        pw[0] = points[3 * pt]; pw[1] = points[3 * pt + 1]; pw[2] = points[3 * pt + 2];
    }
    const bool single = (e - b) <= PS_LM_GROUP;     // the common case: one observation per lane

    double H00 = 0, H10 = 0, H11 = 0, H20 = 0, H21 = 0, H22 = 0, b0 = 0, b1 = 0, b2 = 0;
    ReprojEval ev;
    bool have = false, variable_pose = false;
    int rid_of_obs = -1;
    for (int i = b + sub; i < e; i += PS_LM_GROUP) {
        const LObs o = lobs[i];
        const int pose = PS_POSE_OF(o);
        const Se3 T = se3_load(poses + 12 * pose);
        rid_of_obs = pose_rid[pose];
        variable_pose = rid_of_obs >= 0;
        reproj_eval_obs<true, true, WIDE>(T, pw, &o.u, groups, PS_GRP_OF(o), wide, i, ev);
        have = true;
        const double* J = ev.Jl;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            H00 += J[3 * k] * J[3 * k];
            H10 += J[3 * k + 1] * J[3 * k];
            H11 += J[3 * k + 1] * J[3 * k + 1];
            H20 += J[3 * k + 2] * J[3 * k];
            H21 += J[3 * k + 2] * J[3 * k + 1];
            H22 += J[3 * k + 2] * J[3 * k + 2];
            b0 -= J[3 * k] * ev.r[k];
            b1 -= J[3 * k + 1] * ev.r[k];
            b2 -= J[3 * k + 2] * ev.r[k];
        }
    }
    H00 = group16_sum(H00); H10 = group16_sum(H10); H11 = group16_sum(H11);
    H20 = group16_sum(H20); H21 = group16_sum(H21); H22 = group16_sum(H22);
    b0 = group16_sum(b0); b1 = group16_sum(b1); b2 = group16_sum(b2);

    const double damp = 1.0 + lambda;
    H00 *= damp; H11 *= damp; H22 *= damp;
    // H_ll = C C^T, M = C^-1 (lower): the reciprocal roots first, every quotient a product (ps_rsqrt: no division)
    const double M00 = ps_rsqrt(H00);
    const double l10 = H10 * M00, l20 = H20 * M00;
    const double d1 = H11 - l10 * l10;
    const double M11 = ps_rsqrt(d1);
    const double l21 = (H21 - l20 * l10) * M11;
    const double d2 = H22 - l20 * l20 - l21 * l21;
    const double M22 = ps_rsqrt(d2);
    const double M10 = -l10 * M00 * M11;
    const double M21 = -l21 * M11 * M22;
    const double M20 = -(l20 * M00 + l21 * M10) * M22;
    if (live && sub == 0) {
        if (!(H00 > 0.0) || !(d1 > 0.0) || !(d2 > 0.0)) atomicAdd(&status[ST_LM_FAIL], 1);
        double* ci = Cinv + 6 * (size_t)v;
        ci[0] = M00; ci[1] = M10; ci[2] = M11; ci[3] = M20; ci[4] = M21; ci[5] = M22;
        double* cv = cvec + 3 * (size_t)v;
        cv[0] = M00 * b0;
        cv[1] = M10 * b0 + M11 * b1;
        cv[2] = M20 * b0 + M21 * b1 + M22 * b2;
    }
    // ---- Z rows (one 128-byte line each).  Common case (every landmark of the wave has <= 16 observations): the
    // wave's rows are one contiguous range of Z, so they are staged in LDS and stored as whole 16-byte pieces by
    // consecutive lanes (1 KB per store instruction).
    __shared__ __attribute__((aligned(16))) double zst[4][64 * PS_ZROW];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (__ballot(single || !live) == ~0ull && !(ablate & 2)) {
        const int row0 = __shfl(b, 0, 64);                       // dead groups carry b = e = 0
        const int eend = max(max(__shfl(e, 0, 64), __shfl(e, 16, 64)), max(__shfl(e, 32, 64), __shfl(e, 48, 64)));
        const int nrows = eend - row0;
        if (have) {
            double z[PS_ZROW];
#pragma unroll
            for (int k = 0; k < PS_ZROW; ++k) z[k] = 0.0;        // rows of constant poses are never read
            z[12] = -1.0;
            if (variable_pose) {
                lm_emit_m(ev, M00, M10, M11, M20, M21, M22, z);
                z[9] = ev.pc[0]; z[10] = ev.pc[1]; z[11] = ev.pc[2];
                z[12] = (double)rid_of_obs;
            }
            double2* dst = reinterpret_cast<double2*>(&zst[wv][PS_ZROW * (b + sub - row0)]);
#pragma unroll
            for (int k = 0; k < PS_ZROW / 2; ++k) dst[k] = make_double2(z[2 * k], z[2 * k + 1]);
        }
        __builtin_amdgcn_wave_barrier();
        if (!(ablate & 1)) {
            const double2* src = reinterpret_cast<const double2*>(zst[wv]);
            double2* out = reinterpret_cast<double2*>(Z + PS_ZROW * (size_t)row0);
            for (int k = lane; k < nrows * (PS_ZROW / 2); k += 64) out[k] = src[k];
        }
        return;
    }
    if (!live) return;
    for (int i = b + sub; i < e; i += PS_LM_GROUP) {             // long tracks: a second sweep re-evaluates
        const LObs o = lobs[i];
        const int pose = PS_POSE_OF(o);
        double* z = Z + PS_ZROW * (size_t)i;
        const int rid = pose_rid[pose];
        z[12] = (double)rid;
        if (rid < 0 || (ablate & 1)) continue;
        if (!single) {
            const Se3 T = se3_load(poses + 12 * pose);
            reproj_eval_obs<true, true, WIDE>(T, pw, &o.u, groups, PS_GRP_OF(o), wide, i, ev);
        }
        lm_emit_m(ev, M00, M10, M11, M20, M21, M22, z);
        z[9] = ev.pc[0]; z[10] = ev.pc[1]; z[11] = ev.pc[2];
    }
}

// ---------------------------------------------------------------------------
// pose pass: 33 sums per chunk = 21 (upper J^T J - Z Z^T) + 6 (g) + 6 (diag J^T J, for damping)
// ---------------------------------------------------------------------------
// One workgroup per chunk of one pose's observations (256, or 1024 = four per thread on big
// problems so that the 33 wave reductions are paid once per four observations).  Observation
// records come from a pose-sorted copy (contiguous) that carries the landmark slot, and the pose
// is uniform per workgroup.  The Z row of an observation is NOT read back from HBM (144 B each,
// scattered: that read alone cost 15 of this kernel's 37 us): it is recomputed in registers from
// the Jacobians this kernel evaluates anyway and the landmark's 48-byte factor C^-1 (an L2-resident
// table) -- lm_emit_m + zrow_expand on the same inputs, so the values are those every consumer of the stored row forms.
// acc (upper triangle of a symmetric 6 x 6, row by row: 21 entries) += K^T E K with K = [I | -pc^] and E symmetric 3 x 3
// (e = 00 01 02 11 12 22):  [[E, F], [F^T, H]],  F = -E pc^,  H = pc^ F
PS_DEV void pose_sandwich(const double* __restrict__ e, const double* __restrict__ pc, double* __restrict__ acc) {
    const double x = pc[0], y = pc[1], z = pc[2];
    const double E[3][3] = {{e[0], e[1], e[2]}, {e[1], e[3], e[4]}, {e[2], e[4], e[5]}};
    double F[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        F[i][0] = E[i][2] * y - E[i][1] * z;
        F[i][1] = E[i][0] * z - E[i][2] * x;
        F[i][2] = E[i][1] * x - E[i][0] * y;
    }
    acc[0] += e[0]; acc[1] += e[1]; acc[2] += e[2]; acc[3] += F[0][0]; acc[4] += F[0][1]; acc[5] += F[0][2];
    acc[6] += e[3]; acc[7] += e[4]; acc[8] += F[1][0]; acc[9] += F[1][1]; acc[10] += F[1][2];
    acc[11] += e[5]; acc[12] += F[2][0]; acc[13] += F[2][1]; acc[14] += F[2][2];
    acc[15] += y * F[2][0] - z * F[1][0];
    acc[16] += y * F[2][1] - z * F[1][1];
    acc[17] += y * F[2][2] - z * F[1][2];
    acc[18] += z * F[0][1] - x * F[2][1];
    acc[19] += z * F[0][2] - x * F[2][2];
    acc[20] += x * F[1][2] - y * F[0][2];
}

#define PS_NPOSE_ACC 33
#ifndef PS_POSE_TRANSPOSE
#define PS_POSE_TRANSPOSE 1
#endif
typedef const __attribute__((address_space(1))) void* ps_gptr_t;
typedef __attribute__((address_space(3))) void* ps_lptr_t;

#ifndef PS_POSE_WAVES
#define PS_POSE_WAVES 3
#endif
template <bool WIDE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PS_POSE_WAVES, 8))) void k_pose_pass(
    const PItem* __restrict__ items,
    const LObs* __restrict__ pobs /* observation records in pose order, landmark slot + 1 in the pose bits */,
    const double* __restrict__ poses, const double* __restrict__ points,
    const ObsGroup* __restrict__ groups, const double* __restrict__ Cinv,
    const double* __restrict__ cvec, double* __restrict__ partial, int want_diag /* lambda != 0: the six damping sums too */,
    ObsWide wide, int nitems, int per_xcd /* > 0: XCD-major order of the items (round 6), 0: item = workgroup */)
{
    __shared__ double red[4][PS_NPOSE_ACC];
#if PS_POSE_TRANSPOSE
    __shared__ double tr[4][64 * 17];
#endif
    // Round 6: workgroup b runs on XCD b % 8 (observed dispatch rule; affects speed only).  The items are in pose order and a pose's
    // observations gather points / C^-1 / c of the landmarks it sees -- with item = workgroup every XCD walked ALL poses and pulled
    // the whole landmark table through its own L2 (C3: 60 MB of fabric traffic for 16 MB of records, round-5 verdict weak #4).  With
    // the item list cut into eight contiguous ranges, one per XCD, an XCD sees one stretch of the trajectory and its landmarks once.
    const int item = per_xcd > 0 ? (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (item >= nitems) return;
    const PItem it = items[item];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const Se3 T = se3_load(poses + 12 * (size_t)it.pad);          // pad = pose table index of this chunk
    double acc[PS_NPOSE_ACC];
#pragma unroll
    for (int k = 0; k < PS_NPOSE_ACC; ++k) acc[k] = 0.0;
    for (int i = it.start + threadIdx.x; i < it.end; i += 256) {
        const LObs o = pobs[i];
        const int v = PS_POSE_OF(o) - 1;                           // -1: constant landmark, no Schur term
        const double pw[3] = {points[3 * (size_t)o.point], points[3 * (size_t)o.point + 1], points[3 * (size_t)o.point + 2]};
        double m[6] = {0, 0, 0, 0, 0, 0}, c0 = 0.0, c1 = 0.0, c2 = 0.0;
        if (v >= 0) {
            const double* ci = Cinv + 6 * (size_t)v;
#pragma unroll
            for (int k = 0; k < 6; ++k) m[k] = ci[k];
            c0 = cvec[3 * (size_t)v]; c1 = cvec[3 * (size_t)v + 1]; c2 = cvec[3 * (size_t)v + 2];
        }
        ReprojEval ev;
        reproj_eval_obs<true, true, WIDE>(T, pw, &o.u, groups, PS_GRP_OF(o), wide, i, ev);   // (only the translational columns of J~p = A are used below)
        // J~p = A K with K = [I | -pc^], and Z = K^T M: the observation's contribution is the 3 x 3 sandwich
        //   J~p^T J~p - Z Z^T = K^T (A^T A - M M^T) K,      -J~p^T r~ - Z c = -K^T (A^T r~ + M c)
        // -- 160 multiply-adds instead of the 290 of the two 6 x 6 products formed entry by entry
        double G[6], t3[3];                                  // G = A^T A (00 01 02 11 12 22), t = A^T r~ (+ M c)
        {
            const double* J = ev.Jp;
            G[0] = J[0] * J[0] + J[6] * J[6] + J[12] * J[12];
            G[1] = J[0] * J[1] + J[6] * J[7] + J[12] * J[13];
            G[2] = J[0] * J[2] + J[6] * J[8] + J[12] * J[14];
            G[3] = J[1] * J[1] + J[7] * J[7] + J[13] * J[13];
            G[4] = J[1] * J[2] + J[7] * J[8] + J[13] * J[14];
            G[5] = J[2] * J[2] + J[8] * J[8] + J[14] * J[14];
#pragma unroll
            for (int a = 0; a < 3; ++a) t3[a] = J[a] * ev.r[0] + J[6 + a] * ev.r[1] + J[12 + a] * ev.r[2];
        }
        if (want_diag) {                                     // diag(J~p^T J~p) for Marquardt damping: the sandwich of G alone
            double d21[21];
#pragma unroll
            for (int k = 0; k < 21; ++k) d21[k] = 0.0;
            pose_sandwich(G, ev.pc, d21);
            acc[27] += d21[0]; acc[28] += d21[6]; acc[29] += d21[11]; acc[30] += d21[15]; acc[31] += d21[18]; acc[32] += d21[20];
        }
        if (v >= 0) {
            double m9[9];
            lm_emit_m(ev, m[0], m[1], m[2], m[3], m[4], m[5], m9);
            G[0] -= m9[0] * m9[0] + m9[1] * m9[1] + m9[2] * m9[2];
            G[1] -= m9[0] * m9[3] + m9[1] * m9[4] + m9[2] * m9[5];
            G[2] -= m9[0] * m9[6] + m9[1] * m9[7] + m9[2] * m9[8];
            G[3] -= m9[3] * m9[3] + m9[4] * m9[4] + m9[5] * m9[5];
            G[4] -= m9[3] * m9[6] + m9[4] * m9[7] + m9[5] * m9[8];
            G[5] -= m9[6] * m9[6] + m9[7] * m9[7] + m9[8] * m9[8];
#pragma unroll
            for (int a = 0; a < 3; ++a) t3[a] += m9[3 * a] * c0 + m9[3 * a + 1] * c1 + m9[3 * a + 2] * c2;
        }
        pose_sandwich(G, ev.pc, acc);
        acc[21] -= t3[0]; acc[22] -= t3[1]; acc[23] -= t3[2];
        acc[24] -= ev.pc[1] * t3[2] - ev.pc[2] * t3[1];
        acc[25] -= ev.pc[2] * t3[0] - ev.pc[0] * t3[2];
        acc[26] -= ev.pc[0] * t3[1] - ev.pc[1] * t3[0];
    }
#if PS_POSE_TRANSPOSE
    // 33 sums over the wave's 64 lanes through LDS: every lane stores its accumulators ([lane][17], two halves), lane k
    // adds up column k (64 conflict-free reads, fixed order) -- ~290 instructions per wave instead of the ~730 of 33 DPP
    // butterflies, which cost as much as the evaluation itself when a thread holds one observation (C3)
    {
        double* trw = tr[w];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int k0 = 17 * h2, nk = h2 ? PS_NPOSE_ACC - 17 : 17;
#pragma unroll
            for (int k = 0; k < 17; ++k) if (k < nk) trw[lane * 17 + k] = acc[k0 + k];
            __builtin_amdgcn_wave_barrier();
            if (lane < nk) {
                double s = 0.0;
#pragma unroll 8
                for (int l = 0; l < 64; ++l) s += trw[l * 17 + lane];
                red[w][k0 + lane] = s;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
#else
#pragma unroll
    for (int k = 0; k < PS_NPOSE_ACC; ++k) {
        const double s = wave_sum(acc[k]);
        if (lane == 0) red[w][k] = s;
    }
#endif
    __syncthreads();
    if (threadIdx.x < PS_NPOSE_ACC)
        partial[(size_t)item * PS_NPOSE_ACC + threadIdx.x] =
            ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// one wave (64 lanes, 33 active) per reduced pose: chunk partials -> diagonal S block, g
PS_DEV void pose_finalize_wave(int rid, int lane, const int32_t* __restrict__ pitem_ptr,
                               const double* __restrict__ partial, const int32_t* __restrict__ diag_slot,
                               double lambda, double* __restrict__ S, double* __restrict__ g, double* v /* LDS, 33 */)
{
    if (lane < PS_NPOSE_ACC) {
        double s = 0.0;
        for (int it = pitem_ptr[rid]; it < pitem_ptr[rid + 1]; ++it) s += partial[(size_t)it * PS_NPOSE_ACC + lane];
        v[lane] = s;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < 36) {
        const int r = lane / 6, c = lane % 6;
        const int a = r < c ? r : c, b = r < c ? c : r;
        const int idx = a * 6 - (a * (a - 1)) / 2 + (b - a);   // upper-triangle packed index
        double val = v[idx];
        if (r == c) val += lambda * v[27 + r];
        S[(size_t)diag_slot[rid] * 36 + lane] += val;
    }
    if (lane < 6) g[(size_t)rid * 6 + lane] += v[21 + lane];
}

__global__ __launch_bounds__(64) void k_pose_finalize(
    int nr, const int32_t* __restrict__ pitem_ptr, const double* __restrict__ partial,
    const int32_t* __restrict__ diag_slot, double lambda,
    double* __restrict__ S, double* __restrict__ g)
{
    __shared__ double v[PS_NPOSE_ACC];
    pose_finalize_wave(blockIdx.x, threadIdx.x, pitem_ptr, partial, diag_slot, lambda, S, g, v);
}

// ---------------------------------------------------------------------------
// Schur off-diagonal blocks: one wave per reduced-system block
// ---------------------------------------------------------------------------
// XCD-aware work order: workgroup b runs on XCD b % 8 (observed dispatch rule; affects speed
// only), and order[] lists, per XCD, the blocks of a CONTIGUOUS range of block rows.  All blocks
// that share pose ri's Z rows (and, for neighbouring rows, pose rj's) then hit the same 4 MB L2
// instead of being re-fetched by all eight.
//
// Z rows are scattered, so a lane-per-pair gather issues fully divergent 16-byte loads (41 M L1
// accesses at C3 with the 144-byte rows of round 1).  Instead each wave moves the 64 rows of a
// 32-pair chunk straight into LDS with global_load_lds_dwordx4 (no staging registers, no ds_write
// pass).  A stored row is ONE aligned 128-byte line (M | pc | rid | pad, see PS_ZROW): 7 consecutive
// lanes fetch its first 7 x 16 B, 9 rows per instruction, and because the LDS destination of lane l
// is base + 16 l the rows land at a 112-byte stride, which is conflict-free for the ds_read_b128 of
// the compute phase (28 banks per row: 16 consecutive rows start on 16 distinct multiples of 4).  One L2
// request per row instead of two for the 144-byte rows (8.8 M -> 4.6 M line requests per launch at C3;
// the kernel is bound by exactly that request stream).  Two lanes share a pair
// (lane p + 32 h accumulates block rows 3h .. 3h+2: h = 0 from M_a, h = 1 from pc_a^ M_a; the b row is
// expanded to [M_b; pc_b^ M_b] by both), so a lane carries 18 accumulators: 7 KB of LDS and < 128
// VGPRs per wave => 4 waves per SIMD.  Waves never share LDS data: no workgroup barrier.
#ifndef PS_SP_WAVES
#define PS_SP_WAVES 4                         // waves per SIMD the register budget is held to (128 VGPRs; 5 = 96 VGPRs spills: 60 -> 80 us)
#endif
#define PS_SP_PAIRS 32                        // pairs per chunk: rows a_0..a_31, b_0..b_31
#define PS_SP_ROWD 14                         // doubles per row in LDS (7 x 16 B of the 128-byte line)
#define PS_SP_LDS_PER_WAVE (64 * PS_SP_ROWD)  // doubles: 64 rows x 14

// sum over the 32 lanes of each wave half with DPP row operations (fixed order): lane 31 / 63
// end up with the total of lanes 0-31 / 32-63
PS_DEV double half_sum_dpp(double v) {
    v = dpp_shift_add<0x111, 0xf, 0xf>(v);  // row_shr:1
    v = dpp_shift_add<0x112, 0xf, 0xf>(v);  // row_shr:2
    v = dpp_shift_add<0x114, 0xf, 0xe>(v);  // row_shr:4
    v = dpp_shift_add<0x118, 0xf, 0xc>(v);  // row_shr:8
    v = dpp_shift_add<0x142, 0xa, 0xf>(v);  // row_bcast:15 into rows 1 and 3
    return v;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PS_SP_WAVES, 8))) void k_schur_pairs(
    int per_xcd, const PairItem* __restrict__ xitems /* [8][per_xcd], slot < 0: padding */,
    const int2* __restrict__ pairs, const double* __restrict__ Z, double* __restrict__ S,
    double* __restrict__ Spart /* tiled mode: one partial block per task position, else NULL */, int ablate,
    int first /* items [first, last) of every XCD's list: the whole list, or one half of it (PS_SCHUR_SPLIT) */, int last)
{
    __shared__ __attribute__((aligned(16))) double smem[4 * PS_SP_LDS_PER_WAVE];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double* rows = smem + wv * PS_SP_LDS_PER_WAVE;
    const int local = first + (blockIdx.x >> 3) * 4 + wv;
    if (local >= last) return;
    const size_t pos = (size_t)(blockIdx.x & 7) * per_xcd + local;
    const PairItem it = xitems[pos];
    if (it.slot < 0) return;
    const int p = lane & 31, hf = lane >> 5;                    // pair in the chunk, half of the block
    const int slot = lane / 7, piece = lane - 7 * slot;         // fetch role; lane 63: slot 9 (idle)
    const int32_t* flat = reinterpret_cast<const int32_t*>(pairs) + hf;
    double acc[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) acc[k] = 0.0;
    // lane l holds the Z row index of LDS row l of a chunk (a_p for l < 32, b_p above); the index
    // loads run two chunks ahead of the row fetches so that no chunk waits on them
    int mine = (it.start + p < it.end) ? flat[2 * (size_t)(it.start + p)] : -1;
    int mine1 = (it.start + PS_SP_PAIRS + p < it.end) ? flat[2 * (size_t)(it.start + PS_SP_PAIRS + p)] : -1;
    for (int base = it.start; base < it.end; base += PS_SP_PAIRS) {
        const int n = min(PS_SP_PAIRS, it.end - base);
        // ---- cooperative fetch: instruction k brings rows 9k .. 9k+8 into LDS.  All eight index
        // shuffles are issued first (one wait), the next-but-one chunk's indices are requested
        // BEFORE the rows so that the single vmcnt(0) below never waits on a younger load.
        int zrow[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) zrow[k] = __shfl(mine, (9 * k + slot) & 63, 64);
        const int nb = base + 2 * PS_SP_PAIRS;
        const int mine2 = (nb + p < it.end) ? flat[2 * (size_t)(nb + p)] : -1;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int r = 9 * k + slot;
            if (slot < 9 && r < 2 * PS_SP_PAIRS && zrow[k] >= 0 && !(ablate & 2) && !((ablate & 4) && r < PS_SP_PAIRS))   // (4: no a-rows, timing only)
                __builtin_amdgcn_global_load_lds((ps_gptr_t)(Z + PS_ZROW * (size_t)zrow[k] + 2 * piece),
                                                 (ps_lptr_t)(rows + 9 * PS_SP_ROWD * k), 16, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0): rows have landed in LDS
        __builtin_amdgcn_wave_barrier();
        if (p < n && !(ablate & 1)) {
            // this lane's three rows of Z_a: M_a (h = 0) or pc_a^ M_a (h = 1) ...
            double A[9];
            {
                double ma[12];
                const double2* pa = reinterpret_cast<const double2*>(rows + PS_SP_ROWD * p);
#pragma unroll
                for (int k = 0; k < 6; ++k) { const double2 va = pa[k]; ma[2 * k] = va.x; ma[2 * k + 1] = va.y; }
                zrow_cross(ma, ma + 9, A);
#pragma unroll
                for (int k = 0; k < 9; ++k) A[k] = hf ? A[k] : ma[k];
            }
            // ... against both halves of Z_b, one after the other (M_b is dead once pc_b^ M_b is formed: fewer live registers)
            double mb[12];
            const double2* pb = reinterpret_cast<const double2*>(rows + PS_SP_ROWD * (PS_SP_PAIRS + p));
#pragma unroll
            for (int k = 0; k < 6; ++k) { const double2 vb = pb[k]; mb[2 * k] = vb.x; mb[2 * k + 1] = vb.y; }
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b)
                    acc[6 * a + b] += A[3 * a] * mb[3 * b] + A[3 * a + 1] * mb[3 * b + 1] + A[3 * a + 2] * mb[3 * b + 2];
            double Lb[9];
            zrow_cross(mb, mb + 9, Lb);
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b)
                    acc[6 * a + 3 + b] += A[3 * a] * Lb[3 * b] + A[3 * a + 1] * Lb[3 * b + 1] + A[3 * a + 2] * Lb[3 * b + 2];
        }
        __builtin_amdgcn_wave_barrier();                        // LDS reads done before the next fetch lands
        mine = mine1; mine1 = mine2;
    }
    // ---- reduce the 18 accumulators over the 32 lanes of each half (DPP, fixed order); lanes 31
    // and 63 publish the 36 block entries through LDS for the coalesced, mirrored write
    double* sums = rows;
#pragma unroll
    for (int k = 0; k < 18; ++k) {
        const double t = half_sum_dpp(acc[k]);
        if (p == 31) sums[18 * hf + k] = t;                     // entry (3 hf + k / 6, k % 6) = 18 hf + k
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < 36) {
        const int r = lane / 6, c = lane % 6;
        const double mine_v = sums[lane];
        if (Spart) {
            Spart[pos * 36 + lane] = mine_v;
        } else if (it.slot == it.slotT) {                       // duplicate observation: a diagonal block
            S[(size_t)it.slot * 36 + lane] -= mine_v + sums[c * 6 + r];
        } else {                                                // off-diagonal blocks are still zero here
            S[(size_t)it.slot * 36 + lane] = -mine_v;
            S[(size_t)it.slotT * 36 + c * 6 + r] = -mine_v;
        }
    }
}

// tiled mode: sum the (tile, block) partials of every block in tile order and apply them to S and
// to the mirrored block; one wave per block
__global__ __launch_bounds__(256) void k_schur_combine(
    int nblocks, const PairItem* __restrict__ items, const int32_t* __restrict__ tasks,
    const double* __restrict__ Spart, double* __restrict__ S,
    // workgroups beyond the blocks finalize the pose pass (fin_nr > 0; never when a task writes a diagonal block):
    // one launch less on the critical path
    int fin_nr, const int32_t* __restrict__ pitem_ptr, const double* __restrict__ ppartial,
    const int32_t* __restrict__ diag_slot, double lambda, double* __restrict__ g)
{
    __shared__ double fin_v[4][PS_NPOSE_ACC];
    const int nbw = (nblocks + 3) / 4;
    if ((int)blockIdx.x >= nbw) {
        const int rid = (blockIdx.x - nbw) * 4 + (threadIdx.x >> 6);
        if (rid < fin_nr)
            pose_finalize_wave(rid, threadIdx.x & 63, pitem_ptr, ppartial, diag_slot, lambda, S, g, fin_v[threadIdx.x >> 6]);
        return;
    }
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= nblocks || lane >= 36) return;
    const PairItem it = items[b];
    const int r = lane / 6, c = lane % 6;
    double v = 0.0, vt = 0.0;
    for (int k = it.start; k < it.end; ++k) {
        const double* q = Spart + (size_t)tasks[k] * 36;
        v += q[lane];
        vt += q[c * 6 + r];
    }
    if (it.slot == it.slotT) {
        S[(size_t)it.slot * 36 + lane] -= v + vt;
    } else {
        S[(size_t)it.slot * 36 + lane] -= v;
        S[(size_t)it.slotT * 36 + lane] -= vt;
    }
}

// ---------------------------------------------------------------------------
// pose-pose / prior factors.  scratch row per factor: [H11 | H12 | H22 | g1 | g2]  (3 D^2 + 2 D doubles)
// A workgroup takes 64 factors in two phases:
//   1. one THREAD per factor: E = T_2 T_1^-1 T_obs^-1, xi = log(E) (acos, sin, tan, a square root and the inverse
//      left Jacobian: ~800 instructions), r = S xi, the IRLS scales s_k = sqrt(w(r_k)); s, s r and T_2 T_1^-1 go to LDS.
//      (Round 1 ran this part in all 64 lanes of one wave per factor: 64 copies of the same logarithm.)
//   2. one WAVE per factor, lane (r, c): the scaled Jacobians J1 = -diag(s) S Ad(T_2 T_1^-1), J2 = diag(s) S in LDS,
//      then the three D x D products and the two gradient pieces.
// ---------------------------------------------------------------------------
#define PS_FP_FACTORS 64
template <int D>
__global__ __launch_bounds__(256) void k_factor_pass(
    int nf, const int32_t* __restrict__ f_i, const int32_t* __restrict__ f_j,
    const double* __restrict__ f_Tinv, const int32_t* __restrict__ f_grp,
    const FactorGroup* __restrict__ groups, const double* __restrict__ poses,
    double* __restrict__ scratch, double* __restrict__ dbg /* parity tap or nullptr: [r~ | J~1 | J~2] per factor */)
{
    typedef PoseOps<D> G;
    constexpr int DD = D * D, ROW = 3 * DD + 2 * D, FPB = PS_FP_FACTORS;
    __shared__ double sS[FPB][D], sSr[FPB][D], sT21[FPB][G::W];
    __shared__ int32_t sBin[FPB], sGrp[FPB];
    __shared__ double sJ1[4][36], sJ2[4][36];
    const int f0 = blockIdx.x * FPB;
    if (threadIdx.x < FPB && f0 + (int)threadIdx.x < nf) {
        const int fl = threadIdx.x, f = f0 + fl;
        const int i = f_i[f], j = f_j[f], gi = f_grp[f];
        const bool binary = i >= 0;
        const FactorGroup& grp = groups[gi];
        const typename G::T T2 = G::load(poses + G::W * (size_t)j);
        const typename G::T To = G::load(f_Tinv + G::W * (size_t)f);
        typename G::T E, T21 = T2;
        if (binary) {
            const typename G::T T1i = G::inv(G::load(poses + G::W * (size_t)i));
            E = G::mul(T2, G::mul(T1i, To));            // T_2 (T_1^-1 T_obs^-1)
            T21 = G::mul(T2, T1i);
        } else {
            E = G::mul(T2, To);
        }
        double xi[D];
        G::log(E, xi);
#pragma unroll
        for (int k = 0; k < D; ++k) {
            double rk = 0.0;
            bool present = false;               // an all-zero stiffness row is an absent residual row
#pragma unroll                                  // (rotation-only edges, lowering.py): no weight, no 0 * inf
            for (int m = 0; m < D; ++m) { rk += grp.S[k * D + m] * xi[m]; present = present || grp.S[k * D + m] != 0.0; }
            const double sk = present ? sqrt(ps_loss_weight(grp.loss_id, grp.loss_k, rk)) : 0.0;
            sS[fl][k] = sk;
            sSr[fl][k] = sk * rk;
        }
        G::store(sT21[fl], T21);
        sBin[fl] = binary ? 1 : 0;
        sGrp[fl] = gi;
    }
    __syncthreads();
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool act = lane < DD;
    const int r = act ? lane / D : 0, c = act ? lane - (lane / D) * D : 0;
    for (int q = 0; q < FPB / 4; ++q) {
        const int fl = w * (FPB / 4) + q, f = f0 + fl;
        if (f >= nf) break;                                   // (uniform per wave)
        const FactorGroup& grp = groups[sGrp[fl]];
        if (act) {
            // row r of J~ (scaled by s_r): J1 = -S Ad(T_2 T_1^-1), J2 = S
            const double sk = sS[fl][r];
            double j1 = 0.0;
            if (sBin[fl]) {
#pragma unroll
                for (int m = 0; m < D; ++m) j1 -= grp.S[r * D + m] * G::adj_mem(sT21[fl], m, c);     // (c differs per lane: from LDS)
            }
            sJ1[w][lane] = sk * j1;
            sJ2[w][lane] = sk * grp.S[lane];
        }
        __builtin_amdgcn_wave_barrier();
        if (dbg != nullptr && act) {                          // (uniform: production launches pass nullptr)
            double* o = dbg + (size_t)f * (D + 2 * DD);
            if (lane < D) o[lane] = sSr[fl][lane];
            o[D + lane] = sJ1[w][lane];
            o[D + DD + lane] = sJ2[w][lane];
        }
        if (act) {
            double h11 = 0.0, h12 = 0.0, h22 = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                h11 += sJ1[w][k * D + r] * sJ1[w][k * D + c];
                h12 += sJ1[w][k * D + r] * sJ2[w][k * D + c];
                h22 += sJ2[w][k * D + r] * sJ2[w][k * D + c];
            }
            double* out = scratch + (size_t)f * ROW;
            out[lane] = h11; out[DD + lane] = h12; out[2 * DD + lane] = h22;
            if (c == 0) {
                double g1 = 0.0, g2 = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) { g1 -= sJ1[w][k * D + r] * sSr[fl][k]; g2 -= sJ2[w][k * D + r] * sSr[fl][k]; }
                out[3 * DD + r] = g1; out[3 * DD + D + r] = g2;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// gather factor blocks into S (one thread per entry of every touched block) and g
template <int D>
__global__ __launch_bounds__(256) void k_factor_assemble(
    int nslots, const int32_t* __restrict__ eslots, const int32_t* __restrict__ eptr,
    const int2* __restrict__ eitems /* (scratch offset, transpose) */,
    const int32_t* __restrict__ slot_is_diag,
    int ng, const int32_t* __restrict__ gptr, const int32_t* __restrict__ gitems,
    const double* __restrict__ scratch, double lambda, double* __restrict__ S, double* __restrict__ g)
{
    constexpr int DD = D * D;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nslots * DD) {
        const int si = t / DD, e = t % DD, r = e / D, c = e % D;
        double s = 0.0;
        for (int k = eptr[si]; k < eptr[si + 1]; ++k) {
            const int2 itx = eitems[k];
            s += scratch[(size_t)itx.x + (itx.y ? c * D + r : e)];
        }
        const int slot = eslots[si];
        if (r == c && slot_is_diag[si]) s *= (1.0 + lambda);
        S[(size_t)slot * DD + e] += s;
    }
    const int u = t - nslots * DD;
    if (u >= 0 && u < ng * D) {
        const int rid = u / D, r = u % D;
        double s = 0.0;
        for (int k = gptr[rid]; k < gptr[rid + 1]; ++k) s += scratch[(size_t)gitems[k] + r];
        g[(size_t)rid * D + r] += s;
    }
}
