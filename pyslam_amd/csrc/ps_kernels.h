// ps_kernels.h -- device kernels of the Gauss-Newton / LM iteration (gfx950, fp64).
//
// Pipeline per iteration:
//   k_landmark_pass      16 lanes / landmark  : H_ll, b_l -> chol -> c_l, Z_i = W_i C^-T (LDS-transposed stores)
//   k_pose_pass          WG / pose chunk      : J_p^T J_p - Z Z^T, -J_p^T r - Z c, Z recomputed in registers
//   k_pose_finalize      WG / pose            : chunk partials -> diagonal S block, g
//   k_schur_pairs        wave / (tile, block) : S_ij = -sum Z_i Z_j^T, rows straight into LDS (global_load_lds)
//   k_schur_combine      wave / block         : tiled mode: partials summed in tile order
//   k_factor_pass        wave / pose edge     : pose-pose & prior blocks via LDS-staged 6x6 Jacobians
//   k_factor_assemble    thread / S entry     : gather edge blocks into S, g
//   reduced solve        k_block_jacobi_factor, k_scale_blocks, k_coarse_* (two-level setup),
//                        k_cg_fused_lds / k_cg_fused (one launch per CG iteration), k_coarse_recover;
//                        <= 90 unknowns: k_bsr_to_dense + k_coarse_chol + k_direct_apply;
//                        classic k_pcg_* (two launches per iteration) kept as an independent variant
//   k_backsub            16 lanes / landmark  : dx_l = C^-T (c_l - sum Z_i^T dx_p) (+ fused update of points, poses)
//   k_cost_*, k_reduce3  robust cost, final reductions, results published to pinned host memory
//   k_motion_only_iteration  WG / pose        : problems without landmarks / factors: the whole iteration
// Every reduction has a fixed order: results are bitwise reproducible run to run.
#pragma once
#include "ps_math.h"

struct __attribute__((aligned(32))) LObs {   // one reprojection observation, 32 B
    double u, v, d;
    int32_t pose_grp;      // pose index (low 24 bits) | group (high 8 bits)
    int32_t point;
};
struct PItem { int32_t rid, start, end, pad; };
struct PairItem { int32_t slot, slotT, start, end; };
struct FactorGroup { double S[36]; int32_t loss_id; int32_t pad; double loss_k; };

#define PS_POSE_OF(o) ((o).pose_grp & 0xFFFFFF)
#define PS_GRP_OF(o) (((uint32_t)(o).pose_grp) >> 24)

// status words
enum { ST_LM_FAIL = 0, ST_DIAG_FAIL = 1, ST_PCG_DONE = 2, ST_PCG_ITERS = 3, ST_NWORDS = 8 };
// scalar slots
enum { SC_COST = 0, SC_DXP2 = 1, SC_LINCOST = 2, SC_RR0 = 3, SC_RRFINAL = 4, SC_THRESH = 5, SC_DXL2 = 6, SC_NWORDS = 8 };

// Wave-wide sum with DPP row operations instead of ds_bpermute shuffles (each __shfl_xor of a
// double is two LDS-crossbar permutes, ~100+ cycles of latency; a DPP add is a plain VALU op).
// Classic GCN/CDNA reduction: row_shr 1,2,4(masked),8(masked) -> 16-lane row totals in the
// row's last lane, row_bcast15 / row_bcast31 carry them across rows; lane 63 holds the total,
// which is broadcast with readlane.  Fixed association order => deterministic.
template <int CTRL, int ROW_MASK, int BANK_MASK>
PS_DEV double dpp_shift_add(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, ROW_MASK, BANK_MASK, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, ROW_MASK, BANK_MASK, false);
    const unsigned long long m = ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
    return v + __builtin_bit_cast(double, m);      // lanes masked off by ROW/BANK_MASK add +0.0
}

PS_DEV double wave_sum(double v) {          // every lane gets the total
    v = dpp_shift_add<0x111, 0xf, 0xf>(v);  // row_shr:1
    v = dpp_shift_add<0x112, 0xf, 0xf>(v);  // row_shr:2
    v = dpp_shift_add<0x114, 0xf, 0xe>(v);  // row_shr:4, banks 1-3
    v = dpp_shift_add<0x118, 0xf, 0xc>(v);  // row_shr:8, banks 2-3
    v = dpp_shift_add<0x142, 0xa, 0xf>(v);  // row_bcast:15 into rows 1 and 3
    v = dpp_shift_add<0x143, 0xc, 0xf>(v);  // row_bcast:31 into rows 2 and 3
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)b, 63);
    const unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(b >> 32), 63);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// deterministic block-wide sum (blockDim.x multiple of 64, <= 1024); result valid in every thread
PS_DEV double block_sum(double v, double* lds /* >= 16 doubles */) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds[w] = v;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < nw; ++i) t += lds[i];
    return t;
}

// two block-wide sums sharing one butterfly and one barrier pair (the shuffles of a and b
// interleave, so the pair costs about one reduction's latency)
PS_DEV void block_sum2(double& a, double& b, double* lds /* >= 32 doubles */) {
    a = wave_sum(a); b = wave_sum(b);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { lds[w] = a; lds[16 + w] = b; }
    __syncthreads();
    double sa = 0.0, sb = 0.0;
    for (int i = 0; i < nw; ++i) { sa += lds[i]; sb += lds[16 + i]; }
    a = sa; b = sb;
}

// ---------------------------------------------------------------------------
// landmark pass: 16 lanes cooperate on one landmark (4 landmarks per wave), one observation
// per lane: loads of the 32-byte records and stores of the 144-byte Z rows are contiguous
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

PS_DEV void lm_emit_z(const ReprojEval& ev, double M00, double M10, double M11, double M20, double M21,
                      double M22, double* __restrict__ z) {
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        const double w0 = ev.Jp[a] * ev.Jl[0] + ev.Jp[6 + a] * ev.Jl[3] + ev.Jp[12 + a] * ev.Jl[6];
        const double w1 = ev.Jp[a] * ev.Jl[1] + ev.Jp[6 + a] * ev.Jl[4] + ev.Jp[12 + a] * ev.Jl[7];
        const double w2 = ev.Jp[a] * ev.Jl[2] + ev.Jp[6 + a] * ev.Jl[5] + ev.Jp[12 + a] * ev.Jl[8];
        z[3 * a] = w0 * M00;
        z[3 * a + 1] = w0 * M10 + w1 * M11;
        z[3 * a + 2] = w0 * M20 + w1 * M21 + w2 * M22;
    }
}

__global__ __launch_bounds__(256) void k_landmark_pass(
    int nv, const int32_t* __restrict__ lm_ptr, const int32_t* __restrict__ lm_point,
    const LObs* __restrict__ lobs, const double* __restrict__ poses,
    const double* __restrict__ points, const int32_t* __restrict__ pose_rid,
    const ObsGroup* __restrict__ groups, double lambda,
    double* __restrict__ Z, double* __restrict__ Cinv, double* __restrict__ cvec,
    int32_t* __restrict__ status, int ablate)
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
    for (int i = b + sub; i < e; i += PS_LM_GROUP) {
        const LObs o = lobs[i];
        const int pose = PS_POSE_OF(o);
        const Se3 T = se3_load(poses + 12 * pose);
        variable_pose = pose_rid[pose] >= 0;
        reproj_eval<true, true>(T, pw, &o.u, groups[PS_GRP_OF(o)], ev);
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
    // H_ll = C C^T
    const double l00 = sqrt(H00);
    const double l10 = H10 / l00, l20 = H20 / l00;
    const double d1 = H11 - l10 * l10;
    const double l11 = sqrt(d1);
    const double l21 = (H21 - l20 * l10) / l11;
    const double d2 = H22 - l20 * l20 - l21 * l21;
    const double l22 = sqrt(d2);
    // M = C^-1 (lower)
    const double M00 = 1.0 / l00, M11 = 1.0 / l11, M22 = 1.0 / l22;
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
    // ---- Z rows.  Common case (every landmark of the wave has <= 16 observations): the wave's rows are
    // one contiguous range of Z, so they are transposed through LDS and stored as whole 16-byte pieces by
    // consecutive lanes (1 KB per store instruction) instead of 18 stride-144 8-byte stores per lane.
    __shared__ __attribute__((aligned(16))) double zst[4][64 * 18];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (__ballot(single || !live) == ~0ull && !(ablate & 2)) {
        const int row0 = __shfl(b, 0, 64);                       // dead groups carry b = e = 0
        const int eend = max(max(__shfl(e, 0, 64), __shfl(e, 16, 64)), max(__shfl(e, 32, 64), __shfl(e, 48, 64)));
        const int nrows = eend - row0;
        if (have) {
            double z[18];
            if (variable_pose) lm_emit_z(ev, M00, M10, M11, M20, M21, M22, z);
            else {
#pragma unroll
                for (int k = 0; k < 18; ++k) z[k] = 0.0;         // rows of constant poses are never read
            }
            double2* dst = reinterpret_cast<double2*>(&zst[wv][18 * (b + sub - row0)]);
#pragma unroll
            for (int k = 0; k < 9; ++k) dst[k] = make_double2(z[2 * k], z[2 * k + 1]);
        }
        __builtin_amdgcn_wave_barrier();
        if (!(ablate & 1)) {
            const double2* src = reinterpret_cast<const double2*>(zst[wv]);
            double2* out = reinterpret_cast<double2*>(Z + 18 * (size_t)row0);
            for (int k = lane; k < nrows * 9; k += 64) out[k] = src[k];
        }
        return;
    }
    if (!live) return;
    if (single) {
        if (have && variable_pose && !(ablate & 1)) lm_emit_z(ev, M00, M10, M11, M20, M21, M22, Z + 18 * (size_t)(b + sub));
        return;
    }
    for (int i = b + sub; i < e; i += PS_LM_GROUP) {
        const LObs o = lobs[i];
        const int pose = PS_POSE_OF(o);
        if (pose_rid[pose] < 0) continue;
        const Se3 T = se3_load(poses + 12 * pose);
        reproj_eval<true, true>(T, pw, &o.u, groups[PS_GRP_OF(o)], ev);
        lm_emit_z(ev, M00, M10, M11, M20, M21, M22, Z + 18 * (size_t)i);
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
// table) -- lm_emit_z on the same inputs, so the values are those the landmark pass stored.
#define PS_NPOSE_ACC 33
typedef const __attribute__((address_space(1))) void* ps_gptr_t;
typedef __attribute__((address_space(3))) void* ps_lptr_t;

__global__ __launch_bounds__(256) void k_pose_pass(
    const PItem* __restrict__ items,
    const LObs* __restrict__ pobs /* observation records in pose order, landmark slot + 1 in the pose bits */,
    const double* __restrict__ poses, const double* __restrict__ points,
    const ObsGroup* __restrict__ groups, const double* __restrict__ Cinv,
    const double* __restrict__ cvec, double* __restrict__ partial)
{
    __shared__ double red[4][PS_NPOSE_ACC];
    const PItem it = items[blockIdx.x];
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
        reproj_eval<true, true>(T, pw, &o.u, groups[PS_GRP_OF(o)], ev);
        int n = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = a; b < 6; ++b)
                acc[n++] += ev.Jp[a] * ev.Jp[b] + ev.Jp[6 + a] * ev.Jp[6 + b] + ev.Jp[12 + a] * ev.Jp[12 + b];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            acc[21 + a] -= ev.Jp[a] * ev.r[0] + ev.Jp[6 + a] * ev.r[1] + ev.Jp[12 + a] * ev.r[2];
            acc[27 + a] += ev.Jp[a] * ev.Jp[a] + ev.Jp[6 + a] * ev.Jp[6 + a] + ev.Jp[12 + a] * ev.Jp[12 + a];
        }
        if (v >= 0) {
            double z[18];
            lm_emit_z(ev, m[0], m[1], m[2], m[3], m[4], m[5], z);
            n = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = a; b < 6; ++b)
                    acc[n++] -= z[3 * a] * z[3 * b] + z[3 * a + 1] * z[3 * b + 1] + z[3 * a + 2] * z[3 * b + 2];
#pragma unroll
            for (int a = 0; a < 6; ++a)
                acc[21 + a] -= z[3 * a] * c0 + z[3 * a + 1] * c1 + z[3 * a + 2] * c2;
        }
    }
#pragma unroll
    for (int k = 0; k < PS_NPOSE_ACC; ++k) {
        const double s = wave_sum(acc[k]);
        if (lane == 0) red[w][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < PS_NPOSE_ACC)
        partial[(size_t)blockIdx.x * PS_NPOSE_ACC + threadIdx.x] =
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
// Z rows are 144 B and scattered, so a lane-per-pair gather issues 18 fully divergent 16-byte
// loads per pair (41 M L1 accesses at C3).  Instead each wave moves the 64 rows of a 32-pair
// chunk straight into LDS with global_load_lds_dwordx4 (no staging registers, no ds_write pass):
// 9 consecutive lanes fetch the 9 x 16 B of ONE row, 7 rows per instruction, and because the
// LDS destination of lane l is base + 16 l the rows land at their natural 144-byte stride, which
// is conflict-free for the ds_read_b128 of the compute phase.  Two lanes share a pair (lane
// p + 32 h accumulates block rows 3h .. 3h+2), so a lane carries 18 accumulators instead of 36:
// ~9 KB of LDS and < 128 VGPRs per wave => 4 waves per SIMD, twice the loads in flight of the
// register-staged 64-pair version.  Waves never share LDS data: no workgroup barrier.
#define PS_SP_PAIRS 32                        // pairs per chunk: rows a_0..a_31, b_0..b_31
#define PS_SP_LDS_PER_WAVE 1152               // doubles: 64 rows x 18

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

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_schur_pairs(
    int per_xcd, const PairItem* __restrict__ xitems /* [8][per_xcd], slot < 0: padding */,
    const int2* __restrict__ pairs, const double* __restrict__ Z, double* __restrict__ S,
    double* __restrict__ Spart /* tiled mode: one partial block per task position, else NULL */, int ablate)
{
    __shared__ __attribute__((aligned(16))) double smem[4 * PS_SP_LDS_PER_WAVE];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double* rows = smem + wv * PS_SP_LDS_PER_WAVE;
    const int local = (blockIdx.x >> 3) * 4 + wv;
    if (local >= per_xcd) return;
    const size_t pos = (size_t)(blockIdx.x & 7) * per_xcd + local;
    const PairItem it = xitems[pos];
    if (it.slot < 0) return;
    const int p = lane & 31, hf = lane >> 5;                    // pair in the chunk, half of the block
    const int slot = lane / 9, piece = lane - 9 * slot;         // fetch role; lane 63: slot 7 (idle)
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
        // ---- cooperative fetch: instruction k brings rows 7k .. 7k+6 into LDS.  All ten index
        // shuffles are issued first (one wait), the next-but-one chunk's indices are requested
        // BEFORE the rows so that the single vmcnt(0) below never waits on a younger load.
        int zrow[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) zrow[k] = __shfl(mine, (7 * k + slot) & 63, 64);
        const int nb = base + 2 * PS_SP_PAIRS;
        const int mine2 = (nb + p < it.end) ? flat[2 * (size_t)(nb + p)] : -1;
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const int r = 7 * k + slot;
            if (slot < 7 && r < 2 * PS_SP_PAIRS && zrow[k] >= 0 && !(ablate & 2))
                __builtin_amdgcn_global_load_lds((ps_gptr_t)(Z + 18 * (size_t)zrow[k] + 2 * piece),
                                                 (ps_lptr_t)(rows + 126 * k), 16, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0): rows have landed in LDS
        __builtin_amdgcn_wave_barrier();
        if (p < n && !(ablate & 1)) {
            double za[9];
            const double* pa = rows + 18 * p + 9 * hf;
#pragma unroll
            for (int k = 0; k < 9; ++k) za[k] = pa[k];
#pragma unroll
            for (int bh = 0; bh < 2; ++bh) {
                double zb[10];
                // columns 3bh .. 3bh+2 need b-row entries 9bh .. 9bh+8; read 16-byte aligned
                const double2* pb = reinterpret_cast<const double2*>(rows + 18 * (PS_SP_PAIRS + p) + 8 * bh);
#pragma unroll
                for (int k = 0; k < 5; ++k) { const double2 v = pb[k]; zb[2 * k] = v.x; zb[2 * k + 1] = v.y; }
                const double* q = zb + bh;                      // q[0..8] = entries 9bh .. 9bh+8
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < 3; ++b)
                        acc[6 * a + 3 * bh + b] += za[3 * a] * q[3 * b] + za[3 * a + 1] * q[3 * b + 1] + za[3 * a + 2] * q[3 * b + 2];
            }
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
// pose-pose / prior factors: one wave per factor, Jacobians staged in LDS.
// scratch row per factor: [H11 | H12 | H22 | g1 | g2]  (3 D^2 + 2 D doubles)
// ---------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void k_factor_pass(
    int nf, const int32_t* __restrict__ f_i, const int32_t* __restrict__ f_j,
    const double* __restrict__ f_Tinv, const int32_t* __restrict__ f_grp,
    const FactorGroup* __restrict__ groups, const double* __restrict__ poses,
    double* __restrict__ scratch)
{
    typedef PoseOps<D> G;
    constexpr int DD = D * D, ROW = 3 * DD + 2 * D;
    __shared__ double sJ1[4][36], sJ2[4][36], sr[4][6];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int f = blockIdx.x * 4 + w;
    const bool live = f < nf;
    const bool act = live && lane < DD;
    const int r = lane / D, c = lane % D;
    bool binary = false;
    if (live) {
        const int i = f_i[f], j = f_j[f];
        binary = i >= 0;
        const FactorGroup& grp = groups[f_grp[f]];
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
        double xi[D], s[D];
        G::log(E, xi);
#pragma unroll
        for (int k = 0; k < D; ++k) {
            double rk = 0.0;
            bool present = false;               // an all-zero stiffness row is an absent residual row
#pragma unroll                                  // (rotation-only edges, lowering.py): no weight, no 0 * inf
            for (int m = 0; m < D; ++m) { rk += grp.S[k * D + m] * xi[m]; present = present || grp.S[k * D + m] != 0.0; }
            s[k] = present ? sqrt(ps_loss_weight(grp.loss_id, grp.loss_k, rk)) : 0.0;
            if (lane == 0) sr[w][k] = s[k] * rk;
        }
        if (act) {
            // row r of J~ (scaled by s_r): J1 = -S Ad(T_2 T_1^-1), J2 = S
            double sk = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) if (k == r) sk = s[k];
            double j1 = 0.0;
            if (binary) {
#pragma unroll
                for (int m = 0; m < D; ++m) j1 -= grp.S[r * D + m] * G::adj(T21, m, c);
            }
            sJ1[w][lane] = sk * j1;
            sJ2[w][lane] = sk * grp.S[lane];
        }
    }
    __syncthreads();
    if (!act) return;
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
        for (int k = 0; k < D; ++k) { g1 -= sJ1[w][k * D + r] * sr[w][k]; g2 -= sJ2[w][k * D + r] * sr[w][k]; }
        out[3 * DD + r] = g1; out[3 * DD + D + r] = g2;
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

// ---------------------------------------------------------------------------
// block-Jacobi PCG on the reduced system (BSR, D x D blocks, both triangles)
// ---------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void k_block_jacobi(
    int nr, const int32_t* __restrict__ diag_slot, const double* __restrict__ S,
    double* __restrict__ Minv, int32_t* __restrict__ status)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nr) return;
    double A[D][D], L[D][D], Li[D][D];
    const double* s = S + (size_t)diag_slot[i] * D * D;
#pragma unroll
    for (int r = 0; r < D; ++r)
#pragma unroll
        for (int c = 0; c < D; ++c) { A[r][c] = s[r * D + c]; L[r][c] = 0.0; Li[r][c] = 0.0; }
    bool ok = true;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        double d = A[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
        ok = ok && (d > 0.0);
        const double l = sqrt(d);
        L[j][j] = l;
#pragma unroll
        for (int i2 = j + 1; i2 < D; ++i2) {
            double v = A[i2][j];
#pragma unroll
            for (int k = 0; k < j; ++k) v -= L[i2][k] * L[j][k];
            L[i2][j] = v / l;
        }
    }
    // Li = L^-1 (lower), column by column
#pragma unroll
    for (int c = 0; c < D; ++c) {
        Li[c][c] = 1.0 / L[c][c];
#pragma unroll
        for (int r = c + 1; r < D; ++r) {
            double v = 0.0;
#pragma unroll
            for (int k = c; k < r; ++k) v -= L[r][k] * Li[k][c];
            Li[r][c] = v / L[r][r];
        }
    }
    if (!ok) atomicAdd(&status[ST_DIAG_FAIL], 1);
    double* m = Minv + (size_t)i * D * D;     // A^-1 = Li^T Li
#pragma unroll
    for (int r = 0; r < D; ++r)
#pragma unroll
        for (int c = 0; c < D; ++c) {
            double v = 0.0;
#pragma unroll
            for (int k = (r > c ? r : c); k < D; ++k) v += Li[k][r] * Li[k][c];
            m[r * D + c] = v;
        }
}

// vector-update kernels: each wave owns 64/D whole block rows (a block row never
// straddles two waves, so its D lanes read r[] before any of them overwrites it)
#define PS_PCG_BRW(D) (64 / (D))
#define PS_PCG_BR(D) (4 * PS_PCG_BRW(D))

// x = 0, r = g, z = M^-1 r, partial r.z and r.r
template <int D>
__global__ __launch_bounds__(256) void k_pcg_init(
    int nr, const double* __restrict__ g, const double* __restrict__ Minv,
    double* __restrict__ x, double* __restrict__ r, double* __restrict__ z,
    double* __restrict__ rz_part, double* __restrict__ rr_part, int32_t* __restrict__ status)
{
    __shared__ double lds[16];
    const int t = threadIdx.x;
    const int lane_ = t & 63;
    const int brow = blockIdx.x * PS_PCG_BR(D) + (t >> 6) * PS_PCG_BRW(D) + lane_ / D, rr_ = lane_ % D;
    double prz = 0.0, prr = 0.0;
    if (lane_ < PS_PCG_BRW(D) * D && brow < nr) {
        double rn[D];
#pragma unroll
        for (int c = 0; c < D; ++c) rn[c] = g[(size_t)brow * D + c];
        double zi = 0.0, ri = 0.0;
#pragma unroll
        for (int c = 0; c < D; ++c) {
            zi += Minv[(size_t)brow * D * D + rr_ * D + c] * rn[c];
            if (c == rr_) ri = rn[c];
        }
        const size_t i = (size_t)brow * D + rr_;
        x[i] = 0.0; r[i] = ri; z[i] = zi;
        prz = zi * ri; prr = ri * ri;
    }
    const double a = block_sum(prz, lds);
    const double b = block_sum(prr, lds);
    if (t == 0) { rz_part[blockIdx.x] = a; rr_part[blockIdx.x] = b; }
    if (blockIdx.x == 0 && t == 0) { status[ST_PCG_DONE] = 0; status[ST_PCG_ITERS] = 0; }
}

// A: (beta from the partials) p = z + beta p_old on the fly; q = S p; partial p.q
template <int D>
__global__ __launch_bounds__(256) void k_pcg_spmv(
    int nr, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col_idx,
    const double* __restrict__ S, const double* __restrict__ z,
    const double* __restrict__ p_old, double* __restrict__ p_new, double* __restrict__ q,
    const double* __restrict__ rz_part, const double* __restrict__ rr_part, int npartB,
    double* __restrict__ pq_part, double* __restrict__ hist, int k, double tol2,
    int32_t* __restrict__ status, double* __restrict__ scalars)
{
    __shared__ double lds[4][8];
    constexpr int DD = D * D;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // Every load below is independent: issue them all before the first branch so the
    // kernel pays ONE memory latency here instead of a chain (these kernels are a few us
    // long and latency-, not bandwidth-bound).
    const int done = status[ST_PCG_DONE];
    const int row = blockIdx.x;
    const int rbeg = row_ptr[row], rend = row_ptr[row + 1];
    const double rz_prev = hist[k > 0 ? k - 1 : 0];
    const double thresh_in = scalars[SC_THRESH];
    double rz = 0.0, rr = 0.0;
    for (int i = lane; i < npartB; i += 64) { rz += rz_part[i]; rr += rr_part[i]; }
    const int kk = lane >> 3, r = lane & 7;
    const int b0 = rbeg + w * 8 + kk;
    int cj = 0;
    if (b0 < rend) cj = col_idx[b0];
    if (done) return;
    rz = wave_sum(rz); rr = wave_sum(rr);
    // convergence in the PRECONDITIONED norm r^T M^-1 r: invariant to the block scaling of the
    // system (a 1e12 prior next to unit-weight loop closures), unlike ||r||_2 / ||g||_2
    (void)rr;
    const double thresh = (k == 0) ? tol2 * rz : thresh_in;
    const bool first_wave = (blockIdx.x == 0 && threadIdx.x == 0);
    if (!(rz > thresh)) {                      // converged (also catches rz == 0 and NaN)
        if (first_wave) { status[ST_PCG_DONE] = 1; scalars[SC_RRFINAL] = rz; if (k == 0) scalars[SC_RR0] = rz; }
        return;
    }
    const double beta = (k == 0) ? 0.0 : rz / rz_prev;
    if (first_wave) {
        hist[k] = rz; status[ST_PCG_ITERS] = k + 1; scalars[SC_RRFINAL] = rz;
        if (k == 0) { scalars[SC_THRESH] = thresh; scalars[SC_RR0] = rz; }
    }
    // one workgroup per block row; a wave pass covers 8 blocks x D rows (lane = 8*blk + row),
    // so the row's blocks are fetched with 32-way memory parallelism instead of one at a time
    double acc = 0.0;
    if (r < D) {
        for (int b = b0; b < rend; b += 32) {
            const size_t j = (size_t)(b == b0 ? cj : col_idx[b]) * D;
            const double* sb = S + (size_t)b * DD + r * D;
#pragma unroll
            for (int c = 0; c < D; ++c) acc += sb[c] * (z[j + c] + beta * p_old[j + c]);
        }
    }
    acc += __shfl_xor(acc, 8, 64);
    acc += __shfl_xor(acc, 16, 64);
    acc += __shfl_xor(acc, 32, 64);
    if (lane < 8) lds[w][lane] = acc;
    __syncthreads();
    if (w == 0) {
        double pq = 0.0;
        if (lane < D) {
            const double qr = ((lds[0][lane] + lds[1][lane]) + lds[2][lane]) + lds[3][lane];
            const size_t i = (size_t)row * D + lane;
            const double pn = z[i] + beta * p_old[i];
            p_new[i] = pn; q[i] = qr;
            pq = pn * qr;
        }
        pq = wave_sum(pq);
        if (lane == 0) pq_part[row] = pq;
    }
}

// B: alpha = rz / p.q ; x += alpha p ; r -= alpha q ; z = M^-1 r ; partial r.z, r.r
template <int D>
__global__ __launch_bounds__(256) void k_pcg_update(
    int nr, const double* __restrict__ Minv, const double* __restrict__ p,
    const double* __restrict__ q, double* __restrict__ x, double* __restrict__ r,
    double* __restrict__ z, const double* __restrict__ pq_part, int npartA,
    const double* __restrict__ hist, int k, double* __restrict__ rz_part,
    double* __restrict__ rr_part, const int32_t* __restrict__ status)
{
    __shared__ double lds[16];
    const int t = threadIdx.x;
    // all loads first (independent of alpha), then the reduction that yields alpha
    const int done = status[ST_PCG_DONE];
    const double rzk = hist[k];
    double pq = 0.0;
    for (int i = t; i < npartA; i += 256) pq += pq_part[i];
    const int lane_ = t & 63;
    const int brow = blockIdx.x * PS_PCG_BR(D) + (t >> 6) * PS_PCG_BRW(D) + lane_ / D, rr_ = lane_ % D;
    const bool act = lane_ < PS_PCG_BRW(D) * D && brow < nr;
    double rv[D], qv[D], mv[D], pi = 0.0, xi_ = 0.0;
#pragma unroll
    for (int c = 0; c < D; ++c) { rv[c] = 0.0; qv[c] = 0.0; mv[c] = 0.0; }
    if (act) {
#pragma unroll
        for (int c = 0; c < D; ++c) {
            rv[c] = r[(size_t)brow * D + c];
            qv[c] = q[(size_t)brow * D + c];
            mv[c] = Minv[(size_t)brow * D * D + rr_ * D + c];
        }
        pi = p[(size_t)brow * D + rr_];
        xi_ = x[(size_t)brow * D + rr_];
    }
    if (done) return;
    pq = block_sum(pq, lds);
    const double alpha = rzk / pq;
    double prz = 0.0, prr = 0.0;
    if (act) {
        double zi = 0.0, ri = 0.0;
#pragma unroll
        for (int c = 0; c < D; ++c) {
            const double rn = rv[c] - alpha * qv[c];
            zi += mv[c] * rn;
            if (c == rr_) ri = rn;
        }
        const size_t i = (size_t)brow * D + rr_;
        x[i] = xi_ + alpha * pi;
        r[i] = ri; z[i] = zi;       // same-wave lanes have already loaded r[] (see PS_PCG_BRW)
        prz = zi * ri; prr = ri * ri;
    }
    const double a = block_sum(prz, lds);
    const double b = block_sum(prr, lds);
    if (t == 0) { rz_part[blockIdx.x] = a; rr_part[blockIdx.x] = b; }
}

// ---------------------------------------------------------------------------
// Fused CG: ONE launch per iteration (Chronopoulos-Gear single-reduction form)
// on the explicitly block-Jacobi-scaled system  S^ = L^-1 S L^-T,  g^ = L^-1 g,
// x = L^-T x^   (M = L L^T = diag blocks of S).  Per iteration k:
//   gamma_k = r.r, delta_k = w.r (reduced from the previous launch's partials)
//   beta = gamma_k/gamma_{k-1},  alpha = gamma_k / (delta_k - beta gamma_k / alpha_{k-1})
//   s = w + beta s ; p = r + beta p ; x += alpha p ; r -= alpha s ; w = S^ r
// Every workgroup recomputes r_new at the columns it needs from (r, w, s) of the
// previous launch, so the only global dependency is the launch boundary itself.
// The k = -1 launch (alpha = beta = 0) initialises w = S^ g^ and the first partials.
// ---------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void k_block_jacobi_factor(
    int nr, const int32_t* __restrict__ diag_slot, const double* __restrict__ S,
    double* __restrict__ Linv, int32_t* __restrict__ status,
    // fused CG only (g != NULL): also the start vectors of the scaled system, r = Linv g, w = s = p = x = 0
    const double* __restrict__ g, double* __restrict__ r0, double* __restrict__ w0, double* __restrict__ s0,
    double* __restrict__ p0, double* __restrict__ x0,
    // two-level CG (Bmat != NULL): the coarse basis block of this pose, B_i = L_i^T Ad(T_i) (basis 1: a
    // coarse unknown is a BODY-frame twist eta, the fine correction is x_i = Ad(T_i) eta, x^_i = L_i^T x_i)
    // or the identity (basis 0: hats directly in the scaled coordinates), and bg_i = B_i^T r_i
    const double* __restrict__ poses, const int32_t* __restrict__ pose_of_rid, int basis,
    double* __restrict__ Bmat, double* __restrict__ bg)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (g && i == 0) { status[ST_PCG_DONE] = 0; status[ST_PCG_ITERS] = 0; }
    if (i >= nr) return;
    double A[D][D], L[D][D], Li[D][D];
    const double* s = S + (size_t)diag_slot[i] * D * D;
#pragma unroll
    for (int r = 0; r < D; ++r)
#pragma unroll
        for (int c = 0; c < D; ++c) { A[r][c] = s[r * D + c]; L[r][c] = 0.0; Li[r][c] = 0.0; }
    bool ok = true;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        double d = A[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
        ok = ok && (d > 0.0);
        const double l = sqrt(d);
        L[j][j] = l;
#pragma unroll
        for (int i2 = j + 1; i2 < D; ++i2) {
            double v = A[i2][j];
#pragma unroll
            for (int k = 0; k < j; ++k) v -= L[i2][k] * L[j][k];
            L[i2][j] = v / l;
        }
    }
#pragma unroll
    for (int c = 0; c < D; ++c) {
        Li[c][c] = 1.0 / L[c][c];
#pragma unroll
        for (int r = c + 1; r < D; ++r) {
            double v = 0.0;
#pragma unroll
            for (int k = c; k < r; ++k) v -= L[r][k] * Li[k][c];
            Li[r][c] = v / L[r][r];
        }
    }
    if (!ok) atomicAdd(&status[ST_DIAG_FAIL], 1);
    double* m = Linv + (size_t)i * D * D;
#pragma unroll
    for (int r = 0; r < D; ++r)
#pragma unroll
        for (int c = 0; c < D; ++c) m[r * D + c] = Li[r][c];
    if (g) {
#pragma unroll
        for (int r = 0; r < D; ++r) {
            double v = 0.0;
#pragma unroll
            for (int c = 0; c < D; ++c) v += Li[r][c] * g[(size_t)i * D + c];
            const size_t o = (size_t)i * D + r;
            r0[o] = v; w0[o] = 0.0; s0[o] = 0.0; p0[o] = 0.0; x0[o] = 0.0;
        }
    }
    if (Bmat) {
        typedef PoseOps<D> G;
        double B[D][D];
        if (basis == 1) {
            const typename G::T T = G::load(poses + G::W * (size_t)pose_of_rid[i]);
#pragma unroll
            for (int a = 0; a < D; ++a)
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    double v = 0.0;
#pragma unroll
                    for (int m2 = a; m2 < D; ++m2) v += L[m2][a] * G::adj(T, m2, c);     // (L^T Ad)[a][c]
                    B[a][c] = v;
                }
        } else {
#pragma unroll
            for (int a = 0; a < D; ++a)
#pragma unroll
                for (int c = 0; c < D; ++c) B[a][c] = (a == c) ? 1.0 : 0.0;
        }
        double* bm = Bmat + (size_t)i * D * D;
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
            for (int c = 0; c < D; ++c) bm[a * D + c] = B[a][c];
        if (g) {
            double rr[D];
#pragma unroll
            for (int r = 0; r < D; ++r) {
                double v = 0.0;
#pragma unroll
                for (int c = 0; c < D; ++c) v += Li[r][c] * g[(size_t)i * D + c];
                rr[r] = v;
            }
#pragma unroll
            for (int c = 0; c < D; ++c) {
                double v = 0.0;
#pragma unroll
                for (int a = 0; a < D; ++a) v += B[a][c] * rr[a];
                bg[(size_t)i * D + c] = v;
            }
        }
    }
}

// Sout[out_slot[b]] = Linv_i S_ij Linv_j^T  (one 64-thread workgroup per block; S itself is kept)
template <int D>
__global__ __launch_bounds__(64) void k_scale_blocks(
    int nr, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col_idx,
    const int32_t* __restrict__ brow_of, const double* __restrict__ Linv, const double* __restrict__ S,
    const int32_t* __restrict__ out_slot, double* __restrict__ Sout,
    const double* __restrict__ Bmat /* two-level CG: also SB[b] = S^_b B_j, the input of the coarse row sums */,
    double* __restrict__ SB)
{
    constexpr int DD = D * D;
    __shared__ double sS[36], sT[36], sLi[36], sLj[36], sB[36];
    const int b = blockIdx.x, t = threadIdx.x;
    const int i = brow_of[b], j = col_idx[b];
    if (t < DD) {
        sS[t] = S[(size_t)b * DD + t];
        sLi[t] = Linv[(size_t)i * DD + t];
        sLj[t] = Linv[(size_t)j * DD + t];
        if (Bmat) sB[t] = Bmat[(size_t)j * DD + t];
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
        Sout[(size_t)out_slot[b] * DD + t] = v;
        sS[t] = v;
    }
    if (!Bmat) return;
    __syncthreads();
    if (t < DD) {
        double v = 0.0;
#pragma unroll
        for (int a = 0; a < D; ++a) v += sS[r * D + a] * sB[a * D + c];
        SB[(size_t)out_slot[b] * DD + t] = v;
    }
}

// r0 = g^ = Linv g ; w = s = p = x^ = 0
template <int D>
__global__ __launch_bounds__(256) void k_cg_prepare(
    int nr, const double* __restrict__ g, const double* __restrict__ Linv,
    double* __restrict__ r, double* __restrict__ w, double* __restrict__ s,
    double* __restrict__ p, double* __restrict__ x, int32_t* __restrict__ status,
    const double* __restrict__ Bmat, double* __restrict__ bg /* two-level: bg_i = B_i^T r_i */)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) { status[ST_PCG_DONE] = 0; status[ST_PCG_ITERS] = 0; }
    if (t >= nr * D) return;
    const int i = t / D, rr_ = t % D;
    double v = 0.0;
#pragma unroll
    for (int c = 0; c < D; ++c) v += Linv[(size_t)i * D * D + rr_ * D + c] * g[(size_t)i * D + c];
    r[t] = v; w[t] = 0.0; s[t] = 0.0; p[t] = 0.0; x[t] = 0.0;
    if (Bmat) {                                          // column rr_ of B_i against the whole r_i (recomputed: D^2 flops)
        double acc = 0.0;
#pragma unroll
        for (int a = 0; a < D; ++a) {
            double ra = 0.0;
#pragma unroll
            for (int c = 0; c < D; ++c) ra += Linv[(size_t)i * D * D + a * D + c] * g[(size_t)i * D + c];
            acc += Bmat[(size_t)i * D * D + a * D + rr_] * ra;
        }
        bg[t] = acc;
    }
}

PS_DEV double cg_rnew(double r, double w, double s, double alpha, double beta) {
    return r - alpha * (w + beta * s);
}

// gamma / delta totals for large systems: with thousands of rows every workgroup re-reducing all
// per-row partials would cost O(rows^2) traffic, so one extra single-workgroup launch per iteration
// reduces them once (fixed order) and k_cg_fused reads two scalars (pre_reduced = 1).
__global__ __launch_bounds__(1024) void k_cg_reduce(int nr, const double* __restrict__ gd, double* __restrict__ tot,
                                                     const int32_t* __restrict__ status)
{
    __shared__ double lds[32];
    if (status[ST_PCG_DONE]) return;
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nr; i += 1024) { a += gd[i]; b += gd[nr + i]; }
    block_sum2(a, b, lds);
    if (threadIdx.x == 0) { tot[0] = a; tot[1] = b; }
}

// Split mode: workgroup 0 reduces the fine rows' gamma / delta partials; workgroup 1 + q owns coarse
// block row q of the augmented system [[S^, K], [K^T, I]]:
//   w_new_c[q] = sum_i U[q][i] + r_new_c[q],  then the same vector recurrences as a fine row.
template <int D>
__global__ __launch_bounds__(1024) void k_cg_reduce_split(
    int nr, int ncb, const double* __restrict__ gd, double* __restrict__ tot,
    const double* __restrict__ U, const double* __restrict__ ab,
    const double* __restrict__ r_old, const double* __restrict__ w_old, const double* __restrict__ s_old,
    double* __restrict__ r_new, double* __restrict__ w_new, double* __restrict__ s_new,
    double* __restrict__ p, double* __restrict__ x, double* __restrict__ cgd_out /* [2 ncb] */,
    const int32_t* __restrict__ status,
    const double* __restrict__ Mc /* lagged coarse factor: the coarse-coarse block M (nc x nc), NULL = identity */)
{
    __shared__ double lds[32];
    __shared__ double wpart[16][8];
    __shared__ double srn[400], mrow[8];
    const int t = threadIdx.x;
    // independent loads first (this kernel is latency-bound: ~26-64 workgroups on 256 CUs)
    const int done = status[ST_PCG_DONE];
    if (blockIdx.x == 0) {
        double a = 0.0, b = 0.0;
#pragma unroll 4
        for (int i = t; i < nr; i += 1024) { a += gd[i]; b += gd[nr + i]; }
        if (done) return;
        block_sum2(a, b, lds);
        if (t == 0) { tot[0] = a; tot[1] = b; }
        return;
    }
    const int q = blockIdx.x - 1;
    const double alpha = ab[0], beta = ab[1];
    // sum over i of U[q][i][0..D): flat index e = i*D + c, thread t takes e = t, t + 1024*? ... keep c fixed
    // per thread: with 1024 = 6*170 + 4 not a multiple of D, use the row mapping: rows t, t+1024, ...
    double acc[D];
#pragma unroll
    for (int c = 0; c < D; ++c) acc[c] = 0.0;
    const double* u = U + (size_t)q * nr * D;
#pragma unroll 2
    for (int i = t; i < nr; i += 1024)
#pragma unroll
        for (int c = 0; c < D; ++c) acc[c] += u[(size_t)i * D + c];
    double ri = 0.0, wi = 0.0, si = 0.0, pi = 0.0, xi_ = 0.0;
    if (t < D) {
        const size_t i = (size_t)(nr + q) * D + t;
        ri = r_old[i]; wi = w_old[i]; si = s_old[i]; pi = p[i]; xi_ = x[i];
    }
    const int nc = ncb * D;
    double rn_k = 0.0;                                     // r_new of coarse entry t (for the M row products)
    if (Mc && t < nc) {
        const size_t i = (size_t)nr * D + t;
        rn_k = cg_rnew(r_old[i], w_old[i], s_old[i], alpha, beta);
    }
    if (done) return;
    const int wv = t >> 6, lane = t & 63;
#pragma unroll
    for (int c = 0; c < D; ++c) {
        const double v = wave_sum(acc[c]);
        if (lane == 0) wpart[wv][c] = v;
    }
    if (Mc && t < nc) srn[t] = rn_k;
    __syncthreads();
    if (Mc && wv < D) {                                    // wave c: row q*D + c of M times r_new (coarse part)
        const double* mr = Mc + (size_t)(q * D + wv) * nc;
        double v = 0.0;
        for (int k = lane; k < nc; k += 64) v += mr[k] * srn[k];
        v = wave_sum(v);
        if (lane == 0) mrow[wv] = v;
    }
    if (Mc) __syncthreads();
    double gp = 0.0, dp = 0.0;
    if (t < D) {
        double ws = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) ws += wpart[k][t];
        const size_t i = (size_t)(nr + q) * D + t;
        const double rn = cg_rnew(ri, wi, si, alpha, beta);
        const double wn = ws + (Mc ? mrow[t] : rn);        // coarse-coarse block: M (lagged factor) or the identity
        const double sn = wi + beta * si;
        const double pn = ri + beta * pi;
        s_new[i] = sn; p[i] = pn; x[i] = xi_ + alpha * pn; r_new[i] = rn; w_new[i] = wn;
        gp = rn * rn; dp = wn * rn;
    }
    if (t < 64) {
        gp = wave_sum(gp); dp = wave_sum(dp);
        if (t == 0) { cgd_out[q] = gp; cgd_out[ncb + q] = dp; }
    }
}

template <int D, int NW /* waves per workgroup: 8 for long rows, 1 for short (pose-graph) rows */>
__global__ __launch_bounds__(64 * NW) void k_cg_fused(
    int nr, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col_idx,
    const double* __restrict__ S,
    const double* __restrict__ r_old, const double* __restrict__ w_old, const double* __restrict__ s_old,
    double* __restrict__ r_new, double* __restrict__ w_new, double* __restrict__ s_new,
    double* __restrict__ p, double* __restrict__ x,
    const double* __restrict__ gd_in /* [2*nr] gamma | delta partials */, double* __restrict__ gd_out,
    double* __restrict__ hist /* [0,cap): gamma, [cap,2cap): alpha */, int cap, int k, double tol2,
    int32_t* __restrict__ status, double* __restrict__ scalars,
    int nfine, int wf, int wc /* two-class ELL: fine rows wf blocks wide, the rest wc; wf = 0 => CSR */,
    int ablate /* timing experiments only: 1 skips the SpMV, 2 skips the partial-sum reduction */,
    const double* __restrict__ gd_tot /* non-null: totals already reduced by k_cg_reduce */,
    // split mode (large systems): the matrix holds fine rows only; this kernel also emits
    // U[q][i] = K_iq^T r_new_i, and k_cg_reduce_split owns the ncb coarse rows
    int ncb_split, const int32_t* __restrict__ fine_nnz, const double* __restrict__ cgd_in /* [2 ncb] */,
    double* __restrict__ U, double* __restrict__ ab /* alpha, beta of this launch */)
{
    __shared__ double lds[32];
    __shared__ double part[NW][8];
    constexpr int DD = D * D;
    const int t = threadIdx.x, w = t >> 6, lane = t & 63;
    const int row = blockIdx.x;
    // ---- every independent load first (one memory latency, not a chain)
    const int done = status[ST_PCG_DONE];
    // padded (ELL) rows start at an address computed from the row index, so the column
    // indices load in the same memory round trip as everything else (CSR needs row_ptr first)
    int rbeg, rend;
    if (wf > 0) {
        rbeg = row < nfine ? row * wf : nfine * wf + (row - nfine) * wc;
        rend = rbeg + (row < nfine ? wf : wc);
    } else {
        rbeg = row_ptr[row]; rend = row_ptr[row + 1];
    }
    const double g_prev = hist[k > 0 ? k - 1 : 0];
    const double a_prev = hist[cap + (k > 0 ? k - 1 : 0)];
    const double thresh_in = scalars[SC_THRESH];
    double gs = 0.0, ds = 0.0;
    if (k >= 0 && ablate != 2) {
        if (gd_tot) {
            gs = gd_tot[0]; ds = gd_tot[1];
            if (ncb_split)                       // + the coarse rows' shares (fixed order, every lane the same)
                for (int q = 0; q < ncb_split; ++q) { gs += cgd_in[q]; ds += cgd_in[ncb_split + q]; }
        }
        else for (int i = t; i < nr; i += 64 * NW) { gs += gd_in[i]; ds += gd_in[nr + i]; }
    }
    const int fnz = ncb_split ? fine_nnz[row] : 0;
    const int kk = lane >> 3, r = lane & 7;
    const int b0 = rbeg + w * 8 + kk;
    constexpr int STRIDE = 8 * NW;
    // rows of the dense border K^T (row >= nfine in the ELL layout) have columns 0,1,2,...,nr+ncb-1:
    // their column index is arithmetic, so their vector loads do not wait for a col_idx load.
    const bool dense_row = wf > 0 && row >= nfine;
    int cj0 = 0, cj1 = 0;                        // column blocks of this lane's first two passes
    if (!dense_row) {
        if (b0 < rend) cj0 = col_idx[b0];
        if (b0 + STRIDE < rend) cj1 = col_idx[b0 + STRIDE];
    }
    double ri = 0.0, wi = 0.0, si = 0.0, pi = 0.0, xi_ = 0.0;
    if (t < D) {
        const size_t i = (size_t)row * D + t;
        ri = r_old[i]; wi = w_old[i]; si = s_old[i]; pi = p[i]; xi_ = x[i];
    }
    if (done) return;
    double alpha = 0.0, beta = 0.0;
    if (k >= 0 && ablate == 2) { alpha = 1e-3; beta = 0.5; }
    if (k >= 0 && ablate != 2) {
        if (!gd_tot) block_sum2(gs, ds, lds);
        const double gamma = gs, delta = ds;
        const double thresh = (k == 0) ? tol2 * gamma : thresh_in;
        const bool first = (blockIdx.x == 0 && t == 0);
        if (!(gamma > thresh)) {                     // converged (or gamma == 0 / NaN)
            if (first) { status[ST_PCG_DONE] = 1; scalars[SC_RRFINAL] = gamma; if (k == 0) scalars[SC_RR0] = gamma; }
            return;
        }
        beta = (k == 0) ? 0.0 : gamma / g_prev;
        const double denom = (k == 0) ? delta : delta - beta * gamma / a_prev;
        alpha = gamma / denom;
        if (!(denom > 0.0)) {                        // breakdown: stop, the host reports it
            if (first) { status[ST_PCG_DONE] = 2; scalars[SC_RRFINAL] = gamma; }
            return;
        }
        if (first) {
            hist[k] = gamma; hist[cap + k] = alpha; status[ST_PCG_ITERS] = k + 1; scalars[SC_RRFINAL] = gamma;
            if (k == 0) { scalars[SC_THRESH] = thresh; scalars[SC_RR0] = gamma; }
        }
    }
    if (ncb_split && blockIdx.x == 0 && t == 0) { ab[0] = alpha; ab[1] = beta; }
    // ---- w_new(row) = S^(row,:) r_new, with r_new recomputed per column block
    double acc = 0.0;
    if (r < D && ablate != 1) {
#pragma unroll 2
        for (int b = b0; b < rend; b += STRIDE) {
            int jc;
            if (dense_row) jc = b - rbeg;                  // K^T over the fine columns, then the coarse-coarse row
            else jc = (b == b0) ? cj0 : ((b == b0 + STRIDE) ? cj1 : col_idx[b]);
            const size_t j = (size_t)jc * D;
            const double* sb = S + (size_t)b * DD + r * D;
#pragma unroll
            for (int c = 0; c < D; ++c)
                acc += sb[c] * cg_rnew(r_old[j + c], w_old[j + c], s_old[j + c], alpha, beta);
        }
    }
    if (ncb_split) {
        // transposed border for the coarse rows: U[q][row] = K_iq^T r_new_i.  Lane (group, r) holds
        // row r of K_iq; the 8-lane group is summed with DPP row shifts (total lands in lane r == 7).
        double rown = 0.0;
        if (r < D) {
            const size_t i = (size_t)row * D + r;
            rown = cg_rnew(r_old[i], w_old[i], s_old[i], alpha, beta);
        }
        for (int q0 = 0; q0 < ncb_split; q0 += STRIDE) {
            const int q = q0 + w * 8 + kk;
            double tq[D];
#pragma unroll
            for (int c = 0; c < D; ++c) tq[c] = 0.0;
            if (q < ncb_split && r < D) {
                const double* sb = S + (size_t)(rbeg + fnz + q) * DD + r * D;
#pragma unroll
                for (int c = 0; c < D; ++c) tq[c] = sb[c] * rown;
            }
#pragma unroll
            for (int c = 0; c < D; ++c) {
                tq[c] = dpp_shift_add<0x111, 0xf, 0xf>(tq[c]);
                tq[c] = dpp_shift_add<0x112, 0xf, 0xf>(tq[c]);
                tq[c] = dpp_shift_add<0x114, 0xf, 0xf>(tq[c]);
            }
            if (q < ncb_split && r == 7) {
                double* u = U + ((size_t)q * nr + row) * D;
#pragma unroll
                for (int c = 0; c < D; ++c) u[c] = tq[c];
            }
        }
    }
    acc += __shfl_xor(acc, 8, 64);
    acc += __shfl_xor(acc, 16, 64);
    acc += __shfl_xor(acc, 32, 64);
    if (lane < 8) part[w][lane] = acc;
    __syncthreads();
    if (w == 0) {
        double gp = 0.0, dp = 0.0;
        if (lane < D) {
            double wn = 0.0;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) wn += part[ww][lane];
            const size_t i = (size_t)row * D + lane;
            const double sn = wi + beta * si;
            const double pn = ri + beta * pi;
            const double rn = cg_rnew(ri, wi, si, alpha, beta);
            s_new[i] = sn; p[i] = pn; x[i] = xi_ + alpha * pn; r_new[i] = rn; w_new[i] = wn;
            gp = rn * rn; dp = wn * rn;
        }
        gp = wave_sum(gp); dp = wave_sum(dp);
        if (lane == 0) { gd_out[row] = gp; gd_out[nr + row] = dp; }
    }
}

// Small systems (the whole CG vector fits in LDS: rows * D <= PS_CGV_MAX): ONE global round trip per
// iteration.  k_cg_fused's SpMV needs r_new of the neighbouring block rows, i.e. r/w/s gathered
// through the column indices -- a second, dependent round trip (~1.5 us of a ~6.5 us launch at C3).
// Here every workgroup instead loads the WHOLE r, w, s vectors (coalesced, addresses known at
// launch: ~30 KB at C3, L2-resident) together with its matrix blocks and the column indices, forms
// r_new for every row in LDS once alpha / beta are known, and the SpMV gathers from LDS.
#define PS_CGV_MAX 4096
template <int D, int NW>
__global__ __launch_bounds__(64 * NW) void k_cg_fused_lds(
    int nr, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col_idx,
    const double* __restrict__ S,
    const double* __restrict__ r_old, const double* __restrict__ w_old, const double* __restrict__ s_old,
    double* __restrict__ r_new, double* __restrict__ w_new, double* __restrict__ s_new,
    double* __restrict__ p, double* __restrict__ x,
    const double* __restrict__ gd_in, double* __restrict__ gd_out,
    double* __restrict__ hist, int cap, int k, double tol2,
    int32_t* __restrict__ status, double* __restrict__ scalars,
    int nfine, int wf, int wc)
{
    __shared__ double lds[32];
    __shared__ double part[NW][8];
    __shared__ double rn[PS_CGV_MAX];
    constexpr int DD = D * D, NT = 64 * NW, NV = (PS_CGV_MAX + NT - 1) / NT, NPRE = 4;
    const int t = threadIdx.x, w = t >> 6, lane = t & 63;
    const int row = blockIdx.x, nvec = nr * D;
    // ---- every load of the launch is issued here: one memory latency
    const int done = status[ST_PCG_DONE];
    int rbeg, rend;
    if (wf > 0) {
        rbeg = row < nfine ? row * wf : nfine * wf + (row - nfine) * wc;
        rend = rbeg + (row < nfine ? wf : wc);
    } else {
        rbeg = row_ptr[row]; rend = row_ptr[row + 1];
    }
    const double g_prev = hist[k > 0 ? k - 1 : 0];
    const double a_prev = hist[cap + (k > 0 ? k - 1 : 0)];
    const double thresh_in = scalars[SC_THRESH];
    double gs = 0.0, ds = 0.0;
    if (k >= 0) for (int i = t; i < nr; i += NT) { gs += gd_in[i]; ds += gd_in[nr + i]; }
    double vr[NV], vw[NV], vs[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        const int i = t + q * NT;
        vr[q] = vw[q] = vs[q] = 0.0;
        if (i < nvec) { vr[q] = r_old[i]; vw[q] = w_old[i]; vs[q] = s_old[i]; }
    }
    const int kk = lane >> 3, r = lane & 7;
    const int b0 = rbeg + w * 8 + kk;
    constexpr int STRIDE = 8 * NW;
    const bool dense_row = wf > 0 && row >= nfine;
    int cj[NPRE];
    double sv[NPRE][D];
#pragma unroll
    for (int q = 0; q < NPRE; ++q) {
        const int b = b0 + q * STRIDE;
        cj[q] = 0;
#pragma unroll
        for (int c = 0; c < D; ++c) sv[q][c] = 0.0;
        if (b < rend && r < D) {
            cj[q] = dense_row ? b - rbeg : col_idx[b];
            const double* sb = S + (size_t)b * DD + r * D;
#pragma unroll
            for (int c = 0; c < D; ++c) sv[q][c] = sb[c];
        }
    }
    double ri = 0.0, wi = 0.0, si = 0.0, pi = 0.0, xi_ = 0.0;
    if (t < D) {
        const size_t i = (size_t)row * D + t;
        ri = r_old[i]; wi = w_old[i]; si = s_old[i]; pi = p[i]; xi_ = x[i];
    }
    if (done) return;
    double alpha = 0.0, beta = 0.0;
    if (k >= 0) {
        block_sum2(gs, ds, lds);
        const double gamma = gs, delta = ds;
        const double thresh = (k == 0) ? tol2 * gamma : thresh_in;
        const bool first = (blockIdx.x == 0 && t == 0);
        if (!(gamma > thresh)) {                     // converged (or gamma == 0 / NaN)
            if (first) { status[ST_PCG_DONE] = 1; scalars[SC_RRFINAL] = gamma; if (k == 0) scalars[SC_RR0] = gamma; }
            return;
        }
        beta = (k == 0) ? 0.0 : gamma / g_prev;
        const double denom = (k == 0) ? delta : delta - beta * gamma / a_prev;
        alpha = gamma / denom;
        if (!(denom > 0.0)) {                        // breakdown: stop, the host reports it
            if (first) { status[ST_PCG_DONE] = 2; scalars[SC_RRFINAL] = gamma; }
            return;
        }
        if (first) {
            hist[k] = gamma; hist[cap + k] = alpha; status[ST_PCG_ITERS] = k + 1; scalars[SC_RRFINAL] = gamma;
            if (k == 0) { scalars[SC_THRESH] = thresh; scalars[SC_RR0] = gamma; }
        }
    }
    // ---- r_new of every row into LDS
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        const int i = t + q * NT;
        if (i < nvec) rn[i] = cg_rnew(vr[q], vw[q], vs[q], alpha, beta);
    }
    __syncthreads();
    // ---- w_new(row) = S^(row,:) r_new
    double acc = 0.0;
    if (r < D) {
#pragma unroll
        for (int q = 0; q < NPRE; ++q) {
            if (b0 + q * STRIDE < rend) {
                const double* v = rn + cj[q] * D;
#pragma unroll
                for (int c = 0; c < D; ++c) acc += sv[q][c] * v[c];
            }
        }
        for (int b = b0 + NPRE * STRIDE; b < rend; b += STRIDE) {
            const int jc = dense_row ? b - rbeg : col_idx[b];
            const double* sb = S + (size_t)b * DD + r * D;
            const double* v = rn + jc * D;
#pragma unroll
            for (int c = 0; c < D; ++c) acc += sb[c] * v[c];
        }
    }
    acc += __shfl_xor(acc, 8, 64);
    acc += __shfl_xor(acc, 16, 64);
    acc += __shfl_xor(acc, 32, 64);
    if (lane < 8) part[w][lane] = acc;
    __syncthreads();
    if (w == 0) {
        double gp = 0.0, dp = 0.0;
        if (lane < D) {
            double wn = 0.0;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) wn += part[ww][lane];
            const size_t i = (size_t)row * D + lane;
            const double sn = wi + beta * si;
            const double pn = ri + beta * pi;
            const double rnv = rn[i];
            s_new[i] = sn; p[i] = pn; x[i] = xi_ + alpha * pn; r_new[i] = rnv; w_new[i] = wn;
            gp = rnv * rnv; dp = wn * rnv;
        }
        gp = wave_sum(gp); dp = wave_sum(dp);
        if (lane == 0) { gd_out[row] = gp; gd_out[nr + row] = dp; }
    }
}

// parameter snapshot / restore: both tables in ONE launch (two hipMemcpyAsync are two blit launches, ~5 us each)
__global__ __launch_bounds__(256) void k_copy2(size_t n1, const double* __restrict__ a_src, double* __restrict__ a_dst,
                                               size_t n2, const double* __restrict__ b_src, double* __restrict__ b_dst)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n1 + n2; i += stride) {
        if (i < n1) a_dst[i] = a_src[i]; else b_dst[i - n1] = b_src[i - n1];
    }
}

// ---------------------------------------------------------------------------
// Direct solve of SMALL reduced systems (nr * D <= 90 unknowns: the reference's own examples, sliding
// windows, motion-only problems): BSR -> dense, the LDS-resident blocked Cholesky + inverse of the
// coarse level (k_coarse_chol), x = L^-T (L^-1 g).  Three launches instead of a CG's 10-40.
// ---------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void k_bsr_to_dense(
    int nr, int nnzb, const int32_t* __restrict__ brow_of, const int32_t* __restrict__ col_idx,
    const double* __restrict__ S, double* __restrict__ A)
{
    constexpr int DD = D * D;
    const int n = nr * D;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n * n; t += gridDim.x * blockDim.x) A[t] = 0.0;
    // (single workgroup launch: the zero fill above is complete for this workgroup after the barrier)
    __syncthreads();
    for (int t = threadIdx.x; t < nnzb * DD; t += blockDim.x) {
        const int b = t / DD, e = t % DD;
        A[(size_t)(brow_of[b] * D + e / D) * n + col_idx[b] * D + e % D] = S[t];
    }
}

template <int D>
__global__ __launch_bounds__(256) void k_direct_apply(
    int n, const double* __restrict__ Li, const double* __restrict__ LiT, const double* __restrict__ g,
    double* __restrict__ x, int32_t* __restrict__ status, double* __restrict__ scalars)
{
    __shared__ double sg[96], sy[96];
    const int t = threadIdx.x;
    if (t < n) sg[t] = g[t];
    __syncthreads();
    if (t < n) {                                         // y = L^-1 g   (row t of Li, k <= t)
        double v = 0.0;
        for (int k = 0; k <= t; ++k) v += LiT[(size_t)k * n + t] * sg[k];
        sy[t] = v;
    }
    __syncthreads();
    if (t < n) {                                         // x = L^-T y   (column t of Li, k >= t)
        double v = 0.0;
        for (int k = t; k < n; ++k) v += Li[(size_t)k * n + t] * sy[k];
        x[t] = v;
    }
    if (t == 0) {
        status[ST_PCG_DONE] = 1; status[ST_PCG_ITERS] = 0;
        scalars[SC_RR0] = 1.0; scalars[SC_RRFINAL] = 0.0;
    }
}

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
#define PS_XCG_ROWS 4                         // rows (waves) per workgroup of the SpMV
#define PS_XCG_DROWS 256                      // rows per workgroup of the prolongation

PS_DEV double xcg_total(const double* __restrict__ part, int n, double* lds) {
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) v += part[i];
    return block_sum(v, lds);
}

template <int D>
__global__ __launch_bounds__(64 * PS_XCG_ROWS) void k_xcg_spmv(
    int nr, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col_idx, int wf,
    const double* __restrict__ S, const double* __restrict__ z, const double* __restrict__ p_old,
    double* __restrict__ p_new, double* __restrict__ q, const double* __restrict__ rz_part, int n_rz,
    double* __restrict__ pq_part, double* __restrict__ xstate, int k, double tol2,
    double* __restrict__ hist, int32_t* __restrict__ status, double* __restrict__ scalars)
{
    __shared__ double lds[16];
    __shared__ double wpq[PS_XCG_ROWS];
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
        if (first) { status[ST_PCG_DONE] = 1; scalars[SC_RRFINAL] = rz; if (k == 0) scalars[SC_RR0] = rz; }
        return;
    }
    const double beta = (k == 0) ? 0.0 : rz / rz_prev;
    if (first) {
        xstate[4 + (k & 1)] = rz; hist[k] = rz; status[ST_PCG_ITERS] = k + 1; scalars[SC_RRFINAL] = rz;
        if (k == 0) { xstate[1] = thresh; xstate[2] = rz; scalars[SC_RR0] = rz; }
    }
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * PS_XCG_ROWS + w;
    double pq = 0.0;
    if (row < nr) {
        const int rbeg = wf > 0 ? row * wf : row_ptr[row];
        const int rend = wf > 0 ? rbeg + wf : row_ptr[row + 1];
        const int kk = lane >> 3, r = lane & 7;
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
    }
    if (lane == 0) wpq[w] = pq;
    __syncthreads();
    if (threadIdx.x == 0) {
        double v = 0.0;
#pragma unroll
        for (int ww = 0; ww < PS_XCG_ROWS; ++ww) v += wpq[ww];
        pq_part[blockIdx.x] = v;
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
__global__ __launch_bounds__(256) void k_xcg_ainv(int nc, const double* __restrict__ Lci, float* __restrict__ Ainv)
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
                if (ti != tj || i >= j) { Ainv[(size_t)i * nc + j] = v; Ainv[(size_t)j * nc + i] = v; }
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
    __shared__ double sDi[64 * 36];     // inverse of every diagonal block of L
    const int t = threadIdx.x, nt = blockDim.x;
    for (int k = t; k < nc * nc; k += nt) { sL[k] = A[k]; sX[k] = 0.0; }
    for (int J = 0; J < ncb; ++J) {
        __syncthreads();
        if (t == 0) {                   // D x D Cholesky of the diagonal block + its inverse
            double L[D][D], Mi[D][D];
            bool ok = true;
#pragma unroll
            for (int a = 0; a < D; ++a)
#pragma unroll
                for (int b2 = 0; b2 < D; ++b2) { L[a][b2] = 0.0; Mi[a][b2] = 0.0; }
            double il[D];                       // 1 / L[j][j]: one division per pivot, the rest are multiplies
#pragma unroll
            for (int j = 0; j < D; ++j) {
                double d = sL[(J * D + j) * nc + J * D + j];
#pragma unroll
                for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
                ok = ok && (d > 0.0);
                const double l = sqrt(d);
                L[j][j] = l;
                il[j] = 1.0 / l;
#pragma unroll
                for (int i = j + 1; i < D; ++i) {
                    double v = sL[(J * D + i) * nc + J * D + j];
#pragma unroll
                    for (int k = 0; k < j; ++k) v -= L[i][k] * L[j][k];
                    L[i][j] = v * il[j];
                }
            }
#pragma unroll
            for (int c = 0; c < D; ++c) {
                Mi[c][c] = il[c];
#pragma unroll
                for (int r = c + 1; r < D; ++r) {
                    double v = 0.0;
#pragma unroll
                    for (int k = c; k < r; ++k) v -= L[r][k] * Mi[k][c];
                    Mi[r][c] = v * il[r];
                }
            }
            if (!ok) atomicAdd(&status[ST_DIAG_FAIL], 1);
#pragma unroll
            for (int a = 0; a < D; ++a)
#pragma unroll
                for (int b2 = 0; b2 < D; ++b2) {
                    sL[(J * D + a) * nc + J * D + b2] = L[a][b2];
                    sDi[J * DD + a * D + b2] = Mi[a][b2];
                }
        }
        __syncthreads();
        // panel: L_IJ = A_IJ L_JJ^-T   (entry (a,b) = sum_{k<=b} A_IJ[a][k] Mi[b][k])
        const int m = ncb - J - 1;
        double pv[4];                         // <= 15*36 (D=6) or 31*9 (D=3) entries over 256 threads
        int np = 0;
        for (int idx = t; idx < m * DD; idx += nt, ++np) {
            const int I = J + 1 + idx / DD, e = idx % DD, a = e / D, b2 = e % D;
            double v = 0.0;
            for (int k = 0; k <= b2; ++k) v += sL[(I * D + a) * nc + J * D + k] * sDi[J * DD + b2 * D + k];
            pv[np] = v;
        }
        __syncthreads();
        np = 0;
        for (int idx = t; idx < m * DD; idx += nt, ++np) {
            const int I = J + 1 + idx / DD, e = idx % DD;
            sL[(I * D + e / D) * nc + J * D + e % D] = pv[np];
        }
        __syncthreads();
        // trailing update A_IK -= L_IJ L_KJ^T for J < K <= I
        for (int idx = t; idx < m * m * DD; idx += nt) {
            const int blk = idx / DD, e = idx % DD, a = e / D, b2 = e % D;
            const int I = J + 1 + blk / m, K = J + 1 + blk % m;
            if (K > I) continue;
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) v += sL[(I * D + a) * nc + J * D + k] * sL[(K * D + b2) * nc + J * D + k];
            sL[(I * D + a) * nc + K * D + b2] -= v;
        }
    }
    // X = L^-1 by block rows: X_RC = Mi_R (delta_RC I - sum_{K=C}^{R-1} L_RK X_KC), all C <= R in parallel
    for (int R = 0; R < ncb; ++R) {
        __syncthreads();
        double tv[4];                         // <= 16*36 entries over 256 threads
        int np = 0;
        for (int idx = t; idx < (R + 1) * DD; idx += nt, ++np) {
            const int C = idx / DD, e = idx % DD, a = e / D, b2 = e % D;
            double v = (C == R && a == b2) ? 1.0 : 0.0;
            for (int K = C; K < R; ++K)
#pragma unroll
                for (int k = 0; k < D; ++k) v -= sL[(R * D + a) * nc + K * D + k] * sX[(K * D + k) * nc + C * D + b2];
            tv[np] = v;
        }
        __syncthreads();
        np = 0;
        for (int idx = t; idx < (R + 1) * DD; idx += nt, ++np) {          // stage T in the X_RC slots
            const int C = idx / DD, e = idx % DD;
            sX[(R * D + e / D) * nc + C * D + e % D] = tv[np];
        }
        __syncthreads();
        np = 0;
        for (int idx = t; idx < (R + 1) * DD; idx += nt, ++np) {
            const int C = idx / DD, e = idx % DD, a = e / D, b2 = e % D;
            double v = 0.0;
            for (int k = 0; k <= a; ++k) v += sDi[R * DD + a * D + k] * sX[(R * D + k) * nc + C * D + b2];
            tv[np] = v;
        }
        __syncthreads();
        np = 0;
        for (int idx = t; idx < (R + 1) * DD; idx += nt, ++np) {
            const int C = idx / DD, e = idx % DD;
            sX[(R * D + e / D) * nc + C * D + e % D] = tv[np];
        }
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
// Motion-only problems (no variable landmark, no pose factor: the reduced system is block diagonal --
// reference pipelines/sparse.py:153-161, SURVEY config C5): ONE launch per Gauss-Newton iteration.
// One workgroup per pose: residuals + Jacobians + IRLS of its observations, 33 sums, 6 x 6 Cholesky
// solve, retraction, post-step cost; the last workgroup to arrive sums the per-pose {cost, |dx|^2}
// in pose order and publishes status + scalars to pinned host memory (sequence word).
// ---------------------------------------------------------------------------
#define PS_MO_THREADS 512
__global__ __launch_bounds__(PS_MO_THREADS) void k_motion_only_iteration(
    int nr, const PItem* __restrict__ items, const int32_t* __restrict__ pitem_ptr,
    const LObs* __restrict__ pobs, const double* __restrict__ points, const ObsGroup* __restrict__ groups,
    double* __restrict__ poses, double lambda, int linesearch,
    double* __restrict__ xout /* nr x 6 */, double* __restrict__ partials /* nr x 2: cost, |dx|^2 */,
    int32_t* __restrict__ status, double* __restrict__ scalars, int32_t* __restrict__ arrivals,
    int32_t* __restrict__ hst, double* __restrict__ hsc, long long* __restrict__ hseq, long long seq)
{
    constexpr int NWV = PS_MO_THREADS / 64;
    __shared__ double red[NWV][PS_NPOSE_ACC + 1];
    __shared__ double tot[PS_NPOSE_ACC + 1];
    __shared__ double sT[12];
    __shared__ int s_last;
    const int rid = blockIdx.x, t = threadIdx.x, w = t >> 6, lane = t & 63;
    const int ib = pitem_ptr[rid], ie = pitem_ptr[rid + 1];
    const int start = ib < ie ? items[ib].start : 0, end = ib < ie ? items[ie - 1].end : 0;
    const int pose = ib < ie ? items[ib].pad : 0;
    Se3 T = se3_load(poses + 12 * (size_t)pose);
    double acc[PS_NPOSE_ACC + 1];
#pragma unroll
    for (int k = 0; k <= PS_NPOSE_ACC; ++k) acc[k] = 0.0;
    for (int i = start + t; i < end; i += PS_MO_THREADS) {
        const LObs o = pobs[i];
        const double pw[3] = {points[3 * (size_t)o.point], points[3 * (size_t)o.point + 1], points[3 * (size_t)o.point + 2]};
        ReprojEval ev;
        reproj_eval<true, false>(T, pw, &o.u, groups[PS_GRP_OF(o)], ev);
        int n = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = a; b < 6; ++b)
                acc[n++] += ev.Jp[a] * ev.Jp[b] + ev.Jp[6 + a] * ev.Jp[6 + b] + ev.Jp[12 + a] * ev.Jp[12 + b];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            acc[21 + a] -= ev.Jp[a] * ev.r[0] + ev.Jp[6 + a] * ev.r[1] + ev.Jp[12 + a] * ev.r[2];
            acc[27 + a] += ev.Jp[a] * ev.Jp[a] + ev.Jp[6 + a] * ev.Jp[6 + a] + ev.Jp[12 + a] * ev.Jp[12 + a];
        }
        acc[PS_NPOSE_ACC] += ev.cost;
    }
#pragma unroll
    for (int k = 0; k <= PS_NPOSE_ACC; ++k) {
        const double v = wave_sum(acc[k]);
        if (lane == 0) red[w][k] = v;
    }
    __syncthreads();
    if (t <= PS_NPOSE_ACC) {
        double v = 0.0;
#pragma unroll
        for (int ww = 0; ww < NWV; ++ww) v += red[ww][t];
        tot[t] = v;
    }
    __syncthreads();
    double sq = 0.0;
    if (t == 0) {
        // H = J^T J (+ lambda diag) = L L^T ;  x = H^-1 g
        double H[6][6], x[6];
        bool ok = true;
        int n = 0;
        for (int a = 0; a < 6; ++a)
            for (int b = a; b < 6; ++b) { H[a][b] = tot[n]; H[b][a] = tot[n]; ++n; }
        for (int a = 0; a < 6; ++a) H[a][a] += lambda * tot[27 + a];
        for (int j = 0; j < 6; ++j) {
            double d = H[j][j];
            for (int k = 0; k < j; ++k) d -= H[j][k] * H[j][k];
            ok = ok && (d > 0.0);
            const double l = sqrt(d);
            H[j][j] = l;
            for (int i = j + 1; i < 6; ++i) {
                double v = H[i][j];
                for (int k = 0; k < j; ++k) v -= H[i][k] * H[j][k];
                H[i][j] = v / l;
            }
        }
        for (int i = 0; i < 6; ++i) {
            double v = tot[21 + i];
            for (int k = 0; k < i; ++k) v -= H[i][k] * x[k];
            x[i] = v / H[i][i];
        }
        for (int i = 5; i >= 0; --i) {
            double v = x[i];
            for (int k = i + 1; k < 6; ++k) v -= H[k][i] * x[k];
            x[i] = v / H[i][i];
        }
        if (!ok) atomicAdd(&status[ST_DIAG_FAIL], 1);
        for (int k = 0; k < 6; ++k) { xout[(size_t)rid * 6 + k] = x[k]; sq += x[k] * x[k]; }
        const Se3 Tn = se3_mul(se3_exp(x), T);
        se3_store(poses + 12 * (size_t)pose, Tn);
        se3_store(sT, Tn);
    }
    __syncthreads();
    double cost = tot[PS_NPOSE_ACC];                     // cost at the linearisation point (linesearch == 0)
    if (linesearch) {                                    // cost after the full step
        T = se3_load(sT);
        double c = 0.0;
        for (int i = start + t; i < end; i += PS_MO_THREADS) {
            const LObs o = pobs[i];
            const double pw[3] = {points[3 * (size_t)o.point], points[3 * (size_t)o.point + 1], points[3 * (size_t)o.point + 2]};
            ReprojEval ev;
            reproj_eval<false, false>(T, pw, &o.u, groups[PS_GRP_OF(o)], ev);
            c += ev.cost;
        }
        c = wave_sum(c);
        __syncthreads();
        if (lane == 0) red[w][0] = c;
        __syncthreads();
        cost = 0.0;
#pragma unroll
        for (int ww = 0; ww < NWV; ++ww) cost += red[ww][0];
    }
    if (t == 0) {
        partials[2 * rid] = cost;
        partials[2 * rid + 1] = sq;
        __threadfence();                                 // release this pose's results ...
        s_last = atomicAdd(arrivals, 1) == nr - 1;
        __threadfence();                                 // ... acquire everybody else's
    }
    __syncthreads();
    if (!s_last) return;
    // ---- last workgroup: fixed-order totals, status, publication
    double c = 0.0, q = 0.0;
    for (int i = t; i < nr; i += PS_MO_THREADS) { c += partials[2 * i]; q += partials[2 * i + 1]; }
    // (fixed order: thread-strided partial sums, then the deterministic block reduction)
    __shared__ double lds2[32];
    block_sum2(c, q, lds2);
    if (t == 0) {
        *arrivals = 0;
        scalars[linesearch ? SC_COST : SC_LINCOST] = c;
        scalars[SC_DXP2] = q; scalars[SC_DXL2] = 0.0;
        scalars[SC_RR0] = 1.0; scalars[SC_RRFINAL] = 0.0;
        status[ST_PCG_DONE] = 1; status[ST_PCG_ITERS] = 0;
    }
    __syncthreads();
    __threadfence();
    if (hst) {
        if (t < ST_NWORDS) hst[t] = status[t];
        else if (t < ST_NWORDS + SC_NWORDS) hsc[t - ST_NWORDS] = scalars[t - ST_NWORDS];
        __syncthreads();
        if (t == 0) {
            __threadfence_system();
            *reinterpret_cast<volatile long long*>(hseq) = seq;
        }
    }
}

// ---------------------------------------------------------------------------
// back-substitution, retraction, cost, small reductions.
// `gate`: when non-null the kernel returns unless the CG has flagged convergence
// (status[ST_PCG_DONE]); ps_gn_iteration enqueues this tail right behind the CG launches
// without a host synchronisation and re-runs it in the rare case the CG needed more launches.
// ---------------------------------------------------------------------------
// covariance column (ps_covariance_column): right-hand side of H x = e_k in Schur form.  g and cvec
// are zero on entry.  kind 0: g[index*D + comp] = 1.  kind 1 (landmark slot `index`): c = column
// comp of M = C^-1, and g_j -= Z_j c for every observation of the landmark on a variable pose
// (one thread walks them: duplicates of a pose accumulate in a fixed order).
__global__ __launch_bounds__(64) void k_cov_rhs(
    int kind, int index, int comp, int D, const int32_t* __restrict__ lm_ptr, const LObs* __restrict__ lobs,
    const int32_t* __restrict__ pose_rid, const double* __restrict__ Z, const double* __restrict__ Cinv,
    double* __restrict__ g, double* __restrict__ cvec)
{
    if (threadIdx.x != 0) return;
    if (kind == 0) { g[(size_t)index * D + comp] = 1.0; return; }
    const double* m = Cinv + 6 * (size_t)index;          // M00 M10 M11 M20 M21 M22
    double c[3] = {0.0, 0.0, 0.0};
    if (comp == 0) { c[0] = m[0]; c[1] = m[1]; c[2] = m[3]; }
    else if (comp == 1) { c[1] = m[2]; c[2] = m[4]; }
    else c[2] = m[5];
    cvec[3 * (size_t)index] = c[0]; cvec[3 * (size_t)index + 1] = c[1]; cvec[3 * (size_t)index + 2] = c[2];
    for (int i = lm_ptr[index]; i < lm_ptr[index + 1]; ++i) {
        const int rid = pose_rid[PS_POSE_OF(lobs[i])];
        if (rid < 0) continue;
        const double* z = Z + 18 * (size_t)i;
        for (int a = 0; a < 6; ++a) g[(size_t)rid * 6 + a] -= z[3 * a] * c[0] + z[3 * a + 1] * c[1] + z[3 * a + 2] * c[2];
    }
}

__global__ __launch_bounds__(256) void k_backsub(
    int nv, const int32_t* __restrict__ lm_ptr, const LObs* __restrict__ lobs,
    const int32_t* __restrict__ pose_rid, const double* __restrict__ Z,
    const double* __restrict__ Cinv, const double* __restrict__ cvec,
    const double* __restrict__ xp, double* __restrict__ dxl,
    double* __restrict__ sq_part /* one partial of ||dx_l||^2 per workgroup */,
    const int32_t* __restrict__ gate,
    // fused full-step update (NULL points: back-substitution only).  Workgroups >= nblk_l retract the
    // SE(3) poses instead (nothing in the back-substitution reads `poses` or `points`).
    int nblk_l, const int32_t* __restrict__ lm_point, double* __restrict__ points,
    int P, double* __restrict__ poses, double* __restrict__ sq_part_p)
{
    __shared__ double lds[16];
    if (gate && gate[ST_PCG_DONE] != 1) return;      // (2 = CG breakdown: the host falls back, nothing is applied)
    if ((int)blockIdx.x >= nblk_l) {
        typedef PoseOps<6> G;
        const int i = (blockIdx.x - nblk_l) * blockDim.x + threadIdx.x;
        double sq = 0.0;
        const int rid = (i < P) ? pose_rid[i] : -1;
        if (rid >= 0) {
            double xi[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) { xi[k] = xp[(size_t)rid * 6 + k]; sq += xi[k] * xi[k]; }
            G::store(poses + G::W * (size_t)i, G::mul(G::exp(xi), G::load(poses + G::W * (size_t)i)));
        }
        sq = block_sum(sq, lds);
        if (threadIdx.x == 0) sq_part_p[blockIdx.x - nblk_l] = sq;
        return;
    }
    // 16 lanes per landmark, one observation per lane (same mapping as k_landmark_pass)
    const int v = blockIdx.x * (blockDim.x / PS_LM_GROUP) + threadIdx.x / PS_LM_GROUP;
    const int sub = threadIdx.x & (PS_LM_GROUP - 1);
    const bool live = v < nv;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    if (live) {
        for (int i = lm_ptr[v] + sub; i < lm_ptr[v + 1]; i += PS_LM_GROUP) {
            const int rid = pose_rid[PS_POSE_OF(lobs[i])];
            if (rid < 0) continue;
            const double* z = Z + 18 * (size_t)i;
            const double* x = xp + 6 * (size_t)rid;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                const double xa = x[a];
                a0 -= z[3 * a] * xa; a1 -= z[3 * a + 1] * xa; a2 -= z[3 * a + 2] * xa;
            }
        }
    }
    a0 = group16_sum(a0); a1 = group16_sum(a1); a2 = group16_sum(a2);
    double sq = 0.0;
    if (live && sub == 0) {
        a0 += cvec[3 * (size_t)v]; a1 += cvec[3 * (size_t)v + 1]; a2 += cvec[3 * (size_t)v + 2];
        const double* m = Cinv + 6 * (size_t)v;    // dx = M^T a
        const double d0 = m[0] * a0 + m[1] * a1 + m[3] * a2;
        const double d1 = m[2] * a1 + m[4] * a2;
        const double d2 = m[5] * a2;
        dxl[3 * (size_t)v] = d0; dxl[3 * (size_t)v + 1] = d1; dxl[3 * (size_t)v + 2] = d2;
        sq = d0 * d0 + d1 * d1 + d2 * d2;
        if (points) {
            double* pt = points + 3 * (size_t)lm_point[v];
            pt[0] += d0; pt[1] += d1; pt[2] += d2;
        }
    }
    sq = block_sum(sq, lds);
    if (threadIdx.x == 0) sq_part[blockIdx.x] = sq;
}

template <int D>
__global__ __launch_bounds__(256) void k_update_poses(
    int P, const int32_t* __restrict__ pose_rid, const double* __restrict__ xp,
    double step, double* __restrict__ poses, double* __restrict__ sq_part /* per workgroup, or null */,
    const int32_t* __restrict__ gate)
{
    typedef PoseOps<D> G;
    __shared__ double lds[16];
    if (gate && gate[ST_PCG_DONE] != 1) return;      // (2 = CG breakdown: the host falls back, nothing is applied)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double sq = 0.0;
    const int rid = (i < P) ? pose_rid[i] : -1;
    if (rid >= 0) {
        double xi[D];
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const double v = xp[(size_t)rid * D + k];
            sq += v * v;
            xi[k] = step * v;
        }
        G::store(poses + G::W * (size_t)i, G::mul(G::exp(xi), G::load(poses + G::W * (size_t)i)));
    }
    if (sq_part) {
        sq = block_sum(sq, lds);
        if (threadIdx.x == 0) sq_part[blockIdx.x] = sq;
    }
}

__global__ __launch_bounds__(256) void k_update_points(
    int nv, const int32_t* __restrict__ lm_point, const double* __restrict__ dxl,
    double step, double* __restrict__ points, const int32_t* __restrict__ gate)
{
    if (gate && gate[ST_PCG_DONE] != 1) return;      // (2 = CG breakdown: the host falls back, nothing is applied)
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 3 * nv) return;
    points[3 * (size_t)lm_point[t / 3] + t % 3] += step * dxl[t];
}

// robust cost of the reprojection blocks: one partial per workgroup
__global__ __launch_bounds__(256) void k_cost_reproj(
    long n, const LObs* __restrict__ lobs, const double* __restrict__ poses,
    const double* __restrict__ points, const int32_t* __restrict__ pose_rid,
    const int32_t* __restrict__ point_vid, const ObsGroup* __restrict__ groups,
    int include_all, double* __restrict__ partials, const int32_t* __restrict__ gate)
{
    __shared__ double lds[16];
    if (gate && gate[ST_PCG_DONE] != 1) return;      // (2 = CG breakdown: the host falls back, nothing is applied)
    double c = 0.0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const LObs o = lobs[i];
        const int pose = PS_POSE_OF(o);
        if (!include_all && pose_rid[pose] < 0 && point_vid[o.point] < 0) continue;
        const Se3 T = se3_load(poses + 12 * pose);
        const double pw[3] = {points[3 * o.point], points[3 * o.point + 1], points[3 * o.point + 2]};
        ReprojEval ev;
        reproj_eval<false, false>(T, pw, &o.u, groups[PS_GRP_OF(o)], ev);
        c += ev.cost;
    }
    c = block_sum(c, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = c;
}

template <int D>
__global__ __launch_bounds__(256) void k_cost_factors(
    int nf, const int32_t* __restrict__ f_i, const int32_t* __restrict__ f_j,
    const double* __restrict__ f_Tinv, const int32_t* __restrict__ f_grp,
    const FactorGroup* __restrict__ groups, const double* __restrict__ poses,
    const int32_t* __restrict__ pose_rid, int include_all, double* __restrict__ partials,
    const int32_t* __restrict__ gate)
{
    typedef PoseOps<D> G;
    __shared__ double lds[16];
    if (gate && gate[ST_PCG_DONE] != 1) return;      // (2 = CG breakdown: the host falls back, nothing is applied)
    double cst = 0.0;
    for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < nf; f += gridDim.x * blockDim.x) {
        const int i = f_i[f], j = f_j[f];
        if (!include_all && pose_rid[j] < 0 && (i < 0 || pose_rid[i] < 0)) continue;
        const FactorGroup& grp = groups[f_grp[f]];
        const typename G::T T2 = G::load(poses + G::W * (size_t)j);
        const typename G::T To = G::load(f_Tinv + G::W * (size_t)f);
        typename G::T E;
        if (i >= 0) E = G::mul(T2, G::mul(G::inv(G::load(poses + G::W * (size_t)i)), To));
        else E = G::mul(T2, To);
        double xi[D];
        G::log(E, xi);
#pragma unroll
        for (int k = 0; k < D; ++k) {
            double rk = 0.0;
#pragma unroll
            for (int m = 0; m < D; ++m) rk += grp.S[k * D + m] * xi[m];
            cst += ps_loss_rho(grp.loss_id, grp.loss_k, rk);
        }
    }
    cst = block_sum(cst, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = cst;
}

__global__ __launch_bounds__(256) void k_sumsq_partials(
    long n, const double* __restrict__ v, double scale, double* __restrict__ partials)
{
    __shared__ double lds[16];
    double s = 0.0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const double a = scale * v[i];
        s += a * a;
    }
    s = block_sum(s, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// up to three independent sums in ONE launch: workgroup b reduces partials_b[0..n_b) into out_b
// (fixed order).  Used for {cost, ||dx_pose||^2, ||dx_point||^2} at the end of an iteration.
__global__ __launch_bounds__(256) void k_reduce3(
    int n0, const double* __restrict__ p0, double* __restrict__ o0,
    int n1, const double* __restrict__ p1, double* __restrict__ o1,
    int n2, const double* __restrict__ p2, double* __restrict__ o2, const int32_t* __restrict__ gate,
    // publish (hst != NULL): the status words and the scalar slots go straight to pinned host memory, and
    // the last workgroup to finish stamps a sequence number behind them, so the host ends the iteration
    // by watching that word instead of paying two device-to-host copies and a stream synchronisation
    const int32_t* __restrict__ status, const double* __restrict__ scalars,
    int32_t* __restrict__ hst, double* __restrict__ hsc,
    int32_t* __restrict__ arrivals /* device word, 0 between launches */, long long* __restrict__ hseq, long long seq)
{
    __shared__ double lds[16];
    const bool open = !(gate && gate[ST_PCG_DONE] != 1);
    if (hst && blockIdx.x == 0) {
        const int t = threadIdx.x;
        if (t < ST_NWORDS) hst[t] = status[t];
        else if (t < ST_NWORDS + SC_NWORDS) {
            const int k = t - ST_NWORDS;                 // slots owned by a reduction below are written there
            if (!open || (o0 != scalars + k && o1 != scalars + k && o2 != scalars + k)) hsc[k] = scalars[k];
        }
    }
    const int n = blockIdx.x == 0 ? n0 : (blockIdx.x == 1 ? n1 : n2);
    const double* p = blockIdx.x == 0 ? p0 : (blockIdx.x == 1 ? p1 : p2);
    double* o = blockIdx.x == 0 ? o0 : (blockIdx.x == 1 ? o1 : o2);
    if (open && o) {                                     // block-uniform condition
        double s = 0.0;
        for (int i = threadIdx.x; i < n; i += 256) s += p[i];
        s = block_sum(s, lds);
        if (threadIdx.x == 0) {
            o[0] = s;
            if (hsc && o >= scalars && o < scalars + SC_NWORDS) hsc[o - scalars] = s;
        }
    }
    if (hseq) {
        __syncthreads();                                 // every host-bound store of this workgroup is issued
        if (threadIdx.x == 0) {
            __threadfence_system();
            if (atomicAdd(arrivals, 1) == (int)gridDim.x - 1) {
                *arrivals = 0;
                __threadfence_system();
                *reinterpret_cast<volatile long long*>(hseq) = seq;
            }
        }
    }
}

// sharded iteration: status, scalars and the all-reduced {cost, ||dx_point||^2} to pinned host memory,
// then the sequence word the host is watching (single workgroup)
__global__ __launch_bounds__(64) void k_publish(
    const int32_t* __restrict__ status, const double* __restrict__ scalars, const double* __restrict__ shard,
    int32_t* __restrict__ hst, double* __restrict__ hsc, double* __restrict__ hshard,
    long long* __restrict__ hseq, long long seq)
{
    const int t = threadIdx.x;
    if (t < ST_NWORDS) hst[t] = status[t];
    else if (t < ST_NWORDS + SC_NWORDS) hsc[t - ST_NWORDS] = scalars[t - ST_NWORDS];
    else if (t < ST_NWORDS + SC_NWORDS + 2) hshard[t - ST_NWORDS - SC_NWORDS] = shard[t - ST_NWORDS - SC_NWORDS];
    __syncthreads();
    if (t == 0) {
        __threadfence_system();
        *reinterpret_cast<volatile long long*>(hseq) = seq;
    }
}

__global__ __launch_bounds__(256) void k_reduce_partials(int n, const double* __restrict__ partials,
                                                          double* __restrict__ out)
{
    __shared__ double lds[16];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += partials[i];
    s = block_sum(s, lds);
    if (threadIdx.x == 0) out[0] = s;
}

// debug tap: IRLS-scaled residual / Jacobian blocks in ORIGINAL observation order
__global__ __launch_bounds__(256) void k_debug_reproj(
    long n, const LObs* __restrict__ lobs, const int32_t* __restrict__ lorig,
    const double* __restrict__ poses, const double* __restrict__ points,
    const ObsGroup* __restrict__ groups, double* __restrict__ r, double* __restrict__ jp,
    double* __restrict__ jl)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const LObs o = lobs[i];
    const Se3 T = se3_load(poses + 12 * PS_POSE_OF(o));
    const double pw[3] = {points[3 * o.point], points[3 * o.point + 1], points[3 * o.point + 2]};
    ReprojEval ev;
    reproj_eval<true, true>(T, pw, &o.u, groups[PS_GRP_OF(o)], ev);
    const size_t k = (size_t)lorig[i];
    for (int a = 0; a < 3; ++a) r[3 * k + a] = ev.r[a];
    for (int a = 0; a < 18; ++a) jp[18 * k + a] = ev.Jp[a];
    for (int a = 0; a < 9; ++a) jl[9 * k + a] = ev.Jl[a];
}

// ---------------------------------------------------------------------------
// dense generic path: H = J^T J, g = -J^T r, in-place Cholesky solve (one workgroup)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dense_normal(int m, int n, const double* __restrict__ J,
                                                       const double* __restrict__ r,
                                                       double* __restrict__ H, double* __restrict__ g)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n * n) {
        const int a = t / n, b = t % n;
        double s = 0.0;
        for (int k = 0; k < m; ++k) s += J[(size_t)k * n + a] * J[(size_t)k * n + b];
        H[t] = s;
    }
    if (t < n) {
        double s = 0.0;
        for (int k = 0; k < m; ++k) s -= J[(size_t)k * n + t] * r[k];
        g[t] = s;
    }
}

// H (n x n, row-major, overwritten by its lower Cholesky factor); B (n x nrhs, row-major) <- H^-1 B
__global__ __launch_bounds__(256) void k_dense_chol_solve(int n, int nrhs, double* __restrict__ H,
                                                           double* __restrict__ B, int32_t* __restrict__ status)
{
    const int t = threadIdx.x;
    for (int j = 0; j < n; ++j) {
        __syncthreads();
        if (t == 0) {
            double d = H[(size_t)j * n + j];
            for (int k = 0; k < j; ++k) d -= H[(size_t)j * n + k] * H[(size_t)j * n + k];
            if (!(d > 0.0)) atomicAdd(&status[ST_DIAG_FAIL], 1);
            H[(size_t)j * n + j] = sqrt(d);
        }
        __syncthreads();
        const double l = H[(size_t)j * n + j];
        for (int i = j + 1 + t; i < n; i += 256) {
            double v = H[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) v -= H[(size_t)i * n + k] * H[(size_t)j * n + k];
            H[(size_t)i * n + j] = v / l;
        }
    }
    __syncthreads();
    for (int c = t; c < nrhs; c += 256) {            // one right-hand side per thread
        for (int i = 0; i < n; ++i) {                // L y = b
            double v = B[(size_t)i * nrhs + c];
            for (int k = 0; k < i; ++k) v -= H[(size_t)i * n + k] * B[(size_t)k * nrhs + c];
            B[(size_t)i * nrhs + c] = v / H[(size_t)i * n + i];
        }
        for (int i = n - 1; i >= 0; --i) {           // L^T x = y
            double v = B[(size_t)i * nrhs + c];
            for (int k = i + 1; k < n; ++k) v -= H[(size_t)k * n + i] * B[(size_t)k * nrhs + c];
            B[(size_t)i * nrhs + c] = v / H[(size_t)i * n + i];
        }
    }
}
