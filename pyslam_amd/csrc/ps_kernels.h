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
// This header holds the shared types, status words and reductions; the kernels live in the ps_k_*.h parts included at
// the end (linearize, pcg_classic, cg_fused, xcg, coarse, band, tail), in pipeline order.
#pragma once
#include "ps_math.h"

struct __attribute__((aligned(32))) LObs {   // one reprojection observation, 32 B
    double u, v, d;
    int32_t pose_grp;      // pose index (low 24 bits) | group (high 8 bits)
    int32_t point;
};
struct PItem { int32_t rid, start, end, pad; };
struct PairItem { int32_t slot, slotT, start, end; };
// pose-stationary Schur kernel: a segment of one pose's Z rows (row_start into the row list) and its tasks
struct PoseSeg { int32_t row_start, row_count, task_start, task_end; };
#define PS_PP_SEG 512                         // a-rows in LDS; a segment holds at most PS_PP_SEG - 1 (index 511 marks a padding word)
struct FactorGroup { double S[36]; int32_t loss_id; int32_t pad; double loss_k; };

#define PS_POSE_OF(o) ((o).pose_grp & 0xFFFFFF)
#define PS_GRP_OF(o) (((uint32_t)(o).pose_grp) >> 24)

// status words
enum { ST_LM_FAIL = 0, ST_DIAG_FAIL = 1, ST_PCG_DONE = 2, ST_PCG_ITERS = 3, ST_PERSIST_FAIL = 4 /* k_cg_persist: an exchange timed out */, ST_NWORDS = 8 };
// scalar slots
enum { SC_COST = 0, SC_DXP2 = 1, SC_LINCOST = 2, SC_RR0 = 3, SC_RRFINAL = 4, SC_THRESH = 5, SC_DXL2 = 6, SC_STARTCOST = 7, SC_NWORDS = 8 };

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

#include "ps_k_linearize.h"
#include "ps_k_schur2.h"
#include "ps_k_schur3.h"
#include "ps_k_stream.h"
#include "ps_k_pcg_classic.h"
#include "ps_k_ldi.h"
#include "ps_k_cg_fused.h"
#include "ps_k_cg_persist.h"
#include "ps_k_xcg.h"
#include "ps_k_xcg_persist.h"
#include "ps_k_xcg_persist4.h"
#include "ps_k_coarse.h"
#include "ps_k_band.h"
#include "ps_k_bandpart.h"
#include "ps_k_tail.h"
#include "ps_k_packed.h"
