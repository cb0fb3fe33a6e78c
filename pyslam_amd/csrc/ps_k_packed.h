// ps_k_packed.h -- landmark pass and back-substitution with the wave's lanes PACKED BY OBSERVATION (round 5).
// Part of ps_kernels.h (included from there, after ps_k_linearize.h and ps_k_tail.h; not a stand-alone header).
#pragma once

// ---------------------------------------------------------------------------
// k_landmark_pass / k_backsub give every landmark a fixed group of 16 lanes (one DPP row: the sums are row operations), one
// observation per lane: at BASELINE's ten observations per landmark 40 of a wave's 64 lanes work, and the SQ counters of round 5
// (tools/pmc_sq.sh) show the landmark pass spending its time exactly there -- 561 VALU instructions per wave, 35 % VALU
// utilisation, 1.9 waves resident per SIMD: the evaluation of the observations, not the memory system, is what it waits for.
// Here a wave takes a RUN of consecutive landmarks whose observations fill its 64 lanes, one observation per lane:
//   * wave w owns the landmarks whose first observation row lies in [W w, W (w + 1)), W = 64 - (longest track - 1): their rows
//     are one contiguous range of at most 64 (lmw_first[w .. w + 1], built at create time by a binary search per wave);
//   * a lane finds its landmark from the head mask of the run (one ballot): segment index = popcount below it;
//   * the nine sums H_ll (6) | b_l (3) of a landmark are formed by its head lane from the lanes' values in LDS, in observation
//     order (a fixed order: deterministic), the 3 x 3 factor by the head lane, C^-1 back to the segment's lanes through LDS.
// 55-63 of 64 lanes busy instead of 40: a third fewer waves for the same observations.  Problems with a track longer than 16
// observations keep the 16-lane kernels (their second sweep handles any length).
// ---------------------------------------------------------------------------
#define PS_LMW_MAXOBS 16

// longest and shortest track (observations of one variable landmark): out[0] = max, out[1] = min (initialised to 0, INT_MAX)
__global__ __launch_bounds__(256) void k_lmw_maxobs(int nv, const int32_t* __restrict__ lm_ptr, int32_t* __restrict__ out) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = v < nv ? lm_ptr[v + 1] - lm_ptr[v] : -1;
    int mx = n, mn = n < 0 ? INT_MAX : n;
    for (int off = 32; off; off >>= 1) { mx = max(mx, __shfl_xor(mx, off, 64)); mn = min(mn, __shfl_xor(mn, off, 64)); }
    if ((threadIdx.x & 63) == 0 && mx >= 0) { atomicMax(out, mx); atomicMin(out + 1, mn); }
}

// first[w] = the first landmark whose first row is >= W w  (first[nwaves] = nv)
__global__ __launch_bounds__(256) void k_lmw_items(int nv, int nwaves, int W, const int32_t* __restrict__ lm_ptr, int32_t* __restrict__ first) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w > nwaves) return;
    if (w == nwaves) { first[w] = nv; return; }
    const int target = W * w;
    int lo = 0, hi = nv;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (lm_ptr[mid] < target) lo = mid + 1; else hi = mid; }
    first[w] = lo;
}

// the run's segment structure for this lane: landmark v, its first lane, one past its last lane; `nrows` rows in the run
struct LmwSeg { int v, lane0, lane1, nrows, row0; bool valid, head; };
PS_DEV LmwSeg lmw_segment(int v0, int v1, const int32_t* __restrict__ lm_ptr, int lane, volatile int32_t* flags /* LDS, 64 */) {
    LmwSeg s;
    const int nlm = v1 - v0;                                 // (<= 64: every landmark of a run has at least one row, a run at most 64)
    const int start = (lane < nlm) ? lm_ptr[v0 + lane] : 0;  // lane k: first row of landmark v0 + k
    s.row0 = lm_ptr[v0];
    s.nrows = lm_ptr[v1] - s.row0;
    flags[lane] = 0;
    __builtin_amdgcn_wave_barrier();
    if (lane < nlm) flags[start - s.row0] = 1;               // (a run has at most 64 rows, so at most 64 landmarks)
    __builtin_amdgcn_wave_barrier();
    const unsigned long long hm = __ballot(flags[lane] != 0);
    const unsigned long long le = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
    const unsigned long long below = hm & le, above = hm & ~le;
    s.valid = lane < s.nrows;
    s.v = v0 + __popcll(below) - 1;
    s.lane0 = below ? 63 - __clzll(below) : 0;
    s.lane1 = above ? __ffsll((long long)above) - 1 : s.nrows;
    s.lane1 = min(s.lane1, s.nrows);
    s.head = s.valid && lane == s.lane0;
    return s;
}

// COST (round 5): the pass ALSO sums the robust cost of its observations at this linearisation point (ReprojEval::cost, which
// the evaluation forms anyway) into one partial per workgroup -- the "cost after the step" of an iteration IS the cost at the
// next iteration's linearisation point, so the tail of an iteration that expects a successor runs the successor's landmark pass
// in place of its cost pass (gn_tail, ps_host_cg.h): one evaluation of every observation instead of two.  Such a launch is
// gated like the rest of the tail (`gate`: a no-op until the reduced solve has converged), and a landmark block that is not
// positive definite is reported through `fail_word` (pinned host memory, stamped with `fail_tag`) instead of the status words:
// it belongs to the NEXT call's linearisation, whose status the host folds it into (wait_published).
template <bool WIDE, bool COST>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PS_LM_WAVES, 8))) void k_landmark_pass_packed(
    int nwaves, const int32_t* __restrict__ lmw_first, const int32_t* __restrict__ lm_ptr, const int32_t* __restrict__ lm_point,
    const LObs* __restrict__ lobs, const double* __restrict__ poses,
    const double* __restrict__ points, const int32_t* __restrict__ pose_rid,
    const ObsGroup* __restrict__ groups, double lambda,
    double* __restrict__ Z, double* __restrict__ Cinv, double* __restrict__ cvec,
    int32_t* __restrict__ status, ObsWide wide,
    const int32_t* __restrict__ gate = nullptr, double* __restrict__ cost_part = nullptr,
    long long* __restrict__ fail_word = nullptr, long long fail_tag = 0)
{
    __shared__ __attribute__((aligned(16))) double zst[4][64 * PS_ZROW];    // sums (64 x 9), then C^-1 per segment, then the Z rows
    __shared__ int32_t flags[4][64];
    __shared__ double csum[16];
    if (COST && gate && gate[ST_PCG_DONE] != 1) return;      // (2 = CG breakdown: the host falls back, nothing is applied)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, gw = blockIdx.x * 4 + wv;
    double obs_cost = 0.0;
    const int v0 = gw < nwaves ? lmw_first[gw] : 0, v1 = gw < nwaves ? lmw_first[gw + 1] : 0;
    if (v1 > v0) {                                            // (wave-uniform)
    const LmwSeg sg = lmw_segment(v0, v1, lm_ptr, lane, flags[wv]);
    double* sm = zst[wv];
    const int i = sg.row0 + lane;
    ReprojEval ev;
    bool variable_pose = false;
    int rid_of_obs = -1;
    double hb[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (sg.valid) {
        const int pt = lm_point[sg.v];
        const double pw[3] = {points[3 * (size_t)pt], points[3 * (size_t)pt + 1], points[3 * (size_t)pt + 2]};
        const LObs o = lobs[i];
        const int pose = PS_POSE_OF(o);
        const Se3 T = se3_load(poses + 12 * pose);
        rid_of_obs = pose_rid[pose];
        variable_pose = rid_of_obs >= 0;
        reproj_eval_obs<true, true, WIDE>(T, pw, &o.u, groups, PS_GRP_OF(o), wide, i, ev);
        if (COST) obs_cost = ev.cost;
        const double* J = ev.Jl;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            hb[0] += J[3 * k] * J[3 * k];
            hb[1] += J[3 * k + 1] * J[3 * k];
            hb[2] += J[3 * k + 1] * J[3 * k + 1];
            hb[3] += J[3 * k + 2] * J[3 * k];
            hb[4] += J[3 * k + 2] * J[3 * k + 1];
            hb[5] += J[3 * k + 2] * J[3 * k + 2];
            hb[6] -= J[3 * k] * ev.r[k];
            hb[7] -= J[3 * k + 1] * ev.r[k];
            hb[8] -= J[3 * k + 2] * ev.r[k];
        }
    }
    // (row stride 9 doubles = 18 dwords: lanes l and l + 32 / 9 ... share banks only every 32 rows)
#pragma unroll
    for (int k = 0; k < 9; ++k) sm[9 * lane + k] = hb[k];
    __builtin_amdgcn_wave_barrier();
    // the nine sums of a landmark, spread over its lanes: lane j of an n-lane segment forms quantities j, j + n, ... (each over the
    // landmark's observations IN ORDER: a fixed summation order) -- ten dependent LDS reads per lane at ten observations instead of
    // ninety by the head lane alone; the results go to the head's own nine slots once every lane has read
    {
        const int jpos = lane - sg.lane0, nseg = sg.lane1 - sg.lane0;
        double part[9];
#pragma unroll
        for (int it = 0; it < 9; ++it) {
            const int q = jpos + it * nseg;
            part[it] = 0.0;
            if (sg.valid && q < 9) {
                double acc = 0.0;
                for (int l = sg.lane0; l < sg.lane1; ++l) acc += sm[9 * l + q];
                part[it] = acc;
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 9; ++it) {
            const int q = jpos + it * nseg;
            if (sg.valid && q < 9) sm[9 * sg.lane0 + q] = part[it];
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (sg.head) {
        double a[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) a[k] = sm[9 * lane + k];
        const double damp = 1.0 + lambda;
        const double H00 = a[0] * damp, H10 = a[1], H11 = a[2] * damp, H20 = a[3], H21 = a[4], H22 = a[5] * damp;
        // H_ll = C C^T, M = C^-1 (lower): the reciprocal roots first, every quotient a product (as k_landmark_pass)
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
        if (!(H00 > 0.0) || !(d1 > 0.0) || !(d2 > 0.0)) {
            if (COST && fail_word) *reinterpret_cast<volatile long long*>(fail_word) = fail_tag;
            else atomicAdd(&status[ST_LM_FAIL], 1);
        }
        double* ci = Cinv + 6 * (size_t)sg.v;
        ci[0] = M00; ci[1] = M10; ci[2] = M11; ci[3] = M20; ci[4] = M21; ci[5] = M22;
        double* cv = cvec + 3 * (size_t)sg.v;
        cv[0] = M00 * a[6];
        cv[1] = M10 * a[6] + M11 * a[7];
        cv[2] = M20 * a[6] + M21 * a[7] + M22 * a[8];
        double* mo = sm + 9 * lane;                           // the head's own slot: its sums are consumed
        mo[0] = M00; mo[1] = M10; mo[2] = M11; mo[3] = M20; mo[4] = M21; mo[5] = M22;
    }
    __builtin_amdgcn_wave_barrier();
    double M[6] = {0, 0, 0, 0, 0, 0};
    if (sg.valid) {
#pragma unroll
        for (int k = 0; k < 6; ++k) M[k] = sm[9 * sg.lane0 + k];
    }
    __builtin_amdgcn_wave_barrier();                          // every lane has its C^-1: the area becomes the Z staging
    if (sg.valid) {
        double z[PS_ZROW];
#pragma unroll
        for (int k = 0; k < PS_ZROW; ++k) z[k] = 0.0;        // rows of constant poses are never read
        z[12] = -1.0;
        if (variable_pose) {
            lm_emit_m(ev, M[0], M[1], M[2], M[3], M[4], M[5], z);
            z[9] = ev.pc[0]; z[10] = ev.pc[1]; z[11] = ev.pc[2];
            z[12] = (double)rid_of_obs;
        }
        double2* dst = reinterpret_cast<double2*>(sm + PS_ZROW * lane);
#pragma unroll
        for (int k = 0; k < PS_ZROW / 2; ++k) dst[k] = make_double2(z[2 * k], z[2 * k + 1]);
    }
    __builtin_amdgcn_wave_barrier();
    const double2* src = reinterpret_cast<const double2*>(sm);
    double2* out = reinterpret_cast<double2*>(Z + PS_ZROW * (size_t)sg.row0);
    for (int k = lane; k < sg.nrows * (PS_ZROW / 2); k += 64) out[k] = src[k];
    }
    if (COST) {                                               // one partial per workgroup, fixed order (lanes, then waves)
        obs_cost = block_sum(obs_cost, csum);
        if (threadIdx.x == 0) cost_part[blockIdx.x] = obs_cost;
    }
}

// The robust cost alone, in the landmark pass's structure: wave w evaluates the rows of run w, one per lane, a workgroup writes
// one partial -- the SAME partial sums, bit for bit, as k_landmark_pass_packed<.., true> forms (the cost of an observation is
// a function of its inputs alone: ps_math.h), so a cost is the same number whichever of the two produced it.
template <bool WIDE>
__global__ __launch_bounds__(256) void k_cost_packed(
    int nwaves, const int32_t* __restrict__ lmw_first, const int32_t* __restrict__ lm_ptr, const int32_t* __restrict__ lm_point,
    const LObs* __restrict__ lobs, const double* __restrict__ poses, const double* __restrict__ points,
    const ObsGroup* __restrict__ groups, double* __restrict__ cost_part, const int32_t* __restrict__ gate, ObsWide wide)
{
    __shared__ double csum[16];
    if (gate && gate[ST_PCG_DONE] != 1) return;      // (2 = CG breakdown: the host falls back, nothing is applied)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, gw = blockIdx.x * 4 + wv;
    double c = 0.0;
    const int v0 = gw < nwaves ? lmw_first[gw] : 0, v1 = gw < nwaves ? lmw_first[gw + 1] : 0;
    if (v1 > v0) {
        const int row0 = lm_ptr[v0], nrows = lm_ptr[v1] - row0;
        if (lane < nrows) {
            const int i = row0 + lane;
            const LObs o = lobs[i];
            const Se3 T = se3_load(poses + 12 * PS_POSE_OF(o));
            const double pw[3] = {points[3 * (size_t)o.point], points[3 * (size_t)o.point + 1], points[3 * (size_t)o.point + 2]};
            ReprojEval ev;
            reproj_eval_obs<false, false, WIDE>(T, pw, &o.u, groups, PS_GRP_OF(o), wide, i, ev);
            c = ev.cost;
        }
    }
    c = block_sum(c, csum);
    if (threadIdx.x == 0) cost_part[blockIdx.x] = c;
}

// back-substitution, the same packing: one Z row per lane, the landmark's three sums by its head lane
__global__ __launch_bounds__(256) void k_backsub_packed(
    int nwaves, const int32_t* __restrict__ lmw_first, const int32_t* __restrict__ lm_ptr,
    const double* __restrict__ Z, const double* __restrict__ Cinv, const double* __restrict__ cvec,
    const double* __restrict__ xp, double* __restrict__ dxl,
    double* __restrict__ sq_part /* one partial of ||dx_l||^2 per workgroup */,
    const int32_t* __restrict__ gate,
    // fused full-step update (NULL points: back-substitution only); workgroups >= nblk_l retract the SE(3) poses (as k_backsub)
    int nblk_l, const int32_t* __restrict__ lm_point, double* __restrict__ points,
    int P, const int32_t* __restrict__ pose_rid, double* __restrict__ poses, double* __restrict__ sq_part_p,
    long long* __restrict__ hearly, long long eseq)
{
    __shared__ double lds[16];
    __shared__ double sm3[4][64 * 3];
    __shared__ int32_t flags[4][64];
    const bool closed = gate && gate[ST_PCG_DONE] != 1;
    if (hearly && blockIdx.x == 0 && threadIdx.x == 0) *reinterpret_cast<volatile long long*>(hearly) = closed ? -eseq : eseq;
    if (closed) return;                              // (2 = CG breakdown: the host falls back, nothing is applied)
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
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, gw = blockIdx.x * 4 + wv;
    double sq = 0.0;
    const int v0 = gw < nwaves ? lmw_first[gw] : 0, v1 = gw < nwaves ? lmw_first[gw + 1] : 0;
    if (v1 > v0) {                                            // (wave-uniform)
        const LmwSeg sg = lmw_segment(v0, v1, lm_ptr, lane, flags[wv]);
        double a0 = 0.0, a1 = 0.0, a2 = 0.0;
        if (sg.valid) {
            const double2* zq = reinterpret_cast<const double2*>(Z + PS_ZROW * (size_t)(sg.row0 + lane));
            double z[14];
#pragma unroll
            for (int k = 0; k < 7; ++k) { const double2 t = zq[k]; z[2 * k] = t.x; z[2 * k + 1] = t.y; }
            const int rid = (int)z[12];
            if (rid >= 0) {
                const double* x = xp + 6 * (size_t)rid;
                // Z^T x = M^T (x_rho - pc x x_phi)
                const double y0 = x[0] - (z[10] * x[5] - z[11] * x[4]);
                const double y1 = x[1] - (z[11] * x[3] - z[9] * x[5]);
                const double y2 = x[2] - (z[9] * x[4] - z[10] * x[3]);
                a0 = -(z[0] * y0 + z[3] * y1 + z[6] * y2);
                a1 = -(z[1] * y0 + z[4] * y1 + z[7] * y2);
                a2 = -(z[2] * y0 + z[5] * y1 + z[8] * y2);
            }
        }
        double* sm = sm3[wv];
        sm[3 * lane] = a0; sm[3 * lane + 1] = a1; sm[3 * lane + 2] = a2;
        __builtin_amdgcn_wave_barrier();
        if (sg.head) {
            double b0 = 0.0, b1 = 0.0, b2 = 0.0;
            for (int l = lane; l < sg.lane1; ++l) { b0 += sm[3 * l]; b1 += sm[3 * l + 1]; b2 += sm[3 * l + 2]; }
            const size_t v = (size_t)sg.v;
            b0 += cvec[3 * v]; b1 += cvec[3 * v + 1]; b2 += cvec[3 * v + 2];
            const double* m = Cinv + 6 * v;        // dx = M^T a
            const double d0 = m[0] * b0 + m[1] * b1 + m[3] * b2;
            const double d1 = m[2] * b1 + m[4] * b2;
            const double d2 = m[5] * b2;
            dxl[3 * v] = d0; dxl[3 * v + 1] = d1; dxl[3 * v + 2] = d2;
            sq = d0 * d0 + d1 * d1 + d2 * d2;
            if (points) {
                double* pt = points + 3 * (size_t)lm_point[v];
                pt[0] += d0; pt[1] += d1; pt[2] += d2;
            }
        }
    }
    sq = block_sum(sq, lds);
    if (threadIdx.x == 0) sq_part[blockIdx.x] = sq;
}
