// ps_k_tail.h -- one-launch motion-only iteration, covariance right-hand side, back-substitution, updates, costs, final reductions, dense normal equations.
// Part of ps_kernels.h (included from there, in this order; not a stand-alone header).
#pragma once

// ---------------------------------------------------------------------------
// Motion-only problems (no variable landmark, no pose factor: the reduced system is block diagonal --
// reference pipelines/sparse.py:153-161, SURVEY config C5): ONE launch per Gauss-Newton iteration.
// One workgroup per pose: residuals + Jacobians + IRLS of its observations, 33 sums, 6 x 6 Cholesky
// solve, retraction, post-step cost; the last workgroup to arrive sums the per-pose {cost, |dx|^2}
// in pose order and publishes status + scalars to pinned host memory (sequence word).
// ---------------------------------------------------------------------------
// 6 x 6 normal equations of one pose from the 33 sums (21 upper entries, 6 gradient entries, 6 diagonal entries for the
// damping): H = J^T J + lambda diag = L L^T, x = H^-1 g.  One thread; reciprocal roots (ps_rsqrt) and products instead
// of 6 roots and 27 quotients.  false: a pivot was not positive.
PS_DEV bool mo_solve6(const double* __restrict__ tot, double lambda, double* __restrict__ x) {
    double H[6][6], il[6];
    bool ok = true;
    int n = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = a; b < 6; ++b) { H[a][b] = tot[n]; H[b][a] = tot[n]; ++n; }
#pragma unroll
    for (int a = 0; a < 6; ++a) H[a][a] += lambda * tot[27 + a];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d = H[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= H[j][k] * H[j][k];
        ok = ok && (d > 0.0);
        il[j] = ps_rsqrt(d);                                 // 1 / L_jj
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double v = H[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) v -= H[i][k] * H[j][k];
            H[i][j] = v * il[j];
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double v = tot[21 + i];
#pragma unroll
        for (int k = 0; k < i; ++k) v -= H[i][k] * x[k];
        x[i] = v * il[i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double v = x[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) v -= H[k][i] * x[k];
        x[i] = v * il[i];
    }
    return ok;
}

#define PS_MO_THREADS 512
template <bool WIDE>
__global__ __launch_bounds__(PS_MO_THREADS) void k_motion_only_iteration(
    int nr, const PItem* __restrict__ items, const int32_t* __restrict__ pitem_ptr,
    const LObs* __restrict__ pobs, const double* __restrict__ points, const ObsGroup* __restrict__ groups,
    double* __restrict__ poses, double lambda, int linesearch,
    double* __restrict__ xout /* nr x 6 */, double* __restrict__ partials /* nr x 2: cost, |dx|^2 */,
    int32_t* __restrict__ status, double* __restrict__ scalars, int32_t* __restrict__ arrivals,
    int32_t* __restrict__ hst, double* __restrict__ hsc, long long* __restrict__ hseq, long long seq, ObsWide wide)
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
        reproj_eval_obs<true, false, WIDE>(T, pw, &o.u, groups, PS_GRP_OF(o), wide, i, ev);
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
        double x[6];
        const bool ok = mo_solve6(tot, lambda, x);
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
            reproj_eval_obs<false, false, WIDE>(T, pw, &o.u, groups, PS_GRP_OF(o), wide, i, ev);
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
// The WHOLE solve of a one-pose motion-only problem in one launch (reference pipelines/sparse.py:153-161 builds such a
// Problem per frame and calls solve(): config C5).  The loop of Problem.solve (reference problem.py:130-178) -- iterate,
// record the cost, stop on max_iters / min_update_norm / min_cost / the non-decreasing-step rules, keep and restore the
// best parameters -- needs nothing but the numbers this workgroup produces, so it runs here: one launch and one
// synchronisation per FRAME instead of one per iteration (8.5 iterations on average at C5: 0.27-0.37 ms of launches and
// host round trips for ~70 us of arithmetic).  Arithmetic and summation order are those of k_motion_only_iteration; the
// cost after a step and the cost at the next linearisation point are the same sum, evaluated once.
// ---------------------------------------------------------------------------
#define PS_MO_SOLVE_OBS 4                     // observations per thread: 4 x 512 = the 2 048 of the one-launch limit
struct MoSolveOptions {          // Options of the reference (problem.py:14-40) that the loop reads
    int max_iters, allow_nondecreasing_steps, max_nondecreasing_steps, linesearch;
    double min_update_norm, min_cost, min_cost_decrease, lambda;
};

template <bool WIDE>
__global__ __launch_bounds__(PS_MO_THREADS) void k_motion_only_solve(
    const PItem* __restrict__ items, const int32_t* __restrict__ pitem_ptr,
    const LObs* __restrict__ pobs, const double* __restrict__ points, const ObsGroup* __restrict__ groups,
    double* __restrict__ poses, MoSolveOptions opt, double* __restrict__ xout /* 6: the last step */,
    int32_t* __restrict__ status, double* __restrict__ scalars,
    double* __restrict__ hist /* pinned host: [0] = entries, [1] = iterations, [2] = last |dx|, [3..14] final pose, [15..] cost history */, int hist_cap,
    int32_t* __restrict__ hst, double* __restrict__ hsc, long long* __restrict__ hseq, long long seq, ObsWide wide)
{
    constexpr int NWV = PS_MO_THREADS / 64;
    __shared__ double red[NWV][PS_NPOSE_ACC + 1];
    __shared__ double tot[PS_NPOSE_ACC + 1];
    __shared__ double sT[12], sBest[12];
    __shared__ int s_done;
    const int t = threadIdx.x, w = t >> 6, lane = t & 63;
    const int ib = pitem_ptr[0], ie = pitem_ptr[1];
    const int start = ib < ie ? items[ib].start : 0, end = ib < ie ? items[ie - 1].end : 0;
    const int pose = ib < ie ? items[ib].pad : 0;
    Se3 T = se3_load(poses + 12 * (size_t)pose);
    // observations and their (constant) landmarks: fetched ONCE, kept in registers across the iterations (at most
    // PS_MO_SOLVE_OBS per thread: the one-launch path stops at 2 048 observations per pose)
    LObs obs[PS_MO_SOLVE_OBS];
    double pwr[PS_MO_SOLVE_OBS][3];
#pragma unroll
    for (int q = 0; q < PS_MO_SOLVE_OBS; ++q) {
        const int i = start + t + q * PS_MO_THREADS;
        obs[q] = pobs[i < end ? i : (end > start ? end - 1 : 0)];
        pwr[q][0] = points[3 * (size_t)obs[q].point]; pwr[q][1] = points[3 * (size_t)obs[q].point + 1]; pwr[q][2] = points[3 * (size_t)obs[q].point + 2];
    }
    // thread 0's loop state (Problem.solve's local variables)
    double cost = 0.0, prev_cost = 0.0, last_dx = 100.0;
    int nhist = 0, iters = 0, nondecreasing = 0;
    for (;;) {
        {
            // ---- the 33 sums + the cost at T (k_motion_only_iteration's first phase, same order)
            double acc[PS_NPOSE_ACC + 1];
#pragma unroll
            for (int k = 0; k <= PS_NPOSE_ACC; ++k) acc[k] = 0.0;
#pragma unroll
            for (int q = 0; q < PS_MO_SOLVE_OBS; ++q) {
                const int i = start + t + q * PS_MO_THREADS;
                if (i >= end) break;
                ReprojEval ev;
                reproj_eval_obs<true, false, WIDE>(T, pwr[q], &obs[q].u, groups, PS_GRP_OF(obs[q]), wide, i, ev);
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
            __syncthreads();                             // (red / tot of the previous round have been read)
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
        }
        // ---- thread 0: bookkeeping of the step that led here, then the next step
        if (t == 0) {
            int done = 0;
            const double c_here = tot[PS_NPOSE_ACC];     // cost at the current T
            if (nhist == 0) {                            // the start: Problem.solve's eval_cost
                cost = c_here;
                hist[15 + nhist++] = cost;
            } else if (opt.linesearch) {                 // the step just taken is judged by the cost it led to
                cost = c_here;
                hist[15 + nhist++] = cost;
                done = iters > opt.max_iters || last_dx < opt.min_update_norm || cost < opt.min_cost;
                if (opt.allow_nondecreasing_steps) {
                    if (nondecreasing == 0) se3_store(sBest, T);
                    nondecreasing = (cost >= opt.min_cost_decrease * prev_cost) ? nondecreasing + 1 : 0;
                    if (nondecreasing >= opt.max_nondecreasing_steps) { done = 1; T = se3_load(sBest); }
                } else done = done || cost >= opt.min_cost_decrease * prev_cost;
            }
            double sq = 0.0;
            if (!done) {
                // H = J^T J (+ lambda diag) = L L^T ;  x = H^-1 g   (the function k_motion_only_iteration calls)
                double x[6];
                const bool ok = mo_solve6(tot, opt.lambda, x);
                if (!ok) { atomicAdd(&status[ST_DIAG_FAIL], 1); done = 2; }
                else {
                    for (int k = 0; k < 6; ++k) { xout[k] = x[k]; sq += x[k] * x[k]; }
                    prev_cost = cost;
                    ++iters;
                    last_dx = sqrt(sq);
                    T = se3_mul(se3_exp(x), T);
                    if (!opt.linesearch) {               // the step is judged by the cost of its linearisation point
                        hist[15 + nhist++] = c_here;
                        cost = c_here;
                        done = iters > opt.max_iters || last_dx < opt.min_update_norm || cost < opt.min_cost;
                        if (opt.allow_nondecreasing_steps) {
                            if (nondecreasing == 0) se3_store(sBest, T);
                            // (prev_cost: the cost recorded one iteration earlier -- at the first iteration the start cost,
                            //  i.e. this very number: the reference counts that as a non-decreasing step, and so does this)
                            nondecreasing = (cost >= opt.min_cost_decrease * prev_cost) ? nondecreasing + 1 : 0;
                            if (nondecreasing >= opt.max_nondecreasing_steps) { done = 1; T = se3_load(sBest); }
                        } else done = done || cost >= opt.min_cost_decrease * prev_cost;
                    }
                }
            }
            if (nhist >= hist_cap - 16 && !done) done = 3;             // (the caller sized hist for max_iters + 2 entries)
            se3_store(sT, T);
            s_done = done;
        }
        __syncthreads();
        T = se3_load(sT);
        if (s_done) break;
    }
    if (t == 0) {
        se3_store(poses + 12 * (size_t)pose, T);
        se3_store(hist + 3, T);
        hist[0] = (double)nhist; hist[1] = (double)iters; hist[2] = last_dx;
        scalars[opt.linesearch ? SC_COST : SC_LINCOST] = cost;
        scalars[SC_DXP2] = last_dx * last_dx; scalars[SC_DXL2] = 0.0;
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
        double z[18];
        zrow_expand(Z + PS_ZROW * (size_t)i, Z + PS_ZROW * (size_t)i + 9, z);
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
    int P, double* __restrict__ poses, double* __restrict__ sq_part_p,
    // early word (pinned host memory; NULL: none): has the reduced solve in front of this tail converged?  +seq / -seq.  The
    // host, which is only waiting for the end of the iteration, may enqueue the NEXT linearisation behind the tail on it
    long long* __restrict__ hearly, long long eseq)
{
    __shared__ double lds[16];
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
    // (Round 4 tried the workgroup's rows -- one contiguous range of Z -- as a dense stream through LDS instead of seven 16-byte
    //  pieces per lane at a 128-byte stride: 173 -> 198 us at C4, the staging barrier and the lost occupancy cost more than the
    //  scattered requests; not kept.)
    // 16 lanes per landmark, one observation per lane (same mapping as k_landmark_pass)
    const int v = blockIdx.x * (blockDim.x / PS_LM_GROUP) + threadIdx.x / PS_LM_GROUP;
    const int sub = threadIdx.x & (PS_LM_GROUP - 1);
    const bool live = v < nv;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    if (live) {
        for (int i = lm_ptr[v] + sub; i < lm_ptr[v + 1]; i += PS_LM_GROUP) {
            // the whole 128-byte row: M (9) | pc (3) | reduced pose index (no observation record, no pose_rid gather)
            const double2* zq = reinterpret_cast<const double2*>(Z + PS_ZROW * (size_t)i);
            double z[14];
#pragma unroll
            for (int k = 0; k < 7; ++k) { const double2 t = zq[k]; z[2 * k] = t.x; z[2 * k + 1] = t.y; }
            const int rid = (int)z[12];
            if (rid < 0) continue;
            const double* x = xp + 6 * (size_t)rid;
            // Z^T x = M^T (x_rho - pc x x_phi)
            const double y0 = x[0] - (z[10] * x[5] - z[11] * x[4]);
            const double y1 = x[1] - (z[11] * x[3] - z[9] * x[5]);
            const double y2 = x[2] - (z[9] * x[4] - z[10] * x[3]);
            a0 -= z[0] * y0 + z[3] * y1 + z[6] * y2;
            a1 -= z[1] * y0 + z[4] * y1 + z[7] * y2;
            a2 -= z[2] * y0 + z[5] * y1 + z[8] * y2;
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
    const int32_t* __restrict__ gate, long long* __restrict__ hearly = nullptr /* as k_backsub's */, long long eseq = 0)
{
    typedef PoseOps<D> G;
    __shared__ double lds[16];
    const bool closed = gate && gate[ST_PCG_DONE] != 1;
    if (hearly && blockIdx.x == 0 && threadIdx.x == 0) *reinterpret_cast<volatile long long*>(hearly) = closed ? -eseq : eseq;
    if (closed) return;                              // (2 = CG breakdown: the host falls back, nothing is applied)
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
template <bool WIDE>
__global__ __launch_bounds__(256) void k_cost_reproj(
    long n, const LObs* __restrict__ lobs, const double* __restrict__ poses,
    const double* __restrict__ points, const int32_t* __restrict__ pose_rid,
    const int32_t* __restrict__ point_vid, const ObsGroup* __restrict__ groups,
    int include_all, double* __restrict__ partials, const int32_t* __restrict__ gate, ObsWide wide)
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
        reproj_eval_obs<false, false, WIDE>(T, pw, &o.u, groups, PS_GRP_OF(o), wide, i, ev);
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

// this thread's share (elements t, t + 256, ...) of a 256-thread sum over p[0..n): eight running sums in a fixed
// order, so the loads of a trip are independent -- a single running sum pays one memory latency per element (C4:
// 31 250 landmark partials, 47 us in k_reduce3).  Every final reduction uses it: the sums agree bit for bit.
PS_DEV double strided_sum8(const double* __restrict__ p, int n) {
    double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    int i = threadIdx.x;
    for (; i + 7 * 256 < n; i += 8 * 256) {
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] += p[i + k * 256];
    }
    for (int k = 0; i < n; i += 256, ++k) a[k] += p[i];
    return ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
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
        double s = strided_sum8(p, n);
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
    double s = strided_sum8(partials, n);
    s = block_sum(s, lds);
    if (threadIdx.x == 0) out[0] = s;
}

// debug tap: IRLS-scaled residual / Jacobian blocks in ORIGINAL observation order
template <bool WIDE>
__global__ __launch_bounds__(256) void k_debug_reproj(
    long n, const LObs* __restrict__ lobs, const int32_t* __restrict__ lorig,
    const double* __restrict__ poses, const double* __restrict__ points,
    const ObsGroup* __restrict__ groups, double* __restrict__ r, double* __restrict__ jp,
    double* __restrict__ jl, ObsWide wide)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const LObs o = lobs[i];
    const Se3 T = se3_load(poses + 12 * PS_POSE_OF(o));
    const double pw[3] = {points[3 * o.point], points[3 * o.point + 1], points[3 * o.point + 2]};
    ReprojEval ev;
    reproj_eval_obs<true, true, WIDE>(T, pw, &o.u, groups, PS_GRP_OF(o), wide, i, ev);
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

// ---------------------------------------------------------------------------
// landmark-sharded iteration: the exchange buffer of the ONE sum all-reduce per iteration
//   pack = [upper block triangle of S incl. the diagonal (nup blocks) | g | cost (2) | failure flag]
// S is exactly symmetric (the Schur and factor kernels mirror every off-diagonal block), so ranks exchange
// the upper triangle only -- C4: 23 MB instead of 46 MB over xGMI -- and k_shard_unpack mirrors it back.
// The flag carries ST_LM_FAIL of every shard to all ranks: they fail together instead of one rank leaving
// the others waiting in the next collective.
// ---------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void k_shard_pack(
    long nup, const int32_t* __restrict__ up_slot, const double* __restrict__ S,
    long ntail, const double* __restrict__ tail /* g | cost */, const int32_t* __restrict__ status,
    double* __restrict__ pack)
{
    constexpr int DD = D * D;
    const long total = nup * DD + ntail + 1;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        if (t < nup * DD) pack[t] = S[(size_t)up_slot[t / DD] * DD + t % DD];
        else if (t < total - 1) pack[t] = tail[t - nup * DD];
        else pack[t] = status[ST_LM_FAIL] != 0 ? 1.0 : 0.0;
    }
}

// aggressor of ps_debug_factor_stress (aggressor = 2): workgroups that own a large dynamic LDS allocation and keep writing all of it
__global__ __launch_bounds__(512) void k_lds_scribble(int words, int rounds, double* __restrict__ sink)
{
    extern __shared__ double scr[];
    double acc = 0.0;
    for (int r = 0; r < rounds; ++r) {
        for (int i = threadIdx.x; i < words; i += blockDim.x) scr[i] = (double)(i ^ r) * 1.0e300;
        __syncthreads();
        for (int i = threadIdx.x; i < words; i += blockDim.x) acc += scr[words - 1 - i];
        __syncthreads();
    }
    if (acc == 12345.678) sink[blockIdx.x] = acc;
}

// order-independent bit checksum of n 4-byte words (measurement build: PS_XCG_INV_SUM): *out = sum of (word * (index + 1)) mod 2^64
__global__ __launch_bounds__(256) void k_bit_sum(size_t nwords, const unsigned* __restrict__ a, unsigned long long* __restrict__ out)
{
    unsigned long long s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (size_t)gridDim.x * blockDim.x) s += (unsigned long long)a[i] * (i + 1);
    if (s) atomicAdd(out, s);
}

// bitwise comparison of two arrays (measurement build: PS_XCG_AC_CHECK): cnt[slot] += entries that differ, cnt[2] += 1 per launch
__global__ __launch_bounds__(256) void k_cmp_bits(size_t n, const double* __restrict__ a, const double* __restrict__ b, int32_t* __restrict__ cnt, int slot)
{
    int bad = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        bad += __double_as_longlong(a[i]) != __double_as_longlong(b[i]);
    if (bad) atomicAdd(&cnt[slot], bad);
    if (blockIdx.x == 0 && threadIdx.x == 0 && slot == 0) atomicAdd(&cnt[2], 1);
}

// ---- segment exchange (ps_set_segment_exchange): [tail words | this rank's elements of the packed system], read straight from
// S / g / cost / status at the packed positions (k_shard_pack's layout: the upper blocks in up_slot order, then g, cost (2), the flag)
// -- no pass over the whole 23.5 MB buffer when a rank owns an eighth of it ...
template <int D>
__global__ __launch_bounds__(256) void k_seg_pack(long nup, const int32_t* __restrict__ up_slot, const double* __restrict__ S,
                                                  long ntail, const double* __restrict__ tail /* g | cost */, const int32_t* __restrict__ status,
                                                  long T, long nmine, const int32_t* __restrict__ mine, double* __restrict__ seg_in, long maxlen)
{
    constexpr int DD = D * D;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < maxlen; t += (long)gridDim.x * blockDim.x) {
        double v = 0.0;
        if (t < T) v = t < 2 ? tail[ntail - 2 + t] : (status[ST_LM_FAIL] != 0 ? 1.0 : 0.0);
        else if (t - T < nmine) {
            const long p = mine[t - T];
            v = p < nup * DD ? S[(size_t)up_slot[p / DD] * DD + p % DD] : tail[p - nup * DD];
        }
        seg_in[t] = v;
    }
}
// ... and the sum of what the ranks sent, written where k_shard_unpack would put it (both triangles of S, g, cost, the flag): every
// destination element over its contributors in rank order (no atomics: one thread per destination, a fixed order -- the same number
// on every rank), the tail words over all ranks
template <int D>
__global__ __launch_bounds__(256) void k_seg_sum(long ndst, const int32_t* __restrict__ dst, const int32_t* __restrict__ src_ptr,
                                                 const int32_t* __restrict__ src_off, const double* __restrict__ seg_all,
                                                 long T, int world, long maxlen, long nup, const int32_t* __restrict__ up_slot,
                                                 const int32_t* __restrict__ upT_slot, double* __restrict__ S, long ntail,
                                                 double* __restrict__ tail, int32_t* __restrict__ status)
{
    constexpr int DD = D * D;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < ndst + T; t += (long)gridDim.x * blockDim.x) {
        double v = 0.0;
        if (t < ndst) {
            for (int k = src_ptr[t]; k < src_ptr[t + 1]; ++k) v += seg_all[src_off[k]];
            const long p = dst[t];
            if (p < nup * DD) {
                const long b = p / DD;
                const int e = (int)(p % DD), r = e / D, c = e % D;
                const int s1 = up_slot[b], s2 = upT_slot[b];
                S[(size_t)s1 * DD + e] = v;
                if (s2 != s1) S[(size_t)s2 * DD + c * D + r] = v;
            } else tail[p - nup * DD] = v;
        } else {
            const long w = t - ndst;
            for (int r = 0; r < world; ++r) v += seg_all[(size_t)r * maxlen + w];
            if (w < 2) tail[ntail - 2 + w] = v;
            else if (v != 0.0) status[ST_LM_FAIL] = 1;          // some shard's H_ll was not positive definite
        }
    }
}

template <int D>
__global__ __launch_bounds__(256) void k_shard_unpack(
    long nup, const int32_t* __restrict__ up_slot, const int32_t* __restrict__ upT_slot,
    const double* __restrict__ pack, double* __restrict__ S, long ntail, double* __restrict__ tail,
    int32_t* __restrict__ status)
{
    constexpr int DD = D * D;
    const long total = nup * DD + ntail + 1;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        if (t < nup * DD) {
            const long b = t / DD;
            const int e = (int)(t % DD), r = e / D, c = e % D;
            const double v = pack[t];
            const int s1 = up_slot[b], s2 = upT_slot[b];
            S[(size_t)s1 * DD + e] = v;
            if (s2 != s1) S[(size_t)s2 * DD + c * D + r] = v;
        } else if (t < total - 1) tail[t - nup * DD] = pack[t];
        else if (pack[t] != 0.0) status[ST_LM_FAIL] = 1;        // some shard's H_ll was not positive definite
    }
}
