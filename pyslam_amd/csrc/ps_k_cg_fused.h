// ps_k_cg_fused.h -- fused one-launch-per-iteration CG on the scaled (augmented) system, its setup kernels, snapshot copy, direct small solve.
// Part of ps_kernels.h (included from there, in this order; not a stand-alone header).
#pragma once

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
// One WAVE per pose (4 poses per workgroup), lane (r, c) of the D x D block: the factorisation is a right-looking
// Cholesky with one column scaled and one rank-1 update applied per step by all lanes at once, the inverse six
// independent forward substitutions -- a few hundred instructions per wave instead of the ~3 000 of one thread
// working through the whole block (11 us -> 4 us at C3: this kernel heads the critical path of every reduced solve).
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
    constexpr int DD = D * D;
    __shared__ double sA[4][DD], sLi[4][DD], sB[4][DD], sR[4][D];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + wv;
    if (g && blockIdx.x == 0 && threadIdx.x == 0) { status[ST_PCG_DONE] = 0; status[ST_PCG_ITERS] = 0; }
    if (i >= nr) return;                                    // whole waves
    const bool act = lane < DD;
    const int r = act ? lane / D : 0, c = act ? lane % D : 0;
    double* A = sA[wv];
    double* Li = sLi[wv];
    if (act) { A[lane] = S[(size_t)diag_slot[i] * DD + lane]; Li[lane] = 0.0; }
    __builtin_amdgcn_wave_barrier();
    bool ok = true;
    // A = L L^T in place: after step j column j of A (rows >= j) holds L[:, j]
#pragma unroll
    for (int j = 0; j < D; ++j) {
        const double d = A[j * D + j];
        ok = ok && (d > 0.0);
        const double l = sqrt(d);
        __builtin_amdgcn_wave_barrier();
        if (act && c == j && r >= j) A[lane] = (r == j) ? l : A[lane] / l;
        __builtin_amdgcn_wave_barrier();
        if (act && r > j && c > j) A[lane] -= A[r * D + j] * A[c * D + j];
        __builtin_amdgcn_wave_barrier();
    }
    // L^-1: lane c < D runs the forward substitution of column c
    if (lane < D) {
        const int cc = lane;
        double col[D];
#pragma unroll
        for (int rr = 0; rr < D; ++rr) {
            double v = (rr == cc) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < rr; ++k) v -= (k >= cc) ? A[rr * D + k] * col[k] : 0.0;
            col[rr] = (rr >= cc) ? v / A[rr * D + rr] : 0.0;
            Li[rr * D + cc] = col[rr];
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (!ok && lane == 0) atomicAdd(&status[ST_DIAG_FAIL], 1);
    if (act) Linv[(size_t)i * DD + lane] = Li[lane];
    if (g && lane < D) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) v += Li[lane * D + k] * g[(size_t)i * D + k];
        const size_t o = (size_t)i * D + lane;
        r0[o] = v; w0[o] = 0.0; s0[o] = 0.0; p0[o] = 0.0; x0[o] = 0.0;
        sR[wv][lane] = v;
    }
    if (!Bmat) return;
    if (act) {
        typedef PoseOps<D> G;
        double v = (r == c) ? 1.0 : 0.0;
        if (basis == 1) {
            const double* Tp = poses + G::W * (size_t)pose_of_rid[i];      // (c differs per lane: entries straight from memory)
            v = 0.0;
#pragma unroll
            for (int m2 = 0; m2 < D; ++m2) v += (m2 >= r) ? A[m2 * D + r] * G::adj_mem(Tp, m2, c) : 0.0;     // (L^T Ad)[r][c]
        }
        Bmat[(size_t)i * DD + lane] = v;
        sB[wv][lane] = v;
    }
    __builtin_amdgcn_wave_barrier();
    if (g && lane < D) {
        double v = 0.0;
#pragma unroll
        for (int a = 0; a < D; ++a) v += sB[wv][a * D + lane] * sR[wv][a];
        bg[(size_t)i * D + lane] = v;
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

// The same for big reduced systems (explicit two-level PCG): one WAVE takes PS_SCB_NB consecutive blocks and requests all
// their inputs before it touches any -- two memory round trips per PS_SCB_NB blocks instead of per block (a block per
// 64-thread workgroup is a chain of index load -> operand loads -> three LDS phases with nothing to overlap it but the
// other workgroups of the CU: 59.5 us for C4's 160 k blocks, 140 MB).  Same arithmetic, same order: bit-identical output.
// Measured 4 / 8 / 16 blocks per wave: 45.8 / 46.6 / 61.8 us (3 TB/s, two thirds of it writes).
#define PS_SCB_NB 4
template <int D>
__global__ __launch_bounds__(256) void k_scale_blocks_p(
    int nnzb, const int32_t* __restrict__ col_idx, const int32_t* __restrict__ brow_of,
    const double* __restrict__ Linv, const double* __restrict__ S, const int32_t* __restrict__ out_slot,
    double* __restrict__ Sout, const double* __restrict__ Bmat, double* __restrict__ SB)
{
    constexpr int DD = D * D;
    __shared__ double lds[4][5][36];                       // per wave: S, T, Li, Lj, B
    const int wv = threadIdx.x >> 6, t = threadIdx.x & 63;
    const int b0 = (blockIdx.x * 4 + wv) * PS_SCB_NB;
    if (b0 >= nnzb) return;                                // (whole wave)
    double *sS = lds[wv][0], *sT = lds[wv][1], *sLi = lds[wv][2], *sLj = lds[wv][3], *sB = lds[wv][4];
    int bi[PS_SCB_NB], bj[PS_SCB_NB], bs[PS_SCB_NB];
#pragma unroll
    for (int q = 0; q < PS_SCB_NB; ++q) {
        const int b = min(b0 + q, nnzb - 1);
        bi[q] = brow_of[b]; bj[q] = col_idx[b]; bs[q] = out_slot[b];
    }
    double vs[PS_SCB_NB], vi[PS_SCB_NB], vj[PS_SCB_NB], vb[PS_SCB_NB];
    const int tt = t < DD ? t : 0;
#pragma unroll
    for (int q = 0; q < PS_SCB_NB; ++q) {
        const int b = min(b0 + q, nnzb - 1);
        vs[q] = S[(size_t)b * DD + tt];
        vi[q] = Linv[(size_t)bi[q] * DD + tt];
        vj[q] = Linv[(size_t)bj[q] * DD + tt];
        vb[q] = Bmat ? Bmat[(size_t)bj[q] * DD + tt] : 0.0;
    }
    const int r = tt / D, c = tt % D;
#pragma unroll
    for (int q = 0; q < PS_SCB_NB; ++q) {
        if (b0 + q >= nnzb) break;                         // (wave-uniform)
        if (t < DD) { sS[t] = vs[q]; sLi[t] = vi[q]; sLj[t] = vj[q]; sB[t] = vb[q]; }
        __builtin_amdgcn_wave_barrier();
        double v = 0.0;
#pragma unroll
        for (int a = 0; a < D; ++a) v += sLi[r * D + a] * sS[a * D + c];
        if (t < DD) sT[t] = v;
        __builtin_amdgcn_wave_barrier();
        v = 0.0;
#pragma unroll
        for (int a = 0; a < D; ++a) v += sT[r * D + a] * sLj[c * D + a];
        __builtin_amdgcn_wave_barrier();                   // (everyone has read sS)
        if (t < DD) { Sout[(size_t)bs[q] * DD + t] = v; sS[t] = v; }
        if (Bmat) {
            __builtin_amdgcn_wave_barrier();
            double u = 0.0;
#pragma unroll
            for (int a = 0; a < D; ++a) u += sS[r * D + a] * sB[a * D + c];
            if (t < DD) SB[(size_t)bs[q] * DD + t] = u;
        }
        __builtin_amdgcn_wave_barrier();
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
        if (!(gamma > thresh)) {                     // converged (gamma == 0 too); NaN (a failed block upstream) = breakdown:
            //                                              the gated tail must not apply it
            if (first) { status[ST_PCG_DONE] = (gamma != gamma) ? 2 : 1; scalars[SC_RRFINAL] = gamma; if (k == 0) scalars[SC_RR0] = gamma; }
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
        if (!(gamma > thresh)) {                     // converged (gamma == 0 too); NaN (a failed block upstream) = breakdown:
            //                                              the gated tail must not apply it
            if (first) { status[ST_PCG_DONE] = (gamma != gamma) ? 2 : 1; scalars[SC_RRFINAL] = gamma; if (k == 0) scalars[SC_RR0] = gamma; }
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

// up to four vectors zeroed in ONE launch (four hipMemsetAsync are four fill launches of ~6 us each: the explicit PCG's set-up
// paid 23 us per solve for 100 KB of zeros)
__global__ __launch_bounds__(256) void k_zero4(size_t n0, double* __restrict__ p0, size_t n1, double* __restrict__ p1,
                                               size_t n2, double* __restrict__ p2, size_t n3, double* __restrict__ p3)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x, n = n0 + n1 + n2 + n3;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (i < n0) p0[i] = 0.0;
        else if (i < n0 + n1) p1[i - n0] = 0.0;
        else if (i < n0 + n1 + n2) p2[i - n0 - n1] = 0.0;
        else p3[i - n0 - n1 - n2] = 0.0;
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

// The same solve in ONE launch (whole-iteration calls): S straight from its blocks into LDS, the blocked Cholesky of
// k_coarse_chol WITHOUT forming L^-1, then block forward / backward substitution on g -- the inverse was 2/5 of the
// factor kernel's time and the two extra launches another ~17 us (reduced-solve stage at 30 / 54 / 90 unknowns:
// 29 / 51 / 99 -> 17 / 28 / 50 us; the panel is written by the thread that read it: three barriers per block step, not four).
// LDS: n^2 (L) + 2 n (g -> y -> x, staging) + ncb D^2 (inverse diagonal blocks of L) doubles.
template <int D>
__global__ __launch_bounds__(1024) void k_direct_solve(
    int nr, int nnzb, const int32_t* __restrict__ brow_of, const int32_t* __restrict__ col_idx,
    const double* __restrict__ S, const double* __restrict__ g, double* __restrict__ x,
    int32_t* __restrict__ status, double* __restrict__ scalars)
{
    constexpr int DD = D * D;
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int n = nr * D, t = threadIdx.x, nt = blockDim.x;
    double* sL = sm;                       // n x n, row-major: S, overwritten by L (lower)
    double* sv = sL + n * n;               // n: g, then y = L^-1 g (solved entries live in sw)
    double* sw = sv + n;                   // n: y (forward), then x (backward)
    double* sDi = sw + n;                  // nr x D x D: inverse of every diagonal block of L
    for (int k = t; k < n * n; k += nt) sL[k] = 0.0;
    if (t < n) sv[t] = g[t];
    __syncthreads();
    for (int k = t; k < nnzb * DD; k += nt) {
        const int b = k / DD, e = k % DD;
        sL[(brow_of[b] * D + e / D) * n + col_idx[b] * D + e % D] = S[k];
    }
    for (int J = 0; J < nr; ++J) {
        __syncthreads();
        if (t == 0) {                      // D x D Cholesky of the diagonal block + its inverse (reciprocal roots: no division)
            double L[D][D], Mi[D][D], il[D];
            bool ok = true;
#pragma unroll
            for (int a = 0; a < D; ++a)
#pragma unroll
                for (int b2 = 0; b2 < D; ++b2) { L[a][b2] = 0.0; Mi[a][b2] = 0.0; }
#pragma unroll
            for (int j = 0; j < D; ++j) {
                double d = sL[(J * D + j) * n + J * D + j];
#pragma unroll
                for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
                ok = ok && (d > 0.0);
                il[j] = ps_rsqrt(d);
                L[j][j] = d * il[j];
#pragma unroll
                for (int i = j + 1; i < D; ++i) {
                    double v = sL[(J * D + i) * n + J * D + j];
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
                    sL[(J * D + a) * n + J * D + b2] = L[a][b2];
                    sDi[J * DD + a * D + b2] = Mi[a][b2];
                }
        }
        __syncthreads();
        const int m = nr - J - 1;          // panel: L_IJ = A_IJ L_JJ^-T, one ROW of the block column per thread (read and
        for (int idx = t; idx < m * D; idx += nt) {          // written by the same thread: no barrier in between)
            double* row = sL + ((J + 1) * D + idx) * n + J * D;
            double ar[D], lr[D];
#pragma unroll
            for (int k = 0; k < D; ++k) ar[k] = row[k];
#pragma unroll
            for (int b2 = 0; b2 < D; ++b2) {
                double v = 0.0;
#pragma unroll
                for (int k = 0; k <= b2; ++k) v += ar[k] * sDi[J * DD + b2 * D + k];
                lr[b2] = v;
            }
#pragma unroll
            for (int k = 0; k < D; ++k) row[k] = lr[k];
        }
        __syncthreads();
        for (int idx = t; idx < m * m * DD; idx += nt) {     // trailing update A_IK -= L_IJ L_KJ^T for J < K <= I
            const int blk = idx / DD, e = idx % DD, a = e / D, b2 = e % D;
            const int I = J + 1 + blk / m, K = J + 1 + blk % m;
            if (K > I) continue;
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) v += sL[(I * D + a) * n + J * D + k] * sL[(K * D + b2) * n + J * D + k];
            sL[(I * D + a) * n + K * D + b2] -= v;
        }
    }
    // forward: y_J = L_JJ^-1 g_J, then g_I -= L_IJ y_J for the rows below
    for (int J = 0; J < nr; ++J) {
        __syncthreads();
        if (t < D) {
            double v = 0.0;
            for (int k = 0; k <= t; ++k) v += sDi[J * DD + t * D + k] * sv[J * D + k];
            sw[J * D + t] = v;
        }
        __syncthreads();
        const int i = (J + 1) * D + t;
        if (i < n) {
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) v += sL[i * n + J * D + k] * sw[J * D + k];
            sv[i] -= v;
        }
    }
    __syncthreads();
    if (t < n) sv[t] = sw[t];              // y complete
    // backward: x_J = L_JJ^-T y_J, then y_I -= L_JI^T x_J for the rows above
    for (int J = nr - 1; J >= 0; --J) {
        __syncthreads();
        if (t < D) {
            double v = 0.0;
            for (int k = t; k < D; ++k) v += sDi[J * DD + k * D + t] * sv[J * D + k];
            sw[J * D + t] = v;
        }
        __syncthreads();
        if (t < J * D) {
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) v += sL[(J * D + k) * n + t] * sw[J * D + k];
            sv[t] -= v;
        }
    }
    __syncthreads();
    if (t < n) x[t] = sw[t];
    if (t == 0) {
        status[ST_PCG_DONE] = 1; status[ST_PCG_ITERS] = 0;
        scalars[SC_RR0] = 1.0; scalars[SC_RRFINAL] = 0.0;
    }
}

// ---- restart after a breakdown of the pipelined recurrences (rare; host-driven, see cg_fused_run) ----------------
// xacc (+)= x, then g = gsaved - S xacc: the next pass solves for the correction with the same matrix and preconditioner
__global__ __launch_bounds__(256) void k_vec_accumulate(int n, const double* __restrict__ x, double* __restrict__ xacc, int first)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) xacc[t] = first ? x[t] : xacc[t] + x[t];
}

template <int D>
__global__ __launch_bounds__(256) void k_bsr_residual(
    int nr, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col_idx, const double* __restrict__ S,
    const double* __restrict__ x, const double* __restrict__ gsaved, double* __restrict__ g)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nr * D) return;
    const int i = t / D, r = t % D;
    double v = gsaved[t];
    for (int b = row_ptr[i]; b < row_ptr[i + 1]; ++b) {
        const double* sb = S + (size_t)b * D * D + r * D;
        const double* xv = x + (size_t)col_idx[b] * D;
#pragma unroll
        for (int c = 0; c < D; ++c) v -= sb[c] * xv[c];
    }
    g[t] = v;
}
