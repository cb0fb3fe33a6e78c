// ps_k_band.h -- banded coarse matrix of the explicit two-level PCG: factorisation and inverse.
// Part of ps_kernels.h (included from there; not a stand-alone header).
#pragma once

// ---------------------------------------------------------------------------
// On chain-like problems (sliding-window BA, pose graphs with short loop closures) the coarse matrix A_c = P^T S^ P of
// the explicit two-level PCG is block-banded: a hat function overlaps its neighbours only, and S^ couples a pose to a
// window of poses -- C4: 4 block off-diagonals of 101, C2: 3 of 256.  The general path factors A_c as a dense matrix
// (25 / 64 serial panel launches + triangular inverse + merges: 1.6 ms at C4, ~4 ms at C2, on the side stream, where
// it slowed the CG beside it and made the next set-up wait).  With at most PS_BAND_MAXB block off-diagonals:
//   k_band_chol      ONE workgroup walks the block columns with the active (B + 1) x (B + 1) block window in LDS:
//                    6 x 6 diagonal factor, panel L_ik = A_ik L_kk^-T, trailing update; finished block rows leave as
//                    the scalar band of L by rows (Lrow) and by columns (Lcol), diagonal as reciprocals
//   k_band_inverse   one WAVE per column c of A_c^-1 = L^-T L^-1: forward then backward band substitution, the last
//                    47 unknowns in a lane-shifted register ring (lane j = j steps back), the band row one coalesced
//                    load, the dot product a wave sum; stores (i, c) and (c, i) with the same fp32 value
// C4: 0.15 + 0.08 ms, C2: 0.4 + 0.2 ms.
// ---------------------------------------------------------------------------
#define PS_BAND_MAXB 7                        // block off-diagonals (nodes): scalar half-bandwidth <= 8 D - 1 = 47
#define PS_BAND_W 48                          // row pitch of the band arrays
#define PS_BAND_NB (PS_BAND_MAXB + 1)

// Batched form (round 5, ps_k_bandpart.h): `part` != NULL -- workgroup b factors the diagonal sub-matrix of the nodes
// part[b].x .. part[b].x + part[b].y - 1 (the interior of one chunk of the partitioned factorisation); the band arrays are
// indexed by global row, so the chunks' factors sit side by side in the same Lrow / Lcol / rdiag.  `lda`: row pitch of Ac.
template <int D>
__global__ __launch_bounds__(256) void k_band_chol(
    int ncb, int B, const double* __restrict__ Ac, double* __restrict__ Lrow, double* __restrict__ Lcol,
    double* __restrict__ rdiag, int32_t* __restrict__ status, int lda, const int2* __restrict__ part)
{
    constexpr int DD = D * D, NB = PS_BAND_NB, W = PS_BAND_W;
    __shared__ double Wn[NB][NB][DD];         // block (i, c) of the window at [i % NB][c % NB]
    __shared__ double P[PS_BAND_MAXB * D][D];
    __shared__ double Li2[2][DD];
    __shared__ int bad;
    if (part) {
        const int2 pr = part[blockIdx.x];
        ncb = pr.y;
        Ac += (size_t)pr.x * D * lda + (size_t)pr.x * D;
        Lrow += (size_t)pr.x * D * W; Lcol += (size_t)pr.x * D * W; rdiag += (size_t)pr.x * D;
    }
    const int nc = lda, t = threadIdx.x;
    if (t == 0) bad = 0;
    // block row i of the band (block columns i - B .. i) <-> two values per thread: requested from global memory at the
    // start of a step, stored into the window at its end (the slot is in use until then)
    constexpr int NPRE = (PS_BAND_NB * DD + 255) / 256;
    auto fetch_row = [&](int i, double* v) {
#pragma unroll
        for (int u = 0; u < NPRE; ++u) {
            const int e = t + u * 256;
            const int cb = i - B + e / DD, rr = (e % DD) / D, cc = e % D;
            v[u] = (i < ncb && e < (B + 1) * DD && cb >= 0) ? Ac[(size_t)(i * D + rr) * nc + cb * D + cc] : 0.0;
        }
    };
    auto store_row = [&](int i, const double* v) {
        if (i >= ncb) return;
#pragma unroll
        for (int u = 0; u < NPRE; ++u) {
            const int e = t + u * 256;
            const int cb = i - B + e / DD;
            if (e < (B + 1) * DD && cb >= 0) Wn[i % NB][cb % NB][e % DD] = v[u];
        }
    };
    // L_kk (lower, in place) and its inverse into li, from the block's 36 values in LDS; one thread, in registers
    auto factor_diag = [&](double* Akk, double* li_out) {
        double a[DD], li[DD], rl[D];
#pragma unroll
        for (int e = 0; e < DD; ++e) a[e] = Akk[e];
        bool ok = true;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            double d = a[j * D + j];
#pragma unroll
            for (int m = 0; m < j; ++m) d -= a[j * D + m] * a[j * D + m];
            if (!(d > 0.0)) { ok = false; d = 1.0; }
            rl[j] = rsqrt(d);
            a[j * D + j] = d * rl[j];
#pragma unroll
            for (int i = j + 1; i < D; ++i) {
                double v = a[i * D + j];
#pragma unroll
                for (int m = 0; m < j; ++m) v -= a[i * D + m] * a[j * D + m];
                a[i * D + j] = v * rl[j];
            }
        }
#pragma unroll
        for (int c = 0; c < D; ++c)                          // column c of L^-1
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double v = (i == c) ? 1.0 : 0.0;
#pragma unroll
                for (int m = c; m < i; ++m) v -= a[i * D + m] * li[m * D + c];
                li[i * D + c] = (i < c) ? 0.0 : v * rl[i];
            }
#pragma unroll
        for (int e = 0; e < DD; ++e) { Akk[e] = a[e]; li_out[e] = li[e]; }
        if (!ok) bad = 1;
    };
    for (int i = 0; i <= B; ++i) { double v[NPRE]; fetch_row(i, v); store_row(i, v); }
    __syncthreads();
    if (t == 0) factor_diag(Wn[0][0], Li2[0]);
    __syncthreads();
    // Step k, two barriers: (A) panel L_ik = A_ik L_kk^-T and the finished block row k out to global memory;
    // (B) wave 0 updates and factors the NEXT diagonal block (look-ahead: the serial part, ~0.5 us) while waves 1-3 do
    // the rest of the trailing update, put the panel into the window and bring in block row k + B + 1.
    for (int k = 0; k < ncb; ++k) {
        const int nrow = min(B, ncb - 1 - k) * D;            // scalar rows below the diagonal block
        const double* Li = Li2[k & 1];
        double nxt[NPRE];
        fetch_row(k + B + 1, nxt);
        if (t < nrow * D) {
            const int row = t / D, c = t % D, i = k + 1 + row / D;
            const double* src = Wn[i % NB][k % NB] + (row % D) * D;
            double v = 0.0;
            for (int m = 0; m <= c; ++m) v += src[m] * Li[c * D + m];
            P[row][c] = v;
        }
        for (int e = t; e < (B + 1) * DD; e += 256) {        // block row k is final: out as scalar bands
            const int cb = k - B + e / DD, rr = (e % DD) / D, cc = e % D;
            if (cb < 0) continue;
            const int row = k * D + rr, col = cb * D + cc;
            const double v = Wn[k % NB][cb % NB][rr * D + cc];
            if (col < row) { Lrow[(size_t)row * W + (row - 1 - col)] = v; Lcol[(size_t)col * W + (row - col - 1)] = v; }
            else if (col == row) rdiag[row] = 1.0 / v;
        }
        __syncthreads();
        if (t < 64) {
            if (nrow > 0) {
                double* An = Wn[(k + 1) % NB][(k + 1) % NB];
                if (t < DD) {
                    const int r1 = t / D, r2 = t % D;
                    double v = 0.0;
#pragma unroll
                    for (int m = 0; m < D; ++m) v += P[r1][m] * P[r2][m];
                    An[t] -= v;
                }
                __builtin_amdgcn_wave_barrier();
                if (t == 0) factor_diag(An, Li2[(k + 1) & 1]);
            }
        } else {
            for (int e = t - 64; e < nrow * nrow; e += 192) {    // trailing update A_ij -= L_ik L_jk^T (block lower triangle)
                const int r1 = e / nrow, r2 = e % nrow;
                const int i = k + 1 + r1 / D, j = k + 1 + r2 / D;
                if (j <= i && !(i == k + 1 && j == k + 1)) {
                    double v = 0.0;
#pragma unroll
                    for (int m = 0; m < D; ++m) v += P[r1][m] * P[r2][m];
                    Wn[i % NB][j % NB][(r1 % D) * D + r2 % D] -= v;
                }
            }
            for (int e = t - 64; e < nrow * D; e += 192) {
                const int row = e / D, c = e % D, i = k + 1 + row / D;
                Wn[i % NB][k % NB][(row % D) * D + c] = P[row][c];
            }
        }
        store_row(k + B + 1, nxt);                           // (its slot is free, or the one block row k left in phase A)
        __syncthreads();
    }
    if (t == 0 && bad) atomicAdd(&status[ST_DIAG_FAIL], 1);
}

// lane 0 <- x, lane j <- lane j - 1 (DPP wave_shr:1)
PS_DEV double band_ring_push(double ring, double x) {
    const unsigned long long r = __builtin_bit_cast(unsigned long long, ring), o = __builtin_bit_cast(unsigned long long, x);
    const int lo = __builtin_amdgcn_update_dpp((int)(unsigned)o, (int)(unsigned)r, 0x138, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(unsigned)(o >> 32), (int)(unsigned)(r >> 32), 0x138, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

__global__ __launch_bounds__(256) void k_band_inverse(
    int nc, const double* __restrict__ Lrow, const double* __restrict__ Lcol, const double* __restrict__ rdiag,
    double* __restrict__ Xs /* nc x nc scratch */, float* __restrict__ Ainv)
{
    constexpr int W = PS_BAND_W, U = 8;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= nc) return;
    const bool in = lane < W;
    double* xs = Xs + (size_t)c * nc;
    // the band rows of the NEXT eight steps are requested before the dependent chain of the current eight runs
    double a[U], rd[U], an[U], rdn[U], xv[U], xn[U];
    auto fwd_fetch = [&](int i0, double* av, double* rv) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = min(i0 + u, nc - 1);
            av[u] = in ? Lrow[(size_t)i * W + lane] : 0.0;
            rv[u] = rdiag[i];
        }
    };
    double ring = 0.0;
    fwd_fetch(c, a, rd);
    for (int i0 = c; i0 < nc; i0 += U) {                     // L x = e_c
        if (i0 + U < nc) fwd_fetch(i0 + U, an, rdn);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u;
            if (i < nc) {
                const double s = wave_sum(a[u] * ring);
                const double x = ((i == c ? 1.0 : 0.0) - s) * rd[u];
                if (lane == 0) xs[i] = x;
                ring = band_ring_push(ring, x);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { a[u] = an[u]; rd[u] = rdn[u]; }
    }
    // lane 0's stores of x are read back by every lane of the SAME wave below: ordering within the compute unit is all that
    // is needed (its L1 is coherent for its own waves).  This was __threadfence() until round 3: an agent-scope fence is an
    // L2 write-back + invalidate on this eight-L2 part, one per column wave (2 406 at C2), beside the CG's kernels.
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    auto bwd_fetch = [&](int i0, double* av, double* rv, double* xo) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = max(i0 - u, c);
            av[u] = in ? Lcol[(size_t)i * W + lane] : 0.0;
            rv[u] = rdiag[i];
            xo[u] = xs[i];
        }
    };
    ring = 0.0;
    bwd_fetch(nc - 1, a, rd, xv);
    for (int i0 = nc - 1; i0 >= c; i0 -= U) {                // L^T y = x, rows nc - 1 .. c
        if (i0 - U >= c) bwd_fetch(i0 - U, an, rdn, xn);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 - u;
            if (i >= c) {
                const double s = wave_sum(a[u] * ring);
                const double y = (xv[u] - s) * rd[u];
                if (lane == 0) { const float f = (float)y; Ainv[(size_t)i * nc + c] = f; Ainv[(size_t)c * nc + i] = f; }
                ring = band_ring_push(ring, y);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { a[u] = an[u]; rd[u] = rdn[u]; xv[u] = xn[u]; }
    }
}


// lane j <- lane j + 1, lane 63 <- 0 (DPP wave_shl:1)
PS_DEV double band_shift_down(double v) {
    const unsigned long long r = __builtin_bit_cast(unsigned long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)r, 0x130, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(r >> 32), 0x130, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
PS_DEV double band_lane0(double v) {
    const unsigned long long r = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)r), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(r >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

PS_DEV double band_lane(double v, int idx /* wave-uniform */) {
    const unsigned long long r = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)r, idx), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(r >> 32), idx);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// Round 4: the same columns by RIGHT-LOOKING substitution, the band streamed through LDS.
// k_band_inverse forms every unknown as a 48-lane dot product of the band row with a ring of the last unknowns, the rows
// requested eight steps ahead: 0.19 us per step of a chain of 2 (nc - c) dependent steps -- 0.23 ms at C4, 0.66-1.2 ms at C2,
// for a few MFLOP.  A first rewrite only replaced the dot product (wave-wide sum, six dependent DPP stages) by pending sums --
// lane j carries what the row j steps ahead is owed; a step reads lane 0's, forms the unknown, shifts the pending sums down
// one lane and adds the unknown's column of L -- and gained 25 %: the chain was waiting for its LOADS (eight steps of ~40
// cycles are far shorter than a memory latency).  So the four waves of a workgroup (four neighbouring columns: the same
// rows) now share tiles of 64 band rows, brought into LDS by all 256 threads one tile ahead; a step costs an LDS read that
// does not depend on the chain, two v_readlane, a DPP move and two FMAs.
#define PS_BI2_T 64
// PART (round 5, ps_k_bandpart.h): the batched form -- workgroup b takes four columns of the diagonal sub-matrix its table
// entry names (rows row0 .. row0 + n - 1 of the band arrays: one chunk of the partitioned factorisation, or the whole
// separator system) and writes them as DOUBLES into that sub-matrix's own dense n x n block of Gout.
struct BandInvItem { int32_t row0, n, cmin, pad; int64_t goff; };
template <bool PART>
__global__ __launch_bounds__(256) void k_band_inverse_rl(
    int nc, const double* __restrict__ Lrow, const double* __restrict__ Lcol, const double* __restrict__ rdiag,
    double* __restrict__ Xs /* nc x nc scratch */, float* __restrict__ Ainv,
    const BandInvItem* __restrict__ items, double* __restrict__ Gout, int ldx)
{
    constexpr int W = PS_BAND_W, T = PS_BI2_T, NPT = T * W / 256;          // 12 doubles per thread and tile
    __shared__ double tile[2][T][W];
    __shared__ double rdt[2][T];
    const int t = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
    int cmin = blockIdx.x * 4;
    size_t xoff = 0;
    if (PART) {
        const BandInvItem it = items[blockIdx.x];
        nc = it.n; cmin = it.cmin;
        Lrow += (size_t)it.row0 * W; Lcol += (size_t)it.row0 * W; rdiag += it.row0;
        Gout += it.goff;
        xoff = (size_t)it.row0 * ldx;
    } else ldx = nc;
    const int c = cmin + wv;                                   // (wave-uniform: the per-step tests below are scalar branches)
    const bool live = c < nc, in = lane < W;
    double* xs = Xs + xoff + (size_t)min(c, nc - 1) * ldx;
    // tile k of a sweep: rows row0 + dir * (0 .. 63)
    auto fetch = [&](const double* __restrict__ L, int row0, int dir, double* v, double& rv) {
#pragma unroll
        for (int q = 0; q < NPT; ++q) {
            const int e = t + 256 * q, r = row0 + dir * (e / W);
            v[q] = (r >= 0 && r < nc) ? L[(size_t)r * W + (e % W)] : 0.0;
        }
        const int r = row0 + dir * t;
        rv = (t < T && r >= 0 && r < nc) ? rdiag[r] : 0.0;
    };
    auto stash = [&](int b, const double* v, double rv) {
#pragma unroll
        for (int q = 0; q < NPT; ++q) { const int e = t + 256 * q; tile[b][e / W][e % W] = v[q]; }
        if (t < T) rdt[b][t] = rv;
    };
    double v[NPT], rv;
    // ---- forward, L x = e_c, rows cmin .. nc - 1: after x_i, row i + 1 + j is owed L[i + 1 + j][i] x_i = Lcol[i][j] x_i
    double pend = 0.0;
    fetch(Lcol, cmin, 1, v, rv);
    stash(0, v, rv);
    __syncthreads();
    int b = 0;
    for (int base = cmin; base < nc; base += T, b ^= 1) {
        const bool more = base + T < nc;
        if (more) fetch(Lcol, base + T, 1, v, rv);
        if (live) {
            double xring = 0.0;                                 // the tile's unknowns, pushed in at lane 0: lane j ends up with row base + 63 - j
            const double rdreg = rdt[b][lane];                  // (T == 64: lane u holds 1 / L[u][u] of the tile's row u)
            // sixteen steps at a time: their band entries come out of LDS before the dependent chain of the steps runs (an LDS
            // read inside every step cost the chain its latency: 134 us at C4 that way)
#pragma unroll
            for (int g = 0; g < T / 16; ++g) {
                double a[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) a[k] = in ? tile[b][16 * g + k][lane] : 0.0;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int u = 16 * g + k, i = base + u;
                    double x = 0.0;
                    if (i >= c && i < nc) {
                        x = ((i == c ? 1.0 : 0.0) - band_lane0(pend)) * band_lane(rdreg, u);
                        pend = band_shift_down(pend);
                        pend += a[k] * x;
                    }
                    xring = band_ring_push(xring, x);
                }
            }
            { const int i = base + 63 - lane; if (i >= c && i < nc) xs[i] = xring; }
        }
        if (more) stash(b ^ 1, v, rv);
        __syncthreads();
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");     // (this wave's x, read back by the same wave below: as in k_band_inverse)
    // ---- backward, L^T y = x, rows nc - 1 .. cmin: after y_i, row i - 1 - j is owed L[i][i - 1 - j] y_i = Lrow[i][j] y_i
    pend = 0.0;
    fetch(Lrow, nc - 1, -1, v, rv);
    stash(0, v, rv);
    double xt = 0.0, xn = 0.0;
    if (live && nc - 1 - lane >= c) xt = xs[nc - 1 - lane];
    __syncthreads();
    b = 0;
    for (int top = nc - 1; top >= cmin; top -= T, b ^= 1) {
        const bool more = top - T >= cmin;
        if (more) {
            fetch(Lrow, top - T, -1, v, rv);
            xn = (live && top - T - lane >= c) ? xs[top - T - lane] : 0.0;
        }
        if (live) {
            double yring = 0.0;                                 // lane j ends up with row top - 63 + j
            const double rdreg = rdt[b][lane];
#pragma unroll
            for (int g = 0; g < T / 16; ++g) {
                double a[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) a[k] = in ? tile[b][16 * g + k][lane] : 0.0;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int u = 16 * g + k, i = top - u;
                    double y = 0.0;
                    if (i >= c) {
                        y = (band_lane(xt, u) - band_lane0(pend)) * band_lane(rdreg, u);
                        pend = band_shift_down(pend);
                        pend += a[k] * y;
                    }
                    yring = band_ring_push(yring, y);
                }
            }
            const int i = top - 63 + lane;
            if (i >= c && i <= top) {
                if (PART) { Gout[(size_t)i * nc + c] = yring; Gout[(size_t)c * nc + i] = yring; }
                else { const float f = (float)yring; Ainv[(size_t)i * nc + c] = f; Ainv[(size_t)c * nc + i] = f; }
            }
        }
        if (more) stash(b ^ 1, v, rv);
        xt = xn;
        __syncthreads();
    }
}
