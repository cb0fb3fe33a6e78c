// ps_k_xcg_persist4.h -- k_xcg_persist with FOUR waves per workgroup: one wave per SIMD, 512 registers per lane (round 6).
// Part of ps_kernels.h (included from there, after ps_k_xcg_persist.h; not a stand-alone header).
#pragma once

// ---------------------------------------------------------------------------
// k_xcg_persist runs eight waves per workgroup, one per row: two waves per SIMD, 256 registers per lane, and it sits on that limit
// (52 B per lane of scratch at C4).  Time stamps of its phases (PS_XP_CLOCKS, C4, an interior workgroup, 17.6 us per iteration):
//     t + coarse y 5.2 | columns 1.1 | products + publish 4.9 | first gather pass 3.6 | rest of the gather 1.2 | sums + dots 1.6
// The coarse phase -- y = A_c^-1 t for the 42 rows this workgroup's columns interpolate from, 100 KB of the fp32 inverse -- is a chain
// of L2 round trips, and the rows never change during a solve; there was no place to keep them: LDS holds the matrix, the registers
// are full.  Here the same workgroup (8 rows, the same columns, records, exchange, sums) is FOUR waves with two rows each:
//   * one wave per SIMD = 512 registers per lane (256 VGPRs + 256 AGPRs the compiler moves values through);
//   * the rows of A_c^-1 the workgroup needs are loaded ONCE, into registers: wave w keeps rows w, w + 4, ..., lane l the columns
//     2 l + 128 q (NYW x NQ float2) -- phase 2 is then NYW x NQ multiply-adds per lane and a wave reduction per row, no memory;
//   * PF = 7 blocks per lane and row in registers (+ PL = 4 in LDS): the whole 81-block row of C4 (k_xcg_persist left the eleventh
//     block of eight lanes in L2 and fetched it in every iteration).
// Same recurrences and the same order of every sum as k_xcg_persist / k_xcg_fused1 (rows of y: xcg_coarse_rows' lane order and
// wave tree; the partial pairs: four waves of values instead of four + four of zeros): the three forms give the same bits.
// Conditions (the host checks them, else k_xcg_persist): D = 6, nc even and <= 128 NQ, every workgroup's rows of y <= 4 NYW,
// records <= PS_X4_NR x 256, workgroups <= 256.
// ---------------------------------------------------------------------------
#define PS_X4_NT 256
#define PS_X4_NW 4
#define PS_X4_RPW (PS_XF_ROWS / PS_X4_NW)      // rows per wave
#define PS_X4_NR 16                            // record slots per thread: ncb * rmax * D <= PS_X4_NR * 256

template <int D, int PF, int PL, int NE /* coarse entries per thread: nc <= NE * 256 */, int NYW, int NQ>
__global__ __launch_bounds__(PS_X4_NT) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_xcg_persist4(
    int nr, const int32_t* __restrict__ row_ptr, int wf, const double* __restrict__ S, XcgFusedArgs a,
    const int32_t* __restrict__ rec_cnt /* live records per coarse node */, int nlaunch, double tol2,
    double* __restrict__ hist, int cap, int32_t* __restrict__ status, double* __restrict__ scalars, double* __restrict__ xstate,
    ps_u64* __restrict__ exch /* 2 x E doubles as two granules each; E = nr D + 2 nwg + ncb rmax D */, unsigned salt, unsigned spin_limit,
    long long* __restrict__ dbg /* measurement build (PS_XP_CLOCKS): time stamps, as k_xcg_persist */)
{
    constexpr int NT = PS_X4_NT, NW = PS_X4_NW, RPW = PS_X4_RPW, DD = D * D;
    constexpr int NCOL = (PS_XF_CAP * D + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) double tl[];   // nc: t_{k+1}; behind it nrec: the gathered records; the PL blocks
    __shared__ double su[PS_XF_CAP * D];
    __shared__ double sr[PS_XF_ROWS * D], suo[PS_XF_ROWS * D];
    __shared__ double yl[PS_XF_NODES * D];
    __shared__ double lds[32];
    __shared__ double wred[PS_XF_ROWS][2];
    __shared__ double cw[PS_XF_ROWS][PS_XCG_NSLOT][D];
    __shared__ int bad;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, wg = blockIdx.x;
    const int nc = a.nc, nwg = a.nwg;
    const size_t offG = (size_t)nr * D, offT = offG + 2 * (size_t)nwg, E = offT + (size_t)a.ncb * a.rmax * D;
    const bool chief = wg == 0 && tid == 0;
    const int nrec = a.ncb * a.rmax * D;
    double* trec = tl + ((nc + 1) & ~1);
    double* sml = trec + ((nrec + 1) & ~1);                  // PL blocks per lane and row of the matrix: [((h PL + i) D + c) NT + tid]
    int32_t* sll = reinterpret_cast<int32_t*>(sml + (size_t)PL * RPW * D * NT);     // their LDS slots: [(h PL + i) NT + tid]
    if (tid == 0) bad = 0;
    if (status[ST_PCG_DONE]) return;
    // ---- once: the workgroup's state
    const int row0 = wg * PS_XF_ROWS;
    const int kk = lane >> 3, r = lane & 7;
    const int c0 = a.cptr[wg], ncols = a.cptr[wg + 1] - c0;
    const int n_lo = a.nlo[wg], nrows_y = (a.nhi[wg] - n_lo + 1) * D;
    double rj[NCOL], wj[NCOL], sj[NCOL];
    int jj[NCOL];
#pragma unroll
    for (int q = 0; q < NCOL; ++q) {
        const int e = tid + q * NT, c = e / D, m = e - c * D;
        jj[q] = -1; rj[q] = wj[q] = sj[q] = 0.0;
        if (c < ncols) {
            const int j = a.cols[c0 + c];
            jj[q] = j;
            const size_t o = (size_t)j * D + m;
            rj[q] = a.r_in[o]; wj[q] = a.w_in[o]; sj[q] = a.s_in[o];
        }
    }
    double uo = 0.0, po = 0.0, xo = 0.0;
    const bool own_item = tid < PS_XF_ROWS * D && row0 + tid / D < nr;
    if (own_item) { const size_t o = (size_t)row0 * D + tid; uo = a.u[o]; po = a.p[o]; xo = a.x[o]; }
    double to[NE], tso[NE], sq[NE];
    int en[NE];                                     // live records of this thread's coarse entries
#pragma unroll
    for (int u = 0; u < NE; ++u) {
        const int e = tid + u * NT;
        to[u] = tso[u] = sq[u] = 0.0; en[u] = 0;
        if (e < nc) { to[u] = a.t_in[e]; tso[u] = a.ts_in[e]; en[u] = rec_cnt[e / D]; }
    }
    asm volatile("" ::: "memory");
    unsigned live = 0;                                       // which of this thread's record slots some workgroup writes
#pragma unroll
    for (int u = 0; u < PS_X4_NR; ++u) {
        const int f = tid + u * NT;
        if (f < nrec && (f / D) % a.rmax < rec_cnt[f / (a.rmax * D)]) live |= 1u << u;
    }
    // the matrix: rows wv and wv + 4 of the workgroup, PF blocks per lane and row in registers, PL in LDS
    double sb[RPW][PF > 0 ? PF : 1][D];
    int sl[RPW][PF > 0 ? PF : 1];
    int rbeg[RPW], rend[RPW];
    double bl[RPW], rw0[RPW], rw1[RPW];
    int prow[RPW];
#pragma unroll
    for (int h = 0; h < RPW; ++h) {
        const int row = row0 + wv + NW * h;
        rbeg[h] = row < nr ? (wf > 0 ? row * wf : row_ptr[row]) : 0;
        rend[h] = row < nr ? (wf > 0 ? rbeg[h] + wf : row_ptr[row + 1]) : 0;
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int b = rbeg[h] + kk + 8 * i;
            sl[h][i] = 0;
#pragma unroll
            for (int c = 0; c < D; ++c) sb[h][i][c] = 0.0;
            if (r < D && b < rend[h]) {
                sl[h][i] = (int)a.lidx[b] * D;
                const double* sp = S + (size_t)b * DD + r * D;
#pragma unroll
                for (int c = 0; c < D; ++c) sb[h][i][c] = sp[c];
            }
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int i = 0; i < PL; ++i) {
            const int b = rbeg[h] + kk + 8 * (PF + i);
            int slot = 0;
            double v6[D];
#pragma unroll
            for (int c = 0; c < D; ++c) v6[c] = 0.0;
            if (r < D && b < rend[h]) {
                slot = (int)a.lidx[b] * D;
                const double* sp = S + (size_t)b * DD + r * D;
#pragma unroll
                for (int c = 0; c < D; ++c) v6[c] = sp[c];
            }
            sll[(h * PL + i) * NT + tid] = slot;
#pragma unroll
            for (int c = 0; c < D; ++c) sml[(size_t)((h * PL + i) * D + c) * NT + tid] = v6[c];
            asm volatile("" ::: "memory");                   // (one block's temporaries at a time: a value spilled for the prologue's peak is reloaded in every iteration)
        }
        // (constants of phase 4)
        bl[h] = 0.0; rw0[h] = rw1[h] = 0.0; prow[h] = 0;
        if (row < nr) {
            if (r < D && kk < D) bl[h] = a.Bmat[(size_t)row * DD + r * D + kk];
            const int urow = __builtin_amdgcn_readfirstlane(row);
            prow[h] = a.pnode[urow]; rw0[h] = a.pw0[urow]; rw1[h] = a.pw1[urow];
        }
    }
    const int pfirst = row0 < nr ? a.pnode[row0] : 0;
    int rout = -1;
    if (tid < PS_XCG_NSLOT * D) rout = a.rec_out[wg * PS_XCG_NSLOT + tid / D];
    // the rows of A_c^-1 this workgroup multiplies by, once: wave wv the rows wv, wv + 4, ...; lane l the columns 2 l + 128 q
    float2 ai[NYW][NQ];
#pragma unroll
    for (int y = 0; y < NYW; ++y) {
        const int rr = wv + NW * y;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int j = 2 * lane + 128 * q;
            ai[y][q] = make_float2(0.f, 0.f);
            if (rr < nrows_y && j < nc) ai[y][q] = *reinterpret_cast<const float2*>(a.Ainv + (size_t)(n_lo * D + rr) * nc + j);
        }
        asm volatile("" ::: "memory");
    }
    double gamma = 0.0, delta = 0.0, g_prev = 0.0, a_prev = 0.0, thresh = 0.0;
#ifdef PS_MEASURE
    const int dslot = wg == 0 ? 0 : (wg == nwg / 2 ? 1 : (wg == nwg - 1 ? 2 : -1));
    int dpass = 0;
#define PS_XP_CLK(i) do { if (dbg && dslot >= 0 && tid == 0 && dpass < 64) dbg[((size_t)dslot * 64 + dpass) * 8 + (i)] = (long long)wall_clock64(); } while (0)
#else
#define PS_XP_CLK(i) do { } while (0)
#endif
    for (int k = -1; k < nlaunch - 1; ++k) {
        double alpha = 0.0, beta = 0.0;
        PS_XP_CLK(6);
        if (k >= 0) {
            if (k == 0) thresh = tol2 * gamma;
            if (!(gamma > thresh)) {
                if (chief) { status[ST_PCG_DONE] = (gamma != gamma) ? 2 : 1; scalars[SC_RRFINAL] = gamma; if (k == 0) scalars[SC_RR0] = gamma; }
                break;
            }
            beta = (k == 0) ? 0.0 : gamma / g_prev;
            const double denom = (k == 0) ? delta : delta - beta * gamma / a_prev;
            if (!(denom > 0.0)) { if (chief) { status[ST_PCG_DONE] = 2; scalars[SC_RRFINAL] = gamma; } break; }
            alpha = gamma / denom;
            if (chief) {
                hist[k] = gamma; hist[cap + k] = alpha; status[ST_PCG_ITERS] = k + 1; scalars[SC_RRFINAL] = gamma;
                if (k == 0) { xstate[1] = thresh; xstate[2] = gamma; scalars[SC_RR0] = gamma; }
            }
            g_prev = gamma; a_prev = alpha;
        }
        // ---- 1. t_{k+1} (all of it) into LDS
#pragma unroll
        for (int u = 0; u < NE; ++u) {
            const int e = tid + u * NT;
            if (e < nc) {
                const double ts = sq[u] + beta * tso[u];
                const double tn = to[u] - alpha * ts;
                tl[e] = tn; to[u] = tn; tso[u] = ts;
            }
        }
        __syncthreads();
        // ---- 2. y = A_c^-1 t_{k+1} for the nodes n_lo .. n_hi, from registers (the sums of xcg_coarse_rows, in its order)
#pragma unroll
        for (int y = 0; y < NYW; ++y) {
            const int rr = wv + NW * y;
            if (rr < nrows_y) {                              // (wave-uniform)
                double v = 0.0;
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int j = 2 * lane + 128 * q;
                    float fx = ai[y][q].x, fy = ai[y][q].y;
                    asm volatile("" : "+v"(fx), "+v"(fy));        // (keeps the widening in the loop: hoisted, the rows would take twice the registers)
                    const int jc = j < nc ? j : 0;                 // (no branch: a lane past the row's end reads t[0] and keeps its sum)
                    const double add = (double)fx * tl[jc] + (double)fy * tl[jc + 1];
                    v = j < nc ? v + add : v;
                }
                v = wave_sum(v);
                if (lane == 0) yl[rr] = v;
            }
        }
        __syncthreads();
        PS_XP_CLK(0);
        // ---- 3. the workgroup's columns: s, r, u; the owner's p, x
#pragma unroll
        for (int q = 0; q < NCOL; ++q) {
            int j = jj[q];
            asm volatile("" : "+v"(j));                      // (the column's constants are read again in every iteration: hoisted, 17 registers per item)
            if (j >= 0) {
                const int e = tid + q * NT, c = e / D, m = e - c * D;
                const double sn = wj[q] + beta * sj[q];
                const double rn = rj[q] - alpha * sn;
                const int njq = a.pnode[j], n0 = njq - n_lo;
                const double c0w = a.pw0[j], c1w = a.pw1[j];
                const double* B = a.Bmat + (size_t)j * DD + m * D;
                const bool two = njq + 1 < a.ncb;
                double un = rn;
#pragma unroll
                for (int mm = 0; mm < D; ++mm) {
                    const double yy = c0w * yl[n0 * D + mm] + (two ? c1w * yl[(n0 + 1) * D + mm] : 0.0);
                    un += B[mm] * yy;
                }
                su[c * D + m] = un;
                rj[q] = rn; sj[q] = sn;
                if (j >= row0 && j < row0 + PS_XF_ROWS) { sr[(j - row0) * D + m] = rn; suo[(j - row0) * D + m] = un; }
            }
        }
        if (own_item) {                                      // p_k = u_k + beta p_{k-1}, x_{k+1} = x_k + alpha p_k
            const double pn = uo + beta * po;
            po = pn; xo += alpha * pn;
        }
        __syncthreads();
        PS_XP_CLK(1);
        if (own_item) uo = suo[tid];                         // u_{k+1} of the own rows, for the next iteration's p
        // ---- 4. w_{k+1} = S^ u_{k+1} for the own rows, partials, records of P^T w: published
        const unsigned tag = salt * 4096u + (unsigned)(k + 2);
        ps_u64* buf = exch + (size_t)(k & 1) * E * 2;
#pragma unroll
        for (int h = 0; h < RPW; ++h) {
            asm volatile("" ::: "memory");                   // (one row's LDS reads at a time: the scheduler would hoist both rows')
            const int rw = wv + NW * h, row = row0 + rw;
            if (lane < PS_XCG_NSLOT * D) (&cw[rw][0][0])[lane] = 0.0;
            double g2 = 0.0, d2 = 0.0;
            if (row < nr) {
                double acc = 0.0;
                if (r < D) {
#pragma unroll
                    for (int i = 0; i < PF; ++i) {
                        const double* uc = su + sl[h][i];
#pragma unroll
                        for (int c = 0; c < D; ++c) acc += sb[h][i][c] * uc[c];
                    }
#pragma unroll
                    for (int i = 0; i < PL; ++i) {
                        const double* uc = su + sll[(h * PL + i) * NT + tid];
#pragma unroll
                        for (int c = 0; c < D; ++c) acc += sml[(size_t)((h * PL + i) * D + c) * NT + tid] * uc[c];
                    }
                    for (int b = rbeg[h] + kk + 8 * (PF + PL); b < rend[h]; b += 8) {
                        const double* uc = su + (int)a.lidx[b] * D;
                        const double* sp = S + (size_t)b * DD + r * D;
#pragma unroll
                        for (int c = 0; c < D; ++c) acc += sp[c] * uc[c];
                    }
                }
                acc += __shfl_xor(acc, 8, 64);
                acc += __shfl_xor(acc, 16, 64);
                acc += __shfl_xor(acc, 32, 64);
                double ru = 0.0, wu = 0.0;
                if (lane < D) {
                    cp_put(buf + 2 * ((size_t)row * D + lane), tag, acc);
                    const double un = suo[rw * D + lane];
                    ru = sr[rw * D + lane] * un; wu = acc * un;
                }
                g2 = wave_sum(ru); d2 = wave_sum(wu);
                double v = bl[h] * acc;
                v = dpp_shift_add<0x111, 0xf, 0xf>(v);
                v = dpp_shift_add<0x112, 0xf, 0xf>(v);
                v = dpp_shift_add<0x114, 0xf, 0xa>(v);
                const int rslot = prow[h] - pfirst;
                if (r == 7 && kk < D) {
                    cw[rw][rslot][kk] = rw0[h] * v;
                    if (rslot + 1 < PS_XCG_NSLOT) cw[rw][rslot + 1][kk] = rw1[h] * v;
                }
            }
            if (lane == 0) { wred[rw][0] = g2; wred[rw][1] = d2; }
        }
        __syncthreads();
        if (tid == 0) {
            double g = 0.0, d = 0.0;
#pragma unroll
            for (int ww = 0; ww < PS_XF_ROWS; ++ww) { g += wred[ww][0]; d += wred[ww][1]; }
            cp_put(buf + 2 * (offG + wg), tag, g);
            cp_put(buf + 2 * (offG + nwg + wg), tag, d);
        }
        if (tid < PS_XCG_NSLOT * D && rout >= 0) {
            double v = 0.0;
#pragma unroll
            for (int ww = 0; ww < PS_XF_ROWS; ++ww) v += (&cw[ww][0][0])[tid];
            cp_put(buf + 2 * (offT + (size_t)rout * D + tid % D), tag, v);
        }
        PS_XP_CLK(2);
        // ---- 5. gather what the next iteration needs: w of the columns, every workgroup's partials, every live record
        double gs = 0.0, ds = 0.0;
        {
            const long long t_enter = (long long)wall_clock64();       // (bounded in wall-clock time too: PS_PERSIST_TIMEOUT_TICKS, ps_k_cg_persist.h)
            double rv[PS_X4_NR];
            bool ok = false;
            for (unsigned spins = 0; !ok; ++spins) {
                ok = true;
#pragma unroll
                for (int q = 0; q < NCOL; ++q) {
                    if (jj[q] >= 0) {
                        const int e = tid + q * NT, c = e / D, m = e - c * D;
                        (void)c;
                        ok = xp_get(buf + 2 * ((size_t)jj[q] * D + m), tag, wj[q]) && ok;
                    }
                }
                gs = 0.0; ds = 0.0;
                if (tid < nwg) {                             // (nwg <= 256 = NT: one partial pair per thread)
                    ok = xp_get(buf + 2 * (offG + tid), tag, gs) && ok;
                    ok = xp_get(buf + 2 * (offG + nwg + tid), tag, ds) && ok;
                }
                // the live records, flat, four slots at a time (as k_xcg_persist)
#pragma unroll
                for (int u0 = 0; u0 < PS_X4_NR; u0 += 4) {
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int u = u0; u < u0 + 4; ++u) {
                        rv[u] = 0.0;
                        if (live & (1u << u)) ok = xp_get(buf + 2 * (offT + (size_t)(tid + u * NT)), tag, rv[u]) && ok;
                    }
                }
                ok = __all(ok);
#ifdef PS_MEASURE
                if (spins == 0) PS_XP_CLK(5);
#endif
                if (!ok) {
                    if (spins > spin_limit || (long long)wall_clock64() - t_enter > PS_PERSIST_TIMEOUT_TICKS) { bad = 1; break; }
                    __builtin_amdgcn_s_sleep(PS_CP_SLEEP);
                }
            }
#pragma unroll
            for (int u = 0; u < PS_X4_NR; ++u) { const int f = tid + u * NT; if (f < nrec) trec[f] = rv[u]; }
        }
        __syncthreads();
        PS_XP_CLK(3);
        // the records of every coarse entry summed in record order (as k_xcg_fused1 sums them)
#pragma unroll
        for (int u = 0; u < NE; ++u) {
            const int e = tid + u * NT;
            if (e < nc) {
                const int n = e / D, m = e - n * D;
                double s2 = 0.0;
                for (int c = 0; c < en[u]; ++c) s2 += trec[((size_t)n * a.rmax + c) * D + m];
                sq[u] = s2;
            }
        }
        block_sum2(gs, ds, lds);
        PS_XP_CLK(4);
#ifdef PS_MEASURE
        ++dpass;
#endif
        if (bad) {
            if (tid == 0) { status[ST_PCG_DONE] = 2; status[ST_PERSIST_FAIL] = 1; }
            break;
        }
        gamma = gs; delta = ds;
    }
#undef PS_XP_CLK
    // ---- what the caller reads: x^ (and p) of the own rows
    if (own_item) { const size_t o = (size_t)row0 * D + tid; a.p[o] = po; a.x[o] = xo; }
}
