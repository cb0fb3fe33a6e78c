// ps_k_xcg_persist.h -- the explicit two-level PCG (one-launch-per-iteration form, k_xcg_fused1) as ONE launch per solve (round 5).
// Part of ps_kernels.h (included from there, after ps_k_xcg.h and ps_k_cg_persist.h; not a stand-alone header).
#pragma once

// ---------------------------------------------------------------------------
// k_xcg_fused1 runs one iteration of the single-reduction PCG per launch: at C4 (2 000 poses, 160 000 blocks = 46 MB) 18.7 us, most
// of it the matrix streamed from the Infinity Cache again in every launch.  Everything an iteration hands to the next already goes
// through memory in ONE step (w of the rows, the partials of gamma / delta, the records of P^T w); here that step is an in-launch
// exchange of self-tagged granules (ps_k_cg_persist.h: two 8-byte {tag | half} granules per double, write-through stores, relaxed
// agent-scope loads, two buffers by iteration parity, bounded spins) and the launch stays:
//   * a workgroup keeps its 8 rows of the matrix in registers (PF blocks per lane) and in LDS (PL more: as many as fit 160 KB
//     beside t, the records and the static arrays -- C4: 6 + 4 = the whole row; a row wider than 8 (PF + PL) blocks reads the rest
//     from L2 as before), r and s of ITS COLUMNS (it recomputes them every iteration anyway), t and ts of all coarse entries,
//     u / p / x of its own rows;
//   * per iteration it publishes w of its rows (48 sums), its two partials and its records, and gathers w of its columns (<= 170
//     x 6), all partials (2 per workgroup) and all live records (cnt[node] per node) -- one round trip.
// Same recurrences, same order of the sums as k_xcg_fused1 (launches k = -1, 0, 1, ...), same status / history / scalars.
// Needs every workgroup resident at once: the one-launch form's own condition (<= 256 workgroups) with one workgroup per compute
// unit.  A time-out (spin_limit passes, or one second) reports a breakdown + ST_PERSIST_FAIL: the host repeats the solve launch by
// launch.  Used for long (bundle-adjustment) rows only: on pose graphs there is no matrix stream worth keeping and the exchange
// between up to 256 workgroups costs more than it saves (1 500 poses: 5.48 -> 5.94 ms per solve).  C4: 355 us per launch for 20-21
// iterations against 21 launches of 18.7 us.
// ---------------------------------------------------------------------------
#define PS_XP_NR 8                      // record slots per thread: ncb * rmax * D <= PS_XP_NR * 512

PS_DEV bool xp_get(const ps_u64* g, unsigned tag, double& v) {
    const ps_u64 a = __hip_atomic_load((const ps_gu64*)g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const ps_u64 b = __hip_atomic_load((const ps_gu64*)(g + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v = __longlong_as_double((long long)((a & 0xffffffffull) | (b << 32)));
    return (unsigned)(a >> 32) == tag && (unsigned)(b >> 32) == tag;
}

template <int D, int PF, int PL, int NE /* coarse entries per thread: nc <= NE * 512 */>
__global__ __launch_bounds__(64 * PS_XF_ROWS) void k_xcg_persist(
    int nr, const int32_t* __restrict__ row_ptr, int wf, const double* __restrict__ S, XcgFusedArgs a,
    const int32_t* __restrict__ rec_cnt /* live records per coarse node */, int nlaunch, double tol2,
    double* __restrict__ hist, int cap, int32_t* __restrict__ status, double* __restrict__ scalars, double* __restrict__ xstate,
    ps_u64* __restrict__ exch /* 2 x E doubles as two granules each; E = nr D + 2 nwg + ncb rmax D */, unsigned salt, unsigned spin_limit,
    long long* __restrict__ dbg /* measurement build (PS_XP_CLOCKS): time stamps of workgroups 0, nwg / 2, nwg - 1 ([3][64 passes][8 phases]), else NULL */)
{
    constexpr int NT = 64 * PS_XF_ROWS, DD = D * D;
    constexpr int NCOL = (PS_XF_CAP * D + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) double tl[];   // nc: t_{k+1}; behind it nrec: the gathered records
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
    double* sml = trec + ((nrec + 1) & ~1);                  // PL blocks per lane of the matrix: [(i D + c) NT + tid]
    int32_t* sll = reinterpret_cast<int32_t*>(sml + (size_t)PL * D * NT);     // their LDS slots: [i NT + tid]
    // one block more for the eight lanes kk = 0 of a row (a row of 8 (PF + PL) + 1 blocks -- C4: 81 -- used to fetch it from L2 in
    // every iteration: a round trip in the middle of the products): [(wave 8 + r) D + c], its slot (-1: none) behind
    double* ovf = reinterpret_cast<double*>(sll + (size_t)PL * NT);
    int32_t* ovs = reinterpret_cast<int32_t*>(ovf + (size_t)PS_XF_ROWS * 8 * D);
    if (tid == 0) bad = 0;
    if (status[ST_PCG_DONE]) return;
    // ---- once: the workgroup's state
    const int row0 = wg * PS_XF_ROWS, row = row0 + wv;
    const int kk = lane >> 3, r = lane & 7;
    const int rbeg = row < nr ? (wf > 0 ? row * wf : row_ptr[row]) : 0;
    const int rend = row < nr ? (wf > 0 ? rbeg + wf : row_ptr[row + 1]) : 0;
    const int c0 = a.cptr[wg], ncols = a.cptr[wg + 1] - c0;
    const int n_lo = a.nlo[wg], nrows_y = (a.nhi[wg] - n_lo + 1) * D;
    // (the columns' constants -- node, hat weights, basis row -- are read again in every iteration: L2-resident, and 40 registers
    //  that the matrix needs more)
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
    unsigned live = 0;                                       // which of this thread's record slots some workgroup writes
#pragma unroll
    for (int u = 0; u < PS_XP_NR; ++u) {
        const int f = tid + u * NT;
        if (f < nrec && (f / D) % a.rmax < rec_cnt[f / (a.rmax * D)]) live |= 1u << u;
    }
    double sb[PF > 0 ? PF : 1][D];
    int sl[PF > 0 ? PF : 1];
#pragma unroll
    for (int i = 0; i < PF; ++i) {
        const int b = rbeg + kk + 8 * i;
        sl[i] = 0;
#pragma unroll
        for (int c = 0; c < D; ++c) sb[i][c] = 0.0;
        if (r < D && b < rend) {
            sl[i] = (int)a.lidx[b] * D;
            const double* sp = S + (size_t)b * DD + r * D;
#pragma unroll
            for (int c = 0; c < D; ++c) sb[i][c] = sp[c];
        }
    }
#pragma unroll
    for (int i = 0; i < PL; ++i) {
        const int b = rbeg + kk + 8 * (PF + i);
        int slot = 0;
        double v6[D];
#pragma unroll
        for (int c = 0; c < D; ++c) v6[c] = 0.0;
        if (r < D && b < rend) {
            slot = (int)a.lidx[b] * D;
            const double* sp = S + (size_t)b * DD + r * D;
#pragma unroll
            for (int c = 0; c < D; ++c) v6[c] = sp[c];
        }
        sll[i * NT + tid] = slot;
#pragma unroll
        for (int c = 0; c < D; ++c) sml[(size_t)(i * D + c) * NT + tid] = v6[c];
    }
    if (kk == 0) {
        const int b = rbeg + 8 * (PF + PL);
        int slot = -1;
        double v6[D];
#pragma unroll
        for (int c = 0; c < D; ++c) v6[c] = 0.0;
        if (r < D && b < rend) {
            slot = (int)a.lidx[b] * D;
            const double* sp = S + (size_t)b * DD + r * D;
#pragma unroll
            for (int c = 0; c < D; ++c) v6[c] = sp[c];
        }
        ovs[wv * 8 + r] = slot;
#pragma unroll
        for (int c = 0; c < D; ++c) ovf[(size_t)(wv * 8 + r) * D + c] = v6[c];
    }
    // (constants of phase 4)
    double bl = 0.0, rw0 = 0.0, rw1 = 0.0;
    int prow = 0, pfirst = 0, rout = -1;
    if (tid < PS_XCG_NSLOT * D) rout = a.rec_out[wg * PS_XCG_NSLOT + tid / D];
    if (row < nr) {
        if (r < D && kk < D) bl = a.Bmat[(size_t)row * DD + r * D + kk];
        const int urow = __builtin_amdgcn_readfirstlane(row);
        prow = a.pnode[urow]; pfirst = a.pnode[row0]; rw0 = a.pw0[urow]; rw1 = a.pw1[urow];
    }
    double gamma = 0.0, delta = 0.0, g_prev = 0.0, a_prev = 0.0, thresh = 0.0;
#ifdef PS_MEASURE
    // phase clocks (PS_XP_CLOCKS): time stamps straight to memory, [workgroup slot][pass][phase] -- no loop-carried registers in a
    // kernel that has none to spare
    const int dslot = wg == 0 ? 0 : (wg == nwg / 2 ? 1 : (wg == nwg - 1 ? 2 : -1));
    int dpass = 0;
#define PS_XP_CLK(i) do { if (dbg && dslot >= 0 && tid == 0 && dpass < 64) dbg[((size_t)dslot * 64 + dpass) * 8 + (i)] = (long long)wall_clock64(); } while (0)
#else
#define PS_XP_CLK(i) do { } while (0)
#endif
    for (int k = -1; k < nlaunch - 1; ++k) {
        double alpha = 0.0, beta = 0.0;
        PS_XP_CLK(6);                                        // (pass start)
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
        // ---- 2. y = A_c^-1 t_{k+1} for the nodes n_lo .. n_hi (as k_xcg_fused1: a wave per row)
        {
            int first = n_lo * D;
            asm volatile("" : "+s"(first));                  // (the rows' addresses formed in the loop, not carried through it)
            xcg_coarse_rows<2>(a.Ainv, nc, first, nrows_y, tl, yl, wv, lane, PS_XF_ROWS);
        }
        __syncthreads();
        PS_XP_CLK(0);
        // ---- 3. the workgroup's columns: s, r, u; the owner's p, x
#pragma unroll
        for (int q = 0; q < NCOL; ++q) {
            int j = jj[q];
            asm volatile("" : "+v"(j));                      // (keeps the loads of the column's constants IN the loop: hoisted, 17 registers per item)
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
        if (lane < PS_XCG_NSLOT * D) (&cw[wv][0][0])[lane] = 0.0;
        double g2 = 0.0, d2 = 0.0;
        if (row < nr) {
            double acc = 0.0;
            if (r < D) {
#pragma unroll
                for (int i = 0; i < PF; ++i) {
                    const double* uc = su + sl[i];
#pragma unroll
                    for (int c = 0; c < D; ++c) acc += sb[i][c] * uc[c];
                }
#pragma unroll
                for (int i = 0; i < PL; ++i) {
                    const double* uc = su + sll[i * NT + tid];
#pragma unroll
                    for (int c = 0; c < D; ++c) acc += sml[(size_t)(i * D + c) * NT + tid] * uc[c];
                }
                if (kk == 0) {
                    const int os = ovs[wv * 8 + r];
                    if (os >= 0) {
                        const double* uc = su + os;
#pragma unroll
                        for (int c = 0; c < D; ++c) acc += ovf[(size_t)(wv * 8 + r) * D + c] * uc[c];
                    }
                }
                for (int b = rbeg + kk + 8 * (PF + PL) + (kk == 0 ? 8 : 0); b < rend; b += 8) {
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
                const double un = suo[wv * D + lane];
                ru = sr[wv * D + lane] * un; wu = acc * un;
            }
            g2 = wave_sum(ru); d2 = wave_sum(wu);
            double v = bl * acc;
            v = dpp_shift_add<0x111, 0xf, 0xf>(v);
            v = dpp_shift_add<0x112, 0xf, 0xf>(v);
            v = dpp_shift_add<0x114, 0xf, 0xa>(v);
            const int rslot = prow - pfirst;
            if (r == 7 && kk < D) {
                cw[wv][rslot][kk] = rw0 * v;
                if (rslot + 1 < PS_XCG_NSLOT) cw[wv][rslot + 1][kk] = rw1 * v;
            }
        }
        if (lane == 0) { wred[wv][0] = g2; wred[wv][1] = d2; }
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
            double rv[PS_XP_NR];
            bool ok = false;
            for (unsigned spins = 0; !ok; ++spins) {
                ok = true;
                int tg = tid;
                asm volatile("" : "+v"(tg));                 // (the addresses of the pass formed here, not carried through the solve: two registers each)
#pragma unroll
                for (int q = 0; q < NCOL; ++q) {
                    if (jj[q] >= 0) {
                        const int e = tg + q * NT, c = e / D, m = e - c * D;
                        (void)c;
                        int jq = jj[q];
                        asm volatile("" : "+v"(jq));
                        ok = xp_get(buf + 2 * ((size_t)jq * D + m), tag, wj[q]) && ok;
                    }
                }
                gs = 0.0; ds = 0.0;
                if (tid < nwg) {                             // (nwg <= 256 < NT: one partial pair per thread)
                    ok = xp_get(buf + 2 * (offG + tg), tag, gs) && ok;
                    ok = xp_get(buf + 2 * (offG + nwg + tg), tag, ds) && ok;
                }
                // the live records, flat, four slots at a time (eight at once cost 30 more registers than the matrix can spare)
#pragma unroll
                for (int u0 = 0; u0 < PS_XP_NR; u0 += 4) {
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int u = u0; u < u0 + 4; ++u) {
                        rv[u] = 0.0;
                        if (live & (1u << u)) ok = xp_get(buf + 2 * (offT + (size_t)(tg + u * NT)), tag, rv[u]) && ok;
                    }
                    if ((live >> (u0 + 4)) == 0) break;
                }
                ok = __all(ok);
#ifdef PS_MEASURE
                if (spins == 0) PS_XP_CLK(5);
#endif
                if (!ok) {
                    // (bounded by TIME as well: with 250 workgroups polling, a pass that fails can take far longer than one that succeeds)
                    if (spins > spin_limit || (long long)wall_clock64() - t_enter > PS_PERSIST_TIMEOUT_TICKS) { bad = 1; break; }
                    __builtin_amdgcn_s_sleep(PS_CP_SLEEP);
                }
            }
#pragma unroll
            for (int u = 0; u < PS_XP_NR; ++u) { const int f = tid + u * NT; if (f < nrec) trec[f] = rv[u]; }
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
