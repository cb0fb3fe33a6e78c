// ps_k_cg_persist.h -- the folded two-level CG of small and medium reduced systems in ONE launch (round 5).
// Part of ps_kernels.h (included from there, after ps_k_cg_fused.h; not a stand-alone header).
#pragma once

// ---------------------------------------------------------------------------
// k_cg_fused_lds runs one CG iteration per launch: 5.0-5.4 us each at C3 (a kernel boundary, one memory round trip for the
// 5.7 MB matrix out of the Infinity Cache, one block reduction), 18-21 of them per Gauss-Newton iteration = 40 % of it.
// Nothing in an iteration needs a kernel boundary except that every workgroup needs all of w = S^ r of the others.  Here
//   * the augmented matrix stays IN REGISTERS for the whole solve: a wave owns one "task" = at most 48 blocks of one block row
//     (half a fine row at C3, a fifth of a dense coarse row; the zero padding of an ELL row is left out), lane (slot, r) holds
//     row r of 6 of them (36 doubles);
//   * the vectors r, s, p, x are REPLICATED: every workgroup keeps all n entries (four per thread) and applies the same
//     recurrences with the same alpha / beta -- bitwise the same everywhere, so nobody has to agree on anything;
//   * per iteration ONE exchange: a wave publishes the six sums of its task as self-tagged 8-byte granules {tag | half of the
//     double} with write-through stores, every workgroup gathers all of them with relaxed agent-scope loads (no flags, no
//     fences, no grid barrier: the data is the flag -- /opt/skills/guides/cdna_hip_programming.md guideline 16, form R2),
//     double-buffered by the parity of the iteration (a workgroup can only be one exchange ahead of the slowest);
//   * gamma = r.r and delta = w.r are summed by every workgroup itself, in the same order.
// Measured on the chip before it was built (tools/probes/allgather_probe.hip): 2.0-2.7 us per exchange for 32 workgroups of 512
// threads, whatever their number -- against 5.4 us per launch.  In the kernel: ~4.5 us per CG iteration at C3 (58 workgroups),
// 3.6 on a 100-pose graph; two instantiations by the number of exchanged sums per thread (6: up to 512 tasks, 12: up to 1 024).
// Every spin is bounded: a workgroup that does not see its granules within `spin_limit` passes or one second reports a breakdown
// (ST_PCG_DONE = 2, ST_PERSIST_FAIL) and leaves; the host then solves with the launch-per-iteration kernels and stops using
// this one on the handle.
// Semantics = the launches k = -1, 0, 1, ... of k_cg_fused_lds: same recurrences (Chronopoulos-Gear), same convergence test,
// same status words, history (gamma, alpha per iteration) and scalars; sums in another (fixed) order.
// ---------------------------------------------------------------------------
#define PS_CP_NT 512                    // threads per workgroup (8 waves = 8 tasks)
#define PS_CP_NQ 6                      // blocks per lane slot: a task has at most 8 * PS_CP_NQ blocks
#define PS_CP_TASKB (8 * PS_CP_NQ)
#define PS_CP_NV 4                      // vector entries per thread: n <= PS_CP_NV * PS_CP_NT
#define PS_CP_NE_MAX 12                 // exchanged sums per thread (template NE: 6 or 12): tasks * D <= NE * PS_CP_NT
#define PS_CP_MAXN (PS_CP_NV * PS_CP_NT)
// Wall-clock bound of a spin (wall_clock64: 100 MHz), beside the bound in passes: ONE SECOND.  Round 6 first took the 20 ms
// k_xcg_persist had: with two PROCESSES on one device (tests/test_gpu_sharded.py: two ranks on one GPU) the scheduler suspends a
// process's queues for whole time slices, a spin that straddles a slice sees the clock jump by more than that, and one solve in
// ~5 reported a time-out although every workgroup was resident.  The bound exists so that a grid that can never be resident
// does not hang the stream, not to police latency: residency is decided up front from the device (ps_core.hip: PersistLedger).
#define PS_PERSIST_TIMEOUT_TICKS 100000000LL
#ifndef PS_CP_SLEEP
#define PS_CP_SLEEP 1                   // s_sleep argument between two unsuccessful passes over the exchange (0: none)
#endif

struct CpTask { int32_t row, b0, b1, pad; };
// what k_coarse_recover needs: x = L^-T (x^_f + P y), y = L_c^-T x^_c -- done by the kernel's first workgroup once the solve has
// converged (x == NULL: not done here)
struct CpRecover { int nr, ncb; const int32_t* pnode; const double *pw0, *pw1, *Linv, *Lci, *Bmat; double* x; };

typedef unsigned long long ps_u64;
typedef __attribute__((address_space(1))) ps_u64 ps_gu64;

// Round 6: both granules of a double PUBLISHED by one 16-byte store -- `global_store_dwordx4 ... sc1`, the same bits and cache policy
// as the two relaxed agent-scope 8-byte stores (`global_store_dwordx2 ... sc1`) it replaces, half the write requests.  A granule still
// carries its own tag and lands by an aligned 8-byte half of the access, so a reader that sees one new half and one old one retries
// as before.  (k_xcg_persist at C4: first pass over the exchange 4.7 -> 4.0 us; k_cg_persist at C3: gather 50.4 -> 48.7 us per
// launch.  Readers keep two 8-byte loads per double: 16-byte loads through inline assembly need 128-bit landing tuples --
// k_xcg_persist: scratch 32 -> 190-260 B per lane -- and where there is room for them, k_cg_persist at C3, they measured nothing.)
typedef unsigned ps_u32x4 __attribute__((ext_vector_type(4)));
PS_DEV void cp_put(ps_u64* g, unsigned tag, double v) {
    const ps_u64 b = (ps_u64)__double_as_longlong(v);
    const ps_u32x4 q = {(unsigned)(b & 0xffffffffull), tag, (unsigned)(b >> 32), tag};
    // (`s_nop 1`: on gfx940+ a VALU write to the data registers of a store wider than 64 bits needs two wait states behind it; the
    //  compiler inserts them for its own stores and cannot see into this one -- the next cp_put reuses these very registers)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(g), "v"(q) : "memory");
}

template <int D, int NE, bool PIPE = false>
__global__ __launch_bounds__(PS_CP_NT) void k_cg_persist(
    int n /* augmented unknowns */, int ntasks, const CpTask* __restrict__ tasks,
    const int32_t* __restrict__ row_task0 /* first task of every block row; [rows] = ntasks */,
    const int32_t* __restrict__ acol_idx, const double* __restrict__ Saug,
    const double* __restrict__ r0, const double* __restrict__ w0, const double* __restrict__ s0,
    double* __restrict__ p_io, double* __restrict__ x_io,
    double* __restrict__ hist, int cap, int nlaunch /* iterations k = -1 .. nlaunch - 2 at most */, double tol2,
    int32_t* __restrict__ status, double* __restrict__ scalars,
    ps_u64* __restrict__ exch /* 2 x (ntasks * D) doubles as two granules each */, unsigned salt, unsigned spin_limit,
    long long* __restrict__ dbg /* measurement build: phase clocks of workgroup 0 (8 words), else NULL */, CpRecover rec)
{
    constexpr int DD = D * D;
    __shared__ double rn[PS_CP_MAXN];
    __shared__ double wex[NE * PS_CP_NT];              // the exchanged sums of one iteration
    __shared__ double red[2][2][PS_CP_NT / 64];
    __shared__ int bad;
    const int t = threadIdx.x, wv = t >> 6, lane = t & 63, kk = lane >> 3, r = lane & 7;
    const int task = blockIdx.x * (PS_CP_NT / 64) + wv;
    const bool chief = blockIdx.x == 0 && t == 0;
    const size_t nex = (size_t)ntasks * D;                   // exchanged doubles per iteration
    if (t == 0) bad = 0;
    // ---- the task's blocks into registers (once), the vectors' entries of this thread
    int b0 = 0, b1 = 0;
    if (task < ntasks) { const CpTask tk = tasks[task]; b0 = tk.b0; b1 = tk.b1; }
    int cj[PS_CP_NQ];
    double sv[PS_CP_NQ][D];
#pragma unroll
    for (int q = 0; q < PS_CP_NQ; ++q) {
        const int b = b0 + kk + 8 * q;
        cj[q] = 0;
#pragma unroll
        for (int c = 0; c < D; ++c) sv[q][c] = 0.0;
        if (b < b1 && r < D) {
            cj[q] = acol_idx[b] * D;
            const double* sb = Saug + (size_t)b * DD + r * D;
#pragma unroll
            for (int c = 0; c < D; ++c) sv[q][c] = sb[c];
        }
    }
    double vr[PS_CP_NV], vw[PS_CP_NV], vs[PS_CP_NV], vp[PS_CP_NV], vx[PS_CP_NV];
    int e0[PS_CP_NV], en[PS_CP_NV];                          // first exchanged entry and number of tasks of this entry's row
#pragma unroll
    for (int v = 0; v < PS_CP_NV; ++v) {
        const int i = t + v * PS_CP_NT;
        vr[v] = vw[v] = vs[v] = vp[v] = vx[v] = 0.0; e0[v] = 0; en[v] = 0;
        if (i < n) {
            vr[v] = r0[i]; vw[v] = w0[i]; vs[v] = s0[i]; vp[v] = p_io[i]; vx[v] = x_io[i];
            const int row = i / D, ta = row_task0[row];
            e0[v] = ta * D + (i - row * D); en[v] = row_task0[row + 1] - ta;
        }
    }
    if (status[ST_PCG_DONE]) return;                         // (as every launch of the per-iteration form)
    // (1 / gamma_prev and 1 / alpha_prev are formed while the exchange is in flight: ONE division between an iteration's dot
    //  products and its recurrences -- the per-launch kernels do three, which changes the last bits of alpha and beta, not more)
    double gamma = 0.0, delta = 0.0, inv_gprev = 0.0, inv_aprev = 0.0, thresh = 0.0;
    bool converged = false;
    long long ck[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define PS_CP_CLK(i) do { if (dbg) { const long long now_ = wall_clock64(); ck[i] += now_ - last_; last_ = now_; } } while (0)
    long long last_ = dbg ? wall_clock64() : 0;
    if constexpr (PIPE) {
        // ---- round 6: the PIPELINED recurrences (Ghysels & Vanroose), option "cg_pipelined".  The Chronopoulos-Gear loop below cannot
        // start an iteration's products before its own dot products are summed (r_{k+1} needs alpha_k, alpha_k needs w_k . r_k, w_k is
        // what the exchange delivers): per iteration  gather -> barrier -> dots -> barrier -> recurrences -> barrier -> products, all on
        // the critical path.  Here the product is applied to w_k = S^ r_k, which the END of the previous iteration already knows
        // (w_k = w_{k-1} - alpha z_{k-1}, z = S^ s): the products are published FIRST and gamma = r.r, delta = w.r, alpha, beta are
        // formed while the exchange is in flight;  then  z = q + beta z,  s = w + beta s,  p = r + beta p,  x += alpha p,
        // r -= alpha s,  w -= alpha z  with q = S^ w_k.  Two barriers per iteration instead of three, the dots off the critical path.
        // Same alpha, beta, gamma sequence in exact arithmetic (history, stopping rule, status words as below); one more replicated
        // vector (z).  The operator is the scaled two-level one (eigenvalues 0.74 .. 2.05): the drift of the recurred w, z against the
        // true products is ~ iterations x eps, far below the 1e-12 the solve stops at.
        double vz[PS_CP_NV];
#pragma unroll
        for (int v = 0; v < PS_CP_NV; ++v) vz[v] = 0.0;
        for (int k = -1; k < nlaunch - 1; ++k) {
            double (*rd)[PS_CP_NT / 64] = red[k & 1];
            // ---- the vector to multiply into LDS (pass -1: r_0, whose product is w_0; then w_k), the two dots' partials beside it
            double gs = 0.0, ds = 0.0;
#pragma unroll
            for (int v = 0; v < PS_CP_NV; ++v) {
                const int i = t + v * PS_CP_NT;
                if (i < n) { rn[i] = (k < 0) ? vr[v] : vw[v]; gs += vr[v] * vr[v]; ds += vw[v] * vr[v]; }
            }
            gs = wave_sum(gs); ds = wave_sum(ds);
            if (lane == 0) { rd[0][wv] = gs; rd[1][wv] = ds; }
            PS_CP_CLK(0);
            __syncthreads();
            PS_CP_CLK(1);
            // ---- this wave's task: six sums of S^(row, its blocks) x that vector, published as tagged granules
            const unsigned tag = salt * 4096u + (unsigned)(k + 2);
            ps_u64* buf = exch + (size_t)(k & 1) * nex * 2;
            {
                double acc = 0.0;
#pragma unroll
                for (int q = 0; q < PS_CP_NQ; ++q) {
                    const double* vv = rn + cj[q];
#pragma unroll
                    for (int c = 0; c < D; ++c) acc += sv[q][c] * vv[c];
                }
                acc += __shfl_xor(acc, 8, 64);
                acc += __shfl_xor(acc, 16, 64);
                acc += __shfl_xor(acc, 32, 64);
                if (task < ntasks && lane < D) cp_put(buf + 2 * ((size_t)task * D + lane), tag, acc);
            }
            PS_CP_CLK(2);
            // ---- while the exchange is in flight: gamma_k = r_k . r_k, delta_k = w_k . r_k, the stopping rule, alpha_k, beta_k
            double alpha = 0.0, beta = 0.0;
            bool stop_now = false;
            if (k >= 0) {
                gamma = 0.0; delta = 0.0;
#pragma unroll
                for (int w2 = 0; w2 < PS_CP_NT / 64; ++w2) { gamma += rd[0][w2]; delta += rd[1][w2]; }
                if (k == 0) thresh = tol2 * gamma;
                if (!(gamma > thresh)) {                     // converged (gamma == 0 too); NaN = breakdown
                    if (chief) { status[ST_PCG_DONE] = (gamma != gamma) ? 2 : 1; scalars[SC_RRFINAL] = gamma; if (k == 0) scalars[SC_RR0] = gamma; }
                    converged = !(gamma != gamma);
                    stop_now = true;
                } else {
                    beta = (k == 0) ? 0.0 : gamma * inv_gprev;
                    const double denom = (k == 0) ? delta : delta - beta * gamma * inv_aprev;
                    alpha = gamma / denom;
                    if (!(denom > 0.0)) {                    // breakdown: stop, the host reports it
                        if (chief) { status[ST_PCG_DONE] = 2; scalars[SC_RRFINAL] = gamma; }
                        stop_now = true;
                    } else {
                        if (chief) {
                            hist[k] = gamma; hist[cap + k] = alpha; status[ST_PCG_ITERS] = k + 1; scalars[SC_RRFINAL] = gamma;
                            if (k == 0) { scalars[SC_THRESH] = thresh; scalars[SC_RR0] = gamma; }
                        }
                        inv_gprev = 1.0 / gamma; inv_aprev = 1.0 / alpha;
                    }
                }
            }
            if (stop_now) break;                             // (uniform: every workgroup sums the same numbers in the same order)
            // ---- gather every published sum, then q of every entry = the sum of its row's tasks, in task order, from LDS
            {
                double gv[NE];
                bool ok = false;
                const long long t_enter = (long long)wall_clock64();
                for (unsigned spins = 0; !ok; ++spins) {
                    ok = true;
#pragma unroll
                    for (int v = 0; v < NE; ++v) {
                        const int j = t + v * PS_CP_NT;
                        gv[v] = 0.0;
                        if (j < (int)nex) {
                            const ps_u64* g = buf + 2 * (size_t)j;
                            const ps_u64 a = __hip_atomic_load((const ps_gu64*)g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            const ps_u64 b = __hip_atomic_load((const ps_gu64*)(g + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            ok = ok && (unsigned)(a >> 32) == tag && (unsigned)(b >> 32) == tag;
                            gv[v] = __longlong_as_double((long long)((a & 0xffffffffull) | (b << 32)));
                        }
                    }
                    ok = __all(ok);
                    if (!ok) {
                        if (spins > spin_limit || (long long)wall_clock64() - t_enter > PS_PERSIST_TIMEOUT_TICKS) { bad = 1; break; }
                        if (PS_CP_SLEEP) __builtin_amdgcn_s_sleep(PS_CP_SLEEP);
                        ck[7] += 1;
                    }
                }
#pragma unroll
                for (int v = 0; v < NE; ++v) { const int j = t + v * PS_CP_NT; if (j < (int)nex) wex[j] = gv[v]; }
            }
            PS_CP_CLK(3);
            __syncthreads();
            PS_CP_CLK(4);
            if (bad) {                                       // an exchange timed out: a breakdown the host answers with the other kernels
                if (t == 0) { status[ST_PCG_DONE] = 2; status[ST_PERSIST_FAIL] = 1; }
                break;
            }
#pragma unroll
            for (int v = 0; v < PS_CP_NV; ++v) {
                double q = 0.0;
                for (int m = 0; m < en[v]; ++m) q += wex[e0[v] + m * D];
                if (k < 0) vw[v] = q;                        // w_0 = S^ r_0
                else {
                    vz[v] = q + beta * vz[v];
                    vs[v] = vw[v] + beta * vs[v];
                    vp[v] = vr[v] + beta * vp[v];
                    vx[v] += alpha * vp[v];
                    vr[v] -= alpha * vs[v];
                    vw[v] -= alpha * vz[v];
                }
            }
            PS_CP_CLK(5);
        }
    } else
    for (int k = -1; k < nlaunch - 1; ++k) {
        double alpha = 0.0, beta = 0.0;
        if (k >= 0) {
            if (k == 0) thresh = tol2 * gamma;
            if (!(gamma > thresh)) {                         // converged (gamma == 0 too); NaN = breakdown
                if (chief) { status[ST_PCG_DONE] = (gamma != gamma) ? 2 : 1; scalars[SC_RRFINAL] = gamma; if (k == 0) scalars[SC_RR0] = gamma; }
                converged = !(gamma != gamma);
                break;
            }
            beta = (k == 0) ? 0.0 : gamma * inv_gprev;
            const double denom = (k == 0) ? delta : delta - beta * gamma * inv_aprev;
            alpha = gamma / denom;
            if (!(denom > 0.0)) {                            // breakdown: stop, the host reports it
                if (chief) { status[ST_PCG_DONE] = 2; scalars[SC_RRFINAL] = gamma; }
                break;
            }
            if (chief) {
                hist[k] = gamma; hist[cap + k] = alpha; status[ST_PCG_ITERS] = k + 1; scalars[SC_RRFINAL] = gamma;
                if (k == 0) { scalars[SC_THRESH] = thresh; scalars[SC_RR0] = gamma; }
            }
        }
        const double gamma_now = gamma, alpha_now = alpha;
        // ---- the recurrences on every entry (replicated), r_new into LDS for the products
        double gs = 0.0;
#pragma unroll
        for (int v = 0; v < PS_CP_NV; ++v) {
            const int i = t + v * PS_CP_NT;
            const double sn = vw[v] + beta * vs[v];
            const double pn = vr[v] + beta * vp[v];
            const double rnv = cg_rnew(vr[v], vw[v], vs[v], alpha, beta);
            vs[v] = sn; vp[v] = pn; vx[v] += alpha * pn; vr[v] = rnv;
            if (i < n) { rn[i] = rnv; gs += rnv * rnv; }
        }
        double (*rd)[PS_CP_NT / 64] = red[k & 1];
        gs = wave_sum(gs);
        if (lane == 0) rd[0][wv] = gs;
        PS_CP_CLK(0);
        __syncthreads();
        PS_CP_CLK(1);
        double gamma_next = 0.0;                             // r_new . r_new, known before the exchange
#pragma unroll
        for (int w2 = 0; w2 < PS_CP_NT / 64; ++w2) gamma_next += rd[0][w2];
        // ---- this wave's task: six sums of S^(row, its blocks) r_new, published as tagged granules
        const unsigned tag = salt * 4096u + (unsigned)(k + 2);
        ps_u64* buf = exch + (size_t)(k & 1) * nex * 2;
        {
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < PS_CP_NQ; ++q) {
                const double* v = rn + cj[q];
#pragma unroll
                for (int c = 0; c < D; ++c) acc += sv[q][c] * v[c];
            }
            acc += __shfl_xor(acc, 8, 64);
            acc += __shfl_xor(acc, 16, 64);
            acc += __shfl_xor(acc, 32, 64);
            if (task < ntasks && lane < D) cp_put(buf + 2 * ((size_t)task * D + lane), tag, acc);
        }
        if (k >= 0) { inv_gprev = 1.0 / gamma_now; inv_aprev = 1.0 / alpha_now; }     // (while the exchange is in flight)
        PS_CP_CLK(2);
        // ---- gather every published sum (flat: the loads of a pass are independent, one round trip), then w_new of every entry
        // = the sum of its row's tasks, in task order, from LDS
        {
            double gv[NE];
            bool ok = false;
            const long long t_enter = (long long)wall_clock64();        // (bounded in wall-clock time too: PS_PERSIST_TIMEOUT_TICKS)
            for (unsigned spins = 0; !ok; ++spins) {
                ok = true;
#pragma unroll
                for (int v = 0; v < NE; ++v) {
                    const int j = t + v * PS_CP_NT;
                    gv[v] = 0.0;
                    if (j < (int)nex) {
                        const ps_u64* g = buf + 2 * (size_t)j;
                        const ps_u64 a = __hip_atomic_load((const ps_gu64*)g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const ps_u64 b = __hip_atomic_load((const ps_gu64*)(g + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = ok && (unsigned)(a >> 32) == tag && (unsigned)(b >> 32) == tag;
                        gv[v] = __longlong_as_double((long long)((a & 0xffffffffull) | (b << 32)));
                    }
                }
                ok = __all(ok);
                if (!ok) {
                    if (spins > spin_limit || (long long)wall_clock64() - t_enter > PS_PERSIST_TIMEOUT_TICKS) { bad = 1; break; }
                    if (PS_CP_SLEEP) __builtin_amdgcn_s_sleep(PS_CP_SLEEP);
                    ck[7] += 1;
                }
            }
#pragma unroll
            for (int v = 0; v < NE; ++v) { const int j = t + v * PS_CP_NT; if (j < (int)nex) wex[j] = gv[v]; }
        }
        PS_CP_CLK(3);
        __syncthreads();
        PS_CP_CLK(4);
#pragma unroll
        for (int v = 0; v < PS_CP_NV; ++v) {
            double wsum = 0.0;
            for (int m = 0; m < en[v]; ++m) wsum += wex[e0[v] + m * D];
            vw[v] = wsum;
        }
        // ---- gamma = r.r, delta = w.r over all entries, by every workgroup in the same order
        double ds = 0.0;
#pragma unroll
        for (int v = 0; v < PS_CP_NV; ++v) ds += vw[v] * vr[v];
        ds = wave_sum(ds);
        if (lane == 0) rd[1][wv] = ds;
        __syncthreads();
        PS_CP_CLK(5);
        if (bad) {                                           // an exchange timed out: a breakdown the host answers with the other kernels
            if (t == 0) { status[ST_PCG_DONE] = 2; status[ST_PERSIST_FAIL] = 1; }
            break;
        }
        gamma = gamma_next; delta = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < PS_CP_NT / 64; ++w2) delta += rd[1][w2];
    }
    if (dbg && chief) for (int i = 0; i < 8; ++i) dbg[i] += ck[i];
#undef PS_CP_CLK
    // ---- what the recovery reads (x^) and what a caller that looks at the state finds: written by the first workgroup
    if (blockIdx.x == 0) {
        if (converged && rec.x) {                            // the recovery of k_coarse_recover, on x^ as this workgroup holds it
            const int nc = rec.ncb * D, nf = rec.nr * D;
            __syncthreads();                                 // (every wave has left the loop: rn, wex are free)
#pragma unroll
            for (int v = 0; v < PS_CP_NV; ++v) { const int i = t + v * PS_CP_NT; if (i < n) rn[i] = vx[v]; }
            __syncthreads();
            const double* xc = rn + nf;
            for (int base = 0; base < nc; base += PS_CP_NT / 8) {
                const int kq = base + t / 8, sub = t & 7;
                double v = 0.0;
                if (kq < nc) for (int m = kq + sub; m < nc; m += 8) v += rec.Lci[(size_t)m * nc + kq] * xc[m];
                v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
                if (kq < nc && sub == 0) wex[kq] = v;
            }
            __syncthreads();
            for (int e = t; e < nf; e += PS_CP_NT) {
                const int i = e / D, c = e - i * D, q = rec.pnode[i];
                const double w0 = rec.pw0[i], w1 = rec.pw1[i];
                double z[D];
#pragma unroll
                for (int m = 0; m < D; ++m) z[m] = w0 * wex[q * D + m] + ((q + 1 < rec.ncb) ? w1 * wex[(q + 1) * D + m] : 0.0);
                double v = 0.0;
#pragma unroll
                for (int a = 0; a < D; ++a) {
                    double xhat = rn[i * D + a];
#pragma unroll
                    for (int m = 0; m < D; ++m) xhat += rec.Bmat[(size_t)i * DD + a * D + m] * z[m];
                    v += rec.Linv[(size_t)i * DD + a * D + c] * xhat;
                }
                rec.x[e] = v;
            }
        }
#pragma unroll
        for (int v = 0; v < PS_CP_NV; ++v) {
            const int i = t + v * PS_CP_NT;
            if (i < n) { x_io[i] = vx[v]; p_io[i] = vp[v]; }
        }
    }
}
