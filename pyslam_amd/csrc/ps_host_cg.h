// ps_host_cg.h -- host side of the reduced solve: stage timers, classic PCG driver, coarse-level construction, fused-CG setup / launch / recovery.
// Part of ps_core.hip (one translation unit; included from there, in this order).

namespace {

// ---- stage timers ---------------------------------------------------------
struct StageTimer {
    ps_problem* h;
    int stage;
    hipEvent_t a = nullptr, b = nullptr;
    int slot = -1;
    StageTimer(ps_problem* h_, int st, int level = 2) : h(h_), stage(st) {
        if (h->profiling < level) return;
        // sampled launches only; the two level-1 pairs on DIFFERENT linearisations (the Schur kernel on multiples of
        // "profile_every", the CG launch half a period later): never two pairs -- ~8 us of pipeline bubbles each -- in one iteration
        if (h->profiling == 1 && h->prof_every > 1 &&
            h->prof_tick % h->prof_every != (st == PS_ST_CG_KERNEL ? h->prof_every / 2 : 0)) return;
        if (h->ev_used + 2 > h->ev_pool.size()) {
            for (int i = 0; i < 64; ++i) { hipEvent_t e; hipEventCreate(&e); h->ev_pool.push_back(e); }
        }
        slot = (int)h->ev_used;
        a = h->ev_pool[h->ev_used++];
        b = h->ev_pool[h->ev_used++];
        hipEventRecord(a, h->stream);
    }
    void stop() {                        // idempotent; the destructor calls it too
        if (!a) return;
        hipEventRecord(b, h->stream);
        h->pending.push_back({stage, slot});
        a = nullptr;
    }
    ~StageTimer() { stop(); }
};

void drain_timers(ps_problem* h) {      // call after a stream synchronisation
    for (auto& pr : h->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, h->ev_pool[pr.second], h->ev_pool[pr.second + 1]) == hipSuccess) {
            h->stage_ms[pr.first] += ms;
            h->stage_n[pr.first] += 1;
        }
    }
    h->pending.clear();
    h->ev_used = 0;
}

int side_kick(ps_problem* h);
int ldi_side_kick(ps_problem* h);
int linearize(ps_problem* h, double lambda, bool allow_prelm = false);

// stage timers whose end event has completed (a speculative linearisation enqueued behind the iteration's end is still running
// when the host leaves wait_published: its events stay pending)
void drain_timers_ready(ps_problem* h) {
    std::vector<std::pair<int, int>> keep;
    for (auto& pr : h->pending) {
        float ms = 0.f;
        if (hipEventQuery(h->ev_pool[pr.second + 1]) != hipSuccess) { keep.push_back(pr); continue; }
        if (hipEventElapsedTime(&ms, h->ev_pool[pr.second], h->ev_pool[pr.second + 1]) == hipSuccess) {
            h->stage_ms[pr.first] += ms;
            h->stage_n[pr.first] += 1;
        }
    }
    h->pending.swap(keep);
    if (h->pending.empty()) h->ev_used = 0;
}

int sync(ps_problem* h) {
    HIP_OK(hipStreamSynchronize(h->stream));
    h->persist_release();               // (whatever one-launch solve was in flight has ended)
    drain_timers(h);
    // whatever produced the side stream's inputs has completed: start the next coarse operator NOW, while the GPU is
    // idle between two calls (kicked from the next call's linearize() it started ~35 us into the iteration and the
    // set-up waited ~20 us for its last kernel at C3)
    if (h->side_todo) h->side_ready = true;     // whatever produced the side stream's inputs has completed
    return 0;
}

// End of a published iteration: watch the sequence word k_reduce3's last workgroup writes to pinned host
// memory (a few microseconds cheaper than a stream synchronisation); falls back to the synchronisation
// when stage timers need their events or the word does not show up in ~1 s.
int wait_published(ps_problem* h) {
    volatile long long* w = h->h_seq;
    struct WaitClock {                  // PS_HOST_TIMING: how long the host waits here (= how far ahead of the GPU its enqueueing ran)
        ps_problem* h; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        ~WaitClock() { h->host_wait_ns += std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count(); ++h->host_waits; }
    } wc{h};
    static const int kick_mode = ps_env("PS_SIDE_KICK") ? atoi(ps_env("PS_SIDE_KICK")) : 2;   // 0 next linearize, 1 after the wait, 2 during the wait
    volatile long long* ws = h->h_setup;
    for (long spins = 0; spins < 400000000L; ++spins) {
        // the set-up kernels of this iteration have finished (stamp of k_coarse_mreduce): the side stream's inputs are
        // complete, and the host has nothing to do but wait -- enqueue the next coarse operator now, beside the CG
        if (kick_mode == 2 && h->side_todo && !h->side_ready && *ws == h->setup_seq) { h->side_ready = true; if (side_kick(h)) return -1; }
        // (lagged dense inverse: k_ldi_init has started, S is final -- the Newton-Schulz step may run beside the solve)
        if (h->ldi_side_todo && *ws == h->setup_seq) { if (ldi_side_kick(h)) return -1; }
        // ps_solve expects another iteration and the reduced solve has converged (the tail's first kernel says so): the next
        // linearisation goes in behind the tail NOW, while the GPU runs the tail and the host has nothing to do but wait
        if (h->early_armed && *reinterpret_cast<volatile long long*>(h->h_early) == h->early_seq) {
            h->early_armed = false; h->spec_enqueued = true;
            // (the tail is open: the landmark pass it carries in place of the cost pass runs, at the point this linearises)
            if (h->prelm_pending) { h->prelm_pending = false; h->prelm_valid = true; }
            if (linearize(h, h->lin_lambda, true)) return -1;
        }
        if (*w == h->seq) {
            // (the tail is behind the reduced solve on the stream: a one-launch solve has ended, its compute units are free --
            //  unless the next linearisation was enqueued behind it, which is not a one-launch solver)
            h->persist_release();
            // a landmark block that was not positive definite in the landmark pass the previous tail ran for THIS call's
            // linearisation (k_landmark_pass_packed<.., COST> reports through a word of its own)
            // (two words, by the parity of the tag: this call's own tail may already have run the pass of the NEXT point, whose
            //  failure -- tag + 1 -- goes to the other word and cannot overwrite this one before it has been looked at: round-5 ADVICE)
            if (h->lmfail_check && *reinterpret_cast<volatile long long*>(h->h_lmfail + (h->lmfail_check & 1)) == h->lmfail_check) h->h_status[ST_LM_FAIL] += 1;
            // an exchange of the one-launch CG timed out (a breakdown as far as the caller is concerned: it solves again with the
            // launch-per-iteration kernels): not used on this handle any more
            if (h->h_status[ST_PERSIST_FAIL] && (h->cg_persist || h->xcg_persist)) { h->cg_persist = 0; h->xcg_persist = 0; ++h->cp_failures; }
            if (h->start_cost_pending) {                     // ps_solve's first iteration: the start cost rode in front of it --
                h->start_cost_pending = false;               // from here on the call knows it, as if ps_eval_cost had run first
                h->last_cost = h->ldi_call_start_cost = h->h_scalars[SC_STARTCOST];
            }
            if (h->side_todo) { h->side_ready = true; if (kick_mode >= 1 && side_kick(h)) return -1; }
            // (a refresh of the lagged inverse reads S on the side stream: not once the next linearisation is overwriting it)
            if (h->ldi_side_todo) { if (h->spec_enqueued) h->ldi_side_todo = false; else if (ldi_side_kick(h)) return -1; }
            h->early_armed = false;
            if (h->pending.empty()) return 0;
            if (h->spec_enqueued) { drain_timers_ready(h); return 0; }     // (the next linearisation's events are still ahead)
            // stage timers: everything up to k_reduce3 has completed; an event recorded behind it may
            // need a moment more
            HIP_OK(hipEventSynchronize(h->ev_pool[h->pending.back().second + 1]));
            drain_timers(h);
            return 0;
        }
        __builtin_ia32_pause();
    }
    return sync(h);
}

int read_scalars(ps_problem* h) {
    HIP_OK(hipMemcpyAsync(h->h_scalars, h->scalars, SC_NWORDS * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipMemcpyAsync(h->h_status, h->status, ST_NWORDS * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    return sync(h);
}

// ---- structure building ---------------------------------------------------
struct PairRec { uint64_t key; int32_t a, b, tile; };

template <int D>
int launch_factor_pass(ps_problem* h, double lambda, double* dbg = nullptr) {
    if (h->F == 0) return 0;
    hipLaunchKernelGGL(k_factor_pass<D>, dim3(cdiv(h->F, PS_FP_FACTORS)), dim3(256), 0, h->stream, (int)h->F, h->f_i,
                       h->f_j, h->f_Tinv, h->f_grp, h->fgroups, h->poses, h->fscratch, dbg);
    if (dbg) return 0;                                        // parity tap: blocks only, S and g stay untouched
    const long threads = (long)h->nes * D * D + (long)h->nr * D;
    hipLaunchKernelGGL(k_factor_assemble<D>, dim3(cdiv(threads, 256)), dim3(256), 0, h->stream, h->nes,
                       h->eslots, h->eptr, h->eitems, h->eslot_diag, h->nr, h->gptr, h->gitems,
                       h->fscratch, lambda, h->S, h->g);
    return 0;
}

template <int D>
int pcg_run(ps_problem* h, double tol, int max_iters, int* iters_out, double* relres_out) {
    const int nr = h->nr;
    if (max_iters + 2 > h->hist_cap) return fail("pcg max_iters exceeds the history buffer (4096)");
    hipLaunchKernelGGL(k_block_jacobi<D>, dim3(cdiv(nr, 256)), dim3(256), 0, h->stream, nr, h->diag_slot,
                       h->S, h->Minv, h->status);
    hipLaunchKernelGGL(k_pcg_init<D>, dim3(h->npartB), dim3(256), 0, h->stream, nr, h->g, h->Minv, h->x,
                       h->r, h->z, h->rz_part, h->rr_part, h->status);
    const double tol2 = tol * tol;
    int k = 0;
    int chunk = std::max(4, h->last_pcg_iters + 1);
    bool done = false;
    while (!done) {
        const int n = std::min(chunk, max_iters + 1 - k);   // +1: the launch that only detects convergence
        h->cg_kernel_launches += 2L * n;
        for (int i = 0; i < n; ++i, ++k) {
            double* pold = (k & 1) ? h->p1 : h->p0;
            double* pnew = (k & 1) ? h->p0 : h->p1;
            hipLaunchKernelGGL(k_pcg_spmv<D>, dim3(h->npartA), dim3(256), 0, h->stream, nr, h->row_ptr,
                               h->col_idx, h->S, h->z, pold, pnew, h->q, h->rz_part, h->rr_part, h->npartB,
                               h->pq_part, h->hist, k, tol2, h->status, h->scalars);
            if (k < max_iters)
                hipLaunchKernelGGL(k_pcg_update<D>, dim3(h->npartB), dim3(256), 0, h->stream, nr, h->Minv,
                                   pnew, h->q, h->x, h->r, h->z, h->pq_part, h->npartA, h->hist, k,
                                   h->rz_part, h->rr_part, h->status);
        }
        HIP_OK(hipMemcpyAsync(h->h_status, h->status, ST_NWORDS * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
        HIP_OK(hipMemcpyAsync(h->h_scalars, h->scalars, SC_NWORDS * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_OK(hipStreamSynchronize(h->stream));
        done = h->h_status[ST_PCG_DONE] != 0 || k > max_iters;
        chunk = 8;
    }
    h->last_pcg_iters = h->h_status[ST_PCG_ITERS];
    if (iters_out) *iters_out = h->h_status[ST_PCG_ITERS];
    const double rr0 = h->h_scalars[SC_RR0], rrf = h->h_scalars[SC_RRFINAL];
    if (relres_out) *relres_out = rr0 > 0.0 ? std::sqrt(rrf / rr0) : 0.0;
    if (h->h_status[ST_DIAG_FAIL])
        return fail("reduced system has a non-positive-definite diagonal block (gauge freedom? hold a pose constant or add a prior)");
    return 0;
}

int ensure_cg_buffers(ps_problem* h, int rows, int blocks) {
    const int D = h->D;
    if ((size_t)rows <= h->cg_cap && h->Saug && (size_t)blocks <= h->saug_cap) return 0;
    const size_t nvec = (size_t)rows * D;
    if (h->alloc(&h->cg_xh, nvec)) return -1;
    for (int k = 0; k < 2; ++k)
        if (h->alloc(&h->cg_r[k], nvec) || h->alloc(&h->cg_w[k], nvec) || h->alloc(&h->cg_s[k], nvec) ||
            h->alloc(&h->cg_gd[k], 2 * (size_t)std::max(rows, 1))) return -1;
    double* pv = nullptr;
    if (h->alloc(&pv, nvec)) return -1;
    h->cg_p = pv;                                  // the fused CG's search direction (own rows only)
    if (h->alloc(&h->Saug, (size_t)std::max(blocks, 1) * D * D)) return -1;
    HIP_OK(hipMemsetAsync(h->Saug, 0, (size_t)std::max(blocks, 1) * D * D * sizeof(double), h->stream));
    h->cg_cap = rows; h->saug_cap = (size_t)std::max(blocks, 1);
    if (!h->cg_tot && h->alloc(&h->cg_tot, 2)) return -1;
    return 0;
}

// Coarse nodes (hat functions over the reduced-pose index) + the augmented BSR pattern
// [[S^, K], [K^T, I]].  coarse_req = number of intervals G (ncb = G + 1 nodes).
int build_coarse(ps_problem* h) {
    const bool timing = ps_env("PS_CREATE_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "build_coarse: %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    const int nr = h->nr, D = h->D;
    {   // compute units the solver's stream may use (the one-launch solvers need their whole grid resident)
        const CuBudget cb = ps_stream_cus(h->stream);
        int dev = 0; if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
        h->persist_dev = dev; h->persist_cus = cb.cus; h->persist_masked = cb.masked;
    }
    if (h->ldi_ready) {                 // a rebuilt coarse level changes the sizes the lagged dense inverse was laid out for
        if (h->side) HIP_OK(hipStreamSynchronize(h->side));
        if (h->ldi_stream) HIP_OK(hipStreamSynchronize(h->ldi_stream));
        h->ldi_ready = false; h->ldi_state = 0; h->ldi_cur = -1; h->ldi_side_todo = false; h->ldi_last_its = 0;
    }
    int G = h->coarse_req;
    const int Gmax = 63;                           // nc = (G + 1) D <= 384; LDS-resident factorisation up to nc = 96
    // auto: on from 16 reduced poses, ~18 poses per hat interval, at most 12 intervals while the
    // coarse factorisation is LDS-resident; large systems (split mode, no dense border rows) take 24
    // 48
    const bool sparse_rows = (long)h->nnzb <= 24L * nr;       // pose-graph-like rows (C2: 11 blocks per row)
    if (G < 0) {
        if (nr < 16) G = 0;                        // (systems up to 90 unknowns are solved directly anyway)
        else if (nr > h->cg_split_min_rows) G = 48;                                    // measured: C4 (BA, 2 000 poses) and C2 (10 000-pose chain)
        else if (sparse_rows && nr >= 150) G = std::min(36, nr / 10);    // pose graphs: 200 poses 67 -> 51 iterations, 350: 85 -> 48
        else if (!sparse_rows && nr > 250) G = std::min(32, nr / 16);    // BA: 400 keyframes 31 -> 19 iterations (0.81 -> 0.70 ms), 500: 39 -> 17
        else G = std::min(12, std::max(3, (nr + 9) / 18));
    }
    // large reduced systems (more than cg_explicit_min_rows poses): the two-level preconditioner is APPLIED explicitly
    // (restrict, dense coarse solve, prolong: cg_explicit) instead of folded into the matrix -- the folded form drags
    // a dense border of ncb blocks through every row (C2: 49 of 60 blocks per row).  Without a border the coarse
    // level can be much finer, and its factorisation runs beside the CG on the side stream from the second
    // iteration on.  Pose-graph-like rows (C2: 11 blocks per row, hundreds of CG iterations): one interval per 25
    // poses, up to 400 (C2, 10 000 poses: 250 / 333 / 400 / 500 intervals give 128 / 97 / 84 / 72 CG iterations and
    // 4.4 / 3.8 / 3.7 / 3.9 ms -- beyond 400 the dense fp32 inverse and its band substitutions cost more than they
    // save; the banded factorisation of ps_k_band.h is what makes more than 255 affordable.  Round 4, with the right-looking
    // band substitutions (k_band_inverse_rl: 0.4 instead of 1.0 ms at C2) and cold, reference-terminated solves as the measure: one
    // interval per 20 poses, up to 500 -- ms per iteration at 1 500 / 3 000 / 5 000 / 10 000 poses 1.57 -> 1.23 / 2.34 -> 2.15 /
    // 2.63 -> 2.32 / 3.90 -> 3.74; finer still (one per 12-15) costs more in the set-up of calls 1-3 than its iterations save);
    // bundle-adjustment rows (C4: 80 blocks per row, ~20 iterations -- the factorisation must fit
    // beside a short CG): one per 20 poses, up to 112 (C4: 42 iterations / 3.7 ms folded at 48 -> 20 / 2.6 ms at 100).
    // Measured crossover against the folded single-launch CG (whose coarse level is capped at 12 intervals):
    // pose graphs 400 poses (600: 2.6 -> 1.4 ms, 1 000: 6.1 -> 1.5 ms), bundle adjustment 540 (700: 1.67 -> 1.28 ms).
    // ... those crossovers were measured against the three-launch explicit PCG of round 2 and are kept where the folded CG
    // can have the lagged dense inverse (ldi_possible: it beats both).  Where it cannot -- more than ldi_max_unknowns, or
    // switched off -- the one- / two-launch explicit PCG of round 3 wins from ~250 poses on (tools/path_threshold_probe.py,
    // settled ms folded / explicit: BA 250 keyframes 0.435 / 0.412, 360: 0.53 / 0.45, 480: 0.71 / 0.51, 539: 0.80 / 0.53;
    // SE(3) graphs 250 poses 0.64 / 0.60, 350: 0.82 / 0.72; at 200 keyframes / poses the folded form still leads, 0.31 / 0.40).
    const bool ldi_possible = h->ldi_enable && (long)nr * D <= h->ldi_max_n && (long)nr * D <= PS_LDI_MAXN && nr * D > h->direct_max;
    const int xmin_auto = ldi_possible ? (sparse_rows ? 400 : 540) : 250;
    h->xmin_auto_ldi = ldi_possible;
    const int xmin = std::min(h->cg_split_min_rows, h->cg_explicit_min_rows >= 0 ? h->cg_explicit_min_rows : xmin_auto);
    h->cg_explicit = h->explicit_ok && G != 0 && nr > xmin;
    if (h->cg_explicit && h->coarse_req < 0)
        // (round 5: beyond 2 048 poses -- the two-launch form, whose coarse kernel reads the inverse once per iteration -- one
        //  interval per 10 poses: 3 000 poses 150 -> 300 intervals 1.72 -> 1.66 ms, 5 000 poses 250 -> 500 1.95 -> 1.86 ms; below,
        //  where every workgroup of the one-launch form reads its own rows of the inverse, finer levels lose: 1 500 poses 75 -> 150
        //  intervals 1.10 -> 1.53 ms)
        G = sparse_rows ? std::min(500, std::max(48, nr > 2048 ? nr / 10 : nr / 20)) : std::min(112, std::max(48, nr / 20));
    G = std::min(G, h->cg_explicit ? PS_XCG_MAXNODES - 1 : Gmax);
    if (h->coarse_clamped && h->coarse_req < 0) G = std::min(G, 255);
    if (G > 0 && nr < 2 * G + 1) G = (nr - 1) / 2;
    if (G < 1) G = 0;
    h->G = G; h->coarse_built = true; h->cg_split = false;
    const std::vector<int32_t>& rp = h->h_row_ptr;
    const std::vector<int32_t>& ci = h->h_col_idx;
    int maxlen = 0;
    for (int i = 0; i < nr; ++i) maxlen = std::max(maxlen, rp[i + 1] - rp[i]);
    h->cg_explicit = h->cg_explicit && G > 0;
    const int ncb_pre = (G && !h->cg_explicit) ? G + 1 : 0;
    // pad rows to a common width unless that wastes more than 50 % (hub-like graphs): then CSR
    const bool ell = nr > 0 && (long)(maxlen + ncb_pre) * nr <= (long)(1.5 * (h->nnzb + (long)ncb_pre * nr)) + 64;
    const int wf = ell ? maxlen + ncb_pre : 0;
    const int wc = ell ? nr + ncb_pre : 0;       // coarse rows: K^T (nr blocks) + the coarse-coarse row (ncb blocks)
    h->ell_wf = wf; h->ell_wc = wc;
    if (G == 0) {
        h->ncb = h->nc = 0; h->nr_aug = nr;
        if (!ell) { h->nnzb_aug = h->nnzb; h->arow_ptr = h->row_ptr; h->acol_idx = h->col_idx; h->aug_slot = h->ident_slot;
                    return ensure_cg_buffers(h, nr, h->nnzb); }
        std::vector<int32_t> arp(nr + 1), aci((size_t)nr * wf, 0), slot(h->nnzb);
        for (int i = 0; i < nr; ++i) {
            arp[i] = i * wf;
            for (int b = rp[i]; b < rp[i + 1]; ++b) { slot[b] = i * wf + (b - rp[i]); aci[slot[b]] = ci[b]; }
        }
        arp[nr] = nr * wf;
        h->nnzb_aug = nr * wf;
        if (h->upload(&h->arow_ptr, arp) || h->upload(&h->acol_idx, aci) || h->upload(&h->aug_slot, slot)) return -1;
        if (ensure_cg_buffers(h, nr, h->nnzb_aug)) return -1;
        HIP_OK(hipMemsetAsync(h->Saug, 0, (size_t)h->nnzb_aug * D * D * sizeof(double), h->stream));
        return 0;
    }
    const int ncb = G + 1;
    std::vector<int32_t> pnode(nr), slo(ncb, nr), shi(ncb, 0);
    std::vector<double> pw0(nr), pw1(nr);
    for (int i = 0; i < nr; ++i) {
        const double u = (double)i * G / (double)(nr - 1);
        const int k = std::min(G - 1, (int)u);
        const double th = u - k;
        pnode[i] = k; pw0[i] = 1.0 - th; pw1[i] = th;
        for (int q = k; q <= k + 1; ++q) {
            // a zero weight at q == k + 1 (row exactly on node k) is skipped; at q == k (the very last row) it is
            // kept: every row must lie in the support of its own left node, which owns its vector updates in
            // the explicit PCG (k_xcg_restrict)
            if (q == k + 1 && pw1[i] == 0.0) continue;
            slo[q] = std::min(slo[q], i); shi[q] = std::max(shi[q], i + 1);
        }
    }
    std::vector<int32_t> arp(nr + ncb + 1, 0), aci, slot(h->nnzb), fnz(nr);
    aci.reserve((size_t)h->nnzb + 2 * (size_t)nr * ncb + ncb);
    for (int i = 0; i < nr; ++i) {
        fnz[i] = rp[i + 1] - rp[i];
        for (int b = rp[i]; b < rp[i + 1]; ++b) { slot[b] = (int32_t)aci.size(); aci.push_back(ci[b]); }
        if (!h->cg_explicit) for (int q = 0; q < ncb; ++q) aci.push_back(nr + q);
        if (ell) while ((int)aci.size() < (i + 1) * wf) aci.push_back(0);     // zero-valued padding blocks
        arp[i + 1] = (int32_t)aci.size();
    }
    const bool split = nr > h->cg_split_min_rows;      // big systems: no dense K^T rows in the matrix
    h->cg_split = split && !h->cg_explicit;
    for (int q = 0; q < ncb && !split; ++q) {       // (explicit mode implies split-sized systems: no coarse rows either)
        for (int i = 0; i < nr; ++i) aci.push_back(i);
        for (int q2 = 0; q2 < ncb; ++q2) aci.push_back(nr + q2);
        arp[nr + q + 1] = (int32_t)aci.size();
    }
    if (split) arp.resize(nr + 1);
    if (split && !h->cg_explicit) {
        if (h->alloc(&h->cg_U, (size_t)ncb * nr * D) || h->alloc(&h->cg_cgd[0], 2 * (size_t)ncb) ||
            h->alloc(&h->cg_cgd[1], 2 * (size_t)ncb) || h->alloc(&h->cg_ab, 2)) return -1;
    }
    lap("nodes + augmented pattern");
    // contiguous run of augmented-matrix blocks of fine row i whose column lies in supp(q)
    // (slo / shi increase with q and the row's columns are sorted: two pointers sweep each row once)
    std::vector<int32_t> rlo, rhi, eptr, eq, elo, ehi;
    if (h->cg_explicit) eptr.assign(nr + 1, 0); else { rlo.resize((size_t)nr * ncb); rhi.resize((size_t)nr * ncb); }
    h->max_row_ents = 0;
    for (int i = 0; i < nr; ++i) {
        int lo = rp[i], hi = rp[i];
        const int end = rp[i + 1];
        // explicit mode lists the non-empty runs only: the nodes a row can reach are those of its first and last column
        // (10 000 rows x 400 nodes: 4 ms of empty sweeps otherwise)
        const int q_first = (h->cg_explicit && end > rp[i]) ? pnode[ci[rp[i]]] : 0;
        const int q_last = (h->cg_explicit && end > rp[i]) ? std::min(ncb - 1, pnode[ci[end - 1]] + 1) : ncb - 1;
        for (int q = q_first; q <= q_last; ++q) {
            while (lo < end && ci[lo] < slo[q]) ++lo;
            while (hi < end && ci[hi] < shi[q]) ++hi;
            if (!h->cg_explicit) {
                rlo[(size_t)i * ncb + q] = arp[i] + (lo - rp[i]);
                rhi[(size_t)i * ncb + q] = arp[i] + (hi - rp[i]);
            } else if (lo < hi) {                          // explicit PCG: only the non-empty runs, listed per row
                eq.push_back(q); elo.push_back(arp[i] + (lo - rp[i])); ehi.push_back(arp[i] + (hi - rp[i]));
            }
        }
        if (h->cg_explicit) {
            eptr[i + 1] = (int32_t)eq.size();
            h->max_row_ents = std::max(h->max_row_ents, eptr[i + 1] - eptr[i]);
        }
    }
    if (h->cg_explicit) {
        // segments: for every node pair (q, q') the entries (i in supp(q), node q') in row order
        std::vector<int32_t> sptr((size_t)ncb * ncb + 1, 0), sent, srow;
        for (int q = 0; q < ncb; ++q)
            for (int i = slo[q]; i < shi[q]; ++i)
                for (int e = eptr[i]; e < eptr[i + 1]; ++e) sptr[(size_t)q * ncb + eq[e] + 1]++;
        for (size_t k = 0; k < (size_t)ncb * ncb; ++k) sptr[k + 1] += sptr[k];
        sent.resize(sptr.back()); srow.resize(sptr.back());
        std::vector<int32_t> pos(sptr.begin(), sptr.end() - 1);
        for (int q = 0; q < ncb; ++q)
            for (int i = slo[q]; i < shi[q]; ++i)
                for (int e = eptr[i]; e < eptr[i + 1]; ++e) {
                    const int32_t at = pos[(size_t)q * ncb + eq[e]]++;
                    sent[at] = e; srow[at] = i;
                }
        // block off-diagonals of A_c = P^T S^ P: (q, q') is non-zero iff some row of supp(q) has a run in supp(q')
        h->ac_bw = 0;
        for (int q = 0; q < ncb; ++q)
            for (int q2 = 0; q2 < ncb; ++q2)
                if (sptr[(size_t)q * ncb + q2 + 1] > sptr[(size_t)q * ncb + q2]) h->ac_bw = std::max(h->ac_bw, std::abs(q - q2));
        if (h->ac_bw > PS_BAND_MAXB) h->ac_bw = -1;
        // More than 255 coarse nodes are only affordable with the BANDED factorisation (round-2 ADVICE): a coarse matrix
        // that turns out not to be banded (long loop closures) would take the dense one at ~4x the work and ~7 nc^2
        // doubles of scratch.  The automatic choice then falls back to the old limit and the level is built again.
        if (h->ac_bw < 0 && h->coarse_req < 0 && G > 255 && !h->coarse_clamped) {
            h->coarse_clamped = true; h->coarse_built = false;
            return build_coarse(h);
        }
        if (h->upload(&h->ent_ptr, eptr) || h->upload(&h->ent_q, eq) || h->upload(&h->ent_lo, elo) ||
            h->upload(&h->ent_hi, ehi) || h->upload(&h->seg_ptr, sptr) || h->upload(&h->seg_ent, sent) ||
            h->upload(&h->seg_row, srow)) return -1;
    }
    lap("runs");
    h->ncb = ncb; h->nc = ncb * D; h->nr_aug = nr + ncb; h->nnzb_aug = (int)aci.size();
    if (h->upload(&h->pnode, pnode) || h->upload(&h->slo, slo) || h->upload(&h->shi, shi) ||
        h->upload(&h->pw0, pw0) || h->upload(&h->pw1, pw1) || (!h->cg_explicit && (h->upload(&h->run_lo, rlo) ||
        h->upload(&h->run_hi, rhi))) || h->upload(&h->arow_ptr, arp) || h->upload(&h->acol_idx, aci) ||
        h->upload(&h->aug_slot, slot) || h->upload(&h->fine_nnz, fnz)) return -1;
    lap("uploads");
    // one-launch folded CG (ps_k_cg_persist.h): every block row of the augmented matrix cut into tasks of at most PS_CP_TASKB
    // blocks, one wave each; the vectors replicated in every workgroup (n <= PS_CP_MAXN)
    h->cp_ok = false;
    if (!h->cg_explicit && !split && (long)(nr + ncb) * D <= PS_CP_MAXN) {
        const int rows = nr + ncb;
        std::vector<CpTask> tasks;
        std::vector<int32_t> rt0(rows + 1, 0);
        for (int i = 0; i < rows; ++i) {
            rt0[i] = (int32_t)tasks.size();
            // (a padded fine row: its own blocks and the ncb border blocks come first, the zero padding behind them is left out)
            const int w = i < nr ? std::min(arp[i + 1] - arp[i], fnz[i] + ncb) : arp[i + 1] - arp[i];
            const int nt = std::max(1, cdiv(w, PS_CP_TASKB)), per = cdiv(w, nt);
            for (int k = 0; k < nt; ++k) tasks.push_back(CpTask{i, arp[i] + k * per, std::min(arp[i] + w, arp[i] + (k + 1) * per), 0});
        }
        rt0[rows] = (int32_t)tasks.size();
        // all workgroups must be resident at once: what the DEVICE says this stream can hold (occupancy of the instantiation x
        // the compute units the stream may use, ps_core.hip: ps_stream_cus), not a literal; at most PS_CP_NE_MAX sums per thread
        const int cp_nwg = cdiv((int)tasks.size(), PS_CP_NT / 64);
        const bool cp_ne6 = (long)tasks.size() * D <= 6L * PS_CP_NT;
        int cp_per_cu = 0;
        {
            // (the Chronopoulos-Gear instantiation: it holds more registers than the pipelined one, so its answer covers both)
            const void* kfn = D == 6 ? (cp_ne6 ? (const void*)k_cg_persist<6, 6, false> : (const void*)k_cg_persist<6, 12, false>)
                                     : (cp_ne6 ? (const void*)k_cg_persist<3, 6, false> : (const void*)k_cg_persist<3, 12, false>);
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&cp_per_cu, kfn, PS_CP_NT, 0) != hipSuccess) { (void)hipGetLastError(); cp_per_cu = 0; }
        }
        h->cp_cus_needed = cp_per_cu > 0 ? cdiv(cp_nwg, cp_per_cu) : 0;
        if ((long)tasks.size() * D <= (long)PS_CP_NE_MAX * PS_CP_NT && h->cp_cus_needed > 0 && h->cp_cus_needed <= h->persist_capacity()) {
            CpTask* dt = nullptr;
            if (h->upload(&dt, tasks) || h->upload(&h->cp_row_task0, rt0) ||
                h->alloc(&h->cp_exch, (size_t)4 * tasks.size() * D)) return -1;
            h->cp_tasks = dt;
            HIP_OK(hipMemsetAsync(h->cp_exch, 0, (size_t)4 * tasks.size() * D * sizeof(unsigned long long), h->stream));
            h->cp_ntasks = (int)tasks.size(); h->cp_salt = 0; h->cp_ok = true;
        }
    }
    if (h->alloc(&h->BSZ, (h->cg_explicit ? eq.size() : (size_t)nr * ncb) * D * D) || h->alloc(&h->Bmat2[0], (size_t)nr * D * D) ||
        h->alloc(&h->bgv, (size_t)nr * D) || h->alloc(&h->SB, (size_t)aci.size() * D * D)) return -1;
    h->Bmat = h->Bmat2[0];
    // lagged three-launch setup: folded single-launch CG only, LDS-resident coarse factor, rows that fit the LDS layout
    h->lagx_ok = !h->cg_explicit && !split && h->nc <= 96 && maxlen <= PS_RS_MAXROW;
    h->side_todo = false; h->rows_attr_set = false;
    if (h->lagx_ok) {
        if (h->alloc(&h->Bmat2[1], (size_t)nr * D * D) || h->alloc(&h->X2[0], (size_t)nr * D * h->nc) ||
            h->alloc(&h->X2[1], (size_t)nr * D * h->nc) || h->alloc(&h->Mpart, (size_t)nr * h->nc * (h->nc + 1))) return -1;
        h->rows_lds = ((size_t)4 * maxlen * D * D + (size_t)D * (3 * h->nc + 1) + D * D + 3 * PS_RS_MAXROW) * sizeof(double)
                      + (PS_RS_MAXROW + 3 * 64) * sizeof(int32_t);
        h->rows_lci_lds = h->rows_lds + (size_t)h->nc * h->nc * sizeof(double) <= 160 * 1024;   // L~^-T beside it, when it fits
        if (h->rows_lci_lds) h->rows_lds += (size_t)h->nc * h->nc * sizeof(double);
    }
    if ((!h->cg_explicit && h->alloc(&h->SZ, (size_t)nr * ncb * D * D)) || h->alloc(&h->Ac, (size_t)h->nc * h->nc) ||
        h->alloc(&h->Lci2[0], (size_t)h->nc * h->nc) || h->alloc(&h->LciT2[0], (size_t)h->nc * h->nc) ||
        h->alloc(&h->Lci2[1], (size_t)h->nc * h->nc) || h->alloc(&h->LciT2[1], (size_t)h->nc * h->nc) ||
        h->alloc(&h->tvec, (size_t)h->nc) || h->alloc(&h->chol_scratch, 2 * (size_t)h->nc * h->nc)) return -1;
    if (h->cg_explicit && (h->alloc(&h->xstate, 8) || h->alloc(&h->xy, (size_t)h->nc) || h->alloc(&h->xp2, (size_t)nr * D))) return -1;
    if (h->cg_explicit && h->ac_bw >= 0 &&
        (h->alloc(&h->Lrow, (size_t)h->nc * PS_BAND_W) || h->alloc(&h->Lcol, (size_t)h->nc * PS_BAND_W) || h->alloc(&h->rdiag, (size_t)h->nc))) return -1;
    h->xcg_rt_ok = false;
    if (h->cg_explicit && ncb <= PS_XCG_MAXNODES) {
        // three-launch form: records of P^T q per (SpMV workgroup, node it touches), a node's records contiguous and in
        // workgroup order (the order k_xcg_coarse_rt sums them in)
        if (const char* e = ps_env("PS_XCG_ROWS_RT")) h->xcg_rt_rows = atoi(e);
        const int R = h->xcg_rt_rows, nwg = cdiv(nr, R);
        std::vector<int32_t> wgo((size_t)nwg * PS_XCG_NSLOT, -1), nptr(ncb + 1, 0);
        bool ok = true;
        for (int g = 0; g < nwg && ok; ++g) {
            const int first = g * R, last = std::min(nr, first + R) - 1;
            const int span = pnode[last] + 1 - pnode[first];
            if (span >= PS_XCG_NSLOT) ok = false;
            else for (int s2 = 0; s2 <= span; ++s2) nptr[pnode[first] + s2 + 1]++;
        }
        if (ok) {
            for (int q = 0; q < ncb; ++q) nptr[q + 1] += nptr[q];
            std::vector<int32_t> pos(nptr.begin(), nptr.end() - 1);
            for (int g = 0; g < nwg; ++g) {
                const int first = g * R, last = std::min(nr, first + R) - 1;
                for (int s2 = 0; s2 <= pnode[last] + 1 - pnode[first]; ++s2) wgo[(size_t)g * PS_XCG_NSLOT + s2] = pos[pnode[first] + s2]++;
            }
            if (h->upload(&h->xcg_wg_out, wgo) || h->upload(&h->xcg_nptr, nptr) ||
                h->alloc(&h->tq_part, (size_t)std::max(1, (int)nptr[ncb]) * D) || h->alloc(&h->tvec2, (size_t)h->nc)) return -1;
            h->xcg_rt_ok = true;
        }
    }
    // one-launch form (k_xcg_fused1): per workgroup of 8 rows the distinct columns of its blocks (+ the LDS slot of every
    // block's column), the contiguous range of coarse nodes those columns hang on, and its records of P^T w at a fixed
    // stride per node (so that their addresses need no pointer load)
    h->xf_ok = false;
    // (fp32 rows of an odd-sized inverse are not 8-byte aligned: SE(2) graphs with an odd node count measured no gain)
    // (beyond that, and for odd sizes: the two-launch form, whose coarse kernel takes any nc up to PS_XF2_NEMAX * 1 024)
    h->xf_one_ok = h->nc <= PS_XF_NEMAX * 64 * PS_XF_ROWS && (h->nc & 1) == 0;
    if (h->cg_explicit && (h->xf_one_ok || h->nc <= PS_XF2_NEMAX * 64 * PS_XCG_CROWS_BIG) && nr >= 2 * PS_XF_ROWS) {
        const int R = PS_XF_ROWS, nwg = cdiv(nr, R);
        std::vector<int32_t> cptr(nwg + 1, 0), cols, nlo(nwg), nhi(nwg), rec((size_t)nwg * PS_XCG_NSLOT, -1), cnt(ncb, 0);
        std::vector<uint16_t> lidx((size_t)h->nnzb_aug, 0);
        std::vector<int32_t> mark(nr, -1);
        bool ok = true;
        int rmax = 1;
        for (int g = 0; g < nwg && ok; ++g) {
            const int first = g * R, last = std::min(nr, first + R) - 1;
            const size_t base = cols.size();
            for (int i = first; i <= last; ++i)
                for (int b = rp[i]; b < rp[i + 1]; ++b) if (mark[ci[b]] != g) { mark[ci[b]] = g; cols.push_back(ci[b]); }
            std::sort(cols.begin() + base, cols.end());
            const int n = (int)(cols.size() - base);
            if (n > PS_XF_CAP || n == 0) { ok = false; break; }
            cptr[g + 1] = (int32_t)cols.size();
            int lo = ncb, hi = 0;
            for (int c = 0; c < n; ++c) { const int j = cols[base + c]; lo = std::min(lo, pnode[j]); hi = std::max(hi, std::min(ncb - 1, pnode[j] + 1)); }
            if (hi - lo + 1 > PS_XF_NODES) { ok = false; break; }
            nlo[g] = lo; nhi[g] = hi;
            for (int i = first; i <= last; ++i) {
                const int a0 = arp[i];
                for (int b = rp[i]; b < rp[i + 1]; ++b)
                    lidx[slot[b]] = (uint16_t)(std::lower_bound(cols.begin() + base, cols.end(), ci[b]) - (cols.begin() + base));
                (void)a0;                                   // (padding blocks of an ELL row keep slot 0: their values are zero)
            }
            const int span = pnode[last] + 1 - pnode[first];
            if (span >= PS_XCG_NSLOT) { ok = false; break; }
            for (int s2 = 0; s2 <= span; ++s2) { const int q = pnode[first] + s2; if (q < ncb) rec[(size_t)g * PS_XCG_NSLOT + s2] = cnt[q]++; }
        }
        if (ok) {
            for (int q = 0; q < ncb; ++q) rmax = std::max(rmax, cnt[q]);
            for (int g = 0; g < nwg; ++g) {
                const int first = g * R;
                for (int s2 = 0; s2 < PS_XCG_NSLOT; ++s2) {
                    int32_t& o = rec[(size_t)g * PS_XCG_NSLOT + s2];
                    if (o >= 0) o = (pnode[first] + s2) * rmax + o;
                }
            }
            const size_t nrec = (size_t)ncb * rmax * D;
            if (h->upload(&h->xf_cptr, cptr) || h->upload(&h->xf_cols, cols) || h->upload(&h->xf_lidx, lidx) ||
                h->upload(&h->xf_nlo, nlo) || h->upload(&h->xf_nhi, nhi) || h->upload(&h->xf_rec, rec) ||
                h->alloc(&h->xf_tq[0], nrec) || h->alloc(&h->xf_tq[1], nrec) || h->alloc(&h->xf_ts[0], (size_t)h->nc) ||
                h->alloc(&h->xf_ts[1], (size_t)h->nc) || h->alloc(&h->xf_t[0], (size_t)h->nc) || h->alloc(&h->xf_t[1], (size_t)h->nc)) return -1;
            HIP_OK(hipMemsetAsync(h->xf_tq[0], 0, nrec * sizeof(double), h->stream));
            HIP_OK(hipMemsetAsync(h->xf_tq[1], 0, nrec * sizeof(double), h->stream));
            h->xf_rmax = rmax; h->xf_nwg = nwg; h->xf_nrec = nrec;
            h->xf_ymax = 0;                                 // most rows of y = A_c^-1 t any workgroup needs (k_xcg_persist4 keeps them in registers)
            for (int g = 0; g < nwg; ++g) h->xf_ymax = std::max(h->xf_ymax, (nhi[g] - nlo[g] + 1) * D);
            h->xf_pf = maxlen <= 16 ? 2 : (maxlen <= 48 ? 6 : 8);
            h->xf_ok = true;
            // one launch per SOLVE (ps_k_xcg_persist.h): all workgroups at once (one per compute unit), the records of a node
            // gathered together, the exchange buffer = [w | partials | records] twice (iteration parity), two granules per double
            h->xp_ok = false;
            // (all workgroups resident at once, one per compute unit -- 256 VGPRs, ~150 KB of LDS: as many as the stream's compute
            //  units, ps_core.hip: ps_stream_cus; the exact instantiation's occupancy is checked again where it is launched)
            h->xp_cus_needed = nwg;
            if (h->xf_one_ok && nwg <= h->persist_capacity() && nrec <= (size_t)PS_XP_NR * 64 * PS_XF_ROWS) {
                const size_t words = 4 * ((size_t)nr * D + 2 * (size_t)nwg + nrec);
                if (h->upload(&h->xf_cnt, cnt) || h->alloc(&h->xp_exch, words)) return -1;
                HIP_OK(hipMemsetAsync(h->xp_exch, 0, words * sizeof(unsigned long long), h->stream));
                h->xp_words = words; h->xp_salt = 0; h->xp_ok = true;
            }
        }
    }
    lap("allocations");
    if (!h->lag_status) {
        if (h->alloc(&h->lag_status, ST_NWORDS)) return -1;
        HIP_OK(hipMemsetAsync(h->lag_status, 0, ST_NWORDS * sizeof(int32_t), h->stream));
    }
    if (!h->side) {
        // An ORDINARY non-blocking stream.  Rounds 1-3 created it with the lowest priority so that the side work would not
        // delay the CG launches; on this stack (ROCm 7.2 / gfx950) kernels of a low-priority stream that run beside
        // normal-priority work now and then produce a different result -- same A_c in, a different inverse out, 8 of 40
        // problems in tools/hunt_explicit_flake.py, none with an ordinary stream; a preconditioner either way, so nothing
        // failed, but runs differed in the last bits and one in ~300 side factorisations reported a spurious failure
        // (DESIGN.md section 4).  The priority bought nothing measurable (C3 0.210 / 0.213 ms, C4 2.00 / 1.97 ms).
        // PS_SIDE_LOWPRIO=1 brings the old stream back (to reproduce).
        // PS_SIDE_CUS=n (measurement switch): confine the side stream to n compute units (CU mask, low bits: spread evenly
        // over the XCDs) instead of running it at low priority over the whole chip
        const int side_cus = ps_env("PS_SIDE_CUS") ? atoi(ps_env("PS_SIDE_CUS")) : h->side_cus;
        if (side_cus > 0) {
            uint32_t mask[8] = {};
            for (int b = 0; b < std::min(side_cus, 256); ++b) mask[b >> 5] |= 1u << (b & 31);
            HIP_OK(hipExtStreamCreateWithCUMask(&h->side, 8, mask));
        } else if (ps_env("PS_SIDE_LOWPRIO")) {
            int prio_lo = 0, prio_hi = 0;
            HIP_OK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
            HIP_OK(hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, prio_lo));
        } else {
            if (!ps_pool().take(ps_pool().side_streams, &h->side)) HIP_OK(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
            h->side_poolable = true;
        }
        HIP_OK(hipEventCreateWithFlags(&h->ev_ac, PS_XSTREAM_EVENT_FLAGS));
        HIP_OK(hipEventCreateWithFlags(&h->ev_acdone, PS_XSTREAM_EVENT_FLAGS));
        HIP_OK(hipEventCreateWithFlags(&h->ev_chol, PS_XSTREAM_EVENT_FLAGS));
    }
    if (h->side_pending) { HIP_OK(hipStreamWaitEvent(h->stream, h->ev_chol, 0)); h->side_pending = false; }
    h->acdone_pending = false;                             // (recorded before ev_chol on the same stream)
    h->lci_next = -1; h->lci_cur = 0; h->xcg_tag[0] = h->xcg_tag[1] = -1.0;
    if (ensure_cg_buffers(h, h->nr_aug, h->nnzb_aug)) return -1;
    HIP_OK(hipMemsetAsync(h->Saug, 0, (size_t)h->nnzb_aug * D * D * sizeof(double), h->stream));
    return 0;
}

// device-to-device copy of n doubles as a kernel on `st` (between kernels of a stream a copy kernel keeps everything on one
// queue and one engine; the copies are small)
inline void copy_doubles(hipStream_t st, double* dst, const double* src, size_t n) {
    if (!n) return;
    hipLaunchKernelGGL(k_copy2, dim3((unsigned)std::min<size_t>(2048, cdiv((long)n, 256))), dim3(256), 0, st, n, src, dst, (size_t)0,
                       (const double*)nullptr, (double*)nullptr);
}

// factor A_c = L_c L_c^T and form L_c^-1 (+ transpose) into buffer `buf`: LDS-resident single workgroup up to 90
// unknowns, blocked over the whole chip beyond
template <int D>
int coarse_factor(ps_problem* h, hipStream_t st, int buf, int32_t* stat) {
    const int nc = h->nc, ncb = h->ncb;
    if (nc <= 90) {                                        // 2 nc^2 doubles of dynamic LDS (<= 130 KB)
        const size_t chol_lds = 2 * (size_t)nc * nc * sizeof(double);
        if (ensure_dynamic_lds((const void*)k_coarse_chol<D, true>, (size_t)(chol_lds))) return -1;
        hipLaunchKernelGGL((k_coarse_chol<D, true>), dim3(1), dim3(1024), chol_lds, st, ncb, h->Ac, h->Lci2[buf],
                           h->LciT2[buf], stat, nullptr);
    } else if (h->big_chol) {
        // blocked factorisation over the whole chip (chol_scratch: working copy of A_c, then the tiles' inverses)
        double* A = h->chol_scratch;
        double* Tinv = A + (size_t)nc * nc;
        copy_doubles(st, A, h->Ac, (size_t)nc * nc);
        const int nsteps = cdiv(nc, PS_BC_W);
        for (int s2 = 0; s2 < nsteps; ++s2) {
            const int j0 = s2 * PS_BC_W, w = std::min(PS_BC_W, nc - j0), m = nc - j0 - w;
            hipLaunchKernelGGL(k_bchol_panel, dim3(std::max(1, cdiv((long)m * w, 1024))), dim3(256), 0, st, nc, j0, A,
                               Tinv + (size_t)s2 * PS_BC_W * PS_BC_W, stat);
            if (m > 0) {
                const int nt = cdiv(m, 32);
                hipLaunchKernelGGL(k_bchol_update, dim3(nt * (nt + 1) / 2), dim3(256), 0, st, nc, j0, w, A);
            }
        }
        // L^-1: diagonal blocks by substitution, the rest merged level by level with triangular products
        // (the explicit PCG only reads the lower triangle of L^-1 and overwrites the transpose: no zero fill there)
        const bool dense_out = nc > PS_BI_S0 && !h->cg_explicit;
        if (dense_out) {
            HIP_OK(hipMemsetAsync(h->Lci2[buf], 0, (size_t)nc * nc * sizeof(double), st));
            HIP_OK(hipMemsetAsync(h->LciT2[buf], 0, (size_t)nc * nc * sizeof(double), st));
        }
        const size_t inv_lds = ((size_t)PS_BI_S0 + PS_BC_W) * PS_BI_CW * sizeof(double);
        hipLaunchKernelGGL(k_btri_inverse, dim3(cdiv(nc, PS_BI_CW)), dim3(256), inv_lds, st, nc, A, Tinv, h->Lci2[buf], h->LciT2[buf]);
        for (int s2 = PS_BI_S0; s2 < nc; s2 *= 2) {
            const int pairs = cdiv(nc, 2 * s2), nt = cdiv(s2, PS_BM_T);
            for (int stage = 0; stage < 2; ++stage)
                hipLaunchKernelGGL(k_btri_merge, dim3(pairs * nt * nt), dim3(256), 0, st, nc, s2, stage, A, h->Lci2[buf], h->LciT2[buf]);
        }
        if (dense_out)
            hipLaunchKernelGGL(k_btri_clear, dim3(cdiv((long)nc * nc, 256)), dim3(256), 0, st, nc, h->LciT2[buf]);
    } else {
        hipLaunchKernelGGL((k_coarse_chol<D, false>), dim3(1), dim3(1024), 0, st, ncb, h->Ac, h->Lci2[buf],
                           h->LciT2[buf], stat, h->chol_scratch);
    }
    return 0;
}

// explicit two-level PCG: A_c^-1 (fp32, symmetric) into LciT2[buf] -- banded factorisation + band substitutions when A_c
// has at most PS_BAND_MAXB block off-diagonals, the dense factorisation and k_xcg_ainv otherwise
template <int D>
int xcg_coarse_inverse(ps_problem* h, hipStream_t st, int buf, int32_t* stat) {
    const int nc = h->nc;
    if (h->band_chol && h->ac_bw >= 0 && h->band_part) {
        // partitioned form (round 5, ps_k_bandpart.h): p independent chunk factorisations + the separator system instead of one
        // chain of ncb block steps -- C2 1.41 -> 0.3 ms, C4 0.29 -> 0.16 ms for the same fp32 inverse (to rounding)
        const int B = std::max(h->ac_bw, 1), m = h->band_part_m > 0 ? h->band_part_m : BandPart::auto_m(h->ncb, B);
        if (BandPart::eligible(h->ncb, B, m)) {
            if (!h->bpart || h->bpart->ncb != h->ncb || h->bpart->B != B || h->bpart->m != m || h->bpart->D != D) {
                if (h->bpart) { HIP_OK(hipStreamSynchronize(st)); if (h->side) HIP_OK(hipStreamSynchronize(h->side)); h->dev_bytes -= h->bpart->bytes; }
                h->bpart.reset(new BandPart());
                if (h->bpart->build(h->ncb, D, B, m, st)) { h->bpart.reset(); return -1; }
                h->dev_bytes += h->bpart->bytes;
            }
            return h->bpart->run<D>(st, h->Ac, nc, (float*)h->LciT2[buf], nc, stat);
        }
    }
    if (h->band_chol && h->ac_bw >= 0) {
        HIP_OK(hipMemsetAsync(h->Lrow, 0, (size_t)nc * PS_BAND_W * sizeof(double), st));
        HIP_OK(hipMemsetAsync(h->Lcol, 0, (size_t)nc * PS_BAND_W * sizeof(double), st));
        hipLaunchKernelGGL(k_band_chol<D>, dim3(1), dim3(256), 0, st, h->ncb, std::max(h->ac_bw, 1), (const double*)h->Ac, h->Lrow, h->Lcol, h->rdiag, stat,
                           nc, (const int2*)nullptr);
        static const bool inv_dot = ps_env("PS_BAND_INV_DOT") != nullptr;      // (measurement build: round 2's dot-product form)
        if (inv_dot)
            hipLaunchKernelGGL(k_band_inverse, dim3(cdiv(nc, 4)), dim3(256), 0, st, nc, h->Lrow, h->Lcol, h->rdiag,
                               h->chol_scratch, (float*)h->LciT2[buf]);
        else
            hipLaunchKernelGGL(k_band_inverse_rl<false>, dim3(cdiv(nc, 4)), dim3(256), 0, st, nc, (const double*)h->Lrow, (const double*)h->Lcol,
                               (const double*)h->rdiag, h->chol_scratch, (float*)h->LciT2[buf], (const BandInvItem*)nullptr, (double*)nullptr, nc);
        return 0;
    }
    if (coarse_factor<D>(h, st, buf, stat)) return -1;
    hipLaunchKernelGGL(k_xcg_ainv, dim3(cdiv(nc, PS_AI_T) * (cdiv(nc, PS_AI_T) + 1) / 2), dim3(256), 0, st, nc, h->Lci2[buf], (float*)h->LciT2[buf], nc);
    return 0;
}

template <int D>
int cg_fused_setup(ps_problem* h, int max_iters, bool allow_lag = false, bool rhs_only = false) {
    const int nr = h->nr, cap = h->hist_cap;
    if (max_iters + 2 > cap) return fail("pcg max_iters exceeds the history buffer (4096)");
    if (!h->coarse_built && build_coarse(h)) return -1;
    h->cg_max_launches = max_iters + 2;
    h->cp_recovered = false;
    h->cg_two_level_reduce = h->nr_aug > 2048 || h->cg_split;
    h->cg_short_rows = (long)h->nnzb_aug <= 24L * h->nr_aug;       // pose-graph-like rows: one wave per row
    const int G = h->G, rows = h->nr_aug;
    const int32_t* rp = h->arow_ptr;
    const int32_t* ci = h->acol_idx;
    if (rhs_only) {
        // same matrix (and coarse factor) as the last full setup, new right-hand side h->g
        hipLaunchKernelGGL(k_cg_prepare<D>, dim3(cdiv((long)nr * D, 256)), dim3(256), 0, h->stream, nr, h->g, h->Linv,
                           h->cg_r[0], h->cg_w[0], h->cg_s[0], h->cg_p, h->cg_xh, h->status,
                           G ? h->Bmat : (const double*)nullptr, h->bgv);
        if (G)
            hipLaunchKernelGGL(k_coarse_rhs<D>, dim3(1), dim3(1024), 0, h->stream, nr, h->ncb, h->slo, h->shi, h->pnode,
                               h->pw0, h->pw1, h->LciT2[h->lci_cur], h->arow_ptr, h->Saug, h->tvec, h->cg_r[0], h->cg_w[0],
                               h->cg_s[0], h->cg_p, h->cg_xh, h->cg_split ? 0 : 2, (const int32_t*)nullptr, h->status,
                               h->bgv);
        h->cg_launched = 0;
        return 0;
    }
    if (h->side_pending) {                  // the side stream still reads SB / the basis and writes its factor buffer
        HIP_OK(hipStreamWaitEvent(h->stream, h->ev_chol, 0));
        h->side_pending = false;
    }
    if (G && h->lagx_ok && h->lagx && allow_lag && h->coarse_lag && h->lci_next >= 0) {
        // ---- three launches: everything coarse (basis, factor, X = P L_c^-T) is the previous iteration's
        const int ncb = h->ncb, nc = h->nc;
        const int use = h->lci_next, nb = use ^ 1;
        h->lci_cur = use;
        h->Bmat = h->Bmat2[use];
        hipLaunchKernelGGL(k_block_jacobi_factor<D>, dim3(cdiv(nr, 4)), dim3(256), 0, h->stream, nr, h->diag_slot,
                           h->S, h->Linv, h->status, h->g, h->cg_r[0], h->cg_w[0], h->cg_s[0], h->cg_p, h->cg_xh,
                           h->poses, h->pose_of_rid, h->coarse_basis, h->Bmat2[nb], h->bgv);
        if (!h->rows_attr_set) {
            if (ensure_dynamic_lds((const void*)k_rows_setup<D>, (size_t)(h->rows_lds))) return -1;
            h->rows_attr_set = true;
        }
        static const int rs_ablate = ps_env("PS_RS_ABLATE") ? atoi(ps_env("PS_RS_ABLATE")) : 0;    // (measurement build only)
        hipLaunchKernelGGL(k_rows_setup<D>, dim3(nr), dim3(PS_RS_THREADS), h->rows_lds, h->stream, nr, ncb, h->row_ptr, h->col_idx,
                           h->aug_slot, h->S, h->Linv, h->Bmat2[use], h->Bmat2[nb], h->arow_ptr, h->fine_nnz,
                           h->run_lo, h->run_hi, h->pnode, h->pw0, h->pw1, h->LciT2[use], h->X2[use], h->cg_r[0],
                           h->Saug, h->SB, h->Mpart, h->rows_lci_lds ? 1 : 0, rs_ablate);
        hipLaunchKernelGGL(k_coarse_mreduce<D>, dim3(cdiv((long)nc * (nc + 1) * 8, 256)), dim3(256), 0, h->stream, nr, ncb,
                           h->shi, h->Mpart, h->pnode, h->arow_ptr, h->Saug, h->cg_r[0], h->cg_w[0], h->cg_s[0], h->cg_p,
                           h->cg_xh, h->lag_status, h->status, h->h_setup_dev, ++h->setup_seq);
        h->mc_active = false;
        h->side_todo = true; h->side_ready = false; h->side_buf = nb;
        h->cg_launched = 0;
        h->last_setup_lagx = true;
        return 0;
    }
    h->last_setup_lagx = false;
    h->side_todo = false;                                   // the exact path below recomputes everything coarse
    if (G) h->Bmat = h->Bmat2[h->lagx_ok ? h->lci_cur : 0];
    // block-Jacobi factors + the start vectors of the scaled system (r = Linv g, w = s = p = x = 0)
    hipLaunchKernelGGL(k_block_jacobi_factor<D>, dim3(cdiv(nr, 4)), dim3(256), 0, h->stream, nr, h->diag_slot,
                       h->S, h->Linv, h->status, h->g, h->cg_r[0], h->cg_w[0], h->cg_s[0], h->cg_p, h->cg_xh,
                       h->poses, h->pose_of_rid, h->coarse_basis, G ? h->Bmat : (double*)nullptr, h->bgv);
    hipLaunchKernelGGL(k_scale_blocks<D>, dim3(h->nnzb), dim3(64), 0, h->stream, nr, h->row_ptr, h->col_idx,
                       h->brow_of, h->Linv, h->S, h->aug_slot, h->Saug, G ? h->Bmat : (const double*)nullptr, h->SB);
    if (G) {
        const int ncb = h->ncb, nc = h->nc;
        if (h->side_pending) {                  // the side-stream factorisation still reads A_c / writes its buffer
            HIP_OK(hipStreamWaitEvent(h->stream, h->ev_chol, 0));
            h->side_pending = false;
        }
        hipLaunchKernelGGL(k_coarse_rowsums<D>, dim3(nr), dim3(256), (size_t)(ncb + 1) * D * D * sizeof(double), h->stream,
                           nr, ncb, h->run_lo, h->run_hi, h->acol_idx, h->pnode, h->pw0, h->pw1, h->SB, h->SZ, h->Bmat, h->BSZ);
        hipLaunchKernelGGL(k_coarse_matrix<D>, dim3(cdiv((long)ncb * ncb * D * D, 256)), dim3(256), 0, h->stream,
                           nr, ncb, h->slo, h->shi, h->pnode, h->pw0, h->pw1, h->BSZ, h->Ac);
        // Exact: factor this iteration's A_c on the solver stream (51 us at C3, serial).  Lagged
        // ("coarse_lag", whole-iteration calls only): build the augmented system with the factor of the
        // PREVIOUS iteration's A_c -- any nonsingular L~ gives a consistent system V^T S^ V with
        // V = [I, P L~^-T]; only the coarse-coarse block changes from I to L~^-1 A_c L~^-T -- and factor
        // the current A_c on a side stream while the CG iterates.
        auto launch_chol = [&](hipStream_t st, int buf, int32_t* stat) -> int { return coarse_factor<D>(h, st, buf, stat); };
        // (long sparse chains in split mode: hundreds of CG iterations dwarf the factorisation, and a stale factor
        // costs iterations while the trajectory still moves -- C2: 1 320 -> 1 800 in the second GN step -- so no lag there)
        const bool lag = allow_lag && h->coarse_lag && h->lci_next >= 0 && (!h->cg_split || (long)h->nnzb > 24L * nr) &&
                         !(h->lagx_ok && h->lagx);          // (such systems take the three-launch path above instead)
        if (lag && h->cg_split && !h->Mc && h->alloc(&h->Mc, (size_t)nc * nc)) return -1;
        h->mc_active = lag && h->cg_split;
        const int rpw = nc >= 192 ? 4 : 1;                 // fine block rows per border workgroup
        const int border_lds = (int)((size_t)rpw * D * nc * sizeof(double));
        if (ensure_dynamic_lds((const void*)k_coarse_border<D>, (size_t)border_lds)) return -1;
        if (lag) {
            const int use = h->lci_next;
            h->lci_cur = use;
            // borders K, K^T, the coarse-coarse rows and (last workgroup) the coarse right-hand side
            const CoarseRhsArgs ra{h->slo, h->shi, h->pnode, h->pw0, h->pw1, h->LciT2[use], h->tvec, h->cg_r[0], h->cg_w[0],
                                   h->cg_s[0], h->cg_p, h->cg_xh, h->cg_split ? 0 : 2, h->lag_status, h->status, h->bgv, h->cg_split ? h->Mc : nullptr};
            hipLaunchKernelGGL(k_coarse_border<D>, dim3(cdiv(nr, rpw) + ncb + 1), dim3(256), border_lds, h->stream,
                               nr, ncb, h->SZ, h->Lci2[use], h->arow_ptr, h->fine_nnz, h->Saug, h->cg_split ? 0 : 1, h->Ac, ra, rpw);
            HIP_OK(hipEventRecord(h->ev_ac, h->stream));           // A_c complete, buffer use^1 no longer read
            HIP_OK(hipStreamWaitEvent(h->side, h->ev_ac, 0));
            if (launch_chol(h->side, use ^ 1, h->lag_status)) return -1;
            HIP_OK(hipEventRecord(h->ev_chol, h->side));
            h->lci_next = use ^ 1; h->side_pending = true;
        } else {
            const int buf = h->lci_cur;
            if (launch_chol(h->stream, buf, h->status)) return -1;
            if (h->lagx_ok)                                 // X = P L_c^-T for a lagged setup of the next iteration
                hipLaunchKernelGGL(k_coarse_xbuild<D>, dim3(cdiv((long)nr * D * nc, 256)), dim3(256), 0, h->stream, nr, ncb,
                                   h->pnode, h->pw0, h->pw1, h->Bmat, h->Lci2[buf], h->X2[buf]);
            const CoarseRhsArgs ra{h->slo, h->shi, h->pnode, h->pw0, h->pw1, h->LciT2[buf], h->tvec, h->cg_r[0], h->cg_w[0],
                                   h->cg_s[0], h->cg_p, h->cg_xh, h->cg_split ? 0 : 1, nullptr, h->status, h->bgv, nullptr};
            hipLaunchKernelGGL(k_coarse_border<D>, dim3(cdiv(nr, rpw) + 1), dim3(256), border_lds, h->stream,
                               nr, ncb, h->SZ, h->Lci2[buf], h->arow_ptr, h->fine_nnz, h->Saug, h->cg_split ? 0 : 1,
                               (const double*)nullptr, ra, rpw);
            h->lci_next = buf;
        }
    }
    h->cg_launched = 0;
    return 0;
}

// Three-launch setup: form the NEXT X = P L_c^-T on the side stream -- row sums with the basis of the last set-up (its
// S^_ij B_j are in SB), A_c, factorisation, k_coarse_xbuild.  Called at the start of a whole-iteration call, so the
// work runs beside the linearisation kernels (enqueued right behind them: the GPU is idle when a call starts, nothing
// may delay its first kernels); its inputs were written by the previous call, and the host has synchronised with the
// solver stream since (side_ready), hence no event is needed to order the side work behind them.
template <int D>
int cg_side_kick(ps_problem* h) {
    if (!h->side_todo || !h->side_ready) return 0;     // (not ready: no host synchronisation since the producer was enqueued)
    h->side_todo = false;
    const int nr = h->nr, ncb = h->ncb, nc = h->nc, nb = h->side_buf;
    hipLaunchKernelGGL(k_coarse_rowsums<D>, dim3(nr), dim3(256), (size_t)(ncb + 1) * D * D * sizeof(double), h->side,
                       nr, ncb, h->run_lo, h->run_hi, h->acol_idx, h->pnode, h->pw0, h->pw1, h->SB, h->SZ, h->Bmat2[nb], h->BSZ);
    hipLaunchKernelGGL(k_coarse_matrix<D>, dim3(cdiv((long)ncb * ncb * D * D, 256)), dim3(256), 0, h->side,
                       nr, ncb, h->slo, h->shi, h->pnode, h->pw0, h->pw1, h->BSZ, h->Ac);
    if (coarse_factor<D>(h, h->side, nb, h->lag_status)) return -1;
    hipLaunchKernelGGL(k_coarse_xbuild<D>, dim3(cdiv((long)nr * D * nc, 256)), dim3(256), 0, h->side, nr, ncb,
                       h->pnode, h->pw0, h->pw1, h->Bmat2[nb], h->Lci2[nb], h->X2[nb]);
    HIP_OK(hipEventRecord(h->ev_chol, h->side));
    h->lci_next = nb; h->side_pending = true;
    return 0;
}

int side_kick(ps_problem* h) {
    if (!h->side_todo) return 0;
    return h->D == 6 ? cg_side_kick<6>(h) : cg_side_kick<3>(h);
}

// enqueue `count` more CG launches (launch n runs iteration k = n - 1; converged launches exit at once)
template <int D>
void cg_fused_launch(ps_problem* h, double tol, int count) {
    const int cap = h->hist_cap;
    const int rows = h->cg_split ? h->nr : h->nr_aug;      // matrix rows handled by k_cg_fused
    const int ncbs = h->cg_split ? h->ncb : 0;
    const double tol2 = tol * tol;
    // the whole solve in ONE launch (ps_k_cg_persist.h) when the system fits its layout: every launch the budget still allows
    // (it stops at convergence by itself), from a fresh set-up only
#ifdef PS_MEASURE
    if (ps_env("PS_CP_CLOCKS") && !h->cp_dbg) { hipMalloc(&h->cp_dbg, 64); hipMemset(h->cp_dbg, 0, 64); }
#endif
    if (h->cg_persist && h->cp_ok && h->G > 0 && h->cg_lds && !h->cg_split && !h->cg_two_level_reduce &&
        !h->cg_ablate && h->cg_launched == 0 && h->cg_max_launches > 0 && h->cg_max_launches <= 4090 && !h->no_repeat &&
        h->persist_reserve(h->cp_cus_needed)) {          // (refused while launches of other handles hold the units: launch by launch)
        const int nl = h->cg_max_launches;
        if (++h->cp_salt >= (1u << 20)) {                    // (tags are salt * 4096 + iteration: start over on a cleared buffer)
            hipMemsetAsync(h->cp_exch, 0, (size_t)4 * h->cp_ntasks * D * sizeof(unsigned long long), h->stream);
            h->cp_salt = 1;
        }
#define PS_CP_LAUNCH_(NE, PIPE) hipLaunchKernelGGL((k_cg_persist<D, NE, PIPE>), dim3(cdiv(h->cp_ntasks, PS_CP_NT / 64)), dim3(PS_CP_NT), 0, h->stream, h->nr_aug * D, \
                           h->cp_ntasks, (const CpTask*)h->cp_tasks, h->cp_row_task0, h->acol_idx, h->Saug, h->cg_r[0], h->cg_w[0], h->cg_s[0], h->cg_p, \
                           h->cg_xh, h->hist, cap, nl, tol2, h->status, h->scalars, h->cp_exch, h->cp_salt, h->cp_spin, h->cp_dbg,                  \
                           CpRecover{h->nr, h->ncb, h->pnode, h->pw0, h->pw1, h->Linv, h->Lci2[h->lci_cur], h->Bmat, h->x})
        // (two instantiations by the number of exchanged sums per thread: the small one keeps 40 registers and 24 KB of LDS free)
        // pipelined recurrences only where the last solve was an easy one (ps_core.hip: cg_pipelined)
        const bool pipe = h->cg_pipelined == 2 || (h->cg_pipelined == 1 && h->last_pcg_iters > 0 && h->last_pcg_iters <= 32);
#define PS_CP_LAUNCH(NE) do { if (pipe) PS_CP_LAUNCH_(NE, true); else PS_CP_LAUNCH_(NE, false); } while (0)
        if ((long)h->cp_ntasks * D <= 6L * PS_CP_NT) PS_CP_LAUNCH(6); else PS_CP_LAUNCH(12);
#undef PS_CP_LAUNCH
#undef PS_CP_LAUNCH_
        h->cg_launched = nl; h->cg_kernel_launches += 1; ++h->cp_launches;
        h->cp_recovered = true;                               // (a converged solve leaves x behind: the gated k_coarse_recover is not needed)
        return;
    }
    h->cg_kernel_launches += count;
    for (int i = 0; i < count; ++i, ++h->cg_launched) {
        const int n = h->cg_launched, o = n & 1, nw = o ^ 1;
        // large systems: totals of the previous launch's partials come from a reduce launch
        const double* tot = h->cg_two_level_reduce ? h->cg_tot : nullptr;
        if (tot && !h->cg_split && n > 0)
            hipLaunchKernelGGL(k_cg_reduce, dim3(1), dim3(1024), 0, h->stream, rows, h->cg_gd[o], h->cg_tot, h->status);
#define PS_CG_LAUNCH(NWV)                                                                                          \
        hipLaunchKernelGGL((k_cg_fused<D, NWV>), dim3(rows), dim3(64 * NWV), 0, h->stream, rows, h->arow_ptr,           \
                           h->acol_idx, h->Saug, h->cg_r[o], h->cg_w[o], h->cg_s[o], h->cg_r[nw], h->cg_w[nw],        \
                           h->cg_s[nw], h->cg_p, h->cg_xh, h->cg_gd[o], h->cg_gd[nw], h->hist, cap, n - 1, tol2,      \
                           h->status, h->scalars, h->nr, h->ell_wf, h->ell_wc, h->cg_ablate, tot, ncbs,              \
                           h->fine_nnz, h->cg_cgd[o], h->cg_U, h->cg_ab)
        if (h->cg_lds && !h->cg_short_rows && !h->cg_split && !tot && !h->cg_ablate && rows <= 1024 &&
            (long)rows * D <= PS_CGV_MAX) {
            // small systems: the whole CG vector goes through LDS, one global round trip per launch
            hipLaunchKernelGGL((k_cg_fused_lds<D, 8>), dim3(rows), dim3(512), 0, h->stream, rows, h->arow_ptr,
                               h->acol_idx, h->Saug, h->cg_r[o], h->cg_w[o], h->cg_s[o], h->cg_r[nw], h->cg_w[nw],
                               h->cg_s[nw], h->cg_p, h->cg_xh, h->cg_gd[o], h->cg_gd[nw], h->hist, cap, n - 1, tol2,
                               h->status, h->scalars, h->nr, h->ell_wf, h->ell_wc);
        }
        else if (h->cg_short_rows) { PS_CG_LAUNCH(1); }
        else if (rows > 1024) { PS_CG_LAUNCH(4); }      // many rows: smaller workgroups, more of them in flight
        else { PS_CG_LAUNCH(8); }
#undef PS_CG_LAUNCH
        if (h->cg_split)      // fine totals + the coarse rows of this iteration
            hipLaunchKernelGGL(k_cg_reduce_split<D>, dim3(1 + h->ncb), dim3(1024), 0, h->stream, rows, h->ncb,
                               h->cg_gd[nw], h->cg_tot, h->cg_U, h->cg_ab, h->cg_r[o], h->cg_w[o], h->cg_s[o],
                               h->cg_r[nw], h->cg_w[nw], h->cg_s[nw], h->cg_p, h->cg_xh, h->cg_cgd[nw], h->status,
                               h->mc_active ? h->Mc : (const double*)nullptr);
    }
}

// x = L^-T (x^_f + P y): gated on the CG's convergence flag when `gate` is given
template <int D>
void cg_fused_recover(ps_problem* h, const int32_t* gate) {
    const int nr = h->nr;
    // the one-launch CG recovers x itself when it converges -- exactly when a GATED recovery would run; an ungated one (the
    // step taken at max_iters, a restart's partial iterate) still runs here
    if (gate && h->cp_recovered) { h->cp_recovered = false; return; }
    h->cp_recovered = false;
    if (h->G)
        hipLaunchKernelGGL(k_coarse_recover<D>, dim3(cdiv((long)nr * D, 256)), dim3(256), 0, h->stream, nr, h->ncb,
                           h->pnode, h->pw0, h->pw1, h->Linv, h->Lci2[h->lci_cur], h->cg_xh, h->x, gate, h->Bmat);
    else
        hipLaunchKernelGGL(k_cg_unscale<D>, dim3(cdiv((long)nr * D, 256)), dim3(256), 0, h->stream, nr, h->Linv,
                           h->cg_xh, h->x, gate);
}

int cg_report(ps_problem* h, int* iters_out, double* relres_out) {
    h->prev_pcg_iters = h->last_pcg_iters;
    h->last_pcg_iters = h->h_status[ST_PCG_ITERS];
    if (iters_out) *iters_out = h->h_status[ST_PCG_ITERS];
    const double rr0 = h->h_scalars[SC_RR0], rrf = h->h_scalars[SC_RRFINAL];
    if (relres_out) *relres_out = rr0 > 0.0 ? std::sqrt(rrf / rr0) : 0.0;
    if (h->h_status[ST_LM_FAIL]) {                           // (its NaNs also spoil the reduced system: report the cause)
        h->lci_next = -1;
        return fail("a landmark block H_ll is not positive definite");
    }
    if (h->h_status[ST_DIAG_FAIL]) {
        if (h->lag_status) hipMemsetAsync(h->lag_status, 0, ST_NWORDS * sizeof(int32_t), h->stream);
        h->lci_next = -1;                               // never reuse a factor from a failed solve
        return fail("reduced system has a non-positive-definite diagonal block (gauge freedom? hold a pose constant or add a prior)");
    }
    if (h->h_status[ST_PCG_DONE] == 2) {
        char buf[200];
        snprintf(buf, sizeof buf, "CG breakdown: the reduced system is not positive definite (iteration %d, relative residual %.2e)",
                 h->h_status[ST_PCG_ITERS], rr0 > 0.0 ? std::sqrt(rrf / rr0) : 0.0);
        return fail(buf);
    }
    return 0;
}

// synchronous solve (staged API): poll the convergence flag every chunk.
// A breakdown of the pipelined (Chronopoulos-Gear) recurrences -- delta - beta gamma / alpha <= 0 long before convergence:
// rounding on the singular folded system, seen on the unit right-hand sides of covariance columns of pose graphs -- is
// answered by RESTARTS: keep the iterate, form the true residual g - S x, and solve for the correction with the same
// matrix and two-level preconditioner (up to three times; then the classic two-launch block-Jacobi PCG).
template <int D>
int cg_fused_run(ps_problem* h, double tol, int max_iters, int* iters_out, double* relres_out, bool rhs_only = false) {
    const int nvec = h->nr * D;
    double* xacc = h->q;                  // the classic PCG's buffers are free while the fused CG runs
    double* gsaved = h->z;
    int restarts = 0, total_its = 0;
    double reduced = 1.0;                 // residual reduction achieved by the passes so far
    bool rhs = rhs_only;
    for (;;) {
        const bool forced = h->cg_force_restart && restarts == 0;      // test hook: stop the first pass early and restart
        const double tol_pass = forced ? std::max(1e-4, tol) : std::min(0.1, tol / reduced);
        if (cg_fused_setup<D>(h, max_iters, false, rhs)) return -1;
        int chunk = std::max(h->pcg_chunk, h->last_pcg_iters + 2);
        bool done = false;
        while (!done) {
            const int m = std::min(chunk, max_iters + 2 - h->cg_launched);
            cg_fused_launch<D>(h, tol_pass, m);
            HIP_OK(hipMemcpyAsync(h->h_status, h->status, ST_NWORDS * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
            HIP_OK(hipMemcpyAsync(h->h_scalars, h->scalars, SC_NWORDS * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            HIP_OK(hipStreamSynchronize(h->stream));
            h->persist_release();
            if (h->h_status[ST_PERSIST_FAIL] && h->cg_persist) { h->cg_persist = 0; ++h->cp_failures; }
            done = h->h_status[ST_PCG_DONE] != 0 || h->cg_launched >= max_iters + 2;
            chunk = h->pcg_chunk;
        }
        total_its += h->h_status[ST_PCG_ITERS];
        const bool pretend = forced && h->h_status[ST_PCG_DONE] == 1 && tol_pass > tol;
        if ((h->h_status[ST_PCG_DONE] != 2 && !pretend) || h->h_status[ST_DIAG_FAIL] || h->h_status[ST_LM_FAIL]) break;
        const double rr0 = h->h_scalars[SC_RR0], rrf = h->h_scalars[SC_RRFINAL];
        const double rho = rr0 > 0.0 ? std::sqrt(rrf / rr0) : 1.0;
        ++h->cg_fallbacks;
        if (restarts == 3 || !(rho < 0.5)) {                // no progress to keep: start over with the classic PCG
            if (restarts) copy_doubles(h->stream, h->g, gsaved, (size_t)nvec);
            return pcg_run<D>(h, tol, max_iters, iters_out, relres_out);
        }
        cg_fused_recover<D>(h, nullptr);                    // x of this pass
        if (restarts == 0) copy_doubles(h->stream, gsaved, h->g, (size_t)nvec);
        hipLaunchKernelGGL(k_vec_accumulate, dim3(cdiv(nvec, 256)), dim3(256), 0, h->stream, nvec, (const double*)h->x, xacc, restarts == 0);
        hipLaunchKernelGGL(k_bsr_residual<D>, dim3(cdiv(nvec, 256)), dim3(256), 0, h->stream, h->nr, h->row_ptr, h->col_idx, h->S,
                           (const double*)xacc, (const double*)gsaved, h->g);
        reduced *= rho;
        rhs = true;
        ++restarts;
    }
    cg_fused_recover<D>(h, nullptr);
    if (restarts) {                                         // x = (sum of the earlier passes) + this correction; g as it was
        hipLaunchKernelGGL(k_vec_accumulate, dim3(cdiv(nvec, 256)), dim3(256), 0, h->stream, nvec, (const double*)xacc, h->x, 0);
        copy_doubles(h->stream, h->g, gsaved, (size_t)nvec);
    }
    int its = 0; double rel = 0.0;
    const int rc = cg_report(h, &its, &rel);
    if (iters_out) *iters_out = total_its;
    if (relres_out) *relres_out = rel * reduced;
    h->last_pcg_iters = total_its;
    return rc;
}

int linearize(ps_problem* h, double lambda, bool allow_prelm) {
    h->solver_touched = true;
    h->prelin_valid = false;                                 // (whatever was linearised ahead is replaced)
    // the landmark pass of THIS point already ran in the previous iteration's tail (in place of its cost pass) and nothing has
    // moved since: Z, C^-1, c are in place; a failure it found is reported by this call (lmfail_check, wait_published)
    // (only the whole-iteration calls take it over: they are the ones that look at h_lmfail)
    const bool lm_done = allow_prelm && h->prelm_valid && h->prelm_lambda == lambda && h->nv > 0;
    h->prelm_valid = false;
    h->lin_lmfail_tag = lm_done ? h->prelm_tag : 0;
    if (lm_done) ++h->prelm_used;
    h->lin_lambda = lambda;                                  // (what the held coarse inverse is tagged with, beside the cost)
    h->params_moved_since_lin = false; h->z_foreign = false; // (Z, C^-1, c of THIS point: run below, or taken over from the pass run ahead here)
    ++h->prof_tick;
    h->cov_ready = false;
    h->status_clean = false;
    if (h->ldi_sread_pending) {         // the side stream's inverse update still converts the previous S (ps_host_ldi.h)
        HIP_OK(hipStreamWaitEvent(h->stream, h->ev_ldi_sread, 0));
        h->ldi_sread_pending = false;
    }
    HIP_OK(hipMemsetAsync(h->red, 0, h->red_count * sizeof(double) + ST_NWORDS * sizeof(int32_t), h->stream));   // [S | g | cost | status]
    if (h->nv > 0 && !lm_done) {
        StageTimer t(h, PS_ST_LANDMARK);
        const ObsWide wl{h->sidx_l, h->stiff_tab};
#define PS_LM_LAUNCH(W) hipLaunchKernelGGL(k_landmark_pass<W>, dim3(cdiv(h->nv, 256 / PS_LM_GROUP)), dim3(256), 0, h->stream, h->nv, h->lm_ptr, \
                           h->lm_point, h->lobs, h->poses, h->points, h->pose_rid, h->ogroups, lambda, h->Z,                       \
                           h->Cinv, h->cvec, h->status, h->lm_ablate, wl)
#define PS_LMP_LAUNCH(W) hipLaunchKernelGGL((k_landmark_pass_packed<W, false>), dim3(cdiv(h->lmw_nwaves, 4)), dim3(256), 0, h->stream, h->lmw_nwaves, \
                           h->lmw_first, h->lm_ptr, h->lm_point, h->lobs, h->poses, h->points, h->pose_rid, h->ogroups, lambda, h->Z,      \
                           h->Cinv, h->cvec, h->status, wl, (const int32_t*)nullptr, (double*)nullptr, (long long*)nullptr, 0LL)
        if (h->lm_packed && h->lmw_nwaves > 0 && !h->lm_ablate) { if (h->wide_obs) PS_LMP_LAUNCH(true); else PS_LMP_LAUNCH(false); }
        else if (h->wide_obs) PS_LM_LAUNCH(true); else PS_LM_LAUNCH(false);
#undef PS_LMP_LAUNCH
#undef PS_LM_LAUNCH
    }
    bool fin_in_combine = false, fin_in_pairs = false;
    static const bool schur_split_env = ps_env("PS_SCHUR_SPLIT") != nullptr;      // (measurement build only; read once)
    // option "pose_async" = 1: the pose pass on a second stream beside the pair kernel, joined in front of the finalisation
    const bool pose_side = h->pose_async == 1 && h->npitems > 0 && h->npair_items > 0 && h->D == 6 && !(h->pose_mode && h->schur_mode != 0) &&
                           !h->use_stream && !h->has_diag_tasks && h->schur_pipeline;
    if (pose_side && !h->aux) {
        if (!ps_pool().take(ps_pool().side_streams, &h->aux)) HIP_OK(hipStreamCreateWithFlags(&h->aux, hipStreamNonBlocking));
        HIP_OK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    }
    if (h->npitems > 0) {
        StageTimer t(h, PS_ST_POSE);
        const ObsWide wp{h->sidx_p, h->stiff_tab};
        hipStream_t pst = h->stream;
        if (pose_side) {
            HIP_OK(hipEventRecord(h->ev_fork, h->stream));       // (the landmark pass -- here or in the previous tail -- is in front of it)
            HIP_OK(hipStreamWaitEvent(h->aux, h->ev_fork, 0));
            pst = h->aux;
        }
        // (option "pose_xcd", default on: the pose-ordered items in eight contiguous ranges, one per XCD -- ps_k_linearize.h)
        const int per_xcd = h->pose_xcd ? cdiv(h->npitems, 8) : 0;
        const int nblk = per_xcd ? 8 * per_xcd : h->npitems;
        if (h->wide_obs)
            hipLaunchKernelGGL(k_pose_pass<true>, dim3(nblk), dim3(256), 0, pst, h->pitems, h->pobs,
                               h->poses, h->points, h->ogroups, h->Cinv, h->cvec, h->ppartial, lambda != 0.0 ? 1 : 0, wp, h->npitems, per_xcd);
        else
            hipLaunchKernelGGL(k_pose_pass<false>, dim3(nblk), dim3(256), 0, pst, h->pitems, h->pobs,
                               h->poses, h->points, h->ogroups, h->Cinv, h->cvec, h->ppartial, lambda != 0.0 ? 1 : 0, wp, h->npitems, per_xcd);
        if (pose_side) HIP_OK(hipEventRecord(h->ev_join, h->aux));
        const bool pose_schur = h->pose_mode && h->schur_mode != 0;
        // tiled Schur: the combine launch also finalizes the poses (unless a task writes a diagonal block)
        fin_in_combine = pose_schur ? h->D == 6
                                    : ((h->Spart || h->use_stream) && h->npair_items > 0 && !h->has_diag_tasks && h->D == 6);
        // untiled Schur with the pipelined pair kernel: its trailing workgroups finalize the poses
        fin_in_pairs = !pose_schur && !h->Spart && !h->use_stream && h->schur_pipeline && h->npair_items > 0 && !h->has_diag_tasks && h->D == 6 &&
                       !schur_split_env && !pose_side;
        if (!fin_in_combine && !fin_in_pairs && !pose_side)
            hipLaunchKernelGGL(k_pose_finalize, dim3(h->nr), dim3(64), 0, h->stream, h->nr, h->pitem_ptr,
                               h->ppartial, h->diag_slot, lambda, h->S, h->g);
    }
    if (h->pose_mode && h->schur_mode != 0) {
        // pose-stationary pair products (ps_k_schur3.h) + the combine launch (partials in segment order; it also finalizes the poses)
        StageTimer t(h, PS_ST_SCHUR, 1);
        if (ensure_dynamic_lds((const void*)k_schur_pose, (size_t)PS_PP_LDS_BYTES)) return -1;
        hipLaunchKernelGGL(k_schur_pose, dim3(8 * h->pp_per_xcd), dim3(PS_PP_THREADS), PS_PP_LDS_BYTES, h->stream, h->pp_per_xcd, h->pp_order, h->pp_segs,
                           h->pp_rows, h->pp_tasks, h->pp_pairs, h->Z, h->pp_part, h->schur_ablate);
        hipLaunchKernelGGL(k_schur_combine, dim3(cdiv(h->pp_ncomb, 4) + (fin_in_combine ? cdiv(h->nr, 4) : 0)), dim3(256), 0,
                           h->stream, h->pp_ncomb, h->pp_comb_items, h->pp_comb_tasks, h->pp_part, h->S,
                           fin_in_combine ? h->nr : 0, h->pitem_ptr, h->ppartial, h->diag_slot, lambda, h->g);
    } else if (h->npair_items > 0 && h->use_stream) {
        StageTimer t(h, PS_ST_SCHUR, 1);
        const size_t lds = (size_t)PS_ST_SUBROWS * PS_ST_ROWD * sizeof(double);
        if (!h->st_attr_set) {
            if (ensure_dynamic_lds((const void*)k_schur_stream, (size_t)(lds))) return -1;
            h->st_attr_set = true;
        }
        hipLaunchKernelGGL(k_schur_stream, dim3(h->st_ntiles), dim3(PS_ST_THREADS), lds, h->stream, h->st_tiles, h->st_subs,
                           reinterpret_cast<const uint4*>(h->st_entries), h->Z, h->st_part, h->schur_ablate);
        hipLaunchKernelGGL(k_schur_combine, dim3(cdiv(h->st_ncomb, 4) + (fin_in_combine ? cdiv(h->nr, 4) : 0)), dim3(256), 0,
                           h->stream, h->st_ncomb, h->st_comb_items, h->st_comb_tasks, h->st_part, h->S,
                           fin_in_combine ? h->nr : 0, h->pitem_ptr, h->ppartial, h->diag_slot, lambda, h->g);
    } else if (h->npair_items > 0) {
        StageTimer t(h, PS_ST_SCHUR, 1);
        // PS_SCHUR_SPLIT=1 (measurement switch): the same work as two launches over the two halves of
        // every XCD's list -- what splitting the Schur build into two bands for an overlapped all-reduce would cost
        const bool split2 = schur_split_env;
        const int halfp = split2 ? (h->pair_per_xcd / 2 + 3) / 4 * 4 : h->pair_per_xcd;
        // PS_SCHUR_LDS_PAD=<bytes> (measurement switch): unused dynamic LDS per workgroup, i.e. fewer resident waves per CU
        static const int lds_pad = ps_env("PS_SCHUR_LDS_PAD") ? atoi(ps_env("PS_SCHUR_LDS_PAD")) : 0;
        auto launch_pairs = [&](int nblk, int lds, int from, int to) {
            if (!h->schur_pipeline)
                hipLaunchKernelGGL(k_schur_pairs, dim3(nblk), dim3(256), lds, h->stream, h->pair_per_xcd, h->pair_xitems, h->pairs,
                                   h->Z, h->S, h->Spart, h->schur_ablate, from, to);
            else {
#ifdef PS_MEASURE
                auto k = h->schur_ablate == 0 ? k_schur_pairs_db<0> : h->schur_ablate == 1 ? k_schur_pairs_db<1> :
                         h->schur_ablate == 2 ? k_schur_pairs_db<2> : h->schur_ablate == 3 ? k_schur_pairs_db<3> :
                         h->schur_ablate == 4 ? k_schur_pairs_db<4> : k_schur_pairs_db<5>;
#else
                auto k = k_schur_pairs_db<0>;                   // (the ablation instantiations exist in the measurement build only)
#endif
                hipLaunchKernelGGL(k, dim3(nblk + (fin_in_pairs ? cdiv(h->nr, 4) : 0)), dim3(256), lds, h->stream, h->pair_per_xcd,
                                   h->pair_xitems, h->pairs, h->Z, h->S, h->Spart, from, to, nblk, fin_in_pairs ? h->nr : 0,
                                   h->pitem_ptr, h->ppartial, h->diag_slot, lambda, h->g);
            }
        };
        launch_pairs(8 * (halfp / 4), lds_pad, 0, halfp);
        if (split2 && halfp < h->pair_per_xcd) launch_pairs(8 * ((h->pair_per_xcd - halfp + 3) / 4), 0, halfp, h->pair_per_xcd);
        if (pose_side) {                                     // the pose pass ran beside the pair kernel: its partials are needed from here
            HIP_OK(hipStreamWaitEvent(h->stream, h->ev_join, 0));
            if (!fin_in_combine)
                hipLaunchKernelGGL(k_pose_finalize, dim3(h->nr), dim3(64), 0, h->stream, h->nr, h->pitem_ptr,
                                   h->ppartial, h->diag_slot, lambda, h->S, h->g);
        }

        if (h->Spart)
            hipLaunchKernelGGL(k_schur_combine, dim3(cdiv(h->ncomb, 4) + (fin_in_combine ? cdiv(h->nr, 4) : 0)), dim3(256), 0,
                               h->stream, h->ncomb, h->comb_items, h->comb_tasks, h->Spart, h->S,
                               fin_in_combine ? h->nr : 0, h->pitem_ptr, h->ppartial, h->diag_slot, lambda, h->g);
    }
    if (h->F > 0 && h->nr > 0) {
        StageTimer t(h, PS_ST_EDGES);
        if (h->D == 6) launch_factor_pass<6>(h, lambda); else launch_factor_pass<3>(h, lambda);
    }
    return side_kick(h);                        // next coarse operator: side stream, beside the kernels above
}

// cost partials into cost_partials[0..n); returns n.  The caller reduces them.
// every observation is in the packed landmark pass's runs (no observation of a constant point): costs are summed in that pass's
// structure, by k_cost_packed or by the pass itself
inline bool cost_packed_possible(const ps_problem* h) {
    return h->fuse_cost != 0 && h->lm_packed && h->lmw_nwaves > 0 && !h->lm_ablate && h->Nl == h->N && h->nv > 0 && h->D == 6;
}
int cost_partials_pass(ps_problem* h, int include_all, const int32_t* gate) {
    int n = 0;
    if (h->N > 0 && include_all && cost_packed_possible(h)) {
        // the partial sums of the packed landmark pass (ps_k_packed.h): a cost is the same number, bit for bit, whether that pass
        // summed it on its way (the tail of an iteration that expects a successor, the start cost) or this one did
        const ObsWide wl{h->sidx_l, h->stiff_tab};
        const int nbl = cdiv(h->lmw_nwaves, 4);
        if (h->wide_obs)
            hipLaunchKernelGGL(k_cost_packed<true>, dim3(nbl), dim3(256), 0, h->stream, h->lmw_nwaves, h->lmw_first, h->lm_ptr, h->lm_point,
                               h->lobs, h->poses, h->points, h->ogroups, h->cost_partials, gate, wl);
        else
            hipLaunchKernelGGL(k_cost_packed<false>, dim3(nbl), dim3(256), 0, h->stream, h->lmw_nwaves, h->lmw_first, h->lm_ptr, h->lm_point,
                               h->lobs, h->poses, h->points, h->ogroups, h->cost_partials, gate, wl);
        n += nbl;
    } else if (h->N > 0) {
        const ObsWide wl{h->sidx_l, h->stiff_tab};
        if (h->wide_obs)
            hipLaunchKernelGGL(k_cost_reproj<true>, dim3(h->ncost_obs), dim3(256), 0, h->stream, h->N, h->lobs, h->poses,
                               h->points, h->pose_rid, h->point_vid, h->ogroups, include_all, h->cost_partials, gate, wl);
        else
            hipLaunchKernelGGL(k_cost_reproj<false>, dim3(h->ncost_obs), dim3(256), 0, h->stream, h->N, h->lobs, h->poses,
                               h->points, h->pose_rid, h->point_vid, h->ogroups, include_all, h->cost_partials, gate, wl);
        n += h->ncost_obs;
    }
    if (h->F > 0) {
        if (h->D == 6)
            hipLaunchKernelGGL(k_cost_factors<6>, dim3(h->ncost_fac), dim3(256), 0, h->stream, (int)h->F, h->f_i,
                               h->f_j, h->f_Tinv, h->f_grp, h->fgroups, h->poses, h->pose_rid, include_all,
                               h->cost_partials + n, gate);
        else
            hipLaunchKernelGGL(k_cost_factors<3>, dim3(h->ncost_fac), dim3(256), 0, h->stream, (int)h->F, h->f_i,
                               h->f_j, h->f_Tinv, h->f_grp, h->fgroups, h->poses, h->pose_rid, include_all,
                               h->cost_partials + n, gate);
        n += h->ncost_fac;
    }
    return n;
}

int cost_pass(ps_problem* h, int include_all, int scalar_slot) {
    StageTimer t(h, PS_ST_COST);
    const int n = cost_partials_pass(h, include_all, nullptr);
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, h->stream, n, h->cost_partials,
                       h->scalars + scalar_slot);
    return 0;
}

// workgroups (= partials of ||dx_l||^2) of the back-substitution as it is launched now
inline int nsq_l_now(const ps_problem* h) { return (h->lm_packed && h->lmw_nwaves > 0) ? cdiv(h->lmw_nwaves, 4) : h->nsq_l16; }

int backsub(ps_problem* h, const int32_t* gate = nullptr, bool fuse_update = false, long long* hearly = nullptr, long long eseq = 0) {
    if (h->nv == 0) return 0;
    StageTimer t(h, PS_ST_BACKSUB);
    if (h->lm_packed && h->lmw_nwaves > 0) {                // lanes packed by observation (ps_k_packed.h)
        const int nbl = cdiv(h->lmw_nwaves, 4);               // (= nsq_l_now(h): the partials of ||dx_l||^2 k_reduce3 sums)
        if (fuse_update)
            hipLaunchKernelGGL(k_backsub_packed, dim3(nbl + h->nsq_p), dim3(256), 0, h->stream, h->lmw_nwaves, h->lmw_first, h->lm_ptr, h->Z,
                               h->Cinv, h->cvec, h->x, h->dxl, h->sq_part_l, gate, nbl, h->lm_point, h->points, h->P, h->pose_rid, h->poses,
                               h->sq_part_p, hearly, eseq);
        else
            hipLaunchKernelGGL(k_backsub_packed, dim3(nbl), dim3(256), 0, h->stream, h->lmw_nwaves, h->lmw_first, h->lm_ptr, h->Z,
                               h->Cinv, h->cvec, h->x, h->dxl, h->sq_part_l, gate, nbl, (const int32_t*)nullptr, (double*)nullptr, 0,
                               (const int32_t*)nullptr, (double*)nullptr, (double*)nullptr, hearly, eseq);
        return 0;
    }
    if (fuse_update)       // + full-step landmark update + SE(3) retraction of the poses in the same launch
        hipLaunchKernelGGL(k_backsub, dim3(h->nsq_l16 + h->nsq_p), dim3(256), 0, h->stream, h->nv, h->lm_ptr, h->lobs,
                           h->pose_rid, h->Z, h->Cinv, h->cvec, h->x, h->dxl, h->sq_part_l, gate,
                           h->nsq_l16, h->lm_point, h->points, h->P, h->poses, h->sq_part_p, hearly, eseq);
    else
        hipLaunchKernelGGL(k_backsub, dim3(h->nsq_l16), dim3(256), 0, h->stream, h->nv, h->lm_ptr, h->lobs,
                           h->pose_rid, h->Z, h->Cinv, h->cvec, h->x, h->dxl, h->sq_part_l, gate,
                           h->nsq_l16, (const int32_t*)nullptr, (double*)nullptr, 0, (double*)nullptr, (double*)nullptr, hearly, eseq);
    return 0;
}

int step_norm(ps_problem* h) {
    // standalone ||dx||^2 (ps_step_norm2): partial sums of squares of x and dxl, then two small reduces
    double* part = h->cost_partials;
    HIP_OK(hipMemsetAsync(h->scalars + SC_DXP2, 0, sizeof(double), h->stream));
    HIP_OK(hipMemsetAsync(h->scalars + SC_DXL2, 0, sizeof(double), h->stream));
    if (h->nr > 0) {
        const int b = std::min(256, cdiv((long)h->nr * h->D, 256));
        hipLaunchKernelGGL(k_sumsq_partials, dim3(b), dim3(256), 0, h->stream, (long)h->nr * h->D, h->x, 1.0, part);
        hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, h->stream, b, part, h->scalars + SC_DXP2);
    }
    if (h->nv > 0) {
        const int b = std::min(256, cdiv((long)h->nv * 3, 256));
        hipLaunchKernelGGL(k_sumsq_partials, dim3(b), dim3(256), 0, h->stream, (long)h->nv * 3, h->dxl, 1.0, part + 256);
        hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, h->stream, b, part + 256, h->scalars + SC_DXL2);
    }
    return 0;
}

int apply_update(ps_problem* h, double step, const int32_t* gate = nullptr, bool with_norm = false, long long* hearly = nullptr, long long eseq = 0) {
    h->params_moved_since_lin = true;
    h->prelin_valid = h->prelm_valid = false;   // the parameters move: a linearisation enqueued ahead is of the old point (set again by
                                        // gn_iteration_impl AFTER its tail, for the speculative one enqueued behind that tail)
    StageTimer t(h, PS_ST_UPDATE);
    if (h->nr > 0) {
        double* sq = with_norm ? h->sq_part_p : nullptr;
        if (h->D == 6)
            hipLaunchKernelGGL(k_update_poses<6>, dim3(cdiv(h->P, 256)), dim3(256), 0, h->stream, h->P, h->pose_rid, h->x, step, h->poses, sq, gate, hearly, eseq);
        else
            hipLaunchKernelGGL(k_update_poses<3>, dim3(cdiv(h->P, 256)), dim3(256), 0, h->stream, h->P, h->pose_rid, h->x, step, h->poses, sq, gate, hearly, eseq);
    }
    if (h->nv > 0)
        hipLaunchKernelGGL(k_update_points, dim3(cdiv((long)h->nv * 3, 256)), dim3(256), 0, h->stream, h->nv,
                           h->lm_point, h->dxl, step, h->points, gate);
    return 0;
}

// ---- the robust cost summed by the landmark pass itself (k_landmark_pass_packed<.., COST>, ps_k_packed.h) --------------------
// every observation must be in the packed pass's runs (no observation of a constant point); factors add their own partials
// (a landmark shard: only in the core's own sharded iteration -- ps_set_collective --, which takes the pass over in its next
//  linearisation and carries a failure found ahead through the exchange; a caller that drives the collectives itself keeps the
//  cost-only pass)
inline bool lm_cost_possible(const ps_problem* h) {
    return cost_packed_possible(h) && (!h->shard_out || (h->nccl_allreduce && h->nccl_comm));
}
// enqueue it at the current parameters; -> number of partials in cost_partials.  Leaves Z, C^-1, c of this point behind:
// the caller decides whether that makes prelm_valid (an ungated launch) or prelm_pending (a gated tail).
int lm_cost_enqueue(ps_problem* h, double lambda, const int32_t* gate) {
    const int nbl = cdiv(h->lmw_nwaves, 4);
    if (h->params_moved_since_lin || lambda != h->lin_lambda) h->z_foreign = true;      // (Z, C^-1, c leave the last linearisation's point)
    h->prelm_tag = ++h->prelm_seq;
    {
        StageTimer t(h, PS_ST_LANDMARK);
        const ObsWide wl{h->sidx_l, h->stiff_tab};
#define PS_LMC_LAUNCH(W) hipLaunchKernelGGL((k_landmark_pass_packed<W, true>), dim3(nbl), dim3(256), 0, h->stream, h->lmw_nwaves, h->lmw_first,    \
                           h->lm_ptr, h->lm_point, h->lobs, h->poses, h->points, h->pose_rid, h->ogroups, lambda, h->Z, h->Cinv, h->cvec,        \
                           h->status, wl, gate, h->cost_partials, h->h_lmfail_dev + (h->prelm_tag & 1), h->prelm_tag)
        if (h->wide_obs) PS_LMC_LAUNCH(true); else PS_LMC_LAUNCH(false);
#undef PS_LMC_LAUNCH
    }
    int n = nbl;
    if (h->F > 0) {
        StageTimer t(h, PS_ST_COST);
        hipLaunchKernelGGL(k_cost_factors<6>, dim3(h->ncost_fac), dim3(256), 0, h->stream, (int)h->F, h->f_i, h->f_j, h->f_Tinv, h->f_grp,
                           h->fgroups, h->poses, h->pose_rid, 1, h->cost_partials + nbl, gate);
        n += h->ncost_fac;
    }
    h->prelm_lambda = lambda;
    return n;
}
// the cost at the current parameters (all blocks) into scalars[scalar_slot], by the landmark pass: the next linearisation at
// this point and this lambda finds its landmark pass done
int lm_cost_pass(ps_problem* h, double lambda, int scalar_slot) {
    h->prelin_valid = false;                                 // (Z, C^-1, c are rewritten -- with the same values if nothing moved)
    const int n = lm_cost_enqueue(h, lambda, nullptr);
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, h->stream, n, h->cost_partials, h->scalars + scalar_slot);
    h->prelm_pending = false; h->prelm_valid = true;
    return 0;
}

// back-substitution, update, cost and ||dx||^2 with ONE final reduction launch.  `gate` (device
// status words) makes every kernel a no-op until the CG has flagged convergence.
int gn_tail(ps_problem* h, int linesearch, const int32_t* gate, bool publish = false) {
    h->params_moved_since_lin = true;
    h->prelin_valid = h->prelm_valid = false;   // (as apply_update: every tail moves the parameters)
    h->prelm_pending = false;
    // line-search order (cost AFTER the step): back-substitution, landmark update and pose retraction
    // are one launch when the problem has landmarks (then D == 6)
    const bool fused = linesearch && h->nv > 0 && h->nr > 0 && h->D == 6;
    // the early word (speculative next linearisation, wait_published): stamped by the first kernel of the tail that reads the gate
    long long* const hearly = (publish && h->spec_next && !h->spec_enqueued && gate) ? h->h_early_dev : nullptr;
    const long long eseq = hearly ? ++h->early_seq : 0;
    h->early_armed = hearly != nullptr;
    const bool stamp_in_backsub = h->nv > 0;
    if (backsub(h, gate, fused, stamp_in_backsub ? hearly : nullptr, eseq)) return -1;
    int ncost = 0;
    if (!linesearch) { StageTimer t(h, PS_ST_COST); ncost = cost_partials_pass(h, 0, gate); }
    if (!fused && apply_update(h, 1.0, gate, true, stamp_in_backsub ? nullptr : hearly, eseq)) return -1;
    if (!stamp_in_backsub && (fused || h->nr == 0)) h->early_armed = false;     // (no kernel carried the word)
    // The cost after the step IS the cost at the next linearisation point: when a successor is expected (option "expect_next")
    // its landmark pass runs here, gated like the rest of the tail, and sums the cost as it goes -- every observation is
    // evaluated once instead of twice (C3: the 11-14 us of the cost pass; C4: 75).  Needs every observation in the packed
    // landmark pass's runs (no observation of a constant point) and the published end of the iteration.
    const bool lm_cost = fused && (publish || (h->shard_out && h->nccl_allreduce && h->nccl_comm)) && h->expect_next && lm_cost_possible(h);
    if (lm_cost) {
        ncost = lm_cost_enqueue(h, h->lin_lambda, gate);
        h->prelm_pending = true;
    } else if (linesearch) { StageTimer t(h, PS_ST_COST); ncost = cost_partials_pass(h, 1, gate); }
    double* o_cost = h->shard_out ? h->shard_buf : h->scalars + (linesearch ? SC_COST : SC_LINCOST);
    double* o_dxl = h->shard_out ? h->shard_buf + 1 : h->scalars + SC_DXL2;
    hipLaunchKernelGGL(k_reduce3, dim3(3), dim3(256), 0, h->stream,
                       ncost, h->cost_partials, o_cost,
                       h->nsq_p, h->sq_part_p, h->nr > 0 ? h->scalars + SC_DXP2 : nullptr,
                       nsq_l_now(h), h->sq_part_l, (h->nv > 0 || h->shard_out) ? o_dxl : nullptr, gate,
                       h->status, h->scalars, publish ? h->h_status_dev : nullptr, publish ? h->h_scalars_dev : nullptr,
                       h->arrivals, publish ? h->h_seq_dev : nullptr, publish ? ++h->seq : 0LL);
    return 0;
}

}  // namespace
