// ps_host_iteration.h -- explicit two-level PCG driver, linearisation, tail, the one-synchronisation Gauss-Newton iteration, small host helpers.
// Part of ps_core.hip (one translation unit; included from there, in this order).

namespace {

// ---- explicit two-level PCG (long sparse chains; kernels k_xcg_*) ---------------------------------
// (measurement switch: PS_XCG_AC_MAIN=1 keeps the assembly of A_c on the solver stream, as it was before it moved)
inline bool xcg_ac_on_main() { static const bool v = ps_env("PS_XCG_AC_MAIN") != nullptr; return v; }

// A_c = P^T S^ P from the non-empty (row, node) runs
template <int D>
void xcg_assemble_ac(ps_problem* h, hipStream_t st) {
    const int nr = h->nr, ncb = h->ncb;
    hipLaunchKernelGGL(k_xcoarse_rowsums<D>, dim3(nr), dim3(256), (size_t)(h->max_row_ents + 1) * D * D * sizeof(double), st,
                       nr, h->ent_ptr, h->ent_q, h->ent_lo, h->ent_hi, h->acol_idx, h->pnode, h->pw0, h->pw1, h->SB, h->Bmat, h->BSZ);
    hipLaunchKernelGGL(k_xcoarse_matrix<D>, dim3(cdiv((long)ncb * ncb * D * D, 256)), dim3(256), 0, st,
                       ncb, h->seg_ptr, h->seg_ent, h->seg_row, h->pnode, h->pw0, h->pw1, h->BSZ, h->Ac);
}

template <int D>
void xcg_launch(ps_problem* h, double tol, int count);

template <int D>
int xcg_setup(ps_problem* h, int max_iters, bool allow_lag) {
    const int nr = h->nr, ncb = h->ncb, nc = h->nc;
    if (max_iters + 2 > h->hist_cap) return fail("pcg max_iters exceeds the history buffer (4096)");
    // (the side stream assembles A_c from SB and the basis blocks of the previous lagged set-up: both are rewritten below)
    if (h->acdone_pending) { HIP_OK(hipStreamWaitEvent(h->stream, h->ev_acdone, 0)); h->acdone_pending = false; }
    hipLaunchKernelGGL(k_block_jacobi_factor<D>, dim3(cdiv(nr, 4)), dim3(256), 0, h->stream, nr, h->diag_slot,
                       h->S, h->Linv, h->status, h->g, h->cg_r[0], h->cg_w[0], h->cg_s[0], h->cg_p, h->cg_xh,
                       h->poses, h->pose_of_rid, h->coarse_basis, h->Bmat, h->bgv);
    static const bool scale_pipe = !ps_env("PS_SCALE_NO_PIPE");
    if (scale_pipe)
        hipLaunchKernelGGL(k_scale_blocks_p<D>, dim3(cdiv(h->nnzb, 4 * PS_SCB_NB)), dim3(256), 0, h->stream, h->nnzb, h->col_idx,
                           h->brow_of, h->Linv, h->S, h->aug_slot, h->Saug, h->Bmat, h->SB);
    else
        hipLaunchKernelGGL(k_scale_blocks<D>, dim3(h->nnzb), dim3(64), 0, h->stream, nr, h->row_ptr, h->col_idx,
                           h->brow_of, h->Linv, h->S, h->aug_slot, h->Saug, h->Bmat, h->SB);
    // A_c^-1 lives in LciT2[b] (the transposed factor is not used on this path).  It only PRECONDITIONS here, so any
    // symmetric positive definite stand-in keeps the CG exact: whole-iteration calls use the inverse formed from the
    // PREVIOUS iteration's A_c and form + factor the current one on the side stream while the CG iterates (assembly,
    // factorisation, triangular inverse and product are 5.7 ms of the 12 ms iteration at C2 with 256 nodes; the assembly
    // alone -- k_xcoarse_rowsums + k_xcoarse_matrix -- 59 us of C4's iteration when it ran on the solver stream).
    // "coarse_refresh_every" = k > 1 (landmark-sharded runs on many GPUs, where an iteration is shorter than the side
    // stream's factorisation -- C4 on 8 GPUs: ~1.1 ms against 1.6 ms): only every k-th lagged set-up consumes the newest
    // inverse and starts the next factorisation; the set-ups in between HOLD the inverse they have, assemble no A_c and
    // do not wait for the side stream.  A fixed schedule, not a completion poll: results stay reproducible run to run.
    // (an inverse formed under another damping is a poor stand-in -- 93 CG iterations against 24 on a 600-keyframe BA when lambda
    // goes from 0 to 1e-3: such a call factors its own A_c, on the solver stream)
    // (a damping within a factor of four of the inverse's -- an LM schedule that halves or doubles lambda -- still lags: the
    //  coarse eigenvalues move by that factor at most, and the alternative is a factorisation on the solver stream)
    const bool sparse_rows = (long)h->nnzb <= 24L * nr;
    const double lam_inv = h->lci_next >= 0 ? h->xcg_tag_lambda[h->lci_next] : 0.0, lam_now = h->lin_lambda;
    const bool lam_ok = lam_now == lam_inv || (lam_now > 0.0 && lam_inv > 0.0 && lam_now <= 4.0 * lam_inv && lam_inv <= 4.0 * lam_now);
    // (round 5) ... and not behind a step that at least HALVED the cost on pose-graph-like rows when the factorisation is the cheap
    // partitioned one: the inverse of the start point's A_c is a poor stand-in on the other side of such a step -- C2's second call
    // took 111 CG iterations with it against 71-77 with a current one, 34 x 31 us -- while forming the current one on the solver
    // stream costs 0.33 + 0.06 ms there.  Bundle-adjustment rows lose nothing by lagging (C4: 20 iterations either way) and keep it.
    const bool big_drop = h->prev_cost > 0.0 && h->last_cost >= 0.0 && h->last_cost < 0.5 * h->prev_cost;
    const int bw_now = std::max(h->ac_bw, 1);
    const bool cheap_factor = h->band_chol && h->ac_bw >= 0 && h->band_part &&
                              BandPart::eligible(ncb, bw_now, h->band_part_m > 0 ? h->band_part_m : BandPart::auto_m(ncb, bw_now));
    const bool refactor_now = h->sync_refactor && sparse_rows && big_drop && cheap_factor;
    const bool lag = allow_lag && h->coarse_lag && h->lci_next >= 0 && lam_ok && !refactor_now;
    const bool settled = h->xcg_auto_hold && h->prev_cost > 0.0 && h->last_cost > 0.0 &&
                         std::fabs(h->prev_cost - h->last_cost) <= 1e-4 * h->prev_cost && h->xcg_held < 3;
    // ... or the inverse in use was formed from the A_c of THIS linearisation point (the caller linearises at the same point
    // again -- a damping retry, a repeated step: the same start cost): nothing to refresh, for as long as that lasts
    // (and the same damping: the inverse of an undamped A_c against a system damped with lambda = 1e-3 took 93 iterations where
    // the right one takes 24 -- a 600-keyframe BA, tests/test_gpu_ldi.py)
    const bool same_point = h->xcg_auto_hold && h->last_cost > 0.0 && h->last_cost == h->xcg_tag[h->lci_cur] &&
                            h->lin_lambda == h->xcg_tag_lambda[h->lci_cur];
    // ... or the inverse in use still does its job (round 4): a refresh costs the latency-bound CG launches beside the side
    // stream's factorisation ~5 us each (C4: 0.10-0.15 ms per iteration), a stale coarse inverse costs CG iterations (C4: the
    // inverse of the start point serves the whole 4-iteration solve at 20-21 iterations against 19-20) -- so it is kept
    // while the last solve with it took at most 3 iterations more than the first one did, for at most 8 set-ups in a row,
    // under the damping it was formed with
    // (not behind a step that changed the cost by more than 5 %: the point has moved, and a pose graph's coarse operator with it --
    //  10 000 poses: 122 CG iterations in the third call with the start point's inverse against 88 with the refreshed one)
    const bool moved = !(h->prev_cost > 0.0 && h->last_cost > 0.0 && std::fabs(h->prev_cost - h->last_cost) <= 0.05 * h->prev_cost);
    // (round 5: on bundle-adjustment rows the big first step does not spoil the inverse either -- C4's second call takes 20 iterations
    //  with the start point's inverse against 19-20 with a fresh one -- so there the iteration count alone decides; option
    //  "hold_across_steps" 0 restores round 4's rule)
    const bool moved_matters = moved && (sparse_rows || !h->hold_across_steps);
    const bool keep_ok = h->xcg_adaptive_hold && !moved_matters && h->xcg_its_ref > 0 && h->last_pcg_iters > 0 &&
                         h->last_pcg_iters <= h->xcg_its_ref + 3 && h->xcg_good_held < 8 && h->lci_next >= 0 &&
                         h->lin_lambda == h->xcg_tag_lambda[h->lci_next];
    const bool fresh_waiting = h->lci_next != h->lci_cur;    // a newer inverse has been (or is being) formed and not taken yet
    const bool still_good = keep_ok && !fresh_waiting;       // keep the inverse in use: nothing on the side stream
    const bool take_only = keep_ok && fresh_waiting;         // take the newer one, but do not start the next factorisation
    const bool hold = lag && ((h->xcg_lag_count > 0 &&
                               ((h->xcg_refresh_every > 1 && (h->xcg_lag_count % h->xcg_refresh_every) != 0) || settled)) || same_point ||
                              still_good);
    h->xcg_held = (hold && settled && !same_point) ? h->xcg_held + 1 : 0;
    h->xcg_good_held = ((hold && still_good && !same_point && !settled) || (!hold && take_only)) ? h->xcg_good_held + 1
                                                                                                  : (hold ? h->xcg_good_held : 0);
    h->xcg_setup_cost = h->last_cost; h->xcg_setup_lambda = h->lin_lambda;
    const int32_t* lagst = nullptr;
    h->xcg_side_todo = false;
    if (!hold && h->side_pending) { HIP_OK(hipStreamWaitEvent(h->stream, h->ev_chol, 0)); h->side_pending = false; }
    if (lag) {
        if (!hold) {
            if (h->lci_cur != h->lci_next) h->xcg_ref_pending = true;   // a new inverse: the solve below sets its reference iteration count
            h->lci_cur = h->lci_next;
            if (!take_only) {
                if (xcg_ac_on_main()) xcg_assemble_ac<D>(h, h->stream);
                HIP_OK(hipEventRecord(h->ev_ac, h->stream));   // SB and the basis complete; the side work is enqueued by xcg_side_enqueue
                h->xcg_side_todo = true;
            }
            lagst = h->lag_status;
        }
        ++h->xcg_lag_count;
    } else {
        xcg_assemble_ac<D>(h, h->stream);
        const int buf = h->lci_cur;
        if (xcg_coarse_inverse<D>(h, h->stream, buf, h->status)) return -1;
        h->lci_next = buf; h->xcg_tag[buf] = h->last_cost; h->xcg_tag_lambda[buf] = h->lin_lambda;
        h->xcg_lag_count = 0;
        h->xcg_ref_pending = true;
    }
    {   // PS_XCG_INV_SUM (measurement build): checksum of the coarse inverse this solve consumes
        static const bool inv_sum = ps_env("PS_XCG_INV_SUM") != nullptr;
        if (inv_sum) {
            if (!h->chk_sums) { HIP_OK(hipHostMalloc(&h->chk_sums, 3 * 64 * 8, hipHostMallocMapped)); std::memset(h->chk_sums, 0, 3 * 64 * 8); }
            ++h->chk_nsum;                                   // (this set-up's slot: chk_nsum - 1, also for its side job)
            if (h->chk_nsum <= 64)
                hipLaunchKernelGGL(k_bit_sum, dim3(64), dim3(256), 0, h->stream, (size_t)nc * nc, (const unsigned*)h->LciT2[h->lci_cur], h->chk_sums + 3 * (h->chk_nsum - 1));
        }
    }
    h->xf_active = h->xcg_fused && h->xf_ok && h->xf_skip == 0;
    h->xp_defer = false;
    {   // xstate, the second p buffer and (one- / two-launch form) ts_0 and the records of buffer 0: one launch
        const size_t n0 = 8, n1 = (size_t)nr * D, n2 = h->xf_active ? (size_t)nc : 0, n3 = h->xf_active ? h->xf_nrec : 0;
        hipLaunchKernelGGL(k_zero4, dim3((unsigned)std::min<size_t>(1024, cdiv((long)(n0 + n1 + n2 + n3), 256))), dim3(256), 0, h->stream,
                           n0, h->xstate, n1, h->xp2, n2, h->xf_active ? h->xf_ts[0] : nullptr, n3, h->xf_active ? h->xf_tq[0] : nullptr);
    }
    // one launch while the chip holds every workgroup at once (each forms its own rows of y: cheap for a narrow coarse level and
    // one round of workgroups); beyond that y is computed once, in a launch of its own -- 5 000 SE(3) poses: 5.95 ms with one
    // launch, 2.26 with three, 2.11 with two; 4 000-keyframe BA 2.26 / 1.79 / 1.69 (tools/xf_forms_probe.py)
    h->xf_two = h->xf_active && (!h->xf_one_ok || h->xcg_fused == 2 || h->xf_nwg > 256);
    if (h->xf_skip > 0) --h->xf_skip;
    if (h->xf_active) {
        // one-launch form: t_0 = P^T r_0 (k_xcg_restrict, initialisation mode) into buffer 0, then launch -1 of the fused
        // kernel (alpha = beta = 0): u_0 = M^-1 r_0, w_0 = S^ u_0, the partials of gamma_0 / delta_0 and the records of P^T w_0
        hipLaunchKernelGGL(k_xcg_restrict<D>, dim3(ncb), dim3(256), 0, h->stream, nr, ncb, h->slo, h->shi, h->pnode, h->pw0,
                           h->pw1, h->Bmat, h->cg_r[0], h->cg_r[0], h->cg_w[0], h->cg_p, h->cg_xh, h->cg_gd[1], 0, h->xstate, -1,
                           h->xf_t[0], h->status);
        if (lagst) hipLaunchKernelGGL(k_lag_status_check, dim3(1), dim3(64), 0, h->stream, lagst, h->status);
        h->cg_launched = -1;
        // one launch per SOLVE (ps_k_xcg_persist.h): launch -1 is the first pass of the launch that runs them all (xcg_launch)
        // (long rows only -- bundle adjustment: what the launch saves is the matrix stream of every iteration; with the short rows
        //  of a pose graph there is little to save and the exchange between up to 256 workgroups costs more: 1 500 poses 5.48 -> 5.94 ms)
        h->xp_defer = h->xcg_persist && h->xp_ok && !h->xf_two && h->xf_pf >= 6 && max_iters + 1 <= 4090 && !h->no_repeat &&
                      h->persist_reserve(h->xp_cus_needed);  // (its grid must be resident at once: ps_core.hip, PersistLedger)
        h->cg_max_launches = max_iters + 1;
        if (!h->xp_defer) xcg_launch<D>(h, 0.0, 1);        // launch -1 (cg_launched: -1 -> 0)
        ++h->xf_solves;
        return 0;
    }
    // z_0 = M^-1 r_0 and r_0 . z_0
    hipLaunchKernelGGL(k_xcg_restrict<D>, dim3(ncb), dim3(256), 0, h->stream, nr, ncb, h->slo, h->shi, h->pnode, h->pw0,
                       h->pw1, h->Bmat, h->cg_r[0], h->cg_r[0], h->cg_w[0], h->cg_p, h->cg_xh, h->cg_gd[1], 0, h->xstate, -1,
                       h->tvec, h->status);
    hipLaunchKernelGGL(k_xcg_coarse, dim3(cdiv(nc, 4)), dim3(256), 0, h->stream, nc, (const float*)h->LciT2[h->lci_cur], h->tvec, h->xy, h->status, lagst);
    hipLaunchKernelGGL(k_xcg_prolong<D>, dim3(cdiv(nr, PS_XCG_DROWS)), dim3(PS_XCG_DROWS), 0, h->stream, nr, ncb, h->pnode,
                       h->pw0, h->pw1, h->Bmat, h->cg_r[0], h->xy, h->cg_s[0], h->cg_gd[0], h->status);
    h->cg_launched = 0;
    return 0;
}

// the side-stream half of a lagged setup, enqueued AFTER the first chunk of CG launches so that its ~130 launches
// do not sit in front of them on the host
template <int D>
int xcg_side_enqueue(ps_problem* h) {
    if (!h->xcg_side_todo) return 0;
    h->xcg_side_todo = false;
    const int nc = h->nc, nb = h->lci_cur ^ 1;
    HIP_OK(hipStreamWaitEvent(h->side, h->ev_ac, 0));
    {   // PS_SIDE_DELAY=<rounds> (measurement build, tools/probes/lowprio_hunt.sh): a slow one-workgroup kernel in front of the side job
        // of an ORDINARY side stream -- the job then overlaps the solver stream's later work the way a low-priority one does.
        // Differences with it = a race of this protocol that timing exposes; none = the lowest-priority queue itself
        static const int side_delay = ps_env("PS_SIDE_DELAY") ? atoi(ps_env("PS_SIDE_DELAY")) : 0;
        if (side_delay > 0) {
            if (!h->xy) return fail("PS_SIDE_DELAY: no scratch");
            if (ensure_dynamic_lds((const void*)k_lds_scribble, 96 * 1024)) return -1;
            hipLaunchKernelGGL(k_lds_scribble, dim3(1), dim3(512), 96 * 1024, h->side, 96 * 1024 / 8, side_delay, h->chol_scratch);
        }
    }
    if (!xcg_ac_on_main()) xcg_assemble_ac<D>(h, h->side);
    // (measurement switches of the low-priority hunt: PS_XCG_ACDONE_LATE=1 records ev_acdone at the END of the side job instead of
    //  between assembly and factorisation -- the solver stream then waits for the whole job before it rewrites SB / the basis, which is
    //  safe; PS_XCG_SIDE_PAD=1 puts a one-workgroup dummy kernel between the event and the factorisation)
    static const bool acdone_late = ps_env("PS_XCG_ACDONE_LATE") != nullptr, side_pad = ps_env("PS_XCG_SIDE_PAD") != nullptr;
    if (!acdone_late) HIP_OK(hipEventRecord(h->ev_acdone, h->side));
    if (side_pad) copy_doubles(h->side, h->xy, h->xy, 1);
    h->acdone_pending = true;
    // (measurement switches of the low-priority hunt, tools/probes/lowprio_hunt.sh: PS_XCG_AC_WAIT=1 the solver stream waits for the
    //  side stream's assembly at once; =2 it waits for the whole side job)
    static const int ac_wait = ps_env("PS_XCG_AC_WAIT") ? atoi(ps_env("PS_XCG_AC_WAIT")) : 0;
    if (ac_wait == 1) HIP_OK(hipStreamWaitEvent(h->stream, h->ev_acdone, 0));
    static const bool ac_check = ps_env("PS_XCG_AC_CHECK") != nullptr;
    if (ac_check && !xcg_ac_on_main()) {
        // the same assembly once more on the SOLVER stream (after the side stream's), into buffers of its own, and a bitwise comparison
        const size_t nA = (size_t)h->nc * h->nc;
        if (!h->chk_cnt) {
            int32_t n_ent = 0;
            HIP_OK(hipMemcpy(&n_ent, h->ent_ptr + h->nr, sizeof(n_ent), hipMemcpyDeviceToHost));
            h->chk_bsz_n = (size_t)n_ent * D * D;
            HIP_OK(hipMalloc(&h->chk_Ac, nA * 8)); HIP_OK(hipMalloc(&h->chk_BSZ, std::max<size_t>(h->chk_bsz_n, 1) * 8)); HIP_OK(hipMalloc(&h->chk_cnt, 16));
            HIP_OK(hipMalloc(&h->chk_Ac_side, nA * 8));
            HIP_OK(hipMemset(h->chk_cnt, 0, 16));
        }
        // ... and what the SIDE stream itself sees of A_c right behind its assembly (a copy kernel in front of the factorisation),
        // compared at the end of the side job with A_c as it is then: slot 3
        copy_doubles(h->side, h->chk_Ac_side, h->Ac, nA);
        HIP_OK(hipStreamWaitEvent(h->stream, h->ev_acdone, 0));
        hipLaunchKernelGGL(k_xcoarse_rowsums<D>, dim3(h->nr), dim3(256), (size_t)(h->max_row_ents + 1) * D * D * sizeof(double), h->stream,
                           h->nr, h->ent_ptr, h->ent_q, h->ent_lo, h->ent_hi, h->acol_idx, h->pnode, h->pw0, h->pw1, h->SB, h->Bmat, h->chk_BSZ);
        hipLaunchKernelGGL(k_xcoarse_matrix<D>, dim3(cdiv((long)h->ncb * h->ncb * D * D, 256)), dim3(256), 0, h->stream,
                           h->ncb, h->seg_ptr, h->seg_ent, h->seg_row, h->pnode, h->pw0, h->pw1, h->chk_BSZ, h->chk_Ac);
        hipLaunchKernelGGL(k_cmp_bits, dim3(64), dim3(256), 0, h->stream, nA, (const double*)h->Ac, (const double*)h->chk_Ac, h->chk_cnt, 0);
        hipLaunchKernelGGL(k_cmp_bits, dim3(64), dim3(256), 0, h->stream, h->chk_bsz_n, (const double*)h->BSZ, (const double*)h->chk_BSZ, h->chk_cnt, 1);
    }
    if (xcg_coarse_inverse<D>(h, h->side, nb, h->lag_status)) return -1;
    if (h->chk_sums && h->chk_nsum >= 1 && h->chk_nsum <= 64) {      // PS_XCG_INV_SUM: what the side job factored (A_c as it is at its end) and what it produced
        hipLaunchKernelGGL(k_bit_sum, dim3(64), dim3(256), 0, h->side, (size_t)2 * h->nc * h->nc, (const unsigned*)h->Ac, h->chk_sums + 3 * (h->chk_nsum - 1) + 1);
        hipLaunchKernelGGL(k_bit_sum, dim3(64), dim3(256), 0, h->side, (size_t)h->nc * h->nc, (const unsigned*)h->LciT2[nb], h->chk_sums + 3 * (h->chk_nsum - 1) + 2);
    }
    if (ac_check && h->chk_Ac_side)
        hipLaunchKernelGGL(k_cmp_bits, dim3(64), dim3(256), 0, h->side, (size_t)h->nc * h->nc, (const double*)h->Ac, (const double*)h->chk_Ac_side, h->chk_cnt, 3);
    if (acdone_late) HIP_OK(hipEventRecord(h->ev_acdone, h->side));
    HIP_OK(hipEventRecord(h->ev_chol, h->side));
    if (ac_wait == 2) HIP_OK(hipStreamWaitEvent(h->stream, h->ev_chol, 0));
    h->lci_next = nb; h->side_pending = true; h->xcg_tag[nb] = h->xcg_setup_cost; h->xcg_tag_lambda[nb] = h->xcg_setup_lambda;
    return 0;
}

template <int D>
void xcg_launch(ps_problem* h, double tol, int count) {
    const int nr = h->nr, ncb = h->ncb, nc = h->nc;
    const int n_rz = cdiv(nr, PS_XCG_DROWS);
    double* pbuf[2] = {h->cg_p, h->xp2};
    if (h->xf_active && h->xp_defer && h->cg_launched >= h->cg_max_launches) return;      // (the one launch has run them all)
    if (h->xf_active && h->xp_defer && h->cg_launched == -1) {
        // every iteration of the solve in ONE launch: k = -1 .. max_iters - 1 at most, stops at convergence by itself
        if (count <= 0) return;
        XcgFusedArgs a{};
        a.cptr = h->xf_cptr; a.cols = h->xf_cols; a.lidx = h->xf_lidx; a.nlo = h->xf_nlo; a.nhi = h->xf_nhi;
        a.Ainv = (const float*)h->LciT2[h->lci_cur]; a.nc = nc; a.ncb = ncb;
        a.pnode = h->pnode; a.pw0 = h->pw0; a.pw1 = h->pw1; a.Bmat = h->Bmat;
        a.rec_out = h->xf_rec; a.rmax = h->xf_rmax; a.nwg = h->xf_nwg;
        a.t_in = h->xf_t[0]; a.ts_in = h->xf_ts[0];
        a.r_in = h->cg_r[0]; a.w_in = h->cg_w[0]; a.s_in = h->cg_s[0];
        a.u = h->xp2; a.p = h->cg_p; a.x = h->cg_xh;
        if (++h->xp_salt >= (1u << 20)) {
            hipMemsetAsync(h->xp_exch, 0, h->xp_words * sizeof(unsigned long long), h->stream);
            h->xp_salt = 1;
        }
        const int nl = h->cg_max_launches + 1;               // passes: k = -1 .. max_iters - 1
        if (ps_env("PS_XP_CLOCKS") && !h->xp_dbg) {                                                                     // (measurement build only)
            hipMalloc(&h->xp_dbg, 3 * 64 * 8 * 8); hipMemset(h->xp_dbg, 0, 3 * 64 * 8 * 8);
            fprintf(stderr, "k_xcg_persist: nr %d, nc %d, ncb %d, ell width %d, records %zu, workgroups %d, pf %d\n", nr, nc, h->ncb, h->ell_wf, (size_t)h->xf_nrec, h->xf_nwg, h->xf_pf);
        }
        // (the matrix: PF blocks per lane in registers, PL in LDS behind t and the records; a row wider than 8 (PF + PL) blocks reads
        //  the rest from L2 in every iteration)
        const bool ne2 = nc <= 2 * 64 * PS_XF_ROWS;
        const size_t lds0 = ((size_t)((nc + 1) & ~1) + (((size_t)h->xf_nrec + 1) & ~(size_t)1)) * sizeof(double);
#define PS_XP_LAUNCH(PF, PL, NE) do {                                                                                                         \
            const size_t lds = lds0 + (size_t)(PL) * 64 * PS_XF_ROWS * (D * sizeof(double) + sizeof(int32_t)) + ovf_bytes;                     \
            if (ensure_dynamic_lds((const void*)k_xcg_persist<D, PF, PL, NE>, lds)) { launched = false; break; }                              \
            int per_cu_ = 0;                                                                                                                   \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_, (const void*)k_xcg_persist<D, PF, PL, NE>, 64 * PS_XF_ROWS, lds) != hipSuccess || \
                per_cu_ < 1 || cdiv(h->xf_nwg, per_cu_) > h->persist_capacity()) { launched = false; break; }                                 \
            hipLaunchKernelGGL((k_xcg_persist<D, PF, PL, NE>), dim3(h->xf_nwg), dim3(64 * PS_XF_ROWS), lds, h->stream, nr, h->arow_ptr,        \
                               h->ell_wf, h->Saug, a, h->xf_cnt, nl, tol * tol, h->hist, h->hist_cap, h->status, h->scalars, h->xstate,        \
                               h->xp_exch, h->xp_salt, h->cp_spin, h->xp_dbg); ++h->xp_dbg_launches; } while (0)
        // (LDS: the static arrays + t + the records + PL blocks per lane of the matrix: as many as fit 160 KB)
        const size_t per_pl = (size_t)64 * PS_XF_ROWS * (D * sizeof(double) + sizeof(int32_t));
        const size_t fixed = 11700 + 1024;                   // (the kernel's static arrays, -Rpass-analysis=kernel-resource-usage)
        const size_t ovf_bytes = (size_t)PS_XF_ROWS * 8 * (D * sizeof(double) + sizeof(int32_t));      // (the kk = 0 lanes' one block more)
        const size_t room = 160 * 1024 - fixed - ovf_bytes - std::min<size_t>(lds0, 120 * 1024);
        const int pl = h->xf_pf == 8 ? (room >= 4 * per_pl ? 4 : (room >= 2 * per_pl ? 2 : 0)) : 0;
        bool launched = true, four_done = false;
        // four waves per workgroup, the rows of A_c^-1 and seven blocks per lane and row in registers (ps_k_xcg_persist4.h): BA-like
        // rows (pf 8) whose coarse level fits its register arrays
        if constexpr (D == 6) {
            constexpr int X4_NYW = 11, X4_NQ = 5, PF4 = 6, PL4 = 4;
            const size_t lds4 = lds0 + (size_t)PL4 * PS_X4_RPW * PS_X4_NT * (D * sizeof(double) + sizeof(int32_t));
            if (h->xcg_persist4 && h->xf_pf == 8 && (nc & 1) == 0 && nc <= 128 * X4_NQ && nc <= 3 * PS_X4_NT && h->xf_ymax <= PS_X4_NW * X4_NYW &&
                h->xf_nrec <= (size_t)PS_X4_NR * PS_X4_NT && lds4 + fixed <= 160 * 1024) {
                auto k4 = k_xcg_persist4<6, PF4, PL4, 3, X4_NYW, X4_NQ>;
                int per_cu_ = 0;
                if (!ensure_dynamic_lds((const void*)k4, lds4) &&
                    hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_, (const void*)k4, PS_X4_NT, lds4) == hipSuccess && per_cu_ >= 1 &&
                    cdiv(h->xf_nwg, per_cu_) <= h->persist_capacity()) {
                    hipLaunchKernelGGL(k4, dim3(h->xf_nwg), dim3(PS_X4_NT), lds4, h->stream, nr, h->arow_ptr, h->ell_wf, h->Saug, a, h->xf_cnt,
                                       nl, tol * tol, h->hist, h->hist_cap, h->status, h->scalars, h->xstate, h->xp_exch, h->xp_salt, h->cp_spin, h->xp_dbg);
                    ++h->xp_dbg_launches; ++h->xp4_launches;
                    four_done = true;
                } else (void)hipGetLastError();
            }
        }
        if (four_done) { }
        else if (h->xf_pf == 2) { if (ne2) PS_XP_LAUNCH(2, 0, 2); else PS_XP_LAUNCH(2, 0, 4); }
        else if (pl == 4) { if (ne2) PS_XP_LAUNCH(6, 4, 2); else PS_XP_LAUNCH(6, 4, 4); }
        else if (pl == 2) { if (ne2) PS_XP_LAUNCH(6, 2, 2); else PS_XP_LAUNCH(6, 2, 4); }
        else { if (ne2) PS_XP_LAUNCH(6, 0, 2); else PS_XP_LAUNCH(6, 0, 4); }
        if (!launched) {                                     // (cannot be configured or not resident: the launch-per-iteration form from launch -1 on)
            (void)hipGetLastError();
            h->persist_release(); ++h->cp_refused;
            h->xp_defer = false;
            xcg_launch<D>(h, tol, count);
            return;
        }
#undef PS_XP_LAUNCH
        h->cg_launched = h->cg_max_launches; h->cg_kernel_launches += 1; ++h->xp_launches; ++h->cp_launches;
        return;
    }
    if (h->xf_active) {                                     // ONE launch per iteration (launch index n = k + 1 picks the buffers)
        h->cg_kernel_launches += (h->xf_two ? 2 : 1) * count;
        const size_t lds = h->xf_two ? 0 : (size_t)nc * sizeof(double);
        for (int i = 0; i < count; ++i, ++h->cg_launched) {
            const int k = h->cg_launched, in = (k + 1) & 1, out = in ^ 1;
            XcgFusedArgs a{};
            a.cptr = h->xf_cptr; a.cols = h->xf_cols; a.lidx = h->xf_lidx; a.nlo = h->xf_nlo; a.nhi = h->xf_nhi;
            a.Ainv = (const float*)h->LciT2[h->lci_cur]; a.nc = nc; a.ncb = ncb;
            a.pnode = h->pnode; a.pw0 = h->pw0; a.pw1 = h->pw1; a.Bmat = h->Bmat;
            a.rec_out = h->xf_rec; a.rmax = h->xf_rmax;
            a.tq_in = h->xf_tq[in]; a.tq_out = h->xf_tq[out];
            a.t_in = h->xf_t[in]; a.t_out = h->xf_t[out]; a.ts_in = h->xf_ts[in]; a.ts_out = h->xf_ts[out];
            a.gd_in = h->cg_gd[in]; a.gd_out = h->cg_gd[out]; a.nwg = h->xf_nwg;
            a.r_in = h->cg_r[in]; a.r_out = h->cg_r[out]; a.w_in = h->cg_w[in]; a.w_out = h->cg_w[out];
            a.s_in = h->cg_s[in]; a.s_out = h->cg_s[out];
            a.u = h->xp2; a.p = h->cg_p; a.x = h->cg_xh; a.y = h->xy;
#define PS_XF_LAUNCH(PF, TWO) hipLaunchKernelGGL((k_xcg_fused1<D, PF, TWO>), dim3(h->xf_nwg), dim3(64 * PS_XF_ROWS), lds, h->stream, nr, h->arow_ptr, \
                               h->ell_wf, h->Saug, a, k, tol * tol, h->hist, h->hist_cap, h->status, h->scalars, h->xstate)
            if (h->xf_two) {
                // (one workgroup per compute unit, all in one round: ceil(nc / 256) rows each, at most one per wave)
                static const int f2_rows_env = ps_env("PS_F2_ROWS") ? atoi(ps_env("PS_F2_ROWS")) : 0, f2_ablate = ps_env("PS_F2_ABLATE") ? atoi(ps_env("PS_F2_ABLATE")) : 0;
                const int f2_rows = f2_rows_env > 0 ? f2_rows_env : std::min(PS_XCG_CROWS_BIG, std::max(1, cdiv(nc, 256)));
                if (nc <= 3 * 64 * PS_XCG_CROWS_BIG)
                    hipLaunchKernelGGL((k_xcg_f2_coarse<D, 3>), dim3(cdiv(nc, f2_rows)), dim3(64 * PS_XCG_CROWS_BIG), (size_t)nc * sizeof(double),
                                       h->stream, a, k, tol * tol, h->hist, h->hist_cap, h->status, h->scalars, h->xstate, f2_rows, f2_ablate);
                else
                    hipLaunchKernelGGL((k_xcg_f2_coarse<D, PS_XF2_NEMAX>), dim3(cdiv(nc, f2_rows)), dim3(64 * PS_XCG_CROWS_BIG), (size_t)nc * sizeof(double),
                                       h->stream, a, k, tol * tol, h->hist, h->hist_cap, h->status, h->scalars, h->xstate, f2_rows, f2_ablate);
                static const int pf_env = ps_env("PS_XF2_PF") ? atoi(ps_env("PS_XF2_PF")) : -1;
                // more workgroups than the chip holds at once: the kernel's time is (rounds of workgroups) x (its dependent
                // phases), so registers go to occupancy, not to prefetch (C2, 1 250 workgroups: 24.4 us with PF = 6)
                const int pf = pf_env >= 0 ? pf_env : (h->xf_nwg > 512 ? 0 : h->xf_pf);
                if (pf == 0) PS_XF_LAUNCH(0, true); else if (pf == 2) PS_XF_LAUNCH(2, true); else if (pf == 6) PS_XF_LAUNCH(6, true); else PS_XF_LAUNCH(8, true);
            } else if (h->xf_pf == 2) PS_XF_LAUNCH(2, false); else if (h->xf_pf == 6) PS_XF_LAUNCH(6, false); else PS_XF_LAUNCH(8, false);
#undef PS_XF_LAUNCH
        }
        return;
    }
    if (h->xcg_rt && h->xcg_rt_ok) {                        // three launches per iteration
        const int R = h->xcg_rt_rows, n_pq = cdiv(nr, R);
        h->cg_kernel_launches += 3L * count;
        const XcgRestrictArgs ra{h->Bmat, h->pnode, h->pw0, h->pw1, h->xcg_wg_out, h->tq_part};
        double* tb[2] = {h->tvec, h->tvec2};
        for (int i = 0; i < count; ++i, ++h->cg_launched) {
            const int k = h->cg_launched, b = k & 1;
#define PS_XCG_SPMV_RT(RR) hipLaunchKernelGGL((k_xcg_spmv<D, RR, true>), dim3(n_pq), dim3(64 * RR), 0, h->stream, nr,             \
                               h->arow_ptr, h->acol_idx, h->ell_wf, h->Saug, h->cg_s[0], pbuf[b ^ 1], pbuf[b], h->cg_w[0],       \
                               h->cg_gd[0], n_rz, h->cg_gd[1], h->xstate, k, tol * tol, h->hist, h->status, h->scalars, ra)
            if (R == 4) PS_XCG_SPMV_RT(4); else if (R == 8) PS_XCG_SPMV_RT(8); else PS_XCG_SPMV_RT(16);
#undef PS_XCG_SPMV_RT
            if (ncb <= 256)
                hipLaunchKernelGGL(k_xcg_coarse_rt<D>, dim3(cdiv(nc, PS_XCG_CROWS)), dim3(64 * PS_XCG_CROWS), 0, h->stream, nc,
                                   (const float*)h->LciT2[h->lci_cur], tb[b], tb[b ^ 1], h->xcg_nptr, h->tq_part, h->cg_gd[1], n_pq,
                                   h->xstate, k, h->xy, h->status);
            else
                hipLaunchKernelGGL(k_xcg_coarse_rt_big<D>, dim3(cdiv(nc, PS_XCG_CROWS_BIG)), dim3(64 * PS_XCG_CROWS_BIG),
                                   (size_t)nc * sizeof(double), h->stream, nc, (const float*)h->LciT2[h->lci_cur], tb[b], tb[b ^ 1],
                                   h->xcg_nptr, h->tq_part, h->cg_gd[1], n_pq, h->xstate, k, h->xy, h->status);
            hipLaunchKernelGGL(k_xcg_prolong_rt<D>, dim3(n_rz), dim3(PS_XCG_DROWS), 0, h->stream, nr, ncb, h->pnode, h->pw0,
                               h->pw1, h->Bmat, h->cg_r[b], h->cg_r[b ^ 1], h->cg_w[0], pbuf[b], h->cg_xh, h->cg_gd[1], n_pq,
                               h->xstate, k, h->xy, h->cg_s[0], h->cg_gd[0], h->status);
        }
        return;
    }
    const int n_pq = cdiv(nr, PS_XCG_ROWS);
    h->cg_kernel_launches += 4L * count;
    for (int i = 0; i < count; ++i, ++h->cg_launched) {
        const int k = h->cg_launched, b = k & 1;
        hipLaunchKernelGGL((k_xcg_spmv<D, PS_XCG_ROWS, false>), dim3(n_pq), dim3(64 * PS_XCG_ROWS), 0, h->stream, nr, h->arow_ptr, h->acol_idx,
                           h->ell_wf, h->Saug, h->cg_s[0], pbuf[b ^ 1], pbuf[b], h->cg_w[0], h->cg_gd[0], n_rz, h->cg_gd[1],
                           h->xstate, k, tol * tol, h->hist, h->status, h->scalars, XcgRestrictArgs{});
        hipLaunchKernelGGL(k_xcg_restrict<D>, dim3(ncb), dim3(256), 0, h->stream, nr, ncb, h->slo, h->shi, h->pnode,
                           h->pw0, h->pw1, h->Bmat, h->cg_r[b], h->cg_r[b ^ 1], h->cg_w[0], pbuf[b], h->cg_xh, h->cg_gd[1],
                           n_pq, h->xstate, k, h->tvec, h->status);
        hipLaunchKernelGGL(k_xcg_coarse, dim3(cdiv(nc, 4)), dim3(256), 0, h->stream, nc, (const float*)h->LciT2[h->lci_cur], h->tvec, h->xy, h->status, (const int32_t*)nullptr);
        hipLaunchKernelGGL(k_xcg_prolong<D>, dim3(cdiv(nr, PS_XCG_DROWS)), dim3(PS_XCG_DROWS), 0, h->stream, nr, ncb,
                           h->pnode, h->pw0, h->pw1, h->Bmat, h->cg_r[b ^ 1], h->xy, h->cg_s[0], h->cg_gd[0], h->status);
    }
}

// synchronous solve: poll the convergence flag every chunk, then x = Linv^T x^
template <int D>
int xcg_run(ps_problem* h, double tol, int max_iters, int* iters_out, double* relres_out, bool allow_lag = false) {
    if (xcg_setup<D>(h, max_iters, allow_lag)) return -1;
    int chunk = std::max(32, h->last_pcg_iters + 2);
    bool done = false;
    while (!done) {
        const int m = std::min(chunk, max_iters + 1 - h->cg_launched);
        xcg_launch<D>(h, tol, m);
        if (xcg_side_enqueue<D>(h)) return -1;
        if (read_scalars(h)) return -1;                     // (synchronises: sync() releases the ledger)
        if (h->h_status[ST_PERSIST_FAIL] && h->xcg_persist) { h->xcg_persist = 0; ++h->cp_failures; }
        if (h->xf_active && h->h_status[ST_PCG_DONE] == 2 && !h->h_status[ST_DIAG_FAIL] && !h->h_status[ST_LM_FAIL]) {
            ++h->xf_fallbacks; h->xf_skip = 1;              // breakdown of the one-launch form: again, three launches per iteration
            if (xcg_setup<D>(h, max_iters, false)) return -1;
            chunk = std::max(32, h->last_pcg_iters + 2);
            continue;
        }
        done = h->h_status[ST_PCG_DONE] != 0 || h->cg_launched >= max_iters + 1;
        chunk = std::max(32, h->cg_launched / 4);
    }
    hipLaunchKernelGGL(k_cg_unscale<D>, dim3(cdiv((long)h->nr * D, 256)), dim3(256), 0, h->stream, h->nr, h->Linv,
                       h->cg_xh, h->x, (const int32_t*)nullptr);
    return cg_report(h, iters_out, relres_out);
}

bool use_direct(const ps_problem* h) {
    return h->pcg_variant == 1 && h->nr > 0 && h->nr * h->D <= h->direct_max;
}

// small reduced systems: dense blocked Cholesky in LDS instead of CG (enqueue only)
template <int D>
int direct_solve_enqueue(ps_problem* h) {
    const int nr = h->nr, n = nr * D;
    if (h->direct_fused) {                                  // one launch: no dense copy, no inverse (k_direct_solve)
        const size_t lds = ((size_t)n * n + 2 * (size_t)n + (size_t)nr * D * D) * sizeof(double);
        if (ensure_dynamic_lds((const void*)k_direct_solve<D>, lds)) return -1;
        // (1 024 threads: the trailing update has (nr - J - 1)^2 D^2 / 2 entries per step; measured 256 / 512 / 1 024 threads:
        //  17.8 / 16.7 / 16.6 us at 30 unknowns, 70 / 58 / 50 us at 90)
        static const int nt_env = ps_env("PS_DIRECT_THREADS") ? atoi(ps_env("PS_DIRECT_THREADS")) : 0;
        const int nthreads = nt_env > 0 ? nt_env : 1024;
        hipLaunchKernelGGL(k_direct_solve<D>, dim3(1), dim3(nthreads), lds, h->stream, nr, h->nnzb, h->brow_of, h->col_idx, h->S, h->g, h->x,
                           h->status, h->scalars);
        h->cg_launched = 0;
        return 0;
    }
    if (!h->dA && (h->alloc(&h->dA, (size_t)n * n) || h->alloc(&h->dLi, (size_t)n * n) || h->alloc(&h->dLiT, (size_t)n * n)))
        return -1;
    hipLaunchKernelGGL(k_bsr_to_dense<D>, dim3(1), dim3(256), 0, h->stream, nr, h->nnzb, h->brow_of, h->col_idx, h->S, h->dA);
    const size_t chol_lds = 2 * (size_t)n * n * sizeof(double);
    if (ensure_dynamic_lds((const void*)k_coarse_chol<D, true>, (size_t)(chol_lds))) return -1;
    hipLaunchKernelGGL((k_coarse_chol<D, true>), dim3(1), dim3(1024), chol_lds, h->stream, nr, h->dA, h->dLi, h->dLiT,
                       h->status, nullptr);
    hipLaunchKernelGGL(k_direct_apply<D>, dim3(1), dim3(256), 0, h->stream, n, h->dLi, h->dLiT, h->g, h->x, h->status,
                       h->scalars);
    h->cg_launched = 0;
    return 0;
}

int solve_reduced(ps_problem* h, double tol, int max_iters, int* iters, double* relres) {
    if (h->nr == 0) { if (iters) *iters = 0; if (relres) *relres = 0.0; return 0; }
    StageTimer t(h, PS_ST_PCG);
    if (use_direct(h)) {
        if (h->D == 6 ? direct_solve_enqueue<6>(h) : direct_solve_enqueue<3>(h)) return -1;
        if (read_scalars(h)) return -1;
        return cg_report(h, iters, relres);
    }
    if (h->pcg_variant == 1) {
        if (!h->coarse_built && build_coarse(h)) return -1;
        if (h->cg_explicit)
            return h->D == 6 ? xcg_run<6>(h, tol, max_iters, iters, relres) : xcg_run<3>(h, tol, max_iters, iters, relres);
        return h->D == 6 ? cg_fused_run<6>(h, tol, max_iters, iters, relres) : cg_fused_run<3>(h, tol, max_iters, iters, relres);
    }
    return h->D == 6 ? pcg_run<6>(h, tol, max_iters, iters, relres) : pcg_run<3>(h, tol, max_iters, iters, relres);
}

}  // namespace

namespace {
// ONE-synchronisation iteration for the fused CG: the CG launches (as many as the previous solve
// needed, plus a margin), the recovery of x and the whole tail are enqueued back to back; the tail
// kernels are gated on the device-side convergence flag, so if the CG needed more launches than
// predicted the host simply enqueues more and repeats the (until then no-op) tail.
template <int D>
int gn_solve_and_finish_async(ps_problem* h, double tol, int max_iters, int linesearch,
                              int* iters_out, double* relres_out, StageTimer* total) {
    StageTimer tp(h, PS_ST_PCG);
    if (use_direct(h)) {                                    // small system: three launches, then the (ungated) tail
        if (direct_solve_enqueue<D>(h)) return -1;
        tp.stop();
        if (gn_tail(h, linesearch, nullptr, true)) return -1;
        if (total) total->stop();
        if (wait_published(h)) return -1;
        return cg_report(h, iters_out, relres_out);
    }
    if (!h->coarse_built && build_coarse(h)) return -1;
    if (h->cg_explicit) {                                   // explicit two-level PCG, same one-synchronisation protocol
        if (xcg_setup<D>(h, max_iters, true)) return -1;
        // (+ 3 spare launches behind a step that changed the cost by more than 5 %: the lagged coarse inverse is from the other side of it)
        const bool big_step_x = h->prev_cost > 0.0 && h->last_cost > 0.0 && std::fabs(h->prev_cost - h->last_cost) > 0.05 * h->prev_cost;
        int count = h->last_pcg_iters > 0 ? h->last_pcg_iters + 2 + (big_step_x ? 3 : 0) : 32;
        // The call behind a big step ran with a coarse inverse from the other side of it and took far more iterations than its
        // neighbours (10 000-pose graph, cold: 71 / 111 / 77 / 75 / 72; 1 500 poses 70 / 122 / 87 / 86; BA barely: 19 / 20 / 20):
        // predicting the next call from THAT count enqueued ~36 iterations past convergence, and a launch that finds the solve done
        // still costs its prefetch (12 us each at that size: 0.9 ms of the third call).  Behind such a spike the call before it is the
        // better guide -- the third call needs up to a quarter more than the first across 600 .. 10 000 poses -- and a short second
        // round costs one host round trip (~40 us) where an overshoot costs 25 us per iteration.
        const bool spike = h->prev_pcg_iters > 0 && h->last_pcg_iters > h->prev_pcg_iters + 16;
        if (spike) count = h->prev_pcg_iters + h->prev_pcg_iters / 4 + 2;
        for (;;) {
            count = std::min(count, max_iters + 1 - h->cg_launched);
            // (the side-stream factorisation goes in after the first few iterations' launches: early enough to
            // finish beside the CG, late enough not to delay its start on the host)
            const int head = std::min(count, 12);
            {
                StageTimer tk(h, PS_ST_CG_KERNEL, 1);       // the CG launches alone (the side work goes to another stream)
                xcg_launch<D>(h, tol, head);
                if (xcg_side_enqueue<D>(h)) return -1;
                xcg_launch<D>(h, tol, count - head);
            }
            hipLaunchKernelGGL(k_cg_unscale<D>, dim3(cdiv((long)h->nr * D, 256)), dim3(256), 0, h->stream, h->nr, h->Linv,
                               h->cg_xh, h->x, (const int32_t*)h->status);
            tp.stop();
            if (gn_tail(h, linesearch, h->status, true)) return -1;
            if (total) total->stop();
            if (wait_published(h)) return -1;
            if (h->xf_active && !h->h_status[ST_DIAG_FAIL] && !h->h_status[ST_LM_FAIL] &&
                (h->h_status[ST_PCG_DONE] == 2 || (h->h_status[ST_PCG_DONE] == 0 && h->last_pcg_iters > 0 && h->cg_launched > 2 * h->last_pcg_iters + 16))) {
                // the single-reduction recurrences of the one-launch form broke down (or stalled: far more iterations than
                // the last solve needed) -- nothing has been applied: the same solve again with the three-launch form
                ++h->xf_fallbacks;
                h->xf_skip = 1;
                if (xcg_setup<D>(h, max_iters, false)) return -1;
                count = h->last_pcg_iters > 0 ? h->last_pcg_iters + 2 : 32;
                continue;
            }
            if (h->h_status[ST_PCG_DONE] != 0) break;
            if (h->cg_launched >= max_iters + 1) {          // not converged within max_iters: take the step anyway
                hipLaunchKernelGGL(k_cg_unscale<D>, dim3(cdiv((long)h->nr * D, 256)), dim3(256), 0, h->stream, h->nr, h->Linv,
                                   h->cg_xh, h->x, (const int32_t*)nullptr);
                if (gn_tail(h, linesearch, nullptr, true) || wait_published(h)) return -1;
                break;
            }
            count = std::max(8, h->cg_launched / (spike ? 8 : 2));
        }
        if (cg_report(h, iters_out, relres_out)) return -1;
        if (h->xcg_ref_pending) { h->xcg_ref_pending = false; h->xcg_its_ref = h->last_pcg_iters; }
        return 0;
    }
    ++h->ldi_iter;
    h->ldi_call_start_cost = h->last_cost;                  // cost at this call's linearisation point (-1: unknown)
    if (ldi_decide(h)) {                                    // a lagged dense inverse of S is at hand: 2-6 two-launch iterations
        const int rc = ldi_solve_and_finish<D>(h, tol, max_iters, linesearch, iters_out, relres_out, &tp, total);
        if (rc <= 0) return rc;
        // gave up (nothing applied): the standard solve below, which also re-seeds the inverse
    }
    if (cg_fused_setup<D>(h, max_iters, true)) return -1;
    // launches needed = iterations + 2 (the k = -1 launch and the one that detects convergence): exactly that when the
    // last two solves took the same number of iterations (0.360 -> 0.348 ms at C3), the configured margin otherwise --
    // one iteration too few costs a host round trip (~40 us), one launch too many ~1.5 us
    // (after a cost jump that left the lagged inverse behind the last standard solve is several calls old and from another
    //  phase: C3's trajectory needs 21 iterations where it needed 18 -- six spare launches at ~1.5 us each instead of a miss)
    // (the same after a step that changed the cost by more than 5 %: the lagged basis this set-up uses is from the other side of
    //  it -- the second call of a cold C3 solve needs 21 iterations where the first needed 18, and 18 + 4 launches were one
    //  short: +35 us for the second round)
    const bool big_step = h->prev_cost > 0.0 && h->last_cost > 0.0 && std::fabs(h->prev_cost - h->last_cost) > 0.05 * h->prev_cost;
    const int margin = (h->ldi_moved || big_step) ? std::max(6, h->cg_margin)
                                    : ((h->last_pcg_iters == h->prev_pcg_iters) ? std::min(2, h->cg_margin) : h->cg_margin);
    h->ldi_moved = false;
    // (no prediction -- the first call of a solve: 24.  C3's first iteration needs 18 + 2; 16 meant a second round, +40 us, in every
    //  cold solve; a launch past convergence costs ~1.5 us)
    int count = h->last_pcg_iters > 0 ? h->last_pcg_iters + margin : 24;
    for (;;) {
        count = std::min(count, max_iters + 2 - h->cg_launched);
        { StageTimer tk(h, PS_ST_CG_KERNEL, 1); cg_fused_launch<D>(h, tol, count); }
        cg_fused_recover<D>(h, h->status);
        tp.stop();
        if (gn_tail(h, linesearch, h->status, true)) return -1;
        if (total) total->stop();                       // close the iteration timer before the sync
        if (wait_published(h)) return -1;               // k_reduce3 has published status + scalars to host memory
        if (h->h_status[ST_PCG_DONE] == 2 && !h->h_status[ST_DIAG_FAIL] && !h->h_status[ST_LM_FAIL]) {
            // breakdown of the pipelined recurrences (the gated tail applied nothing): the synchronous solver on the same
            // matrix (it restarts from the iterate it reaches, and ends at the classic PCG if that fails too), then the tail
            if (cg_fused_run<D>(h, tol, max_iters, iters_out, relres_out, true)) return -1;
            if (gn_tail(h, linesearch, nullptr, true) || wait_published(h)) return -1;
            return 0;
        }
        if (h->h_status[ST_PCG_DONE] != 0) break;
        if (h->cg_launched >= max_iters + 2) {          // not converged within max_iters: take the step anyway
            cg_fused_recover<D>(h, nullptr);
            if (gn_tail(h, linesearch, nullptr, true) || wait_published(h)) return -1;
            break;
        }
        count = std::max(8, h->cg_launched / 2);
    }
    if (cg_report(h, iters_out, relres_out)) return -1;
    // a clean standard solve: its operator and CG coefficients seed the lagged dense inverse (side stream)
    if (h->h_status[ST_PCG_DONE] == 1 &&
        ldi_seed_enqueue<D>(h, h->h_status[ST_PCG_ITERS], linesearch ? h->h_scalars[SC_COST] : h->h_scalars[SC_LINCOST])) return -1;
    return 0;
}
}  // namespace

namespace {
struct DevBuf {                      // scoped device allocation for the stateless entry points
    void* p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
    int get(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8) == hipSuccess ? 0 : fail("hipMalloc failed"); }
    template <class T> T* as() { return static_cast<T*>(p); }
};
int need_device() {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("no HIP device visible");
    return 0;
}
}  // namespace

namespace {
// ---- host threads for the structure pass of ps_problem_create (round 4: the O(N) loops over 5 M observations and 22 M pairs at
// C4 were 0.6 s on one core) -------------------------------------------------------------------------------------------------
inline int ps_host_threads(long items, int cap = 16) {
    if (items < 200000) return 1;                             // (a thread start costs more than a small loop)
    return std::max(1, std::min({(int)std::thread::hardware_concurrency(), cap, 16}));
}
// fn(t, T): chunk t of T; T == 1 runs inline
template <class F>
void ps_parallel(int T, F fn) {
    if (T <= 1) { fn(0, 1); return; }
    std::vector<std::thread> pool;
    for (int t = 0; t < T; ++t) pool.emplace_back([&fn, t, T] { fn(t, T); });
    for (auto& th : pool) th.join();
}
// Stable counting sort of the INDICES 0..n-1 by key[i] in [0, nkeys) on several threads: out[position] = i, counts[k] = number
// of items with key k.  Thread t histograms its contiguous chunk; the (key-major, thread-minor) prefix gives every thread its
// own output range per key, so chunk order -- hence stability -- is kept.
inline void parallel_index_sort(long n, size_t nkeys, const int32_t* key, int32_t* out, std::vector<int32_t>& counts) {
    const int T = ps_host_threads(n, (size_t)nkeys > (1u << 18) ? 8 : 16);
    std::vector<std::vector<int32_t>> hist(T, std::vector<int32_t>(nkeys, 0));
    ps_parallel(T, [&](int t, int TT) {
        std::vector<int32_t>& hh = hist[t];
        for (long i = n * t / TT, e = n * (t + 1) / TT; i < e; ++i) hh[key[i]]++;
    });
    counts.assign(nkeys, 0);
    int32_t run = 0;
    for (size_t k = 0; k < nkeys; ++k) {
        for (int t = 0; t < T; ++t) { const int32_t c = hist[t][k]; hist[t][k] = run; run += c; counts[k] += c; }
    }
    ps_parallel(T, [&](int t, int TT) {
        std::vector<int32_t>& hh = hist[t];
        for (long i = n * t / TT, e = n * (t + 1) / TT; i < e; ++i) out[hh[key[i]]++] = (int32_t)i;
    });
}

// stable counting sort of `v` by an integer key in [0, nkeys): O(n + nkeys), used for the big
// host-side orderings of ps_problem_create (std::stable_sort was most of its run time)
template <class T, class KeyFn>
void counting_sort(std::vector<T>& v, size_t nkeys, KeyFn key) {
    std::vector<size_t> pos(nkeys + 1, 0);
    for (const T& x : v) pos[(size_t)key(x) + 1]++;
    for (size_t k = 0; k < nkeys; ++k) pos[k + 1] += pos[k];
    std::vector<T> out(v.size());
    for (const T& x : v) out[pos[(size_t)key(x)]++] = x;
    v.swap(out);
}
}  // namespace
