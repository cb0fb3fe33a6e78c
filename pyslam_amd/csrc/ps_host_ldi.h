// ps_host_ldi.h -- host side of the lagged dense inverse preconditioner (kernels and rationale: ps_k_ldi.h).
// Part of ps_core.hip (one translation unit; included after ps_host_cg.h).
//
// State machine, advanced once per whole-iteration call (ps_gn_iteration, folded single-GPU path):
//   none     --(a standard solve has finished: seed X_0 = c (I + X~ X~^T), 3 Newton-Schulz steps, side stream)-->  seeding
//   seeding  --(two calls later: wait for the side stream, ||R||_F small)-->  valid
//   valid    --(this call solves with it; beside the solve: one Newton-Schulz step against the new S)-->  valid
//   valid    --(the solve did not converge in ldi_cap iterations, or the step did not contract)-->  none (+ the standard path)
// "Two calls later" is a fixed schedule, not a completion poll: results are reproducible run to run.
// Whether a call TRIES the inverse is decided from the costs the previous calls returned (the relative decrease of the last
// step predicts how far S has moved): at most ldi_cost_tol, or unknown.

namespace {

bool ldi_eligible(const ps_problem* h) {
    const long n = (long)h->nr * h->D;
    return h->ldi_enable && h->pcg_variant == 1 && !h->cg_explicit && !h->cg_split && h->G > 0 && n > h->direct_max &&
           n <= h->ldi_max_n && n <= PS_LDI_MAXN && !(h->nccl_allreduce && h->nccl_comm);
}

int ldi_ensure(ps_problem* h) {
    if (h->ldi_ready) return 0;
    const int n = h->nr * h->D, np = (n + 63) / 64 * 64, kp = (h->nc + 15) / 16 * 16;
    h->ldi_n = n; h->ldi_np = np; h->ldi_kp = kp;
    const size_t nn = (size_t)np * np;
    {   // everything the inverse needs -- the direct seed's fp64 workspace included when it will be used -- in one block
        const bool direct = h->ldi_direct || (h->ldi_direct_ok && h->N == 0 && h->F > 0);
        size_t bytes = 6 * nn * 4 + 2 * (size_t)np * kp * 4 + ((size_t)n * h->nc + (size_t)n * h->D + 2 * np + n + nn / 64 + 64) * 8 + 64 * 256;
        if (direct) bytes += 3 * (size_t)n * n * 8 + (size_t)(cdiv(n, PS_BC_W) * PS_BC_W * PS_BC_W) * 8 + 4096;
        if (h->slab_reserve(bytes)) return -1;
    }
    if (h->alloc(&h->ldi_S32, nn) || h->alloc(&h->ldi_X32, nn) || h->alloc(&h->ldi_R32, nn) || h->alloc(&h->ldi_T32, nn) ||
        h->alloc(&h->ldi_Xu[0], nn) || h->alloc(&h->ldi_Xu[1], nn) || h->alloc(&h->ldi_Xt, (size_t)np * kp) ||
        h->alloc(&h->ldi_XtT, (size_t)np * kp) || h->alloc(&h->ldi_x64, (size_t)n * h->nc) ||
        h->alloc(&h->ldi_Linv, (size_t)h->nr * h->D * h->D) || h->alloc(&h->ldi_r[0], (size_t)np) ||
        h->alloc(&h->ldi_r[1], (size_t)np) || h->alloc(&h->ldi_part, (size_t)cdiv(n, PS_LDI_ROWS)) ||
        h->alloc(&h->ldi_fro_part, (size_t)(np / PS_GM_BM) * (np / PS_GM_BN)) || h->alloc(&h->ldi_coef, 4)) return -1;
    {   // column range of every 64-row tile of S^ (from the block pattern), in multiples of the GEMM's K chunk
        std::vector<int2> kr(np / PS_GM_BM);
        for (int tI = 0; tI < np / PS_GM_BM; ++tI) {
            int lo = np, hi = 0;
            for (int row = tI * PS_GM_BM; row < std::min(n, (tI + 1) * PS_GM_BM); ++row) {
                const int br = row / h->D;
                for (int b = h->h_row_ptr[br]; b < h->h_row_ptr[br + 1]; ++b) {
                    lo = std::min(lo, h->h_col_idx[b] * h->D); hi = std::max(hi, (h->h_col_idx[b] + 1) * h->D);
                }
            }
            if ((tI + 1) * PS_GM_BM > n) { lo = std::min(lo, std::max(n, tI * PS_GM_BM)); hi = np; }   // padding rows: identity
            if (lo >= hi) { lo = 0; hi = PS_GM_BK; }
            kr[tI] = make_int2(lo / PS_GM_BK * PS_GM_BK, std::min(np, (hi + PS_GM_BK - 1) / PS_GM_BK * PS_GM_BK));
        }
        if (h->upload(&h->ldi_krange, kr)) return -1;
    }
    // zero outside the block pattern / the real rows, identity on the padding: every GEMM below preserves that structure
    HIP_OK(hipMemsetAsync(h->ldi_S32, 0, nn * sizeof(float), h->stream));
    HIP_OK(hipMemsetAsync(h->ldi_X32, 0, nn * sizeof(float), h->stream));
    HIP_OK(hipMemsetAsync(h->ldi_Xu[0], 0, nn * sizeof(float), h->stream));
    HIP_OK(hipMemsetAsync(h->ldi_Xu[1], 0, nn * sizeof(float), h->stream));
    if (np > n) hipLaunchKernelGGL(k_ldi_pad_identity, dim3(cdiv(np - n, 64)), dim3(64), 0, h->stream, n, np, h->ldi_S32, 1.0f);
    HIP_OK(hipStreamSynchronize(h->stream));
    if (!h->ev_ldi) {
        HIP_OK(hipEventCreateWithFlags(&h->ev_ldi, PS_XSTREAM_EVENT_FLAGS));
        HIP_OK(hipEventCreateWithFlags(&h->ev_ldi_sread, PS_XSTREAM_EVENT_FLAGS));
        HIP_OK(hipEventCreateWithFlags(&h->ev_ldi_ritz, PS_XSTREAM_EVENT_FLAGS));
    }
    const size_t lds = (size_t)np * sizeof(double);
    if (ensure_dynamic_lds((const void*)k_ldi_init, (size_t)(lds))) return -1;
    if (ensure_dynamic_lds((const void*)k_ldi_update, (size_t)(lds))) return -1;
    // pose graphs (no landmark, pose factors only): the direct seed from the start -- their Newton-Schulz seeds never contract
    if (h->ldi_direct_ok && h->N == 0 && h->F > 0) h->ldi_direct = true;
    h->ldi_ready = true;
    return 0;
}

void ldi_gemm(ps_problem* h, hipStream_t st, int M, int N, int K, float alpha, const float* A, int lda, const float* B, int ldb,
              float beta, const float* Dm, int ldd, float gamma, float* C, int ldc, double* fro_part,
              const int2* krange = nullptr, int upper_only = 0, const float* dev_scale = nullptr) {
    hipLaunchKernelGGL(k_ldi_gemm, dim3(N / PS_GM_BN, M / PS_GM_BM), dim3(64 * PS_GM_KS), 0, st, M, N, K, alpha, A, lda, B, ldb, beta, Dm, ldd,
                       gamma, C, ldc, fro_part, krange, upper_only, dev_scale);
}

// one Newton-Schulz step in the scaled coordinates: R = I - S32 X ; T = X + X R ; X = sym(T) (and X_u = unscaled X into `xu`)
template <int D>
void ldi_ns_step(ps_problem* h, hipStream_t st, float* xu, bool want_fro) {
    const int np = h->ldi_np;
    // R = I - S^ X: S^ is banded like S, each row tile only walks its own column range (C3: 544 of 1 216);
    // T = X + X R is symmetric: tiles on and above the diagonal only (k_ldi_sym_unscale mirrors them)
    ldi_gemm(h, st, np, np, np, -1.f, h->ldi_S32, np, h->ldi_X32, np, 0.f, nullptr, 0, 1.f, h->ldi_R32, np,
             want_fro ? h->ldi_fro_part : nullptr, h->ldi_krange, 0);
    ldi_gemm(h, st, np, np, np, 1.f, h->ldi_X32, np, h->ldi_R32, np, 1.f, h->ldi_X32, np, 0.f, h->ldi_T32, np, nullptr, nullptr, 1);
    // X <- mirror(T); the unscaled copy for the solver stream only when asked for (the last step of a seed, a refresh)
    hipLaunchKernelGGL(k_ldi_mirror, dim3(np / 32, np / 32), dim3(256), 0, st, np, h->ldi_T32, h->ldi_X32);
    if (xu) {
        const int tail = cdiv(np - h->ldi_n, 64);
        hipLaunchKernelGGL(k_ldi_sym_unscale<D>, dim3(h->nr * h->nr + tail), dim3(64), 0, st, h->nr, np, h->ldi_X32, h->ldi_Linv,
                           h->ldi_T32, xu);                    // (reads the mirrored X; its symmetrised copy goes to the scratch T)
    }
    if (want_fro)
        hipLaunchKernelGGL(k_ldi_fro_total, dim3(1), dim3(256), 0, st, (np / PS_GM_BM) * (np / PS_GM_BN), h->ldi_fro_part, h->h_ldi_fro_dev);
}

// DIRECT seed: X_u = S^-1 by the blocked dense Cholesky + triangular inverse the explicit PCG uses for wide coarse
// matrices (k_bchol_panel / k_bchol_update, k_btri_inverse / k_btri_merge: ~2 n / 24 + 10 launches) and k_xcg_ainv, on the
// side stream, in place of X_0 + Newton-Schulz.  For systems whose two-level operator is too ill-conditioned a start for the
// fp32 iteration -- pose graphs: their seeds never contracted (kappa ~ 50-100) and they solved with 50-80 CG iterations
// per step for a system of a few hundred unknowns (BASELINE config 1: 100 SE(2) poses, 68 iterations, 0.57 ms).
// `state`: 1 = seed, 3 = refresh (there is no incremental refresh here: the inverse is formed again).
template <int D>
int ldi_direct_enqueue(ps_problem* h, int state, int lag) {
    const int n = h->ldi_n, np = h->ldi_np;
    const size_t nn = (size_t)n * n;
    const int nsteps = cdiv(n, PS_BC_W);
    if (!h->ldi_A64) {
        if (h->alloc(&h->ldi_A64, nn) || h->alloc(&h->ldi_Li, nn) || h->alloc(&h->ldi_LiT, nn) ||
            h->alloc(&h->ldi_Tinv, (size_t)nsteps * PS_BC_W * PS_BC_W) || h->alloc(&h->ldi_stat, ST_NWORDS)) return -1;
    }
    if (!h->ldi_stream && !ps_pool().take(ps_pool().side_streams, &h->ldi_stream))
        HIP_OK(hipStreamCreateWithFlags(&h->ldi_stream, hipStreamNonBlocking));
    hipStream_t st = h->ldi_stream;
    double* A = h->ldi_A64;
    HIP_OK(hipMemsetAsync(A, 0, nn * sizeof(double), st));
    HIP_OK(hipMemsetAsync(h->ldi_stat, 0, ST_NWORDS * sizeof(int32_t), st));
    hipLaunchKernelGGL(k_ldi_dense64<D>, dim3(h->nnzb), dim3(64), 0, st, h->brow_of, h->col_idx, h->S, A, n);
    HIP_OK(hipEventRecord(h->ev_ldi_sread, st));              // S has been read: the next linearisation may overwrite it
    h->ldi_sread_pending = true;
    for (int s2 = 0; s2 < nsteps; ++s2) {
        const int j0 = s2 * PS_BC_W, w = std::min(PS_BC_W, n - j0), m = n - j0 - w;
        hipLaunchKernelGGL(k_bchol_panel, dim3(std::max(1, cdiv((long)m * w, 1024))), dim3(256), 0, st, n, j0, A,
                           h->ldi_Tinv + (size_t)s2 * PS_BC_W * PS_BC_W, h->ldi_stat);
        if (m > 0) {
            const int nt = cdiv(m, 32);
            hipLaunchKernelGGL(k_bchol_update, dim3(nt * (nt + 1) / 2), dim3(256), 0, st, n, j0, w, A);
        }
    }
    const size_t inv_lds = ((size_t)PS_BI_S0 + PS_BC_W) * PS_BI_CW * sizeof(double);
    hipLaunchKernelGGL(k_btri_inverse, dim3(cdiv(n, PS_BI_CW)), dim3(256), inv_lds, st, n, A, h->ldi_Tinv, h->ldi_Li, h->ldi_LiT);
    for (int s2 = PS_BI_S0; s2 < n; s2 *= 2) {
        const int pairs = cdiv(n, 2 * s2), nt = cdiv(s2, PS_BM_T);
        for (int stage = 0; stage < 2; ++stage)
            hipLaunchKernelGGL(k_btri_merge, dim3(pairs * nt * nt), dim3(256), 0, st, n, s2, stage, A, h->ldi_Li, h->ldi_LiT);
    }
    const int wb = h->ldi_cur < 0 ? 0 : (h->ldi_cur ^ 1);
    HIP_OK(hipMemsetAsync(h->ldi_Xu[wb], 0, (size_t)np * np * sizeof(float), st));
    hipLaunchKernelGGL(k_xcg_ainv, dim3(cdiv(n, PS_AI_T) * (cdiv(n, PS_AI_T) + 1) / 2), dim3(256), 0, st, n, h->ldi_Li, h->ldi_Xu[wb], np);
    hipLaunchKernelGGL(k_ldi_direct_done, dim3(1), dim3(64), 0, st, h->ldi_stat, h->h_ldi_fro_dev);
    HIP_OK(hipEventRecord(h->ev_ldi, st));
    h->ldi_state = state; h->ldi_next = wb; h->ldi_ready_at = h->ldi_iter + lag; h->ldi_fro_limit = 0.1;
    h->ldi_refreshed = state == 3;
    h->ldi_next_tag = h->ldi_call_start_cost;
    if (state == 1) ++h->ldi_seeds;
    return 0;
}

// calls between a direct seed and its first use: the factorisation is ~2 n / 24 dependent launches (a fixed schedule by size,
// not a completion poll: the call that first uses the inverse is the same from run to run)
inline int ldi_direct_lag(const ps_problem* h) { return h->ldi_n <= 400 ? 2 : (h->ldi_n <= 800 ? 4 : 6); }

// After a standard (folded two-level) solve of a whole-iteration call: seed the inverse on the side stream from that
// solve's own operator.  Everything the side stream reads from the solver's buffers (S, the block-Jacobi factors, the
// prolongation) is read by its first kernels; ev_ldi_sread marks their end and the next linearisation waits for it.
template <int D>
int ldi_seed_enqueue(ps_problem* h, int its, double cost_now) {
    if (!ldi_eligible(h) || h->ldi_state == 1 || h->ldi_state == 3) return 0;       // (a seed / a re-formed inverse is already in flight)
    // A seed costs the call that starts it ~45 us of enqueueing and the next one ~0.1 ms of GEMMs beside its kernels, and
    // gives ~80 us back per call from then on: with fewer than three calls to come it is a loss.  The caller's stopping
    // rule says how many can come (option "solve_horizon"; a settling step IS a non-decreasing one under the reference's
    // rule, problem.py:177-178: cost >= 0.9 prev) -- C3 under the examples' options: calls 2-4 at 0.31 ms instead of 0.37 / 0.40 / 0.22.
    if (h->solve_horizon >= 0 && h->solve_horizon < 3) return 0;
    if (h->spec_enqueued) return 0;                              // the next linearisation is already overwriting S
    if (h->ldi_iter < h->ldi_no_seed_before) return 0;         // (back-off after rejected seeds)
    // only once the solve has begun to settle: an inverse of an S that the next steps leave far behind is wasted side work
    // (last_cost: the cost the previous call left behind = where this call started; cost_now: what this call returns)
    // ... or when the caller linearises at the SAME point again (same start cost as the call before: a damping retry, a
    // repeated first iteration): the inverse of this very S is what the next call needs
    const bool settling = h->last_cost > 0.0 && cost_now > 0.0 && std::fabs(h->last_cost - cost_now) <= h->ldi_cost_tol * h->last_cost;
    const bool same_point = h->last_cost > 0.0 && h->last_cost == h->ldi_prev_start_cost;
    h->ldi_prev_start_cost = h->last_cost;
    if (!settling && !same_point) return 0;
    if (ldi_ensure(h)) return -1;
    if (h->ldi_direct) return ldi_direct_enqueue<D>(h, 1, ldi_direct_lag(h));
    if (its < 2) return 0;
    const int nr = h->nr, n = h->ldi_n, np = h->ldi_np, nc = h->nc, kp = h->ldi_kp, ncb = h->ncb;
    hipStream_t st = h->side;
    // c = 1.9 / (ritz_min + ritz_max) on the device (k_ldi_ritz): eig(c M0 S^) in (0, 1.9), Newton-Schulz contracts.
    // (`hist` was written by the CG on the solver stream; the host has synchronised with that stream since.)
    hipLaunchKernelGGL(k_ldi_ritz, dim3(1), dim3(64), 0, st, h->hist, h->hist_cap, h->status, h->ldi_coef);
    copy_doubles(st, h->ldi_Linv, h->Linv, (size_t)nr * D * D);
    hipLaunchKernelGGL(k_ldi_scaled_dense<D>, dim3(h->nnzb), dim3(64), 0, st, h->brow_of, h->col_idx, h->S, h->ldi_Linv, h->ldi_S32, np);
    const double* Xsrc = h->ldi_x64;
    if (h->last_setup_lagx) Xsrc = h->X2[h->lci_cur];       // the system was built with the lagged X~ itself
    else hipLaunchKernelGGL(k_coarse_xbuild<D>, dim3(cdiv((long)nr * D * nc, 256)), dim3(256), 0, st, nr, ncb, h->pnode, h->pw0,
                            h->pw1, h->Bmat, h->Lci2[h->lci_cur], h->ldi_x64);
    hipLaunchKernelGGL(k_ldi_seed_prep, dim3(cdiv((long)np * kp, 256)), dim3(256), 0, st, n, nc, np, kp, Xsrc, h->ldi_Xt, h->ldi_XtT);
    HIP_OK(hipEventRecord(h->ev_ldi_sread, st));
    h->ldi_sread_pending = true;
    // X_0 = c (I + X~ X~^T)
    ldi_gemm(h, st, np, np, kp, 1.f, h->ldi_Xt, kp, h->ldi_XtT, np, 0.f, nullptr, 0, 1.f, h->ldi_X32, np, nullptr, nullptr, 0, h->ldi_coef);
    const int wb = h->ldi_cur < 0 ? 0 : (h->ldi_cur ^ 1);
    for (int s = 0; s < h->ldi_seed_steps; ++s) {
        const bool last = s + 1 == h->ldi_seed_steps;
        ldi_ns_step<D>(h, st, last ? h->ldi_Xu[wb] : nullptr, last);
    }
    HIP_OK(hipEventRecord(h->ev_ldi, st));
    h->ldi_state = 1; h->ldi_next = wb; h->ldi_ready_at = h->ldi_iter + h->ldi_seed_lag; h->ldi_fro_limit = 0.1;
    h->ldi_refreshed = false;
    h->ldi_next_tag = h->ldi_call_start_cost;               // the cost at the point whose S this inverse is built from
    ++h->ldi_seeds;
    return 0;
}

// beside an LDI solve: one Newton-Schulz step of the inverse against the S this call linearised (kicked from the host's
// wait once k_ldi_init has stamped the set-up word: S is final then)
template <int D>
int ldi_update_kick(ps_problem* h) {
    if (!h->ldi_side_todo) return 0;
    h->ldi_side_todo = false;
    if (h->ldi_direct) return ldi_direct_enqueue<D>(h, 3, ldi_direct_lag(h));
    hipStream_t st = h->side;
    hipLaunchKernelGGL(k_ldi_scaled_dense<D>, dim3(h->nnzb), dim3(64), 0, st, h->brow_of, h->col_idx, h->S, h->ldi_Linv, h->ldi_S32, h->ldi_np);
    HIP_OK(hipEventRecord(h->ev_ldi_sread, st));
    h->ldi_sread_pending = true;
    const int wb = h->ldi_cur ^ 1;
    ldi_ns_step<D>(h, st, h->ldi_Xu[wb], true);
    HIP_OK(hipEventRecord(h->ev_ldi, st));
    h->ldi_state = 3; h->ldi_next = wb; h->ldi_ready_at = h->ldi_iter + 1; h->ldi_fro_limit = 0.3;
    h->ldi_refreshed = true;
    h->ldi_next_tag = h->ldi_call_start_cost;
    return 0;
}

int ldi_side_kick(ps_problem* h) { return h->D == 6 ? ldi_update_kick<6>(h) : ldi_update_kick<3>(h); }

void ldi_invalidate(ps_problem* h) {
    // (an in-flight seed keeps going: it is consumed or rejected on its own schedule)
    if (h->ldi_state != 1) h->ldi_state = 0;
    h->ldi_cur = -1;
    h->ldi_last_its = 0;
}

// start of a whole-iteration call: take over what the side stream has finished, decide whether this call solves with it
bool ldi_decide(ps_problem* h) {
    if (!ldi_eligible(h) || !h->ldi_ready) return false;
    if ((h->ldi_state == 1 || h->ldi_state == 3) && h->ldi_iter >= h->ldi_ready_at) {
        if (hipEventSynchronize(h->ev_ldi) != hipSuccess) { h->ldi_state = 0; h->ldi_cur = -1; return false; }
        const double rms = std::sqrt(*h->h_ldi_fro / (double)h->ldi_np);    // rms eigenvalue of R = I - S^ X before the last step
        const bool ok = std::isfinite(rms) && rms < h->ldi_fro_limit && (h->ldi_state == 1 || h->ldi_update_ok);
        h->ldi_last_rms = rms;
        if (ok) { h->ldi_cur = h->ldi_next; h->ldi_tag = h->ldi_next_tag; if (h->ldi_state == 1) h->ldi_rejects = 0; h->ldi_state = 2; }
        else {
            // a seed that did not contract in its Newton-Schulz steps (an operator whose preconditioned condition number is
            // large: long pose graphs) is side work for nothing: wait 8, 16, 32 ... calls before the next one
            if (h->ldi_state == 1) { h->ldi_rejects = std::min(h->ldi_rejects + 1, 10); h->ldi_no_seed_before = h->ldi_iter + (4L << h->ldi_rejects); }
            // ... unless the direct seed has not been tried on this problem yet: the next standard solve seeds with it
            if (h->ldi_state == 1 && !h->ldi_direct && h->ldi_direct_ok) { h->ldi_direct = true; h->ldi_rejects = 0; h->ldi_no_seed_before = 0; }
            h->ldi_state = 0; h->ldi_cur = -1; h->ldi_last_its = 0;
        }
    }
    // (state 3 before its ready_at: a re-formed inverse is in flight into the OTHER buffer; the one in use stays valid)
    if (!(h->ldi_state == 2 || h->ldi_state == 3) || h->ldi_cur < 0) return false;
    // How far is this call's linearisation point from the one the inverse was built at?  Judged by the cost (scale-free, and
    // the host has it for nothing): the cost at the inverse's point (its tag) against the cost this call starts from.
    // Either unknown (parameters replaced from outside): try, ldi_cap bounds the damage.
    if (h->ldi_tag > 0.0 && h->last_cost > 0.0 && std::fabs(h->ldi_tag - h->last_cost) > h->ldi_cost_tol * h->ldi_tag) {
        ldi_invalidate(h);        // the problem has moved too far: the standard path solves and re-seeds
        h->ldi_moved = true;      // (... with a wider launch margin: its last iteration count is from another phase of the solve)
        return false;
    }
    return true;
}

// The LDI solve of a whole-iteration call: classic PCG (k_pcg_spmv) preconditioned with X_u, gated tail, one
// synchronisation.  Returns 0 = solved and published, 1 = gave up (nothing applied: the caller runs the standard path).
template <int D>
int ldi_solve_and_finish(ps_problem* h, double tol, int max_iters, int linesearch, int* iters_out, double* relres_out,
                         StageTimer* tp, StageTimer* total) {
    const int nr = h->nr, n = h->ldi_n, np = h->ldi_np, nwg = cdiv(n, PS_LDI_ROWS);
    const size_t lds = (size_t)np * sizeof(double);
    const float* Xu = h->ldi_Xu[h->ldi_cur];
    const double tol2 = tol * tol;
    const int cap = std::min(h->ldi_cap, max_iters);
    hipLaunchKernelGGL(k_ldi_init, dim3(nwg), dim3(256), lds, h->stream, n, np, Xu, h->g, h->x, h->ldi_r[0], h->z, h->ldi_part,
                       h->status, h->h_setup_dev, ++h->setup_seq);
    // One Newton-Schulz step beside this solve only when the previous solve with this inverse took more than
    // ldi_refresh_its iterations: a refresh costs the latency-bound launches beside it ~70 us at C3 (measured), one PCG
    // iteration ~11 us, so a settled solve runs with NO side-stream work at all and a drifting S switches the refresh on
    // (beyond 2 048 unknowns a refresh -- two n^3 GEMMs beside the solve -- costs more than the iterations it saves: a 420-keyframe
    //  BA, 2 514 unknowns, ran 0.39 ms per call at 7 iterations and 0.69 once 8 iterations switched the refresh on; standard
    //  solver 0.60.  There the inverse simply ages until ldi_cap gives it up and the next standard solve re-seeds.)
    const bool refresh = h->ldi_state == 2 && h->ldi_last_its > h->ldi_refresh_its + (n > 2048 ? 4 : 0);   // (not while one is in flight)
    h->ldi_side_todo = refresh; h->ldi_update_ok = false;
    // launch sequence: spmv(0) update(0) spmv(1) update(1) ... ; spmv(k) is the launch that detects convergence of iteration
    // k, so a solve of m iterations needs 2 m + 1 launches and ends on an spmv -- exactly that many are enqueued when the
    // last two solves agreed (a launch past convergence still costs ~4.7 us: its loads are issued before the flag is read)
    int step = 0;                                            // launches enqueued so far
    const int guess = h->ldi_last_its > 0 ? h->ldi_last_its + (h->ldi_last_its == h->ldi_prev_its ? 0 : 1) : 4;
    int upto = 2 * std::min(guess, cap) + 1;
    for (;;) {
        h->cg_kernel_launches += upto - step;
        for (; step < upto; ++step) {
            const int k = step >> 1;
            double* pold = (k & 1) ? h->p1 : h->p0;
            double* pnew = (k & 1) ? h->p0 : h->p1;
            if ((step & 1) == 0)
                hipLaunchKernelGGL(k_pcg_spmv<D>, dim3(nr), dim3(256), 0, h->stream, nr, h->row_ptr, h->col_idx, h->S, h->z, pold, pnew,
                                   h->q, h->ldi_part, h->ldi_part, nwg, h->pq_part, h->hist, k, tol2, h->status, h->scalars);
            else
                hipLaunchKernelGGL(k_ldi_update, dim3(nwg), dim3(256), lds, h->stream, n, np, Xu, pnew, h->q, h->x, h->ldi_r[k & 1],
                                   h->ldi_r[(k + 1) & 1], h->z, h->pq_part, nr, h->hist, k, h->ldi_part, h->status);
        }
        if (tp) tp->stop();
        if (gn_tail(h, linesearch, h->status, true)) return -1;
        if (total) total->stop();
        if (wait_published(h)) return -1;
        if (h->h_status[ST_LM_FAIL] || h->h_status[ST_DIAG_FAIL]) return 1;     // the standard path reports these
        if (h->h_status[ST_PCG_DONE] == 1) break;
        // Not converged yet.  Will it be within the cap?  The iterations so far give the rate (r.z drops by a constant factor per
        // iteration to a good approximation): an inverse from a point the solve has left needs 20+ iterations, and finding
        // that out by running all of ldi_cap of them costs the call ~0.13 ms plus a synchronisation per four launches.
        bool hopeless = false;
        {
            const double rr0 = h->h_scalars[SC_RR0], rrf = h->h_scalars[SC_RRFINAL];
            const int done_its = h->h_status[ST_PCG_ITERS];
            if (done_its >= 3 && rr0 > 0.0 && rrf > 0.0 && rrf < rr0) {
                const double per_it = std::log(rrf / rr0) / done_its;                // < 0
                hopeless = std::log(tol2) / per_it > (double)cap + 1.0;
            } else if (done_its >= 3 && !(rrf < rr0)) hopeless = true;               // not contracting at all
        }
        if (h->h_status[ST_PCG_DONE] == 2 || step >= 2 * cap + 1 || hopeless) {   // NaN / not (going to be) converged within the cap
            ++h->ldi_fallbacks;
            h->ldi_side_todo = false;
            ldi_invalidate(h);
            return 1;
        }
        upto = std::min(step + 4, 2 * cap + 1);
    }
    h->ldi_prev_its = h->ldi_last_its;
    h->ldi_last_its = h->h_status[ST_PCG_ITERS];
    h->ldi_update_ok = true;                                 // X contracted on this S: the Newton-Schulz step beside it is sound
    ++h->ldi_solves;
    // (own report: the standard solver's launch-count prediction -- last_pcg_iters -- must not see these counts)
    if (iters_out) *iters_out = h->h_status[ST_PCG_ITERS];
    const double rr0 = h->h_scalars[SC_RR0], rrf = h->h_scalars[SC_RRFINAL];
    if (relres_out) *relres_out = rr0 > 0.0 ? std::sqrt(rrf / rr0) : 0.0;
    return 0;
}

}  // namespace
