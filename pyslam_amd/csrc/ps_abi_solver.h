// ps_abi_solver.h -- C ABI: staged calls, whole iterations, sharded protocol, covariance columns, options, profiling, dense normal solve.
// Part of ps_core.hip (one translation unit; included from there, in this order).

int ps_get_info(ps_problem* h, ps_problem_info* info) {
    if (!h || !info) return fail("null argument");
    info->dof = h->D; info->num_poses = h->P; info->num_reduced = h->nr; info->num_points = h->L;
    info->num_var_points = h->nv; info->num_obs = h->N; info->num_edges = h->F; info->num_priors = 0;
    info->reduced_nnzb = h->nnzb; info->num_pairs = h->npairs;
    info->reduce_count = ((long)h->nnzb + h->nr) / 2 * h->D * h->D + (long)h->nr * h->D + 3;
    info->device_bytes = (int64_t)h->dev_bytes;
    info->cg_restarts = h->cg_fallbacks;
    info->cg_kernel_launches = h->cg_kernel_launches;
    info->ldi_solves = h->ldi_solves; info->ldi_fallbacks = h->ldi_fallbacks; info->ldi_seeds = h->ldi_seeds;
    info->xcg_fused_solves = h->xf_solves; info->xcg_fused_fallbacks = h->xf_fallbacks;
    info->cg_persist_solves = h->cp_launches; info->cg_persist_failures = h->cp_failures;
    info->cg_persist_refused = h->cp_refused; info->persist_cus = h->persist_cus;
    info->persist_cus_needed = h->cp_ok ? h->cp_cus_needed : (h->xp_ok ? h->xp_cus_needed : 0);
    info->landmark_passes_taken_over = h->prelm_used;
    info->xcg_persist4_solves = h->xp4_launches;
    return 0;
}

int ps_eval_cost(ps_problem* h, int include_all_constant, double* cost) {
    if (!h || !cost) return fail("null argument");
    // (all blocks, every observation in the packed landmark pass's runs: that pass sums the cost -- the same sum, bit for bit, as
    //  the tail of a whole-iteration call that expects a successor forms, ps_host_cg.h: gn_tail -- and leaves this point's
    //  landmark pass done for a linearisation that follows)
    if (include_all_constant && h->fuse_cost == 1 && lm_cost_possible(h)) { if (lm_cost_pass(h, h->lin_lambda, SC_COST)) return -1; }
    else if (cost_pass(h, include_all_constant, SC_COST)) return -1;
    if (read_scalars(h)) return -1;
    *cost = h->h_scalars[SC_COST];
    if (include_all_constant) h->last_cost = *cost;         // the cost AT the current parameters (what ps_gn_iteration's line-search
    return 0;                                               // cost is for the parameters it leaves behind)
}

int ps_linearize(ps_problem* h, double lambda) {
    if (!h) return fail("null argument");
    h->last_cost = h->prev_cost = -1.0;                     // staged API: the cost history of whole-iteration calls ends here
    return linearize(h, lambda);
}

// exchange buffer of the landmark-sharded iteration: slot lists of the upper block triangle + the buffer itself
static int ensure_shard_pack(ps_problem* h) {
    if (h->shard_pack) return 0;
    const int nr = h->nr, DD = h->D * h->D;
    std::vector<int32_t> up, upT;
    for (int r = 0; r < nr; ++r)
        for (int b = h->h_row_ptr[r]; b < h->h_row_ptr[r + 1]; ++b) {
            const int c = h->h_col_idx[b];
            if (c < r) continue;
            up.push_back(b);
            const int32_t* lo = h->h_col_idx.data() + h->h_row_ptr[c];
            const int32_t* hi = h->h_col_idx.data() + h->h_row_ptr[c + 1];
            const int32_t* it = std::lower_bound(lo, hi, r);
            if (it == hi || *it != r) return fail("reduced block pattern is not symmetric");
            upT.push_back((int32_t)(it - h->h_col_idx.data()));
        }
    h->nup = (long)up.size();
    h->pack_count = h->nup * DD + (long)nr * h->D + 2 + 1;
    if (h->upload(&h->up_slot, up) || h->upload(&h->upT_slot, upT) || h->alloc(&h->shard_pack, (size_t)h->pack_count)) return -1;
    return 0;
}

int ps_reduce_buffer(ps_problem* h, void** dev_ptr, int64_t* count) {
    if (!h || !dev_ptr || !count) return fail("null argument");
    if (ensure_shard_pack(h)) return -1;
    *dev_ptr = h->shard_pack; *count = h->pack_count;
    return 0;
}

int ps_shard_pack(ps_problem* h) {
    if (!h) return fail("null argument");
    if (ensure_shard_pack(h)) return -1;
    const long ntail = (long)h->nr * h->D + 2;
    const int nb = (int)std::min<long>(4096, cdiv(h->pack_count, 256));
    if (h->D == 6) hipLaunchKernelGGL(k_shard_pack<6>, dim3(nb), dim3(256), 0, h->stream, h->nup, h->up_slot, h->S, ntail, h->g, h->status, h->shard_pack);
    else hipLaunchKernelGGL(k_shard_pack<3>, dim3(nb), dim3(256), 0, h->stream, h->nup, h->up_slot, h->S, ntail, h->g, h->status, h->shard_pack);
    return 0;
}

int ps_shard_unpack(ps_problem* h) {
    if (!h) return fail("null argument");
    h->prelin_valid = h->prelm_valid = false;            // S and g are overwritten with the all-reduced system
    if (ensure_shard_pack(h)) return -1;
    const long ntail = (long)h->nr * h->D + 2;
    const int nb = (int)std::min<long>(4096, cdiv(h->pack_count, 256));
    if (h->D == 6) hipLaunchKernelGGL(k_shard_unpack<6>, dim3(nb), dim3(256), 0, h->stream, h->nup, h->up_slot, h->upT_slot, h->shard_pack, h->S, ntail, h->g, h->status);
    else hipLaunchKernelGGL(k_shard_unpack<3>, dim3(nb), dim3(256), 0, h->stream, h->nup, h->up_slot, h->upT_slot, h->shard_pack, h->S, ntail, h->g, h->status);
    return 0;
}

// A tolerance <= 0 asks for the DEFAULT (include/pyslam_hip.h: ps_solve_reduced): what meets SURVEY 8d's bar on the step
// (|dx - dx_spsolve| <= 1e-8 |dx|) without the caller knowing cond(M^-1 S) -- error <= cond x relative residual
static inline double resolve_pcg_tol(const ps_problem* h, double tol) {
    if (tol > 0.0) return tol;
    return h->nv > 0 ? 1e-12 : 1e-14;      // Schur complements of a bundle adjustment / pose graphs (priors ~1e6 beside loops ~1)
}

int ps_solve_reduced(ps_problem* h, double tol, int max_iters, int* iters_out, double* relres_out) {
    if (!h) return fail("null argument");
    tol = resolve_pcg_tol(h, tol);
    const int rc = solve_reduced(h, tol, max_iters, iters_out, relres_out);
    if (rc == 0 && h->h_status[ST_LM_FAIL]) return fail("a landmark block H_ll is not positive definite");
    return rc;
}

// the staged entry points that read Z, C^-1, c of the last ps_linearize (include/pyslam_hip.h: ps_eval_cost)
static int z_of_last_linearize(ps_problem* h, const char* who) {
    if (!h->z_foreign || h->nv == 0) return 0;
    char buf[320];
    snprintf(buf, sizeof buf, "%s: Z, C^-1 and c no longer belong to the last ps_linearize -- a landmark pass has since run at another point "
             "(ps_eval_cost after the parameters moved, or a whole-iteration call that expected a successor): call ps_linearize again", who);
    return fail(buf);
}

int ps_backsub(ps_problem* h) {
    if (!h) return fail("null argument");
    if (z_of_last_linearize(h, "ps_backsub")) return -1;
    return backsub(h);
}

int ps_get_dx(ps_problem* h, double* dx_pose, double* dx_point) {
    if (!h) return fail("null argument");
    std::vector<double> tmp;
    if (dx_pose && h->nr) HIP_OK(hipMemcpyAsync(dx_pose, h->x, (size_t)h->nr * h->D * sizeof(double), hipMemcpyDefault, h->stream));
    if (dx_point && h->nv) {
        tmp.resize((size_t)h->nv * 3);
        HIP_OK(hipMemcpyAsync(tmp.data(), h->dxl, tmp.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    }
    if (sync(h)) return -1;
    for (int s2 = 0; dx_point && s2 < h->nv; ++s2)       // internal slot order -> the caller's vid order
        std::memcpy(dx_point + 3 * (size_t)h->h_vid_of_slot[s2], &tmp[3 * (size_t)s2], 3 * sizeof(double));
    return 0;
}

int ps_step_norm2(ps_problem* h, double* norm2) {
    if (!h || !norm2) return fail("null argument");
    if (step_norm(h) || read_scalars(h)) return -1;
    *norm2 = h->h_scalars[SC_DXP2] + h->h_scalars[SC_DXL2];
    return 0;
}

int ps_apply_update(ps_problem* h, double step) {
    if (!h) return fail("null argument");
    h->prelin_valid = h->prelm_valid = false;
    h->last_cost = h->prev_cost = -1.0;                     // parameters move without a cost: history unknown from here
    return apply_update(h, step);
}

int ps_snapshot_params(ps_problem* h) {
    if (!h) return fail("null argument");
    h->snap_valid = true;
    h->snap_cost = h->last_cost;
    const size_t n1 = (size_t)h->P * h->PW, n2 = (size_t)h->L * 3;
    if (n1 + n2)
        hipLaunchKernelGGL(k_copy2, dim3((unsigned)std::min<size_t>(2048, cdiv((long)(n1 + n2), 256))), dim3(256), 0, h->stream,
                           n1, (const double*)h->poses, h->poses_snap, n2, (const double*)h->points, h->points_snap);
    return 0;
}

// the end of ps_solve: the best parameters become the current ones by exchanging the two sets of tables -- no copy, nothing for
// the caller's final synchronisation to wait for.  The snapshot is consumed (its tables now hold the iterate that was given up)
static int restore_params_by_exchange(ps_problem* h) {
    if (!h->snap_valid) return fail("ps_solve: no snapshot to restore");
    h->params_moved_since_lin = true;
    h->prelin_valid = h->prelm_valid = false;
    h->last_cost = h->snap_cost; h->prev_cost = -1.0;
    std::swap(h->poses, h->poses_snap);
    std::swap(h->points, h->points_snap);
    h->snap_valid = false;
    return 0;
}

int ps_restore_params(ps_problem* h) {
    if (!h) return fail("null argument");
    if (!h->snap_valid) return fail("ps_restore_params: no snapshot (none taken, or ps_solve has consumed it)");
    h->params_moved_since_lin = true;
    h->prelin_valid = h->prelm_valid = false;
    h->last_cost = h->snap_cost; h->prev_cost = -1.0;       // the snapshot's own cost (if it was known), no step history
    const size_t n1 = (size_t)h->P * h->PW, n2 = (size_t)h->L * 3;
    if (n1 + n2)
        hipLaunchKernelGGL(k_copy2, dim3((unsigned)std::min<size_t>(2048, cdiv((long)(n1 + n2), 256))), dim3(256), 0, h->stream,
                           n1, (const double*)h->poses_snap, h->poses, n2, (const double*)h->points_snap, h->points);
    return 0;
}

int ps_get_params(ps_problem* h, double* poses, double* points) {
    if (!h) return fail("null argument");
    // hipMemcpyDefault: the destination may be host memory or the caller's own device buffer
    if (poses && h->P) HIP_OK(hipMemcpyAsync(poses, h->poses, (size_t)h->P * h->PW * sizeof(double), hipMemcpyDefault, h->stream));
    if (points && h->L) HIP_OK(hipMemcpyAsync(points, h->points, (size_t)h->L * 3 * sizeof(double), hipMemcpyDefault, h->stream));
    return sync(h);
}

int ps_set_params(ps_problem* h, const double* poses, const double* points) {
    if (!h) return fail("null argument");
    h->params_moved_since_lin = true;
    h->prelin_valid = h->prelm_valid = false;
    h->last_cost = h->prev_cost = -1.0;
    if (poses && h->P) HIP_OK(hipMemcpyAsync(h->poses, poses, (size_t)h->P * h->PW * sizeof(double), hipMemcpyDefault, h->stream));
    if (points && h->L) HIP_OK(hipMemcpyAsync(h->points, points, (size_t)h->L * 3 * sizeof(double), hipMemcpyDefault, h->stream));
    return sync(h);
}

int ps_gn_finish(ps_problem* h, int linesearch, double* cost_out, double* dx_pose_norm2, double* dx_point_norm2) {
    if (!h) return fail("null argument");
    if (z_of_last_linearize(h, "ps_gn_finish")) return -1;
    HIP_OK(hipMemsetAsync(h->scalars + SC_DXP2, 0, sizeof(double), h->stream));
    HIP_OK(hipMemsetAsync(h->scalars + SC_DXL2, 0, sizeof(double), h->stream));
    if (gn_tail(h, linesearch, nullptr)) return -1;
    if (read_scalars(h)) return -1;
    if (h->h_status[ST_LM_FAIL]) return fail("a landmark block H_ll is not positive definite");
    if (cost_out) *cost_out = linesearch ? h->h_scalars[SC_COST] : h->h_scalars[SC_LINCOST];
    if (dx_pose_norm2) *dx_pose_norm2 = h->h_scalars[SC_DXP2];
    if (dx_point_norm2) *dx_point_norm2 = h->h_scalars[SC_DXL2];
    return 0;
}

int ps_gn_solve_finish(ps_problem* h, double pcg_tol, int pcg_max_iters, int linesearch, double* cost_out,
                       double* dx_pose_norm2, double* dx_point_norm2, int* pcg_iters_out, double* pcg_relres_out) {
    if (!h) return fail("null argument");
    pcg_tol = resolve_pcg_tol(h, pcg_tol);
    if (z_of_last_linearize(h, "ps_gn_solve_finish")) return -1;
    HIP_OK(hipMemsetAsync(h->scalars + SC_DXP2, 0, sizeof(double), h->stream));
    HIP_OK(hipMemsetAsync(h->scalars + SC_DXL2, 0, sizeof(double), h->stream));
    if (h->nr > 0 && h->pcg_variant == 1) {
        const int rc = h->D == 6
            ? gn_solve_and_finish_async<6>(h, pcg_tol, pcg_max_iters, linesearch, pcg_iters_out, pcg_relres_out, nullptr)
            : gn_solve_and_finish_async<3>(h, pcg_tol, pcg_max_iters, linesearch, pcg_iters_out, pcg_relres_out, nullptr);
        if (rc) return -1;
    } else {
        if (solve_reduced(h, pcg_tol, pcg_max_iters, pcg_iters_out, pcg_relres_out)) return -1;
        if (gn_tail(h, linesearch, nullptr)) return -1;
        if (read_scalars(h)) return -1;
    }
    if (h->h_status[ST_LM_FAIL]) return fail("a landmark block H_ll is not positive definite");
    if (cost_out) *cost_out = linesearch ? h->h_scalars[SC_COST] : h->h_scalars[SC_LINCOST];
    if (dx_pose_norm2) *dx_pose_norm2 = h->h_scalars[SC_DXP2];
    if (dx_point_norm2) *dx_point_norm2 = h->h_scalars[SC_DXL2];
    return 0;
}

int ps_set_segment_exchange(ps_problem* h, void* nccl_all_gather_fn, int32_t world, int32_t rank, int64_t maxlen,
                            int64_t n_mine, const int64_t* mine, int64_t n_dst, const int64_t* dst, const int64_t* src_ptr,
                            const int64_t* src_off) {
    if (!h) return fail("null argument");
    if (!nccl_all_gather_fn) { h->seg_allgather = nullptr; return 0; }
    if (!(h->nccl_allreduce && h->nccl_comm)) return fail("ps_set_segment_exchange: call ps_set_collective first");
    if (ensure_shard_pack(h)) return -1;
    const int64_t T = 3, body = h->pack_count - T;
    if (world < 1 || rank < 0 || rank >= world || n_mine < 0 || n_dst < 0 || maxlen < T + n_mine || (n_mine && !mine) ||
        (n_dst && (!dst || !src_ptr || !src_off)))
        return fail("ps_set_segment_exchange: inconsistent plan");
    // every index is checked here, once: a plan made for another block pattern would scatter into the wrong slots silently
    for (int64_t k = 0; k < n_mine; ++k) if (mine[k] < 0 || mine[k] >= body) return fail("ps_set_segment_exchange: an element of this rank lies outside [upper(S) | g]");
    for (int64_t k = 0; k < n_dst; ++k) {
        if (dst[k] < 0 || dst[k] >= body || src_ptr[k + 1] < src_ptr[k]) return fail("ps_set_segment_exchange: bad destination table");
        for (int64_t q = src_ptr[k]; q < src_ptr[k + 1]; ++q) {
            const int64_t r = src_off[q] / maxlen, o = src_off[q] % maxlen;
            if (src_off[q] < 0 || r >= world || o < T || (q > src_ptr[k] && src_off[q] / maxlen <= src_off[q - 1] / maxlen))
                return fail("ps_set_segment_exchange: bad source table (offsets must be rank * maxlen + 3 + k, ascending rank)");
        }
    }
    if (n_dst && src_ptr[0] != 0) return fail("ps_set_segment_exchange: src_ptr must start at 0");
    if ((int64_t)world * maxlen > INT32_MAX || body > INT32_MAX) return fail("ps_set_segment_exchange: the exchange does not fit 32-bit plan entries");
    const std::vector<int32_t> vm(mine, mine + n_mine), vd(dst, dst + n_dst), vp(src_ptr, src_ptr + (n_dst ? n_dst + 1 : 0)),
                               vo(src_off, src_off + (n_dst ? src_ptr[n_dst] : 0));
    if (h->upload(&h->seg_mine, vm) || h->upload(&h->seg_dst, vd) || h->upload(&h->seg_src_ptr, vp) || h->upload(&h->seg_src_off, vo) ||
        h->alloc(&h->seg_in, (size_t)maxlen) || h->alloc(&h->seg_all, (size_t)maxlen * world)) return -1;
    h->seg_world = world; h->seg_rank = rank; h->seg_maxlen = maxlen; h->seg_nmine = n_mine; h->seg_ndst = n_dst;
    h->seg_allgather = (ps_problem::allgather_fn)nccl_all_gather_fn;
    return 0;
}

// [upper(S) | g | cost | flag] summed over the ranks: one all-reduce of the whole packed buffer between k_shard_pack and
// k_shard_unpack, or (round 6) the all-gather of the ranks' segments, read from and summed into S / g / cost / status directly
static int exchange_reduced_system(ps_problem* h) {
    enum { NCCL_F64 = 8, NCCL_SUM = 0 };
    if (!h->seg_allgather) {
        { StageTimer tpk(h, PS_ST_PACK); if (ps_shard_pack(h)) return -1; }
        {
            StageTimer tar(h, PS_ST_ALLREDUCE);
            if (h->nccl_allreduce(h->shard_pack, h->shard_pack, (size_t)h->pack_count, NCCL_F64, NCCL_SUM, h->nccl_comm, h->stream))
                return fail("ncclAllReduce of the reduced system failed");
        }
        { StageTimer tpk(h, PS_ST_PACK); if (ps_shard_unpack(h)) return -1; }
        return 0;
    }
    h->prelin_valid = h->prelm_valid = false;            // (as ps_shard_unpack: S and g are overwritten with the summed system)
    const long T = 3, ntail = (long)h->nr * h->D + 2;
    const unsigned nb1 = (unsigned)std::min<long>(2048, cdiv(h->seg_maxlen, 256)), nb2 = (unsigned)std::min<long>(4096, cdiv(h->seg_ndst + T, 256));
    {
        StageTimer tpk(h, PS_ST_PACK);
        if (h->D == 6) hipLaunchKernelGGL(k_seg_pack<6>, dim3(nb1), dim3(256), 0, h->stream, h->nup, h->up_slot, h->S, ntail, h->g, h->status, T, h->seg_nmine, h->seg_mine, h->seg_in, h->seg_maxlen);
        else hipLaunchKernelGGL(k_seg_pack<3>, dim3(nb1), dim3(256), 0, h->stream, h->nup, h->up_slot, h->S, ntail, h->g, h->status, T, h->seg_nmine, h->seg_mine, h->seg_in, h->seg_maxlen);
    }
    {
        StageTimer tar(h, PS_ST_ALLREDUCE);
        if (h->seg_allgather(h->seg_in, h->seg_all, (size_t)h->seg_maxlen, NCCL_F64, h->nccl_comm, h->stream))
            return fail("ncclAllGather of the reduced system's segments failed");
    }
    {
        StageTimer tpk(h, PS_ST_PACK);
        if (h->D == 6) hipLaunchKernelGGL(k_seg_sum<6>, dim3(nb2), dim3(256), 0, h->stream, h->seg_ndst, h->seg_dst, h->seg_src_ptr, h->seg_src_off, h->seg_all, T, h->seg_world, h->seg_maxlen, h->nup, h->up_slot, h->upT_slot, h->S, ntail, h->g, h->status);
        else hipLaunchKernelGGL(k_seg_sum<3>, dim3(nb2), dim3(256), 0, h->stream, h->seg_ndst, h->seg_dst, h->seg_src_ptr, h->seg_src_off, h->seg_all, T, h->seg_world, h->seg_maxlen, h->nup, h->up_slot, h->upT_slot, h->S, ntail, h->g, h->status);
    }
    return 0;
}

int ps_set_collective(ps_problem* h, void* nccl_all_reduce_fn, void* nccl_comm) {
    if (!h) return fail("null argument");
    h->nccl_allreduce = (ps_problem::allreduce_fn)nccl_all_reduce_fn;
    h->nccl_comm = nccl_comm;
    return 0;
}

int ps_shard_buffer(ps_problem* h, void** dev_ptr) {
    if (!h || !dev_ptr) return fail("null argument");
    *dev_ptr = h->shard_buf;
    return 0;
}

// Sharded second half WITHOUT a host synchronisation: (first != 0: CG setup,) CG launches, gated
// tail; cost and ||dx_point||^2 of this shard land in ps_shard_buffer for the caller's all-reduce.
// Returns 1 when this was the last, ungated pass (max_iters exhausted), else 0.
int ps_gn_solve_finish_enqueue(ps_problem* h, double pcg_tol, int pcg_max_iters, int linesearch, int first) {
    if (!h) return fail("null argument");
    if (h->nr == 0 || h->pcg_variant != 1) return fail("ps_gn_solve_finish_enqueue needs the fused CG and a reduced system");
    pcg_tol = resolve_pcg_tol(h, pcg_tol);
    if (first && z_of_last_linearize(h, "ps_gn_solve_finish_enqueue")) return -1;
    h->shard_out = true;
    // (a caller that drives the collectives itself cannot repeat a solve on ONE rank -- the others have applied their tails and
    //  wait in the next collective: it gets neither the explicit PCG's one-launch-per-iteration form, whose recurrences may break
    //  down, nor the one-launch-per-SOLVE forms, whose exchange may time out under a foreign load on the device)
    h->no_repeat = !(h->nccl_allreduce && h->nccl_comm);
    struct Reset { ps_problem* h; ~Reset() { h->shard_out = false; h->no_repeat = false; } } reset{h};
    if (first) {
        HIP_OK(hipMemsetAsync(h->scalars + SC_DXP2, 0, sizeof(double), h->stream));
        if (!h->coarse_built && build_coarse(h)) return -1;
        // (a caller that drives the collectives itself cannot repeat a solve whose single-reduction recurrences broke
        //  down: it gets the three-launch form; the core's own sharded iteration below handles the fallback)
        if (!(h->nccl_allreduce && h->nccl_comm) && h->xf_skip == 0) h->xf_skip = 1;
        if (h->cg_explicit) { if (h->D == 6 ? xcg_setup<6>(h, pcg_max_iters, true) : xcg_setup<3>(h, pcg_max_iters, true)) return -1; }
        else if (h->D == 6 ? cg_fused_setup<6>(h, pcg_max_iters, true) : cg_fused_setup<3>(h, pcg_max_iters, true)) return -1;
    }
    // (the explicit PCG runs iteration k in launch group k: one group less than the fused CG's launches)
    const int limit = pcg_max_iters + (h->cg_explicit ? 1 : 2);
    // (behind a step that changed the cost by more than 5 % the lagged operators are from the other side of it: spare launches,
    //  as in gn_solve_and_finish_async; no prediction -- the first call of a solve -- 24 launches for the folded CG)
    const bool big_step = h->prev_cost > 0.0 && h->last_cost > 0.0 && std::fabs(h->prev_cost - h->last_cost) > 0.05 * h->prev_cost;
    const int margin = big_step ? (h->cg_explicit ? 5 : std::max(6, h->cg_margin))
                                : ((h->cg_explicit || h->last_pcg_iters == h->prev_pcg_iters) ? std::min(2, h->cg_margin) : h->cg_margin);
    int count = first ? (h->last_pcg_iters > 0 ? h->last_pcg_iters + margin : (h->cg_explicit ? 32 : 24)) : std::max(8, h->cg_launched / 2);
    count = std::min(count, limit - h->cg_launched);
    const bool last = count <= 0;
    const int32_t* gate = last ? nullptr : h->status;
    if (h->cg_explicit) {
        if (!last) {
            const int head = std::min(count, 12);           // (see gn_solve_and_finish_async)
            if (h->D == 6) { xcg_launch<6>(h, pcg_tol, head); if (xcg_side_enqueue<6>(h)) return -1; xcg_launch<6>(h, pcg_tol, count - head); }
            else { xcg_launch<3>(h, pcg_tol, head); if (xcg_side_enqueue<3>(h)) return -1; xcg_launch<3>(h, pcg_tol, count - head); }
        }
        if (h->D == 6) hipLaunchKernelGGL(k_cg_unscale<6>, dim3(cdiv((long)h->nr * 6, 256)), dim3(256), 0, h->stream, h->nr, h->Linv, h->cg_xh, h->x, gate);
        else hipLaunchKernelGGL(k_cg_unscale<3>, dim3(cdiv((long)h->nr * 3, 256)), dim3(256), 0, h->stream, h->nr, h->Linv, h->cg_xh, h->x, gate);
    } else if (h->D == 6) { if (!last) cg_fused_launch<6>(h, pcg_tol, count); cg_fused_recover<6>(h, gate); }
    else { if (!last) cg_fused_launch<3>(h, pcg_tol, count); cg_fused_recover<3>(h, gate); }
    if (gn_tail(h, linesearch, last ? nullptr : h->status)) return -1;
    return last ? 1 : 0;
}

// Synchronise and read back: done flag, the (all-reduced) shard buffer, ||dx_pose||^2, CG statistics.
int ps_gn_result(ps_problem* h, int* done, double* shard2 /* [2] */, double* dx_pose_norm2,
                 int* pcg_iters_out, double* pcg_relres_out) {
    if (!h) return fail("null argument");
    double sb[2] = {0.0, 0.0};
    HIP_OK(hipMemcpyAsync(sb, h->shard_buf, sizeof(sb), hipMemcpyDeviceToHost, h->stream));
    if (read_scalars(h)) return -1;
    if (h->h_status[ST_LM_FAIL]) return fail("a landmark block H_ll is not positive definite");
    if (done) *done = h->h_status[ST_PCG_DONE];
    if (shard2) { shard2[0] = sb[0]; shard2[1] = sb[1]; }
    if (dx_pose_norm2) *dx_pose_norm2 = h->h_scalars[SC_DXP2];
    // the cost history the lagged operators' hold / try decisions read (same on every rank: the buffer was all-reduced)
    if (h->h_status[ST_PCG_DONE]) { h->prev_cost = h->last_cost; h->last_cost = sb[0]; }
    if (cg_report(h, pcg_iters_out, pcg_relres_out)) return -1;
    if (h->h_status[ST_PCG_DONE] && h->xcg_ref_pending) { h->xcg_ref_pending = false; h->xcg_its_ref = h->last_pcg_iters; }
    return 0;
}

// ps_gn_iteration; `start_cost_out` != NULL: also the cost AT the linearisation point (all blocks, ps_eval_cost's sum) --
// the start cost of Problem.solve (reference problem.py:133), whose pass is enqueued in front of the iteration instead of
// costing a call and a synchronisation of its own
static int gn_iteration_impl(ps_problem* h, double lambda, double pcg_tol, int pcg_max_iters, int linesearch,
                             double* cost_out, double* dx_norm_out, int* pcg_iters_out, double* pcg_relres_out,
                             double* start_cost_out) {
    if (!h) return fail("null argument");
    pcg_tol = resolve_pcg_tol(h, pcg_tol);
    h->start_cost_pending = false;
    h->lmfail_check = 0;
    if (start_cost_out) {
        const bool published_path = !(h->nccl_allreduce && h->nccl_comm) && h->nr > 0 && h->pcg_variant == 1 &&
                                    !(h->mo_fused && h->nv == 0 && h->F == 0 && h->D == 6 && h->N == h->Np && h->max_pose_obs <= 2048);
        if (!published_path) {                              // paths that end otherwise: a call of its own
            if (ps_eval_cost(h, 1, start_cost_out)) return -1;
            start_cost_out = nullptr;
        } else {
            const bool ahead = h->prelin_valid && h->prelin_lambda == lambda;       // (this point is linearised already)
            if (!ahead && h->fuse_cost == 1 && lm_cost_possible(h)) { if (lm_cost_pass(h, lambda, SC_STARTCOST)) return -1; }
            else if (cost_pass(h, 1, SC_STARTCOST)) return -1;
            h->start_cost_pending = true;
        }
    }
    struct CallClock {
        ps_problem* h; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        ~CallClock() { h->host_call_ns += std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count(); ++h->host_calls; }
    } cc{h};
    if (h->nccl_allreduce && h->nccl_comm) {
        // landmark-sharded iteration, everything on the solver's stream in ONE call:
        // linearize -> RCCL sum of [S | g | cost] -> replicated CG + gated shard-local tail ->
        // RCCL sum of {cost, ||dx_point||^2} -> one synchronisation
        if (h->nr == 0 || h->pcg_variant != 1) return fail("the sharded iteration needs the fused CG and a reduced system");
        enum { NCCL_F64 = 8, NCCL_SUM = 0 };
        StageTimer total(h, PS_ST_TOTAL, 2);
        // (round 6: as on one GPU, the landmark pass of this point may have run in the previous call's tail, summing that call's
        //  cost on its way -- every observation evaluated once per iteration on the shards too)
        if (linearize(h, lambda, true)) return -1;
        h->lmfail_check = h->lin_lmfail_tag;
        if (h->lin_lmfail_tag && *reinterpret_cast<volatile long long*>(h->h_lmfail + (h->lin_lmfail_tag & 1)) == h->lin_lmfail_tag) {
            // a landmark block that was not positive definite in the pass run ahead (the host has seen its word: the previous call
            // waited behind it): into the device status word NOW, so that the flag travels with the exchange and EVERY rank fails
            // this call -- a rank that gave up alone would leave the others waiting in the next collective
            HIP_OK(hipMemsetD32Async((hipDeviceptr_t)(h->status + ST_LM_FAIL), 1, 1, h->stream));
        }
        // ONE exchange over ranks of [upper(S) | g | cost | failure flag]
        if (exchange_reduced_system(h)) return -1;
        int first = 1, done = 0;
        double sb[2] = {0.0, 0.0}, dxp2 = 0.0;
        for (;;) {
            const int last = ps_gn_solve_finish_enqueue(h, pcg_tol, pcg_max_iters, linesearch, first);
            if (last < 0) return -1;
            {
                StageTimer tar(h, PS_ST_ALLREDUCE);
                if (h->nccl_allreduce(h->shard_buf, h->shard_buf, 2, NCCL_F64, NCCL_SUM, h->nccl_comm, h->stream))
                    return fail("ncclAllReduce of the shard scalars failed");
            }
            hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, h->stream, h->status, h->scalars, h->shard_buf,
                               h->h_status_dev, h->h_scalars_dev, h->h_shard_dev, h->h_seq_dev, ++h->seq);
            total.stop();
            if (wait_published(h)) return -1;
            if (h->h_status[ST_LM_FAIL]) return fail("a landmark block H_ll is not positive definite");
            done = h->h_status[ST_PCG_DONE];
            if (h->h_status[ST_PERSIST_FAIL])
                // not a numerical event (those are the same on every rank): THIS rank's one-launch solver did not get its workgroups
                // resident together within a second.  Solving again here alone would leave the other ranks waiting in a collective
                return fail("the one-launch CG timed out on this rank (its workgroups were not resident together: is another process using "
                            "this GPU?); a landmark-sharded iteration cannot be repeated on one rank alone -- run with the options "
                            "cg_persist 0 and xcg_persist 0 on a shared device");
            if (h->xf_active && done == 2 && !h->h_status[ST_DIAG_FAIL]) {
                // breakdown of the one-launch form's recurrences (the same on every rank: the solve is replicated): nothing
                // applied; set the solve up again in the three-launch form
                ++h->xf_fallbacks; h->xf_skip = 1; first = 1;
                continue;
            }
            sb[0] = h->h_shard[0]; sb[1] = h->h_shard[1];
            dxp2 = h->h_scalars[SC_DXP2];
            if (cg_report(h, pcg_iters_out, pcg_relres_out)) return -1;
            if (done && h->xcg_ref_pending) { h->xcg_ref_pending = false; h->xcg_its_ref = h->last_pcg_iters; }   // (coarse_adaptive_hold)
            if (done || last) break;
            first = 0;
        }
        if (h->prelm_pending) {                              // the tail ran the next linearisation's landmark pass in place of the cost pass
            h->prelm_pending = false;
            h->prelm_valid = h->h_status[ST_PCG_DONE] == 1 && !h->h_status[ST_LM_FAIL] && !h->h_status[ST_DIAG_FAIL];
        }
        if (cost_out) *cost_out = sb[0];
        if (dx_norm_out) *dx_norm_out = std::sqrt(dxp2 + sb[1]);
        h->prev_cost = h->last_cost; h->last_cost = sb[0];      // (all-reduced: the same on every rank)
        return 0;
    }
    if (h->mo_fused && h->nv == 0 && h->F == 0 && h->nr > 0 && h->D == 6 && h->N == h->Np && h->pcg_variant == 1 &&
        h->max_pose_obs <= 2048) {
        // motion-only: block-diagonal reduced system, the whole iteration is ONE launch (one workgroup per pose;
        // beyond ~2 000 observations per pose one workgroup is slower than the multi-kernel path)
        StageTimer total(h, PS_ST_TOTAL, 2);
        h->cov_ready = false;
        if (!h->mo_partials && h->alloc(&h->mo_partials, 2 * (size_t)h->nr)) return -1;
        if (!h->status_clean) {
            HIP_OK(hipMemsetAsync(h->status, 0, ST_NWORDS * sizeof(int32_t), h->stream));
            h->status_clean = true;
        }
        const ObsWide wp{h->sidx_p, h->stiff_tab};
        const long long seq_now = ++h->seq;
        if (h->wide_obs)
            hipLaunchKernelGGL(k_motion_only_iteration<true>, dim3(h->nr), dim3(PS_MO_THREADS), 0, h->stream, h->nr, h->pitems, h->pitem_ptr,
                           h->pobs, h->points, h->ogroups, h->poses, lambda, linesearch, h->x, h->mo_partials, h->status,
                           h->scalars, h->arrivals + 1, h->h_status_dev, h->h_scalars_dev, h->h_seq_dev, seq_now, wp);
        else
            hipLaunchKernelGGL(k_motion_only_iteration<false>, dim3(h->nr), dim3(PS_MO_THREADS), 0, h->stream, h->nr, h->pitems, h->pitem_ptr,
                           h->pobs, h->points, h->ogroups, h->poses, lambda, linesearch, h->x, h->mo_partials, h->status,
                           h->scalars, h->arrivals + 1, h->h_status_dev, h->h_scalars_dev, h->h_seq_dev, seq_now, wp);
        total.stop();
        if (wait_published(h)) return -1;
        if (pcg_iters_out) *pcg_iters_out = 0;
        if (pcg_relres_out) *pcg_relres_out = 0.0;
        if (h->h_status[ST_DIAG_FAIL]) {
            h->status_clean = false;
            return fail("reduced system has a non-positive-definite diagonal block (gauge freedom? hold a pose constant or add a prior)");
        }
        if (cost_out) *cost_out = linesearch ? h->h_scalars[SC_COST] : h->h_scalars[SC_LINCOST];
        if (dx_norm_out) *dx_norm_out = std::sqrt(h->h_scalars[SC_DXP2]);
        return 0;
    }
    {
        StageTimer total(h, PS_ST_TOTAL, 2);  // closed before the last synchronising read-back
        h->spec_enqueued = false;
        if (h->prelin_valid && h->prelin_lambda == lambda) h->prelin_valid = false;     // linearised ahead, behind the last call's tail
        else if (linearize(h, lambda, true)) return -1;
        h->lmfail_check = h->lin_lmfail_tag;   // (its landmark pass may have run in the previous call's tail: wait_published)
        if (h->nr > 0 && h->pcg_variant == 1) {
            const int rc = h->D == 6
                ? gn_solve_and_finish_async<6>(h, pcg_tol, pcg_max_iters, linesearch, pcg_iters_out, pcg_relres_out, &total)
                : gn_solve_and_finish_async<3>(h, pcg_tol, pcg_max_iters, linesearch, pcg_iters_out, pcg_relres_out, &total);
            if (rc) return -1;
        } else {
            if (solve_reduced(h, pcg_tol, pcg_max_iters, pcg_iters_out, pcg_relres_out)) return -1;
            if (gn_tail(h, linesearch, nullptr)) return -1;
            total.stop();
            if (read_scalars(h)) return -1;
        }
    }
    if (h->prelm_pending) {                                  // the tail ran the next linearisation's landmark pass in place of the cost pass
        h->prelm_pending = false;
        h->prelm_valid = h->h_status[ST_PCG_DONE] == 1 && !h->h_status[ST_LM_FAIL] && !h->h_status[ST_DIAG_FAIL];
    }
    if (h->spec_enqueued) {                                  // the next iteration's linearisation is in the queue: valid if this one ended well
        h->spec_enqueued = false;
        h->prelin_valid = h->h_status[ST_PCG_DONE] == 1 && !h->h_status[ST_LM_FAIL] && !h->h_status[ST_DIAG_FAIL];
        h->prelin_lambda = lambda;
    }
    if (!h->guards.empty()) { hipStreamSynchronize(h->stream); if (h->side) hipStreamSynchronize(h->side); if (h->ldi_stream) hipStreamSynchronize(h->ldi_stream); h->check_guards("after ps_gn_iteration"); }
    if (h->h_status[ST_LM_FAIL]) return fail("a landmark block H_ll is not positive definite");
    if (cost_out) *cost_out = linesearch ? h->h_scalars[SC_COST] : h->h_scalars[SC_LINCOST];
    if (start_cost_out) {                                    // (last_cost: what ps_eval_cost would have left, if no wait took it over)
        *start_cost_out = h->h_scalars[SC_STARTCOST];
        if (h->start_cost_pending) { h->start_cost_pending = false; h->last_cost = *start_cost_out; }
    }
    // (without a line search the cost returned is the cost at the START point: what the parameters left behind cost is unknown,
    //  and the tags that compare linearisation points by their cost must not take one for the other -- round-3 ADVICE)
    h->prev_cost = h->last_cost; h->last_cost = linesearch ? h->h_scalars[SC_COST] : -1.0;
    // a slot whose reduction did not run (no reduced poses / no variable landmarks) is stale: count it as 0
    if (dx_norm_out) *dx_norm_out = std::sqrt((h->nr > 0 ? h->h_scalars[SC_DXP2] : 0.0) + (h->nv > 0 ? h->h_scalars[SC_DXL2] : 0.0));
    return 0;
}

int ps_gn_iteration(ps_problem* h, double lambda, double pcg_tol, int pcg_max_iters, int linesearch,
                    double* cost_out, double* dx_norm_out, int* pcg_iters_out, double* pcg_relres_out) {
    return gn_iteration_impl(h, lambda, pcg_tol, pcg_max_iters, linesearch, cost_out, dx_norm_out, pcg_iters_out, pcg_relres_out, nullptr);
}

// The loop of Problem.solve (reference pyslam/problem.py:130-178), driven from here: no interpreter between two iterations,
// the start cost's pass rides in front of the first iteration.  Statement for statement pyslam_amd/problem.py: device_solve.
int ps_solve(ps_problem* h, const ps_solve_options* o, double pcg_tol, int pcg_max_iters, double* cost_history, int32_t cap,
             int32_t* n_history, int32_t* iterations, double* last_dx_norm, int32_t* pcg_iters, double* pcg_relres, double* iter_ms) {
    if (!h || !o || !cost_history || !n_history) return fail("null argument");
    if (h->nccl_allreduce && h->nccl_comm) return 1;          // sharded: every rank's start cost needs the all-reduce (the caller loops)
    if (o->max_iters < 0 || (long)o->max_iters + 2 > cap) return 1;
    if (ps_reset_solver_state(h)) return -1;
    int n = 0, it = 0, nd = 0;
    double cost = 0.0, dxn = 0.0, last_ratio = 1.0;
    bool done = false;
    while (!done) {
        ++it;
        // iterations the stopping rules still allow after this one if its step is non-decreasing (option "solve_horizon")
        h->solve_horizon = !o->allow_nondecreasing_steps ? 0
                         : std::max(0, std::min(o->max_nondecreasing_steps - (nd + 1), o->max_iters + 1 - it));
        // Will there be another iteration?  Surely (short of ||dx|| / min_cost stopping it) if even a non-decreasing step leaves
        // the solve running; without allow_nondecreasing_steps, likely while the steps still cut the cost in half.  Then the next
        // linearisation is enqueued behind this iteration's tail (wait_published).  Not while the lagged dense inverse may seed: its
        // side stream reads S after the call.
        h->expect_next = it <= o->max_iters && (o->allow_nondecreasing_steps ? h->solve_horizon >= 1 : (it >= 2 && last_ratio < 0.5));
        h->spec_next = h->expect_next && !(ldi_eligible(h) && h->solve_horizon >= 3);
        const auto t0 = std::chrono::steady_clock::now();
        double c0 = 0.0, c = 0.0, rel = 0.0;
        int its = 0;
        const int rc_it = gn_iteration_impl(h, o->lm_lambda, pcg_tol, pcg_max_iters, o->linesearch, &c, &dxn, &its, &rel, it == 1 ? &c0 : nullptr);
        h->spec_next = false; h->expect_next = 0;
        if (rc_it) return -1;
        if (it == 1) { cost = c0; cost_history[n++] = c0; h->prev_cost = c0; }
        const double prev = cost;
        cost = c;
        last_ratio = prev > 0.0 ? cost / prev : 1.0;
        if (pcg_iters) pcg_iters[it - 1] = its;
        if (pcg_relres) pcg_relres[it - 1] = rel;
        if (iter_ms) iter_ms[it - 1] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        cost_history[n++] = cost;
        done = it > o->max_iters || dxn < o->min_update_norm || cost < o->min_cost;
        if (o->allow_nondecreasing_steps) {
            if (nd == 0 && ps_snapshot_params(h)) return -1;
            if (cost >= o->min_cost_decrease * prev) ++nd; else nd = 0;
            if (nd >= o->max_nondecreasing_steps) { done = true; if (restore_params_by_exchange(h)) return -1; }
        } else done = done || cost >= o->min_cost_decrease * prev;
    }
    h->solve_horizon = -1;
    *n_history = n;
    if (iterations) *iterations = it;
    if (last_dx_norm) *last_dx_norm = dxn;
    return 0;
}

// Whole Problem.solve() loop of a one-pose motion-only problem in ONE launch (k_motion_only_solve)
int ps_motion_only_solve(ps_problem* h, const ps_solve_options* o, double* cost_history, int32_t cap, int32_t* n_history,
                         int32_t* iterations, double* last_dx_norm, double* pose12_out) {
    if (!h || !o || !cost_history || !n_history) return fail("null argument");
    h->prelin_valid = h->prelm_valid = false;
    const bool eligible = h->mo_fused && h->nv == 0 && h->F == 0 && h->nr == 1 && h->P == 1 && h->D == 6 && h->N == h->Np &&
                          h->pcg_variant == 1 && h->max_pose_obs <= 2048;
    const int need = o->max_iters + 2;                        // the start cost + at most max_iters + 1 iterations
    if (!eligible || need + 16 > PS_MO_HIST_WORDS || need > cap) return 1;       // not an error: the caller iterates itself
    h->cov_ready = false;
    h->last_cost = h->prev_cost = -1.0;
    if (!h->status_clean) {
        HIP_OK(hipMemsetAsync(h->status, 0, ST_NWORDS * sizeof(int32_t), h->stream));
        h->status_clean = true;
    }
    MoSolveOptions mo{};
    mo.max_iters = o->max_iters; mo.allow_nondecreasing_steps = o->allow_nondecreasing_steps;
    mo.max_nondecreasing_steps = o->max_nondecreasing_steps; mo.linesearch = o->linesearch;
    mo.min_update_norm = o->min_update_norm; mo.min_cost = o->min_cost; mo.min_cost_decrease = o->min_cost_decrease;
    mo.lambda = o->lm_lambda;
    const ObsWide wp{h->sidx_p, h->stiff_tab};
    const long long seq_now = ++h->seq;
    StageTimer total(h, PS_ST_TOTAL, 2);
    if (h->wide_obs)
        hipLaunchKernelGGL(k_motion_only_solve<true>, dim3(1), dim3(PS_MO_THREADS), 0, h->stream, h->pitems, h->pitem_ptr, h->pobs,
                           h->points, h->ogroups, h->poses, mo, h->x, h->status, h->scalars, h->h_mo_hist_dev, PS_MO_HIST_WORDS,
                           h->h_status_dev, h->h_scalars_dev, h->h_seq_dev, seq_now, wp);
    else
        hipLaunchKernelGGL(k_motion_only_solve<false>, dim3(1), dim3(PS_MO_THREADS), 0, h->stream, h->pitems, h->pitem_ptr, h->pobs,
                           h->points, h->ogroups, h->poses, mo, h->x, h->status, h->scalars, h->h_mo_hist_dev, PS_MO_HIST_WORDS,
                           h->h_status_dev, h->h_scalars_dev, h->h_seq_dev, seq_now, wp);
    total.stop();
    if (wait_published(h)) return -1;
    if (h->h_status[ST_DIAG_FAIL]) {
        h->status_clean = false;
        return fail("reduced system has a non-positive-definite diagonal block (gauge freedom? hold a pose constant or add a prior)");
    }
    const int n = (int)h->h_mo_hist[0];
    if (n < 1 || n > cap) return fail("ps_motion_only_solve: cost history out of range");
    for (int k = 0; k < n; ++k) cost_history[k] = h->h_mo_hist[15 + k];
    if (pose12_out) for (int k = 0; k < 12; ++k) pose12_out[k] = h->h_mo_hist[3 + k];
    *n_history = n;
    if (iterations) *iterations = (int32_t)h->h_mo_hist[1];
    if (last_dx_norm) *last_dx_norm = h->h_mo_hist[2];
    h->last_cost = cost_history[n - 1];
    return 0;
}

int ps_covariance_begin(ps_problem* h) {
    if (!h) return fail("null argument");
    h->prelin_valid = h->prelm_valid = false;
    if (linearize(h, 0.0)) return -1;
    if (h->nr > 0 && h->pcg_variant == 1 && !use_direct(h) && !h->coarse_built && build_coarse(h)) return -1;
    if (h->nr > 0 && h->pcg_variant == 1 && !use_direct(h) && !h->cg_explicit) {
        const int rc = h->D == 6 ? cg_fused_setup<6>(h, 16) : cg_fused_setup<3>(h, 16);
        if (rc) return -1;
    }
    if (read_scalars(h)) return -1;
    if (h->h_status[ST_LM_FAIL]) return fail("a landmark block H_ll is not positive definite");
    if (h->h_status[ST_DIAG_FAIL]) return fail("reduced system has a non-positive-definite diagonal block (gauge freedom? hold a pose constant or add a prior)");
    if (h->h_slot_of_vid.size() != h->h_vid_of_slot.size()) {
        h->h_slot_of_vid.assign(h->h_vid_of_slot.size(), 0);
        for (size_t s2 = 0; s2 < h->h_vid_of_slot.size(); ++s2) h->h_slot_of_vid[h->h_vid_of_slot[s2]] = (int32_t)s2;
    }
    h->cov_ready = true;
    return 0;
}

int ps_covariance_column(ps_problem* h, int kind, int index, int comp, double tol, int max_iters,
                         int* iters_out, double* relres_out) {
    if (!h) return fail("null argument");
    tol = resolve_pcg_tol(h, tol);
    if (!h->cov_ready) return fail("ps_covariance_column: call ps_covariance_begin first (any linearisation invalidates it)");
    if (kind == 0 ? (index < 0 || index >= h->nr || comp < 0 || comp >= h->D)
                  : (kind != 1 || index < 0 || index >= h->nv || comp < 0 || comp >= 3))
        return fail("ps_covariance_column: parameter index / component out of range");
    // right-hand side of H x = e in Schur form: c_l = C_l^-1 r_l, g = r_p - sum_l Z_l c_l
    if (h->nr) HIP_OK(hipMemsetAsync(h->g, 0, (size_t)h->nr * h->D * sizeof(double), h->stream));
    if (h->nv) HIP_OK(hipMemsetAsync(h->cvec, 0, (size_t)h->nv * 3 * sizeof(double), h->stream));
    const int slot = kind == 1 ? h->h_slot_of_vid[index] : index;
    hipLaunchKernelGGL(k_cov_rhs, dim3(1), dim3(64), 0, h->stream, kind, slot, comp, h->D, h->lm_ptr, h->lobs,
                       h->pose_rid, h->Z, h->Cinv, h->g, h->cvec);
    int its = 0; double rel = 0.0;
    if (h->nr > 0) {
        int rc;
        if (use_direct(h)) {
            rc = h->D == 6 ? direct_solve_enqueue<6>(h) : direct_solve_enqueue<3>(h);
            if (!rc) rc = read_scalars(h);
            if (!rc) rc = cg_report(h, &its, &rel);
        } else if (h->pcg_variant == 1 && h->cg_explicit)
            rc = h->D == 6 ? xcg_run<6>(h, tol, max_iters, &its, &rel) : xcg_run<3>(h, tol, max_iters, &its, &rel);
        else if (h->pcg_variant == 1)
            rc = h->D == 6 ? cg_fused_run<6>(h, tol, max_iters, &its, &rel, true) : cg_fused_run<3>(h, tol, max_iters, &its, &rel, true);
        else
            rc = h->D == 6 ? pcg_run<6>(h, tol, max_iters, &its, &rel) : pcg_run<3>(h, tol, max_iters, &its, &rel);
        if (rc) return -1;
    }
    if (backsub(h)) return -1;
    if (iters_out) *iters_out = its;
    if (relres_out) *relres_out = rel;
    return 0;
}

int ps_get_reduced_system(ps_problem* h, int32_t* row_ptr, int32_t* col_idx, double* vals, double* g) {
    if (!h) return fail("null argument");
    if (row_ptr) std::memcpy(row_ptr, h->h_row_ptr.data(), h->h_row_ptr.size() * sizeof(int32_t));
    if (col_idx) std::memcpy(col_idx, h->h_col_idx.data(), h->h_col_idx.size() * sizeof(int32_t));
    if (vals && h->nnzb) HIP_OK(hipMemcpyAsync(vals, h->S, (size_t)h->nnzb * h->D * h->D * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (g && h->nr) HIP_OK(hipMemcpyAsync(g, h->g, (size_t)h->nr * h->D * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    return sync(h);
}

int ps_get_landmark_factors(ps_problem* h, double* cinv, double* c) {
    if (!h) return fail("null argument");
    if (z_of_last_linearize(h, "ps_get_landmark_factors")) return -1;
    std::vector<double> t6((size_t)h->nv * 6), t3((size_t)h->nv * 3);
    if (h->nv) {
        HIP_OK(hipMemcpyAsync(t6.data(), h->Cinv, t6.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_OK(hipMemcpyAsync(t3.data(), h->cvec, t3.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    }
    if (sync(h)) return -1;
    for (int s2 = 0; s2 < h->nv; ++s2) {
        const size_t v = (size_t)h->h_vid_of_slot[s2];
        if (cinv) std::memcpy(cinv + 6 * v, &t6[6 * (size_t)s2], 6 * sizeof(double));
        if (c) std::memcpy(c + 3 * v, &t3[3 * (size_t)s2], 3 * sizeof(double));
    }
    return 0;
}

int ps_debug_reproj_blocks(ps_problem* h, double* r, double* jpose, double* jpoint) {
    if (!h || !r || !jpose || !jpoint) return fail("null argument");
    if (h->N == 0) return 0;
    double *dr, *djp, *djl;
    HIP_OK(hipMalloc((void**)&dr, (size_t)h->N * 3 * sizeof(double)));
    HIP_OK(hipMalloc((void**)&djp, (size_t)h->N * 18 * sizeof(double)));
    HIP_OK(hipMalloc((void**)&djl, (size_t)h->N * 9 * sizeof(double)));
    const ObsWide wl{h->sidx_l, h->stiff_tab};
    if (h->wide_obs)
        hipLaunchKernelGGL(k_debug_reproj<true>, dim3(cdiv(h->N, 256)), dim3(256), 0, h->stream, h->N, h->lobs, h->lorig,
                       h->poses, h->points, h->ogroups, dr, djp, djl, wl);
    else
        hipLaunchKernelGGL(k_debug_reproj<false>, dim3(cdiv(h->N, 256)), dim3(256), 0, h->stream, h->N, h->lobs, h->lorig,
                       h->poses, h->points, h->ogroups, dr, djp, djl, wl);
    hipMemcpyAsync(r, dr, (size_t)h->N * 3 * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    hipMemcpyAsync(jpose, djp, (size_t)h->N * 18 * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    hipMemcpyAsync(jpoint, djl, (size_t)h->N * 9 * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    const int rc = sync(h);
    hipFree(dr); hipFree(djp); hipFree(djl);
    return rc;
}

int ps_debug_factor_blocks(ps_problem* h, double* r, double* j1, double* j2) {
    if (!h || !r || !j1 || !j2) return fail("null argument");
    if (h->F == 0) return 0;
    const int D = h->D, DD = D * D, ROW = D + 2 * DD;
    double* dbg;
    HIP_OK(hipMalloc((void**)&dbg, (size_t)h->F * ROW * sizeof(double)));
    // the production kernel itself, with its tap open (binary factors first, then the priors: creation order)
    const int rc0 = D == 6 ? launch_factor_pass<6>(h, 0.0, dbg) : launch_factor_pass<3>(h, 0.0, dbg);
    std::vector<double> t((size_t)h->F * ROW);
    hipMemcpyAsync(t.data(), dbg, t.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    const int rc = sync(h);
    hipFree(dbg);
    if (rc0 || rc) return -1;
    for (long f = 0; f < h->F; ++f) {
        std::memcpy(r + f * D, &t[(size_t)f * ROW], D * sizeof(double));
        std::memcpy(j1 + f * DD, &t[(size_t)f * ROW + D], DD * sizeof(double));
        std::memcpy(j2 + f * DD, &t[(size_t)f * ROW + D + DD], DD * sizeof(double));
    }
    return 0;
}

int ps_debug_table_checksums(ps_problem* h, uint64_t* out, int capacity, int* count) {
    if (!h || !out || !count) return fail("null argument");
    if (sync(h)) return -1;
    struct Tab { const void* p; size_t bytes; };
    const bool tiled = h->schur_tiles > 1 && h->npair_items > 0;
    const Tab tabs[16] = {
        {h->point_vid, (size_t)h->L * 4}, {h->lobs, (size_t)h->N * sizeof(LObs)}, {h->lorig, (size_t)h->N * 4},
        {h->lm_ptr, ((size_t)h->nv + 1) * 4}, {h->lm_point, (size_t)h->nv * 4}, {h->pose_of_rid, (size_t)std::max(h->nr, 1) * 4},
        {h->pitems, (size_t)h->npitems * sizeof(PItem)}, {h->pitem_ptr, ((size_t)h->nr + 1) * 4}, {h->pobs, (size_t)h->Np * sizeof(LObs)},
        {h->gather_lists ? h->pairs : nullptr, (size_t)h->npairs * 8}, {h->pair_xitems, (size_t)8 * h->pair_per_xcd * sizeof(PairItem)},
        {tiled ? h->comb_items : nullptr, (size_t)h->ncomb * sizeof(PairItem)}, {tiled ? h->comb_tasks : nullptr, (size_t)h->npair_items * 4},
        {h->row_ptr, ((size_t)h->nr + 1) * 4}, {h->col_idx, (size_t)h->nnzb * 4}, {h->diag_slot, (size_t)h->nr * 4}};
    *count = 16;
    std::vector<unsigned char> buf;
    for (int t = 0; t < 16 && t < capacity; ++t) {
        uint64_t hsh = 0;
        if (tabs[t].p && tabs[t].bytes) {
            buf.resize(tabs[t].bytes);
            HIP_OK(hipMemcpy(buf.data(), tabs[t].p, tabs[t].bytes, hipMemcpyDeviceToHost));
            hsh = 1469598103934665603ull;
            // eight bytes at a time (the tables are 4-byte multiples; a tail shorter than a word is folded byte by byte)
            size_t i = 0;
            for (; i + 8 <= buf.size(); i += 8) { uint64_t w; std::memcpy(&w, &buf[i], 8); hsh = (hsh ^ w) * 1099511628211ull; }
            for (; i < buf.size(); ++i) hsh = (hsh ^ buf[i]) * 1099511628211ull;
        }
        out[t] = hsh;
    }
    return 0;
}

// the folded / explicit crossover depends on whether the lagged dense inverse can apply (ps_host_cg.h: build_coarse): an
// option that changes that answer has the coarse level rebuilt by the next solve
static void relook_path(ps_problem* h) {
    const long n = (long)h->nr * h->D;
    const bool possible = h->ldi_enable && n <= h->ldi_max_n && n <= PS_LDI_MAXN && n > h->direct_max;
    if (h->coarse_built && possible != h->xmin_auto_ldi && h->cg_explicit_min_rows < 0) h->coarse_built = false;
}

int ps_set_option(ps_problem* h, const char* name, double value) {
    if (!h || !name) return fail("null argument");
    const std::string n(name);
    // what a caller's own loop says about its next call leaves what was computed ahead in place
    if (n == "expect_next") { h->expect_next = value != 0; return 0; }
    if (n == "solve_horizon") { h->solve_horizon = value < 0 ? -1 : (int)std::min(value, 1e6); return 0; }
    h->prelin_valid = h->prelm_valid = false;
    if (n == "pcg_variant") { if (value != 0 && value != 1) return fail("pcg_variant must be 0 or 1"); h->pcg_variant = (int)value; }
    else if (n == "coarse_groups") {
        if (value < -1 || value >= PS_XCG_MAXNODES) return fail("coarse_groups out of range (-1 auto, 0 off, else number of hat intervals; above 63 only for the explicit two-level PCG, at most 1023)");
        h->coarse_req = (int)value; h->coarse_built = false;
    }
#ifdef PS_MEASURE
    else if (n == "cg_ablate") h->cg_ablate = (int)value;
    else if (n == "schur_ablate") h->schur_ablate = (int)value;
    else if (n == "lm_ablate") h->lm_ablate = (int)value;
#else
    else if (n == "cg_ablate" || n == "schur_ablate" || n == "lm_ablate") {
        if (value != 0.0) return fail(n + ": timing experiments exist in the measurement build only (__graft_entry__.build_measure(), PYSLAM_AMD_MEASURE=1)");
    }
#endif
    else if (n == "schur_pipeline") h->schur_pipeline = value != 0.0;
    else if (n == "schur_mode") {
        if (value != 0.0 && value != 1.0) return fail("schur_mode must be 0 (gather kernels) or 1 (pose-stationary kernel)");
        if (value == 0.0 && !h->gather_lists) return fail("schur_mode 0: the gather kernels' pair lists were not built for this handle (create it under PS_SCHUR_MODE=0 or 2)");
        h->schur_mode = (int)value;
    }
    else if (n == "schur_stream") { if (value != 0.0 && !h->st_tiles) return fail("schur_stream: the streaming lists were not built for this problem"); h->use_stream = value != 0.0; }
    else if (n == "coarse_lag") h->coarse_lag = value != 0.0;
    else if (n == "cg_force_restart") h->cg_force_restart = value != 0.0;
    else if (n == "xcg_restrict_fused") h->xcg_rt = value != 0;
    else if (n == "band_chol") { h->band_chol = value != 0; h->lci_next = -1; }
    else if (n == "lm_packed") h->lm_packed = value != 0;
    else if (n == "pose_async") h->pose_async = (int)value;
    else if (n == "pose_xcd") h->pose_xcd = value != 0;
    else if (n == "xcg_persist4") h->xcg_persist4 = value != 0;
    else if (n == "cg_pipelined") { if (value != 0 && value != 1 && value != 2) return fail("cg_pipelined must be 0, 1 or 2"); h->cg_pipelined = (int)value; }
    else if (n == "fuse_cost") h->fuse_cost = (int)value;       // 0 off, 1 on, 2 = in the tails only (not the start cost / ps_eval_cost)
    else if (n == "sync_refactor") h->sync_refactor = value != 0;
    else if (n == "hold_across_steps") h->hold_across_steps = value != 0;
    else if (n == "band_part") { h->band_part = value != 0; h->lci_next = -1; }
    else if (n == "band_part_chunk") { if (value < 0 || value > 4096) return fail("band_part_chunk must be 0 (automatic) .. 4096 nodes"); h->band_part_m = (int)value; h->lci_next = -1; }
    else if (n == "coarse_auto_hold") h->xcg_auto_hold = value != 0;
    else if (n == "coarse_adaptive_hold") h->xcg_adaptive_hold = value != 0;
    else if (n == "xcg_fused") { if (value != 0 && value != 1 && value != 2) return fail("xcg_fused must be 0, 1 or 2"); h->xcg_fused = (int)value; }
    else if (n == "lagged_inverse") { h->ldi_enable = value != 0; if (!h->ldi_enable) { h->ldi_cur = -1; if (h->ldi_state != 1) h->ldi_state = 0; } relook_path(h); }
    else if (n == "ldi_max_unknowns") { if (value < 0 || value > PS_LDI_MAXN) return fail("ldi_max_unknowns out of range (0 .. 3328)"); h->ldi_max_n = (int)value; relook_path(h); }
    else if (n == "ldi_cap") { if (value < 1 || value > 64) return fail("ldi_cap out of range (1 .. 64)"); h->ldi_cap = (int)value; }
    else if (n == "ldi_cost_tol") { if (!(value >= 0)) return fail("ldi_cost_tol must be >= 0"); h->ldi_cost_tol = value; }
    else if (n == "ldi_refresh_its") { if (value < 0 || value > 64) return fail("ldi_refresh_its out of range (0 .. 64)"); h->ldi_refresh_its = (int)value; }
    else if (n == "ldi_direct") { h->ldi_direct_ok = value != 0.0; h->ldi_direct = value > 0.0; }
    else if (n == "direct_fused") h->direct_fused = value != 0.0;
    else if (n == "ldi_seed_lag") { if (value < 1 || value > 16) return fail("ldi_seed_lag out of range (1 .. 16)"); h->ldi_seed_lag = (int)value; }
    else if (n == "ldi_seed_steps") { if (value < 1 || value > 40) return fail("ldi_seed_steps out of range (1 .. 40)"); h->ldi_seed_steps = (int)value; }
    else if (n == "coarse_refresh_every") { if (value < 1 || value > 16) return fail("coarse_refresh_every must be 1..16"); h->xcg_refresh_every = (int)value; }
    else if (n == "coarse_lag_x") { h->lagx = value != 0.0; h->lci_next = -1; h->side_todo = false; }
    else if (n == "cg_lds") h->cg_lds = value != 0.0;
    else if (n == "cg_persist") h->cg_persist = value != 0.0;
    else if (n == "xcg_persist") h->xcg_persist = value != 0.0;
    else if (n == "cg_persist_spin") { if (value < 0 || value > 1e7) return fail("cg_persist_spin out of range"); h->cp_spin = (unsigned)value; }
    else if (n == "cg_explicit") { h->explicit_ok = value != 0.0; h->coarse_built = false; }
    else if (n == "big_chol") h->big_chol = value != 0.0;
    else if (n == "fused_motion_only") h->mo_fused = value != 0.0;
    else if (n == "direct_max_unknowns") { if (value < 0 || value > 90) return fail("direct_max_unknowns must be 0..90"); h->direct_max = (int)value; }
    else if (n == "coarse_basis") { h->coarse_basis = value != 0.0; h->lci_next = -1; h->side_todo = false; }
    else if (n == "profile_every") { if (value < 1) return fail("profile_every must be >= 1"); h->prof_every = (int)value; }
    else if (n == "cg_margin") { if (value < 0 || value > 64) return fail("cg_margin out of range"); h->cg_margin = (int)value; }
    else if (n == "cg_split_min_rows") { h->cg_split_min_rows = (int)value; h->coarse_built = false; }
    else if (n == "cg_explicit_min_rows") { h->cg_explicit_min_rows = (int)value; h->coarse_built = false; }
    else if (n == "pcg_chunk") { if (value < 1 || value > 4096) return fail("pcg_chunk out of range"); h->pcg_chunk = (int)value; }
    else return fail("unknown option: " + n);
    return 0;
}

// Forget everything the solver carries from one whole-iteration call to the next, as if the handle had just been
// created: the lagged coarse factor / inverse and their tags, the lagged dense inverse of S (seeds in flight are waited
// for, then dropped), held coarse inverses, the launch-count predictions, the cost history.  Tables, parameters, options
// and the structures built at create time stay.  The first ps_gn_iteration after this runs the exact (un-lagged) set-up,
// exactly like the first iteration of a fresh handle.  What a caller that starts a NEW solve on a live handle calls
// (Problem.solve; bench.py's cold solves).
int ps_reset_solver_state(ps_problem* h) {
    if (!h) return fail("null argument");
    h->prelin_valid = h->prelm_valid = false;
    if (!h->solver_touched) {                               // nothing linearised since creation / the last reset: only the history
        h->last_cost = h->prev_cost = h->snap_cost = -1.0;
        h->solve_horizon = -1;
        return 0;
    }
    h->solver_touched = false;
    HIP_OK(hipStreamSynchronize(h->stream));
    if (h->side) HIP_OK(hipStreamSynchronize(h->side));
    if (h->ldi_stream) HIP_OK(hipStreamSynchronize(h->ldi_stream));
    drain_timers(h);
    // lagged coarse level (folded and explicit forms)
    h->lci_next = -1; h->lci_cur = 0;
    h->side_todo = false; h->side_ready = false; h->side_pending = false; h->acdone_pending = false;
    h->xcg_side_todo = false; h->xcg_lag_count = 0; h->xcg_held = 0;
    h->xcg_its_ref = 0; h->xcg_good_held = 0; h->xcg_ref_pending = false;
    h->xcg_tag[0] = h->xcg_tag[1] = -1.0; h->xcg_tag_lambda[0] = h->xcg_tag_lambda[1] = 0.0;
    h->xcg_setup_cost = -1.0; h->xcg_setup_lambda = 0.0;
    h->mc_active = false; h->last_setup_lagx = false; h->xf_skip = 0;
    if (h->lag_status) HIP_OK(hipMemsetAsync(h->lag_status, 0, ST_NWORDS * sizeof(int32_t), h->stream));
    // lagged dense inverse
    h->ldi_state = 0; h->ldi_cur = -1; h->ldi_next = -1; h->ldi_iter = 0; h->ldi_ready_at = 0;
    h->ldi_side_todo = false; h->ldi_update_ok = false; h->ldi_sread_pending = false; h->ldi_refreshed = false;
    h->ldi_last_its = h->ldi_prev_its = 0; h->ldi_rejects = 0; h->ldi_no_seed_before = 0;
    h->ldi_tag = h->ldi_next_tag = h->ldi_call_start_cost = -1.0; h->ldi_prev_start_cost = -2.0;
    h->ldi_moved = false; h->ldi_last_rms = 0.0;
    if (h->ldi_ready && !(h->ldi_direct_ok && h->N == 0 && h->F > 0)) h->ldi_direct = false;   // (a rejected seed had switched the direct seed on)
    // predictions and history
    h->last_pcg_iters = 0; h->prev_pcg_iters = -1;
    h->last_cost = h->prev_cost = h->snap_cost = -1.0;
    h->solve_horizon = -1;
    h->cov_ready = false;
    return 0;
}

/* the hash of the HIP sources this library was compiled from (-DPS_BUILD_SHA, set by __graft_entry__.build()) */
#ifndef PS_BUILD_SHA
#define PS_BUILD_SHA "unknown"
#endif
const char* ps_build_sha(void) { return PS_BUILD_SHA; }

int ps_set_profiling(ps_problem* h, int enabled) {
    if (!h) return fail("null argument");
    h->profiling = enabled < 0 ? 0 : (enabled > 2 ? 2 : enabled);
    return 0;
}

int ps_get_stage_times(ps_problem* h, double* ms, int64_t* counts, int reset) {
    if (!h) return fail("null argument");
    if (!h->pending.empty() && sync(h)) return -1;       // staged calls only enqueue: collect their events
    for (int i = 0; i < PS_NUM_STAGES; ++i) {
        if (ms) ms[i] = h->stage_ms[i];
        if (counts) counts[i] = h->stage_n[i];
        if (reset) { h->stage_ms[i] = 0.0; h->stage_n[i] = 0; }
    }
    return 0;
}

int ps_dense_normal_solve(const double* J, const double* r, int32_t m, int32_t n, double* dx, double* covariance) {
    if (!J || !r || !dx || m <= 0 || n <= 0) return fail("bad argument");
    if (n > 2048) return fail("generic (host-evaluated) path supports at most 2048 unknowns");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("no HIP device visible");
    const int nrhs = covariance ? n + 1 : 1;
    DevBuf bJ, br, bH, bB, bst, bg;                   // scoped: released on every return path
    if (bJ.get((size_t)m * n * sizeof(double)) || br.get((size_t)m * sizeof(double)) || bH.get((size_t)n * n * sizeof(double)) ||
        bB.get((size_t)n * nrhs * sizeof(double)) || bst.get(ST_NWORDS * sizeof(int32_t)) || bg.get((size_t)n * sizeof(double))) return -1;
    double *dJ = bJ.as<double>(), *dr = br.as<double>(), *dH = bH.as<double>(), *dB = bB.as<double>(), *dg = bg.as<double>();
    int32_t* dst = bst.as<int32_t>();
    HIP_OK(hipMemset(dst, 0, ST_NWORDS * sizeof(int32_t)));
    HIP_OK(hipMemcpy(dJ, J, (size_t)m * n * sizeof(double), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dr, r, (size_t)m * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_dense_normal, dim3(cdiv((long)n * n, 256)), dim3(256), 0, 0, m, n, dJ, dr, dH, dg);
    // B = [g | I]
    std::vector<double> B((size_t)n * nrhs, 0.0), gh(n);
    HIP_OK(hipMemcpy(gh.data(), dg, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) { B[(size_t)i * nrhs] = gh[i]; if (covariance) B[(size_t)i * nrhs + 1 + i] = 1.0; }
    HIP_OK(hipMemcpy(dB, B.data(), B.size() * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_dense_chol_solve, dim3(1), dim3(256), 0, 0, n, nrhs, dH, dB, dst);
    HIP_OK(hipMemcpy(B.data(), dB, B.size() * sizeof(double), hipMemcpyDeviceToHost));
    int32_t st[ST_NWORDS];
    HIP_OK(hipMemcpy(st, dst, sizeof(st), hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) {
        dx[i] = B[(size_t)i * nrhs];
        if (covariance) for (int j = 0; j < n; ++j) covariance[(size_t)i * n + j] = B[(size_t)i * nrhs + 1 + j];
    }
    if (st[ST_DIAG_FAIL]) return fail("normal matrix is not positive definite");
    return 0;
}
