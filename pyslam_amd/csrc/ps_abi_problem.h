// ps_abi_problem.h -- C ABI: error state, ps_problem_create / destroy (table upload, pair lists, XCD tiles).
// Part of ps_core.hip (one translation unit; included from there, in this order).


const char* ps_last_error(void) { return g_err.c_str(); }

int ps_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// One-off costs of a process, paid here instead of inside its first solve: the HIP runtime loads this library's code object
// at the FIRST kernel launch (5.8 ms on the MI355X box: it used to sit in the start-cost pass of the first Problem.solve,
// tools/first_call_probe.py) and sets its allocator up at the first hipMalloc (~150 ms, seen as "parameter tables" of the
// first ps_problem_create).  The binding calls this once from require_gpu().
int ps_warm_up(void) {
    static std::once_flag once;
    static int rc = 0;
    std::call_once(once, [] {
        void* p = nullptr;
        if (hipMalloc(&p, 256) != hipSuccess) { rc = fail("hipMalloc failed (no usable HIP device?)"); return; }
        hipLaunchKernelGGL(k_zero4, dim3(1), dim3(256), 0, 0, (size_t)8, (double*)p, (size_t)0, (double*)nullptr, (size_t)0,
                           (double*)nullptr, (size_t)0, (double*)nullptr);
        // (the runtime's own fill and copy kernels are resolved at their first use too: every linearisation starts with a
        //  hipMemsetAsync, the staged calls read their scalars back with hipMemcpyAsync)
        double hostw[4] = {0, 0, 0, 0};
        (void)hipMemsetAsync(p, 0, 256, 0);
        (void)hipMemcpyAsync((char*)p + 128, p, 64, hipMemcpyDeviceToDevice, 0);
        (void)hipMemcpyAsync(hostw, p, sizeof(hostw), hipMemcpyDeviceToHost, 0);
        if (hipDeviceSynchronize() != hipSuccess) rc = fail("the warm-up launch failed");
        hipFree(p);
        // The runtime also builds its object for every KERNEL lazily, at that kernel's first launch (~0.2 ms each: the first
        // whole-iteration call of a process took 6-8 ms for ~35 distinct kernels -- not under rocprofv3, which resolves them all
        // when it loads).  Asking for the attributes of the kernels of the whole-iteration paths resolves them here.
        hipFuncAttributes a;
#define PS_TOUCH(...) (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&__VA_ARGS__))
        PS_TOUCH(k_landmark_pass<false>); PS_TOUCH(k_landmark_pass_packed<false, false>); PS_TOUCH(k_landmark_pass_packed<false, true>); PS_TOUCH(k_backsub_packed); PS_TOUCH(k_pose_pass<false>); PS_TOUCH(k_pose_finalize); PS_TOUCH(k_schur_pairs_db<0>);
        PS_TOUCH(k_schur_combine); PS_TOUCH(k_backsub); PS_TOUCH(k_cost_reproj<false>); PS_TOUCH(k_cost_packed<false>); PS_TOUCH(k_reduce3); PS_TOUCH(k_reduce_partials);
        PS_TOUCH(k_copy2); PS_TOUCH(k_zero4); PS_TOUCH(k_lag_status_check);
        PS_TOUCH(k_block_jacobi_factor<6>); PS_TOUCH(k_scale_blocks<6>); PS_TOUCH(k_scale_blocks_p<6>); PS_TOUCH(k_rows_setup<6>);
        PS_TOUCH(k_coarse_rowsums<6>); PS_TOUCH(k_coarse_matrix<6>); PS_TOUCH(k_coarse_chol<6, true>); PS_TOUCH(k_coarse_border<6>);
        PS_TOUCH(k_coarse_mreduce<6>); PS_TOUCH(k_coarse_xbuild<6>); PS_TOUCH(k_coarse_recover<6>); PS_TOUCH(k_cg_fused_lds<6, 8>);
        PS_TOUCH(k_cg_fused<6, 8>); PS_TOUCH(k_cg_unscale<6>); PS_TOUCH(k_update_poses<6>); PS_TOUCH(k_update_points);
        PS_TOUCH(k_xcoarse_rowsums<6>); PS_TOUCH(k_xcoarse_matrix<6>); PS_TOUCH(k_band_chol<6>); PS_TOUCH(k_band_inverse); PS_TOUCH(k_band_inverse_rl<false>);
        PS_TOUCH(k_band_inverse_rl<true>); PS_TOUCH(k_bp_v); PS_TOUCH(k_bp_t); PS_TOUCH(k_bp_w); PS_TOUCH(k_bp_dense_sep); PS_TOUCH(k_bp_dense);
        PS_TOUCH(k_xcg_restrict<6>); PS_TOUCH(k_xcg_f2_coarse<6, 3>); PS_TOUCH(k_xcg_fused1<6, 8, false>); PS_TOUCH(k_xcg_fused1<6, 0, true>);
        PS_TOUCH(k_xcg_fused1<6, 2, false>); PS_TOUCH(k_xcg_fused1<6, 6, false>);
        PS_TOUCH(k_factor_pass<6>); PS_TOUCH(k_factor_assemble<6>); PS_TOUCH(k_cost_factors<6>);
        PS_TOUCH(k_factor_pass<3>); PS_TOUCH(k_factor_assemble<3>); PS_TOUCH(k_cost_factors<3>); PS_TOUCH(k_block_jacobi_factor<3>);
        PS_TOUCH(k_update_poses<3>); PS_TOUCH(k_direct_solve<6>); PS_TOUCH(k_direct_solve<3>);
        PS_TOUCH(k_motion_only_solve<false>); PS_TOUCH(k_motion_only_iteration<false>);
        PS_TOUCH(k_pcg_spmv<6>); PS_TOUCH(k_pcg_spmv<3>); PS_TOUCH(k_ldi_init); PS_TOUCH(k_ldi_update); PS_TOUCH(k_ldi_gemm); PS_TOUCH(k_ldi_mirror);
        PS_TOUCH(k_ldi_ritz); PS_TOUCH(k_ldi_seed_prep); PS_TOUCH(k_ldi_fro_total); PS_TOUCH(k_ldi_pad_identity);
        PS_TOUCH(k_ldi_scaled_dense<6>); PS_TOUCH(k_ldi_sym_unscale<6>); PS_TOUCH(k_ldi_scaled_dense<3>); PS_TOUCH(k_ldi_sym_unscale<3>);
        PS_TOUCH(k_shard_pack<6>); PS_TOUCH(k_shard_unpack<6>); PS_TOUCH(k_publish); PS_TOUCH(k_cov_rhs);
        PS_TOUCH(k_scale_blocks<3>); PS_TOUCH(k_rows_setup<3>); PS_TOUCH(k_coarse_rowsums<3>); PS_TOUCH(k_coarse_matrix<3>);
        PS_TOUCH(k_coarse_chol<3, true>); PS_TOUCH(k_coarse_border<3>); PS_TOUCH(k_coarse_mreduce<3>); PS_TOUCH(k_coarse_xbuild<3>);
        PS_TOUCH(k_coarse_recover<3>); PS_TOUCH(k_cg_fused_lds<3, 8>); PS_TOUCH(k_cg_unscale<3>);
#undef PS_TOUCH
    });
    return rc;
}

int ps_problem_destroy(ps_problem* h) {
    if (!h) return 0;
    if (h->cp_dbg) {
        long long c[8]; hipMemcpy(c, h->cp_dbg, 64, hipMemcpyDeviceToHost);
        int khz = 100000; hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
        fprintf(stderr, "k_cg_persist, workgroup 0, %ld launches, us per launch: recurrences %.2f | sync %.2f | products + publish %.2f | gather %.2f "
                "(%.1f failed passes) | sync %.2f | dots + sync %.2f\n", h->cp_launches, c[0] * 1e3 / khz / h->cp_launches, c[1] * 1e3 / khz / h->cp_launches,
                c[2] * 1e3 / khz / h->cp_launches, c[3] * 1e3 / khz / h->cp_launches, (double)c[7] / h->cp_launches, c[4] * 1e3 / khz / h->cp_launches,
                c[5] * 1e3 / khz / h->cp_launches);
        hipFree(h->cp_dbg);
    }
    if (h->xp_dbg) {                                      // (PS_XP_CLOCKS: the LAST launch's time stamps, averaged over its passes)
        std::vector<long long> c(3 * 64 * 8); hipMemcpy(c.data(), h->xp_dbg, c.size() * 8, hipMemcpyDeviceToHost);
        int khz = 100000; hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
        for (int w = 0; w < 3; ++w) {
            double ph[6] = {0, 0, 0, 0, 0, 0}; int np = 0;
            for (int p = 1; p < 64; ++p) {                   // (pass 0 = launch -1: no scalars yet; skipped)
                const long long* t = &c[((size_t)w * 64 + p) * 8];
                if (!t[6] || !t[4]) break;
                ph[0] += t[0] - t[6]; ph[1] += t[1] - t[0]; ph[2] += t[2] - t[1]; ph[3] += t[5] - t[2]; ph[4] += t[3] - t[5]; ph[5] += t[4] - t[3]; ++np;
            }
            const double u = 1e3 / khz / std::max(np, 1);
            fprintf(stderr, "k_xcg_persist, workgroup %s, last launch, %d passes, us per pass: t + coarse y %.2f | columns %.2f | products + publish %.2f | "
                    "first gather pass %.2f | rest of the gather %.2f | record sums + dots %.2f | sum %.2f\n", w == 0 ? "0" : (w == 1 ? "nwg/2" : "nwg-1"), np,
                    ph[0] * u, ph[1] * u, ph[2] * u, ph[3] * u, ph[4] * u, ph[5] * u, (ph[0] + ph[1] + ph[2] + ph[3] + ph[4] + ph[5]) * u);
        }
        hipFree(h->xp_dbg);
    }
    if (h->chk_sums) {
        hipDeviceSynchronize();
        fprintf(stderr, "PS_XCG_INV_SUM %p:", (void*)h);
        for (int k = 0; k < std::min(h->chk_nsum, 64); ++k) fprintf(stderr, " [%d use %016llx | side A_c %016llx -> inv %016llx]", k, h->chk_sums[3 * k], h->chk_sums[3 * k + 1], h->chk_sums[3 * k + 2]);
        fprintf(stderr, "\n");
        hipHostFree(h->chk_sums);
    }
    if (h->chk_cnt) {
        int32_t c[4] = {}; hipDeviceSynchronize(); hipMemcpy(c, h->chk_cnt, 16, hipMemcpyDeviceToHost);
        fprintf(stderr, "PS_XCG_AC_CHECK: %d comparisons, entries that differ between the side stream's assembly and the solver stream's: A_c %d, BSZ %d; A_c behind the assembly vs at the end of the side job: %d\n", c[2], c[0], c[1], c[3]);
        hipFree(h->chk_Ac_side); hipFree(h->chk_Ac); hipFree(h->chk_BSZ); hipFree(h->chk_cnt);
    }
    if (ps_env("PS_HOST_TIMING") && h->host_calls)
        fprintf(stderr, "ps_gn_iteration: %ld calls, %.1f us per call on the host, of which %.1f us waiting for the GPU (%ld waits)\n",
                h->host_calls, h->host_call_ns * 1e-3 / h->host_calls, h->host_wait_ns * 1e-3 / h->host_calls, h->host_waits);
    hipStreamSynchronize(h->stream);
    h->persist_release();
    if (h->side) hipStreamSynchronize(h->side);
    if (h->ldi_stream) hipStreamSynchronize(h->ldi_stream);
    for (void* p : h->allocs) hipFree(p);
    h->arena_release();            // arena block, its pinned mirror and the pinned result words go back to the process-wide pool
    for (hipEvent_t e : h->ev_pool) hipEventDestroy(e);
    if (h->ev_ldi) { hipEventDestroy(h->ev_ldi); hipEventDestroy(h->ev_ldi_sread); hipEventDestroy(h->ev_ldi_ritz); }
    if (h->aux) { hipStreamSynchronize(h->aux); if (!ps_pool().give(ps_pool().side_streams, h->aux)) hipStreamDestroy(h->aux); hipEventDestroy(h->ev_fork); hipEventDestroy(h->ev_join); }
    if (h->ldi_stream && !ps_pool().give(ps_pool().side_streams, h->ldi_stream)) hipStreamDestroy(h->ldi_stream);
    if (h->side) { hipStreamSynchronize(h->side); if (!h->side_poolable || !ps_pool().give(ps_pool().side_streams, h->side)) hipStreamDestroy(h->side); hipEventDestroy(h->ev_ac); hipEventDestroy(h->ev_chol); hipEventDestroy(h->ev_acdone); }
    if (h->own_stream && h->stream && !ps_pool().give(ps_pool().streams, h->stream)) hipStreamDestroy(h->stream);
    delete h;
    return 0;
}

// ps_problem_desc::flags & PS_DESC_DEVICE_TABLES: the small tables (groups, factors, index maps) are walked on the host, so they are
// brought over once (one D2H copy each, no pinned staging on the caller's side); the observation columns stay where they are when
// the structure is built on the device (csrc/ps_host_build.h reads them in place) and come over only for the host builder; the
// parameter tables (PS_DESC_DEVICE_PARAMS) never leave the device.
struct DescStage {
    ps_problem_desc d;
    std::vector<std::vector<char>> bufs;
    int pull_bytes(const void** field, size_t bytes) {
        if (!bytes || !*field) return 0;
        bufs.emplace_back(bytes);
        if (hipMemcpy(bufs.back().data(), *field, bytes, hipMemcpyDeviceToHost) != hipSuccess)
            return fail("ps_problem_create: copy of a device-resident table failed (PS_DESC_DEVICE_TABLES set on a host pointer?)");
        *field = bufs.back().data();
        return 0;
    }
    int pull(const double** f, size_t n) { return pull_bytes((const void**)f, n * sizeof(double)); }
    int pull(const int32_t** f, size_t n) { return pull_bytes((const void**)f, n * sizeof(int32_t)); }
    bool obs_pending = false;
    int pull_obs() {
        if (!obs_pending) return 0;
        obs_pending = false;
        const size_t N = (size_t)std::max<int64_t>(0, d.num_obs);
        return pull(&d.obs_pose, N) || pull(&d.obs_point, N) || pull(&d.obs_uvd, 3 * N) || pull(&d.obs_grp, N);
    }
    int stage(const ps_problem_desc* in) {
        d = *in;
        if (!(d.flags & PS_DESC_DEVICE_TABLES)) return 0;
        const size_t P = std::max(0, d.num_poses), L = std::max(0, d.num_points), N = (size_t)std::max<int64_t>(0, d.num_obs);
        const size_t E = (size_t)std::max<int64_t>(0, d.num_edges), Q = (size_t)std::max<int64_t>(0, d.num_priors);
        const size_t X = (size_t)std::max<int64_t>(0, d.num_extra_pairs), PW = d.dof == 6 ? 12 : 6, DD = (size_t)d.dof * d.dof;
        obs_pending = N > 0;       // the observation columns come over only if the HOST builder is going to walk them (pull_obs)
        return pull(&d.pose_rid, P) || pull(&d.point_vid, L) || pull(&d.cams, 5 * (size_t)std::max(0, d.num_cams)) || pull(&d.stiff3, 9 * (size_t)std::max(0, d.num_stiff3)) ||
               pull(&d.obs_groups, 4 * (size_t)std::max(0, d.num_obs_groups)) || pull(&d.e_i, E) || pull(&d.e_j, E) ||
               pull(&d.e_Tobs_inv, PW * E) || pull(&d.e_grp, E) || pull(&d.u_i, Q) || pull(&d.u_Tobs_inv, PW * Q) || pull(&d.u_grp, Q) ||
               pull(&d.stiffd, DD * (size_t)std::max(0, d.num_stiffd)) || pull(&d.edge_groups, 3 * (size_t)std::max(0, d.num_edge_groups)) ||
               pull(&d.extra_pair_i, X) || pull(&d.extra_pair_j, X);
    }
};

int ps_problem_create(const ps_problem_desc* d_in, void* stream, ps_problem** out) {
    const bool timing = ps_env("PS_CREATE_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "ps_problem_create: %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    if (!d_in || !out) return fail("null argument");
    *out = nullptr;
    if (d_in->flags & ~(uint32_t)(PS_DESC_DEVICE_PARAMS | PS_DESC_DEVICE_TABLES)) return fail("unknown bits in ps_problem_desc.flags");
    if (d_in->dof != 6 && d_in->dof != 3) return fail("dof must be 6 (SE3) or 3 (SE2)");
    if (d_in->num_obs > 0 && d_in->dof != 6) return fail("reprojection blocks need SE(3) poses");
    if (d_in->num_poses >= (1 << 24)) return fail("more than 2^24 poses");
    if (d_in->num_obs >= (1L << 31) / 18) return fail("too many observations for 32-bit indexing");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("no HIP device visible");
    DescStage staged;
    if (staged.stage(d_in)) return -1;
    const ps_problem_desc* d = &staged.d;
    const bool params_resident = (d->flags & PS_DESC_DEVICE_PARAMS) != 0;
    lap("device-resident tables");

    ps_problem* h = new ps_problem();
    struct Guard { ps_problem* h; bool ok = false; ~Guard() { if (!ok) ps_problem_destroy(h); } } guard{h};
    if (stream) h->stream = (hipStream_t)stream;
    else {
        if (!ps_pool().take(ps_pool().streams, &h->stream)) HIP_OK(hipStreamCreate(&h->stream));
        h->own_stream = true;
    }
    if (h->arena_begin()) return -1;
    const int D = h->D = d->dof;
    const int PW = h->PW = (D == 6 ? 12 : 6);
    const int DD = D * D;
    h->P = d->num_poses; h->L = d->num_points; h->N = d->num_obs;
    const int P = h->P, L = h->L;
    const long N = h->N;

    // ---- parameter tables
    if (h->upload(&h->poses, d->poses, (size_t)P * PW, params_resident)) return -1;
    if (h->upload(&h->points, d->points, (size_t)L * 3, params_resident)) return -1;
    if (h->upload(&h->pose_rid, d->pose_rid, (size_t)P)) return -1;
    if (h->alloc(&h->poses_snap, (size_t)P * PW) || h->alloc(&h->points_snap, (size_t)L * 3)) return -1;
    int nr = 0, nv = 0;
    for (int i = 0; i < P; ++i) if (d->pose_rid[i] >= 0) nr = std::max(nr, d->pose_rid[i] + 1);
    for (int i = 0; i < L; ++i) if (d->point_vid[i] >= 0) nv = std::max(nv, d->point_vid[i] + 1);
    h->nr = nr; h->nv = nv;
    std::vector<int32_t> point_of_vid(nv, -1);
    for (int i = 0; i < L; ++i) if (d->point_vid[i] >= 0) point_of_vid[d->point_vid[i]] = i;
    for (int v = 0; v < nv; ++v) if (point_of_vid[v] < 0) return fail("point_vid is not a dense 0..nv-1 numbering");
    // Internal landmark order ("slots"): by the lowest pose index that observes the landmark, so the
    // Z rows a pose (and a reduced-system block row) touches come from a compact address range and
    // stay in the 4 MB per-XCD L2, whatever order the caller numbered the landmarks in.
    // Structure build on the device (csrc/ps_host_build.h): the observation tables and the pair list are produced by kernels
    // from the caller's observation columns; the host builder below stays as its test oracle and for what the device build
    // does not cover.  PS_CREATE_DEVICE: 0 = host builder, 1 = device build from 200 000 observations up (default: below
    // that the host builder takes a fraction of a millisecond), 2 = always.
    const bool wide = d->num_obs_groups > 255;
    const int create_dev_env = ps_create_env("PS_CREATE_DEVICE") ? atoi(ps_create_env("PS_CREATE_DEVICE")) : 1;
    const bool dev_build = create_dev_env != 0 && D == 6 && N > 0 && nr > 0 && nv > 0 && !wide && (create_dev_env == 2 || N >= 200000) &&
                           !(ps_create_env("PS_SCHUR_MODE") && atoi(ps_create_env("PS_SCHUR_MODE")) != 0) &&
                           !(ps_create_env("PS_SCHUR_STREAM") && atoi(ps_create_env("PS_SCHUR_STREAM")) != 0) &&
                           !ps_create_env("PS_SCHUR_TILE_KB") && !ps_create_env("PS_PAIRS_BY_LANDMARK");
    if (!dev_build && staged.pull_obs()) return -1;
    std::vector<int32_t> first_pose(dev_build ? 0 : L, INT32_MAX);
    if (!dev_build)
    {
        const int T = ps_host_threads(N, 8);
        std::vector<std::vector<int32_t>> fp(T > 1 ? T : 0);
        ps_parallel(T, [&](int t, int TT) {
            std::vector<int32_t>& mine = TT > 1 ? fp[t] : first_pose;
            if (TT > 1) mine.assign(L, INT32_MAX);
            for (long i = N * t / TT, e = N * (t + 1) / TT; i < e; ++i) {
                const int pt = d->obs_point[i];
                if (pt >= 0 && pt < L) mine[pt] = std::min(mine[pt], d->obs_pose[i]);
            }
        });
        for (auto& v2 : fp) for (int i = 0; i < L; ++i) first_pose[i] = std::min(first_pose[i], v2[i]);
    }
    std::vector<int32_t>& vid_of_slot = h->h_vid_of_slot;
    vid_of_slot.resize(nv);
    if (!dev_build) {   // vids in ascending first pose, ties in vid order (what the stable sort over 500 000 landmarks gave: 40 ms at C4)
        std::vector<int32_t> fkey(nv), cnts;
        for (int v = 0; v < nv; ++v) { const int32_t f = first_pose[point_of_vid[v]]; fkey[v] = (f < 0 || f >= P) ? P : f; }
        parallel_index_sort(nv, (size_t)P + 1, fkey.data(), vid_of_slot.data(), cnts);
    }
    std::vector<int32_t> lm_point(dev_build ? 0 : nv), point_slot(dev_build ? 0 : L, -1);
    for (int s2 = 0; s2 < nv && !dev_build; ++s2) { lm_point[s2] = point_of_vid[vid_of_slot[s2]]; point_slot[lm_point[s2]] = s2; }
    {
        std::vector<char> seen(nr, 0);
        for (int i = 0; i < P; ++i) if (d->pose_rid[i] >= 0) {
            if (seen[d->pose_rid[i]]) return fail("pose_rid has duplicates");
            seen[d->pose_rid[i]] = 1;
        }
        for (int i = 0; i < nr; ++i) if (!seen[i]) return fail("pose_rid is not a dense 0..nr-1 numbering");
    }

    if (!dev_build && h->upload(&h->point_vid, point_slot)) return -1;     // device-side 'vid' = internal slot

    lap("parameter tables");
    // ---- observation groups.  Up to 255 rows (camera, stiffness, loss id, loss k): one device group per row.  More --
    // typically one stiffness per observation -- : the device groups are the distinct (camera, loss id, loss k) CLASSES
    // (at most 255 of those) and the stiffness travels as a per-observation index into the stiffness table ("wide").
    std::vector<int32_t> class_of_row(std::max(1, d->num_obs_groups), 0);
    std::vector<ObsGroup> og;
    {
        auto fill = [&](ObsGroup& o, const double* row, bool with_s) -> int {
            const int cam = (int)row[0], st = (int)row[1];
            if (cam < 0 || cam >= d->num_cams || st < 0 || st >= d->num_stiff3) return fail("obs group index out of range");
            const double* c = d->cams + 5 * cam;
            o.cu = c[0]; o.cv = c[1]; o.fu = c[2]; o.fv = c[3];
            o.cam_type = c[4] < 0.0 ? 1 : 0;            // cams row: baseline b >= 0 = stereo, b = -1 = RGB-D
            o.b = o.cam_type ? 0.0 : c[4];
            for (int k = 0; k < 9; ++k) o.S[k] = with_s ? d->stiff3[9 * st + k] : (k % 4 == 0 ? 1.0 : 0.0);
            o.loss_id = (int)row[2]; o.loss_k = row[3];
            return 0;
        };
        if (!wide) {
            og.resize(std::max(1, d->num_obs_groups));
            for (int gi = 0; gi < d->num_obs_groups; ++gi) { if (fill(og[gi], d->obs_groups + 4 * gi, true)) return -1; class_of_row[gi] = gi; }
        } else {
            std::vector<std::array<double, 3>> keys;        // (camera, loss id, loss k) of the classes found so far
            for (int gi = 0; gi < d->num_obs_groups; ++gi) {
                const double* row = d->obs_groups + 4 * gi;
                // every row's stiffness index is read on the device through the per-observation column (round-3 ADVICE: only
                // the first row of a class used to be range-checked)
                if ((int)row[1] < 0 || (int)row[1] >= d->num_stiff3) return fail("obs group index out of range");
                int c = -1;
                for (size_t q = 0; q < keys.size(); ++q)
                    if (keys[q][0] == row[0] && keys[q][1] == row[2] && keys[q][2] == row[3]) { c = (int)q; break; }
                if (c < 0) {
                    if (keys.size() == 255) return fail("more than 255 distinct (camera, loss) classes among the observation groups");
                    keys.push_back({row[0], row[2], row[3]});
                    og.emplace_back();
                    if (fill(og.back(), row, false)) return -1;
                    c = (int)keys.size() - 1;
                }
                class_of_row[gi] = c;
            }
            if (og.empty()) og.emplace_back();
            std::vector<double> st(d->stiff3, d->stiff3 + 9 * (size_t)d->num_stiff3);
            if (h->upload(&h->stiff_tab, st)) return -1;
        }
    }
    h->wide_obs = wide;
    if (h->upload(&h->ogroups, og)) return -1;

    lap("observation groups");
    // ---- observations sorted by landmark: variable points (by vid) first, then constant points
    if ((long)nv + (long)L + 1 >= (1L << 31)) return fail("too many landmarks for 32-bit sort keys");
    DevObsBuild devo;
    if (dev_build && (h->arena_flush() || devo.run(h, d, d_in, nr, nv, point_of_vid, vid_of_slot))) return -1;
    const size_t NH = dev_build ? 0 : (size_t)N;       // sizes of the host builder's tables
    std::vector<int32_t> order(NH), lkey(NH), lcounts;
    if (!dev_build) {   // keys (and the range checks the fill below relies on) on several threads
        std::atomic<int> bad{0};
        ps_parallel(ps_host_threads(N), [&](int t, int TT) {
            for (long i = N * t / TT, e = N * (t + 1) / TT; i < e; ++i) {
                const int pose = d->obs_pose[i], pt = d->obs_point[i], grp = d->obs_grp[i];
                if (pose < 0 || pose >= P || pt < 0 || pt >= L || grp < 0 || grp >= d->num_obs_groups) { bad = 1; lkey[i] = 0; continue; }
                const int v = point_slot[pt];
                lkey[i] = v >= 0 ? v : nv + pt;
            }
        });
        if (bad) return fail("observation index out of range");
    }
    if (!dev_build) parallel_index_sort(N, (size_t)nv + (size_t)L + 1, lkey.data(), order.data(), lcounts);
    std::vector<LObs> lobs(NH);
    std::vector<int32_t> lorig(NH), lm_ptr(nv + 1, 0), sidx_l(wide ? N : 0);
    long Nl = dev_build ? devo.Nl : 0;
    for (int v = 0; v < nv && !dev_build; ++v) { lm_ptr[v + 1] = lcounts[v]; Nl += lcounts[v]; }
    if (!dev_build) ps_parallel(ps_host_threads(N), [&](int t, int TT) {
        for (long k = N * t / TT, e = N * (t + 1) / TT; k < e; ++k) {
            const long i = order[k];
            const int pose = d->obs_pose[i], pt = d->obs_point[i], grp = d->obs_grp[i];
            LObs& o = lobs[k];
            o.u = d->obs_uvd[3 * i]; o.v = d->obs_uvd[3 * i + 1]; o.d = d->obs_uvd[3 * i + 2];
            o.pose_grp = (int32_t)((uint32_t)pose | ((uint32_t)class_of_row[grp] << 24));
            o.point = pt;
            lorig[k] = (int32_t)i;
            if (wide) sidx_l[k] = (int32_t)d->obs_groups[4 * (size_t)grp + 1];
        }
    });
    { std::vector<int32_t>().swap(lkey); std::vector<int32_t>().swap(lcounts); }
    for (int v = 0; v < nv; ++v) lm_ptr[v + 1] += lm_ptr[v];
    h->Nl = Nl;
    if (!dev_build) if (h->upload(&h->lobs, lobs) || h->upload(&h->lorig, lorig) || h->upload(&h->lm_ptr, lm_ptr) ||
        h->upload(&h->lm_point, lm_point) || (wide && h->upload(&h->sidx_l, sidx_l))) return -1;
    if (h->alloc(&h->Z, (size_t)Nl * PS_ZROW) || h->alloc(&h->Cinv, (size_t)nv * 6) ||
        h->alloc(&h->cvec, (size_t)nv * 3) || h->alloc(&h->dxl, (size_t)nv * 3)) return -1;
    if (h->zero(h->dxl, (size_t)nv * 3 * sizeof(double))) return -1;

    lap("landmark sort + lobs");
    // ---- pose segments (observations on variable poses), chunks of 256
    std::vector<int32_t> pcount(nr + 1, 0);
    std::vector<int32_t> pidx;
    if (dev_build) pcount = devo.pcount;
    else {   // observations by reduced pose (constant poses: key nr, cut off afterwards), stable in landmark order
        std::vector<int32_t> pkey((size_t)N), pord((size_t)N), pcnt;
        ps_parallel(ps_host_threads(N), [&](int t, int TT) {
            for (long k = N * t / TT, e = N * (t + 1) / TT; k < e; ++k) { const int r = d->pose_rid[PS_POSE_OF(lobs[k])]; pkey[k] = r >= 0 ? r : nr; }
        });
        parallel_index_sort(N, (size_t)nr + 1, pkey.data(), pord.data(), pcnt);
        for (int r = 0; r < nr; ++r) pcount[r + 1] = pcount[r] + pcnt[r];
        pord.resize((size_t)pcount[nr]);
        pidx.swap(pord);
    }
    const long Np = h->Np = pcount[nr];
    for (int r = 0; r < nr; ++r) h->max_pose_obs = std::max(h->max_pose_obs, pcount[r + 1] - pcount[r]);
    std::vector<PItem> pitems;
    std::vector<int32_t> pitem_ptr(nr + 1, 0);
    std::vector<int32_t> pose_of_rid(std::max(nr, 1), 0);
    for (int p = 0; p < P; ++p) if (d->pose_rid[p] >= 0) pose_of_rid[d->pose_rid[p]] = p;
    if (h->upload(&h->pose_of_rid, pose_of_rid)) return -1;
    // chunk of observations per workgroup: 512 (two per thread) once that still fills the chip
    // (measured with the LDS-transposed reduction of k_pose_pass: 256 / 512 / 1024 observations per workgroup give
    //  28.3 / 25.9 / 28.6 us at C3 and 0.218 / 0.192 / 0.202 ms at C4)
    int pchunk = Np >= 512L * 256 ? 512 : 256;
    if (const char* e = ps_env("PS_POSE_CHUNK")) pchunk = std::max(256, atoi(e) / 256 * 256);
    for (int r = 0; r < nr; ++r) {
        for (int s = pcount[r]; s < pcount[r + 1]; s += pchunk)
            pitems.push_back({r, s, std::min(s + pchunk, pcount[r + 1]), pose_of_rid[r]});
        pitem_ptr[r + 1] = (int32_t)pitems.size();
    }
    // pose-sorted copy of the observation records; the pose bits (uniform per chunk) carry the landmark slot + 1
    if (nv >= (1 << 24) - 1) return fail("too many variable landmarks for the 24-bit slot field");
    std::vector<LObs> pobs(dev_build ? 0 : (size_t)Np);
    if (!dev_build) ps_parallel(ps_host_threads(Np), [&](int t, int TT) {
        for (long k = Np * t / TT, e = Np * (t + 1) / TT; k < e; ++k) {
            pobs[k] = lobs[pidx[k]];
            const int slot = point_slot[pobs[k].point];          // -1: constant point
            pobs[k].pose_grp = (int32_t)(((uint32_t)PS_GRP_OF(pobs[k]) << 24) | (uint32_t)(slot + 1));
        }
    });
    if (wide) {
        std::vector<int32_t> sidx_p((size_t)Np);
        for (long k = 0; k < Np; ++k) sidx_p[k] = sidx_l[pidx[k]];
        if (h->upload(&h->sidx_p, sidx_p)) return -1;
    }
    h->npitems = (int)pitems.size();
    if (h->upload(&h->pitems, pitems) || h->upload(&h->pitem_ptr, pitem_ptr) ||
        (!dev_build && h->upload(&h->pobs, pobs)) || h->alloc(&h->ppartial, (size_t)pitems.size() * PS_NPOSE_ACC)) return -1;

    lap("pose segments + pobs");
    // ---- pose factors: edges then priors
    const long E = d->num_edges, Q = d->num_priors, F = h->F = E + Q;
    std::vector<FactorGroup> fg(std::max(1, d->num_edge_groups));
    for (int gi = 0; gi < d->num_edge_groups; ++gi) {
        const double* row = d->edge_groups + 3 * gi;
        const int st = (int)row[0];
        if (st < 0 || st >= d->num_stiffd) return fail("edge group stiffness index out of range");
        std::memset(&fg[gi], 0, sizeof(FactorGroup));
        for (int k = 0; k < DD; ++k) fg[gi].S[k] = d->stiffd[(size_t)DD * st + k];
        fg[gi].loss_id = (int)row[1]; fg[gi].loss_k = row[2];
    }
    std::vector<int32_t> f_i(F), f_j(F), f_grp(F);
    std::vector<double> f_T((size_t)F * PW);
    for (long f = 0; f < E; ++f) {
        f_i[f] = d->e_i[f]; f_j[f] = d->e_j[f]; f_grp[f] = d->e_grp[f];
        std::memcpy(&f_T[(size_t)f * PW], d->e_Tobs_inv + (size_t)f * PW, PW * sizeof(double));
    }
    for (long u = 0; u < Q; ++u) {
        f_i[E + u] = -1; f_j[E + u] = d->u_i[u]; f_grp[E + u] = d->u_grp[u];
        std::memcpy(&f_T[(size_t)(E + u) * PW], d->u_Tobs_inv + (size_t)u * PW, PW * sizeof(double));
    }
    for (long f = 0; f < F; ++f)
        if (f_j[f] < 0 || f_j[f] >= P || f_i[f] >= P || f_grp[f] < 0 || f_grp[f] >= d->num_edge_groups)
            return fail("pose factor index out of range");
    if (h->upload(&h->fgroups, fg) || h->upload(&h->f_i, f_i) || h->upload(&h->f_j, f_j) ||
        h->upload(&h->f_grp, f_grp) || h->upload(&h->f_Tinv, f_T)) return -1;
    const int FROW = 3 * DD + 2 * D;
    if (h->alloc(&h->fscratch, (size_t)F * FROW)) return -1;

    lap("pose factors");
    // ---- pose-stationary Schur lists (round 4; kernel k_schur_pose, csrc/ps_k_schur3.h; NOT the default: slower, see below).
    // A workgroup takes a SEGMENT of up to PS_PP_SEG rows of ONE pose i -- its Z rows on variable landmarks, in landmark
    // order -- brings them into LDS once, and its waves work through the tasks (segment, partner pose j > i): the pairs
    // (a = local index of pose i's row, b = the Z row of pose j on the same landmark).  Only the partner rows are gathered:
    // 0.5 M + 2.25 M row fetches at C3 instead of 4.5 M.  One partial block per task, summed in segment order by
    // k_schur_combine (fixed order).  Not built when a landmark is observed twice from one pose (a diagonal-block task: the
    // gather kernels handle that) or when a Z row index does not fit 23 bits.
    std::vector<uint64_t> pose_keys;              // upper block keys (ri < rj) that receive pairs, unsorted, with repeats
    std::vector<PoseSeg> psegs;
    std::vector<int32_t> pseg_rows;
    std::vector<PairItem> ptasks;                 // slot / slotT are filled once the block pattern exists
    std::vector<uint64_t> ptask_key;
    std::vector<uint32_t> ppairs;
    std::vector<long> pseg_pairs;                 // pairs per segment (XCD balancing)
    // Measured (tools/schur_probe.py): 0.121 ms at C3 against 0.046 for the pipelined gather kernel, 1.01 ms
    // against 0.515 at C4 -- a workgroup needs 157 KB of LDS (the segment + two chunk buffers per wave), so ONE is resident per
    // CU and the dependent loads of its start-up (order -> segment -> row indices -> rows -> pair words -> partner rows) are
    // covered by nothing; with products, fetches, fill and stores all removed the skeleton still takes 0.06 ms.  Kept as an
    // independent variant (parity-tested against the gather kernels and the oracle); built with PS_SCHUR_MODE=1 (this
    // kernel only) or 2 (both list sets: option "schur_mode" switches).
    const int schur_mode_env = ps_create_env("PS_SCHUR_MODE") ? atoi(ps_create_env("PS_SCHUR_MODE")) : 0;
    bool pose_mode = schur_mode_env != 0 && D == 6 && nv > 0 && nr > 0 && Nl > 0 && Nl < (1L << 23);
    if (pose_mode) {
        std::vector<int32_t> rid_row((size_t)Nl), lm_row((size_t)Nl);
        for (int v = 0; v < nv; ++v)
            for (int a = lm_ptr[v]; a < lm_ptr[v + 1]; ++a) { lm_row[a] = v; rid_row[a] = d->pose_rid[PS_POSE_OF(lobs[a])]; }
        struct Tri { int32_t rb, al, b; };
        struct Part {                                   // what one host thread produces for its range of poses
            std::vector<PoseSeg> segs; std::vector<int32_t> rows; std::vector<PairItem> tasks; std::vector<uint64_t> keys;
            std::vector<uint32_t> pairs; std::vector<long> seg_pairs; bool diag = false;
        };
        const int nth = std::max(1, std::min({(int)std::thread::hardware_concurrency(), 16, nr}));
        std::vector<Part> parts(nth);
        auto work = [&](int th) {
            Part& o = parts[th];
            const int r0 = (int)((long)nr * th / nth), r1 = (int)((long)nr * (th + 1) / nth);
            std::vector<Tri> tri, sorted;
            std::vector<int32_t> cnt((size_t)nr + 1, 0), touched;
            for (int r = r0; r < r1; ++r) {
                // pose r's Z rows (variable landmarks only), ascending
                int s0 = pcount[r];
                const int s_end = pcount[r + 1];
                while (s0 < s_end) {
                    PoseSeg sg{(int32_t)o.rows.size(), 0, (int32_t)o.tasks.size(), 0};
                    tri.clear();
                    int al = 0;
                    for (; s0 < s_end && al < PS_PP_SEG - 1; ++s0) {
                        const int k = pidx[s0];
                        if (k >= Nl) continue;                   // an observation of a constant point: no Z row
                        o.rows.push_back(k);
                        const int v = lm_row[k];
                        for (int b = lm_ptr[v]; b < lm_ptr[v + 1]; ++b) {
                            const int rb = rid_row[b];
                            if (rb > r) tri.push_back({rb, al, b});
                            else if (rb == r && b != k) o.diag = true;
                        }
                        ++al;
                    }
                    sg.row_count = al;
                    // tasks: the pairs by partner (stable counting sort: pairs of a task stay in landmark order)
                    touched.clear();
                    for (const Tri& t : tri) { if (cnt[t.rb]++ == 0) touched.push_back(t.rb); }
                    std::sort(touched.begin(), touched.end());
                    int32_t run = 0;
                    for (int32_t rb : touched) { const int32_t c = cnt[rb]; cnt[rb] = run; run += c; }
                    sorted.resize(tri.size());
                    for (const Tri& t : tri) sorted[cnt[t.rb]++] = t;
                    // every task's words are padded to a multiple of 32 (one product pass of the kernel); a padding word carries
                    // the reserved a-row index and the task's last partner row (a harmless fetch)
                    int32_t prev = 0;
                    for (int32_t rb : touched) {
                        const int32_t end = cnt[rb];
                        const int32_t w0 = (int32_t)o.pairs.size();
                        for (int32_t q = prev; q < end; ++q) o.pairs.push_back(((uint32_t)sorted[q].al << 23) | (uint32_t)sorted[q].b);
                        while ((o.pairs.size() - (size_t)w0) % 32) o.pairs.push_back(((uint32_t)511u << 23) | (uint32_t)sorted[end - 1].b);
                        o.tasks.push_back({-1, -1, w0, (int32_t)o.pairs.size()});
                        o.keys.push_back(((uint64_t)r << 32) | (uint32_t)rb);
                        prev = end;
                        cnt[rb] = 0;
                    }
                    sg.task_end = (int32_t)o.tasks.size();
                    if (al > 0) { o.segs.push_back(sg); o.seg_pairs.push_back((long)tri.size()); }
                    else { o.rows.resize(sg.row_start); }
                }
            }
        };
        if (nth == 1) work(0);
        else {
            std::vector<std::thread> pool;
            for (int th = 0; th < nth; ++th) pool.emplace_back(work, th);
            for (auto& t : pool) t.join();
        }
        size_t tot_pairs = 0, tot_tasks = 0;
        for (const Part& o : parts) { pose_mode = pose_mode && !o.diag; tot_pairs += o.pairs.size(); tot_tasks += o.tasks.size(); }
        if (tot_pairs >= (1UL << 31) || tot_tasks >= (1UL << 31) / 36) pose_mode = false;
        if (pose_mode) {
            for (const Part& o : parts) {                   // concatenate in pose order, rebasing the offsets
                const int32_t rb = (int32_t)pseg_rows.size(), tb = (int32_t)ptasks.size(), pb = (int32_t)ppairs.size();
                for (PoseSeg sg : o.segs) { sg.row_start += rb; sg.task_start += tb; sg.task_end += tb; psegs.push_back(sg); }
                for (PairItem t : o.tasks) { t.start += pb; t.end += pb; ptasks.push_back(t); }
                pseg_rows.insert(pseg_rows.end(), o.rows.begin(), o.rows.end());
                ptask_key.insert(ptask_key.end(), o.keys.begin(), o.keys.end());
                ppairs.insert(ppairs.end(), o.pairs.begin(), o.pairs.end());
                pseg_pairs.insert(pseg_pairs.end(), o.seg_pairs.begin(), o.seg_pairs.end());
            }
            pose_keys = ptask_key;
        }
    }
    const bool gather_lists = !pose_mode || schur_mode_env == 2;      // the pair lists of the gather kernels (k_schur_pairs[_db])
    lap("pose-stationary lists");
    // ---- Schur pairs per landmark (upper-triangle block keys)
    // Landmark tiles: when Z (144 B per row) is much larger than the eight 4 MB L2s, the pair list is
    // cut into tiles of consecutive landmarks (consecutive Z rows) and ONE XCD works through a whole
    // tile: a Z row is then only ever requested by one L2 instead of by up to eight.  A block that
    // receives pairs from several tiles gets one partial per (tile, block) task, summed in tile order
    // by k_schur_combine (fixed order => deterministic).  Measured at C3 (72 MB of Z): 8 tiles of 9 MB
    // beat both no tiling (71 -> 65 us) and L2-sized 2-4 MB tiles (71-82 us: five times more tasks,
    // and the per-task prologue/epilogue costs more than the extra L2 hits save).
    std::vector<PairRec> prs;
    int ntiles = 1;
    bool tiles_forced = ps_create_env("PS_SCHUR_TILE_KB") != nullptr;
    const double zbytes = 8.0 * PS_ZROW * (double)Nl;
    {
        double tile_kb = 9216.0, min_mb = 16.0;
        if (const char* e = ps_create_env("PS_SCHUR_TILE_KB")) tile_kb = atof(e);
        if (const char* e = ps_create_env("PS_SCHUR_TILE_MIN_MB")) min_mb = atof(e);
        if (tile_kb > 0 && zbytes > min_mb * 1048576.0)
            ntiles = 8 * (int)std::ceil(zbytes / (8.0 * tile_kb * 1024.0));
    }
    // Device build of the pair list (csrc/ps_host_build.h): pairs generated, sorted and left in HBM; the host receives the task
    // starts and keys only.
    const bool dev_pairs = gather_lists && dev_build && ntiles < 65536;
    // (a device build leaves no observation tables on the host: the host pair loop below would see empty lists and the reduced
    //  system would silently lose every landmark coupling -- round-4 ADVICE.  65 536 tiles = 590 GB of Z rows: not reachable)
    if (dev_build && gather_lists && !dev_pairs) return fail("too many landmark tiles for the device pair build");
    DevPairBuild devb;
    std::vector<long> lm_pairs_before(nv + 1, 0);
    if (dev_pairs) {
        if (h->arena_flush() || devb.prepare(h, nv, nr, Nl)) return -1;
    } else {
        for (int v = 0; v < nv && gather_lists; ++v) {
            long nvar = 0;
            for (int a = lm_ptr[v]; a < lm_ptr[v + 1]; ++a) nvar += d->pose_rid[PS_POSE_OF(lobs[a])] >= 0;
            lm_pairs_before[v + 1] = lm_pairs_before[v] + nvar * (nvar - 1) / 2;
        }
    }
    const long total_pairs = dev_pairs ? (long)devb.total : (gather_lists ? lm_pairs_before[nv] : 0);
    if (total_pairs >= (1L << 31)) return fail("too many Schur pairs for 32-bit indexing");
    if (dev_pairs) {
        // the same decisions as the host loop below, on lists that are never on the host
        if (h->alloc(&h->pairs, (size_t)total_pairs, true)) return -1;
        uint64_t* pout = reinterpret_cast<uint64_t*>(h->pairs);
        const bool keep_env = ps_env("PS_SCHUR_KEEP_TILES") != nullptr;
        bool done = false;
        if (ntiles > 1 && zbytes <= 128.0 * 1048576.0 && !keep_env) {
            if (devb.build(1, pout)) return -1;
            if (devb.task_start.size() >= 2 * 256 * 12) { ntiles = 1; done = true; }
        }
        if (!done && ntiles == 1) { if (devb.build(1, pout)) return -1; done = true; }
        if (!done) {
            if (devb.build(ntiles, pout)) return -1;
            if (total_pairs > 0) {
                std::vector<uint64_t> tk(devb.task_key);
                std::sort(tk.begin(), tk.end());
                const size_t nblocks = std::unique(tk.begin(), tk.end()) - tk.begin(), ntask = devb.task_start.size();
                const bool blocks_fill_chip = nblocks >= 2 * 256 * 12 && zbytes <= 128.0 * 1048576.0 && !keep_env;
                if (!((size_t)total_pairs >= 64 * ntask && !blocks_fill_chip)) { ntiles = 1; if (devb.build(1, pout)) return -1; }
            }
        }
    }
    // one unit of work per tile: the tile's pairs in (block row, block column, landmark) order, written to its
    // slice of prs; tiles are independent, so they are built by a few host threads
    // Tiles multiply the tasks (one per (tile, block) with pairs): when a task is left with a handful of pairs -- a landmark
    // shard of a multi-GPU run: 36 pairs per block at C4 / 8 -- the per-task skeleton dominates and the
    // untiled list wins (C4 / 8 shard: Schur stage 0.159 -> 0.119 ms).  Decided on the generated list: fewer than 64 pairs
    // per task on average -> generated again without tiles.
    for (int attempt = 0; attempt < 2 && gather_lists && !dev_pairs; ++attempt) {
        auto tile_of = [&](int v) {
            return (ntiles > 1 && total_pairs > 0)
                ? (int)std::min<long>(ntiles - 1, (long)((double)ntiles * (double)lm_pairs_before[v] / (double)total_pairs)) : 0;
        };
        std::vector<int> tile_begin(ntiles + 1, nv);
        {
            int t_prev = -1;
            for (int v = 0; v < nv; ++v) {
                const int t = tile_of(v);
                for (int q = t_prev + 1; q <= t; ++q) tile_begin[q] = v;
                t_prev = std::max(t_prev, t);
            }
            tile_begin[ntiles] = nv;
            for (int q = ntiles - 1; q >= 0; --q) tile_begin[q] = std::min(tile_begin[q], tile_begin[q + 1]);
        }
        prs.resize((size_t)total_pairs);
        // One tile (the untiled list: C3): pose by pose on several host threads.  The pairs of pose r -- its rows a in landmark
        // order against the rows b of the same landmark whose pose comes later (or the same pose again, b > a: a diagonal block)
        // -- sorted by partner (stable) ARE the (block row, block column, landmark) order the two counting passes over the whole
        // list produce, so the threads' outputs are simply concatenated (C3: 70 ms -> 7 ms of ps_problem_create).
        auto build_untiled_by_pose = [&]() {
            std::vector<int32_t> rid_row((size_t)lm_ptr[nv]), lm_row((size_t)lm_ptr[nv]);
            for (int v = 0; v < nv; ++v)
                for (int a = lm_ptr[v]; a < lm_ptr[v + 1]; ++a) { lm_row[a] = v; rid_row[a] = d->pose_rid[PS_POSE_OF(lobs[a])]; }
            const long nrows = lm_ptr[nv];
            const int nth = std::max(1, std::min({(int)std::thread::hardware_concurrency(), 16, nr}));
            std::vector<std::vector<PairRec>> outs(nth);
            auto work = [&](int th) {
                std::vector<PairRec>& o = outs[th];
                const int r0 = (int)((long)nr * th / nth), r1 = (int)((long)nr * (th + 1) / nth);
                std::vector<PairRec> tri;
                std::vector<int32_t> cnt((size_t)nr + 1, 0), touched;
                for (int r = r0; r < r1; ++r) {
                    tri.clear();
                    for (int s0 = pcount[r]; s0 < pcount[r + 1]; ++s0) {
                        const int k = pidx[s0];
                        if (k >= nrows) continue;                // an observation of a constant point: no Z row
                        const int v = lm_row[k];
                        for (int b = lm_ptr[v]; b < lm_ptr[v + 1]; ++b) {
                            const int rb = rid_row[b];
                            if (rb > r || (rb == r && b > k)) tri.push_back({((uint64_t)r << 32) | (uint32_t)rb, k, b, 0});
                        }
                    }
                    touched.clear();
                    for (const PairRec& t : tri) { if (cnt[(uint32_t)t.key]++ == 0) touched.push_back((int32_t)(uint32_t)t.key); }
                    std::sort(touched.begin(), touched.end());
                    int32_t run = 0;
                    for (int32_t rb : touched) { const int32_t c = cnt[rb]; cnt[rb] = run; run += c; }
                    const size_t base = o.size();
                    o.resize(base + tri.size());
                    for (const PairRec& t : tri) o[base + cnt[(uint32_t)t.key]++] = t;
                    for (int32_t rb : touched) cnt[rb] = 0;
                }
            };
            if (nth == 1) work(0);
            else {
                std::vector<std::thread> pool;
                for (int th = 0; th < nth; ++th) pool.emplace_back(work, th);
                for (auto& t : pool) t.join();
            }
            size_t at = 0;
            for (auto& o : outs) { std::copy(o.begin(), o.end(), prs.begin() + at); at += o.size(); }
            prs.resize(at);
        };
        // When all of Z stays in the Infinity Cache the untiled list is the likely winner (the rule below): build IT first and
        // keep it if the blocks alone fill the chip -- no tiled list is generated only to be thrown away (C3: 23 -> 7 ms)
        const bool by_pose = nr > 0 && !ps_create_env("PS_PAIRS_BY_LANDMARK");
        if (by_pose && attempt == 0 && ntiles > 1 && !tiles_forced && zbytes <= 128.0 * 1048576.0 && !ps_env("PS_SCHUR_KEEP_TILES")) {
            build_untiled_by_pose();
            size_t nblocks0 = 0;
            for (size_t k = 0; k < prs.size(); ++k) nblocks0 += (k == 0 || prs[k].key != prs[k - 1].key);
            if (nblocks0 >= 2 * 256 * 12) { ntiles = 1; break; }
            prs.resize((size_t)total_pairs);
        }
        if (ntiles == 1 && by_pose) { build_untiled_by_pose(); break; }
        auto build_tile = [&](int tile) {
            const int v0 = tile_begin[tile], v1 = tile_begin[tile + 1];
            std::vector<PairRec> loc;
            loc.reserve((size_t)(lm_pairs_before[v1] - lm_pairs_before[v0]));
            for (int v = v0; v < v1; ++v)
                for (int a = lm_ptr[v]; a < lm_ptr[v + 1]; ++a) {
                    const int ra = d->pose_rid[PS_POSE_OF(lobs[a])];
                    if (ra < 0) continue;
                    for (int b = a + 1; b < lm_ptr[v + 1]; ++b) {
                        const int rb = d->pose_rid[PS_POSE_OF(lobs[b])];
                        if (rb < 0) continue;
                        if (ra <= rb) loc.push_back({((uint64_t)ra << 32) | (uint32_t)rb, a, b, tile});
                        else loc.push_back({((uint64_t)rb << 32) | (uint32_t)ra, b, a, tile});
                    }
                }
            // (block row, block column): two stable counting passes, least significant first
            counting_sort(loc, (size_t)std::max(nr, 1), [](const PairRec& x) { return (uint32_t)x.key; });
            counting_sort(loc, (size_t)std::max(nr, 1), [](const PairRec& x) { return (uint32_t)(x.key >> 32); });
            std::copy(loc.begin(), loc.end(), prs.begin() + lm_pairs_before[v0]);
        };
        {
            const int nthreads = std::max(1, std::min({ntiles, 16, (int)std::thread::hardware_concurrency()}));
            if (nthreads <= 1) {
                for (int t = 0; t < ntiles; ++t) build_tile(t);
            } else {
                std::atomic<int> next{0};
                std::vector<std::thread> pool;
                for (int k = 0; k < nthreads; ++k)
                    pool.emplace_back([&] { for (int t = next++; t < ntiles; t = next++) build_tile(t); });
                for (auto& th : pool) th.join();
            }
        }
        if (ntiles == 1 || tiles_forced || prs.empty()) break;
        size_t ntask = 0;
        std::vector<uint64_t> task_keys;
        for (size_t k = 0; k < prs.size(); ++k)
            if (k == 0 || prs[k].key != prs[k - 1].key || prs[k].tile != prs[k - 1].tile) { ++ntask; task_keys.push_back(prs[k].key); }
        std::sort(task_keys.begin(), task_keys.end());
        const size_t nblocks = std::unique(task_keys.begin(), task_keys.end()) - task_keys.begin();
        // Tiles also buy parallelism (tasks) and L2 locality, and cost a combine launch, a partial per task and short tasks.
        // When the blocks alone fill the chip twice over (2 x 256 CUs x 12 waves of the pipelined pair kernel) and all of Z
        // stays in the 256 MiB Infinity Cache, the untiled list wins (C3: stage 56 -> 48 us); at C4 (Z = 640 MB) the tiles'
        // locality is worth 0.53 against 0.77 ms.
        const bool blocks_fill_chip = nblocks >= 2 * 256 * 12 && zbytes <= 128.0 * 1048576.0 && !ps_env("PS_SCHUR_KEEP_TILES");
        if (prs.size() >= 64 * ntask && !blocks_fill_chip) break;
        ntiles = 1;
    }
    if (!gather_lists) ntiles = 1;
    h->schur_tiles = ntiles;
    const long npairs_l = dev_pairs ? total_pairs : (long)prs.size();
    h->npairs = gather_lists ? npairs_l : (long)ppairs.size();
    if (prs.size() >= (1UL << 31)) return fail("too many Schur pairs for 32-bit indexing");

    lap("pair generation + sort");
    // tasks: runs of equal (tile, block) in the pair list -- start, block key, tile.  The device build has them already; the host
    // build finds them (and copies its records into the kernels' int2 form) on several threads
    std::vector<int32_t> task_start, task_tile;
    std::vector<uint64_t> task_keyv;
    std::vector<int2> pairs(dev_pairs ? 0 : prs.size());
    if (dev_pairs) {
        task_start.swap(devb.task_start); task_tile.swap(devb.task_tile); task_keyv.swap(devb.task_key);
    } else {
        const long np = (long)prs.size();
        const int T = ps_host_threads(np);
        std::vector<std::vector<int32_t>> starts(T);
        ps_parallel(T, [&](int t, int TT) {
            for (long k = np * t / TT, e = np * (t + 1) / TT; k < e; ++k) {
                pairs[k] = make_int2(prs[k].a, prs[k].b);
                if (k == 0 || prs[k].key != prs[k - 1].key || prs[k].tile != prs[k - 1].tile) starts[t].push_back((int32_t)k);
            }
        });
        size_t ntask = 0;
        for (auto& v2 : starts) ntask += v2.size();
        task_start.reserve(ntask); task_tile.reserve(ntask); task_keyv.reserve(ntask);
        for (auto& v2 : starts)
            for (int32_t k : v2) { task_start.push_back(k); task_tile.push_back(prs[k].tile); task_keyv.push_back(prs[k].key); }
        prs.clear(); prs.shrink_to_fit();
    }
    // ---- block pattern of the reduced system
    std::vector<uint64_t> keys;                 // upper keys (ri <= rj)
    keys.reserve(task_keyv.size() + nr + F + d->num_extra_pairs);
    for (int r = 0; r < nr; ++r) keys.push_back(((uint64_t)r << 32) | (uint32_t)r);
    keys.insert(keys.end(), task_keyv.begin(), task_keyv.end());
    keys.insert(keys.end(), pose_keys.begin(), pose_keys.end());
    for (long f = 0; f < E; ++f) {
        const int ra = d->pose_rid[f_i[f]], rb = d->pose_rid[f_j[f]];
        if (ra >= 0 && rb >= 0 && ra != rb)
            keys.push_back(((uint64_t)std::min(ra, rb) << 32) | (uint32_t)std::max(ra, rb));
    }
    for (long k = 0; k < d->num_extra_pairs; ++k) {
        const int ra = d->extra_pair_i[k], rb = d->extra_pair_j[k];
        if (ra < 0 || rb < 0 || ra >= nr || rb >= nr) return fail("extra pair index out of range");
        keys.push_back(((uint64_t)std::min(ra, rb) << 32) | (uint32_t)std::max(ra, rb));
    }
    std::sort(keys.begin(), keys.end());
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    std::vector<int32_t>& row_ptr = h->h_row_ptr;
    std::vector<int32_t>& col_idx = h->h_col_idx;
    row_ptr.assign(nr + 1, 0);
    for (uint64_t k : keys) {
        const int a = (int)(k >> 32), b = (int)(uint32_t)k;
        row_ptr[a + 1]++;
        if (a != b) row_ptr[b + 1]++;
    }
    for (int r = 0; r < nr; ++r) row_ptr[r + 1] += row_ptr[r];
    const long nnzb_l = row_ptr[nr];
    if (nnzb_l * DD >= (1L << 31)) return fail("reduced system too large for 32-bit block offsets");
    const int nnzb = h->nnzb = (int)nnzb_l;
    col_idx.assign(nnzb, 0);
    {
        std::vector<int32_t> f2(row_ptr.begin(), row_ptr.end() - 1);
        // lower part first needs sorted columns per row: insert (b,a) pairs in key order gives
        // ascending a for row b; then (a,b) gives ascending b >= a.  Do two passes.
        for (uint64_t k : keys) { const int a = (int)(k >> 32), b = (int)(uint32_t)k; if (a != b) col_idx[f2[b]++] = a; }
        for (uint64_t k : keys) { const int a = (int)(k >> 32), b = (int)(uint32_t)k; col_idx[f2[a]++] = b; }
    }
    auto slot_of = [&](int a, int b) -> int {
        const int32_t* lo = col_idx.data() + row_ptr[a];
        const int32_t* hi = col_idx.data() + row_ptr[a + 1];
        const int32_t* it = std::lower_bound(lo, hi, b);
        return (it != hi && *it == b) ? (int)(it - col_idx.data()) : -1;
    };
    std::vector<int32_t> diag_slot(nr);
    for (int r = 0; r < nr; ++r) diag_slot[r] = slot_of(r, r);
    if (h->upload(&h->row_ptr, row_ptr) || h->upload(&h->col_idx, col_idx) || h->upload(&h->diag_slot, diag_slot)) return -1;
    h->red_count = (long)nnzb * DD + (long)nr * D + 2;
    if (h->alloc(&h->red, (size_t)h->red_count + ST_NWORDS / 2)) return -1;       // + the status words: one memset clears both
    h->status = reinterpret_cast<int32_t*>(h->red + h->red_count);
    h->S = h->red; h->g = h->red + (size_t)nnzb * DD; h->red_cost = h->g + (size_t)nr * D;

    lap("block pattern");
    // one work item (task) per (tile, block) that has pairs
    std::vector<PairItem> pitm(task_start.size());
    {
        const long nt = (long)pitm.size();
        std::atomic<int> diag{0};
        ps_parallel(ps_host_threads(nt * 16), [&](int t, int TT) {
            for (long q = nt * t / TT, e = nt * (t + 1) / TT; q < e; ++q) {
                const uint64_t key = task_keyv[q];
                const int a = (int)(key >> 32), b = (int)(uint32_t)key;
                pitm[q] = {slot_of(a, b), slot_of(b, a), task_start[q], q + 1 < nt ? task_start[q + 1] : (int32_t)npairs_l};
                if (a == b) diag = 1;
            }
        });
        if (diag) h->has_diag_tasks = true;
    }
    h->npair_items = (int)pitm.size();
    if (!dev_pairs && h->upload(&h->pairs, pairs)) return -1;
    {   // per-XCD work lists.  Untiled: items are sorted by block row, so equal contiguous shares of
        // the PAIRS (not of the items) give each XCD a contiguous range of block rows with balanced
        // work.  Tiled: XCD x takes tiles x, x + 8, ... (tiles hold equal pair counts).
        std::vector<std::vector<int32_t>> lists(8);
        const double total = (double)npairs_l;
        for (size_t k = 0; k < pitm.size(); ++k) {
            const int x = ntiles > 1 ? (task_tile[k] & 7)
                                     : (total > 0 ? std::min(7, (int)(8.0 * pitm[k].start / total)) : 0);
            lists[x].push_back((int32_t)k);
        }
        // longest tasks first (within each tile): the short ones fill the tail of the XCD's schedule
        if (!ps_env("PS_SCHUR_NO_LPT"))
            for (auto& l : lists)
                std::stable_sort(l.begin(), l.end(), [&](int32_t x, int32_t y) {
                    if (task_tile[x] != task_tile[y]) return task_tile[x] < task_tile[y];
                    return pitm[x].end - pitm[x].start > pitm[y].end - pitm[y].start; });
        size_t mx = 0;
        for (auto& l : lists) mx = std::max(mx, l.size());
        mx = std::max<size_t>((mx + 3) / 4 * 4, 4);
        std::vector<PairItem> xit(8 * mx, PairItem{-1, -1, 0, 0});
        std::vector<int32_t> pos_of_task(pitm.size(), -1);
        for (int x = 0; x < 8; ++x)
            for (size_t q = 0; q < lists[x].size(); ++q) {
                xit[x * mx + q] = pitm[lists[x][q]];
                pos_of_task[lists[x][q]] = (int32_t)(x * mx + q);
            }
        h->pair_per_xcd = (int)mx;
        if (h->upload(&h->pair_xitems, xit)) return -1;
        if (ntiles > 1 && !pitm.empty()) {
            // per-block task lists in tile order (tasks are numbered tile-major); partials are
            // addressed by dispatch position
            std::vector<std::pair<int32_t, int32_t>> bt(pitm.size());      // (slot, task)
            for (size_t k = 0; k < pitm.size(); ++k) bt[k] = {pitm[k].slot, (int32_t)k};
            std::stable_sort(bt.begin(), bt.end(), [](const std::pair<int32_t, int32_t>& x, const std::pair<int32_t, int32_t>& y) {
                return x.first < y.first; });
            std::vector<PairItem> citm;
            std::vector<int32_t> ctasks(bt.size());
            for (size_t k = 0; k < bt.size(); ++k) {
                ctasks[k] = pos_of_task[bt[k].second];
                if (k == 0 || bt[k].first != bt[k - 1].first) {
                    if (!citm.empty()) citm.back().end = (int32_t)k;
                    citm.push_back({pitm[bt[k].second].slot, pitm[bt[k].second].slotT, (int32_t)k, 0});
                }
            }
            citm.back().end = (int32_t)bt.size();
            h->ncomb = (int)citm.size();
            if (h->upload(&h->comb_items, citm) || h->upload(&h->comb_tasks, ctasks)) return -1;
            if (h->alloc(&h->Spart, xit.size() * 36)) return -1;
        }
    }
    if (pose_mode && !ptasks.empty()) {
        for (size_t t = 0; t < ptasks.size(); ++t) {
            const int a = (int)(ptask_key[t] >> 32), b = (int)(uint32_t)ptask_key[t];
            ptasks[t].slot = slot_of(a, b); ptasks[t].slotT = slot_of(b, a);
        }
        // XCD x sweeps a contiguous range of poses (its segments in pose order: neighbouring poses share landmarks, hence
        // partner rows -- temporal locality in that XCD's L2 and in the Infinity Cache); ranges balanced by pair count
        long total = 0;
        for (long c : pseg_pairs) total += c + 64;           // (+ a per-segment constant: the row fill)
        std::vector<std::vector<int32_t>> lists(8);
        long run = 0;
        for (size_t q = 0; q < psegs.size(); ++q) {
            lists[std::min(7, (int)(8.0 * (double)run / (double)std::max(total, 1L)))].push_back((int32_t)q);
            run += pseg_pairs[q] + 64;
        }
        size_t mx = 0;
        for (auto& l : lists) mx = std::max(mx, l.size());
        mx = std::max<size_t>(mx, 1);
        std::vector<int32_t> order(8 * mx, -1);
        for (int x = 0; x < 8; ++x) for (size_t q = 0; q < lists[x].size(); ++q) order[x * mx + q] = lists[x][q];
        // per-block task lists in segment order (tasks are numbered pose-, segment-, partner-major)
        std::vector<std::pair<int32_t, int32_t>> bt(ptasks.size());
        for (size_t k = 0; k < ptasks.size(); ++k) bt[k] = {ptasks[k].slot, (int32_t)k};
        std::stable_sort(bt.begin(), bt.end(), [](const std::pair<int32_t, int32_t>& x, const std::pair<int32_t, int32_t>& y) {
            return x.first < y.first; });
        std::vector<PairItem> citm;
        std::vector<int32_t> ctasks(bt.size());
        for (size_t k = 0; k < bt.size(); ++k) {
            ctasks[k] = bt[k].second;
            if (k == 0 || bt[k].first != bt[k - 1].first) {
                if (!citm.empty()) citm.back().end = (int32_t)k;
                citm.push_back({ptasks[bt[k].second].slot, ptasks[bt[k].second].slotT, (int32_t)k, 0});
            }
        }
        citm.back().end = (int32_t)bt.size();
        // device records: a wave takes a contiguous run of the segment's tasks (their words are contiguous), runs balanced by
        // product passes
        std::vector<PoseSegW> psegw(psegs.size());
        for (size_t q = 0; q < psegs.size(); ++q) {
            const PoseSeg& sg = psegs[q];
            PoseSegW& o = psegw[q];
            o.row_start = sg.row_start; o.row_count = sg.row_count;
            const long w0 = sg.task_start < sg.task_end ? ptasks[sg.task_start].start : 0;
            const long wn = sg.task_start < sg.task_end ? ptasks[sg.task_end - 1].end - w0 : 0;
            int tcur = sg.task_start;
            o.wt[0] = tcur;
            for (int w = 1; w <= PS_PP_WAVES; ++w) {
                const long target = w0 + wn * w / PS_PP_WAVES;
                while (tcur < sg.task_end && (w == PS_PP_WAVES || ptasks[tcur].end <= target)) ++tcur;
                // (a task that straddles the target goes to the next wave; the last wave takes the rest)
                o.wt[w] = tcur;
            }
        }
        h->pp_per_xcd = (int)mx; h->pp_ntasks = (int)ptasks.size(); h->pp_ncomb = (int)citm.size();
        if (h->upload(&h->pp_order, order) || h->upload(&h->pp_segs, psegw) || h->upload(&h->pp_rows, pseg_rows) ||
            h->upload(&h->pp_tasks, ptasks) || h->upload(&h->pp_pairs, ppairs) || h->upload(&h->pp_comb_items, citm) ||
            h->upload(&h->pp_comb_tasks, ctasks) || h->alloc(&h->pp_part, ptasks.size() * 36)) return -1;
        h->pose_mode = true;
    }
    h->gather_lists = gather_lists;

    lap("pair items + XCD lists");
    // ---- streaming Schur kernel (k_schur_stream): landmark tiles, per-tile block tables, per-lane-pair entry lists
    {
        // OFF unless PS_SCHUR_STREAM=1: measured slower than the gather kernel on MI355X (C3 0.112-0.137 ms against 0.060,
        // C4 0.79 against 0.57 ms): 57 MB of partial blocks written and read back plus 28 MB of padded entry words
        // outweigh the 9x fewer L2 requests, which L2 / Infinity Cache absorb well enough.
        const char* env = ps_create_env("PS_SCHUR_STREAM");
        h->stream_mode = env ? atoi(env) : 0;                   // 0 off (default), 1 on whenever it can be built
        if (h->stream_mode != 0 && total_pairs > 0 && D == 6) {
            std::vector<StreamTile> tiles;
            std::vector<StreamSub> subs;
            std::vector<uint32_t> ents;
            std::vector<std::pair<int32_t, int32_t>> part_of;   // (block slot, partial index): combine lists
            std::vector<int32_t> part_slotT;
            bool ok = true;
            struct KV { uint64_t key; int32_t val; };
            std::vector<KV> table(4096, KV{~0ULL, -1});
            std::vector<uint32_t> used;
            auto find_or_add = [&](uint64_t key, int32_t next) -> int32_t {
                size_t i = (size_t)((key * 0x9E3779B97F4A7C15ULL) >> 52) & 4095;
                for (;;) {
                    if (table[i].key == key) return table[i].val;
                    if (table[i].key == ~0ULL) { table[i] = KV{key, next}; used.push_back((uint32_t)i); return next; }
                    i = (i + 1) & 4095;
                }
            };
            auto rid_of = [&](int row) { return d->pose_rid[PS_POSE_OF(lobs[row])]; };
            long part_total = 0, ent_pairs = 0;
            // Tiles are cut by the accumulator capacity (PS_ST_CAP blocks) AND by a landmark count that leaves the chip
            // enough workgroups: a multiple of its 256 CUs, about 5 M pairs per round of 256 tiles (C3: 256 tiles of 196
            // landmarks, C4: 1 024 of 489).  Fewer, larger tiles would write fewer partial blocks but leave CUs idle.
            long ttarget = 256L * std::max<long>(1, std::lround((double)total_pairs / 5.0e6));
            if (const char* e2 = ps_create_env("PS_ST_TILES")) ttarget = std::max(1L, atol(e2));
            const int tile_lm_cap = std::max(32, cdiv(nv, ttarget));
            int v = 0;
            while (v < nv && ok) {
                // grow the tile landmark by landmark while its block set fits the accumulators of one workgroup
                for (uint32_t i : used) table[i] = KV{~0ULL, -1};
                used.clear();
                int nblk = 0;
                const int v0 = v;
                std::vector<uint64_t> keys_local;
                for (; v < nv && v - v0 < tile_lm_cap; ++v) {
                    const int a0 = lm_ptr[v], a1 = lm_ptr[v + 1];
                    if (a1 - a0 >= 65536) { ok = false; break; }
                    const size_t used_before = used.size();
                    int added = 0;
                    // (stop inserting as soon as the landmark overflows the accumulators: the 4 096-slot table holds at most
                    //  PS_ST_CAP + 1 keys that way -- a landmark seen by ~90 variable poses used to fill it and spin, ADVICE)
                    for (int a = a0; a < a1 && nblk + added <= PS_ST_CAP; ++a) {
                        const int ra = rid_of(a);
                        if (ra < 0) continue;
                        for (int b = a + 1; b < a1 && nblk + added <= PS_ST_CAP; ++b) {
                            const int rb = rid_of(b);
                            if (rb < 0) continue;
                            const uint64_t key = ((uint64_t)std::min(ra, rb) << 32) | (uint32_t)std::max(ra, rb);
                            if (find_or_add(key, nblk + added) == nblk + added) { keys_local.push_back(key); ++added; }
                        }
                    }
                    if (nblk + added > PS_ST_CAP) {              // this landmark does not fit any more: undo it, close the tile
                        for (size_t q = used_before; q < used.size(); ++q) table[used[q]] = KV{~0ULL, -1};
                        used.resize(used_before);
                        keys_local.resize(nblk);
                        if (v == v0) ok = false;                  // a single landmark with more pairs than accumulators
                        break;
                    }
                    nblk += added;
                }
                if (!ok) break;
                if (nblk == 0) continue;                          // landmarks without a pair (seen by < 2 variable poses)
                const int v1 = v;
                // local block numbering: ascending key (fixed, independent of the hash order)
                std::vector<int32_t> order(nblk), local_of(nblk);
                for (int b = 0; b < nblk; ++b) order[b] = b;
                std::sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return keys_local[x] < keys_local[y]; });
                for (int b = 0; b < nblk; ++b) local_of[order[b]] = b;
                StreamTile tl{};
                tl.row0 = lm_ptr[v0]; tl.sub0 = (int32_t)subs.size(); tl.part0 = (int32_t)part_total; tl.nblk = nblk;
                for (int b = 0; b < nblk; ++b) {
                    const uint64_t key = keys_local[order[b]];
                    const int ra = (int)(key >> 32), rb = (int)(uint32_t)key;
                    part_of.push_back({slot_of(ra, rb), (int32_t)(part_total + b)});
                    part_slotT.push_back(slot_of(rb, ra));
                    if (ra == rb) h->has_diag_tasks = true;
                }
                part_total += nblk;
                // sub-tiles: <= PS_ST_SUBROWS rows each, cut at landmark boundaries, about equal
                const int nrows_t = lm_ptr[v1] - lm_ptr[v0];
                const int nsub = std::max(1, cdiv(nrows_t, PS_ST_SUBROWS));
                const int target = cdiv(nrows_t, nsub);            // rows per sub-tile aimed at
                int vs = v0;
                while (vs < v1) {
                    int ve = vs + 1;
                    while (ve < v1 && lm_ptr[ve] - lm_ptr[vs] < target && lm_ptr[ve + 1] - lm_ptr[vs] <= PS_ST_SUBROWS) ++ve;
                    if (lm_ptr[ve] - lm_ptr[vs] > PS_ST_SUBROWS) { ok = false; break; }     // one landmark longer than a sub-tile
                    StreamSub sb{};
                    sb.row = lm_ptr[vs]; sb.nrows = lm_ptr[ve] - lm_ptr[vs];
                    // per (slot q, lane pair): the row pairs of its block inside this sub-tile, in landmark order
                    std::vector<std::vector<uint32_t>> lists((size_t)PS_ST_SLOTS * PS_ST_PAIRS);
                    for (int vv = vs; vv < ve; ++vv)
                        for (int a = lm_ptr[vv]; a < lm_ptr[vv + 1]; ++a) {
                            const int ra = rid_of(a);
                            if (ra < 0) continue;
                            for (int b = a + 1; b < lm_ptr[vv + 1]; ++b) {
                                const int rb = rid_of(b);
                                if (rb < 0) continue;
                                const uint64_t key = ((uint64_t)std::min(ra, rb) << 32) | (uint32_t)std::max(ra, rb);
                                const int lb = local_of[find_or_add(key, -1)];
                                const int la = (ra <= rb ? a : b) - sb.row, lbr = (ra <= rb ? b : a) - sb.row;
                                lists[(size_t)(lb / PS_ST_PAIRS) * PS_ST_PAIRS + lb % PS_ST_PAIRS].push_back((uint32_t)la | ((uint32_t)lbr << 16));
                                ++ent_pairs;
                            }
                        }
                    for (int q = 0; q < PS_ST_SLOTS; ++q)
                        for (int w = 0; w < PS_ST_THREADS / 64; ++w) {
                            size_t steps = 0;
                            for (int l = 0; l < 64; ++l) steps = std::max(steps, lists[(size_t)q * PS_ST_PAIRS + w * 64 + l].size());
                            const size_t nbatch = (steps + 7) / 8;           // batches of 8 steps, lane major inside a batch
                            sb.off[q][w] = (int32_t)(ents.size() / 4); sb.steps[q][w] = (int32_t)nbatch;
                            for (size_t bt = 0; bt < nbatch; ++bt)
                                for (int l = 0; l < 64; ++l) {
                                    const auto& li = lists[(size_t)q * PS_ST_PAIRS + w * 64 + l];
                                    for (size_t k = 0; k < 8; ++k) ents.push_back(bt * 8 + k < li.size() ? li[bt * 8 + k] : PS_ST_NONE);
                                }
                        }
                    subs.push_back(sb);
                    vs = ve;
                }
                if (!ok) break;
                tl.nsub = (int32_t)subs.size() - tl.sub0;
                tiles.push_back(tl);
                if (ents.size() / 4 >= (1UL << 31) - (1UL << 20)) ok = false;
            }
            // worth it only if a partial block collects several pairs (else the partials outweigh the gathers)
            const double pairs_per_partial = part_total ? (double)total_pairs / (double)part_total : 0.0;
            if (ok && !tiles.empty() && (h->stream_mode == 1 || pairs_per_partial >= 4.0)) {
                // combine lists: per block slot its partials in tile order
                std::stable_sort(part_of.begin(), part_of.end(), [](const std::pair<int32_t, int32_t>& x, const std::pair<int32_t, int32_t>& y) {
                    return x.first < y.first; });
                std::vector<PairItem> citm;
                std::vector<int32_t> ctasks(part_of.size());
                for (size_t k = 0; k < part_of.size(); ++k) {
                    ctasks[k] = part_of[k].second;
                    if (k == 0 || part_of[k].first != part_of[k - 1].first) {
                        if (!citm.empty()) citm.back().end = (int32_t)k;
                        citm.push_back({part_of[k].first, part_slotT[part_of[k].second], (int32_t)k, 0});
                    }
                }
                citm.back().end = (int32_t)part_of.size();
                h->st_ntiles = (int)tiles.size(); h->st_ncomb = (int)citm.size();
                if (h->upload(&h->st_tiles, tiles) || h->upload(&h->st_subs, subs) || h->upload(&h->st_entries, ents) ||
                    h->upload(&h->st_comb_items, citm) || h->upload(&h->st_comb_tasks, ctasks) ||
                    h->alloc(&h->st_part, (size_t)part_total * 36)) return -1;
                h->use_stream = true;
                if (timing) fprintf(stderr, "ps_problem_create: streaming Schur: %d tiles, %zu sub-tiles, %ld partial blocks (%.1f pairs each), "
                                            "%.1f MB of entry words (%.2fx the pairs)\n", h->st_ntiles, subs.size(), part_total, pairs_per_partial,
                                    ents.size() * 4e-6, (double)ents.size() / (double)std::max<long>(1, ent_pairs));
            }
        }
    }
    lap("streaming Schur lists");
    // ---- factor contribution lists
    {
        struct C { int32_t slot, off, tr; };
        std::vector<C> cs;
        std::vector<std::vector<int32_t>> gl(nr);
        for (long f = 0; f < F; ++f) {
            const int ra = f_i[f] >= 0 ? d->pose_rid[f_i[f]] : -1, rb = d->pose_rid[f_j[f]];
            const int32_t base = (int32_t)(f * FROW);
            if ((size_t)f * FROW >= (1UL << 31)) return fail("too many pose factors for 32-bit scratch offsets");
            if (ra >= 0) { cs.push_back({diag_slot[ra], base, 0}); gl[ra].push_back(base + 3 * DD); }
            if (rb >= 0) { cs.push_back({diag_slot[rb], base + 2 * DD, 0}); gl[rb].push_back(base + 3 * DD + D); }
            if (ra >= 0 && rb >= 0) {
                if (ra == rb) return fail("pose-pose edge connects a pose with itself");
                cs.push_back({slot_of(ra, rb), base + DD, 0});
                cs.push_back({slot_of(rb, ra), base + DD, 1});
            }
        }
        std::stable_sort(cs.begin(), cs.end(), [](const C& x, const C& y) { return x.slot < y.slot; });
        std::vector<int32_t> eslots, eptr, ediag, gptr(nr + 1, 0), gitems;
        std::vector<int2> eitems(cs.size());
        for (size_t k = 0; k < cs.size(); ++k) {
            eitems[k] = make_int2(cs[k].off, cs[k].tr);
            if (k == 0 || cs[k].slot != cs[k - 1].slot) { eslots.push_back(cs[k].slot); eptr.push_back((int32_t)k); }
        }
        eptr.push_back((int32_t)cs.size());
        for (int32_t s : eslots) {
            // diagonal iff the slot is some row's diag slot: find its row by binary search on row_ptr
            const int row = (int)(std::upper_bound(row_ptr.begin(), row_ptr.end(), s) - row_ptr.begin()) - 1;
            ediag.push_back(col_idx[s] == row ? 1 : 0);
        }
        for (int r = 0; r < nr; ++r) { gptr[r + 1] = gptr[r] + (int32_t)gl[r].size(); gitems.insert(gitems.end(), gl[r].begin(), gl[r].end()); }
        h->nes = (int)eslots.size();
        if (h->upload(&h->eslots, eslots) || h->upload(&h->eptr, eptr) || h->upload(&h->eslot_diag, ediag) ||
            h->upload(&h->eitems, eitems) || h->upload(&h->gptr, gptr) || h->upload(&h->gitems, gitems)) return -1;
    }

    lap("factor lists");
    // ---- PCG workspace
    const size_t nvec = (size_t)nr * D;
    h->npartA = std::max(1, nr);      // k_pcg_spmv: one workgroup (and one p.q partial) per block row
    h->npartB = std::max(1, cdiv(nr, D == 6 ? PS_PCG_BR(6) : PS_PCG_BR(3)));
    h->hist_cap = 4098;            // classic PCG uses [0,cap); the fused CG needs 2*cap (gamma | alpha)
    if (h->alloc(&h->x, nvec) || h->alloc(&h->r, nvec) || h->alloc(&h->z, nvec) || h->alloc(&h->p0, nvec) ||
        h->alloc(&h->p1, nvec) || h->alloc(&h->q, nvec) || h->alloc(&h->Minv, (size_t)nr * DD) ||
        h->alloc(&h->rz_part, h->npartB) || h->alloc(&h->rr_part, h->npartB) || h->alloc(&h->pq_part, h->npartA) ||
        h->alloc(&h->hist, 2 * (size_t)h->hist_cap)) return -1;
    {
        std::vector<int32_t> brow_of(nnzb), ident(nnzb);
        for (int r = 0; r < nr; ++r) for (int b = row_ptr[r]; b < row_ptr[r + 1]; ++b) brow_of[b] = r;
        for (int b = 0; b < nnzb; ++b) ident[b] = b;
        if (h->upload(&h->brow_of, brow_of) || h->upload(&h->ident_slot, ident)) return -1;
        if (h->alloc(&h->Linv, (size_t)nr * DD)) return -1;
    }
    if (h->zero(h->x, nvec * sizeof(double)) || h->zero(h->p0, nvec * sizeof(double)) || h->zero(h->p1, nvec * sizeof(double))) return -1;

    lap("pcg workspace");
    // ---- scalars
    h->ncost_obs = N > 0 ? std::min(2048, cdiv(N, 256)) : 0;
    h->ncost_fac = F > 0 ? std::min(1024, cdiv(F, 256)) : 0;
    if (h->alloc(&h->scalars, SC_NWORDS)) return -1;
    if (h->zero(h->scalars, SC_NWORDS * sizeof(double)) || h->zero(h->status, ST_NWORDS * sizeof(int32_t))) return -1;
    h->nsq_l = h->nsq_l16 = nv > 0 ? cdiv(nv, 256 / PS_LM_GROUP) : 0;
    // ---- runs of landmarks per wave for the packed landmark pass / back-substitution (ps_k_packed.h)
    h->lmw_nwaves = 0;
    if (nv > 0 && Nl > 0 && D == 6) {
        int mx = 0, mn = INT_MAX;
        if (!dev_build) {
            for (int v = 0; v < nv; ++v) { const int n = lm_ptr[v + 1] - lm_ptr[v]; mx = std::max(mx, n); mn = std::min(mn, n); }
        } else {
            int32_t init[2] = {0, INT_MAX}, *d_mm = nullptr;
            if (h->alloc(&d_mm, 2, true)) return -1;
            HIP_OK(hipMemcpyAsync(d_mm, init, sizeof(init), hipMemcpyHostToDevice, h->stream));
            hipLaunchKernelGGL(k_lmw_maxobs, dim3(cdiv(nv, 256)), dim3(256), 0, h->stream, nv, h->lm_ptr, d_mm);
            HIP_OK(hipMemcpyAsync(init, d_mm, sizeof(init), hipMemcpyDeviceToHost, h->stream));
            HIP_OK(hipStreamSynchronize(h->stream));
            mx = init[0]; mn = init[1];
        }
        if (mx <= PS_LMW_MAXOBS && mn >= 1) {
            const int W = 64 - (mx - 1), nw = cdiv(Nl, W);
            if (!dev_build) {
                std::vector<int32_t> first((size_t)nw + 1, nv);
                for (int w = 0, v = 0; w < nw; ++w) { while (v < nv && lm_ptr[v] < W * w) ++v; first[w] = v; }
                if (h->upload(&h->lmw_first, first)) return -1;
            } else {
                if (h->alloc(&h->lmw_first, (size_t)nw + 1, true)) return -1;
                hipLaunchKernelGGL(k_lmw_items, dim3(cdiv(nw + 1, 256)), dim3(256), 0, h->stream, nv, nw, W, h->lm_ptr, h->lmw_first);
            }
            h->lmw_nwaves = nw;
            h->nsq_l = std::max(h->nsq_l, cdiv(nw, 4));
        }
    }
    h->nsq_p = nr > 0 ? cdiv(P, 256) : 0;
    if (h->alloc(&h->sq_part_l, (size_t)h->nsq_l) || h->alloc(&h->sq_part_p, (size_t)h->nsq_p) ||
        h->alloc(&h->shard_buf, 2)) return -1;
    // (cost partials: one per workgroup of the cost pass, or of the packed landmark pass when that sums the cost; + the factors')
    if (h->alloc(&h->cost_partials, (size_t)std::max({h->ncost_obs + h->ncost_fac, cdiv(h->lmw_nwaves, 4) + h->ncost_fac, 512}) + 8)) return -1;
    if (h->zero(h->shard_buf, 2 * sizeof(double))) return -1;
    // the pinned, host-mapped result words: one block (from the pool), five windows at 256-byte offsets
    if (!ps_pool().take(ps_pool().host_words, &h->words_host))
        HIP_OK(hipHostMalloc(&h->words_host, PS_WORDS_BYTES, hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(h->words_host, 0, PS_WORDS_BYTES);
    char* wdev = nullptr;
    HIP_OK(hipHostGetDevicePointer((void**)&wdev, h->words_host, 0));
    static_assert(SC_NWORDS * sizeof(double) <= 256 && ST_NWORDS * sizeof(int32_t) <= 256, "result words outgrew their windows");
    char* whost = (char*)h->words_host;
    h->h_scalars = (double*)(whost + 0);      h->h_scalars_dev = (double*)(wdev + 0);
    h->h_status = (int32_t*)(whost + 256);    h->h_status_dev = (int32_t*)(wdev + 256);
    h->h_seq = (long long*)(whost + 512);     h->h_seq_dev = (long long*)(wdev + 512);
    h->h_shard = (double*)(whost + 768);      h->h_shard_dev = (double*)(wdev + 768);
    h->h_setup = (long long*)(whost + 1024);  h->h_setup_dev = (long long*)(wdev + 1024);
    h->h_ldi_fro = (double*)(whost + 1280);   h->h_ldi_fro_dev = (double*)(wdev + 1280);
    h->h_early = (long long*)(whost + 1536);  h->h_early_dev = (long long*)(wdev + 1536);
    h->h_lmfail = (long long*)(whost + 1792); h->h_lmfail_dev = (long long*)(wdev + 1792);
    h->h_mo_hist = (double*)(whost + 2048);   h->h_mo_hist_dev = (double*)(wdev + 2048);     // PS_MO_HIST_WORDS doubles
    if (h->alloc(&h->arrivals, 2)) return -1;
    if (h->zero(h->arrivals, 2 * sizeof(int32_t))) return -1;
    if (h->arena_close()) return -1;
    HIP_OK(hipStreamSynchronize(h->stream));
    // The structures of the two-level preconditioner (node lists, augmented pattern, per-workgroup column lists: host work
    // of a few milliseconds at C2 / C4) used to be built by the first solve, i.e. inside the first Gauss-Newton iteration
    // (C2: 9-10 ms against 3 ms for the later ones).  They depend on nothing but the block pattern: built here.  An option
    // that changes them (coarse_groups, cg_explicit, ...) rebuilds them on the next solve as before.
    if (nr >= 16 && nr * D > h->direct_max && h->pcg_variant == 1 && !ps_env("PS_LAZY_COARSE")) {
        if (build_coarse(h)) return -1;
        HIP_OK(hipStreamSynchronize(h->stream));
        lap("two-level structures");
    }
    // a pose graph that will get the direct seed of the lagged inverse (ps_host_ldi.h): its stream is taken here -- the
    // first hipStreamCreate of a process costs ~5 ms, which belongs to set-up, not into the third iteration of a 0.1 ms solve
    if (ldi_eligible(h) && h->ldi_direct_ok && h->N == 0 && h->F > 0 && !h->ldi_stream &&
        !ps_pool().take(ps_pool().side_streams, &h->ldi_stream))
        HIP_OK(hipStreamCreateWithFlags(&h->ldi_stream, hipStreamNonBlocking));
    lap("scalars + final sync");
    guard.ok = true;
    *out = h;
    return 0;
}
