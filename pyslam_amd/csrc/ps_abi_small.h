// ps_abi_small.h -- C ABI: frame-to-frame RANSAC and dense photometric alignment.
// Part of ps_core.hip (one translation unit; included from there, in this order).

// ---- frame-to-frame RANSAC (reference pyslam/pipelines/ransac.py) -------------------------------
int ps_ransac_transforms(const double* pts_1, const double* pts_2, int32_t batch, int32_t n, double* T_out) {
    if (!pts_1 || !pts_2 || !T_out || batch < 0 || n <= 0) return fail("bad argument");
    if (batch == 0) return 0;
    if (need_device()) return -1;
    const size_t nb = (size_t)batch * n * 3 * sizeof(double);
    DevBuf a, b, t;
    if (a.get(nb) || b.get(nb) || t.get((size_t)batch * 16 * sizeof(double))) return -1;
    HIP_OK(hipMemcpy(a.p, pts_1, nb, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(b.p, pts_2, nb, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_ransac_transforms, dim3(cdiv(batch, 64)), dim3(64), 0, 0, batch, n, a.as<double>(), b.as<double>(),
                       t.as<double>());
    HIP_OK(hipMemcpy(T_out, t.p, (size_t)batch * 16 * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

int ps_ransac_cost(const double* T, int32_t num_hyp, const double* pts_1, const double* obs_2, int32_t num_pts,
                   const double* cam5, double thresh, uint8_t* masks, int32_t* counts) {
    if (!T || !pts_1 || !obs_2 || !cam5 || num_hyp < 0 || num_pts < 0) return fail("bad argument");
    if (num_hyp == 0) return 0;
    if (need_device()) return -1;
    DevBuf dT, dp, dobs, dcam, dcnt, dmask;
    const size_t pb = (size_t)num_pts * 3 * sizeof(double);
    if (dT.get((size_t)num_hyp * 16 * sizeof(double)) || dp.get(pb) || dobs.get(pb) || dcam.get(5 * sizeof(double)) ||
        dcnt.get((size_t)num_hyp * sizeof(int32_t)) || dmask.get((size_t)num_hyp * num_pts)) return -1;
    HIP_OK(hipMemcpy(dT.p, T, (size_t)num_hyp * 16 * sizeof(double), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dp.p, pts_1, pb, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dobs.p, obs_2, pb, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dcam.p, cam5, 5 * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_ransac_hypotheses, dim3(num_hyp), dim3(256), 0, 0, num_pts, 0, (const int32_t*)nullptr,
                       dp.as<double>(), (const double*)nullptr, dobs.as<double>(), dcam.as<double>(), thresh,
                       dT.as<double>(), dcnt.as<int32_t>(), dmask.as<uint8_t>());
    if (masks) HIP_OK(hipMemcpy(masks, dmask.p, (size_t)num_hyp * num_pts, hipMemcpyDeviceToHost));
    if (counts) HIP_OK(hipMemcpy(counts, dcnt.p, (size_t)num_hyp * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_OK(hipDeviceSynchronize());
    return 0;
}

int ps_ransac_frame_to_frame(const double* pts_1, const double* pts_2, const double* obs_2, int32_t num_pts,
                             const int32_t* sample_idx, int32_t num_hyp, int32_t set_size, const double* cam5,
                             double thresh, double* T_all, int32_t* counts, int32_t* best_index,
                             int32_t* best_count, double* T_best, uint8_t* best_mask) {
    if (!pts_1 || !pts_2 || !obs_2 || !sample_idx || !cam5 || num_pts <= 0 || num_hyp <= 0 || set_size <= 0)
        return fail("bad argument");
    for (size_t k = 0; k < (size_t)num_hyp * set_size; ++k)
        if (sample_idx[k] < 0 || sample_idx[k] >= num_pts) return fail("sample index out of range");
    if (need_device()) return -1;
    DevBuf dp1, dp2, dobs, dcam, didx, dT, dcnt, dmask, dbest, dTb, dbm;
    const size_t pb = (size_t)num_pts * 3 * sizeof(double);
    if (dp1.get(pb) || dp2.get(pb) || dobs.get(pb) || dcam.get(5 * sizeof(double)) ||
        didx.get((size_t)num_hyp * set_size * sizeof(int32_t)) || dT.get((size_t)num_hyp * 16 * sizeof(double)) ||
        dcnt.get((size_t)num_hyp * sizeof(int32_t)) || dmask.get((size_t)num_hyp * num_pts) ||
        dbest.get(2 * sizeof(int32_t)) || dTb.get(16 * sizeof(double)) || dbm.get((size_t)num_pts)) return -1;
    HIP_OK(hipMemcpy(dp1.p, pts_1, pb, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dp2.p, pts_2, pb, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dobs.p, obs_2, pb, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dcam.p, cam5, 5 * sizeof(double), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(didx.p, sample_idx, (size_t)num_hyp * set_size * sizeof(int32_t), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_ransac_hypotheses, dim3(num_hyp), dim3(256), 0, 0, num_pts, set_size, didx.as<int32_t>(),
                       dp1.as<double>(), dp2.as<double>(), dobs.as<double>(), dcam.as<double>(), thresh,
                       dT.as<double>(), dcnt.as<int32_t>(), dmask.as<uint8_t>());
    hipLaunchKernelGGL(k_ransac_best, dim3(1), dim3(256), 0, 0, num_hyp, num_pts, dcnt.as<int32_t>(), dT.as<double>(),
                       dmask.as<uint8_t>(), dbest.as<int32_t>(), dTb.as<double>(), dbm.as<uint8_t>());
    int32_t bi[2];
    HIP_OK(hipMemcpy(bi, dbest.p, sizeof(bi), hipMemcpyDeviceToHost));
    if (best_index) *best_index = bi[0];
    if (best_count) *best_count = bi[1];
    if (T_best) HIP_OK(hipMemcpy(T_best, dTb.p, 16 * sizeof(double), hipMemcpyDeviceToHost));
    if (best_mask) HIP_OK(hipMemcpy(best_mask, dbm.p, (size_t)num_pts, hipMemcpyDeviceToHost));
    if (T_all) HIP_OK(hipMemcpy(T_all, dT.p, (size_t)num_hyp * 16 * sizeof(double), hipMemcpyDeviceToHost));
    if (counts) HIP_OK(hipMemcpy(counts, dcnt.p, (size_t)num_hyp * sizeof(int32_t), hipMemcpyDeviceToHost));
    return 0;
}

// ---- dense photometric alignment (reference pyslam/residuals/photometric_residual.py) ------------------
struct ps_photo {
    hipStream_t stream = nullptr;
    PhotoArgs args{};
    std::vector<void*> allocs;
    double *pose = nullptr, *partials = nullptr, *out = nullptr;
    int nparts = 0;
    double h_out[PS_PHOTO_NOUT];
    int upload(const double** dst, const double* src, size_t n) {
        void* p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(double)) != hipSuccess) return fail("hipMalloc failed");
        allocs.push_back(p);
        if (n && hipMemcpy(p, src, n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return fail("hipMemcpy failed");
        *dst = (const double*)p;
        return 0;
    }
    ~ps_photo() { for (void* p : allocs) hipFree(p); }
};

namespace {
int photo_pass(ps_photo* h, int with_normal, int update) {
    hipLaunchKernelGGL(k_photo_pass, dim3(h->nparts), dim3(256), 0, h->stream, h->args, (const double*)h->pose, with_normal,
                       h->partials);
    hipLaunchKernelGGL(k_photo_finish, dim3(1), dim3(256), 0, h->stream, h->nparts, (const double*)h->partials, with_normal,
                       update, h->pose, h->out);
    return 0;
}
int photo_fetch(ps_photo* h) {
    HIP_OK(hipMemcpyAsync(h->h_out, h->out, sizeof(h->h_out), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
    return 0;
}
}  // namespace

int ps_photometric_create(const ps_photo_desc* d, void* stream, ps_photo** out) {
    if (!d || !out) return fail("null argument");
    *out = nullptr;
    if (d->num_pixels < 0 || d->height <= 0 || d->width <= 0) return fail("bad image or pixel count");
    if (d->num_pixels > 0 && (!d->pt_ref || !d->im_ref || !d->im_jac || !d->tri_jac_d)) return fail("null pixel table");
    if (!d->im_track) return fail("null tracking image");
    if (d->cam_type != 0 && d->cam_type != 1) return fail("cam_type must be 0 (stereo) or 1 (RGB-D)");
    if (d->loss_id < 0 || d->loss_id > 5) return fail("unknown loss id");
    if (need_device()) return -1;
    std::unique_ptr<ps_photo> h(new ps_photo);
    h->stream = (hipStream_t)stream;
    PhotoArgs& a = h->args;
    const size_t n = (size_t)d->num_pixels;
    a.n = d->num_pixels; a.h = d->height; a.w = d->width;
    if (h->upload(&a.pt_ref, d->pt_ref, 3 * n) || h->upload(&a.im_ref, d->im_ref, n) ||
        h->upload(&a.im_jac, d->im_jac, 2 * n) || h->upload(&a.tri_jac_d, d->tri_jac_d, 3 * n) ||
        h->upload(&a.image, d->im_track, (size_t)d->height * d->width)) return -1;
    a.cu = d->cam[0]; a.cv = d->cam[1]; a.fu = d->cam[2]; a.fv = d->cam[3]; a.b = d->cam[4];
    a.cam_type = d->cam_type; a.cam_w = (double)d->cam_w; a.cam_h = (double)d->cam_h;
    a.var_i = d->intensity_covar; a.var_d = d->depth_covar;
    a.loss_id = d->loss_id; a.loss_k = d->loss_k;
    h->nparts = std::max(1, cdiv(d->num_pixels, 256 * PS_PHOTO_PPT));
    const double ident[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
    const double* tmp = nullptr;
    if (h->upload(&tmp, ident, 12)) return -1;
    h->pose = const_cast<double*>(tmp);
    std::vector<double> zeros((size_t)h->nparts * PS_PHOTO_NACC + PS_PHOTO_NOUT, 0.0);
    if (h->upload(&tmp, zeros.data(), (size_t)h->nparts * PS_PHOTO_NACC)) return -1;
    h->partials = const_cast<double*>(tmp);
    if (h->upload(&tmp, zeros.data(), (size_t)PS_PHOTO_NOUT)) return -1;
    h->out = const_cast<double*>(tmp);
    *out = h.release();
    return 0;
}

int ps_photometric_destroy(ps_photo* h) {
    if (!h) return 0;
    if (h->stream) hipStreamSynchronize(h->stream); else hipDeviceSynchronize();
    delete h;
    return 0;
}

int ps_photometric_set_pose(ps_photo* h, const double* pose12) {
    if (!h || !pose12) return fail("null argument");
    HIP_OK(hipMemcpyAsync(h->pose, pose12, 12 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
    return 0;
}

int ps_photometric_get_pose(ps_photo* h, double* pose12) {
    if (!h || !pose12) return fail("null argument");
    HIP_OK(hipMemcpyAsync(pose12, h->pose, 12 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
    return 0;
}

int ps_photometric_eval_cost(ps_photo* h, double* cost, int64_t* num_valid) {
    if (!h) return fail("null handle");
    if (photo_pass(h, 0, 0) || photo_fetch(h)) return -1;
    if (cost) *cost = h->h_out[42];
    if (num_valid) *num_valid = (int64_t)h->h_out[43];
    return 0;
}

int ps_photometric_normal_equations(ps_photo* h, double* H36, double* b6, double* cost, int64_t* num_valid) {
    if (!h) return fail("null handle");
    if (photo_pass(h, 1, 0) || photo_fetch(h)) return -1;
    if (H36) std::copy(h->h_out, h->h_out + 36, H36);
    if (b6) std::copy(h->h_out + 36, h->h_out + 42, b6);
    if (cost) *cost = h->h_out[42];
    if (num_valid) *num_valid = (int64_t)h->h_out[43];
    return 0;
}

int ps_photometric_iteration(ps_photo* h, int32_t split_params, int32_t linesearch, double* dx6, double* cost) {
    if (!h) return fail("null handle");
    if (photo_pass(h, 1, split_params ? 2 : 1) || photo_fetch(h)) return -1;
    if (h->h_out[43] < 6.0) return fail("photometric alignment: fewer than 6 valid pixels");
    if (h->h_out[50] != 0.0) return fail("photometric alignment: normal equations are not positive definite");
    if (dx6) std::copy(h->h_out + 44, h->h_out + 50, dx6);
    double c = h->h_out[42];
    if (linesearch) {
        if (photo_pass(h, 0, 0) || photo_fetch(h)) return -1;
        c = h->h_out[42];
    }
    if (cost) *cost = c;
    return 0;
}

// ---- host-evaluated generic path at scale: CG on J^T J dx = rhs, J sparse in HBM (csrc/ps_sparse.h) -----------------
int ps_sparse_normal_solve(int32_t m, int32_t n, const int32_t* j_row_ptr, const int32_t* j_col, const double* j_val,
                           const int32_t* jt_row_ptr, const int32_t* jt_col, const double* jt_val,
                           const double* r, const double* rhs, double tol, int32_t max_iters,
                           double* dx, int32_t* iters_out, double* relres_out) {
    if (!j_row_ptr || !j_col || !j_val || !jt_row_ptr || !jt_col || !jt_val || !dx || m <= 0 || n <= 0 || (!r && !rhs))
        return fail("bad argument");
    if (need_device()) return -1;
    const size_t nnz = (size_t)j_row_ptr[m];
    if ((size_t)jt_row_ptr[n] != nnz) return fail("J and its transpose have different numbers of non-zeros");
    DevBuf bJr, bJc, bJv, bTr, bTc, bTv, be, bb, bM, bx, br, bz, bp, bq, by, bpart, bsc;
    const int nb = std::max(1, std::min(1024, cdiv(n, 256)));
    if (bJr.get((size_t)(m + 1) * 4) || bJc.get(nnz * 4) || bJv.get(nnz * 8) || bTr.get((size_t)(n + 1) * 4) ||
        bTc.get(nnz * 4) || bTv.get(nnz * 8) || be.get((size_t)m * 8) || bb.get((size_t)n * 8) || bM.get((size_t)n * 8) ||
        bx.get((size_t)n * 8) || br.get((size_t)n * 8) || bz.get((size_t)n * 8) || bp.get((size_t)n * 8) ||
        bq.get((size_t)n * 8) || by.get((size_t)m * 8) || bpart.get((size_t)nb * 8) || bsc.get(SPS_N * 8)) return -1;
    HIP_OK(hipMemcpy(bJr.p, j_row_ptr, (size_t)(m + 1) * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(bJc.p, j_col, nnz * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(bJv.p, j_val, nnz * 8, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(bTr.p, jt_row_ptr, (size_t)(n + 1) * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(bTc.p, jt_col, nnz * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(bTv.p, jt_val, nnz * 8, hipMemcpyHostToDevice));
    HIP_OK(hipMemset(bsc.p, 0, SPS_N * 8));
    hipStream_t st = 0;
    const int gm = cdiv(m, 256), gn = cdiv(n, 256);
    if (rhs) HIP_OK(hipMemcpy(bb.p, rhs, (size_t)n * 8, hipMemcpyHostToDevice));
    else {                                                          // rhs = -J^T r
        HIP_OK(hipMemcpy(be.p, r, (size_t)m * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_sp_spmv, dim3(gn), dim3(256), 0, st, n, bTr.as<int32_t>(), bTc.as<int32_t>(), bTv.as<double>(),
                           be.as<double>(), bb.as<double>(), -1.0);
    }
    hipLaunchKernelGGL(k_sp_diag, dim3(gn), dim3(256), 0, st, n, bTr.as<int32_t>(), bTv.as<double>(), bM.as<double>());
    hipLaunchKernelGGL(k_sp_init, dim3(nb), dim3(256), 0, st, n, bb.as<double>(), bM.as<double>(), bx.as<double>(),
                       br.as<double>(), bz.as<double>(), bp.as<double>(), bpart.as<double>());
    hipLaunchKernelGGL(k_sp_scalars, dim3(1), dim3(256), 0, st, nb, bpart.as<double>(), bsc.as<double>(), (int)SPS_RZ, tol * tol, 1);
    double sc[SPS_N] = {};
    int launched = 0;
    while (launched < max_iters) {
        const int chunk = std::min(32, max_iters - launched);
        for (int k = 0; k < chunk; ++k, ++launched) {
            hipLaunchKernelGGL(k_sp_dir, dim3(nb), dim3(256), 0, st, n, bz.as<double>(), bp.as<double>(), bsc.as<double>());
            hipLaunchKernelGGL(k_sp_spmv, dim3(gm), dim3(256), 0, st, m, bJr.as<int32_t>(), bJc.as<int32_t>(), bJv.as<double>(),
                               bp.as<double>(), by.as<double>(), 1.0);
            hipLaunchKernelGGL(k_sp_spmv, dim3(gn), dim3(256), 0, st, n, bTr.as<int32_t>(), bTc.as<int32_t>(), bTv.as<double>(),
                               by.as<double>(), bq.as<double>(), 1.0);
            hipLaunchKernelGGL(k_sp_dot, dim3(nb), dim3(256), 0, st, n, bp.as<double>(), bq.as<double>(), bpart.as<double>());
            hipLaunchKernelGGL(k_sp_scalars, dim3(1), dim3(256), 0, st, nb, bpart.as<double>(), bsc.as<double>(), (int)SPS_PQ, tol * tol, 0);
            hipLaunchKernelGGL(k_sp_update, dim3(nb), dim3(256), 0, st, n, bp.as<double>(), bq.as<double>(), bM.as<double>(),
                               bx.as<double>(), br.as<double>(), bz.as<double>(), bsc.as<double>(), bpart.as<double>());
            hipLaunchKernelGGL(k_sp_scalars, dim3(1), dim3(256), 0, st, nb, bpart.as<double>(), bsc.as<double>(), (int)SPS_RZ, tol * tol, 0);
        }
        HIP_OK(hipMemcpy(sc, bsc.p, sizeof(sc), hipMemcpyDeviceToHost));
        if (sc[SPS_DONE] != 0.0) break;
    }
    HIP_OK(hipMemcpy(dx, bx.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    if (iters_out) *iters_out = (int32_t)sc[SPS_ITERS];
    if (relres_out) *relres_out = sc[SPS_RZ0] > 0.0 ? std::sqrt(sc[SPS_RZ] / sc[SPS_RZ0]) : 0.0;
    if (sc[SPS_DONE] == 2.0) return fail("sparse normal-equation CG broke down (J^T J is not positive semi-definite to rounding, or a NaN in J)");
    return 0;
}

// ---- host-evaluated generic path, DIRECT: (J^T J) dx = -J^T r (or = rhs) by a dense blocked Cholesky on the device, n <= 8 192
// (csrc/ps_sparse.h "Round 5").  relres_out: ||rhs - J^T J dx|| / ||rhs|| of the ORIGINAL system after the refinement steps.
int ps_sparse_normal_direct(int32_t m, int32_t n, const int32_t* j_row_ptr, const int32_t* j_col, const double* j_val,
                            const int32_t* jt_row_ptr, const int32_t* jt_col, const double* jt_val,
                            const double* r, const double* rhs, int32_t refine_steps, double* dx, double* relres_out) {
    if (!j_row_ptr || !j_col || !j_val || !jt_row_ptr || !jt_col || !jt_val || !dx || m <= 0 || n <= 0 || (!r && !rhs))
        return fail("bad argument");
    if (n > PS_SPD_MAXN) return fail("ps_sparse_normal_direct: more unknowns than the dense direct solve takes (8192)");
    if (need_device()) return -1;
    const size_t nnz = (size_t)j_row_ptr[m];
    if ((size_t)jt_row_ptr[n] != nnz) return fail("J and its transpose have different numbers of non-zeros");
    const int nsteps = cdiv(n, PS_BC_W), nb = cdiv(n, 4);
    DevBuf bJr, bJc, bJv, bTr, bTc, bTv, be, bb, bH, bA, bTi, bx, bd, bres, bpart, bst;
    if (bJr.get((size_t)(m + 1) * 4) || bJc.get(nnz * 4) || bJv.get(nnz * 8) || bTr.get((size_t)(n + 1) * 4) || bTc.get(nnz * 4) ||
        bTv.get(nnz * 8) || be.get((size_t)m * 8) || bb.get((size_t)n * 8) || bH.get((size_t)n * n * 8) || bA.get((size_t)n * n * 8) ||
        bTi.get((size_t)nsteps * PS_BC_W * PS_BC_W * 8) || bx.get((size_t)n * 8) || bd.get((size_t)n * 8) || bres.get((size_t)n * 8) ||
        bpart.get((size_t)2 * nb * 8) || bst.get(ST_NWORDS * 4)) return -1;
    HIP_OK(hipMemcpy(bJr.p, j_row_ptr, (size_t)(m + 1) * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(bJc.p, j_col, nnz * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(bJv.p, j_val, nnz * 8, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(bTr.p, jt_row_ptr, (size_t)(n + 1) * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(bTc.p, jt_col, nnz * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(bTv.p, jt_val, nnz * 8, hipMemcpyHostToDevice));
    HIP_OK(hipMemset(bst.p, 0, ST_NWORDS * 4));
    hipStream_t st = 0;
    if (rhs) HIP_OK(hipMemcpy(bb.p, rhs, (size_t)n * 8, hipMemcpyHostToDevice));
    else {
        HIP_OK(hipMemcpy(be.p, r, (size_t)m * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_sp_spmv, dim3(cdiv(n, 256)), dim3(256), 0, st, n, bTr.as<int32_t>(), bTc.as<int32_t>(), bTv.as<double>(),
                           be.as<double>(), bb.as<double>(), -1.0);
    }
    double* H = bH.as<double>();
    double* A = bA.as<double>();
    hipLaunchKernelGGL(k_spd_normal, dim3(cdiv((long)n * n, 256)), dim3(256), 0, st, n, bJr.as<int32_t>(), bJc.as<int32_t>(), bJv.as<double>(),
                       bTr.as<int32_t>(), bTc.as<int32_t>(), bTv.as<double>(), H);
    HIP_OK(hipMemcpyAsync(A, H, (size_t)n * n * 8, hipMemcpyDeviceToDevice, st));
    for (int s2 = 0; s2 < nsteps; ++s2) {                      // blocked right-looking Cholesky over the whole chip
        const int j0 = s2 * PS_BC_W, w = std::min(PS_BC_W, n - j0), mrem = n - j0 - w;
        hipLaunchKernelGGL(k_bchol_panel, dim3(std::max(1, cdiv((long)mrem * w, 1024))), dim3(256), 0, st, n, j0, A,
                           bTi.as<double>() + (size_t)s2 * PS_BC_W * PS_BC_W, bst.as<int32_t>());
        if (mrem > 0) {
            const int nt = cdiv(mrem, 32);
            hipLaunchKernelGGL(k_bchol_update, dim3(nt * (nt + 1) / 2), dim3(256), 0, st, n, j0, w, A);
        }
    }
    auto solve_into = [&](double* v) {                         // v <- (L L^T)^-1 v
        hipLaunchKernelGGL(k_spd_subst, dim3(1), dim3(512), 0, st, n, (const double*)A, (const double*)bTi.as<double>(), v, 0);
        hipLaunchKernelGGL(k_spd_subst, dim3(1), dim3(512), 0, st, n, (const double*)A, (const double*)bTi.as<double>(), v, 1);
    };
    HIP_OK(hipMemcpyAsync(bx.p, bb.p, (size_t)n * 8, hipMemcpyDeviceToDevice, st));
    solve_into(bx.as<double>());
    std::vector<double> part((size_t)2 * nb);
    double rel = 0.0;
    for (int it = 0;; ++it) {                                  // residual of the original system; refine through the factor
        hipLaunchKernelGGL(k_spd_residual, dim3(nb), dim3(256), 0, st, n, (const double*)H, (const double*)bb.as<double>(),
                           (const double*)bx.as<double>(), bres.as<double>(), bpart.as<double>());
        HIP_OK(hipMemcpy(part.data(), bpart.p, part.size() * 8, hipMemcpyDeviceToHost));
        double rr = 0.0, bbn = 0.0;
        for (int k = 0; k < nb; ++k) { rr += part[2 * k]; bbn += part[2 * k + 1]; }
        rel = bbn > 0.0 ? std::sqrt(rr / bbn) : 0.0;
        if (it >= refine_steps || !(rel > 1e-15)) break;
        solve_into(bres.as<double>());
        hipLaunchKernelGGL(k_spd_axpy, dim3(cdiv(n, 256)), dim3(256), 0, st, n, (const double*)bres.as<double>(), bx.as<double>());
    }
    int32_t stw[ST_NWORDS];
    HIP_OK(hipMemcpy(stw, bst.p, sizeof(stw), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(dx, bx.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    if (relres_out) *relres_out = rel;
    if (stw[ST_DIAG_FAIL]) return fail("normal matrix is not positive definite to rounding (dense direct solve)");
    return 0;
}

// ---- the band factorisation kernels on their own (csrc/ps_k_band.h, ps_k_bandpart.h): inverse of a symmetric positive definite
// matrix with `bw` block off-diagonals of D x D blocks, A given dense on the host (row-major, nc = ncb D; the lower triangle is
// read).  chunk_nodes < 0: the one-workgroup column walk (k_band_chol + k_band_inverse_rl); 0: the partitioned form with its
// automatic chunk size; > 0: that many interior nodes per chunk.  ainv_out: nc x nc fp32 (what the explicit two-level PCG
// keeps).  elapsed_us (may be NULL): GPU time of the factorisation + inverse launches, measured with events.
int ps_debug_factor_stress(const double* a, int32_t ncb, int32_t dof, int32_t bw, int32_t mode, int32_t launches, int32_t lowprio,
                           int32_t aggressor, int32_t* n_diff, int32_t* n_pivot) {
    // mode bits: 8 = the input is PRODUCED on the victim stream in front of every factorisation (the buffer first holds 2 A, then a
    // many-workgroup copy kernel writes A: a factorisation that starts before its producer has finished reads a mix of both);
    // 16 = an event is recorded between producer and factorisation (as xcg_side_enqueue does with ev_acdone)
    const bool produce = (mode & 8) != 0, with_event = (mode & 16) != 0;
    mode &= 7;
    if (!a || !n_diff || !n_pivot || ncb <= 0 || (dof != 3 && dof != 6) || mode < 0 || mode > 3 || launches < 1) return fail("bad argument");
    if (mode <= 1 && (bw < 1 || bw > PS_BAND_MAXB)) return fail("bad band width");
    if (need_device()) return -1;
    const int nc = ncb * dof;
    if (mode == 2 && nc > 90) return fail("mode 2 (LDS-resident k_coarse_chol) needs ncb dof <= 90");
    DevBuf bA, bInv, bLr, bLc, brd, bXs, bst, bLi, bLiT, bsc, ga, gb;
    const size_t gn = (size_t)16 << 20;
    if (bA.get((size_t)nc * nc * 8) || bInv.get((size_t)nc * nc * 4) || bst.get(ST_NWORDS * 4) || bLr.get((size_t)nc * PS_BAND_W * 8) ||
        bLc.get((size_t)nc * PS_BAND_W * 8) || brd.get((size_t)nc * 8) || bXs.get((size_t)nc * nc * 8) || bLi.get((size_t)nc * nc * 8) ||
        bLiT.get((size_t)nc * nc * 8) || bsc.get((size_t)2 * nc * nc * 8) || ga.get(gn * 8) || gb.get(gn * 8)) return -1;
    HIP_OK(hipMemcpy(bA.p, a, (size_t)nc * nc * 8, hipMemcpyHostToDevice));
    HIP_OK(hipMemset(ga.p, 0, gn * 8));
    DevBuf bA1, bA2;
    hipEvent_t evp = nullptr;
    if (produce) {
        std::vector<double> a2((size_t)nc * nc);
        for (size_t k = 0; k < a2.size(); ++k) a2[k] = 2.0 * a[k];
        if (bA1.get((size_t)nc * nc * 8) || bA2.get((size_t)nc * nc * 8)) return -1;
        HIP_OK(hipMemcpy(bA1.p, a, (size_t)nc * nc * 8, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(bA2.p, a2.data(), (size_t)nc * nc * 8, hipMemcpyHostToDevice));
        HIP_OK(hipEventCreateWithFlags(&evp, hipEventDisableTiming));
    }
    struct EvGuard { hipEvent_t e; ~EvGuard() { if (e) hipEventDestroy(e); } } evg{evp};
    hipStream_t vs = nullptr, as = nullptr;
    int lo = 0, hi = 0;
    HIP_OK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    if (lowprio) HIP_OK(hipStreamCreateWithPriority(&vs, hipStreamNonBlocking, lo)); else HIP_OK(hipStreamCreateWithFlags(&vs, hipStreamNonBlocking));
    HIP_OK(hipStreamCreateWithFlags(&as, hipStreamNonBlocking));
    struct Streams { hipStream_t a, b; ~Streams() { hipStreamDestroy(a); hipStreamDestroy(b); } } streams{vs, as};
    std::unique_ptr<BandPart> bp;
    if (mode == 1) {
        const int m = BandPart::auto_m(ncb, bw);
        if (!BandPart::eligible(ncb, bw, m)) return fail("mode 1: the partitioned factorisation does not apply to this shape");
        bp.reset(new BandPart());
        if (bp->build(ncb, dof, bw, m, vs)) return -1;
    }
    const size_t chol_lds = 2 * (size_t)nc * nc * sizeof(double);
    if (mode == 2 && (dof == 6 ? ensure_dynamic_lds((const void*)k_coarse_chol<6, true>, chol_lds) : ensure_dynamic_lds((const void*)k_coarse_chol<3, true>, chol_lds))) return -1;
    auto run = [&](hipStream_t st) -> int {
        if (produce) {
            copy_doubles(st, bA.as<double>(), bA2.as<double>(), (size_t)nc * nc);       // 2 A ...
            copy_doubles(st, bA.as<double>(), bA1.as<double>(), (size_t)nc * nc);       // ... then A, by a kernel of many workgroups
            if (with_event) HIP_OK(hipEventRecord(evp, st));
        }
        HIP_OK(hipMemsetAsync(bst.p, 0, ST_NWORDS * 4, st));
        HIP_OK(hipMemsetAsync(bInv.p, 0, (size_t)nc * nc * 4, st));
        if (mode == 0) {
            HIP_OK(hipMemsetAsync(bLr.p, 0, (size_t)nc * PS_BAND_W * 8, st));
            HIP_OK(hipMemsetAsync(bLc.p, 0, (size_t)nc * PS_BAND_W * 8, st));
            if (dof == 6) hipLaunchKernelGGL(k_band_chol<6>, dim3(1), dim3(256), 0, st, ncb, bw, bA.as<double>(), bLr.as<double>(), bLc.as<double>(), brd.as<double>(), bst.as<int32_t>(), nc, (const int2*)nullptr);
            else hipLaunchKernelGGL(k_band_chol<3>, dim3(1), dim3(256), 0, st, ncb, bw, bA.as<double>(), bLr.as<double>(), bLc.as<double>(), brd.as<double>(), bst.as<int32_t>(), nc, (const int2*)nullptr);
            hipLaunchKernelGGL(k_band_inverse_rl<false>, dim3(cdiv(nc, 4)), dim3(256), 0, st, nc, (const double*)bLr.as<double>(), (const double*)bLc.as<double>(),
                               (const double*)brd.as<double>(), bXs.as<double>(), bInv.as<float>(), (const BandInvItem*)nullptr, (double*)nullptr, nc);
        } else if (mode == 1) {
            if (dof == 6 ? bp->run<6>(st, bA.as<double>(), nc, bInv.as<float>(), nc, bst.as<int32_t>()) : bp->run<3>(st, bA.as<double>(), nc, bInv.as<float>(), nc, bst.as<int32_t>())) return -1;
        } else {
            if (mode == 2) {
                if (dof == 6) hipLaunchKernelGGL((k_coarse_chol<6, true>), dim3(1), dim3(1024), chol_lds, st, ncb, bA.as<double>(), bLi.as<double>(), bLiT.as<double>(), bst.as<int32_t>(), nullptr);
                else hipLaunchKernelGGL((k_coarse_chol<3, true>), dim3(1), dim3(1024), chol_lds, st, ncb, bA.as<double>(), bLi.as<double>(), bLiT.as<double>(), bst.as<int32_t>(), nullptr);
            } else {
                if (dof == 6) hipLaunchKernelGGL((k_coarse_chol<6, false>), dim3(1), dim3(1024), 0, st, ncb, bA.as<double>(), bLi.as<double>(), bLiT.as<double>(), bst.as<int32_t>(), bsc.as<double>());
                else hipLaunchKernelGGL((k_coarse_chol<3, false>), dim3(1), dim3(1024), 0, st, ncb, bA.as<double>(), bLi.as<double>(), bLiT.as<double>(), bst.as<int32_t>(), bsc.as<double>());
            }
            hipLaunchKernelGGL(k_xcg_ainv, dim3(cdiv(nc, PS_AI_T) * (cdiv(nc, PS_AI_T) + 1) / 2), dim3(256), 0, st, nc, bLi.as<double>(), bInv.as<float>(), nc);
        }
        return 0;
    };
    std::vector<float> ref((size_t)nc * nc), out((size_t)nc * nc);
    int32_t stw[ST_NWORDS];
    if (run(vs)) return -1;
    HIP_OK(hipStreamSynchronize(vs));
    HIP_OK(hipMemcpy(ref.data(), bInv.p, ref.size() * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(stw, bst.p, sizeof(stw), hipMemcpyDeviceToHost));
    if (stw[ST_DIAG_FAIL]) return fail("ps_debug_factor_stress: the input is not positive definite (idle reference run)");
    *n_diff = *n_pivot = 0;
    for (int k = 0; k < launches; ++k) {
        if (aggressor == 1) for (int q = 0; q < 3; ++q) copy_doubles(as, gb.as<double>(), ga.as<double>(), gn);
        if (aggressor == 2) {                                 // workgroups with 96 KB of LDS each, scribbling over all of it
            const size_t lds = 96 * 1024;
            if (ensure_dynamic_lds((const void*)k_lds_scribble, lds)) return -1;
            for (int q = 0; q < 2; ++q) hipLaunchKernelGGL(k_lds_scribble, dim3(1024), dim3(512), lds, as, (int)(lds / 8), 6, gb.as<double>());
        }
        if (run(vs)) return -1;
        HIP_OK(hipStreamSynchronize(vs));
        HIP_OK(hipMemcpy(out.data(), bInv.p, out.size() * 4, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(stw, bst.p, sizeof(stw), hipMemcpyDeviceToHost));
        if (std::memcmp(out.data(), ref.data(), out.size() * 4)) ++*n_diff;
        if (stw[ST_DIAG_FAIL]) ++*n_pivot;
    }
    HIP_OK(hipDeviceSynchronize());
    if (bp) bp.reset();
    return 0;
}

int ps_debug_band_inverse(const double* a, int32_t ncb, int32_t dof, int32_t bw, int32_t chunk_nodes, float* ainv_out, double* elapsed_us) {
    if (!a || !ainv_out || ncb <= 0 || (dof != 3 && dof != 6) || bw < 1 || bw > PS_BAND_MAXB) return fail("bad argument");
    if (need_device()) return -1;
    const int nc = ncb * dof;
    DevBuf bA, bInv, bLr, bLc, brd, bXs, bst;
    if (bA.get((size_t)nc * nc * 8) || bInv.get((size_t)nc * nc * 4) || bst.get(ST_NWORDS * 4)) return -1;
    HIP_OK(hipMemcpy(bA.p, a, (size_t)nc * nc * 8, hipMemcpyHostToDevice));
    HIP_OK(hipMemset(bst.p, 0, ST_NWORDS * 4));
    HIP_OK(hipMemset(bInv.p, 0, (size_t)nc * nc * 4));
    hipStream_t st = 0;
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    struct Ev { hipEvent_t a, b; ~Ev() { hipEventDestroy(a); hipEventDestroy(b); } } evs{e0, e1};
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {                       // (the second run is the one timed: the first loads the kernels)
        if (chunk_nodes < 0) {
            if (rep == 0 && (bLr.get((size_t)nc * PS_BAND_W * 8) || bLc.get((size_t)nc * PS_BAND_W * 8) || brd.get((size_t)nc * 8) ||
                             bXs.get((size_t)nc * nc * 8))) return -1;
            HIP_OK(hipMemsetAsync(bLr.p, 0, (size_t)nc * PS_BAND_W * 8, st));
            HIP_OK(hipMemsetAsync(bLc.p, 0, (size_t)nc * PS_BAND_W * 8, st));
            HIP_OK(hipEventRecord(e0, st));
            if (dof == 6) hipLaunchKernelGGL(k_band_chol<6>, dim3(1), dim3(256), 0, st, ncb, bw, bA.as<double>(), bLr.as<double>(), bLc.as<double>(),
                                             brd.as<double>(), bst.as<int32_t>(), nc, (const int2*)nullptr);
            else hipLaunchKernelGGL(k_band_chol<3>, dim3(1), dim3(256), 0, st, ncb, bw, bA.as<double>(), bLr.as<double>(), bLc.as<double>(),
                                    brd.as<double>(), bst.as<int32_t>(), nc, (const int2*)nullptr);
            hipLaunchKernelGGL(k_band_inverse_rl<false>, dim3(cdiv(nc, 4)), dim3(256), 0, st, nc, (const double*)bLr.as<double>(),
                               (const double*)bLc.as<double>(), (const double*)brd.as<double>(), bXs.as<double>(), bInv.as<float>(),
                               (const BandInvItem*)nullptr, (double*)nullptr, nc);
            HIP_OK(hipEventRecord(e1, st));
        } else {
            static thread_local std::unique_ptr<BandPart> bp;
            const int m = chunk_nodes > 0 ? chunk_nodes : BandPart::auto_m(ncb, bw);
            if (rep == 0) {
                if (2 * bw - 1 > PS_BAND_MAXB || m < bw) return fail("partitioned band factorisation: needs 2 bw - 1 <= 7 and chunks of at least bw nodes");
                bp.reset(new BandPart());
                if (bp->build(ncb, dof, bw, m, st)) { bp.reset(); return -1; }
            }
            HIP_OK(hipEventRecord(e0, st));
            const int rc = dof == 6 ? bp->run<6>(st, bA.as<double>(), nc, bInv.as<float>(), nc, bst.as<int32_t>())
                                    : bp->run<3>(st, bA.as<double>(), nc, bInv.as<float>(), nc, bst.as<int32_t>());
            HIP_OK(hipEventRecord(e1, st));
            if (rc) { bp.reset(); return -1; }
            if (rep == 1) { HIP_OK(hipStreamSynchronize(st)); bp.reset(); }
        }
        HIP_OK(hipStreamSynchronize(st));
    }
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    if (elapsed_us) *elapsed_us = 1e3 * ms;
    int32_t stw[ST_NWORDS];
    HIP_OK(hipMemcpy(stw, bst.p, sizeof(stw), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(ainv_out, bInv.p, (size_t)nc * nc * 4, hipMemcpyDeviceToHost));
    if (stw[ST_DIAG_FAIL]) return fail("band factorisation: the matrix is not positive definite");
    return 0;
}
