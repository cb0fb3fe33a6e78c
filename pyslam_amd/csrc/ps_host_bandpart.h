// ps_host_bandpart.h -- host side of the partitioned band factorisation (kernels and the algebra: ps_k_bandpart.h).
// Part of ps_core.hip (one translation unit; included from there before ps_host_cg.h).
#pragma once

struct BandPart {
    int ncb = 0, D = 0, B = 0, nc = 0, s = 0, p = 0, nS = 0, m = 0, ldx = 0, ntiles = 0, n_items_ch = 0, n_items_T = 0;
    size_t bytes = 0;
    std::vector<void*> owned;
    BandPartDev dev{};
    int2* d_chunks = nullptr;
    BandInvItem *d_items_ch = nullptr, *d_items_T = nullptr;
    BandPartTile* d_tiles = nullptr;
    double *G = nullptr, *V = nullptr, *T = nullptr, *Tinv = nullptr, *W = nullptr, *Lrow = nullptr, *Lcol = nullptr, *rdiag = nullptr,
           *TLrow = nullptr, *TLcol = nullptr, *Trdiag = nullptr, *Xs = nullptr;

    ~BandPart() { for (void* q : owned) hipFree(q); }

    // interior nodes per chunk: m + B ~ sqrt(B ncb) minimises the chain m + (p - 1) B of dependent block steps
    static int auto_m(int ncb, int B) { return std::max(std::max(B, 4), (int)std::lround(std::sqrt((double)B * ncb)) - B); }
    // worth it (and possible) only when the separator system fits the band kernels and there are at least three chunks
    static bool eligible(int ncb, int B, int m) { return B >= 1 && 2 * B - 1 <= PS_BAND_MAXB && m >= B && ncb >= 3 * m + 2 * B; }

    template <typename T2>
    int put(T2** out, const std::vector<T2>& v) {
        if (get(out, v.size())) return -1;
        if (!v.empty()) HIP_OK(hipMemcpy(*out, v.data(), v.size() * sizeof(T2), hipMemcpyHostToDevice));
        return 0;
    }
    template <typename T2>
    int get(T2** out, size_t n) {
        void* q = nullptr;
        const size_t b = std::max<size_t>(n, 1) * sizeof(T2);
        if (hipMalloc(&q, b) != hipSuccess) return fail("hipMalloc (partitioned band factorisation) failed");
        owned.push_back(q); bytes += b; *out = (T2*)q;
        return 0;
    }

    int build(int ncb_, int D_, int B_, int m_, hipStream_t st) {
        ncb = ncb_; D = D_; B = B_; m = m_; nc = ncb * D; s = B * D;
        std::vector<int2> chunks, seps;                       // (first node, nodes)
        for (int n = 0; n < ncb;) {
            int mm = std::min(m, ncb - n);
            int rest = ncb - (n + mm);
            if (rest <= 2 * B) { mm = ncb - n; rest = 0; }   // (no tail chunk smaller than a separator: the last chunk takes the rest)
            chunks.push_back(make_int2(n, mm)); n += mm;
            if (rest > 0) { seps.push_back(make_int2(n, B)); n += B; }
        }
        p = (int)chunks.size(); nS = (int)seps.size() * s;
        if (p < 2) return fail("partitioned band factorisation: fewer than two chunks");
        std::vector<int32_t> row_seg(nc), row_loc(nc), ch_row0(p), ch_n(p), sep_row0(seps.size());
        std::vector<int64_t> goff(p);
        std::vector<BandInvItem> items_ch, items_T;
        std::vector<BandPartTile> tiles;
        int64_t gsz = 0;
        int maxn = 0;
        for (int a = 0; a < p; ++a) {
            const int r0 = chunks[a].x * D, n = chunks[a].y * D;
            ch_row0[a] = r0; ch_n[a] = n; goff[a] = gsz; gsz += (int64_t)n * n; maxn = std::max(maxn, n);
            for (int l = 0; l < n; ++l) { row_seg[r0 + l] = a; row_loc[r0 + l] = l; }
            for (int c = 0; c < n; c += 4) items_ch.push_back(BandInvItem{r0, n, c, 0, goff[a]});
            for (int b = 0; b <= a; ++b) {
                const int nb = chunks[b].y * D;
                for (int i0 = 0; i0 < n; i0 += PS_BP_T)
                    for (int j0 = 0; j0 < nb; j0 += PS_BP_T)
                        if (a != b || j0 <= i0) tiles.push_back(BandPartTile{a, b, i0, j0});
            }
        }
        for (size_t x = 0; x < seps.size(); ++x) {
            sep_row0[x] = seps[x].x * D;
            for (int l = 0; l < s; ++l) { row_seg[sep_row0[x] + l] = -1 - (int)x; row_loc[sep_row0[x] + l] = l; }
        }
        for (int c = 0; c < nS; c += 4) items_T.push_back(BandInvItem{0, nS, c, 0, 0});
        ldx = std::max(maxn, nS);
        ntiles = (int)tiles.size(); n_items_ch = (int)items_ch.size(); n_items_T = (int)items_T.size();
        int32_t *d_row_seg, *d_row_loc, *d_ch_row0, *d_ch_n, *d_sep_row0;
        int64_t* d_goff;
        if (put(&d_chunks, chunks) || put(&d_row_seg, row_seg) || put(&d_row_loc, row_loc) || put(&d_ch_row0, ch_row0) ||
            put(&d_ch_n, ch_n) || put(&d_sep_row0, sep_row0) || put(&d_goff, goff) || put(&d_items_ch, items_ch) ||
            put(&d_items_T, items_T) || put(&d_tiles, tiles)) return -1;
        if (get(&G, (size_t)gsz) || get(&V, (size_t)nc * 2 * s) || get(&T, (size_t)nS * nS) || get(&Tinv, (size_t)nS * nS) ||
            get(&W, (size_t)nc * nS) || get(&Lrow, (size_t)nc * PS_BAND_W) || get(&Lcol, (size_t)nc * PS_BAND_W) ||
            get(&rdiag, (size_t)nc) || get(&TLrow, (size_t)nS * PS_BAND_W) || get(&TLcol, (size_t)nS * PS_BAND_W) ||
            get(&Trdiag, (size_t)nS) || get(&Xs, (size_t)nc * ldx)) return -1;
        // band entries in front of a chunk's (or the separator system's) first column are never written: zero once, the written
        // positions are the same in every run
        HIP_OK(hipMemsetAsync(Lrow, 0, (size_t)nc * PS_BAND_W * sizeof(double), st));
        HIP_OK(hipMemsetAsync(Lcol, 0, (size_t)nc * PS_BAND_W * sizeof(double), st));
        HIP_OK(hipMemsetAsync(TLrow, 0, (size_t)nS * PS_BAND_W * sizeof(double), st));
        HIP_OK(hipMemsetAsync(TLcol, 0, (size_t)nS * PS_BAND_W * sizeof(double), st));
        dev = BandPartDev{nc, nc, s, p, nS, 2 * s, d_row_seg, d_row_loc, d_ch_row0, d_ch_n, d_goff, d_sep_row0};
        return 0;
    }

    // A (nc x nc, row pitch lda, lower triangle read) -> its inverse as fp32 (row pitch ldo), 9 launches on `st`
    template <int DD_>
    int run(hipStream_t st, const double* A, int lda, float* Ainv, int ldo, int32_t* stat) {
        BandPartDev bp = dev; bp.lda = lda;
        hipLaunchKernelGGL(k_band_chol<DD_>, dim3(p), dim3(256), 0, st, 0, B, A, Lrow, Lcol, rdiag, stat, lda, (const int2*)d_chunks);
        hipLaunchKernelGGL(k_band_inverse_rl<true>, dim3(n_items_ch), dim3(256), 0, st, 0, (const double*)Lrow, (const double*)Lcol,
                           (const double*)rdiag, Xs, (float*)nullptr, (const BandInvItem*)d_items_ch, G, ldx);
        hipLaunchKernelGGL(k_bp_v, dim3(cdiv((long)nc * 2 * s, 256)), dim3(256), 0, st, bp, A, (const double*)G, V);
        hipLaunchKernelGGL(k_bp_t, dim3(cdiv((long)nS * nS, 256)), dim3(256), 0, st, bp, A, (const double*)V, T);
        const int nT = nS / DD_, BT = std::max(1, std::min(2 * B - 1, nT - 1));      // separator x couples to x - 1, x, x + 1: 2 B - 1 node offsets
        hipLaunchKernelGGL(k_band_chol<DD_>, dim3(1), dim3(256), 0, st, nT, BT, (const double*)T, TLrow, TLcol,
                           Trdiag, stat, nS, (const int2*)nullptr);
        hipLaunchKernelGGL(k_band_inverse_rl<true>, dim3(n_items_T), dim3(256), 0, st, 0, (const double*)TLrow, (const double*)TLcol,
                           (const double*)Trdiag, Xs, (float*)nullptr, (const BandInvItem*)d_items_T, Tinv, ldx);
        hipLaunchKernelGGL(k_bp_w, dim3(cdiv((long)nc * nS, 256)), dim3(256), 0, st, bp, (const double*)V, (const double*)Tinv, W);
        hipLaunchKernelGGL(k_bp_dense_sep, dim3(cdiv((long)nc * nS, 256)), dim3(256), 0, st, bp, (const double*)W, (const double*)Tinv, Ainv, ldo);
        hipLaunchKernelGGL(k_bp_dense, dim3(ntiles), dim3(256), 0, st, bp, (const BandPartTile*)d_tiles, (const double*)G,
                           (const double*)V, (const double*)W, Ainv, ldo);
        return 0;
    }
};
