// ps_core.hip -- host side of the C ABI declared in include/pyslam_hip.h:
// table upload, iteration-invariant structure (segment / pair / block lists),
// kernel sequencing on one HIP stream, hipEvent stage timers.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared \
//              -Iinclude pyslam_amd/csrc/ps_core.hip -o pyslam_amd/lib/libpyslam_hip.so
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "pyslam_hip.h"
#include "ps_kernels.h"
#include "ps_ransac.h"
#include "ps_photo.h"

namespace {

thread_local std::string g_err;

int fail(const std::string& msg) { g_err = msg; return -1; }

#define HIP_OK(expr)                                                                   \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess)                                                          \
            return fail(std::string(#expr) + ": " + hipGetErrorString(e_));            \
    } while (0)

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace

struct ps_problem {
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int D = 6, PW = 12;
    int P = 0, nr = 0, L = 0, nv = 0;
    long N = 0, Nl = 0, Np = 0;     // observations: all / on variable points / on variable poses
    long F = 0;                     // pose factors (edges + priors)
    long npairs = 0;
    int nnzb = 0;
    size_t dev_bytes = 0;
    std::vector<void*> allocs;

    // parameter tables
    double *poses = nullptr, *points = nullptr, *poses_snap = nullptr, *points_snap = nullptr;
    int32_t *pose_rid = nullptr, *point_vid = nullptr;
    // reprojection
    ObsGroup* ogroups = nullptr;
    LObs* lobs = nullptr;
    int32_t *lorig = nullptr, *lm_ptr = nullptr, *lm_point = nullptr;
    double *Z = nullptr, *Cinv = nullptr, *cvec = nullptr, *dxl = nullptr;
    LObs* pobs = nullptr;           // observation records in pose-sorted order (landmark slot + 1 in the pose bits)
    PItem* pitems = nullptr;
    int npitems = 0;
    int32_t* pitem_ptr = nullptr;
    double* ppartial = nullptr;
    int2* pairs = nullptr;
    int npair_items = 0, pair_per_xcd = 0;
    PairItem* pair_xitems = nullptr;   // work items in per-XCD dispatch order
    // tiled Schur (Z larger than the L2s): per-task partial blocks + per-block task lists
    int schur_tiles = 1, ncomb = 0;
    bool has_diag_tasks = false;    // some landmark is observed twice from one pose: a pair task writes a diagonal block
    double* Spart = nullptr;
    PairItem* comb_items = nullptr;     // slot, slotT, [start, end) into comb_tasks
    int32_t* comb_tasks = nullptr;
    // factors
    FactorGroup* fgroups = nullptr;
    int32_t *f_i = nullptr, *f_j = nullptr, *f_grp = nullptr;
    double *f_Tinv = nullptr, *fscratch = nullptr;
    int32_t *eslots = nullptr, *eptr = nullptr, *eslot_diag = nullptr, *gptr = nullptr, *gitems = nullptr;
    int2* eitems = nullptr;
    int nes = 0;
    // reduced system
    int32_t *row_ptr = nullptr, *col_idx = nullptr, *diag_slot = nullptr;
    double* red = nullptr;          // [S (nnzb*D*D) | g (nr*D) | cost]
    long red_count = 0;
    double *S = nullptr, *g = nullptr, *red_cost = nullptr;
    std::vector<int32_t> h_row_ptr, h_col_idx;
    std::vector<int32_t> h_vid_of_slot;   // landmarks are stored in locality order; external order is vid
    // pcg
    double *x = nullptr, *r = nullptr, *z = nullptr, *p0 = nullptr, *p1 = nullptr, *q = nullptr, *Minv = nullptr;
    double *rz_part = nullptr, *rr_part = nullptr, *pq_part = nullptr, *hist = nullptr;
    int npartA = 0, npartB = 0, hist_cap = 0, last_pcg_iters = 0;
    // fused CG (one launch per iteration) on the block-Jacobi-scaled system
    int pcg_variant = 1;            // 1 = fused single-reduction CG, 0 = classic two-launch PCG
    int pcg_chunk = 8;              // launches between host polls of the 'done' flag
    double *Linv = nullptr, *cg_r[2] = {}, *cg_w[2] = {}, *cg_s[2] = {}, *cg_gd[2] = {}, *cg_xh = nullptr;
    int32_t* brow_of = nullptr;
    double* cg_p = nullptr;
    double* Saug = nullptr;         // scaled (and, with a coarse level, augmented) matrix the CG runs on
    int32_t* ident_slot = nullptr;
    // two-level preconditioner (coarse level), built lazily by build_coarse()
    int coarse_req = -1;            // requested number of groups: -1 = auto, 0 = off
    int G = 0, ncb = 0, nc = 0, nr_aug = 0, nnzb_aug = 0;
    size_t cg_cap = 0, saug_cap = 0; // CG vectors / matrix are allocated for this many block rows / blocks
    int32_t *pnode = nullptr, *slo = nullptr, *shi = nullptr, *run_lo = nullptr, *run_hi = nullptr,
            *arow_ptr = nullptr, *acol_idx = nullptr, *aug_slot = nullptr, *fine_nnz = nullptr;
    double *pw0 = nullptr, *pw1 = nullptr, *SZ = nullptr, *Ac = nullptr, *tvec = nullptr, *chol_scratch = nullptr;
    // coarse basis P_iq = w(i,q) B_i: B_i = L_i^T Ad(T_i) (coarse_basis 1, rigid-motion aware) or I (0)
    int coarse_basis = 1;
    double *Bmat = nullptr, *bgv = nullptr, *SB = nullptr, *BSZ = nullptr;
    int32_t *ent_ptr = nullptr, *ent_q = nullptr, *ent_lo = nullptr, *ent_hi = nullptr;   // explicit PCG: non-empty (row, node) runs
    int32_t *seg_ptr = nullptr, *seg_ent = nullptr, *seg_row = nullptr;                     // ... grouped by (node, node') for A_c
    int max_row_ents = 0;
    int32_t* pose_of_rid = nullptr;
    // coarse factor L_c^-1 (and transpose), double-buffered: with "coarse_lag" the factorisation of THIS
    // iteration's A_c runs on a side stream while the CG iterates with the previous iteration's factor
    double *Lci2[2] = {}, *LciT2[2] = {};
    int lci_cur = 0;                // buffer the current augmented system was built with
    int lci_next = -1;              // buffer holding (or receiving, see side_pending) the newest factor; -1: none
    bool xcg_side_todo = false;     // explicit PCG: lagged setup whose side-stream half is not enqueued yet
    bool side_pending = false;      // a side-stream factorisation is in flight: wait for ev_chol before reuse
    int coarse_lag = 1;
    hipStream_t side = nullptr;
    hipEvent_t ev_ac = nullptr, ev_chol = nullptr;
    int32_t* lag_status = nullptr;  // ST_DIAG_FAIL of the side-stream factorisation
    double* Mc = nullptr;           // split mode: dense coarse-coarse block M of the lagged system
    bool mc_active = false;         // the current system was built with a lagged factor in split mode
    bool coarse_built = false;
    int cg_ablate = 0, schur_ablate = 0, lm_ablate = 0;
    int max_pose_obs = 0;           // most observations on one variable pose
    int mo_fused = 1;               // motion-only problems: one launch per iteration (k_motion_only_iteration)
    double* mo_partials = nullptr;
    bool status_clean = true;       // no failure flag can be pending in the device status words
    int direct_max = 90;            // reduced systems up to this many unknowns are solved directly (0: never)
    double *dA = nullptr, *dLi = nullptr, *dLiT = nullptr;
    int big_chol = 1;               // nc > 90: multi-workgroup blocked factorisation (0: one workgroup out of L2)
    // explicit two-level PCG (long sparse chains)
    int explicit_ok = 1;
    bool cg_explicit = false;
    double *xstate = nullptr, *xy = nullptr, *xp2 = nullptr;
    int cg_lds = 1;                 // small systems: k_cg_fused_lds (whole vector through LDS)
    int prof_every = 1;             // profiling level 1: time the Schur kernel of every n-th linearisation only
    long prof_tick = 0;
    int cg_fallbacks = 0;           // solves repeated with the classic PCG after a breakdown of the pipelined CG
    int cg_margin = 4;              // CG launches enqueued beyond the previous solve's iteration count
    bool cg_two_level_reduce = false, cg_short_rows = false;
    double* cg_tot = nullptr;
    // split mode (large systems): coarse rows are owned by k_cg_reduce_split
    bool cg_split = false;
    int cg_split_min_rows = 1024;
    int cg_explicit_min_rows = -1;  // explicit two-level PCG beyond this many reduced poses (-1: 400 for pose-graph rows, 540 for BA rows)
    double *cg_U = nullptr, *cg_cgd[2] = {}, *cg_ab = nullptr;
    int cg_launched = 0;            // CG launches enqueued since the last setup
    bool cov_ready = false;         // ps_covariance_begin has linearised and set the reduced solver up; cleared by linearize()
    std::vector<int32_t> h_slot_of_vid;
    int ell_wf = 0, ell_wc = 0;     // two-class ELL widths of the CG matrix (0 = CSR)
    // scalars
    double *cost_partials = nullptr, *scalars = nullptr, *h_scalars = nullptr;
    int32_t *status = nullptr, *h_status = nullptr;
    double* h_scalars_dev = nullptr;     // device-side aliases of the pinned, host-mapped result words
    int32_t* h_status_dev = nullptr;
    double *h_shard = nullptr, *h_shard_dev = nullptr;   // host-mapped copy of shard_buf (sharded iteration)
    long long *h_seq = nullptr, *h_seq_dev = nullptr;   // sequence number stamped by the last k_reduce3 workgroup
    long long seq = 0;
    int32_t* arrivals = nullptr;
    int ncost_obs = 0, ncost_fac = 0, nsq = 0;
    // native RCCL: function pointer + communicator handed over by the binding (ps_set_collective)
    typedef int (*allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
    allreduce_fn nccl_allreduce = nullptr;
    void* nccl_comm = nullptr;
    double* shard_buf = nullptr;    // {cost, ||dx_point||^2} of this landmark shard, for the caller's all-reduce
    bool shard_out = false;         // k_reduce3 writes cost / ||dx_point||^2 there instead of into scalars
    double *sq_part_l = nullptr, *sq_part_p = nullptr;   // per-workgroup partials of ||dx_point||^2, ||dx_pose||^2
    int nsq_l = 0, nsq_p = 0;
    // profiling
    int profiling = 0;              // 0 off, 1 = iteration total + Schur kernel only, 2 = every stage
    hipEvent_t ev[2 * PS_NUM_STAGES] = {};
    std::vector<std::pair<int, int>> pending;   // (stage, event slot) recorded, not yet read
    double stage_ms[PS_NUM_STAGES] = {};
    int64_t stage_n[PS_NUM_STAGES] = {};
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;

    template <typename T>
    int alloc(T** out, size_t n) {
        *out = nullptr;
        const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) return fail(std::string("hipMalloc: ") + hipGetErrorString(e));
        allocs.push_back(p);
        dev_bytes += bytes;
        *out = (T*)p;
        return 0;
    }
    template <typename T>
    int upload(T** out, const std::vector<T>& v) {
        if (alloc(out, v.size())) return -1;
        if (!v.empty()) HIP_OK(hipMemcpy(*out, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
        return 0;
    }
    template <typename T>
    int upload(T** out, const T* src, size_t n) {
        if (alloc(out, n)) return -1;
        if (n) HIP_OK(hipMemcpy(*out, src, n * sizeof(T), hipMemcpyHostToDevice));
        return 0;
    }
};

namespace {

// ---- stage timers ---------------------------------------------------------
struct StageTimer {
    ps_problem* h;
    int stage;
    hipEvent_t a = nullptr, b = nullptr;
    int slot = -1;
    StageTimer(ps_problem* h_, int st, int level = 2) : h(h_), stage(st) {
        if (h->profiling < level) return;
        if (h->profiling == 1 && h->prof_every > 1 && h->prof_tick % h->prof_every != 0) return;   // sampled launches only
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

int sync(ps_problem* h) {
    HIP_OK(hipStreamSynchronize(h->stream));
    drain_timers(h);
    return 0;
}

// End of a published iteration: watch the sequence word k_reduce3's last workgroup writes to pinned host
// memory (a few microseconds cheaper than a stream synchronisation); falls back to the synchronisation
// when stage timers need their events or the word does not show up in ~1 s.
int wait_published(ps_problem* h) {
    volatile long long* w = h->h_seq;
    for (long spins = 0; spins < 400000000L; ++spins) {
        if (*w == h->seq) {
            if (h->pending.empty()) return 0;
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
int launch_factor_pass(ps_problem* h, double lambda) {
    if (h->F == 0) return 0;
    hipLaunchKernelGGL(k_factor_pass<D>, dim3(cdiv(h->F, 4)), dim3(256), 0, h->stream, (int)h->F, h->f_i,
                       h->f_j, h->f_Tinv, h->f_grp, h->fgroups, h->poses, h->fscratch);
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
    const bool timing = getenv("PS_CREATE_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "build_coarse: %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    const int nr = h->nr, D = h->D;
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
    // iteration on.  Pose-graph-like rows (C2: 11 blocks per row, hundreds of CG iterations): one interval per 40
    // poses, up to 255; bundle-adjustment rows (C4: 80 blocks per row, ~20 iterations -- the factorisation must fit
    // beside a short CG): one per 20 poses, up to 112 (C4: 42 iterations / 3.7 ms folded at 48 -> 20 / 2.6 ms at 100).
    // Measured crossover against the folded single-launch CG (whose coarse level is capped at 12 intervals):
    // pose graphs 400 poses (600: 2.6 -> 1.4 ms, 1 000: 6.1 -> 1.5 ms), bundle adjustment 540 (700: 1.67 -> 1.28 ms).
    const int xmin = std::min(h->cg_split_min_rows, h->cg_explicit_min_rows >= 0 ? h->cg_explicit_min_rows : (sparse_rows ? 400 : 540));
    h->cg_explicit = h->explicit_ok && G != 0 && nr > xmin;
    if (h->cg_explicit && h->coarse_req < 0)
        G = sparse_rows ? std::min(255, std::max(48, nr / 40)) : std::min(112, std::max(48, nr / 20));
    G = std::min(G, h->cg_explicit ? 255 : Gmax);
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
        for (int q = 0; q < ncb; ++q) {
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
    if (h->alloc(&h->BSZ, (h->cg_explicit ? eq.size() : (size_t)nr * ncb) * D * D) || h->alloc(&h->Bmat, (size_t)nr * D * D) ||
        h->alloc(&h->bgv, (size_t)nr * D) || h->alloc(&h->SB, (size_t)aci.size() * D * D)) return -1;
    if ((!h->cg_explicit && h->alloc(&h->SZ, (size_t)nr * ncb * D * D)) || h->alloc(&h->Ac, (size_t)h->nc * h->nc) ||
        h->alloc(&h->Lci2[0], (size_t)h->nc * h->nc) || h->alloc(&h->LciT2[0], (size_t)h->nc * h->nc) ||
        h->alloc(&h->Lci2[1], (size_t)h->nc * h->nc) || h->alloc(&h->LciT2[1], (size_t)h->nc * h->nc) ||
        h->alloc(&h->tvec, (size_t)h->nc) || h->alloc(&h->chol_scratch, 2 * (size_t)h->nc * h->nc)) return -1;
    if (h->cg_explicit && (h->alloc(&h->xstate, 8) || h->alloc(&h->xy, (size_t)h->nc) || h->alloc(&h->xp2, (size_t)nr * D))) return -1;
    lap("allocations");
    if (!h->lag_status) {
        if (h->alloc(&h->lag_status, ST_NWORDS)) return -1;
        HIP_OK(hipMemsetAsync(h->lag_status, 0, ST_NWORDS * sizeof(int32_t), h->stream));
    }
    if (!h->side) {
        int prio_lo = 0, prio_hi = 0;                      // lowest priority: the side work must not delay the CG launches
        HIP_OK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
        HIP_OK(hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, prio_lo));
        HIP_OK(hipEventCreateWithFlags(&h->ev_ac, hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&h->ev_chol, hipEventDisableTiming));
    }
    if (h->side_pending) { HIP_OK(hipStreamWaitEvent(h->stream, h->ev_chol, 0)); h->side_pending = false; }
    h->lci_next = -1; h->lci_cur = 0;
    if (ensure_cg_buffers(h, h->nr_aug, h->nnzb_aug)) return -1;
    HIP_OK(hipMemsetAsync(h->Saug, 0, (size_t)h->nnzb_aug * D * D * sizeof(double), h->stream));
    return 0;
}

// factor A_c = L_c L_c^T and form L_c^-1 (+ transpose) into buffer `buf`: LDS-resident single workgroup up to 90
// unknowns, blocked over the whole chip beyond
template <int D>
int coarse_factor(ps_problem* h, hipStream_t st, int buf, int32_t* stat) {
    const int nc = h->nc, ncb = h->ncb;
    if (nc <= 90) {                                        // 2 nc^2 doubles of dynamic LDS (<= 130 KB)
        const size_t chol_lds = 2 * (size_t)nc * nc * sizeof(double);
        HIP_OK(hipFuncSetAttribute((const void*)k_coarse_chol<D, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)chol_lds));
        hipLaunchKernelGGL((k_coarse_chol<D, true>), dim3(1), dim3(1024), chol_lds, st, ncb, h->Ac, h->Lci2[buf],
                           h->LciT2[buf], stat, nullptr);
    } else if (h->big_chol) {
        // blocked factorisation over the whole chip (chol_scratch: working copy of A_c, then the tiles' inverses)
        double* A = h->chol_scratch;
        double* Tinv = A + (size_t)nc * nc;
        HIP_OK(hipMemcpyAsync(A, h->Ac, (size_t)nc * nc * sizeof(double), hipMemcpyDeviceToDevice, st));
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

template <int D>
int cg_fused_setup(ps_problem* h, int max_iters, bool allow_lag = false, bool rhs_only = false) {
    const int nr = h->nr, cap = h->hist_cap;
    if (max_iters + 2 > cap) return fail("pcg max_iters exceeds the history buffer (4096)");
    if (!h->coarse_built && build_coarse(h)) return -1;
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
    // block-Jacobi factors + the start vectors of the scaled system (r = Linv g, w = s = p = x = 0)
    hipLaunchKernelGGL(k_block_jacobi_factor<D>, dim3(cdiv(nr, 256)), dim3(256), 0, h->stream, nr, h->diag_slot,
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
        const bool lag = allow_lag && h->coarse_lag && h->lci_next >= 0 && (!h->cg_split || (long)h->nnzb > 24L * nr);
        if (lag && h->cg_split && !h->Mc && h->alloc(&h->Mc, (size_t)nc * nc)) return -1;
        h->mc_active = lag && h->cg_split;
        const int rpw = nc >= 192 ? 4 : 1;                 // fine block rows per border workgroup
        const int border_lds = (int)((size_t)rpw * D * nc * sizeof(double));
        HIP_OK(hipFuncSetAttribute((const void*)k_coarse_border<D>, hipFuncAttributeMaxDynamicSharedMemorySize, border_lds));
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

// enqueue `count` more CG launches (launch n runs iteration k = n - 1; converged launches exit at once)
template <int D>
void cg_fused_launch(ps_problem* h, double tol, int count) {
    const int cap = h->hist_cap;
    const int rows = h->cg_split ? h->nr : h->nr_aug;      // matrix rows handled by k_cg_fused
    const int ncbs = h->cg_split ? h->ncb : 0;
    const double tol2 = tol * tol;
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
    if (h->G)
        hipLaunchKernelGGL(k_coarse_recover<D>, dim3(cdiv((long)nr * D, 256)), dim3(256), 0, h->stream, nr, h->ncb,
                           h->pnode, h->pw0, h->pw1, h->Linv, h->Lci2[h->lci_cur], h->cg_xh, h->x, gate, h->Bmat);
    else
        hipLaunchKernelGGL(k_cg_unscale<D>, dim3(cdiv((long)nr * D, 256)), dim3(256), 0, h->stream, nr, h->Linv,
                           h->cg_xh, h->x, gate);
}

int cg_report(ps_problem* h, int* iters_out, double* relres_out) {
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

// synchronous solve (staged API): poll the convergence flag every chunk
template <int D>
int cg_fused_run(ps_problem* h, double tol, int max_iters, int* iters_out, double* relres_out, bool rhs_only = false) {
    if (cg_fused_setup<D>(h, max_iters, false, rhs_only)) return -1;
    int chunk = std::max(h->pcg_chunk, h->last_pcg_iters + 2);
    bool done = false;
    while (!done) {
        const int m = std::min(chunk, max_iters + 2 - h->cg_launched);
        cg_fused_launch<D>(h, tol, m);
        HIP_OK(hipMemcpyAsync(h->h_status, h->status, ST_NWORDS * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
        HIP_OK(hipMemcpyAsync(h->h_scalars, h->scalars, SC_NWORDS * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_OK(hipStreamSynchronize(h->stream));
        done = h->h_status[ST_PCG_DONE] != 0 || h->cg_launched >= max_iters + 2;
        chunk = h->pcg_chunk;
    }
    if (h->h_status[ST_PCG_DONE] == 2 && !h->h_status[ST_DIAG_FAIL]) {
        // the pipelined (Chronopoulos-Gear) recurrences lost positivity -- rounding on an ill-conditioned system, seen on
        // unit right-hand sides of covariance columns -- : repeat with the classic two-launch block-Jacobi PCG
        ++h->cg_fallbacks;
        return pcg_run<D>(h, tol, max_iters, iters_out, relres_out);
    }
    cg_fused_recover<D>(h, nullptr);
    return cg_report(h, iters_out, relres_out);
}

int linearize(ps_problem* h, double lambda) {
    ++h->prof_tick;
    h->cov_ready = false;
    h->status_clean = false;
    HIP_OK(hipMemsetAsync(h->red, 0, h->red_count * sizeof(double) + ST_NWORDS * sizeof(int32_t), h->stream));   // [S | g | cost | status]
    if (h->nv > 0) {
        StageTimer t(h, PS_ST_LANDMARK);
        hipLaunchKernelGGL(k_landmark_pass, dim3(cdiv(h->nv, 256 / PS_LM_GROUP)), dim3(256), 0, h->stream, h->nv, h->lm_ptr,
                           h->lm_point, h->lobs, h->poses, h->points, h->pose_rid, h->ogroups, lambda, h->Z,
                           h->Cinv, h->cvec, h->status, h->lm_ablate);
    }
    bool fin_in_combine = false;
    if (h->npitems > 0) {
        StageTimer t(h, PS_ST_POSE);
        hipLaunchKernelGGL(k_pose_pass, dim3(h->npitems), dim3(256), 0, h->stream, h->pitems, h->pobs,
                           h->poses, h->points, h->ogroups, h->Cinv, h->cvec, h->ppartial);
        // tiled Schur: the combine launch also finalizes the poses (unless a task writes a diagonal block)
        fin_in_combine = h->Spart && h->npair_items > 0 && !h->has_diag_tasks && h->D == 6;
        if (!fin_in_combine)
            hipLaunchKernelGGL(k_pose_finalize, dim3(h->nr), dim3(64), 0, h->stream, h->nr, h->pitem_ptr,
                               h->ppartial, h->diag_slot, lambda, h->S, h->g);
    }
    if (h->npair_items > 0) {
        StageTimer t(h, PS_ST_SCHUR, 1);
        hipLaunchKernelGGL(k_schur_pairs, dim3(8 * (h->pair_per_xcd / 4)), dim3(256), 0, h->stream,
                           h->pair_per_xcd, h->pair_xitems, h->pairs, h->Z, h->S, h->Spart, h->schur_ablate);
        if (h->Spart)
            hipLaunchKernelGGL(k_schur_combine, dim3(cdiv(h->ncomb, 4) + (fin_in_combine ? cdiv(h->nr, 4) : 0)), dim3(256), 0,
                               h->stream, h->ncomb, h->comb_items, h->comb_tasks, h->Spart, h->S,
                               fin_in_combine ? h->nr : 0, h->pitem_ptr, h->ppartial, h->diag_slot, lambda, h->g);
    }
    if (h->F > 0 && h->nr > 0) {
        StageTimer t(h, PS_ST_EDGES);
        if (h->D == 6) launch_factor_pass<6>(h, lambda); else launch_factor_pass<3>(h, lambda);
    }
    return 0;
}

// cost partials into cost_partials[0..n); returns n.  The caller reduces them.
int cost_partials_pass(ps_problem* h, int include_all, const int32_t* gate) {
    int n = 0;
    if (h->N > 0) {
        hipLaunchKernelGGL(k_cost_reproj, dim3(h->ncost_obs), dim3(256), 0, h->stream, h->N, h->lobs, h->poses,
                           h->points, h->pose_rid, h->point_vid, h->ogroups, include_all, h->cost_partials, gate);
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

int backsub(ps_problem* h, const int32_t* gate = nullptr, bool fuse_update = false) {
    if (h->nv == 0) return 0;
    StageTimer t(h, PS_ST_BACKSUB);
    if (fuse_update)       // + full-step landmark update + SE(3) retraction of the poses in the same launch
        hipLaunchKernelGGL(k_backsub, dim3(h->nsq_l + h->nsq_p), dim3(256), 0, h->stream, h->nv, h->lm_ptr, h->lobs,
                           h->pose_rid, h->Z, h->Cinv, h->cvec, h->x, h->dxl, h->sq_part_l, gate,
                           h->nsq_l, h->lm_point, h->points, h->P, h->poses, h->sq_part_p);
    else
        hipLaunchKernelGGL(k_backsub, dim3(h->nsq_l), dim3(256), 0, h->stream, h->nv, h->lm_ptr, h->lobs,
                           h->pose_rid, h->Z, h->Cinv, h->cvec, h->x, h->dxl, h->sq_part_l, gate,
                           h->nsq_l, (const int32_t*)nullptr, (double*)nullptr, 0, (double*)nullptr, (double*)nullptr);
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

int apply_update(ps_problem* h, double step, const int32_t* gate = nullptr, bool with_norm = false) {
    StageTimer t(h, PS_ST_UPDATE);
    if (h->nr > 0) {
        double* sq = with_norm ? h->sq_part_p : nullptr;
        if (h->D == 6)
            hipLaunchKernelGGL(k_update_poses<6>, dim3(cdiv(h->P, 256)), dim3(256), 0, h->stream, h->P, h->pose_rid, h->x, step, h->poses, sq, gate);
        else
            hipLaunchKernelGGL(k_update_poses<3>, dim3(cdiv(h->P, 256)), dim3(256), 0, h->stream, h->P, h->pose_rid, h->x, step, h->poses, sq, gate);
    }
    if (h->nv > 0)
        hipLaunchKernelGGL(k_update_points, dim3(cdiv((long)h->nv * 3, 256)), dim3(256), 0, h->stream, h->nv,
                           h->lm_point, h->dxl, step, h->points, gate);
    return 0;
}

// back-substitution, update, cost and ||dx||^2 with ONE final reduction launch.  `gate` (device
// status words) makes every kernel a no-op until the CG has flagged convergence.
int gn_tail(ps_problem* h, int linesearch, const int32_t* gate, bool publish = false) {
    // line-search order (cost AFTER the step): back-substitution, landmark update and pose retraction
    // are one launch when the problem has landmarks (then D == 6)
    const bool fused = linesearch && h->nv > 0 && h->nr > 0 && h->D == 6;
    if (backsub(h, gate, fused)) return -1;
    int ncost = 0;
    if (!linesearch) { StageTimer t(h, PS_ST_COST); ncost = cost_partials_pass(h, 0, gate); }
    if (!fused && apply_update(h, 1.0, gate, true)) return -1;
    if (linesearch) { StageTimer t(h, PS_ST_COST); ncost = cost_partials_pass(h, 1, gate); }
    double* o_cost = h->shard_out ? h->shard_buf : h->scalars + (linesearch ? SC_COST : SC_LINCOST);
    double* o_dxl = h->shard_out ? h->shard_buf + 1 : h->scalars + SC_DXL2;
    hipLaunchKernelGGL(k_reduce3, dim3(3), dim3(256), 0, h->stream,
                       ncost, h->cost_partials, o_cost,
                       h->nsq_p, h->sq_part_p, h->nr > 0 ? h->scalars + SC_DXP2 : nullptr,
                       h->nsq_l, h->sq_part_l, (h->nv > 0 || h->shard_out) ? o_dxl : nullptr, gate,
                       h->status, h->scalars, publish ? h->h_status_dev : nullptr, publish ? h->h_scalars_dev : nullptr,
                       h->arrivals, publish ? h->h_seq_dev : nullptr, publish ? ++h->seq : 0LL);
    return 0;
}

// ---- explicit two-level PCG (long sparse chains; kernels k_xcg_*) ---------------------------------
template <int D>
int xcg_setup(ps_problem* h, int max_iters, bool allow_lag) {
    const int nr = h->nr, ncb = h->ncb, nc = h->nc;
    if (max_iters + 2 > h->hist_cap) return fail("pcg max_iters exceeds the history buffer (4096)");
    hipLaunchKernelGGL(k_block_jacobi_factor<D>, dim3(cdiv(nr, 256)), dim3(256), 0, h->stream, nr, h->diag_slot,
                       h->S, h->Linv, h->status, h->g, h->cg_r[0], h->cg_w[0], h->cg_s[0], h->cg_p, h->cg_xh,
                       h->poses, h->pose_of_rid, h->coarse_basis, h->Bmat, h->bgv);
    hipLaunchKernelGGL(k_scale_blocks<D>, dim3(h->nnzb), dim3(64), 0, h->stream, nr, h->row_ptr, h->col_idx,
                       h->brow_of, h->Linv, h->S, h->aug_slot, h->Saug, h->Bmat, h->SB);
    if (h->side_pending) { HIP_OK(hipStreamWaitEvent(h->stream, h->ev_chol, 0)); h->side_pending = false; }
    hipLaunchKernelGGL(k_xcoarse_rowsums<D>, dim3(nr), dim3(256), (size_t)(h->max_row_ents + 1) * D * D * sizeof(double), h->stream,
                       nr, h->ent_ptr, h->ent_q, h->ent_lo, h->ent_hi, h->acol_idx, h->pnode, h->pw0, h->pw1, h->SB, h->Bmat, h->BSZ);
    hipLaunchKernelGGL(k_xcoarse_matrix<D>, dim3(cdiv((long)ncb * ncb * D * D, 256)), dim3(256), 0, h->stream,
                       ncb, h->seg_ptr, h->seg_ent, h->seg_row, h->pnode, h->pw0, h->pw1, h->BSZ, h->Ac);
    // A_c^-1 lives in LciT2[b] (the transposed factor is not used on this path).  It only PRECONDITIONS here, so any
    // symmetric positive definite stand-in keeps the CG exact: whole-iteration calls use the inverse formed from the
    // PREVIOUS iteration's A_c and factor the current one on the side stream while the CG iterates (the factorisation,
    // triangular inverse and product are 5.6 ms of the 12 ms iteration at C2 with 256 nodes).
    const bool lag = allow_lag && h->coarse_lag && h->lci_next >= 0;
    const int32_t* lagst = nullptr;
    h->xcg_side_todo = false;
    if (lag) {
        h->lci_cur = h->lci_next;
        HIP_OK(hipEventRecord(h->ev_ac, h->stream));       // A_c complete; the side work is enqueued by xcg_side_enqueue
        h->xcg_side_todo = true;
        lagst = h->lag_status;
    } else {
        const int buf = h->lci_cur;
        if (coarse_factor<D>(h, h->stream, buf, h->status)) return -1;
        hipLaunchKernelGGL(k_xcg_ainv, dim3(cdiv(nc, PS_AI_T) * (cdiv(nc, PS_AI_T) + 1) / 2), dim3(256), 0, h->stream, nc, h->Lci2[buf], (float*)h->LciT2[buf]);
        h->lci_next = buf;
    }
    HIP_OK(hipMemsetAsync(h->xstate, 0, 8 * sizeof(double), h->stream));
    HIP_OK(hipMemsetAsync(h->xp2, 0, (size_t)nr * D * sizeof(double), h->stream));
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
    if (coarse_factor<D>(h, h->side, nb, h->lag_status)) return -1;
    hipLaunchKernelGGL(k_xcg_ainv, dim3(cdiv(nc, PS_AI_T) * (cdiv(nc, PS_AI_T) + 1) / 2), dim3(256), 0, h->side, nc, h->Lci2[nb], (float*)h->LciT2[nb]);
    HIP_OK(hipEventRecord(h->ev_chol, h->side));
    h->lci_next = nb; h->side_pending = true;
    return 0;
}

template <int D>
void xcg_launch(ps_problem* h, double tol, int count) {
    const int nr = h->nr, ncb = h->ncb, nc = h->nc;
    const int n_pq = cdiv(nr, PS_XCG_ROWS), n_rz = cdiv(nr, PS_XCG_DROWS);
    double* pbuf[2] = {h->cg_p, h->xp2};
    for (int i = 0; i < count; ++i, ++h->cg_launched) {
        const int k = h->cg_launched, b = k & 1;
        hipLaunchKernelGGL(k_xcg_spmv<D>, dim3(n_pq), dim3(64 * PS_XCG_ROWS), 0, h->stream, nr, h->arow_ptr, h->acol_idx,
                           h->ell_wf, h->Saug, h->cg_s[0], pbuf[b ^ 1], pbuf[b], h->cg_w[0], h->cg_gd[0], n_rz, h->cg_gd[1],
                           h->xstate, k, tol * tol, h->hist, h->status, h->scalars);
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
        if (read_scalars(h)) return -1;
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
    if (!h->dA && (h->alloc(&h->dA, (size_t)n * n) || h->alloc(&h->dLi, (size_t)n * n) || h->alloc(&h->dLiT, (size_t)n * n)))
        return -1;
    hipLaunchKernelGGL(k_bsr_to_dense<D>, dim3(1), dim3(256), 0, h->stream, nr, h->nnzb, h->brow_of, h->col_idx, h->S, h->dA);
    const size_t chol_lds = 2 * (size_t)n * n * sizeof(double);
    HIP_OK(hipFuncSetAttribute((const void*)k_coarse_chol<D, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)chol_lds));
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
        int count = h->last_pcg_iters > 0 ? h->last_pcg_iters + 2 : 32;
        for (;;) {
            count = std::min(count, max_iters + 1 - h->cg_launched);
            // (the side-stream factorisation goes in after the first few iterations' launches: early enough to
            // finish beside the CG, late enough not to delay its start on the host)
            const int head = std::min(count, 12);
            xcg_launch<D>(h, tol, head);
            if (xcg_side_enqueue<D>(h)) return -1;
            xcg_launch<D>(h, tol, count - head);
            hipLaunchKernelGGL(k_cg_unscale<D>, dim3(cdiv((long)h->nr * D, 256)), dim3(256), 0, h->stream, h->nr, h->Linv,
                               h->cg_xh, h->x, (const int32_t*)h->status);
            tp.stop();
            if (gn_tail(h, linesearch, h->status, true)) return -1;
            if (total) total->stop();
            if (wait_published(h)) return -1;
            if (h->h_status[ST_PCG_DONE] != 0) break;
            if (h->cg_launched >= max_iters + 1) {          // not converged within max_iters: take the step anyway
                hipLaunchKernelGGL(k_cg_unscale<D>, dim3(cdiv((long)h->nr * D, 256)), dim3(256), 0, h->stream, h->nr, h->Linv,
                                   h->cg_xh, h->x, (const int32_t*)nullptr);
                if (gn_tail(h, linesearch, nullptr, true) || wait_published(h)) return -1;
                break;
            }
            count = std::max(8, h->cg_launched / 2);
        }
        return cg_report(h, iters_out, relres_out);
    }
    if (cg_fused_setup<D>(h, max_iters, true)) return -1;
    int count = h->last_pcg_iters > 0 ? h->last_pcg_iters + h->cg_margin : 16;
    for (;;) {
        count = std::min(count, max_iters + 2 - h->cg_launched);
        cg_fused_launch<D>(h, tol, count);
        cg_fused_recover<D>(h, h->status);
        tp.stop();
        if (gn_tail(h, linesearch, h->status, true)) return -1;
        if (total) total->stop();                       // close the iteration timer before the sync
        if (wait_published(h)) return -1;               // k_reduce3 has published status + scalars to host memory
        if (h->h_status[ST_PCG_DONE] == 2 && !h->h_status[ST_DIAG_FAIL]) {
            // breakdown of the pipelined recurrences (the gated tail applied nothing): classic PCG, then the tail
            ++h->cg_fallbacks;
            if (pcg_run<D>(h, tol, max_iters, iters_out, relres_out)) return -1;
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
    return cg_report(h, iters_out, relres_out);
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

// ===========================================================================
// C ABI
// ===========================================================================
extern "C" {

const char* ps_last_error(void) { return g_err.c_str(); }

int ps_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int ps_problem_destroy(ps_problem* h) {
    if (!h) return 0;
    hipStreamSynchronize(h->stream);
    if (h->side) hipStreamSynchronize(h->side);
    for (void* p : h->allocs) hipFree(p);
    if (h->h_scalars) hipHostFree(h->h_scalars);
    if (h->h_status) hipHostFree(h->h_status);
    if (h->h_seq) hipHostFree(h->h_seq);
    if (h->h_shard) hipHostFree(h->h_shard);
    for (hipEvent_t e : h->ev_pool) hipEventDestroy(e);
    if (h->side) { hipStreamSynchronize(h->side); hipStreamDestroy(h->side); hipEventDestroy(h->ev_ac); hipEventDestroy(h->ev_chol); }
    if (h->own_stream) hipStreamDestroy(h->stream);
    delete h;
    return 0;
}

int ps_problem_create(const ps_problem_desc* d, void* stream, ps_problem** out) {
    const bool timing = getenv("PS_CREATE_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "ps_problem_create: %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    if (!d || !out) return fail("null argument");
    *out = nullptr;
    if (d->dof != 6 && d->dof != 3) return fail("dof must be 6 (SE3) or 3 (SE2)");
    if (d->num_obs > 0 && d->dof != 6) return fail("reprojection blocks need SE(3) poses");
    if (d->num_poses >= (1 << 24)) return fail("more than 2^24 poses");
    if (d->num_obs_groups > 255) return fail("more than 255 observation groups");
    if (d->num_obs >= (1L << 31) / 18) return fail("too many observations for 32-bit indexing");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("no HIP device visible");

    ps_problem* h = new ps_problem();
    struct Guard { ps_problem* h; bool ok = false; ~Guard() { if (!ok) ps_problem_destroy(h); } } guard{h};
    if (stream) h->stream = (hipStream_t)stream;
    else { HIP_OK(hipStreamCreate(&h->stream)); h->own_stream = true; }
    const int D = h->D = d->dof;
    const int PW = h->PW = (D == 6 ? 12 : 6);
    const int DD = D * D;
    h->P = d->num_poses; h->L = d->num_points; h->N = d->num_obs;
    const int P = h->P, L = h->L;
    const long N = h->N;

    // ---- parameter tables
    if (h->upload(&h->poses, d->poses, (size_t)P * PW)) return -1;
    if (h->upload(&h->points, d->points, (size_t)L * 3)) return -1;
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
    std::vector<int32_t> first_pose(L, INT32_MAX);
    for (long i = 0; i < N; ++i) {
        const int pt = d->obs_point[i];
        if (pt >= 0 && pt < L) first_pose[pt] = std::min(first_pose[pt], d->obs_pose[i]);
    }
    std::vector<int32_t>& vid_of_slot = h->h_vid_of_slot;
    vid_of_slot.resize(nv);
    for (int v = 0; v < nv; ++v) vid_of_slot[v] = v;
    std::stable_sort(vid_of_slot.begin(), vid_of_slot.end(), [&](int32_t a, int32_t b) {
        return first_pose[point_of_vid[a]] < first_pose[point_of_vid[b]]; });
    std::vector<int32_t> lm_point(nv), point_slot(L, -1);
    for (int s2 = 0; s2 < nv; ++s2) { lm_point[s2] = point_of_vid[vid_of_slot[s2]]; point_slot[lm_point[s2]] = s2; }
    {
        std::vector<char> seen(nr, 0);
        for (int i = 0; i < P; ++i) if (d->pose_rid[i] >= 0) {
            if (seen[d->pose_rid[i]]) return fail("pose_rid has duplicates");
            seen[d->pose_rid[i]] = 1;
        }
        for (int i = 0; i < nr; ++i) if (!seen[i]) return fail("pose_rid is not a dense 0..nr-1 numbering");
    }

    if (h->upload(&h->point_vid, point_slot)) return -1;     // device-side 'vid' = internal slot

    lap("parameter tables");
    // ---- observation groups
    std::vector<ObsGroup> og(std::max(1, d->num_obs_groups));
    for (int gi = 0; gi < d->num_obs_groups; ++gi) {
        const double* row = d->obs_groups + 4 * gi;
        const int cam = (int)row[0], st = (int)row[1];
        if (cam < 0 || cam >= d->num_cams || st < 0 || st >= d->num_stiff3) return fail("obs group index out of range");
        const double* c = d->cams + 5 * cam;
        ObsGroup& o = og[gi];
        o.cu = c[0]; o.cv = c[1]; o.fu = c[2]; o.fv = c[3];
        o.cam_type = c[4] < 0.0 ? 1 : 0;            // cams row: baseline b >= 0 = stereo, b = -1 = RGB-D
        o.b = o.cam_type ? 0.0 : c[4];
        for (int k = 0; k < 9; ++k) o.S[k] = d->stiff3[9 * st + k];
        o.loss_id = (int)row[2]; o.loss_k = row[3];
    }
    if (h->upload(&h->ogroups, og)) return -1;

    lap("observation groups");
    // ---- observations sorted by landmark: variable points (by vid) first, then constant points
    std::vector<int64_t> order(N);
    for (long i = 0; i < N; ++i) order[i] = i;
    auto lm_key = [&](long i) -> int64_t {
        const int v = point_slot[d->obs_point[i]];
        return v >= 0 ? (int64_t)v : (int64_t)nv + d->obs_point[i];
    };
    counting_sort(order, (size_t)nv + (size_t)L + 1, [&](int64_t a) { return lm_key(a); });
    std::vector<LObs> lobs(N);
    std::vector<int32_t> lorig(N), lm_ptr(nv + 1, 0);
    long Nl = 0;
    for (long k = 0; k < N; ++k) {
        const long i = order[k];
        const int pose = d->obs_pose[i], pt = d->obs_point[i], grp = d->obs_grp[i];
        if (pose < 0 || pose >= P || pt < 0 || pt >= L || grp < 0 || grp >= d->num_obs_groups)
            return fail("observation index out of range");
        LObs& o = lobs[k];
        o.u = d->obs_uvd[3 * i]; o.v = d->obs_uvd[3 * i + 1]; o.d = d->obs_uvd[3 * i + 2];
        o.pose_grp = (int32_t)((uint32_t)pose | ((uint32_t)grp << 24));
        o.point = pt;
        lorig[k] = (int32_t)i;
        const int v = point_slot[pt];
        if (v >= 0) { lm_ptr[v + 1] += 1; ++Nl; }
    }
    for (int v = 0; v < nv; ++v) lm_ptr[v + 1] += lm_ptr[v];
    h->Nl = Nl;
    if (h->upload(&h->lobs, lobs) || h->upload(&h->lorig, lorig) || h->upload(&h->lm_ptr, lm_ptr) ||
        h->upload(&h->lm_point, lm_point)) return -1;
    if (h->alloc(&h->Z, (size_t)Nl * 18) || h->alloc(&h->Cinv, (size_t)nv * 6) ||
        h->alloc(&h->cvec, (size_t)nv * 3) || h->alloc(&h->dxl, (size_t)nv * 3)) return -1;
    HIP_OK(hipMemsetAsync(h->dxl, 0, std::max<size_t>(1, (size_t)nv * 3) * sizeof(double), h->stream));

    lap("landmark sort + lobs");
    // ---- pose segments (observations on variable poses), chunks of 256
    std::vector<int32_t> pcount(nr + 1, 0);
    for (long k = 0; k < N; ++k) { const int r = d->pose_rid[PS_POSE_OF(lobs[k])]; if (r >= 0) pcount[r + 1]++; }
    for (int r = 0; r < nr; ++r) pcount[r + 1] += pcount[r];
    const long Np = h->Np = pcount[nr];
    for (int r = 0; r < nr; ++r) h->max_pose_obs = std::max(h->max_pose_obs, pcount[r + 1] - pcount[r]);
    std::vector<int32_t> pidx(Np), fill(pcount.begin(), pcount.end() - 1);
    for (long k = 0; k < N; ++k) { const int r = d->pose_rid[PS_POSE_OF(lobs[k])]; if (r >= 0) pidx[fill[r]++] = (int32_t)k; }
    std::vector<PItem> pitems;
    std::vector<int32_t> pitem_ptr(nr + 1, 0);
    std::vector<int32_t> pose_of_rid(std::max(nr, 1), 0);
    for (int p = 0; p < P; ++p) if (d->pose_rid[p] >= 0) pose_of_rid[d->pose_rid[p]] = p;
    if (h->upload(&h->pose_of_rid, pose_of_rid)) return -1;
    // chunk of observations per workgroup: 1024 (four per thread) once that still fills the chip
    const int pchunk = Np >= 1024L * 512 ? 1024 : 256;
    for (int r = 0; r < nr; ++r) {
        for (int s = pcount[r]; s < pcount[r + 1]; s += pchunk)
            pitems.push_back({r, s, std::min(s + pchunk, pcount[r + 1]), pose_of_rid[r]});
        pitem_ptr[r + 1] = (int32_t)pitems.size();
    }
    // pose-sorted copy of the observation records; the pose bits (uniform per chunk) carry the landmark slot + 1
    if (nv >= (1 << 24) - 1) return fail("too many variable landmarks for the 24-bit slot field");
    std::vector<LObs> pobs((size_t)Np);
    for (long k = 0; k < Np; ++k) {
        pobs[k] = lobs[pidx[k]];
        const int slot = point_slot[pobs[k].point];              // -1: constant point
        pobs[k].pose_grp = (int32_t)(((uint32_t)PS_GRP_OF(pobs[k]) << 24) | (uint32_t)(slot + 1));
    }
    h->npitems = (int)pitems.size();
    if (h->upload(&h->pitems, pitems) || h->upload(&h->pitem_ptr, pitem_ptr) ||
        h->upload(&h->pobs, pobs) || h->alloc(&h->ppartial, (size_t)pitems.size() * PS_NPOSE_ACC)) return -1;

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
    {
        const double zbytes = 144.0 * (double)lm_ptr[nv];
        double tile_kb = 9216.0, min_mb = 16.0;
        if (const char* e = getenv("PS_SCHUR_TILE_KB")) tile_kb = atof(e);
        if (const char* e = getenv("PS_SCHUR_TILE_MIN_MB")) min_mb = atof(e);
        if (tile_kb > 0 && zbytes > min_mb * 1048576.0)
            ntiles = 8 * (int)std::ceil(zbytes / (8.0 * tile_kb * 1024.0));
    }
    std::vector<long> lm_pairs_before(nv + 1, 0);
    for (int v = 0; v < nv; ++v) {
        long nvar = 0;
        for (int a = lm_ptr[v]; a < lm_ptr[v + 1]; ++a) nvar += d->pose_rid[PS_POSE_OF(lobs[a])] >= 0;
        lm_pairs_before[v + 1] = lm_pairs_before[v] + nvar * (nvar - 1) / 2;
    }
    const long total_pairs = lm_pairs_before[nv];
    // one unit of work per tile: the tile's pairs in (block row, block column, landmark) order, written to its
    // slice of prs; tiles are independent, so they are built by a few host threads
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
    h->schur_tiles = ntiles;
    h->npairs = (long)prs.size();
    if (prs.size() >= (1UL << 31)) return fail("too many Schur pairs for 32-bit indexing");

    lap("pair generation + sort");
    // ---- block pattern of the reduced system
    std::vector<uint64_t> keys;                 // upper keys (ri <= rj)
    keys.reserve(prs.size() / 8 + nr + F + d->num_extra_pairs);
    for (int r = 0; r < nr; ++r) keys.push_back(((uint64_t)r << 32) | (uint32_t)r);
    for (size_t k = 0; k < prs.size(); ++k) if (k == 0 || prs[k].key != prs[k - 1].key) keys.push_back(prs[k].key);
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
    // pair list + one work item (task) per (tile, block) that has pairs
    std::vector<int2> pairs(prs.size());
    std::vector<PairItem> pitm;
    std::vector<int32_t> task_tile;
    for (size_t k = 0; k < prs.size(); ++k) {
        pairs[k] = make_int2(prs[k].a, prs[k].b);
        if (k == 0 || prs[k].key != prs[k - 1].key || prs[k].tile != prs[k - 1].tile) {
            const int a = (int)(prs[k].key >> 32), b = (int)(uint32_t)prs[k].key;
            if (!pitm.empty()) pitm.back().end = (int32_t)k;
            pitm.push_back({slot_of(a, b), slot_of(b, a), (int32_t)k, 0});
            if (a == b) h->has_diag_tasks = true;
            task_tile.push_back(prs[k].tile);
        }
    }
    if (!pitm.empty()) pitm.back().end = (int32_t)prs.size();
    h->npair_items = (int)pitm.size();
    if (h->upload(&h->pairs, pairs)) return -1;
    {   // per-XCD work lists.  Untiled: items are sorted by block row, so equal contiguous shares of
        // the PAIRS (not of the items) give each XCD a contiguous range of block rows with balanced
        // work.  Tiled: XCD x takes tiles x, x + 8, ... (tiles hold equal pair counts).
        std::vector<std::vector<int32_t>> lists(8);
        const double total = (double)pairs.size();
        for (size_t k = 0; k < pitm.size(); ++k) {
            const int x = ntiles > 1 ? (task_tile[k] & 7)
                                     : (total > 0 ? std::min(7, (int)(8.0 * pitm[k].start / total)) : 0);
            lists[x].push_back((int32_t)k);
        }
        // longest tasks first (within each tile): the short ones fill the tail of the XCD's schedule
        if (!getenv("PS_SCHUR_NO_LPT"))
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
    prs.clear(); prs.shrink_to_fit();

    lap("pair items + XCD lists");
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
    HIP_OK(hipMemsetAsync(h->x, 0, std::max<size_t>(1, nvec) * sizeof(double), h->stream));
    HIP_OK(hipMemsetAsync(h->p0, 0, std::max<size_t>(1, nvec) * sizeof(double), h->stream));
    HIP_OK(hipMemsetAsync(h->p1, 0, std::max<size_t>(1, nvec) * sizeof(double), h->stream));

    lap("pcg workspace");
    // ---- scalars
    h->ncost_obs = N > 0 ? std::min(2048, cdiv(N, 256)) : 0;
    h->ncost_fac = F > 0 ? std::min(1024, cdiv(F, 256)) : 0;
    if (h->alloc(&h->cost_partials, (size_t)std::max(h->ncost_obs + h->ncost_fac, 512) + 8) ||
        h->alloc(&h->scalars, SC_NWORDS)) return -1;
    HIP_OK(hipMemsetAsync(h->scalars, 0, SC_NWORDS * sizeof(double), h->stream));
    HIP_OK(hipMemsetAsync(h->status, 0, ST_NWORDS * sizeof(int32_t), h->stream));
    h->nsq_l = nv > 0 ? cdiv(nv, 256 / PS_LM_GROUP) : 0;
    h->nsq_p = nr > 0 ? cdiv(P, 256) : 0;
    if (h->alloc(&h->sq_part_l, (size_t)h->nsq_l) || h->alloc(&h->sq_part_p, (size_t)h->nsq_p) ||
        h->alloc(&h->shard_buf, 2)) return -1;
    HIP_OK(hipMemsetAsync(h->shard_buf, 0, 2 * sizeof(double), h->stream));
    HIP_OK(hipHostMalloc((void**)&h->h_scalars, SC_NWORDS * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
    HIP_OK(hipHostMalloc((void**)&h->h_status, ST_NWORDS * sizeof(int32_t), hipHostMallocMapped | hipHostMallocCoherent));
    HIP_OK(hipHostGetDevicePointer((void**)&h->h_scalars_dev, h->h_scalars, 0));
    HIP_OK(hipHostGetDevicePointer((void**)&h->h_status_dev, h->h_status, 0));
    HIP_OK(hipHostMalloc((void**)&h->h_seq, sizeof(long long), hipHostMallocMapped | hipHostMallocCoherent));
    HIP_OK(hipHostMalloc((void**)&h->h_shard, 2 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
    HIP_OK(hipHostGetDevicePointer((void**)&h->h_shard_dev, h->h_shard, 0));
    HIP_OK(hipHostGetDevicePointer((void**)&h->h_seq_dev, h->h_seq, 0));
    *h->h_seq = 0;
    if (h->alloc(&h->arrivals, 2)) return -1;
    HIP_OK(hipMemsetAsync(h->arrivals, 0, 2 * sizeof(int32_t), h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
    lap("scalars + final sync");
    guard.ok = true;
    *out = h;
    return 0;
}

int ps_get_info(ps_problem* h, ps_problem_info* info) {
    if (!h || !info) return fail("null argument");
    info->dof = h->D; info->num_poses = h->P; info->num_reduced = h->nr; info->num_points = h->L;
    info->num_var_points = h->nv; info->num_obs = h->N; info->num_edges = h->F; info->num_priors = 0;
    info->reduced_nnzb = h->nnzb; info->num_pairs = h->npairs; info->reduce_count = h->red_count;
    info->device_bytes = (int64_t)h->dev_bytes;
    return 0;
}

int ps_eval_cost(ps_problem* h, int include_all_constant, double* cost) {
    if (!h || !cost) return fail("null argument");
    if (cost_pass(h, include_all_constant, SC_COST)) return -1;
    if (read_scalars(h)) return -1;
    *cost = h->h_scalars[SC_COST];
    return 0;
}

int ps_linearize(ps_problem* h, double lambda) {
    if (!h) return fail("null argument");
    return linearize(h, lambda);
}

int ps_reduce_buffer(ps_problem* h, void** dev_ptr, int64_t* count) {
    if (!h || !dev_ptr || !count) return fail("null argument");
    *dev_ptr = h->red; *count = h->red_count;
    return 0;
}

int ps_solve_reduced(ps_problem* h, double tol, int max_iters, int* iters_out, double* relres_out) {
    if (!h) return fail("null argument");
    const int rc = solve_reduced(h, tol, max_iters, iters_out, relres_out);
    if (rc == 0 && h->h_status[ST_LM_FAIL]) return fail("a landmark block H_ll is not positive definite");
    return rc;
}

int ps_backsub(ps_problem* h) {
    if (!h) return fail("null argument");
    return backsub(h);
}

int ps_get_dx(ps_problem* h, double* dx_pose, double* dx_point) {
    if (!h) return fail("null argument");
    std::vector<double> tmp;
    if (dx_pose && h->nr) HIP_OK(hipMemcpyAsync(dx_pose, h->x, (size_t)h->nr * h->D * sizeof(double), hipMemcpyDeviceToHost, h->stream));
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
    return apply_update(h, step);
}

int ps_snapshot_params(ps_problem* h) {
    if (!h) return fail("null argument");
    const size_t n1 = (size_t)h->P * h->PW, n2 = (size_t)h->L * 3;
    if (n1 + n2)
        hipLaunchKernelGGL(k_copy2, dim3((unsigned)std::min<size_t>(2048, cdiv((long)(n1 + n2), 256))), dim3(256), 0, h->stream,
                           n1, (const double*)h->poses, h->poses_snap, n2, (const double*)h->points, h->points_snap);
    return 0;
}

int ps_restore_params(ps_problem* h) {
    if (!h) return fail("null argument");
    const size_t n1 = (size_t)h->P * h->PW, n2 = (size_t)h->L * 3;
    if (n1 + n2)
        hipLaunchKernelGGL(k_copy2, dim3((unsigned)std::min<size_t>(2048, cdiv((long)(n1 + n2), 256))), dim3(256), 0, h->stream,
                           n1, (const double*)h->poses_snap, h->poses, n2, (const double*)h->points_snap, h->points);
    return 0;
}

int ps_get_params(ps_problem* h, double* poses, double* points) {
    if (!h) return fail("null argument");
    if (poses && h->P) HIP_OK(hipMemcpyAsync(poses, h->poses, (size_t)h->P * h->PW * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (points && h->L) HIP_OK(hipMemcpyAsync(points, h->points, (size_t)h->L * 3 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    return sync(h);
}

int ps_set_params(ps_problem* h, const double* poses, const double* points) {
    if (!h) return fail("null argument");
    if (poses && h->P) HIP_OK(hipMemcpyAsync(h->poses, poses, (size_t)h->P * h->PW * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (points && h->L) HIP_OK(hipMemcpyAsync(h->points, points, (size_t)h->L * 3 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    return sync(h);
}

int ps_gn_finish(ps_problem* h, int linesearch, double* cost_out, double* dx_pose_norm2, double* dx_point_norm2) {
    if (!h) return fail("null argument");
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
    h->shard_out = true;
    struct Reset { ps_problem* h; ~Reset() { h->shard_out = false; } } reset{h};
    if (first) {
        HIP_OK(hipMemsetAsync(h->scalars + SC_DXP2, 0, sizeof(double), h->stream));
        if (!h->coarse_built && build_coarse(h)) return -1;
        if (h->cg_explicit) { if (h->D == 6 ? xcg_setup<6>(h, pcg_max_iters, true) : xcg_setup<3>(h, pcg_max_iters, true)) return -1; }
        else if (h->D == 6 ? cg_fused_setup<6>(h, pcg_max_iters, true) : cg_fused_setup<3>(h, pcg_max_iters, true)) return -1;
    }
    // (the explicit PCG runs iteration k in launch group k: one group less than the fused CG's launches)
    const int limit = pcg_max_iters + (h->cg_explicit ? 1 : 2);
    const int margin = h->cg_explicit ? 2 : h->cg_margin;
    int count = first ? (h->last_pcg_iters > 0 ? h->last_pcg_iters + margin : (h->cg_explicit ? 32 : 16)) : std::max(8, h->cg_launched / 2);
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
    return cg_report(h, pcg_iters_out, pcg_relres_out);
}

int ps_gn_iteration(ps_problem* h, double lambda, double pcg_tol, int pcg_max_iters, int linesearch,
                    double* cost_out, double* dx_norm_out, int* pcg_iters_out, double* pcg_relres_out) {
    if (!h) return fail("null argument");
    if (h->nccl_allreduce && h->nccl_comm) {
        // landmark-sharded iteration, everything on the solver's stream in ONE call:
        // linearize -> RCCL sum of [S | g | cost] -> replicated CG + gated shard-local tail ->
        // RCCL sum of {cost, ||dx_point||^2} -> one synchronisation
        if (h->nr == 0 || h->pcg_variant != 1) return fail("the sharded iteration needs the fused CG and a reduced system");
        enum { NCCL_F64 = 8, NCCL_SUM = 0 };
        StageTimer total(h, PS_ST_TOTAL, 2);
        if (linearize(h, lambda)) return -1;
        if (h->nccl_allreduce(h->red, h->red, (size_t)h->red_count, NCCL_F64, NCCL_SUM, h->nccl_comm, h->stream))
            return fail("ncclAllReduce of the reduced system failed");
        int first = 1, done = 0;
        double sb[2] = {0.0, 0.0}, dxp2 = 0.0;
        for (;;) {
            const int last = ps_gn_solve_finish_enqueue(h, pcg_tol, pcg_max_iters, linesearch, first);
            if (last < 0) return -1;
            if (h->nccl_allreduce(h->shard_buf, h->shard_buf, 2, NCCL_F64, NCCL_SUM, h->nccl_comm, h->stream))
                return fail("ncclAllReduce of the shard scalars failed");
            hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, h->stream, h->status, h->scalars, h->shard_buf,
                               h->h_status_dev, h->h_scalars_dev, h->h_shard_dev, h->h_seq_dev, ++h->seq);
            total.stop();
            if (wait_published(h)) return -1;
            if (h->h_status[ST_LM_FAIL]) return fail("a landmark block H_ll is not positive definite");
            done = h->h_status[ST_PCG_DONE];
            sb[0] = h->h_shard[0]; sb[1] = h->h_shard[1];
            dxp2 = h->h_scalars[SC_DXP2];
            if (cg_report(h, pcg_iters_out, pcg_relres_out)) return -1;
            if (done || last) break;
            first = 0;
        }
        if (cost_out) *cost_out = sb[0];
        if (dx_norm_out) *dx_norm_out = std::sqrt(dxp2 + sb[1]);
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
        hipLaunchKernelGGL(k_motion_only_iteration, dim3(h->nr), dim3(PS_MO_THREADS), 0, h->stream, h->nr, h->pitems, h->pitem_ptr,
                           h->pobs, h->points, h->ogroups, h->poses, lambda, linesearch, h->x, h->mo_partials, h->status,
                           h->scalars, h->arrivals + 1, h->h_status_dev, h->h_scalars_dev, h->h_seq_dev, ++h->seq);
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
        if (linearize(h, lambda)) return -1;
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
    if (h->h_status[ST_LM_FAIL]) return fail("a landmark block H_ll is not positive definite");
    if (cost_out) *cost_out = linesearch ? h->h_scalars[SC_COST] : h->h_scalars[SC_LINCOST];
    // a slot whose reduction did not run (no reduced poses / no variable landmarks) is stale: count it as 0
    if (dx_norm_out) *dx_norm_out = std::sqrt((h->nr > 0 ? h->h_scalars[SC_DXP2] : 0.0) + (h->nv > 0 ? h->h_scalars[SC_DXL2] : 0.0));
    return 0;
}

int ps_covariance_begin(ps_problem* h) {
    if (!h) return fail("null argument");
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
    hipLaunchKernelGGL(k_debug_reproj, dim3(cdiv(h->N, 256)), dim3(256), 0, h->stream, h->N, h->lobs, h->lorig,
                       h->poses, h->points, h->ogroups, dr, djp, djl);
    hipMemcpyAsync(r, dr, (size_t)h->N * 3 * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    hipMemcpyAsync(jpose, djp, (size_t)h->N * 18 * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    hipMemcpyAsync(jpoint, djl, (size_t)h->N * 9 * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    const int rc = sync(h);
    hipFree(dr); hipFree(djp); hipFree(djl);
    return rc;
}

int ps_set_option(ps_problem* h, const char* name, double value) {
    if (!h || !name) return fail("null argument");
    const std::string n(name);
    if (n == "pcg_variant") { if (value != 0 && value != 1) return fail("pcg_variant must be 0 or 1"); h->pcg_variant = (int)value; }
    else if (n == "coarse_groups") {
        if (value < -1 || value > 255) return fail("coarse_groups out of range (-1 auto, 0 off, else number of hat intervals; above 63 only for the explicit two-level PCG)");
        h->coarse_req = (int)value; h->coarse_built = false;
    }
    else if (n == "cg_ablate") h->cg_ablate = (int)value;
    else if (n == "schur_ablate") h->schur_ablate = (int)value;
    else if (n == "lm_ablate") h->lm_ablate = (int)value;
    else if (n == "coarse_lag") h->coarse_lag = value != 0.0;
    else if (n == "cg_lds") h->cg_lds = value != 0.0;
    else if (n == "cg_explicit") { h->explicit_ok = value != 0.0; h->coarse_built = false; }
    else if (n == "big_chol") h->big_chol = value != 0.0;
    else if (n == "fused_motion_only") h->mo_fused = value != 0.0;
    else if (n == "direct_max_unknowns") { if (value < 0 || value > 90) return fail("direct_max_unknowns must be 0..90"); h->direct_max = (int)value; }
    else if (n == "coarse_basis") { h->coarse_basis = value != 0.0; h->lci_next = -1; }
    else if (n == "profile_every") { if (value < 1) return fail("profile_every must be >= 1"); h->prof_every = (int)value; }
    else if (n == "cg_margin") { if (value < 0 || value > 64) return fail("cg_margin out of range"); h->cg_margin = (int)value; }
    else if (n == "cg_split_min_rows") { h->cg_split_min_rows = (int)value; h->coarse_built = false; }
    else if (n == "cg_explicit_min_rows") { h->cg_explicit_min_rows = (int)value; h->coarse_built = false; }
    else if (n == "pcg_chunk") { if (value < 1 || value > 4096) return fail("pcg_chunk out of range"); h->pcg_chunk = (int)value; }
    else return fail("unknown option: " + n);
    return 0;
}

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
    double *dJ = nullptr, *dr = nullptr, *dH = nullptr, *dB = nullptr;
    int32_t* dst = nullptr;
    const int nrhs = covariance ? n + 1 : 1;
    HIP_OK(hipMalloc((void**)&dJ, (size_t)m * n * sizeof(double)));
    HIP_OK(hipMalloc((void**)&dr, (size_t)m * sizeof(double)));
    HIP_OK(hipMalloc((void**)&dH, (size_t)n * n * sizeof(double)));
    HIP_OK(hipMalloc((void**)&dB, (size_t)n * nrhs * sizeof(double)));
    HIP_OK(hipMalloc((void**)&dst, ST_NWORDS * sizeof(int32_t)));
    HIP_OK(hipMemset(dst, 0, ST_NWORDS * sizeof(int32_t)));
    HIP_OK(hipMemcpy(dJ, J, (size_t)m * n * sizeof(double), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dr, r, (size_t)m * sizeof(double), hipMemcpyHostToDevice));
    double* dg = nullptr;
    HIP_OK(hipMalloc((void**)&dg, (size_t)n * sizeof(double)));
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
    hipFree(dJ); hipFree(dr); hipFree(dH); hipFree(dB); hipFree(dst); hipFree(dg);
    if (st[ST_DIAG_FAIL]) return fail("normal matrix is not positive definite");
    return 0;
}

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

}  // extern "C"
