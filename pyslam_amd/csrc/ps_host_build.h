// ps_host_build.h -- ps_problem_create's structure build on the GPU (round 4; counterpart of the per-iteration bookkeeping of
// reference pyslam/problem.py:294-329, which this build does once per structure).
//
// What moves to the device: everything whose size is the number of observations or of Schur pairs -- the pair list of the
// gather kernels (22.5 M pairs = 180 MB at C4, the largest table of a handle: generated, sorted and left where the kernels
// read it, never on the host).  What stays on the host: everything whose size is the number of blocks or tasks (block
// pattern, work items, XCD lists, two-level structures: 10^4 .. 10^5 records).
//
// The lists are BIT-IDENTICAL to the host builder's (ps_abi_problem.h keeps it: PS_CREATE_DEVICE=0, and it is what problems
// below the size threshold use): pairs are generated in the host builder's order (landmark by landmark, row a < row b over
// the rows of variable poses), keyed (tile, block row, block column) and sorted with a STABLE radix sort -- the order the
// host's two stable counting passes per tile produce.  tests/test_gpu_create.py holds the two builds against each other
// table by table (ps_debug_table_checksums).
#pragma once
#include "ps_sort.h"

namespace {

// reduced pose index of every Z row (observation in landmark order), -1 for a constant pose
__global__ __launch_bounds__(256) void k_build_row_rid(long n, const LObs* __restrict__ lobs, const int32_t* __restrict__ pose_rid,
                                                        int32_t* __restrict__ rid_row)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) rid_row[i] = pose_rid[PS_POSE_OF(lobs[i])];
}

// pairs per landmark: nvar (nvar - 1) / 2 over its rows on variable poses; cnt[nv] = 0 (the scan's total lands there)
__global__ __launch_bounds__(256) void k_build_pair_counts(int nv, const int32_t* __restrict__ lm_ptr, const int32_t* __restrict__ rid_row,
                                                            long long* __restrict__ cnt)
{
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v > nv) return;
    long long nvar = 0;
    if (v < nv) for (int a = lm_ptr[v]; a < lm_ptr[v + 1]; ++a) nvar += rid_row[a] >= 0;
    cnt[v] = nvar * (nvar - 1) / 2;
}

// The pairs of landmark v at before[v] .. : (row a, row b), a < b, both on variable poses, in (a, b) order -- the host builder's
// generation order.  Key: (tile, lower reduced pose, higher reduced pose); value: (Z row of the lower pose, Z row of the higher)
// = the int2 the Schur kernels read.  PACKED keys ((tile nr + lo) nr + hi) fit 32 bits for every size that fits a GPU today.
template <typename K, bool PACKED>
__global__ __launch_bounds__(256) void k_build_pairs(int nv, const int32_t* __restrict__ lm_ptr, const int32_t* __restrict__ rid_row,
                                                      const long long* __restrict__ before, int ntiles, int nr, K* __restrict__ keys,
                                                      uint64_t* __restrict__ vals)
{
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= nv) return;
    long long at = before[v];
    const long long total = before[nv];
    int tile = 0;
    if (ntiles > 1 && total > 0) {
        const long long t = (long long)((double)ntiles * (double)at / (double)total);       // (the host's tile_of, same arithmetic)
        tile = (int)(t < (long long)(ntiles - 1) ? t : (long long)(ntiles - 1));
    }
    const int a0 = lm_ptr[v], a1 = lm_ptr[v + 1];
    for (int a = a0; a < a1; ++a) {
        const int ra = rid_row[a];
        if (ra < 0) continue;
        for (int b = a + 1; b < a1; ++b) {
            const int rb = rid_row[b];
            if (rb < 0) continue;
            const bool ord = ra <= rb;
            const uint32_t lo = ord ? ra : rb, hi = ord ? rb : ra, rlo = ord ? a : b, rhi = ord ? b : a;
            keys[at] = PACKED ? (K)(((K)tile * (K)nr + lo) * (K)nr + hi) : (K)(((uint64_t)tile << 48) | ((uint64_t)lo << 24) | hi);
            vals[at] = (uint64_t)rlo | ((uint64_t)rhi << 32);
            ++at;
        }
    }
}

template <typename K>
__global__ __launch_bounds__(256) void k_build_task_flags(long n, const K* __restrict__ keys, uint8_t* __restrict__ flags)
{
    const long k = (long)blockIdx.x * 256 + threadIdx.x;
    if (k < n) flags[k] = (k == 0 || keys[k] != keys[k - 1]) ? 1 : 0;
}

template <typename K>
__global__ __launch_bounds__(256) void k_build_gather_keys(int n, const int32_t* __restrict__ idx, const K* __restrict__ keys,
                                                            uint64_t* __restrict__ out)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k < n) out[k] = (uint64_t)keys[idx[k]];
}

// ---- observation tables (landmark-sorted lobs / lorig / lm_ptr, landmark slots, pose-sorted pobs) -------------------------------
// status word: 1 = an observation's pose / point / group index is out of range
__global__ __launch_bounds__(256) void k_build_first_pose(long n, const int32_t* __restrict__ obs_pose, const int32_t* __restrict__ obs_point,
                                                           const int32_t* __restrict__ obs_grp, int P, int L, int G, int32_t* __restrict__ first_pose,
                                                           int32_t* __restrict__ bad)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int pose = obs_pose[i], pt = obs_point[i], grp = obs_grp[i];
    if (pose < 0 || pose >= P || pt < 0 || pt >= L || grp < 0 || grp >= G) { *bad = 1; return; }
    atomicMin(&first_pose[pt], pose);
}
// sort key of landmark (vid) v: the lowest pose that observes it (P: nobody does); value: v
__global__ __launch_bounds__(256) void k_build_slot_keys(int nv, int P, const int32_t* __restrict__ point_of_vid, const int32_t* __restrict__ first_pose,
                                                          uint32_t* __restrict__ key, uint32_t* __restrict__ val)
{
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= nv) return;
    const int32_t f = first_pose[point_of_vid[v]];
    key[v] = (uint32_t)((f < 0 || f >= P) ? P : f);
    val[v] = (uint32_t)v;
}
// slot s holds landmark vid_of_slot[s]: lm_point[s] = its point, point_slot[that point] = s (point_slot starts at -1)
__global__ __launch_bounds__(256) void k_build_slots(int nv, const uint32_t* __restrict__ vid_of_slot, const int32_t* __restrict__ point_of_vid,
                                                      int32_t* __restrict__ lm_point, int32_t* __restrict__ point_slot)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= nv) return;
    const int32_t pt = point_of_vid[vid_of_slot[s]];
    lm_point[s] = pt;
    point_slot[pt] = s;
}
// landmark key of observation i: its slot, or nv + point for a constant point (behind every variable landmark); counts per slot
__global__ __launch_bounds__(256) void k_build_obs_keys(long n, const int32_t* __restrict__ obs_point, const int32_t* __restrict__ point_slot, int nv,
                                                         uint32_t* __restrict__ key, uint32_t* __restrict__ val, int32_t* __restrict__ lcount)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int pt = obs_point[i], s = point_slot[pt];
    key[i] = (uint32_t)(s >= 0 ? s : nv + pt);
    val[i] = (uint32_t)i;
    if (s >= 0) atomicAdd(&lcount[s], 1);
}
// record k of the landmark order = observation order[k]; its reduced pose (nr: constant) is the key of the pose order
__global__ __launch_bounds__(256) void k_build_lobs(long n, const uint32_t* __restrict__ order, const int32_t* __restrict__ obs_pose,
                                                     const int32_t* __restrict__ obs_point, const int32_t* __restrict__ obs_grp,
                                                     const double* __restrict__ obs_uvd, const int32_t* __restrict__ pose_rid, int nr,
                                                     LObs* __restrict__ lobs, int32_t* __restrict__ lorig, uint32_t* __restrict__ pkey,
                                                     uint32_t* __restrict__ pval, int32_t* __restrict__ pcount)
{
    const long k = (long)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const long i = order[k];
    const int pose = obs_pose[i];
    LObs o;
    o.u = obs_uvd[3 * i]; o.v = obs_uvd[3 * i + 1]; o.d = obs_uvd[3 * i + 2];
    o.pose_grp = (int32_t)((uint32_t)pose | ((uint32_t)obs_grp[i] << 24));
    o.point = obs_point[i];
    lobs[k] = o;
    lorig[k] = (int32_t)i;
    const int r = pose_rid[pose];
    pkey[k] = (uint32_t)(r >= 0 ? r : nr);
    pval[k] = (uint32_t)k;
    if (r >= 0) atomicAdd(&pcount[r], 1);
}
// pose-sorted copy: the pose bits (uniform per chunk) carry the landmark slot + 1
__global__ __launch_bounds__(256) void k_build_pobs(long n, const uint32_t* __restrict__ pidx, const LObs* __restrict__ lobs,
                                                     const int32_t* __restrict__ point_slot, LObs* __restrict__ pobs)
{
    const long k = (long)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    LObs o = lobs[pidx[k]];
    const int slot = point_slot[o.point];
    o.pose_grp = (int32_t)(((uint32_t)PS_GRP_OF(o) << 24) | (uint32_t)(slot + 1));
    pobs[k] = o;
}
__global__ __launch_bounds__(256) void k_build_fill_i32(long n, int32_t* __restrict__ p, int32_t v)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

// scratch of one device build: freed when the build ends
struct DevScratch {
    std::vector<void*> ptrs;
    ~DevScratch() { for (void* p : ptrs) (void)hipFree(p); }
    template <typename T>
    int get(T** out, size_t n) {
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T));
        if (e != hipSuccess) return fail(std::string("hipMalloc (structure build): ") + hipGetErrorString(e));
        ptrs.push_back(p);
        *out = (T*)p;
        return 0;
    }
};

// The observation tables on the device: landmark slots (by first observing pose, ties by vid), observations in landmark order
// (lobs / lorig / lm_ptr), in pose order (pobs), the per-pose counts back on the host.  Every ordering is a STABLE radix sort of
// indices by the key the host builder's counting sorts use, so every table is the host builder's, bit for bit.
struct DevObsBuild {
    DevScratch scratch;
    std::vector<int32_t> pcount;                // nr + 1: observations on variable poses before pose r (pose order)
    long Nl = 0, Np = 0;
    static int bits_for(uint64_t maxval) { int b = 1; while (b < 32 && (maxval >> b)) ++b; return b; }
    template <typename T>
    int input(ps_problem* h, const T* host_or_dev, bool resident, size_t n, const T** out) {
        if (resident) { *out = host_or_dev; return 0; }
        T* p = nullptr;
        if (scratch.get(&p, n)) return -1;
        if (n) HIP_OK(hipMemcpyAsync(p, host_or_dev, n * sizeof(T), hipMemcpyHostToDevice, h->stream));
        *out = p;
        return 0;
    }
    // d: the descriptor with HOST tables (staged if the caller's were resident); src: the caller's own descriptor (its obs_* are
    // device pointers when PS_DESC_DEVICE_TABLES is set: used where they lie)
    int run(ps_problem* h, const ps_problem_desc* d, const ps_problem_desc* src, int nr, int nv, const std::vector<int32_t>& point_of_vid,
            std::vector<int32_t>& vid_of_slot) {
        const long N = h->N;
        const int P = h->P, L = h->L;
        const bool res = (src->flags & PS_DESC_DEVICE_TABLES) != 0;
        const int32_t *obs_pose, *obs_point, *obs_grp, *pov;
        const double* obs_uvd;
        if (input(h, res ? src->obs_pose : d->obs_pose, res, (size_t)N, &obs_pose) || input(h, res ? src->obs_point : d->obs_point, res, (size_t)N, &obs_point) ||
            input(h, res ? src->obs_grp : d->obs_grp, res, (size_t)N, &obs_grp) || input(h, res ? src->obs_uvd : d->obs_uvd, res, 3 * (size_t)N, &obs_uvd) ||
            input(h, point_of_vid.data(), false, (size_t)nv, &pov)) return -1;
        const size_t M = (size_t)std::max<long>(N, nv);
        int32_t *first_pose, *bad, *lcount, *pcount_dev, *pscan;
        uint32_t *k0, *k1, *v0, *v1, *vos;
        const size_t tmp_bytes = ps_sort_tmp_bytes(M + 1);
        char* tmp;
        if (scratch.get(&first_pose, (size_t)L) || scratch.get(&bad, 1) || scratch.get(&lcount, (size_t)nv + 1) || scratch.get(&pcount_dev, (size_t)nr + 1) ||
            scratch.get(&pscan, (size_t)nr + 1) || scratch.get(&k0, M) || scratch.get(&k1, M) || scratch.get(&v0, M) || scratch.get(&v1, M) ||
            scratch.get(&vos, (size_t)nv) || scratch.get(&tmp, tmp_bytes)) return -1;
        if (h->alloc(&h->point_vid, (size_t)L, true) || h->alloc(&h->lm_point, (size_t)nv, true) || h->alloc(&h->lobs, (size_t)N, true) ||
            h->alloc(&h->lorig, (size_t)N, true) || h->alloc(&h->lm_ptr, (size_t)nv + 1, true)) return -1;
        hipStream_t st = h->stream;
        HIP_OK(hipMemsetAsync(bad, 0, sizeof(int32_t), st));
        HIP_OK(hipMemsetAsync(lcount, 0, ((size_t)nv + 1) * sizeof(int32_t), st));
        HIP_OK(hipMemsetAsync(pcount_dev, 0, ((size_t)nr + 1) * sizeof(int32_t), st));
        hipLaunchKernelGGL(k_build_fill_i32, dim3(cdiv((long)L, 256)), dim3(256), 0, st, (long)L, first_pose, INT32_MAX);
        hipLaunchKernelGGL(k_build_fill_i32, dim3(cdiv((long)L, 256)), dim3(256), 0, st, (long)L, h->point_vid, -1);
        hipLaunchKernelGGL(k_build_first_pose, dim3(cdiv(N, 256)), dim3(256), 0, st, N, obs_pose, obs_point, obs_grp, P, L, d->num_obs_groups, first_pose, bad);
        // an index out of range must be known before anything is indexed with it
        int32_t bad_h = 0;
        HIP_OK(hipMemcpyAsync(&bad_h, bad, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        if (bad_h) return fail("observation index out of range");
        // landmark slots
        hipLaunchKernelGGL(k_build_slot_keys, dim3(cdiv(nv, 256)), dim3(256), 0, st, nv, P, pov, first_pose, k0, v0);
        HIP_OK(ps_sort_pairs_k32_v32(tmp, tmp_bytes, k0, k1, v0, vos, (size_t)nv, bits_for((uint64_t)P), st));
        hipLaunchKernelGGL(k_build_slots, dim3(cdiv(nv, 256)), dim3(256), 0, st, nv, vos, pov, h->lm_point, h->point_vid);
        vid_of_slot.resize((size_t)nv);
        HIP_OK(hipMemcpyAsync(vid_of_slot.data(), vos, (size_t)nv * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        // landmark order
        hipLaunchKernelGGL(k_build_obs_keys, dim3(cdiv(N, 256)), dim3(256), 0, st, N, obs_point, h->point_vid, nv, k0, v0, lcount);
        HIP_OK(ps_sort_pairs_k32_v32(tmp, tmp_bytes, k0, k1, v0, v1, (size_t)N, bits_for((uint64_t)nv + (uint64_t)L), st));
        HIP_OK(ps_scan_exclusive_i32(tmp, tmp_bytes, lcount, h->lm_ptr, (size_t)nv + 1, st));
        // records in landmark order + pose order
        hipLaunchKernelGGL(k_build_lobs, dim3(cdiv(N, 256)), dim3(256), 0, st, N, v1, obs_pose, obs_point, obs_grp, obs_uvd, h->pose_rid, nr, h->lobs, h->lorig,
                           k0, v0, pcount_dev);
        HIP_OK(ps_sort_pairs_k32_v32(tmp, tmp_bytes, k0, k1, v0, v1, (size_t)N, bits_for((uint64_t)nr), st));
        HIP_OK(ps_scan_exclusive_i32(tmp, tmp_bytes, pcount_dev, pscan, (size_t)nr + 1, st));
        pcount.assign((size_t)nr + 1, 0);
        int32_t nl = 0;
        HIP_OK(hipMemcpyAsync(pcount.data(), pscan, ((size_t)nr + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIP_OK(hipMemcpyAsync(&nl, h->lm_ptr + nv, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        Nl = nl; Np = pcount[nr];
        if (h->alloc(&h->pobs, (size_t)Np, true)) return -1;
        if (Np) hipLaunchKernelGGL(k_build_pobs, dim3(cdiv(Np, 256)), dim3(256), 0, st, Np, v1, h->lobs, h->point_vid, h->pobs);
        HIP_OK(hipStreamSynchronize(st));           // (the scratch is freed when this object goes)
        return 0;
    }
};

// The gather kernels' pair list on the device.  In: the handle's lobs / lm_ptr / pose_rid tables (on the device: the caller has
// flushed the arena).  prepare(): row rids, pair counts, their scan -> total.  build(ntiles): keys + values, stable sort, values
// into `pairs`, task starts and keys back on the host (task = run of equal (tile, block) keys).
struct DevPairBuild {
    ps_problem* h = nullptr;
    DevScratch scratch;
    int nv = 0, nr = 0;
    long nrows = 0;
    long long total = 0;
    int32_t* rid_row = nullptr;
    long long *cnt = nullptr, *before = nullptr;
    void* tmp = nullptr; size_t tmp_bytes = 0;
    void *keys = nullptr, *keys_sorted = nullptr;
    uint64_t *vals = nullptr, *task_keys_dev = nullptr;
    uint8_t* flags = nullptr;
    int32_t *starts_dev = nullptr, *count_dev = nullptr;
    // result of the last build()
    std::vector<int32_t> task_start, task_tile;
    std::vector<uint64_t> task_key;             // (lower reduced pose << 32) | higher

    int prepare(ps_problem* h_, int nv_, int nr_, long nrows_) {
        h = h_; nv = nv_; nr = nr_; nrows = nrows_;
        if (scratch.get(&rid_row, (size_t)nrows) || scratch.get(&cnt, (size_t)nv + 1) || scratch.get(&before, (size_t)nv + 1)) return -1;
        tmp_bytes = ps_sort_tmp_bytes((size_t)nv + 1);
        void* t0 = nullptr;
        if (scratch.get((char**)&t0, tmp_bytes)) return -1;
        if (nrows) hipLaunchKernelGGL(k_build_row_rid, dim3(cdiv(nrows, 256)), dim3(256), 0, h->stream, nrows, h->lobs, h->pose_rid, rid_row);
        hipLaunchKernelGGL(k_build_pair_counts, dim3(cdiv(nv + 1, 256)), dim3(256), 0, h->stream, nv, h->lm_ptr, rid_row, cnt);
        HIP_OK(ps_scan_exclusive_i64(t0, tmp_bytes, cnt, before, (size_t)nv + 1, h->stream));
        HIP_OK(hipMemcpyAsync(&total, before + nv, sizeof(long long), hipMemcpyDeviceToHost, h->stream));
        HIP_OK(hipStreamSynchronize(h->stream));
        return 0;
    }
    // key layout for `ntiles` tiles
    // (PS_CREATE_KEYS64=1 at create: the 64-bit keys of problems beyond tiles x poses^2 = 2^32 on any problem -- no problem that fits
    //  a test reaches them by size: tests/test_gpu_create.py holds them against the host builder this way)
    static bool packed32(int ntiles, int nr) {
        return (uint64_t)ntiles * (uint64_t)nr * (uint64_t)nr <= 0xFFFFFFFFull && !ps_create_env("PS_CREATE_KEYS64");
    }
    int reserve(uint64_t* pairs_out) {
        (void)pairs_out;
        if (keys) return 0;
        const size_t n = (size_t)total;
        tmp_bytes = ps_sort_tmp_bytes(n);
        // 33 B of transient scratch per pair (+ rocPRIM's temporary): two key arrays (sized for 64-bit keys: build() may be called
        // again with another tile count, i.e. another key width), the values, one flag byte.  The task starts and task keys reuse
        // what is dead once the sort has run -- the unsorted keys and the unsorted values (round-4 ADVICE: they had 12 B per pair
        // of their own).
        if (scratch.get((char**)&tmp, tmp_bytes) || scratch.get((uint64_t**)&keys, n) || scratch.get((uint64_t**)&keys_sorted, n) ||
            scratch.get(&vals, n) || scratch.get(&flags, n) || scratch.get(&count_dev, 1)) return -1;
        starts_dev = reinterpret_cast<int32_t*>(keys);      // (n x 4 B of the n x 8 B)
        task_keys_dev = vals;
        return 0;
    }
    template <typename K, bool PACKED>
    int run(int ntiles, int bits, uint64_t* pairs_out) {
        const long n = (long)total;
        K* k0 = (K*)keys; K* k1 = (K*)keys_sorted;
        hipLaunchKernelGGL((k_build_pairs<K, PACKED>), dim3(cdiv(nv, 256)), dim3(256), 0, h->stream, nv, h->lm_ptr, rid_row, before, ntiles, nr, k0, vals);
        if (sizeof(K) == 4) HIP_OK(ps_sort_pairs_k32_v64(tmp, tmp_bytes, (const uint32_t*)k0, (uint32_t*)k1, vals, pairs_out, (size_t)n, bits, h->stream));
        else HIP_OK(ps_sort_pairs_k64_v64(tmp, tmp_bytes, (const uint64_t*)k0, (uint64_t*)k1, vals, pairs_out, (size_t)n, bits, h->stream));
        hipLaunchKernelGGL(k_build_task_flags<K>, dim3(cdiv(n, 256)), dim3(256), 0, h->stream, n, k1, flags);
        HIP_OK(ps_select_flagged_indices(tmp, tmp_bytes, flags, starts_dev, count_dev, (size_t)n, h->stream));
        int32_t ntask = 0;
        HIP_OK(hipMemcpyAsync(&ntask, count_dev, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
        HIP_OK(hipStreamSynchronize(h->stream));
        hipLaunchKernelGGL(k_build_gather_keys<K>, dim3(cdiv(ntask, 256)), dim3(256), 0, h->stream, (int)ntask, starts_dev, k1, task_keys_dev);
        task_start.resize((size_t)ntask);
        std::vector<uint64_t> raw((size_t)ntask);
        if (ntask) {
            HIP_OK(hipMemcpyAsync(task_start.data(), starts_dev, (size_t)ntask * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
            HIP_OK(hipMemcpyAsync(raw.data(), task_keys_dev, (size_t)ntask * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
        }
        HIP_OK(hipStreamSynchronize(h->stream));
        task_key.resize((size_t)ntask); task_tile.resize((size_t)ntask);
        for (int32_t q = 0; q < ntask; ++q) {
            const uint64_t k = raw[q];
            uint64_t tile, lo, hi;
            if (PACKED) { hi = k % (uint64_t)nr; lo = (k / (uint64_t)nr) % (uint64_t)nr; tile = k / (uint64_t)nr / (uint64_t)nr; }
            else { hi = k & 0xFFFFFFull; lo = (k >> 24) & 0xFFFFFFull; tile = k >> 48; }
            task_key[q] = ((uint64_t)lo << 32) | (uint64_t)hi;
            task_tile[q] = (int32_t)tile;
        }
        return 0;
    }
    // pairs_out: the handle's pair table (total pairs x 8 B)
    int build(int ntiles, uint64_t* pairs_out) {
        if (total == 0) { task_start.clear(); task_key.clear(); task_tile.clear(); return 0; }
        if (reserve(pairs_out)) return -1;
        if (packed32(ntiles, nr)) {
            const uint64_t mx = (uint64_t)ntiles * (uint64_t)nr * (uint64_t)nr - 1;
            int bits = 1;
            while (bits < 32 && (mx >> bits)) ++bits;
            return run<uint32_t, true>(ntiles, bits, pairs_out);
        }
        int tb = 1;
        while ((ntiles - 1) >> tb) ++tb;
        return run<uint64_t, false>(ntiles, 48 + tb, pairs_out);
    }
};
}  // namespace
