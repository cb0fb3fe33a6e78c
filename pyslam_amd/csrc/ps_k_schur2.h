// pyslam_amd / HIP (gfx950): k_schur_pairs_db -- the gather kernel of ps_k_linearize.h with TWO chunks of every wave in
// flight.  (test infrastructure: none; product kernel, selected by the option "schur_pipeline")
//
// k_schur_pairs runs fetch -> wait -> products chunk after chunk: the round trip of a chunk's 64 rows (every chunk has an
// L2 miss in it) and the products never overlap inside a wave, only across the 16 waves of a CU, and the time of the
// kernel follows  a + b / (resident waves)  (PS_SCHUR_LDS_PAD sweep).  Here a wave owns two 6 KB
// buffers: while chunk c is multiplied out of one, chunk c+1 is landing in the other, and the fetch of chunk c+2 is
// issued into the first the moment its products are done -- 24 chunks per CU in flight all the time (3 workgroups x 4
// waves x 2 buffers = 144 KB of the 160 KB LDS) instead of 16 for ~70 % of the time.
//
//   * Rows land as 96 B (M | pc: 6 x 16 B; the rid word and the padding of the 128-byte line are not fetched): the chunk
//     is one stream of 384 16-byte pieces, piece q = 64 k + lane of instruction k belongs to row q / 6 -- six
//     global_load_lds_dwordx4 per chunk with ALL lanes active and no branch around them, so exactly 6 + 1 vector-memory
//     operations are issued per step and `s_waitcnt vmcnt(7)` means "chunk c has landed" whatever chunk c+1 is doing
//     (vmcnt retires in order).  Pairs past the end of the task are clamped to its last pair, never predicated.
//   * The pair indices of chunk c+4 are requested at step c, BEFORE the rows of chunk c+2, so the same wait covers the
//     indices step c+2 shuffles.
//   * The products read LDS through inline assembly: the compiler's wait-count pass makes every LDS load it knows about
//     wait for ALL direct-to-LDS loads in flight (it cannot tell the two buffers apart), which would serialise the
//     pipeline again; reads it does not know about are ordered by hand (lgkmcnt(0) inside the statement, wave barriers
//     around the phase).
#pragma once

#define PS_SQ_ROWD 12                         // doubles per row in LDS
#define PS_SQ_BUF (64 * PS_SQ_ROWD)           // doubles per chunk buffer (64 rows: a_0..a_31, b_0..b_31)
#define PS_SQ_BUFSTEP (4 * PS_SQ_BUF * 8)      // bytes from a wave's buffer 0 to its buffer 1

typedef double ps_d2 __attribute__((ext_vector_type(2)));
#ifndef PS_SQ_ASM_CLOBBER
#define PS_SQ_ASM_CLOBBER "memory"
#endif

// pair indices of one chunk -> LDS: lane l brings the Z row of LDS row l (a_p for l < 32, b_p above), 4 B per lane
PS_DEV void sq_fetch_idx(const int32_t* __restrict__ src /* this lane's word */, int32_t* slot /* wave-uniform, 64 words */) {
    __builtin_amdgcn_global_load_lds((ps_gptr_t)src, (ps_lptr_t)slot, 4, 0, 0);
}

// rows of one chunk -> LDS buffer; the indices are read back from the LDS slot they landed in (the caller's wait covers it).
// ia[k] = LDS byte address of the index word of the row this lane's k-th piece belongs to, in slot 0; slot_off = 256 * slot
template <bool HOT /* timing only: every row from the first 64 rows of Z (cache hits) */>
PS_DEV void sq_fetch(const double* __restrict__ Z, const uint32_t (&ia)[6], uint32_t slot_off, double* buf, const int (&off_of)[6]) {
    int zr[6];
    asm volatile(
        "ds_read_b32 %0, %6\n\t"
        "ds_read_b32 %1, %7\n\t"
        "ds_read_b32 %2, %8\n\t"
        "ds_read_b32 %3, %9\n\t"
        "ds_read_b32 %4, %10\n\t"
        "ds_read_b32 %5, %11\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(zr[0]), "=&v"(zr[1]), "=&v"(zr[2]), "=&v"(zr[3]), "=&v"(zr[4]), "=&v"(zr[5])
        : "v"(ia[0] + slot_off), "v"(ia[1] + slot_off), "v"(ia[2] + slot_off), "v"(ia[3] + slot_off), "v"(ia[4] + slot_off),
          "v"(ia[5] + slot_off)
        : "memory");
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        if (HOT) zr[k] &= 63;
        __builtin_amdgcn_global_load_lds((ps_gptr_t)(Z + PS_ZROW * (size_t)zr[k] + off_of[k]), (ps_lptr_t)(buf + 128 * k), 16, 0, 0);
    }
}

// products of the chunk whose rows p and 32 + p start at LDS byte addresses aa / ab (rows have landed): this lane's pair,
// half hf of the block
PS_DEV void sq_products(uint32_t aa, uint32_t ab, int hf, double (&acc)[18]) {
    ps_d2 va[6], vb[6];
    asm volatile(
        "ds_read_b128 %0, %12\n\t"
        "ds_read_b128 %1, %12 offset:16\n\t"
        "ds_read_b128 %2, %12 offset:32\n\t"
        "ds_read_b128 %3, %12 offset:48\n\t"
        "ds_read_b128 %4, %12 offset:64\n\t"
        "ds_read_b128 %5, %12 offset:80\n\t"
        "ds_read_b128 %6, %13\n\t"
        "ds_read_b128 %7, %13 offset:16\n\t"
        "ds_read_b128 %8, %13 offset:32\n\t"
        "ds_read_b128 %9, %13 offset:48\n\t"
        "ds_read_b128 %10, %13 offset:64\n\t"
        "ds_read_b128 %11, %13 offset:80\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(va[0]), "=&v"(va[1]), "=&v"(va[2]), "=&v"(va[3]), "=&v"(va[4]), "=&v"(va[5]),
          "=&v"(vb[0]), "=&v"(vb[1]), "=&v"(vb[2]), "=&v"(vb[3]), "=&v"(vb[4]), "=&v"(vb[5])
        : "v"(aa), "v"(ab)
        : PS_SQ_ASM_CLOBBER);
    double A[9];
    {
        double ma[12];
#pragma unroll
        for (int k = 0; k < 6; ++k) { ma[2 * k] = va[k].x; ma[2 * k + 1] = va[k].y; }
        zrow_cross(ma, ma + 9, A);
#pragma unroll
        for (int k = 0; k < 9; ++k) A[k] = hf ? A[k] : ma[k];
    }
    double mb[12];
#pragma unroll
    for (int k = 0; k < 6; ++k) { mb[2 * k] = vb[k].x; mb[2 * k + 1] = vb[k].y; }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
            acc[6 * a + b] += A[3 * a] * mb[3 * b] + A[3 * a + 1] * mb[3 * b + 1] + A[3 * a + 2] * mb[3 * b + 2];
    double Lb[9];
    zrow_cross(mb, mb + 9, Lb);
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
            acc[6 * a + 3 + b] += A[3 * a] * Lb[3 * b] + A[3 * a + 1] * Lb[3 * b + 1] + A[3 * a + 2] * Lb[3 * b + 2];
}

#define PS_SQ_WAIT_CHUNK() __builtin_amdgcn_s_waitcnt(0x0F70 | 7)      /* vmcnt(7), nothing else */
#define PS_SQ_WAIT_ALL() __builtin_amdgcn_s_waitcnt(0x0F70)            /* vmcnt(0) */

template <int ABL /* timing only: 1 = no products, 2 = no row fetch, 4 = every row fetched from the first 64 rows of Z */>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_schur_pairs_db(
    int per_xcd, const PairItem* __restrict__ xitems, const int2* __restrict__ pairs, const double* __restrict__ Z,
    double* __restrict__ S, double* __restrict__ Spart, int first, int last,
    // untiled lists have no combine launch to finalize the pose pass in: workgroups beyond the pair items do it here
    // (fin_nr > 0; never when a task writes a diagonal block) -- one launch less on the critical path
    int fin_first_block, int fin_nr, const int32_t* __restrict__ pitem_ptr, const double* __restrict__ ppartial,
    const int32_t* __restrict__ diag_slot, double lambda, double* __restrict__ g)
{
    constexpr int ablate = ABL;
    __shared__ __attribute__((aligned(16))) double sbuf[2 * 4 * PS_SQ_BUF];      // [buffer][wave][64 rows x 12]
    __shared__ __attribute__((aligned(16))) int32_t sidx[4 * 256];          // per wave: the index words of four chunks
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double* const buf0 = sbuf + wv * PS_SQ_BUF;
    double* const buf1 = buf0 + 4 * PS_SQ_BUF;
    int32_t* const islot = sidx + wv * 256;
    if ((int)blockIdx.x >= fin_first_block) {
        const int rid = ((int)blockIdx.x - fin_first_block) * 4 + wv;
        if (rid < fin_nr) pose_finalize_wave(rid, lane, pitem_ptr, ppartial, diag_slot, lambda, S, g, buf0);
        return;
    }
    const int local = first + (blockIdx.x >> 3) * 4 + wv;
    if (local >= last) return;
    const size_t pos = (size_t)(blockIdx.x & 7) * per_xcd + local;
    PairItem it = xitems[pos];
    // one item per wave: scalar registers (uniform branches in the pipeline below, no exec-mask loop)
    it.slot = __builtin_amdgcn_readfirstlane(it.slot); it.slotT = __builtin_amdgcn_readfirstlane(it.slotT);
    it.start = __builtin_amdgcn_readfirstlane(it.start); it.end = __builtin_amdgcn_readfirstlane(it.end);
    if (it.slot < 0) return;
    const int p = lane & 31, hf = lane >> 5;
    const int32_t* flat = reinterpret_cast<const int32_t*>(pairs) + hf;
    uint32_t ia[6];
    int off_of[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int q = 64 * k + lane, row = q / 6;
        ia[k] = (uint32_t)(uintptr_t)(ps_lptr_t)(islot + row);
        off_of[k] = 2 * (q - 6 * row);
    }
    const uint32_t aa = (uint32_t)(uintptr_t)(ps_lptr_t)(buf0 + PS_SQ_ROWD * p), ab = aa + 32 * PS_SQ_ROWD * 8;
    double acc[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) acc[k] = 0.0;
    const int npairs = it.end - it.start, nch = (npairs + 31) >> 5, last_pair = max(it.end - 1, 0);
    // the word of chunk c this lane brings in (clamped: rows past the end fetch the task's last pair again)
#define PS_SQ_IDX(c) (flat + 2 * (size_t)min(it.start + 32 * (c) + p, last_pair))
    // Every vector-memory operation from here to the end of the pipeline is a direct-to-LDS load issued in program order
    // (no register destination, so the compiler has nothing to wait for and nothing to hoist; a spill in this stretch
    // would break the count: __graft_entry__.build() refuses a k_schur_pairs_db that uses scratch).  Queue: I0 I1 I2 | R(0) I3 R(1) | then per
    // step I(c+4) R(c+2): "at most 7 younger operations outstanding" always reads "chunk c and the indices of c+2 are in".
    sq_fetch_idx(PS_SQ_IDX(0), islot + 0);
    sq_fetch_idx(PS_SQ_IDX(1), islot + 64);
    sq_fetch_idx(PS_SQ_IDX(2), islot + 128);
    PS_SQ_WAIT_ALL();
    __builtin_amdgcn_wave_barrier();
    if (!(ablate & 2)) sq_fetch<(ABL & 4) != 0>(Z, ia, 0, buf0, off_of);
    sq_fetch_idx(PS_SQ_IDX(3), islot + 192);
    if (nch > 1 && !(ablate & 2)) sq_fetch<(ABL & 4) != 0>(Z, ia, 256, buf1, off_of);
    int c = 0;
    // step c: chunk c is multiplied, the indices of chunk c + 4 are requested (into the slot chunk c's came in), chunk c + 2
    // is fetched into the buffer just read.  Buffer and slot are wave-uniform: scalar adds to the lanes' LDS addresses.
    for (; c + 2 < nch; ++c) {
        const uint32_t boff = (c & 1) * PS_SQ_BUFSTEP;
        PS_SQ_WAIT_CHUNK();
        __builtin_amdgcn_wave_barrier();
        if (!(ablate & 1)) sq_products(aa + boff, ab + boff, hf, acc);   // all 32 pairs are real: only a task's last chunk is ragged
        __builtin_amdgcn_wave_barrier();
        sq_fetch_idx(PS_SQ_IDX(c + 4), islot + 64 * (c & 3));
        if (!(ablate & 2)) sq_fetch<(ABL & 4) != 0>(Z, ia, 256 * ((c + 2) & 3), (c & 1) ? buf1 : buf0, off_of);
    }
    if (c + 1 < nch) {                                          // last but one: the last chunk is still landing
        PS_SQ_WAIT_CHUNK();
        __builtin_amdgcn_wave_barrier();
        if (!(ablate & 1)) sq_products(aa + (c & 1) * PS_SQ_BUFSTEP, ab + (c & 1) * PS_SQ_BUFSTEP, hf, acc);
        __builtin_amdgcn_wave_barrier();
        ++c;
    }
    {
        PS_SQ_WAIT_ALL();
        __builtin_amdgcn_wave_barrier();
        if (p < npairs - 32 * c && !(ablate & 1)) sq_products(aa + (c & 1) * PS_SQ_BUFSTEP, ab + (c & 1) * PS_SQ_BUFSTEP, hf, acc);
        __builtin_amdgcn_wave_barrier();
    }
#undef PS_SQ_IDX
    // ---- as k_schur_pairs: 18 half-wave sums, lanes 31 / 63 publish, coalesced mirrored write
    double* sums = buf0;
#pragma unroll
    for (int k = 0; k < 18; ++k) {
        const double t = half_sum_dpp(acc[k]);
        if (p == 31) sums[18 * hf + k] = t;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < 36) {
        const int r = lane / 6, cc = lane % 6;
        const double mine_v = sums[lane];
        if (Spart) {
            Spart[pos * 36 + lane] = mine_v;
        } else if (it.slot == it.slotT) {
            S[(size_t)it.slot * 36 + lane] -= mine_v + sums[cc * 6 + r];
        } else {
            S[(size_t)it.slot * 36 + lane] = -mine_v;
            S[(size_t)it.slotT * 36 + cc * 6 + r] = -mine_v;
        }
    }
}
