// pyslam_amd / HIP (gfx950): k_schur_pose -- the POSE-STATIONARY form of the Schur pair products (round 4).
// (NOT the default: measured 2-2.6x slower than the gather kernel k_schur_pairs_db; its lists are built and it
//  runs only with PS_SCHUR_MODE=1|2 at create + option "schur_mode" -- kept with its parity test as the record of that experiment)
//
// The gather kernels (k_schur_pairs, k_schur_pairs_db) fetch BOTH rows of every pair: 4.46 M row fetches for 0.5 M distinct
// rows at C3, every row requested 9 times, and the pipelined kernel is bound by what a CU can ingest (its ablation: 38 us with
// every fetch a cache hit against 48).  Here a workgroup owns a segment of ONE pose's rows -- up to PS_PP_SEG Z rows, brought
// into LDS once -- and only the partner rows travel: 0.5 M + 2.23 M fetches.
//
// A first version gathered the partner rows into registers, one chunk ahead: 0.119 ms at C3 against 0.046 for the pipelined
// gather kernel -- a task (segment, partner) has ~58 pairs, i.e. one or two chunks, so nothing overlapped.  This one keeps
// k_schur_pairs_db's pipeline (two chunks of a wave in flight, rows and pair words straight into LDS under a hand-counted
// vmcnt, LDS read through inline assembly) and runs it ACROSS tasks: a wave owns a contiguous run of the segment's tasks,
// whose pair words are contiguous in memory, every task padded to a multiple of 32 words (one product pass: 32 pairs x two
// lanes), so the fetch stream is one long task and only the accumulators know where a task ends (half-wave sums, 36 stores
// into the task's partial block, accumulators cleared).  A chunk is 64 partner rows = 64 pairs = two product passes.
// k_schur_combine sums a block's partials in segment order (fixed order: results are bit-reproducible run to run).
//
// Pair word: local a-row index << 23 | Z row of the partner; a-row index PS_PP_PAD marks a padding word (no product).
#pragma once

#define PS_PP_THREADS 512
#define PS_PP_WAVES (PS_PP_THREADS / 64)
#define PS_PP_PAD 511u
#define PS_PP_BUFSTEP (PS_PP_WAVES * PS_SQ_BUF * 8)          // bytes from a wave's buffer 0 to its buffer 1
#define PS_PP_LDS_BYTES ((PS_PP_SEG * PS_SQ_ROWD + 2 * PS_PP_WAVES * PS_SQ_BUF) * 8 + PS_PP_WAVES * (256 + 64) * 4)

// rows of one chunk -> LDS buffer: as sq_fetch, the Z row is the low 23 bits of the pair word
PS_DEV void pp_fetch(const double* __restrict__ Z, const uint32_t (&ia)[6], uint32_t slot_off, double* buf, const int (&off_of)[6]) {
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
    for (int k = 0; k < 6; ++k)
        __builtin_amdgcn_global_load_lds((ps_gptr_t)(Z + PS_ZROW * (size_t)(zr[k] & 0x7FFFFF) + off_of[k]), (ps_lptr_t)(buf + 128 * k), 16, 0, 0);
}

PS_DEV uint32_t pp_read_word(uint32_t lds_addr) {
    uint32_t w;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(w) : "v"(lds_addr) : "memory");
    return w;
}

struct PoseSegW { int32_t row_start, row_count, wt[PS_PP_WAVES + 1]; };     // rows of the segment; tasks [wt[w], wt[w+1]) belong to wave w

__global__ __launch_bounds__(PS_PP_THREADS) void k_schur_pose(
    int per_xcd, const int32_t* __restrict__ order /* [8][per_xcd] segment indices, -1: none */,
    const PoseSegW* __restrict__ segs, const int32_t* __restrict__ seg_rows, const PairItem* __restrict__ tasks,
    const uint32_t* __restrict__ pairs, const double* __restrict__ Z, double* __restrict__ Spart, int ablate)
{
    extern __shared__ __attribute__((aligned(16))) double pp_lds[];
    double* const arows = pp_lds;                                           // [PS_PP_SEG][12]
    double* const sbuf = pp_lds + PS_PP_SEG * PS_SQ_ROWD;                    // [buffer][wave][64 rows x 12]
    int32_t* const sidx = reinterpret_cast<int32_t*>(sbuf + 2 * PS_PP_WAVES * PS_SQ_BUF);   // per wave: the words of four chunks
    int32_t* const stend = sidx + PS_PP_WAVES * 256;                        // per wave: the ends of its next 64 tasks
    const int sidx_seg = order[(size_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3)];
    if (sidx_seg < 0) return;
    const PoseSegW* sg = segs + sidx_seg;
    const int row_start = sg->row_start, row_count = sg->row_count;
    // ---- the segment's rows: 6 pieces of 16 bytes per row (M | pc; the rid word and the padding are not fetched).  Every thread
    // requests its six row indices, then its six pieces, then stores: two memory latencies for the whole fill (a loop over the
    // pieces paid two per piece: 12 dependent round trips, most of the first version's time)
    if (!(ablate & 4)) {
        constexpr int NP = PS_PP_SEG * 6 / PS_PP_THREADS;
        int zr[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int q = threadIdx.x + PS_PP_THREADS * k;
            zr[k] = q < row_count * 6 ? seg_rows[row_start + q / 6] : -1;
        }
        double2 v[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int q = threadIdx.x + PS_PP_THREADS * k;
            v[k] = make_double2(0.0, 0.0);
            if (zr[k] >= 0) v[k] = *reinterpret_cast<const double2*>(Z + PS_ZROW * (size_t)zr[k] + 2 * (q % 6));
        }
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int q = threadIdx.x + PS_PP_THREADS * k;
            if (zr[k] >= 0) *reinterpret_cast<double2*>(arows + PS_SQ_ROWD * (q / 6) + 2 * (q % 6)) = v[k];
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, p = lane & 31, hf = lane >> 5;
    int t = __builtin_amdgcn_readfirstlane(sg->wt[wv]);
    const int t_last = __builtin_amdgcn_readfirstlane(sg->wt[wv + 1]);
    if (t >= t_last) return;
    const int P0 = __builtin_amdgcn_readfirstlane(tasks[t].start), P1 = __builtin_amdgcn_readfirstlane(tasks[t_last - 1].end);
    // (from here to the end every vector-memory operation of the wave is a direct-to-LDS load or a store: a load with a register
    //  destination would make the compiler drain the whole pipeline -- a task's end, needed after every second pass or so, comes
    //  from LDS, where the ends of the wave's next 64 tasks are brought by one more direct load)
    int32_t* const tend = stend + wv * 64;
    const uint32_t tend_base = (uint32_t)(uintptr_t)(ps_lptr_t)tend;
    int t0 = t;
    sq_fetch_idx(&tasks[min(t0 + lane, t_last - 1)].end, tend);
    int cur_end = 0;
    double* const buf0 = sbuf + wv * PS_SQ_BUF;
    double* const buf1 = buf0 + PS_PP_WAVES * PS_SQ_BUF;
    int32_t* const islot = sidx + wv * 256;
    uint32_t ia[6];
    int off_of[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int q = 64 * k + lane, row = q / 6;
        ia[k] = (uint32_t)(uintptr_t)(ps_lptr_t)(islot + row);
        off_of[k] = 2 * (q - 6 * row);
    }
    const uint32_t a_base = (uint32_t)(uintptr_t)(ps_lptr_t)arows;
    const uint32_t b_base = (uint32_t)(uintptr_t)(ps_lptr_t)(buf0 + PS_SQ_ROWD * p);
    const uint32_t w_base = (uint32_t)(uintptr_t)(ps_lptr_t)(islot + p);
    double acc[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) acc[k] = 0.0;
    const int nwords = P1 - P0, nch = (nwords + 63) >> 6;
    const uint32_t* const wlast = pairs + (P1 - 1);
#define PS_PP_IDX(c) (reinterpret_cast<const int32_t*>(min(pairs + P0 + 64 * (c) + lane, wlast)))
    // A task's partial block leaves as 18 stores from lanes 31 / 63.  Stores count in vmcnt like the loads: issued right after
    // the products they would sit in front of the next chunk's fetch in the queue and the next wait would have to see them
    // COMPLETE (a write round trip per task, i.e. per chunk: the first pipelined version ran at 0.13 ms at C3, slower than
    // without the pipeline).  So the sums of the (at most two) tasks that end inside a chunk are held in registers and
    // stored AFTER the step's fetches -- the youngest operations in the queue -- and the next wait allows for them:
    // vmcnt(7 + 18 per stored task).  One step later they are older than a whole chunk of loads and have long retired.
    double sumA[18], sumB[18];
    int tA = -1, tB = -1;
    // one product pass: pairs [pos, pos + 32) of chunk buffer `boff`, half h2; at a task's end its half-wave sums go to sumA / sumB
    auto pass = [&](int pos, uint32_t boff, int slot, int h2) {
        const uint32_t w = pp_read_word(w_base + 256u * (uint32_t)slot + 128u * (uint32_t)h2);
        if ((w >> 23) != PS_PP_PAD && !(ablate & 1))
            sq_products(a_base + PS_SQ_ROWD * 8u * (w >> 23), b_base + boff + PS_SQ_ROWD * 8u * 32u * (uint32_t)h2, hf, acc);
        if (pos + 32 == cur_end) {                            // wave-uniform: the task is complete
            if (tA < 0) {
                tA = t;
#pragma unroll
                for (int k = 0; k < 18; ++k) { sumA[k] = half_sum_dpp(acc[k]); acc[k] = 0.0; }
            } else {
                tB = t;
#pragma unroll
                for (int k = 0; k < 18; ++k) { sumB[k] = half_sum_dpp(acc[k]); acc[k] = 0.0; }
            }
            ++t;
            if (t < t_last) {
                if (t - t0 == 64) {                           // (a wave with more than 64 tasks: the next 64 ends, behind a full wait)
                    t0 = t;
                    sq_fetch_idx(&tasks[min(t0 + lane, t_last - 1)].end, tend);
                    PS_SQ_WAIT_ALL();
                    __builtin_amdgcn_wave_barrier();
                }
                cur_end = __builtin_amdgcn_readfirstlane((int)pp_read_word(tend_base + 4u * (uint32_t)(t - t0)));
            }
        }
    };
    // the held partial blocks -> global memory; returns how many tasks were stored (18 stores each)
    auto flush = [&]() -> int {
        int n = 0;
        if (ablate & 8) { tA = tB = -1; return 0; }
        if (tA >= 0) {
            if (p == 31) {
#pragma unroll
                for (int k = 0; k < 18; ++k) Spart[(size_t)tA * 36 + 18 * hf + k] = sumA[k];
            }
            tA = -1; ++n;
        }
        if (tB >= 0) {
            if (p == 31) {
#pragma unroll
                for (int k = 0; k < 18; ++k) Spart[(size_t)tB * 36 + 18 * hf + k] = sumB[k];
            }
            tB = -1; ++n;
        }
        return n;
    };
#define PS_PP_WAIT(extra) do { if ((extra) == 0) __builtin_amdgcn_s_waitcnt(0x0F70 | 7); /* vmcnt(7) */          \
                               else if ((extra) == 1) __builtin_amdgcn_s_waitcnt(0x4F70 | 9); /* vmcnt(25) */      \
                               else __builtin_amdgcn_s_waitcnt(0x8F70 | 11); /* vmcnt(43) */ } while (0)
    // Queue as in k_schur_pairs_db: I0 I1 I2 | R(0) I3 R(1) | then per step I(c+4) R(c+2) [+ the step's stores]; "at most 7
    // (+ 18 per stored task) younger operations outstanding" reads "chunk c and the words of c+2 are in".
    sq_fetch_idx(PS_PP_IDX(0), islot + 0);
    sq_fetch_idx(PS_PP_IDX(1), islot + 64);
    sq_fetch_idx(PS_PP_IDX(2), islot + 128);
    PS_SQ_WAIT_ALL();
    __builtin_amdgcn_wave_barrier();
    cur_end = __builtin_amdgcn_readfirstlane((int)pp_read_word(tend_base));
    if (!(ablate & 2)) pp_fetch(Z, ia, 0, buf0, off_of);
    sq_fetch_idx(PS_PP_IDX(3), islot + 192);
    if (nch > 1 && !(ablate & 2)) pp_fetch(Z, ia, 256, buf1, off_of);
    int c = 0, stored = 0;
    for (; c + 2 < nch; ++c) {
        const uint32_t boff = (c & 1) * PS_PP_BUFSTEP;
        PS_PP_WAIT(stored);
        __builtin_amdgcn_wave_barrier();
        pass(P0 + 64 * c, boff, c & 3, 0);
        pass(P0 + 64 * c + 32, boff, c & 3, 1);               // (a chunk before the last one is full: both passes are real)
        __builtin_amdgcn_wave_barrier();
        sq_fetch_idx(PS_PP_IDX(c + 4), islot + 64 * (c & 3));
        if (!(ablate & 2)) pp_fetch(Z, ia, 256 * ((c + 2) & 3), (c & 1) ? buf1 : buf0, off_of);
        stored = flush();
    }
    if (c + 1 < nch) {                                          // last but one: the last chunk is still landing
        PS_PP_WAIT(stored);
        __builtin_amdgcn_wave_barrier();
        pass(P0 + 64 * c, (c & 1) * PS_PP_BUFSTEP, c & 3, 0);
        pass(P0 + 64 * c + 32, (c & 1) * PS_PP_BUFSTEP, c & 3, 1);
        __builtin_amdgcn_wave_barrier();
        flush();
        ++c;
    }
    {
        PS_SQ_WAIT_ALL();
        __builtin_amdgcn_wave_barrier();
        pass(P0 + 64 * c, (c & 1) * PS_PP_BUFSTEP, c & 3, 0);
        if (P0 + 64 * c + 32 < P1) pass(P0 + 64 * c + 32, (c & 1) * PS_PP_BUFSTEP, c & 3, 1);
        flush();
    }
#undef PS_PP_WAIT
#undef PS_PP_IDX
}
