// ps_k_stream.h -- streaming Schur kernel: landmark tiles with the block accumulators held in registers.
// Part of ps_kernels.h (included after ps_k_linearize.h; not a stand-alone header).
#pragma once

// ---------------------------------------------------------------------------
// k_schur_pairs GATHERS: a wave owns one reduced-system block and fetches the two Z rows of each of its pairs from
// wherever they lie -- every row crosses L2 -> LDS about nine times (once per pair it takes part in), and the kernel
// is bound by that request stream and its miss latency.  This kernel STREAMS instead: a
// workgroup owns a TILE of consecutive landmarks (landmarks are stored by the first pose that sees them, so a tile's
// observations fall into a narrow window of poses and touch a bounded set of <= PS_ST_CAP reduced-system blocks),
// reads the tile's Z rows ONCE, contiguously, into LDS, and keeps one accumulator set per touched block in registers:
//   * 512 threads; thread t owns blocks t and t + 512 of the tile, 36 accumulators each (a whole 6 x 6 block per lane:
//     both Z rows of a pair are read from LDS once -- with two lanes per block, as in k_schur_pairs, the random-row LDS
//     reads, which conflict ~3-way, bound the kernel);
//   * the tile's rows go through LDS in sub-tiles of <= PS_ST_SUBROWS rows (112 bytes each: M | pc | rid, the stride that
//     spreads random rows over all banks for 16-byte reads); for each sub-tile and each of the two block slots q the
//     host has listed, per lane, the (a, b) row pairs of its q-th block inside the sub-tile ("entries", one 32-bit
//     word per step and lane, in batches of 8 steps, padded to the longest list of the wave);
//   * at the end every lane writes its accumulators as the tile's PARTIAL of that block; k_schur_combine sums the
//     partials of a block over the tiles in tile order (fixed order => deterministic), as for the tiled gather kernel.
// Traffic: Z once (128 B per observation) + 4 B per pair (+ padding) + 288 B per (tile, block) partial, written and
// read once -- instead of ~2 x 128 B per pair through L2.
// ---------------------------------------------------------------------------
#define PS_ST_THREADS 512
#define PS_ST_PAIRS PS_ST_THREADS            // blocks per slot (one lane per block)
#define PS_ST_SLOTS 2                        // block slots per lane
#define PS_ST_CAP (PS_ST_PAIRS * PS_ST_SLOTS)
#define PS_ST_ROWD 14                        // doubles per row in LDS (7 x 16 B of the 128-byte line)
#define PS_ST_SUBROWS 1280                   // rows per sub-tile in LDS (140 KB)
#define PS_ST_NONE 0xFFFFFFFFu

struct StreamTile {                          // one workgroup
    int32_t row0, nsub;                      // first Z row of the tile, number of sub-tiles
    int32_t sub0;                            // first entry of this tile in the sub-tile table
    int32_t part0;                           // first partial block of this tile
    int32_t nblk, pad0, pad1, pad2;
};
struct StreamSub {                           // one sub-tile: rows [row, row + nrows) and, per (slot q, wave w), its entry list
    int32_t row, nrows;
    int32_t off[PS_ST_SLOTS][PS_ST_THREADS / 64];      // first 16-byte unit of the list of (q, w)
    int32_t steps[PS_ST_SLOTS][PS_ST_THREADS / 64];    // its length in batches of 8 steps (128 units = 2 KB per batch)
};

// one (a, b) pair into the 36 accumulators of its block: Z_a Z_b^T with Z = [M; pc^ M]
PS_DEV void stream_pair(const double* __restrict__ rows, unsigned e, double* __restrict__ acc) {
    const int ra = (int)(e & 0xFFFFu), rb = (int)(e >> 16);
    double za[18], zb[18];
    {
        double m[12];
        const double2* p = reinterpret_cast<const double2*>(rows + PS_ST_ROWD * ra);
#pragma unroll
        for (int k = 0; k < 6; ++k) { const double2 v = p[k]; m[2 * k] = v.x; m[2 * k + 1] = v.y; }
        zrow_expand(m, m + 9, za);
    }
    {
        double m[12];
        const double2* p = reinterpret_cast<const double2*>(rows + PS_ST_ROWD * rb);
#pragma unroll
        for (int k = 0; k < 6; ++k) { const double2 v = p[k]; m[2 * k] = v.x; m[2 * k + 1] = v.y; }
        zrow_expand(m, m + 9, zb);
    }
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c)
            acc[6 * r + c] += za[3 * r] * zb[3 * c] + za[3 * r + 1] * zb[3 * c + 1] + za[3 * r + 2] * zb[3 * c + 2];
}

// all steps of one (sub-tile, slot) list for this wave.  The list is stored in BATCHES of 8 steps, lane major
// (batch, lane, 8 words): a lane fetches a batch with two 16-byte loads, one batch ahead of its use -- loaded word
// by word the compiler sinks every load next to its use and each step waits out a memory latency (2 us per step).
PS_DEV void stream_list(const double* __restrict__ rows, const uint4* __restrict__ ent, int nbatch, int lane,
                        double* __restrict__ acc, int ablate) {
    if (nbatch == 0) return;
    uint4 e0 = ent[lane * 2], e1 = ent[lane * 2 + 1];
    for (int b = 0; b < nbatch; ++b) {
        const int bn = min(b + 1, nbatch - 1);
        const uint4 f0 = ent[(bn * 64 + lane) * 2], f1 = ent[(bn * 64 + lane) * 2 + 1];
        if (!(ablate & 1)) {
            if (e0.x != PS_ST_NONE) stream_pair(rows, e0.x, acc);
            if (e0.y != PS_ST_NONE) stream_pair(rows, e0.y, acc);
            if (e0.z != PS_ST_NONE) stream_pair(rows, e0.z, acc);
            if (e0.w != PS_ST_NONE) stream_pair(rows, e0.w, acc);
            if (e1.x != PS_ST_NONE) stream_pair(rows, e1.x, acc);
            if (e1.y != PS_ST_NONE) stream_pair(rows, e1.y, acc);
            if (e1.z != PS_ST_NONE) stream_pair(rows, e1.z, acc);
            if (e1.w != PS_ST_NONE) stream_pair(rows, e1.w, acc);
        }
        e0 = f0; e1 = f1;
    }
}

__global__ __launch_bounds__(PS_ST_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_schur_stream(
    const StreamTile* __restrict__ tiles, const StreamSub* __restrict__ subs, const uint4* __restrict__ entries /* 16-byte units */,
    const double* __restrict__ Z, double* __restrict__ Spart, int ablate)
{
    extern __shared__ __attribute__((aligned(16))) double rows[];     // PS_ST_SUBROWS x 14
    const StreamTile tl = tiles[blockIdx.x];
    const int t = threadIdx.x, w = t >> 6, lane = t & 63;
    double acc0[36], acc1[36];
#pragma unroll
    for (int k = 0; k < 36; ++k) { acc0[k] = 0.0; acc1[k] = 0.0; }
    const int slot = lane / 7, piece = lane - 7 * slot;              // fetch role: 9 rows per wave-instruction (lane 63 idle)
    for (int s = 0; s < tl.nsub; ++s) {
        const StreamSub& sb = subs[tl.sub0 + s];
        const int srow = sb.row, nrows = sb.nrows;
        __syncthreads();                                             // everyone is done with the previous sub-tile's rows
        // ---- rows [srow, srow + nrows): the first 112 bytes of every 128-byte line straight into LDS
        for (int r0 = w * 9; r0 < nrows; r0 += (PS_ST_THREADS / 64) * 9) {
            const int r = r0 + slot;
            if (slot < 9 && r < nrows && !(ablate & 2))
                __builtin_amdgcn_global_load_lds((ps_gptr_t)(Z + PS_ZROW * (size_t)(srow + r) + 2 * piece),
                                                 (ps_lptr_t)(rows + PS_ST_ROWD * r0), 16, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);                          // vmcnt(0)
        __syncthreads();
        stream_list(rows, entries + sb.off[0][w], sb.steps[0][w], lane, acc0, ablate);
        stream_list(rows, entries + sb.off[1][w], sb.steps[1][w], lane, acc1, ablate);
    }
    // ---- partial blocks: thread t holds blocks t and t + 512 of the tile
    double* out = Spart + (size_t)tl.part0 * 36;
    if (t < tl.nblk)
#pragma unroll
        for (int k = 0; k < 36; ++k) out[(size_t)t * 36 + k] = acc0[k];
    if (t + PS_ST_PAIRS < tl.nblk)
#pragma unroll
        for (int k = 0; k < 36; ++k) out[(size_t)(t + PS_ST_PAIRS) * 36 + k] = acc1[k];
}
