// Probe for a persistent CG: NWG workgroups of NT threads exchange an n-vector of doubles per iteration through self-tagged
// 8-byte granules {tag32 | half32} (two per double), sc1 stores, relaxed agent-scope loads, double-buffered by iteration parity.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/agp tools/probes/allgather_probe.hip && /tmp/agp [nwg] [nt] [n] [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;

template <bool PLAIN>
__device__ __forceinline__ void put(u64* g, unsigned tag, double v) {
    const u64 b = (u64)__double_as_longlong(v);
    if (PLAIN) {        // stays in this XCD's L2: only readers on the SAME XCD ever see it before the kernel ends
        *(volatile u64*)g = ((u64)tag << 32) | (b & 0xffffffffull);
        *(volatile u64*)(g + 1) = ((u64)tag << 32) | (b >> 32);
        return;
    }
    __hip_atomic_store((gu64*)g, ((u64)tag << 32) | (b & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store((gu64*)(g + 1), ((u64)tag << 32) | (b >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int NT, int NV, bool ONEXCD>
__global__ __launch_bounds__(NT) void k_probe(int n, int iters, unsigned salt, u64* __restrict__ ex /* 2 x n x 2 granules */,
                                              double* __restrict__ out, int* __restrict__ fail, long long* __restrict__ clk, int work,
                                              int nwg_arg, int* __restrict__ ticket)
{
    __shared__ double v[4096];
    __shared__ double red[16];
    __shared__ int bad;
    __shared__ int my;
    const int t = threadIdx.x;
    int wg = blockIdx.x, nwg = gridDim.x;
    if (ONEXCD) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if ((xcc & 15u) != 0u) return;
        if (t == 0) my = atomicAdd(ticket, 1);
        __syncthreads();
        wg = my; nwg = nwg_arg;
        if (wg >= nwg) return;
    }
    const int per = (n + nwg - 1) / nwg, lo = wg * per, hi = min(n, lo + per);
    if (t == 0) bad = 0;
    for (int i = t; i < n; i += NT) v[i] = 1.0 + 1e-3 * i;
    __syncthreads();
    long long t0 = wall_clock64();
    double acc = 0.0;
    for (int k = 0; k < iters; ++k) {
        const unsigned tag = salt * 4096u + (unsigned)(k + 1);
        u64* buf = ex + (size_t)(k & 1) * n * 2;
        // "SpMV": own entries = f(all of v)  (a stand-in with the same LDS traffic shape)
        double mine = 0.0;
        if (lo + t < hi) {
            const int i = lo + t;
            double s = 0.0;
            for (int j = 0; j < work; ++j) s += v[(i * 7 + j * 13) % n] * 1e-3;
            mine = 0.5 * v[i] + s + 1.0 / (k + 2);
            put<ONEXCD>(buf + 2 * (size_t)i, tag, mine);
        }
        // gather every entry
        double g[NV];
        bool ok = false;
        for (unsigned spins = 0; !ok; ++spins) {
            ok = true;
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                const int i = t + q * NT;
                if (i < n) {
                    const u64 a = __hip_atomic_load((gu64*)(buf + 2 * (size_t)i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const u64 b = __hip_atomic_load((gu64*)(buf + 2 * (size_t)i + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = ok && (unsigned)(a >> 32) == tag && (unsigned)(b >> 32) == tag;
                    g[q] = __longlong_as_double((long long)((a & 0xffffffffull) | (b << 32)));
                }
            }
            ok = __all(ok);
            if (!ok && spins > 200000u) { bad = 1; break; }
            if (!ok) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();            // everyone has read the old v for its "SpMV", and gathered
        if (bad) { if (t == 0) atomicAdd(fail, 1); return; }
#pragma unroll
        for (int q = 0; q < NV; ++q) { const int i = t + q * NT; if (i < n) v[i] = g[q]; }
        // a dot product over the whole vector (block reduction), as the CG's gamma / delta
        double s = 0.0;
        for (int q = 0; q < NV; ++q) { const int i = t + q * NT; if (i < n) s += g[q] * g[q]; }
        for (int off = 32; off; off >>= 1) s += __shfl_xor(s, off, 64);
        if ((t & 63) == 0) red[t >> 6] = s;
        __syncthreads();
        double tot = 0.0;
        for (int w = 0; w < NT / 64; ++w) tot += red[w];
        acc += tot * 1e-9;
        __syncthreads();
    }
    long long t1 = wall_clock64();
    if (t == 0) { out[wg] = acc + v[n - 1]; clk[wg] = t1 - t0; }
}

int main(int argc, char** argv) {
    const int nwg = argc > 1 ? atoi(argv[1]) : 32, nt = argc > 2 ? atoi(argv[2]) : 512, n = argc > 3 ? atoi(argv[3]) : 1280;
    const int iters = argc > 4 ? atoi(argv[4]) : 40, work = argc > 5 ? atoi(argv[5]) : 44;
    u64* ex; double* out; int* fail; long long* clk;
    CK(hipMalloc(&ex, (size_t)4 * n * 8)); CK(hipMemset(ex, 0, (size_t)4 * n * 8));
    CK(hipMalloc(&out, nwg * 8)); CK(hipMalloc(&fail, 4)); CK(hipMemset(fail, 0, 4)); CK(hipMalloc(&clk, nwg * 8));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int wc_khz = 0; CK(hipDeviceGetAttribute(&wc_khz, hipDeviceAttributeWallClockRate, 0));
    unsigned salt = 1;
    const bool onexcd = argc > 6 && atoi(argv[6]) != 0;
    int* ticket; CK(hipMalloc(&ticket, 4));
    auto launch = [&](int it) {
        CK(hipMemsetAsync(ticket, 0, 4, st));
        if (onexcd) {
            if (nt == 512) hipLaunchKernelGGL((k_probe<512, 4, true>), dim3(8 * nwg), dim3(512), 0, st, n, it, salt, ex, out, fail, clk, work, nwg, ticket);
            else hipLaunchKernelGGL((k_probe<1024, 2, true>), dim3(8 * nwg), dim3(1024), 0, st, n, it, salt, ex, out, fail, clk, work, nwg, ticket);
        } else {
            if (nt == 512) hipLaunchKernelGGL((k_probe<512, 4, false>), dim3(nwg), dim3(512), 0, st, n, it, salt, ex, out, fail, clk, work, nwg, ticket);
            else if (nt == 256) hipLaunchKernelGGL((k_probe<256, 8, false>), dim3(nwg), dim3(256), 0, st, n, it, salt, ex, out, fail, clk, work, nwg, ticket);
            else hipLaunchKernelGGL((k_probe<1024, 2, false>), dim3(nwg), dim3(1024), 0, st, n, it, salt, ex, out, fail, clk, work, nwg, ticket);
        }
        ++salt;
    };
    for (int w = 0; w < 3; ++w) launch(iters);
    CK(hipStreamSynchronize(st));
    for (int rep = 0; rep < 2; ++rep)
        for (int it : {iters, 2 * iters}) {
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < 10; ++r) launch(it);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<long long> hc(nwg); CK(hipMemcpy(hc.data(), clk, nwg * 8, hipMemcpyDeviceToHost));
            long long mx = 0; for (auto c : hc) mx = c > mx ? c : mx;
            int hf; CK(hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost));
            printf("nwg %d nt %d n %d iters %d: %.2f us per launch, in-kernel loop %.2f us = %.3f us per iteration (fail %d)\n", nwg, nt, n, it,
                   ms * 1e3 / 10, mx * 1e3 / wc_khz, mx * 1e3 / wc_khz / it, hf);
        }
    std::vector<double> ho(nwg); CK(hipMemcpy(ho.data(), out, nwg * 8, hipMemcpyDeviceToHost));
    bool same = true; for (int i = 1; i < nwg; ++i) same = same && ho[i] == ho[0];
    printf("all workgroups agree: %s (%.17g)\n", same ? "yes" : "NO", ho[0]);
    return 0;
}
