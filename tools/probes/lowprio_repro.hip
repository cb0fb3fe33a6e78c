// lowprio_repro.hip -- does a PLAIN HIP kernel on a lowest-priority stream give another result when normal-priority work runs
// beside it?  (round-5 verdict item 6: root-cause the side-stream corruption of csrc/ps_host_cg.h or bound it to the stack.)
// Measurement infrastructure, not product code.  Nothing in here hand-counts vmcnt, loads straight into LDS or uses DPP:
// the victim is an LDS-resident right-looking Cholesky + inverse by substitutions, __syncthreads() between dependent steps --
// the shape of k_coarse_chol / k_band_chol -- and the aggressor a streaming copy on an ordinary stream.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/lowprio_repro.hip -o /tmp/lowprio_repro && /tmp/lowprio_repro [launches=3000]
// Prints, for the victim on a lowest-priority and on an ordinary stream, with and without the aggressor: launches whose n x n
// output differs from the idle reference in any bit, and how many of those reported a non-positive pivot.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int N = 96;

__global__ __launch_bounds__(256) void victim(const double* __restrict__ A, double* __restrict__ Linv, int* __restrict__ bad) {
    __shared__ double L[N * N], X[N * N];
    const int t = threadIdx.x;
    for (int i = t; i < N * N; i += 256) { L[i] = A[i]; X[i] = 0.0; }
    __syncthreads();
    for (int j = 0; j < N; ++j) {                       // right-looking Cholesky, one column per step
        const double d = L[j * N + j];
        if (t == 0 && !(d > 0.0)) atomicAdd(bad, 1);
        const double l = sqrt(d);
        __syncthreads();
        for (int i = j + t; i < N; i += 256) L[i * N + j] = (i == j) ? l : L[i * N + j] / l;
        __syncthreads();
        for (int e = t; e < (N - j - 1) * (N - j - 1); e += 256) {
            const int r = j + 1 + e / (N - j - 1), c = j + 1 + e % (N - j - 1);
            if (c <= r) L[r * N + c] -= L[r * N + j] * L[c * N + j];
        }
        __syncthreads();
    }
    for (int c = t; c < N; c += 256)                    // L^-1 by forward substitution, a column per thread
        for (int r = c; r < N; ++r) {
            double v = (r == c) ? 1.0 : 0.0;
            for (int k = c; k < r; ++k) v -= L[r * N + k] * X[k * N + c];
            X[r * N + c] = v / L[r * N + r];
        }
    __syncthreads();
    for (int i = t; i < N * N; i += 256) Linv[i] = X[i];
}
__global__ void aggressor(const double* __restrict__ a, double* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i] * 1.0000001;
}

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 3000;
    std::vector<double> hA(N * N, 0.0), ref(N * N), out(N * N);
    srand(7);
    for (int i = 0; i < N; ++i) for (int j = 0; j <= i; ++j) { const double v = (rand() % 2001 - 1000) * 1e-3; hA[i * N + j] = hA[j * N + i] = (i == j) ? N + 1.0 + v : v; }
    double *dA, *dX, *ga, *gb; int* dbad;
    const size_t gn = 32u << 20;
    OK(hipMalloc(&dA, N * N * 8)); OK(hipMalloc(&dX, N * N * 8)); OK(hipMalloc(&dbad, 4)); OK(hipMalloc(&ga, gn * 8)); OK(hipMalloc(&gb, gn * 8));
    OK(hipMemcpy(dA, hA.data(), N * N * 8, hipMemcpyHostToDevice)); OK(hipMemset(ga, 0, gn * 8));
    int lo = 0, hi = 0; OK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t s_low, s_norm, s_agg;
    OK(hipStreamCreateWithPriority(&s_low, hipStreamNonBlocking, lo)); OK(hipStreamCreateWithFlags(&s_norm, hipStreamNonBlocking));
    OK(hipStreamCreateWithFlags(&s_agg, hipStreamNonBlocking));
    OK(hipMemset(dbad, 0, 4));
    hipLaunchKernelGGL(victim, dim3(1), dim3(256), 0, s_norm, dA, dX, dbad); OK(hipStreamSynchronize(s_norm));
    OK(hipMemcpy(ref.data(), dX, N * N * 8, hipMemcpyDeviceToHost));
    printf("priority range: lowest %d, highest %d; %d launches per case\n", lo, hi, launches);
    for (int with_agg = 0; with_agg < 2; ++with_agg)
        for (int low = 0; low < 2; ++low) {
            hipStream_t vs = low ? s_low : s_norm;
            int diff = 0, pivots = 0;
            for (int k = 0; k < launches; ++k) {
                if (with_agg) for (int q = 0; q < 3; ++q) hipLaunchKernelGGL(aggressor, dim3(4096), dim3(256), 0, s_agg, ga, gb, gn);
                OK(hipMemsetAsync(dbad, 0, 4, vs));
                hipLaunchKernelGGL(victim, dim3(1), dim3(256), 0, vs, dA, dX, dbad);
                OK(hipStreamSynchronize(vs));
                int b = 0; OK(hipMemcpy(&b, dbad, 4, hipMemcpyDeviceToHost)); OK(hipMemcpy(out.data(), dX, N * N * 8, hipMemcpyDeviceToHost));
                if (memcmp(out.data(), ref.data(), N * N * 8)) ++diff;
                if (b) ++pivots;
            }
            OK(hipDeviceSynchronize());
            printf("victim on %-16s stream, aggressor %-3s: %d of %d outputs differ from the idle reference, %d non-positive pivots\n",
                   low ? "LOWEST-priority" : "ordinary", with_agg ? "ON" : "off", diff, launches, pivots);
        }
    return 0;
}
