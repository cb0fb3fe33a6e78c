// chain of N dependent small kernels: stream launches vs one hipGraph launch (measurement only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>
#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_small(double* a, const double* b, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = b[i] * 1.0000001 + 1e-9;
}
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 16, n = argc > 2 ? atoi(argv[2]) : 256 * 512, reps = 200;
    double *a, *b; OK(hipMalloc(&a, n * 8)); OK(hipMalloc(&b, n * 8)); OK(hipMemset(a, 0, n * 8)); OK(hipMemset(b, 0, n * 8));
    hipStream_t st; OK(hipStreamCreate(&st));
    hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
    auto chain = [&]() { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_small, dim3((n + 255) / 256), dim3(256), 0, st, (i & 1) ? a : b, (i & 1) ? b : a, n); };
    for (int w = 0; w < 20; ++w) chain();
    OK(hipStreamSynchronize(st));
    // (a) stream: device time of `reps` chains back to back, and host wall time of one chain + sync
    OK(hipEventRecord(e0, st)); for (int r = 0; r < reps; ++r) chain(); OK(hipEventRecord(e1, st)); OK(hipStreamSynchronize(st));
    float ms; OK(hipEventElapsedTime(&ms, e0, e1));
    printf("stream  : %d kernels/chain, %.2f us per chain (device, back to back), %.2f us per kernel\n", N, ms * 1e3 / reps, ms * 1e3 / reps / N);
    double wall = 0; for (int r = 0; r < reps; ++r) { auto t0 = std::chrono::steady_clock::now(); chain(); OK(hipStreamSynchronize(st)); wall += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); }
    printf("stream  : %.2f us per chain (host wall, launch + sync each chain)\n", wall / reps);
    // (b) graph
    hipGraph_t g; hipGraphExec_t ge;
    OK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal)); chain(); OK(hipStreamEndCapture(st, &g));
    OK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 20; ++w) OK(hipGraphLaunch(ge, st));
    OK(hipStreamSynchronize(st));
    OK(hipEventRecord(e0, st)); for (int r = 0; r < reps; ++r) OK(hipGraphLaunch(ge, st)); OK(hipEventRecord(e1, st)); OK(hipStreamSynchronize(st));
    OK(hipEventElapsedTime(&ms, e0, e1));
    printf("graph   : %d kernels/chain, %.2f us per chain (device, back to back), %.2f us per kernel\n", N, ms * 1e3 / reps, ms * 1e3 / reps / N);
    wall = 0; for (int r = 0; r < reps; ++r) { auto t0 = std::chrono::steady_clock::now(); OK(hipGraphLaunch(ge, st)); OK(hipStreamSynchronize(st)); wall += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); }
    printf("graph   : %.2f us per chain (host wall, launch + sync each chain)\n", wall / reps);
    return 0;
}
