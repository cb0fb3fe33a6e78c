// FETCH_SIZE calibration (MI355X_MICROARCH.md: "other access widths are uncalibrated"): kernels that read a KNOWN number of bytes
// from a buffer far larger than the Infinity Cache, per-lane access width 4 / 8 / 16 bytes streaming, and 8-byte random gathers
// (one per 128-byte line / one per 64-byte half line).   hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o /tmp/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/fc -o fc -- /tmp/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <typename T> __global__ void k_stream(const T* __restrict__ p, size_t n, double* out) {
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const T v = p[i];
        const float* f = reinterpret_cast<const float*>(&v);
        acc += f[0];
    }
    if (acc == 12345.678) out[0] = acc;
}
// one 8-byte load per `stride` bytes (stride 128: one per cache line, 64: two per line), index scrambled so that neighbouring lanes hit far-apart lines
__global__ void k_gather8(const double* __restrict__ p, size_t nlines, int stride, double* out) {
    double acc = 0.0;
    const size_t per = stride / 8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nlines; i += (size_t)gridDim.x * blockDim.x) {
        const size_t j = (i * 2654435761ull) % nlines;          // a permutation of the lines when nlines is a power of two
        acc += p[j * per];
    }
    if (acc == 12345.678) out[0] = acc;
}
int main() {
    const size_t bytes = 2ull << 30;                            // 2 GiB >> 256 MiB Infinity Cache
    void* buf; double* out;
    hipMalloc(&buf, bytes); hipMalloc(&out, 8); hipMemset(buf, 0, bytes);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_stream<float>, dim3(4096), dim3(256), 0, 0, (const float*)buf, bytes / 4, out);
        hipLaunchKernelGGL(k_stream<float2>, dim3(4096), dim3(256), 0, 0, (const float2*)buf, bytes / 8, out);
        hipLaunchKernelGGL(k_stream<float4>, dim3(4096), dim3(256), 0, 0, (const float4*)buf, bytes / 16, out);
        hipLaunchKernelGGL(k_gather8, dim3(4096), dim3(256), 0, 0, (const double*)buf, bytes / 128, 128, out);
        hipLaunchKernelGGL(k_gather8, dim3(4096), dim3(256), 0, 0, (const double*)buf, bytes / 64, 64, out);
    }
    hipDeviceSynchronize();
    printf("bytes per streaming kernel %zu; gather128: %zu loads of 8 B (one per 128-B line); gather64: %zu loads (one per 64 B)\n", bytes, bytes / 128, bytes / 64);
    return 0;
}
