// ps_sort.hip -- see ps_sort.h.  rocPRIM's device-wide radix sort / scan / select behind plain functions (gfx950 only).
#include <cstring>
#include <algorithm>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include "ps_sort.h"

namespace {
template <class K, class V>
size_t sort_bytes(size_t n) {
    size_t b = 0;
    (void)rocprim::radix_sort_pairs(nullptr, b, (const K*)nullptr, (K*)nullptr, (const V*)nullptr, (V*)nullptr, n, 0u, 8u * (unsigned)sizeof(K),
                                    (hipStream_t)0);
    return b;
}
template <class K, class V>
hipError_t sort_pairs(void* tmp, size_t tmp_bytes, const K* kin, K* kout, const V* vin, V* vout, size_t n, int bits, hipStream_t s) {
    if (n == 0) return hipSuccess;
    size_t need = tmp_bytes;
    return rocprim::radix_sort_pairs(tmp, need, kin, kout, vin, vout, n, 0u, (unsigned)std::max(1, bits), s);
}
}  // namespace

size_t ps_sort_tmp_bytes(size_t n) {
    n = std::max<size_t>(n, 1);
    size_t b = std::max({sort_bytes<uint32_t, uint64_t>(n), sort_bytes<uint64_t, uint64_t>(n), sort_bytes<uint32_t, uint32_t>(n)});
    size_t q = 0;
    (void)rocprim::exclusive_scan(nullptr, q, (const long long*)nullptr, (long long*)nullptr, 0LL, n, rocprim::plus<long long>(), (hipStream_t)0);
    b = std::max(b, q);
    q = 0;
    (void)rocprim::select(nullptr, q, rocprim::counting_iterator<int32_t>(0), (const uint8_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, n,
                          (hipStream_t)0);
    b = std::max(b, q);
    return b + 256;
}

hipError_t ps_sort_pairs_k32_v64(void* tmp, size_t tmp_bytes, const uint32_t* kin, uint32_t* kout, const uint64_t* vin, uint64_t* vout,
                                 size_t n, int bits, hipStream_t s) { return sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, n, bits, s); }
hipError_t ps_sort_pairs_k64_v64(void* tmp, size_t tmp_bytes, const uint64_t* kin, uint64_t* kout, const uint64_t* vin, uint64_t* vout,
                                 size_t n, int bits, hipStream_t s) { return sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, n, bits, s); }
hipError_t ps_sort_pairs_k32_v32(void* tmp, size_t tmp_bytes, const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout,
                                 size_t n, int bits, hipStream_t s) { return sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, n, bits, s); }

hipError_t ps_scan_exclusive_i64(void* tmp, size_t tmp_bytes, const long long* in, long long* out, size_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    return rocprim::exclusive_scan(tmp, tmp_bytes, in, out, 0LL, n, rocprim::plus<long long>(), s);
}
hipError_t ps_scan_exclusive_i32(void* tmp, size_t tmp_bytes, const int32_t* in, int32_t* out, size_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    return rocprim::exclusive_scan(tmp, tmp_bytes, in, out, (int32_t)0, n, rocprim::plus<int32_t>(), s);
}
hipError_t ps_select_flagged_indices(void* tmp, size_t tmp_bytes, const uint8_t* flags, int32_t* out, int32_t* count, size_t n, hipStream_t s) {
    if (n == 0) return hipMemsetAsync(count, 0, sizeof(int32_t), s);
    return rocprim::select(tmp, tmp_bytes, rocprim::counting_iterator<int32_t>(0), flags, out, count, n, s);
}
