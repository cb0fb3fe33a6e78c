// ps_sort.h -- device-wide primitives of the structure build (ps_problem_create on the GPU): stable LSD radix sorts, an exclusive
// scan and a flag compaction.  Implemented in ps_sort.hip, a translation unit of its own: these are the library primitives of
// rocPRIM (/opt/rocm/include/rocprim), whose templates are kept out of the kernel file's compile.  Everything is enqueued on
// the given stream; `tmp` is device scratch of at least *_tmp_bytes(n) bytes.  Stability is what makes the device-built lists
// bit-identical to the host builder's counting sorts (tests/test_gpu_create.py).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

#define PS_HIDDEN __attribute__((visibility("hidden")))

// scratch bytes for any of the calls below on n items (the maximum over them)
PS_HIDDEN size_t ps_sort_tmp_bytes(size_t n);
// stable sort of (key, value) pairs by the low `bits` bits of the key
PS_HIDDEN hipError_t ps_sort_pairs_k32_v64(void* tmp, size_t tmp_bytes, const uint32_t* kin, uint32_t* kout, const uint64_t* vin,
                                           uint64_t* vout, size_t n, int bits, hipStream_t s);
PS_HIDDEN hipError_t ps_sort_pairs_k64_v64(void* tmp, size_t tmp_bytes, const uint64_t* kin, uint64_t* kout, const uint64_t* vin,
                                           uint64_t* vout, size_t n, int bits, hipStream_t s);
PS_HIDDEN hipError_t ps_sort_pairs_k32_v32(void* tmp, size_t tmp_bytes, const uint32_t* kin, uint32_t* kout, const uint32_t* vin,
                                           uint32_t* vout, size_t n, int bits, hipStream_t s);
// out[i] = sum of in[0..i)
PS_HIDDEN hipError_t ps_scan_exclusive_i64(void* tmp, size_t tmp_bytes, const long long* in, long long* out, size_t n, hipStream_t s);
PS_HIDDEN hipError_t ps_scan_exclusive_i32(void* tmp, size_t tmp_bytes, const int32_t* in, int32_t* out, size_t n, hipStream_t s);
// out[0..*count) = the indices i with flags[i] != 0, ascending; *count is a device word
PS_HIDDEN hipError_t ps_select_flagged_indices(void* tmp, size_t tmp_bytes, const uint8_t* flags, int32_t* out, int32_t* count, size_t n,
                                               hipStream_t s);
