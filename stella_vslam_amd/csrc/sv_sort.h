// Hand-written device sorts / scans shared by the BA structure builder (ba_pairs.hip) and the BoW merge-join (bucket_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

// in-place exclusive scan of data[0 .. n), total to data[n] (data 16-byte aligned); `scratch` = sv_scan_scratch_ints(n) ints (0 for small n:
// one workgroup; beyond that tile sums, their scan, and a scan of every tile behind its offset)
size_t sv_scan_scratch_ints(size_t n);
void sv_scan_i32(hipStream_t s, int* data, int n, int* scratch);
// stable least-significant-digit radix sort (6-bit digits) of n (u32 key, u64 value) pairs by the low `bits` key bits, ping-ponging between
// the two buffer sets; `hist` = sv_sort_hist_ints(n) ints of scratch.  The data starts in set `start`; returns the set that holds the result
// (start ^ (passes & 1)).
int sv_sort_passes(int bits);
size_t sv_sort_hist_ints(size_t n);
// n_dev != null: the element count is read on the device (*n_dev <= n; n sizes the launches and the scratch)
int sv_sort_pairs(hipStream_t s, unsigned* const keys[2], unsigned long long* const vals[2], int start, int n, int bits, int* hist, const int* n_dev = nullptr);
// n <= SV_SORT_SMALL_MAX (key, index) pairs sorted by (key, index) in ONE workgroup's LDS (bitonic network on the 64-bit composites)
#define SV_SORT_SMALL_MAX 16384
bool sv_sort_small(hipStream_t s, const unsigned* keys_in, const int* idx_in, int n, unsigned* keys_out, int* idx_out);  // false: not available on this device
