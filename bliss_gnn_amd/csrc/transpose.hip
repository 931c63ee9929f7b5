// By-source index of a block (the transposed CSR the SpMM backward gathers through).
//
// Index plumbing, not arithmetic: a stable LSD radix sort of (source id, edge index) pairs -- rocPRIM's
// device radix sort restricted to the bits a block-local source id needs -- followed by one binary
// search per source for its list start.  Stable => every source's edges stay in ascending edge order,
// which fixes the fp32 summation order of the backward and makes gradients bitwise reproducible.
// Works on capacity-padded arrays with the true edge count on the device (padded entries get the key
// n_src_cap and sort behind every real edge).
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include "common.cuh"
#include "bliss_gnn.h"
#include "prof.h"

namespace {

__global__ void __launch_bounds__(256) k_tr_keys(const int* __restrict__ src, const int* __restrict__ nnz_dev, int nnz_host,
                                                 int cap_b, int pad_key, int* __restrict__ keys, int* __restrict__ vals) {
  const int nnz = nnz_dev ? *nnz_dev : nnz_host;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < cap_b; i += gridDim.x * 256) {
    keys[i] = i < nnz ? src[i] : pad_key;
    vals[i] = i;
  }
}

__global__ void __launch_bounds__(256) k_tr_indptr(const int* __restrict__ keys_sorted, const int* __restrict__ nnz_dev,
                                                   int nnz_host, int n_src_cap, int* __restrict__ t_indptr) {
  const int nnz = nnz_dev ? *nnz_dev : nnz_host;
  for (int j = blockIdx.x * 256 + threadIdx.x; j <= n_src_cap; j += gridDim.x * 256) {
    int lo = 0, hi = nnz;                              // first position whose key >= j
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (keys_sorted[mid] < j) lo = mid + 1; else hi = mid;
    }
    t_indptr[j] = lo;
  }
}

inline unsigned bits_for(int n) { unsigned b = 1; while ((1ll << b) <= (long long)n) ++b; return b; }
inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" {

int64_t bliss_block_transpose_temp_bytes(int32_t cap_b, int32_t n_src_cap) {
  if (cap_b <= 0 || n_src_cap <= 0) return 0;
  size_t sort_bytes = 0;
  int* p = nullptr;
  if (rocprim::radix_sort_pairs(nullptr, sort_bytes, p, p, p, p, (size_t)cap_b, 0, bits_for(n_src_cap)) != hipSuccess) return -1;
  return (int64_t)(3 * up256((size_t)cap_b * 4) + up256(sort_bytes));
}

int bliss_block_transpose(const int32_t* src, const int32_t* nnz_dev, int32_t nnz_host, int32_t cap_b, int32_t n_src_cap,
                          int32_t* t_indptr, int32_t* t_edge, void* temp, int64_t temp_bytes, void* stream) {
  if (!t_indptr || n_src_cap <= 0 || cap_b < 0) return BLISS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (cap_b == 0) {                                  // no edges: every list is empty (kernel, not hipMemsetAsync: graph-safe)
    int g0 = (n_src_cap + 256) / 256;
    if (g0 > 2048) g0 = 2048;
    k_tr_indptr<<<g0, 256, 0, st>>>(nullptr, nullptr, 0, n_src_cap, t_indptr);
    return (int)hipGetLastError();
  }
  if (!src || !t_edge || !temp) return BLISS_EINVAL;
  const size_t arr = up256((size_t)cap_b * 4);
  int* keys_in = (int*)temp;
  int* keys_out = (int*)((char*)temp + arr);
  int* vals_in = (int*)((char*)temp + 2 * arr);
  void* sort_tmp = (char*)temp + 3 * arr;
  size_t sort_bytes = (size_t)temp_bytes - 3 * arr;
  if ((int64_t)(3 * arr) > temp_bytes) return BLISS_EINVAL;
  int grid = (cap_b + 255) / 256;
  if (grid > 2048) grid = 2048;
  PROF_LAUNCH(BK_TRANSPOSE, st, {
    k_tr_keys<<<grid, 256, 0, st>>>(src, nnz_dev, nnz_host, cap_b, n_src_cap, keys_in, vals_in);
    hipError_t e = rocprim::radix_sort_pairs(sort_tmp, sort_bytes, keys_in, keys_out, vals_in, t_edge, (size_t)cap_b, 0,
                                             bits_for(n_src_cap), st);
    if (e != hipSuccess) return (int)e;
    int g2 = (n_src_cap + 256) / 256;
    if (g2 > 2048) g2 = 2048;
    k_tr_indptr<<<g2, 256, 0, st>>>(keys_out, nnz_dev, nnz_host, n_src_cap, t_indptr);
  });
  return (int)hipGetLastError();
}

}  // extern "C"
