// Graph preparation on the device: edge list -> the int32 CSC training graph of train_lightning.py:334-346, 373.
//
//   dgl.remove_self_loop  : drop edges with src == dst; the survivors keep their order and are renumbered 0..E'-1
//   dgl.add_self_loop     : append (v, v) for v = 0..V-1 with edge ids E'..E'+V-1
//   --undirected          : g.add_edges(dst, src) appends the reverse of EVERY edge (self loops included) after that
//   g.formats(['csc'])    : stable counting sort by destination => inside a column edges are in ascending edge id,
//                           which puts the self loop last ([DGL-recalled])
//
// One-time index plumbing, not arithmetic: a flag scan for the renumbering, a scatter into the expanded edge list, one
// stable LSD radix sort (rocPRIM, only the bits a node id needs) keyed by destination, and a binary search per node for
// the column starts.  No host synchronisation: outputs are sized for the worst case (no self loops removed) and the
// true edge count is left on the device.
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include "common.cuh"
#include "bliss_gnn.h"

namespace {

__global__ void __launch_bounds__(256) k_prep_flags(const int* __restrict__ src, const int* __restrict__ dst, int n_edges,
                                                    int num_nodes, int* __restrict__ flags, int* __restrict__ err) {
  for (int e = blockIdx.x * 256 + threadIdx.x; e < n_edges; e += gridDim.x * 256) {
    const int s = src[e], d = dst[e];
    if ((unsigned)s >= (unsigned)num_nodes || (unsigned)d >= (unsigned)num_nodes) { atomicOr(err, 1); flags[e] = 0; continue; }
    flags[e] = s != d;
  }
}

// expanded edge list (keys = destination, srcv = source, vals = edge id = position in the list)
__global__ void __launch_bounds__(256) k_prep_expand(const int* __restrict__ src, const int* __restrict__ dst, int n_edges,
                                                     int num_nodes, int undirected, int cap, const int* __restrict__ flags,
                                                     const int* __restrict__ newid, int* __restrict__ keys,
                                                     int* __restrict__ srcv, int* __restrict__ vals,
                                                     long long* __restrict__ n_out) {
  const int kept = n_edges > 0 ? newid[n_edges - 1] + flags[n_edges - 1] : 0;
  const int n1 = kept + num_nodes, total = undirected ? 2 * n1 : n1;
  const int tid = blockIdx.x * 256 + threadIdx.x, nth = gridDim.x * 256;
  if (tid == 0) *n_out = total;
  for (int e = tid; e < n_edges; e += nth) {
    if (!flags[e]) continue;
    const int id = newid[e], s = src[e], d = dst[e];
    keys[id] = d; srcv[id] = s;
    if (undirected) { keys[n1 + id] = s; srcv[n1 + id] = d; }
  }
  for (int v = tid; v < num_nodes; v += nth) {
    keys[kept + v] = v; srcv[kept + v] = v;
    if (undirected) { keys[n1 + kept + v] = v; srcv[n1 + kept + v] = v; }
  }
  for (int i = tid; i < cap; i += nth) {
    vals[i] = i;
    if (i >= total) keys[i] = num_nodes;          // padding sorts behind every real edge
  }
}

__global__ void __launch_bounds__(256) k_prep_finish(const int* __restrict__ keys_sorted, const int* __restrict__ eid_sorted,
                                                     const int* __restrict__ srcv, const long long* __restrict__ n_out,
                                                     int num_nodes, long long* __restrict__ indptr, int* __restrict__ indices) {
  const int total = (int)*n_out;
  const int tid = blockIdx.x * 256 + threadIdx.x, nth = gridDim.x * 256;
  for (int p = tid; p < total; p += nth) indices[p] = srcv[eid_sorted[p]];
  for (int v = tid; v <= num_nodes; v += nth) {
    int lo = 0, hi = total;                        // first position whose destination >= v
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (keys_sorted[mid] < v) lo = mid + 1; else hi = mid;
    }
    indptr[v] = lo;
  }
}

inline unsigned bits_for(int n) { unsigned b = 1; while ((1ll << b) <= (long long)n) ++b; return b; }
inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

struct PrepLayout {
  size_t flags, newid, keys_in, keys_out, vals_in, srcv, scan_tmp, sort_tmp, scan_bytes, sort_bytes, total;
};

bool prep_layout(int64_t n_edges, int64_t cap, int num_nodes, PrepLayout& L) {
  int* p = nullptr;
  L.scan_bytes = L.sort_bytes = 0;
  if (n_edges > 0 && rocprim::exclusive_scan(nullptr, L.scan_bytes, p, p, 0, (size_t)n_edges, rocprim::plus<int>()) != hipSuccess) return false;
  if (rocprim::radix_sort_pairs(nullptr, L.sort_bytes, p, p, p, p, (size_t)cap, 0, bits_for(num_nodes)) != hipSuccess) return false;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += up256(bytes); return o; };
  L.flags = take((size_t)n_edges * 4);  L.newid = take((size_t)n_edges * 4);
  L.keys_in = take((size_t)cap * 4);    L.keys_out = take((size_t)cap * 4);
  L.vals_in = take((size_t)cap * 4);    L.srcv = take((size_t)cap * 4);
  L.scan_tmp = take(L.scan_bytes);      L.sort_tmp = take(L.sort_bytes);
  L.total = off;
  return true;
}

}  // namespace

extern "C" {

int64_t bliss_graph_prepare_capacity(int64_t n_edges, int32_t num_nodes, int undirected) {
  const int64_t cap = (n_edges + num_nodes) * (undirected ? 2 : 1);
  return (n_edges < 0 || num_nodes <= 0 || cap >= (1ll << 31)) ? -1 : cap;
}

int64_t bliss_graph_prepare_temp_bytes(int64_t n_edges, int32_t num_nodes, int undirected) {
  const int64_t cap = bliss_graph_prepare_capacity(n_edges, num_nodes, undirected);
  PrepLayout L;
  if (cap < 0 || !prep_layout(n_edges, cap, num_nodes, L)) return -1;
  return (int64_t)L.total;
}

int bliss_graph_prepare(const int32_t* coo_src, const int32_t* coo_dst, int64_t n_edges, int32_t num_nodes, int undirected,
                        int64_t* indptr, int32_t* indices, int32_t* eid, int64_t* n_out, int32_t* err, void* temp,
                        int64_t temp_bytes, void* stream) {
  const int64_t cap = bliss_graph_prepare_capacity(n_edges, num_nodes, undirected);
  PrepLayout L;
  if (cap < 0 || !indptr || !indices || !eid || !n_out || !err || !temp || (n_edges > 0 && (!coo_src || !coo_dst))) return BLISS_EINVAL;
  if (!prep_layout(n_edges, cap, num_nodes, L) || (int64_t)L.total > temp_bytes) return BLISS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  char* t = (char*)temp;
  int *flags = (int*)(t + L.flags), *newid = (int*)(t + L.newid), *keys_in = (int*)(t + L.keys_in), *keys_out = (int*)(t + L.keys_out),
      *vals_in = (int*)(t + L.vals_in), *srcv = (int*)(t + L.srcv);
  auto grid = [](int64_t n) { int64_t g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g)); };
  if (n_edges > 0) {
    k_prep_flags<<<grid(n_edges), 256, 0, st>>>(coo_src, coo_dst, (int)n_edges, num_nodes, flags, err);
    hipError_t e = rocprim::exclusive_scan(t + L.scan_tmp, L.scan_bytes, flags, newid, 0, (size_t)n_edges, rocprim::plus<int>(), st);
    if (e != hipSuccess) return (int)e;
  }
  k_prep_expand<<<grid(cap), 256, 0, st>>>(coo_src, coo_dst, (int)n_edges, num_nodes, undirected, (int)cap, flags, newid, keys_in, srcv,
                                           vals_in, (long long*)n_out);
  hipError_t e = rocprim::radix_sort_pairs(t + L.sort_tmp, L.sort_bytes, keys_in, keys_out, vals_in, eid, (size_t)cap, 0,
                                           bits_for(num_nodes), st);
  if (e != hipSuccess) return (int)e;
  k_prep_finish<<<grid(cap), 256, 0, st>>>(keys_out, eid, srcv, (const long long*)n_out, num_nodes, (long long*)indptr, indices);
  return (int)hipGetLastError();
}

}  // extern "C"
