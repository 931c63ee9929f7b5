// Per-block message passing for SAGEConv('mean') with sampled edge weights, gfx950.
//
// Replaces DGL's g-SpMM behind  graph.update_all(fn.u_mul_e('h','_edge_weight','m'), fn.mean('m','neigh'))
// (dglnn.SAGEConv.forward, called from model.py:321-329) and its autograd backward, plus
// th.norm(h, dim=1) of model.py:318-320.
//
// HBM-bound gather kernels: one 64-lane wave owns one output row; a lane owns 4 consecutive bf16
// features (8-byte loads, 512 B per wave-instruction for a 256-wide row); the row's (src, w) pairs
// are fetched 64 at a time, coalesced, and broadcast by lane; 4 neighbour rows are kept in flight.
// fp32 accumulation, one rounding to bf16 at the store.  The backward is a gather too (through the
// by-source transposed index), so it is deterministic and needs no float atomics.
#include "common.cuh"
#include "bliss_gnn.h"
#include "prof.h"

namespace {

#define SP_TPB 256

struct f4 { float x, y, z, w; };

__device__ __forceinline__ f4 load4(const bf16_t* p) {
  uint2 v = *reinterpret_cast<const uint2*>(p);
  f4 r;
  r.x = __uint_as_float(v.x << 16); r.y = __uint_as_float(v.x & 0xffff0000u);
  r.z = __uint_as_float(v.y << 16); r.w = __uint_as_float(v.y & 0xffff0000u);
  return r;
}

template <bool OUT_F32>
__device__ __forceinline__ void store4(void* out, int64_t off, f4 a) {
  if (OUT_F32) {
    *reinterpret_cast<float4*>((float*)out + off) = make_float4(a.x, a.y, a.z, a.w);
  } else {
    uint2 v;
    v.x = (uint32_t)f2bf(a.x) | ((uint32_t)f2bf(a.y) << 16);
    v.y = (uint32_t)f2bf(a.z) | ((uint32_t)f2bf(a.w) << 16);
    *reinterpret_cast<uint2*>((bf16_t*)out + off) = v;
  }
}

// FWD: row = destination i, edge list = CSR row, neighbour = src[e], coefficient = w[e]
// BWD: row = source j, edge list = t_edge[t_indptr[j]..], neighbour = dst[e], coefficient = w[e]/deg(dst[e])
template <bool VEC4, bool OUT_F32, bool BWD>
__global__ void __launch_bounds__(SP_TPB) k_spmm(const int* __restrict__ row_ptr, const int* __restrict__ nbr_or_tedge,
                                                const int* __restrict__ dst, const int* __restrict__ blk_indptr,
                                                const bf16_t* __restrict__ w, const bf16_t* __restrict__ h, int64_t h_stride,
                                                int n_rows, int dim, int mean, void* out, int64_t out_stride) {
  const int lane = lane_id();
  const int row = blockIdx.x * (SP_TPB / 64) + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const int beg = row_ptr[row], end = row_ptr[row + 1];
  float inv = 1.0f;
  if (!BWD && mean) inv = 1.0f / (float)max(end - beg, 1);
  constexpr int W = VEC4 ? 4 : 1;
  for (int col0 = 0; col0 < dim; col0 += 64 * W) {
    const int col = col0 + lane * W;
    const bool act = col < dim;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int base = beg; base < end; base += 64) {
      // coalesced fetch of up to 64 (neighbour, coefficient) pairs
      int my_n = 0;
      float my_c = 0.f;
      if (base + lane < end) {
        int e = base + lane;
        if (BWD) {
          e = nbr_or_tedge[e];
          int d = dst[e];
          my_n = d;
          float c = w ? bf2f(w[e]) : 1.0f;
          if (mean) c = c / (float)max(blk_indptr[d + 1] - blk_indptr[d], 1);
          my_c = c;
        } else {
          my_n = nbr_or_tedge[e];
          my_c = w ? bf2f(w[e]) : 1.0f;
        }
      }
      const int cnt = min(64, end - base);
      int j = 0;
      for (; j + 4 <= cnt; j += 4) {
        int n0 = __shfl(my_n, j), n1 = __shfl(my_n, j + 1), n2 = __shfl(my_n, j + 2), n3 = __shfl(my_n, j + 3);
        float c0 = __shfl(my_c, j), c1 = __shfl(my_c, j + 1), c2 = __shfl(my_c, j + 2), c3 = __shfl(my_c, j + 3);
        if (act) {
          if (VEC4) {
            f4 a0 = load4(h + n0 * h_stride + col), a1 = load4(h + n1 * h_stride + col);
            f4 a2 = load4(h + n2 * h_stride + col), a3 = load4(h + n3 * h_stride + col);
            acc.x += c0 * a0.x; acc.y += c0 * a0.y; acc.z += c0 * a0.z; acc.w += c0 * a0.w;
            acc.x += c1 * a1.x; acc.y += c1 * a1.y; acc.z += c1 * a1.z; acc.w += c1 * a1.w;
            acc.x += c2 * a2.x; acc.y += c2 * a2.y; acc.z += c2 * a2.z; acc.w += c2 * a2.w;
            acc.x += c3 * a3.x; acc.y += c3 * a3.y; acc.z += c3 * a3.z; acc.w += c3 * a3.w;
          } else {
            float a0 = bf2f(h[n0 * h_stride + col]), a1 = bf2f(h[n1 * h_stride + col]);
            float a2 = bf2f(h[n2 * h_stride + col]), a3 = bf2f(h[n3 * h_stride + col]);
            acc.x += c0 * a0; acc.x += c1 * a1; acc.x += c2 * a2; acc.x += c3 * a3;
          }
        }
      }
      for (; j < cnt; ++j) {
        int n0 = __shfl(my_n, j);
        float c0 = __shfl(my_c, j);
        if (act) {
          if (VEC4) {
            f4 a0 = load4(h + n0 * h_stride + col);
            acc.x += c0 * a0.x; acc.y += c0 * a0.y; acc.z += c0 * a0.z; acc.w += c0 * a0.w;
          } else {
            acc.x += c0 * bf2f(h[n0 * h_stride + col]);
          }
        }
      }
    }
    if (act) {
      acc.x *= inv; acc.y *= inv; acc.z *= inv; acc.w *= inv;
      if (VEC4) store4<OUT_F32>(out, row * out_stride + col, acc);
      else if (OUT_F32) ((float*)out)[row * out_stride + col] = acc.x;
      else ((bf16_t*)out)[row * out_stride + col] = f2bf(acc.x);
    }
  }
}

// ||h_j||_2 per row  (model.py:318-320)
__global__ void __launch_bounds__(SP_TPB) k_embed_norm(const bf16_t* __restrict__ h, int n_rows, int dim, int64_t stride,
                                                      bf16_t* __restrict__ out, int vec4) {
  const int lane = lane_id();
  const int row = blockIdx.x * (SP_TPB / 64) + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const bf16_t* p = h + row * stride;
  float s = 0.f;
  if (vec4) {
    for (int c = lane * 4; c < dim; c += 256) {
      f4 a = load4(p + c);
      s += a.x * a.x; s += a.y * a.y; s += a.z * a.z; s += a.w * a.w;
    }
  } else {
    for (int c = lane; c < dim; c += 64) { float a = bf2f(p[c]); s += a * a; }
  }
  for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
  if (lane == 0) out[row] = f2bf(sqrtf(s));
}

template <bool BWD>
int launch_spmm(const int* row_ptr, const int* idx, const int* dst, const int* blk_indptr, const void* w, const void* h,
                int64_t h_stride, int n_rows, int dim, int mean, void* out, int64_t out_stride, int out_fp32, hipStream_t st) {
  if (n_rows <= 0 || dim <= 0) return 0;
  const bool vec4 = (dim % 4 == 0) && (h_stride % 4 == 0) && (out_stride % 4 == 0) &&
                    (((uintptr_t)h) % 8 == 0) && (((uintptr_t)out) % (out_fp32 ? 16 : 8) == 0);
  dim3 grid((n_rows + SP_TPB / 64 - 1) / (SP_TPB / 64)), block(SP_TPB);
#define GO(V, F) PROF_LAUNCH(BWD ? BK_SPMM_BWD : BK_SPMM_FWD, st, k_spmm<V, F, BWD><<<grid, block, 0, st>>>(row_ptr, idx, dst, blk_indptr, (const bf16_t*)w, (const bf16_t*)h, h_stride, n_rows, dim, mean, out, out_stride))
  if (vec4) { if (out_fp32) GO(true, true); else GO(true, false); }
  else      { if (out_fp32) GO(false, true); else GO(false, false); }
#undef GO
  return (int)hipGetLastError();
}

}  // namespace

extern "C" {

int bliss_spmm_fwd(const int32_t* indptr, const int32_t* src, const void* w, const void* h, int64_t h_stride,
                   int32_t n_dst, int32_t dim, int mean, void* out, int64_t out_stride, int out_fp32, void* stream) {
  if (!indptr || !h || !out) return BLISS_EINVAL;   // src may be NULL for an edgeless block
  return launch_spmm<false>(indptr, src, nullptr, nullptr, w, h, h_stride, n_dst, dim, mean, out, out_stride, out_fp32, (hipStream_t)stream);
}

int bliss_spmm_bwd(const int32_t* t_indptr, const int32_t* t_edge, const int32_t* dst, const int32_t* indptr,
                   const void* w, const void* gout, int64_t gout_stride, int32_t n_src, int32_t dim, int mean,
                   void* gh, int64_t gh_stride, int out_fp32, void* stream) {
  if (!t_indptr || !indptr || !gout || !gh) return BLISS_EINVAL;   // t_edge/dst may be NULL for an edgeless block
  return launch_spmm<true>(t_indptr, t_edge, dst, indptr, w, gout, gout_stride, n_src, dim, mean, gh, gh_stride, out_fp32, (hipStream_t)stream);
}

int bliss_embed_norm(const void* h, int32_t n_rows, int32_t dim, int64_t row_stride, void* out, void* stream) {
  if (!h || !out) return BLISS_EINVAL;
  if (n_rows <= 0) return 0;
  const int vec4 = (dim % 4 == 0) && (row_stride % 4 == 0) && (((uintptr_t)h) % 8 == 0);
  hipStream_t st = (hipStream_t)stream;
  PROF_LAUNCH(BK_EMBED_NORM, st, k_embed_norm<<<(n_rows + 3) / 4, SP_TPB, 0, st>>>((const bf16_t*)h, n_rows, dim, row_stride, (bf16_t*)out, vec4));
  return (int)hipGetLastError();
}

}  // extern "C"
