// Per-block message passing for SAGEConv('mean') with sampled edge weights, gfx950.
//
// Replaces DGL's g-SpMM behind  graph.update_all(fn.u_mul_e('h','_edge_weight','m'), fn.mean('m','neigh'))
// (dglnn.SAGEConv.forward, called from model.py:321-329) and its autograd backward, plus
// th.norm(h, dim=1) of model.py:318-320.
//
// HBM-bound gather kernels: a 64-lane wave owns 64 consecutive edges; a lane owns 4 consecutive bf16
// features (8-byte loads, 512 B per wave-instruction for a 256-wide row); the (neighbour, coefficient)
// pairs are fetched coalesced and broadcast by lane; 4 neighbour rows are kept in flight.  fp32
// accumulation, one rounding to bf16 at the store.  The backward is a gather too (through the by-source
// transposed index), so it is deterministic and needs no float atomics.
#include "common.cuh"
#include "bliss_gnn.h"
#include "prof.h"

extern "C" int bliss_spmm_chunk_edges(int32_t nnz_bound);

namespace {

#define SP_TPB 256

struct f4 { float x, y, z, w; };

// broadcast from a lane whose index is wave-uniform: v_readlane (scalar result, no LDS round trip like __shfl's ds_bpermute)
__device__ __forceinline__ int bcast_i(int v, int j) { return __builtin_amdgcn_readlane(v, j); }
__device__ __forceinline__ float bcast_f(float v, int j) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j)); }

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

// Balanced ("merge-style") traversal: a wave owns EC consecutive entries of the edge list (bliss_spmm_chunk_edges), whatever
// rows they belong to, so a 30 000-edge hub row and a 3-edge row cost the same per wave.  Rows that lie
// inside one chunk are finished and stored by that wave; a row cut by chunk boundaries leaves fp32
// partials (one "head" and one "tail" slot per chunk) that k_spmm_fixup adds IN CHUNK ORDER -- the sum
// order is fixed, so results are bitwise reproducible without float atomics.
// FWD: row = destination (CSR row), neighbour = src[e], coefficient = w[e]            (x 1/deg at the store)
// BWD: row = source (t_edge groups the edges by source), neighbour = dst[e], coefficient = w[e]/deg(dst[e])

template <bool VEC4, bool OUT_F32>
__device__ __forceinline__ void store_row(void* out, int64_t off, f4 a, float scale) {
  a.x *= scale; a.y *= scale; a.z *= scale; a.w *= scale;
  if (VEC4) store4<OUT_F32>(out, off, a);
  else if (OUT_F32) ((float*)out)[off] = a.x;
  else ((bf16_t*)out)[off] = f2bf(a.x);
}

template <bool VEC4, bool OUT_F32, bool BWD>
__global__ void __launch_bounds__(SP_TPB) k_spmm(const int* __restrict__ row_ptr, const int* __restrict__ t_edge,
                                                const int* __restrict__ src, const int* __restrict__ dst,
                                                const int* __restrict__ blk_indptr, const bf16_t* __restrict__ w,
                                                const bf16_t* __restrict__ h, int64_t h_stride, int n_rows,
                                                const int* __restrict__ nnz_dev, int nnz_host, int dim,
                                                int mean, void* out, int64_t out_stride, float* __restrict__ part, int EC) {
  const int nnz = nnz_dev ? min(*nnz_dev, nnz_host) : nnz_host;
  const int lane = lane_id();
  const int chunk = blockIdx.x * (SP_TPB / 64) + (threadIdx.x >> 6);
  const int nchunks = nnz > 0 ? (nnz + EC - 1) / EC : 1;
  if (chunk >= nchunks) return;
  const int c0 = chunk * EC, c1 = min(nnz, c0 + EC), cnt = c1 - c0;
  int my_row = -1, my_n = 0;
  float my_c = 0.f;
  if (lane < cnt) {
    const int e = BWD ? t_edge[c0 + lane] : c0 + lane;
    my_row = BWD ? src[e] : dst[e];
    my_n = BWD ? dst[e] : src[e];
    float c = w ? bf2f(w[e]) : 1.0f;
    if (BWD && mean) { const int d = dst[e]; c = c / (float)max(blk_indptr[d + 1] - blk_indptr[d], 1); }
    my_c = c;
  }
  constexpr int W = VEC4 ? 4 : 1;
  const f4 zero = {0.f, 0.f, 0.f, 0.f};
  for (int col0 = 0; col0 < dim; col0 += 64 * W) {
    const int col = col0 + lane * W;
    const bool act = col < dim;
    // finish row r with the accumulated value: store, or leave a partial for the fix-up pass
    auto flush = [&](int r, f4 acc) {
      const int rb = row_ptr[r], re = row_ptr[r + 1];
      const bool starts = rb >= c0, ends = re <= c1;
      if (!act) return;
      if (starts && ends) {
        const float scale = (!BWD && mean) ? 1.0f / (float)max(re - rb, 1) : 1.0f;
        store_row<VEC4, OUT_F32>(out, r * out_stride + col, acc, scale);
      } else {
        float* q = part + ((int64_t)chunk * 2 + (starts ? 1 : 0)) * dim + col;
        q[0] = acc.x;
        if (VEC4) { q[1] = acc.y; q[2] = acc.z; q[3] = acc.w; }
      }
    };
    if (cnt == 0) continue;                                 // rows without edges are cleared by k_spmm_fixup
    int cur = bcast_i(my_row, 0);
    f4 acc = zero;
    for (int j = 0; j < cnt;) {
      const int r = bcast_i(my_row, j);
      if (r != cur) { flush(cur, acc); acc = zero; cur = r; }
      const int run = __popcll(__ballot(my_row == r && lane >= j));
      const int end = j + run;
      for (; j + 4 <= end; j += 4) {
        int n0 = bcast_i(my_n, j), n1 = bcast_i(my_n, j + 1), n2 = bcast_i(my_n, j + 2), n3 = bcast_i(my_n, j + 3);
        float k0 = bcast_f(my_c, j), k1 = bcast_f(my_c, j + 1), k2 = bcast_f(my_c, j + 2), k3 = bcast_f(my_c, j + 3);
        if (act) {
          if (VEC4) {
            f4 a0 = load4(h + n0 * h_stride + col), a1 = load4(h + n1 * h_stride + col);
            f4 a2 = load4(h + n2 * h_stride + col), a3 = load4(h + n3 * h_stride + col);
            acc.x += k0 * a0.x; acc.y += k0 * a0.y; acc.z += k0 * a0.z; acc.w += k0 * a0.w;
            acc.x += k1 * a1.x; acc.y += k1 * a1.y; acc.z += k1 * a1.z; acc.w += k1 * a1.w;
            acc.x += k2 * a2.x; acc.y += k2 * a2.y; acc.z += k2 * a2.z; acc.w += k2 * a2.w;
            acc.x += k3 * a3.x; acc.y += k3 * a3.y; acc.z += k3 * a3.z; acc.w += k3 * a3.w;
          } else {
            float a0 = bf2f(h[n0 * h_stride + col]), a1 = bf2f(h[n1 * h_stride + col]);
            float a2 = bf2f(h[n2 * h_stride + col]), a3 = bf2f(h[n3 * h_stride + col]);
            acc.x += k0 * a0; acc.x += k1 * a1; acc.x += k2 * a2; acc.x += k3 * a3;
          }
        }
      }
      for (; j < end; ++j) {
        int n0 = bcast_i(my_n, j);
        float k0 = bcast_f(my_c, j);
        if (act) {
          if (VEC4) {
            f4 a0 = load4(h + n0 * h_stride + col);
            acc.x += k0 * a0.x; acc.y += k0 * a0.y; acc.z += k0 * a0.z; acc.w += k0 * a0.w;
          } else acc.x += k0 * bf2f(h[n0 * h_stride + col]);
        }
      }
    }
    flush(cur, acc);
  }
}

// rows cut by chunk boundaries: tail partial of the chunk they start in + head partials of the following chunks;
// rows without edges (isolated or capacity-padded) are cleared here, one wave per row
template <bool OUT_F32, bool BWD, bool VEC4>
__global__ void __launch_bounds__(SP_TPB) k_spmm_fixup(const int* __restrict__ row_ptr, int n_rows, int dim, int mean,
                                                      const float* __restrict__ part, void* out, int64_t out_stride, int EC) {
  const int lane = lane_id();
  const int r = blockIdx.x * (SP_TPB / 64) + (threadIdx.x >> 6);
  if (r >= n_rows) return;
  const int rb = row_ptr[r], re = row_ptr[r + 1];
  constexpr int W = VEC4 ? 4 : 1;
  const f4 zero = {0.f, 0.f, 0.f, 0.f};
  if (re <= rb) {                                           // no edges (incl. capacity-padded rows): the output row is zero
    for (int col = lane * W; col < dim; col += 64 * W) store_row<VEC4, OUT_F32>(out, r * out_stride + col, zero, 1.0f);
    return;
  }
  const int c = rb / EC, c_end = (re - 1) / EC;
  if (c_end == c) return;                                   // finished by its chunk
  const float scale = (!BWD && mean) ? 1.0f / (float)(re - rb) : 1.0f;
  for (int col = lane * W; col < dim; col += 64 * W) {
    f4 sum = zero;
    if (VEC4) {
      const float4 v = *reinterpret_cast<const float4*>(part + ((int64_t)c * 2 + 1) * dim + col);
      sum.x = v.x; sum.y = v.y; sum.z = v.z; sum.w = v.w;
#pragma unroll 4
      for (int cc = c + 1; cc <= c_end; ++cc) {
        const float4 u = *reinterpret_cast<const float4*>(part + ((int64_t)cc * 2) * dim + col);
        sum.x += u.x; sum.y += u.y; sum.z += u.z; sum.w += u.w;
      }
    } else {
      sum.x = part[((int64_t)c * 2 + 1) * dim + col];
      for (int cc = c + 1; cc <= c_end; ++cc) sum.x += part[((int64_t)cc * 2) * dim + col];
    }
    store_row<VEC4, OUT_F32>(out, r * out_stride + col, sum, scale);
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

// ---- feature gather + row norms in one pass ---------------------------------------------------------------------------
// blocks[0].srcdata['features'] = g.ndata['features'][input_nodes] (train_lightning.py:138) and the embed_norm of those rows
// (model.py:318-320): a wave copies one row (W bf16 per lane and access) through its slice of LDS to `out`, then takes the
// sum of squares from LDS in exactly k_embed_norm's order, so the norm has the bits k_embed_norm(out) would give.
// HBM-bound: 2 * K * F bytes in, the same out (SURVEY 8d "feature gather").
template <int W>
__global__ void __launch_bounds__(SP_TPB) k_gather_rows_norm(const bf16_t* __restrict__ feat, int64_t f_stride, const int* __restrict__ ids,
                                                            int n_rows, int dim, bf16_t* __restrict__ out, int64_t o_stride,
                                                            bf16_t* __restrict__ norm_out, int norm_vec4) {
  extern __shared__ __attribute__((aligned(16))) bf16_t rows_sh[];
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  const int row = blockIdx.x * (SP_TPB / 64) + wave;
  if (row >= n_rows) return;                          // wave-uniform; no workgroup barrier below
  const int dpad = (dim + 7) & ~7;
  bf16_t* sh = rows_sh + (size_t)wave * dpad;
  const bf16_t* p = feat + (int64_t)ids[row] * f_stride;
  bf16_t* q = out + (int64_t)row * o_stride;
  if (W == 4) {
    for (int c = lane * 4; c < dim; c += 256) { const uint2 v = *reinterpret_cast<const uint2*>(p + c); *reinterpret_cast<uint2*>(q + c) = v; *reinterpret_cast<uint2*>(sh + c) = v; }
  } else if (W == 2) {
    for (int c = lane * 2; c < dim; c += 128) { const uint32_t v = *reinterpret_cast<const uint32_t*>(p + c); *reinterpret_cast<uint32_t*>(q + c) = v; *reinterpret_cast<uint32_t*>(sh + c) = v; }
  } else {
    for (int c = lane; c < dim; c += 64) { const bf16_t v = p[c]; q[c] = v; sh[c] = v; }
  }
  if (!norm_out) return;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();                    // the wave's own LDS writes, in order, before its reads
  float s = 0.f;
  if (norm_vec4) {
    for (int c = lane * 4; c < dim; c += 256) {
      const float a = bf2f(sh[c]), b = bf2f(sh[c + 1]), cc = bf2f(sh[c + 2]), d = bf2f(sh[c + 3]);
      s += a * a; s += b * b; s += cc * cc; s += d * d;
    }
  } else {
    for (int c = lane; c < dim; c += 64) { const float a = bf2f(sh[c]); s += a * a; }
  }
  for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
  if (lane == 0) norm_out[row] = f2bf(sqrtf(s));
}

// ---- SAGE hidden-layer epilogue in one pass ----------------------------------------------------------------------
// model.py:321-333 + :318-320 of the next layer:   rst = fc_self(h_dst) + h_neigh;  h = dropout(relu(rst));  ||h_j||
// The reference runs four element-wise kernels for this; here one wave walks a row once: bf16 add (fp32, one rounding),
// ReLU, dropout (keep with probability 1-p, scale 1/(1-p), one rounding) and the fp32 sum of squares of what it stores
// (the same accumulation order as k_embed_norm).  The dropout bits come from a counter-based hash of (seed, launch
// counter, element index): the same Bernoulli(1-p) law as torch's fused_dropout, a different (device-resident,
// graph-replayable) random stream.  ctr[0] = launch counter, ctr[1] (and ctr[2..65]) = tickets: the last workgroup to finish bumps the
// counter, so every workgroup of a launch has read the same value.
__device__ __forceinline__ uint32_t drop_hash(uint32_t seed, uint32_t ctr, uint32_t idx) {
  uint32_t x = idx * 0x9e3779b1u + seed;
  x ^= ctr * 0x85ebca77u + 0x165667b1u;
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;    // lowbias32 finaliser
  x += ctr; x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12;
  return x;
}

template <bool VEC4>
__global__ void __launch_bounds__(SP_TPB) k_sage_epilogue(const bf16_t* __restrict__ a, int64_t a_stride, const bf16_t* __restrict__ b,
                                                         int64_t b_stride, int n_rows, int dim, uint32_t drop_thresh, float scale,
                                                         uint32_t seed, unsigned long long* ctr, bf16_t* __restrict__ out,
                                                         int64_t out_stride, bf16_t* __restrict__ norm_out) {
  const int lane = lane_id();
  const int row = blockIdx.x * (SP_TPB / 64) + (threadIdx.x >> 6);
  const uint32_t c = drop_thresh ? (uint32_t)__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
  if (row < n_rows) {
    const bf16_t *pa = a + row * a_stride, *pb = b + row * b_stride;
    bf16_t* po = out + row * out_stride;
    float s = 0.f;
    constexpr int W = VEC4 ? 4 : 1;
    for (int col = lane * W; col < dim; col += 64 * W) {
      float v[W];
      if (VEC4) {
        const f4 x = load4(pa + col), y = load4(pb + col);
        v[0] = x.x + y.x; v[1 % W] = x.y + y.y; v[2 % W] = x.z + y.z; v[3 % W] = x.w + y.w;
      } else v[0] = bf2f(pa[col]) + bf2f(pb[col]);
#pragma unroll
      for (int i = 0; i < W; ++i) {
        float t = rbf(v[i]);                                  // the add's bf16 result
        t = t > 0.f ? t : 0.f;                                // relu
        if (drop_thresh) {
          const bool keep = drop_hash(seed, c, (uint32_t)(row * dim + col + i)) >= drop_thresh;
          t = keep ? rbf(t * scale) : 0.f;
        }
        v[i] = t;
        s += t * t;
      }
      if (VEC4) {
        uint2 o;
        o.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1 % W]) << 16);
        o.y = (uint32_t)f2bf(v[2 % W]) | ((uint32_t)f2bf(v[3 % W]) << 16);
        *reinterpret_cast<uint2*>(po + col) = o;
      } else po[col] = f2bf(v[0]);
    }
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    if (lane == 0 && norm_out) norm_out[row] = f2bf(sqrtf(s));
  }
  if (drop_thresh) {                                          // launch counter: bumped once, by the last workgroup
    // two-level ticket (ctr[2 + g], g = workgroup % 64, then ctr[1]): a same-address atomic costs ~10 ns, and the 1200
    // workgroups of the input layer's launch spent 12 of its 20 us queueing on ONE ticket word
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned G = gridDim.x < 64u ? gridDim.x : 64u, g = blockIdx.x % G;
      const unsigned long long members = (gridDim.x - g + G - 1) / G;
      if (atomicAdd(ctr + 2 + g, 1ull) == members - 1) {
        __hip_atomic_store(ctr + 2 + g, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (atomicAdd(ctr + 1, 1ull) == (unsigned long long)G - 1) {
          __hip_atomic_store(ctr + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          atomicAdd(ctr, 1ull);
        }
      }
    }
  }
}

// d(a) = d(b) = dout * scale where the stored output is positive (kept and past the ReLU), else 0
template <bool VEC4>
__global__ void __launch_bounds__(SP_TPB) k_sage_epilogue_bwd(const bf16_t* __restrict__ dout, int64_t d_stride, const bf16_t* __restrict__ out,
                                                             int64_t out_stride, int n_rows, int dim, float scale,
                                                             bf16_t* __restrict__ din, int64_t din_stride) {
  const int lane = lane_id();
  const int row = blockIdx.x * (SP_TPB / 64) + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  constexpr int W = VEC4 ? 4 : 1;
  for (int col = lane * W; col < dim; col += 64 * W) {
    if (VEC4) {
      const f4 g = load4(dout + row * d_stride + col), o = load4(out + row * out_stride + col);
      f4 r;
      r.x = o.x > 0.f ? g.x * scale : 0.f; r.y = o.y > 0.f ? g.y * scale : 0.f;
      r.z = o.z > 0.f ? g.z * scale : 0.f; r.w = o.w > 0.f ? g.w * scale : 0.f;
      store4<false>(din, row * din_stride + col, r);
    } else {
      const float g = bf2f(dout[row * d_stride + col]);
      din[row * din_stride + col] = f2bf(bf2f(out[row * out_stride + col]) > 0.f ? g * scale : 0.f);
    }
  }
}

template <bool BWD>
int launch_spmm(const int* row_ptr, const int* t_edge, const int* src, const int* dst, const int* blk_indptr, const void* w,
                const void* h, int64_t h_stride, int n_rows, const int* nnz_dev, int nnz, int dim, int mean, void* out,
                int64_t out_stride, int out_fp32, float* part, hipStream_t st) {
  if (n_rows <= 0 || dim <= 0) return 0;
  const int EC = bliss_spmm_chunk_edges(nnz);
  if (nnz > EC && !part) return BLISS_EINVAL;
  const bool vec4 = (dim % 4 == 0) && (h_stride % 4 == 0) && (out_stride % 4 == 0) &&
                    (((uintptr_t)h) % 8 == 0) && (((uintptr_t)out) % (out_fp32 ? 16 : 8) == 0);
  const int nchunks = nnz > 0 ? (nnz + EC - 1) / EC : 1;
  dim3 grid((nchunks + SP_TPB / 64 - 1) / (SP_TPB / 64)), block(SP_TPB), gfix((n_rows + SP_TPB / 64 - 1) / (SP_TPB / 64));
#define GO(V, F) PROF_LAUNCH(BWD ? BK_SPMM_BWD : BK_SPMM_FWD, st, k_spmm<V, F, BWD><<<grid, block, 0, st>>>(row_ptr, t_edge, src, dst, blk_indptr, (const bf16_t*)w, (const bf16_t*)h, h_stride, n_rows, nnz_dev, nnz, dim, mean, out, out_stride, part, EC))
  if (vec4) { if (out_fp32) GO(true, true); else GO(true, false); }
  else      { if (out_fp32) GO(false, true); else GO(false, false); }
#undef GO
  {
    const bool fv = vec4 && (((uintptr_t)part) % 16 == 0);
#define FIX(F, V) PROF_LAUNCH(BK_SPMM_FIXUP, st, k_spmm_fixup<F, BWD, V><<<gfix, block, 0, st>>>(row_ptr, n_rows, dim, mean, part, out, out_stride, EC))
    if (out_fp32) { if (fv) FIX(true, true); else FIX(true, false); }
    else          { if (fv) FIX(false, true); else FIX(false, false); }
#undef FIX
  }
  return (int)hipGetLastError();
}

}  // namespace

extern "C" {

// A wave walks its chunk edge by edge (a dependent chain of broadcasts and row loads), so the chunk length trades
// parallelism against per-chunk overhead: the sampled blocks of this path hold 2 K - 500 K edges, far too few for
// 64-edge chunks to fill the chip: 64 edges per wave from 200 K edges up, 32 from 50 K, 16 below.
int bliss_spmm_chunk_edges(int32_t nnz_bound) { return nnz_bound >= 200000 ? 64 : (nnz_bound >= 50000 ? 32 : 16); }

int bliss_spmm_fwd(const int32_t* indptr, const int32_t* src, const int32_t* dst, const void* w, const void* h,
                   int64_t h_stride, int32_t n_dst, const int32_t* nnz_dev, int32_t nnz, int32_t dim, int mean, void* out,
                   int64_t out_stride, int out_fp32, float* partials, void* stream) {
  if (!indptr || !h || !out || nnz < 0 || (nnz > 0 && (!src || !dst))) return BLISS_EINVAL;
  return launch_spmm<false>(indptr, nullptr, src, dst, nullptr, w, h, h_stride, n_dst, nnz_dev, nnz, dim, mean, out, out_stride, out_fp32,
                            partials, (hipStream_t)stream);
}

int bliss_spmm_bwd(const int32_t* t_indptr, const int32_t* t_edge, const int32_t* src, const int32_t* dst,
                   const int32_t* indptr, const void* w, const void* gout, int64_t gout_stride, int32_t n_src,
                   const int32_t* nnz_dev, int32_t nnz, int32_t dim, int mean, void* gh, int64_t gh_stride, int out_fp32,
                   float* partials, void* stream) {
  if (!t_indptr || !indptr || !gout || !gh || nnz < 0 || (nnz > 0 && (!t_edge || !src || !dst))) return BLISS_EINVAL;
  return launch_spmm<true>(t_indptr, t_edge, src, dst, indptr, w, gout, gout_stride, n_src, nnz_dev, nnz, dim, mean, gh, gh_stride,
                           out_fp32, partials, (hipStream_t)stream);
}

int bliss_sage_epilogue_fwd(const void* a, int64_t a_stride, const void* b, int64_t b_stride, int32_t n_rows, int32_t dim,
                            float p_drop, uint32_t seed, uint64_t* ctr, void* out, int64_t out_stride, void* norm_out, void* stream) {
  if (!a || !b || !out || n_rows < 0 || dim <= 0 || p_drop < 0.f || p_drop >= 1.f || (p_drop > 0.f && !ctr)) return BLISS_EINVAL;
  if (n_rows == 0) return 0;
  const bool v4 = dim % 4 == 0 && a_stride % 4 == 0 && b_stride % 4 == 0 && out_stride % 4 == 0 && ((uintptr_t)a) % 8 == 0 &&
                  ((uintptr_t)b) % 8 == 0 && ((uintptr_t)out) % 8 == 0;
  const uint32_t thresh = p_drop > 0.f ? (uint32_t)((double)p_drop * 4294967296.0) : 0u;
  const float scale = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
  hipStream_t st = (hipStream_t)stream;
  const int grid = (n_rows + SP_TPB / 64 - 1) / (SP_TPB / 64);
  if (v4) k_sage_epilogue<true><<<grid, SP_TPB, 0, st>>>((const bf16_t*)a, a_stride, (const bf16_t*)b, b_stride, n_rows, dim, thresh, scale, seed,
                                                        (unsigned long long*)ctr, (bf16_t*)out, out_stride, (bf16_t*)norm_out);
  else k_sage_epilogue<false><<<grid, SP_TPB, 0, st>>>((const bf16_t*)a, a_stride, (const bf16_t*)b, b_stride, n_rows, dim, thresh, scale, seed,
                                                      (unsigned long long*)ctr, (bf16_t*)out, out_stride, (bf16_t*)norm_out);
  return (int)hipGetLastError();
}

int bliss_sage_epilogue_bwd(const void* dout, int64_t dout_stride, const void* out, int64_t out_stride, int32_t n_rows, int32_t dim,
                            float p_drop, void* din, int64_t din_stride, void* stream) {
  if (!dout || !out || !din || n_rows < 0 || dim <= 0 || p_drop < 0.f || p_drop >= 1.f) return BLISS_EINVAL;
  if (n_rows == 0) return 0;
  const bool v4 = dim % 4 == 0 && dout_stride % 4 == 0 && out_stride % 4 == 0 && din_stride % 4 == 0 && ((uintptr_t)dout) % 8 == 0 &&
                  ((uintptr_t)out) % 8 == 0 && ((uintptr_t)din) % 8 == 0;
  const float scale = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
  hipStream_t st = (hipStream_t)stream;
  const int grid = (n_rows + SP_TPB / 64 - 1) / (SP_TPB / 64);
  if (v4) k_sage_epilogue_bwd<true><<<grid, SP_TPB, 0, st>>>((const bf16_t*)dout, dout_stride, (const bf16_t*)out, out_stride, n_rows, dim, scale,
                                                            (bf16_t*)din, din_stride);
  else k_sage_epilogue_bwd<false><<<grid, SP_TPB, 0, st>>>((const bf16_t*)dout, dout_stride, (const bf16_t*)out, out_stride, n_rows, dim, scale,
                                                          (bf16_t*)din, din_stride);
  return (int)hipGetLastError();
}

int bliss_gather_rows(const void* feat, int64_t feat_stride, const int32_t* ids, int32_t n_rows, int32_t dim, void* out,
                      int64_t out_stride, void* norm_out, void* stream) {
  if (!feat || !ids || !out || dim <= 0 || feat_stride < dim || out_stride < dim) return BLISS_EINVAL;
  if (n_rows <= 0) return 0;
  const size_t lds = (size_t)(SP_TPB / 64) * ((dim + 7) & ~7) * sizeof(bf16_t);
  if (lds > 48 * 1024) return BLISS_EINVAL;           // (caller gathers and norms separately)
  hipStream_t st = (hipStream_t)stream;
  const int grid = (n_rows + SP_TPB / 64 - 1) / (SP_TPB / 64);
  const uintptr_t al = (uintptr_t)feat | (uintptr_t)out;
  const int norm_vec4 = (dim % 4 == 0) && (out_stride % 4 == 0) && (((uintptr_t)out) % 8 == 0);     // = bliss_embed_norm(out)
  if (dim % 4 == 0 && feat_stride % 4 == 0 && out_stride % 4 == 0 && al % 8 == 0)
    k_gather_rows_norm<4><<<grid, SP_TPB, lds, st>>>((const bf16_t*)feat, feat_stride, ids, n_rows, dim, (bf16_t*)out, out_stride, (bf16_t*)norm_out, norm_vec4);
  else if (dim % 2 == 0 && feat_stride % 2 == 0 && out_stride % 2 == 0 && al % 4 == 0)
    k_gather_rows_norm<2><<<grid, SP_TPB, lds, st>>>((const bf16_t*)feat, feat_stride, ids, n_rows, dim, (bf16_t*)out, out_stride, (bf16_t*)norm_out, norm_vec4);
  else
    k_gather_rows_norm<1><<<grid, SP_TPB, lds, st>>>((const bf16_t*)feat, feat_stride, ids, n_rows, dim, (bf16_t*)out, out_stride, (bf16_t*)norm_out, norm_vec4);
  return (int)hipGetLastError();
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
