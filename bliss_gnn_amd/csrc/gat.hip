// GATv2 attention path for gfx950 -- the message-passing part of custom_GATv2Conv.forward (model.py:48-112):
//   e_ij[h]   = attn[h,:] . leaky_relu(el[src_j,h,:] + er[dst_i,h,:])           model.py:82-86  (returned as "attention", :108-110)
//   a_ij[h]   = softmax over the in-edges of i                                    model.py:88-90  (dglnn.functional.edge_softmax)
//   out[i,h,:] = sum_j a_ij[h] * el[src_j,h,:]                                    model.py:98     (update_all(u_mul_e, sum))
// and their backward passes, plus calculate_alpha's GAT branch (bandit_sampler.py:146-154).
//
// DGL materialises the [B, H, D'] edge tensor (363 MB at layer 0 of the Reddit config); here every
// kernel recomputes el+er on the fly from the two gathered rows, so only [B, H] logits/attention ever touch HBM.
// Balanced traversal: a wave owns a fixed number of consecutive edges (like csrc/spmm.hip); row sums that cross
// chunk boundaries go through fp32 partials and the shared fix-up kernel, so everything is deterministic.
// Arithmetic of the forward kernels (round 3): the reference runs this path as a chain of bf16 tensor ops, so every op
// rounds (u_add_v, leaky_relu, * attn, the sum over D in fp32; e - max, exp, the per-destination sum, the division); the
// kernels round at the same points (tests/golden/gat*_model_exp3.npz: a reference run of model.py's own forward), the
// per-destination sum of the softmax is exact fixed point like every copy_e_sum (common.cuh), and the aggregation adds
// fp32 products with one rounding.  F32 variants (no intermediate rounding, float outputs) exist for the north star's
// 1e-4 check against fp32 math on the same bf16 operands.  The backward kernels differentiate the fp32 formulas.
#include "common.cuh"
#include "bliss_gnn.h"
#include "prof.h"

namespace {

#define GAT_TPB 256
#define GAT_EC 16          // edges per wave in the per-edge dot kernels
#define GAT_MAXH 8

__device__ __forceinline__ float lrelu(float x, float s) { return x > 0.f ? x : s * x; }

// per-edge, per-head dot product of two gathered rows.
//   MODE 0 (logits):  e[e,h]  = sum_d attn[h,d] * lrelu(feat[src_e,h,d] + feat[dst_e,h,d])
//   MODE 1 (d_a):     out[e,h] = sum_d g[dst_e,h,d] * feat[src_e,h,d]
struct g4 { float x, y, z, w; };
__device__ __forceinline__ g4 gload4(const bf16_t* p) {
  const uint2 v = *reinterpret_cast<const uint2*>(p);
  g4 r;
  r.x = __uint_as_float(v.x << 16); r.y = __uint_as_float(v.x & 0xffff0000u);
  r.z = __uint_as_float(v.y << 16); r.w = __uint_as_float(v.y & 0xffff0000u);
  return r;
}
__device__ __forceinline__ int gb_i(int v, int j) { return __builtin_amdgcn_readlane(v, j); }    // wave-uniform lane index

template <int MODE, bool VEC4, bool F32>
__global__ void __launch_bounds__(GAT_TPB) k_gat_edge_dot(const int* __restrict__ src, const int* __restrict__ dst,
                                                         const int* __restrict__ nnz_dev, int nnz_host,
                                                         const bf16_t* __restrict__ feat, int64_t feat_stride,
                                                         const bf16_t* __restrict__ g, int64_t g_stride,
                                                         const bf16_t* __restrict__ attn, int H, int D, float slope,
                                                         void* __restrict__ out_v) {
  bf16_t* out = (bf16_t*)out_v;
  float* out32 = (float*)out_v;
  const int nnz = nnz_dev ? min(*nnz_dev, nnz_host) : nnz_host;
  const int lane = lane_id();
  const int chunk = blockIdx.x * (GAT_TPB / 64) + (threadIdx.x >> 6);
  const int e0 = chunk * GAT_EC, e1 = min(nnz, e0 + GAT_EC);
  // MODE 0, bf16 mode: model.py:82-86 op by op -- rbf(el + er), rbf(leaky_relu), rbf(* attn), fp32 sum over D, one rounding
  auto term = [&](float t, float x, float y) -> float {
    if (F32) return t * lrelu(x + y, slope);
    return rbf(t * rbf(lrelu(rbf(x + y), slope)));
  };
  const int HD = H * D;
  constexpr int W = VEC4 ? 4 : 1;
  // the chunk's endpoints once, coalesced; broadcast per edge with v_readlane
  int my_s = 0, my_d = 0;
  if (e0 + lane < e1) { my_s = src[e0 + lane]; my_d = dst[e0 + lane]; }
  for (int e = e0; e < e1; ++e) {
    const bf16_t* a = feat + (int64_t)gb_i(my_s, e - e0) * feat_stride;
    const bf16_t* b = (MODE == 0 ? feat + (int64_t)gb_i(my_d, e - e0) * feat_stride : g + (int64_t)gb_i(my_d, e - e0) * g_stride);
    float part[GAT_MAXH];
#pragma unroll
    for (int h = 0; h < GAT_MAXH; ++h) part[h] = 0.f;
    for (int idx = lane * W; idx < HD; idx += 64 * W) {
      float v;
      if (VEC4) {                                       // D % 4 == 0: the four columns belong to one head
        const g4 x = gload4(a + idx), y = gload4(b + idx);
        if (MODE == 0) {
          const g4 t = gload4(attn + idx);
          v = term(t.x, x.x, y.x) + term(t.y, x.y, y.y) + term(t.z, x.z, y.z) + term(t.w, x.w, y.w);
        } else v = x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
      } else {
        const float x = bf2f(a[idx]), y = bf2f(b[idx]);
        v = MODE == 0 ? term(bf2f(attn[idx]), x, y) : x * y;
      }
      const int hd = idx / D;
#pragma unroll
      for (int h = 0; h < GAT_MAXH; ++h) if (h == hd) part[h] += v;
    }
#pragma unroll
    for (int h = 0; h < GAT_MAXH; ++h) {
      if (h < H) {
        float s = part[h];
        for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
        if (lane == 0) { if (F32) out32[(int64_t)e * H + h] = s; else out[(int64_t)e * H + h] = f2bf(s); }
      }
    }
  }
}

// edge softmax over the in-edges of every destination (one wave per destination), forward and backward
//   FWD: a = exp(e - max) / sum                      BWD: de = a * (da - sum_e' a da)
template <bool BWD>
__global__ void __launch_bounds__(GAT_TPB) k_gat_softmax(const int* __restrict__ indptr, int n_dst, const bf16_t* __restrict__ x,
                                                        const bf16_t* __restrict__ a_in, int H, bf16_t* __restrict__ out, int* err) {
  const int lane = lane_id();
  const int row = blockIdx.x * (GAT_TPB / 64) + (threadIdx.x >> 6);
  if (row >= n_dst) return;
  const int beg = indptr[row], end = indptr[row + 1];
  for (int h = 0; h < H; ++h) {
    if (!BWD) {
      // [DGL-recalled] edge_softmax = max, exp(e - max), sum, divide -- four tensor ops in the dtype of e: every one rounds to
      // bf16, and the sum is a copy_e_sum (exact, rounded once).  exp through double: the correctly rounded fp32 value, which
      // is what torch's CPU kernel rounds to bf16 (csrc/exp3.hip does the same, exhaustively tested there)
      float m = -__builtin_inff();
      for (int e = beg + lane; e < end; e += 64) m = fmaxf(m, bf2f(x[(int64_t)e * H + h]));
      for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
      int bad = 0;
      int64_t s = 0;
      for (int e = beg + lane; e < end; e += 64) {
        const bf16_t sc = f2bf((float)exp((double)rbf(bf2f(x[(int64_t)e * H + h]) - m)));
        out[(int64_t)e * H + h] = sc;                   // (own element, re-read by the same lane below)
        s += bf_to_fixed(sc, FRAC_DST, &bad);
      }
      for (int d = 32; d >= 1; d >>= 1) {
        const int lo = __shfl_xor((int)(s & 0xffffffffll), d), hi = __shfl_xor((int)(s >> 32), d);
        s += ((int64_t)hi << 32) | (uint32_t)lo;
      }
      float ssum = bf2f(fixed_to_bf(s, FRAC_DST, &bad));
      if (__any(bad != 0)) ssum = __builtin_nanf("");   // a non-finite logit: the reference's sum, and with it the whole row, is NaN
      for (int e = beg + lane; e < end; e += 64) out[(int64_t)e * H + h] = f2bf(bf2f(out[(int64_t)e * H + h]) / ssum);
      if (bad && err) atomicOr(err, bad);
    } else {
      float t = 0.f;                                   // x = d_a, a_in = a
      for (int e = beg + lane; e < end; e += 64) t += bf2f(a_in[(int64_t)e * H + h]) * bf2f(x[(int64_t)e * H + h]);
      for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d);
      for (int e = beg + lane; e < end; e += 64) {
        const float a = bf2f(a_in[(int64_t)e * H + h]);
        out[(int64_t)e * H + h] = f2bf(a * (bf2f(x[(int64_t)e * H + h]) - t));
      }
    }
  }
}

// the same softmax in plain fp32 on float logits, float result (the 1e-4 check)
__global__ void __launch_bounds__(GAT_TPB) k_gat_softmax_f32(const int* __restrict__ indptr, int n_dst, const float* __restrict__ x, int H,
                                                            float* __restrict__ out) {
  const int lane = lane_id();
  const int row = blockIdx.x * (GAT_TPB / 64) + (threadIdx.x >> 6);
  if (row >= n_dst) return;
  const int beg = indptr[row], end = indptr[row + 1];
  for (int h = 0; h < H; ++h) {
    float m = -__builtin_inff();
    for (int e = beg + lane; e < end; e += 64) m = fmaxf(m, x[(int64_t)e * H + h]);
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    float s = 0.f;
    for (int e = beg + lane; e < end; e += 64) s += expf(x[(int64_t)e * H + h] - m);
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    for (int e = beg + lane; e < end; e += 64) out[(int64_t)e * H + h] = expf(x[(int64_t)e * H + h] - m) / s;
  }
}

// Row sums of per-edge vectors that are never materialised:
//   val(e, col) = coef[e, h(col)] * feat[nbr_e, col]                                   (LBWD == false: aggregation fwd / bwd)
//   val(e, col) = de[e, h(col)] * attn[col] * lrelu'(feat[src_e, col] + feat[dst_e, col])   (LBWD == true: logits backward)
// rows = destinations (BY_SRC == false, CSR order) or sources (BY_SRC == true, through t_edge).  Merge-style chunks of
// 64 edges per wave; rows cut by chunk boundaries leave partials for k_gat_fixup (same scheme as csrc/spmm.hip).
// With LBWD && !BY_SRC the kernel also accumulates d_attn[col] = sum_e de[e,h] * lrelu(x) (fp32 atomics per workgroup).
#define GEC 16                                          // edges per wave: every edge costs up to 2 KB of row gathers, so short
                                                        // chunks (many waves in flight) beat long serial ones
template <bool LBWD, bool BY_SRC, bool VEC4, bool F32 = false>
__global__ void __launch_bounds__(GAT_TPB) k_gat_rows(const int* __restrict__ row_ptr, const int* __restrict__ t_edge,
                                                     const int* __restrict__ src, const int* __restrict__ dst,
                                                     const int* __restrict__ nnz_dev, int nnz_host,
                                                     const bf16_t* __restrict__ coef, const bf16_t* __restrict__ feat,
                                                     int64_t feat_stride, const bf16_t* __restrict__ attn, int H, int D,
                                                     float slope, bf16_t* __restrict__ out, int64_t out_stride,
                                                     float* __restrict__ part, float* __restrict__ d_attn,
                                                     const bf16_t* __restrict__ coef2 = nullptr, const bf16_t* __restrict__ g2 = nullptr,
                                                     int64_t g2_stride = 0) {
  // (LBWD && BY_SRC with g2: the aggregation's backward rides along -- val(e, col) += coef2[e, h] * g2[dst_e, col], so that
  // d el_j = sum over out-edges of (d e attn lrelu' + a d rst_i) comes out of ONE pass over the source's edges)
  __shared__ float sh_attn[LBWD && !BY_SRC ? 2048 : 1];
  const int nnz = nnz_dev ? min(*nnz_dev, nnz_host) : nnz_host;
  const int lane = lane_id();
  const int chunk = blockIdx.x * (GAT_TPB / 64) + (threadIdx.x >> 6);
  const int nchunks = nnz > 0 ? (nnz + GEC - 1) / GEC : 1;
  const int HD = H * D;
  constexpr int W = VEC4 ? 4 : 1;
  const bool do_attn = LBWD && !BY_SRC && d_attn != nullptr;
  if (do_attn) { for (int i = threadIdx.x; i < HD && i < 2048; i += GAT_TPB) sh_attn[i] = 0.f; __syncthreads(); }
  if (chunk < nchunks && nnz > 0) {
    const int c0 = chunk * GEC, c1 = min(nnz, c0 + GEC), cnt = c1 - c0;
    int my_row = -1, my_e = 0, my_s = 0, my_d = 0;       // the chunk's edges once, coalesced; broadcast per edge with v_readlane
    if (lane < cnt) {
      my_e = BY_SRC ? t_edge[c0 + lane] : c0 + lane;
      my_s = src[my_e]; my_d = dst[my_e];
      my_row = BY_SRC ? my_s : my_d;
    }
    for (int col0 = 0; col0 < HD; col0 += 64 * W) {
      const int col = col0 + lane * W;
      const bool act = col < HD;
      const int hd = act ? col / D : 0;                 // VEC4 requires D % 4 == 0: the lane's columns share a head
      float at[W], acc[W], attn_acc[W];
#pragma unroll
      for (int i = 0; i < W; ++i) { at[i] = (LBWD && act) ? bf2f(attn[col + i]) : 0.f; acc[i] = 0.f; attn_acc[i] = 0.f; }
      int cur = gb_i(my_row, 0);
      auto flush = [&](int r) {
        if (act) {
          const int rb = row_ptr[r], re = row_ptr[r + 1];
          const bool starts = rb >= c0, ends = re <= c1;
          if (starts && ends) {
            if (F32) {
              float* o32 = reinterpret_cast<float*>(out) + r * out_stride + col;
#pragma unroll
              for (int i = 0; i < W; ++i) o32[i] = acc[i];
            } else if (VEC4) {
              uint2 v;
              v.x = (uint32_t)f2bf(acc[0]) | ((uint32_t)f2bf(acc[1 % W]) << 16);
              v.y = (uint32_t)f2bf(acc[2 % W]) | ((uint32_t)f2bf(acc[3 % W]) << 16);
              *reinterpret_cast<uint2*>(out + r * out_stride + col) = v;
            } else out[r * out_stride + col] = f2bf(acc[0]);
          } else {
            float* q = part + ((int64_t)chunk * 2 + (starts ? 1 : 0)) * HD + col;
#pragma unroll
            for (int i = 0; i < W; ++i) q[i] = acc[i];
          }
        }
#pragma unroll
        for (int i = 0; i < W; ++i) acc[i] = 0.f;
      };
      for (int j = 0; j < cnt; ++j) {
        const int r = gb_i(my_row, j), e = gb_i(my_e, j), s_ = gb_i(my_s, j), d_ = gb_i(my_d, j);
        if (r != cur) { flush(cur); cur = r; }
        if (act) {
          const float cf = F32 ? reinterpret_cast<const float*>(coef)[(int64_t)e * H + hd] : bf2f(coef[(int64_t)e * H + hd]);
          if (LBWD && BY_SRC && g2) {
            const float cf2 = bf2f(coef2[(int64_t)e * H + hd]);
            if (VEC4) {
              const g4 f = gload4(g2 + (int64_t)d_ * g2_stride + col);
              acc[0] += cf2 * f.x; acc[1 % W] += cf2 * f.y; acc[2 % W] += cf2 * f.z; acc[3 % W] += cf2 * f.w;
            } else acc[0] += cf2 * bf2f(g2[(int64_t)d_ * g2_stride + col]);
          }
          if (LBWD) {
            float xs[W], xd[W];
            if (VEC4) {
              const g4 a = gload4(feat + (int64_t)s_ * feat_stride + col), b = gload4(feat + (int64_t)d_ * feat_stride + col);
              xs[0] = a.x; xs[1 % W] = a.y; xs[2 % W] = a.z; xs[3 % W] = a.w;
              xd[0] = b.x; xd[1 % W] = b.y; xd[2 % W] = b.z; xd[3 % W] = b.w;
            } else { xs[0] = bf2f(feat[(int64_t)s_ * feat_stride + col]); xd[0] = bf2f(feat[(int64_t)d_ * feat_stride + col]); }
#pragma unroll
            for (int i = 0; i < W; ++i) {
              const float x = xs[i] + xd[i];
              acc[i] += cf * at[i] * (x > 0.f ? 1.f : slope);
              if (do_attn) attn_acc[i] += cf * lrelu(x, slope);
            }
          } else {
            const int nb = BY_SRC ? d_ : s_;
            if (VEC4) {
              const g4 f = gload4(feat + (int64_t)nb * feat_stride + col);
              acc[0] += cf * f.x; acc[1 % W] += cf * f.y; acc[2 % W] += cf * f.z; acc[3 % W] += cf * f.w;
            } else acc[0] += cf * bf2f(feat[(int64_t)nb * feat_stride + col]);
          }
        }
      }
      flush(cur);
      if (do_attn && act) {
#pragma unroll
        for (int i = 0; i < W; ++i) if (col + i < 2048) atomicAdd(&sh_attn[col + i], attn_acc[i]);
      }
    }
  }
  if (do_attn) {
    __syncthreads();
    for (int i = threadIdx.x; i < HD && i < 2048; i += GAT_TPB) { const float v = sh_attn[i]; if (v != 0.f) atomicAdd(d_attn + i, v); }
  }
}

// rows without edges -> 0; rows cut by chunk boundaries -> tail partial of their first chunk + head partials of the rest
// (added in chunk order: deterministic).  VEC4: a lane owns four consecutive columns (float4 partials, 8-byte stores).
template <bool VEC4, bool F32 = false>
__global__ void __launch_bounds__(GAT_TPB) k_gat_fixup(const int* __restrict__ row_ptr, int n_rows, int HD,
                                                      const float* __restrict__ part, bf16_t* __restrict__ out, int64_t out_stride) {
  const int lane = lane_id();
  const int r = blockIdx.x * (GAT_TPB / 64) + (threadIdx.x >> 6);
  if (r >= n_rows) return;
  const int rb = row_ptr[r], re = row_ptr[r + 1];
  constexpr int W = VEC4 ? 4 : 1;
  if (F32) {                                            // float rows (the 1e-4 check): element-wise, same order of additions
    float* o32 = reinterpret_cast<float*>(out);
    if (re <= rb) { for (int col = lane; col < HD; col += 64) o32[r * out_stride + col] = 0.f; return; }
    const int c = rb / GEC, c_end = (re - 1) / GEC;
    if (c_end == c) return;
    for (int col = lane; col < HD; col += 64) {
      float sum = part[((int64_t)c * 2 + 1) * HD + col];
      for (int cc = c + 1; cc <= c_end; ++cc) sum += part[((int64_t)cc * 2) * HD + col];
      o32[r * out_stride + col] = sum;
    }
    return;
  }
  if (re <= rb) {
    for (int col = lane * W; col < HD; col += 64 * W) {
      if (VEC4) *reinterpret_cast<uint2*>(out + r * out_stride + col) = make_uint2(0u, 0u); else out[r * out_stride + col] = 0;
    }
    return;
  }
  const int c = rb / GEC, c_end = (re - 1) / GEC;
  if (c_end == c) return;
  for (int col = lane * W; col < HD; col += 64 * W) {
    if (VEC4) {
      float4 sum = *reinterpret_cast<const float4*>(part + ((int64_t)c * 2 + 1) * HD + col);
#pragma unroll 4
      for (int cc = c + 1; cc <= c_end; ++cc) {
        const float4 v = *reinterpret_cast<const float4*>(part + ((int64_t)cc * 2) * HD + col);
        sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
      }
      uint2 o;
      o.x = (uint32_t)f2bf(sum.x) | ((uint32_t)f2bf(sum.y) << 16);
      o.y = (uint32_t)f2bf(sum.z) | ((uint32_t)f2bf(sum.w) << 16);
      *reinterpret_cast<uint2*>(out + r * out_stride + col) = o;
    } else {
      float sum = part[((int64_t)c * 2 + 1) * HD + col];
      for (int cc = c + 1; cc <= c_end; ++cc) sum += part[((int64_t)cc * 2) * HD + col];
      out[r * out_stride + col] = f2bf(sum);
    }
  }
}

// calculate_alpha, model == 'gat' (bandit_sampler.py:148-154): exact per-destination sums of q_ij and a_ij (the
// arithmetic contract of common.cuh), alpha = nan_to_num(a / sum a) * sum q.  One wave per destination.
__global__ void __launch_bounds__(GAT_TPB) k_gat_alpha(const int* __restrict__ indptr, int n_dst, const bf16_t* __restrict__ q,
                                                      const bf16_t* __restrict__ a, bf16_t* __restrict__ alpha, int* err) {
  const int lane = lane_id();
  const int row = blockIdx.x * (GAT_TPB / 64) + (threadIdx.x >> 6);
  if (row >= n_dst) return;
  const int beg = indptr[row], end = indptr[row + 1];
  int bad = 0;
  int64_t sq = 0, sa = 0;
  for (int e = beg + lane; e < end; e += 64) { sq += bf_to_fixed(q[e], FRAC_DST, &bad); sa += bf_to_fixed(a[e], FRAC_DST, &bad); }
  for (int d = 32; d >= 1; d >>= 1) {
    int lo = __shfl_xor((int)(sq & 0xffffffffll), d), hi = __shfl_xor((int)(sq >> 32), d);
    sq += ((int64_t)hi << 32) | (uint32_t)lo;
    lo = __shfl_xor((int)(sa & 0xffffffffll), d); hi = __shfl_xor((int)(sa >> 32), d);
    sa += ((int64_t)hi << 32) | (uint32_t)lo;
  }
  const float qsum = bf2f(fixed_to_bf(sq, FRAC_DST, &bad));       // :150 copy_e_sum(mfg, q_ij)
  const float asum = bf2f(fixed_to_bf(sa, FRAC_DST, &bad));       // :151 copy_e_sum(mfg, attention)
  for (int e = beg + lane; e < end; e += 64) {
    float f = rbf(bf2f(a[e]) / asum);                             // :152 e_div_v
    if (f != f) f = 0.f;                                          // :153 torch.nan_to_num (nan -> 0, +-inf -> +-max bf16)
    else if (f == __builtin_inff()) f = 3.3895313892515355e38f;
    else if (f == -__builtin_inff()) f = -3.3895313892515355e38f;
    alpha[e] = f2bf(f * qsum);                                    // :154 e_dot_v on scalars
  }
  if (bad) atomicOr(err, bad);
}

}  // namespace

extern "C" {

int bliss_gat_chunk_edges(void) { return GEC; }

static int gat_logits(const int32_t* src, const int32_t* dst, const int32_t* nnz_dev, int32_t nnz, const void* feat,
                      int64_t feat_stride, const void* attn, int32_t heads, int32_t head_dim, float negative_slope, void* e_out,
                      void* stream, bool f32) {
  if (!feat || !attn || !e_out || heads <= 0 || heads > GAT_MAXH || head_dim <= 0 || nnz < 0) return BLISS_EINVAL;
  if (nnz == 0) return 0;
  if (!src || !dst) return BLISS_EINVAL;
  const int chunks = (nnz + GAT_EC - 1) / GAT_EC;
  const bool v4 = head_dim % 4 == 0 && feat_stride % 4 == 0 && ((uintptr_t)feat) % 8 == 0 && ((uintptr_t)attn) % 8 == 0;
  const dim3 grid((chunks + 3) / 4);
  hipStream_t st = (hipStream_t)stream;
#define LG(V, F) k_gat_edge_dot<0, V, F><<<grid, GAT_TPB, 0, st>>>(src, dst, nnz_dev, nnz, (const bf16_t*)feat, feat_stride, nullptr, 0, \
                                                                  (const bf16_t*)attn, heads, head_dim, negative_slope, e_out)
  if (f32) { if (v4) LG(true, true); else LG(false, true); }
  else { if (v4) LG(true, false); else LG(false, false); }
#undef LG
  return (int)hipGetLastError();
}

int bliss_gat_logits(const int32_t* src, const int32_t* dst, const int32_t* nnz_dev, int32_t nnz, const void* feat,
                     int64_t feat_stride, const void* attn, int32_t heads, int32_t head_dim, float negative_slope, void* e_out,
                     void* stream) {
  return gat_logits(src, dst, nnz_dev, nnz, feat, feat_stride, attn, heads, head_dim, negative_slope, e_out, stream, false);
}

int bliss_gat_logits_f32(const int32_t* src, const int32_t* dst, const int32_t* nnz_dev, int32_t nnz, const void* feat,
                         int64_t feat_stride, const void* attn, int32_t heads, int32_t head_dim, float negative_slope, float* e_out,
                         void* stream) {
  return gat_logits(src, dst, nnz_dev, nnz, feat, feat_stride, attn, heads, head_dim, negative_slope, e_out, stream, true);
}

int bliss_gat_edge_dot(const int32_t* src, const int32_t* dst, const int32_t* nnz_dev, int32_t nnz, const void* feat,
                       int64_t feat_stride, const void* g, int64_t g_stride, int32_t heads, int32_t head_dim, void* out, void* stream) {
  if (!feat || !g || !out || heads <= 0 || heads > GAT_MAXH || head_dim <= 0 || nnz < 0) return BLISS_EINVAL;
  if (nnz == 0) return 0;
  if (!src || !dst) return BLISS_EINVAL;
  const int chunks = (nnz + GAT_EC - 1) / GAT_EC;
  const bool v4 = head_dim % 4 == 0 && feat_stride % 4 == 0 && g_stride % 4 == 0 && ((uintptr_t)feat) % 8 == 0 && ((uintptr_t)g) % 8 == 0;
  if (v4) k_gat_edge_dot<1, true, false><<<(chunks + 3) / 4, GAT_TPB, 0, (hipStream_t)stream>>>(src, dst, nnz_dev, nnz, (const bf16_t*)feat, feat_stride, (const bf16_t*)g,
                                                                        g_stride, nullptr, heads, head_dim, 0.f, (bf16_t*)out);
  else k_gat_edge_dot<1, false, false><<<(chunks + 3) / 4, GAT_TPB, 0, (hipStream_t)stream>>>(src, dst, nnz_dev, nnz, (const bf16_t*)feat, feat_stride, (const bf16_t*)g,
                                                                        g_stride, nullptr, heads, head_dim, 0.f, (bf16_t*)out);
  return (int)hipGetLastError();
}

int bliss_gat_edge_softmax(const int32_t* indptr, int32_t n_dst, const void* x, const void* a_or_null, int32_t heads, int backward,
                           void* out, void* stream) {
  if (!indptr || !x || !out || heads <= 0 || (backward == 1 && !a_or_null) || backward < 0 || backward > 2) return BLISS_EINVAL;
  if (n_dst <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (backward == 2) k_gat_softmax_f32<<<(n_dst + 3) / 4, GAT_TPB, 0, st>>>(indptr, n_dst, (const float*)x, heads, (float*)out);
  else if (backward) k_gat_softmax<true><<<(n_dst + 3) / 4, GAT_TPB, 0, st>>>(indptr, n_dst, (const bf16_t*)x, (const bf16_t*)a_or_null, heads, (bf16_t*)out, nullptr);
  else k_gat_softmax<false><<<(n_dst + 3) / 4, GAT_TPB, 0, st>>>(indptr, n_dst, (const bf16_t*)x, nullptr, heads, (bf16_t*)out, nullptr);
  return (int)hipGetLastError();
}

int bliss_gat_rows(int which, const int32_t* row_ptr, int32_t n_rows, const int32_t* t_edge, const int32_t* src, const int32_t* dst,
                   const int32_t* nnz_dev, int32_t nnz, const void* coef, const void* feat, int64_t feat_stride, const void* attn,
                   int32_t heads, int32_t head_dim, float negative_slope, void* out, int64_t out_stride, float* partials,
                   float* d_attn, void* stream) {
  if (!row_ptr || !coef || !feat || !out || heads <= 0 || heads > GAT_MAXH || head_dim <= 0 || n_rows <= 0 || nnz < 0) return BLISS_EINVAL;
  if (nnz > 0 && (!src || !dst || !partials)) return BLISS_EINVAL;
  if (which == 4) {             // the forward aggregation with float coefficients and float rows (no rounding: the 1e-4 check)
    if (nnz > 0 && (!src || !dst || !partials)) return BLISS_EINVAL;
    hipStream_t st4 = (hipStream_t)stream;
    const int chunks4 = nnz > 0 ? (nnz + GEC - 1) / GEC : 1;
    k_gat_rows<false, false, false, true><<<(chunks4 + 3) / 4, GAT_TPB, 0, st4>>>(row_ptr, t_edge, src, dst, nnz_dev, nnz, (const bf16_t*)coef, (const bf16_t*)feat,
                                                                                 feat_stride, (const bf16_t*)attn, heads, head_dim, negative_slope,
                                                                                 (bf16_t*)out, out_stride, partials, d_attn);
    k_gat_fixup<false, true><<<(n_rows + 3) / 4, GAT_TPB, 0, st4>>>(row_ptr, n_rows, heads * head_dim, partials, (bf16_t*)out, out_stride);
    return (int)hipGetLastError();
  }
  if ((which & 1) && nnz > 0 && !t_edge) return BLISS_EINVAL;
  if ((which & 2) && (!attn || heads * head_dim > 2048)) return BLISS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int HD = heads * head_dim;
  const int chunks = nnz > 0 ? (nnz + GEC - 1) / GEC : 1;
  dim3 grid((chunks + 3) / 4), block(GAT_TPB);
#define ARGS row_ptr, t_edge, src, dst, nnz_dev, nnz, (const bf16_t*)coef, (const bf16_t*)feat, feat_stride, (const bf16_t*)attn, heads, head_dim, negative_slope, (bf16_t*)out, out_stride, partials, d_attn
  const bool v4 = head_dim % 4 == 0 && feat_stride % 4 == 0 && out_stride % 4 == 0 && ((uintptr_t)feat) % 8 == 0 &&
                  ((uintptr_t)out) % 8 == 0 && (!attn || ((uintptr_t)attn) % 8 == 0);
#define GO(L, S) do { if (v4) k_gat_rows<L, S, true><<<grid, block, 0, st>>>(ARGS); else k_gat_rows<L, S, false><<<grid, block, 0, st>>>(ARGS); } while (0)
  switch (which) {             // bit 0: rows are sources (through t_edge); bit 1: logits backward
    case 0: GO(false, false); break;
    case 1: GO(false, true); break;
    case 2: GO(true, false); break;
    case 3: GO(true, true); break;
    default: return BLISS_EINVAL;
  }
#undef GO
#undef ARGS
  if (v4 && ((uintptr_t)partials) % 16 == 0) k_gat_fixup<true><<<(n_rows + 3) / 4, GAT_TPB, 0, st>>>(row_ptr, n_rows, HD, partials, (bf16_t*)out, out_stride);
  else k_gat_fixup<false><<<(n_rows + 3) / 4, GAT_TPB, 0, st>>>(row_ptr, n_rows, HD, partials, (bf16_t*)out, out_stride);
  return (int)hipGetLastError();
}

int bliss_gat_rows_src_fused(const int32_t* t_indptr, int32_t n_src, const int32_t* t_edge, const int32_t* src, const int32_t* dst,
                             const int32_t* nnz_dev, int32_t nnz, const void* de, const void* a_drop, const void* feat, int64_t feat_stride,
                             const void* g, int64_t g_stride, const void* attn, int32_t heads, int32_t head_dim, float negative_slope,
                             void* out, int64_t out_stride, float* partials, void* stream) {
  if (!t_indptr || !de || !a_drop || !feat || !g || !attn || !out || heads <= 0 || heads > GAT_MAXH || head_dim <= 0 || n_src <= 0 || nnz < 0)
    return BLISS_EINVAL;
  if (nnz > 0 && (!src || !dst || !partials || !t_edge)) return BLISS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int HD = heads * head_dim;
  const int chunks = nnz > 0 ? (nnz + GEC - 1) / GEC : 1;
  const dim3 grid((chunks + 3) / 4), block(GAT_TPB);
  const bool v4 = head_dim % 4 == 0 && feat_stride % 4 == 0 && out_stride % 4 == 0 && g_stride % 4 == 0 && ((uintptr_t)feat) % 8 == 0 &&
                  ((uintptr_t)out) % 8 == 0 && ((uintptr_t)attn) % 8 == 0 && ((uintptr_t)g) % 8 == 0;
#define SRCF(V) k_gat_rows<true, true, V><<<grid, block, 0, st>>>(t_indptr, t_edge, src, dst, nnz_dev, nnz, (const bf16_t*)de, (const bf16_t*)feat, \
    feat_stride, (const bf16_t*)attn, heads, head_dim, negative_slope, (bf16_t*)out, out_stride, partials, nullptr, (const bf16_t*)a_drop, \
    (const bf16_t*)g, g_stride)
  if (v4) SRCF(true); else SRCF(false);
#undef SRCF
  if (v4 && ((uintptr_t)partials) % 16 == 0) k_gat_fixup<true><<<(n_src + 3) / 4, GAT_TPB, 0, st>>>(t_indptr, n_src, HD, partials, (bf16_t*)out, out_stride);
  else k_gat_fixup<false><<<(n_src + 3) / 4, GAT_TPB, 0, st>>>(t_indptr, n_src, HD, partials, (bf16_t*)out, out_stride);
  return (int)hipGetLastError();
}

int bliss_gat_alpha(const int32_t* indptr, int32_t n_dst, const void* q_ij, const void* a_ij, void* alpha_out, int32_t* err, void* stream) {
  if (!indptr || !q_ij || !a_ij || !alpha_out || !err) return BLISS_EINVAL;
  if (n_dst <= 0) return 0;
  k_gat_alpha<<<(n_dst + 3) / 4, GAT_TPB, 0, (hipStream_t)stream>>>(indptr, n_dst, (const bf16_t*)q_ij, (const bf16_t*)a_ij, (bf16_t*)alpha_out, err);
  return (int)hipGetLastError();
}

}  // extern "C"
