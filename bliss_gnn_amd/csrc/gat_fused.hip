// GATv2 message passing, one workgroup per DESTINATION row (round 3) -- custom_GATv2Conv.forward, model.py:82-99, and its
// backward by destination.  csrc/gat.hip runs the same arithmetic as five launches per direction that each re-gather the
// 2 KB feature rows of every edge (logits: source + destination row; aggregation: source row again; backward: seven row
// gathers per edge) and pass [B, H] tensors through memory in between: 760 us forward and 1.2 ms backward per step on the
// Reddit-like config (profiles/r03_d_gat_step_timeline.txt).  Here a workgroup of 8 waves owns one destination i:
//   forward   er_i stays in registers; pass 1 gathers el_j per in-edge -> logits e_ij[h] (stored: the reference returns them
//             as "attention", model.py:108-110) and the per-head maximum; pass 2 is the edge softmax over the stored logits
//             (exact per-destination sum like every copy_e_sum), with attention dropout (model.py:88) folded in; pass 3
//             gathers el_j again (L2-warm) and accumulates  sum_j a_ij el_j.  Two row gathers per edge instead of three, one
//             launch instead of five, no partial/fix-up pass (a row never leaves its workgroup).
//   backward  (by destination) g_i = d rst_i and er_i in registers; pass 1: d a_ij = g_i . el_j and t = sum a d a; pass 2:
//             d e_ij = a (d a - t) (softmax backward); pass 3: d er_i = sum_j d e_ij attn lrelu'(el_j + er_i) and this row's
//             share of d attn.  The by-SOURCE half (d el_j: out-degrees are far more skewed) stays on the merge-style
//             kernel of gat.hip, now with the aggregation's backward folded in.
// The forward rounds exactly where gat.hip's kernels round (= where the reference's bf16 tensor ops round): the golden
// fixtures of tests/golden/gat*_model_exp3.npz hold for both paths.  bf16 conversions use the hardware's
// v_cvt_pk_bf16_f32 (round to nearest even, like common.cuh:f2bf on every finite input).
#include "common.cuh"
#include "bliss_gnn.h"

namespace {

#define GF_TPB 512
#define GF_WAVES (GF_TPB / 64)
#define GF_MAXH 8
#define GF_ITER 4                      // a lane owns up to 4 groups of W consecutive columns: H*D <= 64 * W * 4

__device__ __forceinline__ float rbf_hw(float f) { return (float)(__bf16)f; }
__device__ __forceinline__ bf16_t f2bf_hw(float f) { const __bf16 b = (__bf16)f; return __builtin_bit_cast(unsigned short, b); }
__device__ __forceinline__ float lrelu_f(float x, float s) { return x > 0.f ? x : s * x; }

struct g4f { float v[4]; };
template <bool VEC4>
__device__ __forceinline__ g4f ldrow(const bf16_t* p) {            // W consecutive elements of a row as floats
  g4f r;
  if (VEC4) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    r.v[0] = __uint_as_float(u.x << 16); r.v[1] = __uint_as_float(u.x & 0xffff0000u);
    r.v[2] = __uint_as_float(u.y << 16); r.v[3] = __uint_as_float(u.y & 0xffff0000u);
  } else { r.v[0] = bf2f(p[0]); r.v[1] = r.v[2] = r.v[3] = 0.f; }
  return r;
}

__device__ __forceinline__ uint32_t gf_drop_hash(uint32_t seed, uint32_t ctr, uint32_t idx) {      // = drop_hash of spmm.hip / sage.hip
  uint32_t x = idx * 0x9e3779b1u + seed;
  x ^= ctr * 0x85ebca77u + 0x165667b1u;
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  x += ctr; x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12;
  return x;
}

struct GatFused {
  const int* indptr; const int* src; int n_dst; const int* n_dst_dev;
  const bf16_t* feat; long long feat_stride; const bf16_t* attn; int H, D; float slope;
  bf16_t* e; bf16_t* a; bf16_t* ad;                    // [nnz, H]: logits, softmax, softmax after dropout (== a when p == 0)
  bf16_t* rst; long long rst_stride;                   // forward output [n_dst, H*D]
  unsigned drop_thresh; float drop_scale; unsigned seed; unsigned long long* ctr; unsigned* ctr_used;
  // backward
  const bf16_t* g; long long g_stride;                 // d rst [n_dst, H*D]
  bf16_t* de;                                          // [nnz, H] out: d e (softmax backward), scratch for d a before that
  bf16_t* d_er; long long der_stride;                  // [n_dst, H*D] out
  float* dattn_part;                                   // [n_dst, H*D] out: this row's share of d attn
};

// A wave's share of a row's edges, SWEEP = 64 * GF_WAVES edges at a time: lane l of wave w holds the source id of edge
// base + l * GF_WAVES + w, fetched with ONE coalesced load per sweep; the loop over the wave's edges then reads the id with
// v_readlane (wave-uniform, no dependent global load per edge) and keeps the rows of TWO edges in flight.
// a row of one edge as the lane's NG column groups, still packed (two registers per group of four bf16): FOUR edges' rows are
// kept in flight per wave -- the loop is bound by the latency of these gathers, not by their bytes
template <bool VEC4> struct RawGroup { uint2 u; };
template <bool VEC4, int NG>
__device__ __forceinline__ void ld_edge_raw(RawGroup<VEC4> (&r)[NG], const bf16_t* el, const int (&coff)[NG]) {
  // UNCONDITIONAL loads from clamped (always valid) column offsets: a load under a lane-dependent branch makes the compiler wait
  // for everything outstanding (csrc/sage.hip found the same); columns beyond H*D are masked where the values are used
#pragma unroll
  for (int c = 0; c < NG; ++c) {
    if (VEC4) r[c].u = *reinterpret_cast<const uint2*>(el + coff[c]);
    else { r[c].u.x = el[coff[c]]; r[c].u.y = 0u; }
  }
}
template <bool VEC4, int W, int NG>
__device__ __forceinline__ void unpack_row(float (&x)[NG][W], const RawGroup<VEC4> (&r)[NG]) {
#pragma unroll
  for (int c = 0; c < NG; ++c) {
    if (VEC4) {
      x[c][0] = __uint_as_float(r[c].u.x << 16); x[c][1 % W] = __uint_as_float(r[c].u.x & 0xffff0000u);
      x[c][2 % W] = __uint_as_float(r[c].u.y << 16); x[c][3 % W] = __uint_as_float(r[c].u.y & 0xffff0000u);
    } else x[c][0] = __uint_as_float(r[c].u.x << 16);
  }
}
#define GF_FLIGHT 2

// per-edge, per-head coefficients ([nnz, H] bf16 arrays) of a sweep: lane l fetches those of ITS edge once (agent-scope loads:
// another wave of the workgroup may have written them), the edge loop broadcasts them with v_readlane
template <int NH>
__device__ __forceinline__ void ld_edge_coefs(float (&cv)[NH], const bf16_t* arr, long long e, int H, bool valid) {
  // one agent-scope load per edge where the H values form an aligned 2 / 4 / 8 / 16-byte unit (each atomic load is waited for)
#pragma unroll
  for (int h = 0; h < NH; ++h) cv[h] = 0.f;
  if (!valid) return;
  const bf16_t* q = arr + e * H;
  if (NH >= 4 && H == 4) {
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    cv[0] = bf2f((bf16_t)(v & 0xffffu)); cv[1 % NH] = bf2f((bf16_t)((v >> 16) & 0xffffu));
    cv[2 % NH] = bf2f((bf16_t)((v >> 32) & 0xffffu)); cv[3 % NH] = bf2f((bf16_t)(v >> 48));
  } else if (NH >= 8 && H == 8) {
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long w = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(q) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    cv[0] = bf2f((bf16_t)(v & 0xffffu)); cv[1 % NH] = bf2f((bf16_t)((v >> 16) & 0xffffu));
    cv[2 % NH] = bf2f((bf16_t)((v >> 32) & 0xffffu)); cv[3 % NH] = bf2f((bf16_t)(v >> 48));
    cv[4 % NH] = bf2f((bf16_t)(w & 0xffffu)); cv[5 % NH] = bf2f((bf16_t)((w >> 16) & 0xffffu));
    cv[6 % NH] = bf2f((bf16_t)((w >> 32) & 0xffffu)); cv[7 % NH] = bf2f((bf16_t)(w >> 48));
  } else if (NH >= 2 && H == 2) {
    const unsigned v = __hip_atomic_load(reinterpret_cast<const unsigned*>(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    cv[0] = bf2f((bf16_t)(v & 0xffffu)); cv[1 % NH] = bf2f((bf16_t)(v >> 16));
  } else {
#pragma unroll
    for (int h = 0; h < NH; ++h)
      if (h < H) cv[h] = bf2f(__hip_atomic_load(q + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  }
}
__device__ __forceinline__ float bcast_f32(float v, int j) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j)); }
// the coefficient of this lane's column group (head hd) out of the edge's H broadcast values
template <int NH>
__device__ __forceinline__ float pick_head(const float (&sv)[NH], int hd) {
  float r = 0.f;
#pragma unroll
  for (int h = 0; h < NH; ++h) r = (h == hd) ? sv[h] : r;
  return r;
}

// per-head sums of a lane's partial values: part[h] over the wave (all lanes get the totals)
template <int NH>
__device__ __forceinline__ void wave_sum_heads(float (&part)[NH], int H) {
#pragma unroll
  for (int h = 0; h < NH; ++h)
    if (h < H) { float s = part[h]; for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d); part[h] = s; }
}

template <bool VEC4, int HG>
__global__ void __launch_bounds__(GF_TPB) k_gat_fwd(GatFused p) {
  constexpr int W = VEC4 ? 4 : 1;
  // HG > 0: "a column group is a head" (VEC4, D == 256: group c of 64 lanes x 4 columns IS head c, H == HG <= 4) -- the Reddit
  // config's 4 x 256; every per-head selection below is then a compile-time index.  HG == 0: any H <= 8, D (head by compare)
  constexpr int NG = HG ? HG : GF_ITER, NH = HG ? HG : GF_MAXH;
  __shared__ float sh_acc[GF_WAVES][NG * 64 * W];          // cross-wave reduction of the output row (32 KiB when VEC4)
  __shared__ float sh_max[GF_WAVES][GF_MAXH];
  __shared__ unsigned long long sh_sum[GF_MAXH];
  __shared__ int sh_bad;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int row = blockIdx.x;
  int S = p.n_dst;
  if (p.n_dst_dev) { const int t = *p.n_dst_dev; S = t < S ? t : S; }
  const int H = HG ? HG : p.H, D = p.D, HD = H * D;
  const uint32_t ctr = p.drop_thresh ? (uint32_t)__hip_atomic_load(p.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
  if (row == 0 && tid == 0 && p.drop_thresh && p.ctr_used) *p.ctr_used = ctr;
  if (row >= S) {                                      // capacity padding: finite zeros (and its ticket for the dropout counter)
    for (int c = tid; c < HD; c += GF_TPB) p.rst[(long long)row * p.rst_stride + c] = 0;
    return;
  }
  const int beg = p.indptr[row], end = p.indptr[row + 1];
  // this lane's columns: group c covers columns c*64*W + lane*W .. +W-1 (one head per group when VEC4: D % 4 == 0)
  float er[NG][W], at[NG][W], acc[NG][W];
  int hd[NG], coff[NG];
#pragma unroll
  for (int c = 0; c < NG; ++c) {
    const int col = c * 64 * W + lane * W;
    hd[c] = col < HD ? col / D : -1;
    coff[c] = col < HD ? col : 0;                       // (masked columns re-read the row's first W elements; at = 0 there)
    const g4f x = col < HD ? ldrow<VEC4>(p.feat + (long long)row * p.feat_stride + col) : g4f{{0.f, 0.f, 0.f, 0.f}};
    const g4f t = col < HD ? ldrow<VEC4>(p.attn + col) : g4f{{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int j = 0; j < W; ++j) { er[c][j] = x.v[j]; at[c][j] = t.v[j]; acc[c][j] = 0.f; }
  }
  if (tid < GF_MAXH) sh_sum[tid] = 0ull;
  if (tid == 0) sh_bad = 0;
  // ---- pass 1: logits (model.py:82-86, op by op like k_gat_edge_dot<0>) and the per-head maximum
  float mx[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) mx[h] = -__builtin_inff();
  auto logits_of = [&](const float (&x)[NG][W], int e) {
    float part[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) part[h] = 0.f;
#pragma unroll
    for (int c = 0; c < NG; ++c) {
      if (HG || hd[c] >= 0) {
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < W; ++j) v += rbf_hw(at[c][j] * rbf_hw(lrelu_f(rbf_hw(x[c][j] + er[c][j]), p.slope)));
        if (HG) part[c % NH] += v;
        else {
          if (HG) part[c % NH] += v;
          else {
#pragma unroll
            for (int h = 0; h < NH; ++h) if (h == hd[c]) part[h] += v;
          }
        }
      }
    }
    wave_sum_heads(part, H);
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      if (h < H) {
        const bf16_t eb = f2bf_hw(part[h]);
        if (lane == h) p.e[(long long)e * H + h] = eb;
        mx[h] = fmaxf(mx[h], bf2f(eb));
      }
    }
  };
  for (int base = beg; base < end; base += 64 * GF_WAVES) {
    const int my_e = base + lane * GF_WAVES + wave;
    const int my_s = my_e < end ? p.src[my_e] : 0;
    const int left = end - base - wave;
    const int n_mine = left <= 0 ? 0 : min(64, (left + GF_WAVES - 1) / GF_WAVES);
    for (int j = 0; j < n_mine; j += GF_FLIGHT) {
      RawGroup<VEC4> raw[GF_FLIGHT][NG];
#pragma unroll
      for (int q = 0; q < GF_FLIGHT; ++q)               // (edges beyond the wave's share re-read edge j: discarded below)
        ld_edge_raw<VEC4, NG>(raw[q], p.feat + (long long)__builtin_amdgcn_readlane(my_s, j + q < n_mine ? j + q : j) * p.feat_stride, coff);
#pragma unroll
      for (int q = 0; q < GF_FLIGHT; ++q) {
        if (j + q < n_mine) {                           // (wave-uniform)
          float x[NG][W];
          unpack_row<VEC4, W, NG>(x, raw[q]);
          logits_of(x, base + (j + q) * GF_WAVES + wave);
        }
      }
    }
  }
  if (lane < NH) {
    float m = -__builtin_inff();
#pragma unroll
    for (int h = 0; h < NH; ++h) if (h == lane) m = mx[h];
    sh_max[wave][lane] = m;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's logits have left for L2 (write-through) before the barrier
  __syncthreads();
  // ---- pass 2: edge softmax over the stored logits (model.py:88-90; [DGL-recalled] four bf16 ops, exact sum), attention dropout
  const int cnt = (end - beg) * H;
  int bad = 0;
  for (int i = tid; i < cnt; i += GF_TPB) {
    const int h = i % H;
    float m = sh_max[0][h];
#pragma unroll
    for (int w2 = 1; w2 < GF_WAVES; ++w2) m = fmaxf(m, sh_max[w2][h]);
    const bf16_t x = __hip_atomic_load(p.e + (long long)beg * H + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bf16_t sc = f2bf((float)exp((double)rbf(bf2f(x) - m)));
    p.a[(long long)beg * H + i] = sc;                  // (this thread reads it back below)
    const long long fx = bf_to_fixed(sc, FRAC_DST, &bad);
    if (fx) atomicAdd(&sh_sum[h], (unsigned long long)fx);
  }
  if (bad) atomicOr(&sh_bad, bad);
  __syncthreads();
  for (int i = tid; i < cnt; i += GF_TPB) {
    const int h = i % H;
    int b2 = 0;
    float ssum = bf2f(fixed_to_bf((long long)sh_sum[h], FRAC_DST, &b2));
    if (sh_bad) ssum = __builtin_nanf("");             // a non-finite logit: the reference's sum, and with it the row, is NaN
    const long long o = (long long)beg * H + i;
    const bf16_t av = f2bf(bf2f(p.a[o]) / ssum);
    p.a[o] = av;
    if (p.drop_thresh) {                               // nn.Dropout on a bf16 tensor: a * mask / (1 - p), one rounding
      const bool keep = gf_drop_hash(p.seed, ctr, (uint32_t)o) >= p.drop_thresh;
      p.ad[o] = keep ? f2bf(bf2f(av) * p.drop_scale) : (bf16_t)0;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // ---- pass 3: out = sum_j a_ij el_j (model.py:98), fp32 products and sums, one rounding
  const bf16_t* aw = p.drop_thresh ? p.ad : p.a;
  for (int base = beg; base < end; base += 64 * GF_WAVES) {
    const int my_e = base + lane * GF_WAVES + wave;
    const int my_s = my_e < end ? p.src[my_e] : 0;
    float my_a[NH];
    ld_edge_coefs(my_a, aw, my_e, H, my_e < end);
    const int left = end - base - wave;
    const int n_mine = left <= 0 ? 0 : min(64, (left + GF_WAVES - 1) / GF_WAVES);
    auto accumulate = [&](const float (&x)[NG][W], int j) {
      float sv[NH];
#pragma unroll
      for (int h = 0; h < NH; ++h) sv[h] = h < H ? bcast_f32(my_a[h], j) : 0.f;
#pragma unroll
      for (int c = 0; c < NG; ++c) {
        if (HG || hd[c] >= 0) {
          const float cf = (HG ? sv[c % NH] : pick_head<NH>(sv, hd[c]));
#pragma unroll
          for (int jj = 0; jj < W; ++jj) acc[c][jj] += cf * x[c][jj];
        }
      }
    };
    for (int j = 0; j < n_mine; j += GF_FLIGHT) {
      RawGroup<VEC4> raw[GF_FLIGHT][NG];
#pragma unroll
      for (int q = 0; q < GF_FLIGHT; ++q)               // (edges beyond the wave's share re-read edge j: discarded below)
        ld_edge_raw<VEC4, NG>(raw[q], p.feat + (long long)__builtin_amdgcn_readlane(my_s, j + q < n_mine ? j + q : j) * p.feat_stride, coff);
#pragma unroll
      for (int q = 0; q < GF_FLIGHT; ++q) {
        if (j + q < n_mine) {                           // (wave-uniform)
          float x[NG][W];
          unpack_row<VEC4, W, NG>(x, raw[q]);
          accumulate(x, j + q);
        }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < NG; ++c)
#pragma unroll
    for (int j = 0; j < W; ++j) sh_acc[wave][c * 64 * W + lane * W + j] = acc[c][j];
  __syncthreads();
  for (int col = tid; col < HD; col += GF_TPB) {
    float s = sh_acc[0][col];
#pragma unroll
    for (int w2 = 1; w2 < GF_WAVES; ++w2) s += sh_acc[w2][col];   // fixed order: bitwise reproducible
    p.rst[(long long)row * p.rst_stride + col] = f2bf(s);
  }
}

// the dropout stream's launch counter moves on behind the forward kernel (every workgroup has read it by then).  A ticket taken
// by each of the ~5 K workgroups on the counter's own cache line cost 40-70 us per launch (scratch/gatbench.py): every
// workgroup's first load queued behind the other workgroups' atomics
__global__ void k_gat_bump(unsigned long long* ctr) { if (threadIdx.x == 0) *ctr += 1ull; }

// backward by destination: d a, softmax backward, d er and the row's share of d attn
template <bool VEC4, int HG>
__global__ void __launch_bounds__(GF_TPB, 4) k_gat_bwd_dst(GatFused p) {
  constexpr int W = VEC4 ? 4 : 1;
  // HG > 0: "a column group is a head" (VEC4, D == 256: group c of 64 lanes x 4 columns IS head c, H == HG <= 4) -- the Reddit
  // config's 4 x 256; every per-head selection below is then a compile-time index.  HG == 0: any H <= 8, D (head by compare)
  constexpr int NG = HG ? HG : GF_ITER, NH = HG ? HG : GF_MAXH;
  __shared__ float sh_acc[GF_WAVES][NG * 64 * W];
  __shared__ float sh_t[GF_WAVES][GF_MAXH];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int row = blockIdx.x;
  int S = p.n_dst;
  if (p.n_dst_dev) { const int t = *p.n_dst_dev; S = t < S ? t : S; }
  const int H = HG ? HG : p.H, D = p.D, HD = H * D;
  if (row >= p.n_dst) return;
  if (row >= S) {
    for (int c = tid; c < HD; c += GF_TPB) { p.d_er[(long long)row * p.der_stride + c] = 0; p.dattn_part[(long long)row * HD + c] = 0.f; }
    return;
  }
  const int beg = p.indptr[row], end = p.indptr[row + 1];
  float er[NG][W], at[NG][W], gr[NG][W];
  int hd[NG], coff[NG];
#pragma unroll
  for (int c = 0; c < NG; ++c) {
    const int col = c * 64 * W + lane * W;
    hd[c] = col < HD ? col / D : -1;
    coff[c] = col < HD ? col : 0;
    const g4f z = g4f{{0.f, 0.f, 0.f, 0.f}};
    const g4f x = col < HD ? ldrow<VEC4>(p.feat + (long long)row * p.feat_stride + col) : z;
    const g4f t = col < HD ? ldrow<VEC4>(p.attn + col) : z;
    const g4f gg = col < HD ? ldrow<VEC4>(p.g + (long long)row * p.g_stride + col) : z;
#pragma unroll
    for (int j = 0; j < W; ++j) { er[c][j] = x.v[j]; at[c][j] = t.v[j]; gr[c][j] = gg.v[j]; }
  }
  // ---- pass 1: d a_ij[h] = g_i[h,:] . el_j[h,:] (bf16, like k_gat_edge_dot<1>), through the dropout mask; t[h] = sum a d a
  float tp[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) tp[h] = 0.f;
  for (int base = beg; base < end; base += 64 * GF_WAVES) {
    const int my_e = base + lane * GF_WAVES + wave;
    const int my_s = my_e < end ? p.src[my_e] : 0;
    float my_a[NH], my_ad[NH];
    ld_edge_coefs(my_a, p.a, my_e, H, my_e < end);
    if (p.drop_thresh) ld_edge_coefs(my_ad, p.ad, my_e, H, my_e < end);
    const int left = end - base - wave;
    const int n_mine = left <= 0 ? 0 : min(64, (left + GF_WAVES - 1) / GF_WAVES);
    auto da_of = [&](const float (&x)[NG][W], int j) {
      const int e = base + j * GF_WAVES + wave;
      float part[NH];
#pragma unroll
      for (int h = 0; h < NH; ++h) part[h] = 0.f;
#pragma unroll
      for (int c = 0; c < NG; ++c) {
        if (HG || hd[c] >= 0) {
          float v = 0.f;
#pragma unroll
          for (int jj = 0; jj < W; ++jj) v += gr[c][jj] * x[c][jj];
          if (HG) part[c % NH] += v;
          else {
#pragma unroll
            for (int h = 0; h < NH; ++h) if (h == hd[c]) part[h] += v;
          }
        }
      }
      wave_sum_heads(part, H);
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        if (h < H) {
          float da = rbf_hw(part[h]);
          if (p.drop_thresh) da = (bcast_f32(my_ad[h], j) != 0.f) ? rbf_hw(da * p.drop_scale) : 0.f;   // dropout backward (mask = what the forward kept)
          if (lane == h) p.de[(long long)e * H + h] = f2bf_hw(da);
          tp[h] += bcast_f32(my_a[h], j) * da;
        }
      }
    };
    for (int j = 0; j < n_mine; j += GF_FLIGHT) {
      RawGroup<VEC4> raw[GF_FLIGHT][NG];
#pragma unroll
      for (int q = 0; q < GF_FLIGHT; ++q)               // (edges beyond the wave's share re-read edge j: discarded below)
        ld_edge_raw<VEC4, NG>(raw[q], p.feat + (long long)__builtin_amdgcn_readlane(my_s, j + q < n_mine ? j + q : j) * p.feat_stride, coff);
#pragma unroll
      for (int q = 0; q < GF_FLIGHT; ++q) {
        if (j + q < n_mine) {                           // (wave-uniform)
          float x[NG][W];
          unpack_row<VEC4, W, NG>(x, raw[q]);
          da_of(x, j + q);
        }
      }
    }
  }
  if (lane < NH) {
    float t = 0.f;
#pragma unroll
    for (int h = 0; h < NH; ++h) if (h == lane) t = tp[h];
    sh_t[wave][lane] = t;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // ---- pass 2: d e = a (d a - t)   (k_gat_softmax<true>)
  const int cnt = (end - beg) * H;
  for (int i = tid; i < cnt; i += GF_TPB) {
    const int h = i % H;
    float t = sh_t[0][h];
#pragma unroll
    for (int w2 = 1; w2 < GF_WAVES; ++w2) t += sh_t[w2][h];
    const long long o = (long long)beg * H + i;
    const float da = bf2f(__hip_atomic_load(p.de + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    p.de[o] = f2bf(bf2f(p.a[o]) * (da - t));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // ---- pass 3: d er_i = sum_j d e attn lrelu'(el_j + er_i);  d attn += d e lrelu(el_j + er_i)   (k_gat_rows<true, false>)
  float dacc[NG][W], aacc[NG][W];
#pragma unroll
  for (int c = 0; c < NG; ++c)
#pragma unroll
    for (int j = 0; j < W; ++j) { dacc[c][j] = 0.f; aacc[c][j] = 0.f; }
  for (int base = beg; base < end; base += 64 * GF_WAVES) {
    const int my_e = base + lane * GF_WAVES + wave;
    const int my_s = my_e < end ? p.src[my_e] : 0;
    float my_de[NH];
    ld_edge_coefs(my_de, p.de, my_e, H, my_e < end);
    const int left = end - base - wave;
    const int n_mine = left <= 0 ? 0 : min(64, (left + GF_WAVES - 1) / GF_WAVES);
    auto grads_of = [&](const float (&x)[NG][W], int j) {
      float sv[NH];
#pragma unroll
      for (int h = 0; h < NH; ++h) sv[h] = h < H ? bcast_f32(my_de[h], j) : 0.f;
#pragma unroll
      for (int c = 0; c < NG; ++c) {
        if (HG || hd[c] >= 0) {
          const float cf = (HG ? sv[c % NH] : pick_head<NH>(sv, hd[c]));
#pragma unroll
          for (int jj = 0; jj < W; ++jj) {
            const float sx = x[c][jj] + er[c][jj];
            dacc[c][jj] += cf * at[c][jj] * (sx > 0.f ? 1.f : p.slope);
            aacc[c][jj] += cf * lrelu_f(sx, p.slope);
          }
        }
      }
    };
    for (int j = 0; j < n_mine; j += GF_FLIGHT) {
      RawGroup<VEC4> raw[GF_FLIGHT][NG];
#pragma unroll
      for (int q = 0; q < GF_FLIGHT; ++q)               // (edges beyond the wave's share re-read edge j: discarded below)
        ld_edge_raw<VEC4, NG>(raw[q], p.feat + (long long)__builtin_amdgcn_readlane(my_s, j + q < n_mine ? j + q : j) * p.feat_stride, coff);
#pragma unroll
      for (int q = 0; q < GF_FLIGHT; ++q) {
        if (j + q < n_mine) {                           // (wave-uniform)
          float x[NG][W];
          unpack_row<VEC4, W, NG>(x, raw[q]);
          grads_of(x, j + q);
        }
      }
    }
  }
  // two cross-wave reductions through the same LDS buffer
#pragma unroll
  for (int c = 0; c < NG; ++c)
#pragma unroll
    for (int j = 0; j < W; ++j) sh_acc[wave][c * 64 * W + lane * W + j] = dacc[c][j];
  __syncthreads();
  for (int col = tid; col < HD; col += GF_TPB) {
    float s = sh_acc[0][col];
#pragma unroll
    for (int w2 = 1; w2 < GF_WAVES; ++w2) s += sh_acc[w2][col];
    p.d_er[(long long)row * p.der_stride + col] = f2bf(s);
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < NG; ++c)
#pragma unroll
    for (int j = 0; j < W; ++j) sh_acc[wave][c * 64 * W + lane * W + j] = aacc[c][j];
  __syncthreads();
  for (int col = tid; col < HD; col += GF_TPB) {
    float s = sh_acc[0][col];
#pragma unroll
    for (int w2 = 1; w2 < GF_WAVES; ++w2) s += sh_acc[w2][col];
    p.dattn_part[(long long)row * HD + col] = s;
  }
}

// d attn[col] = sum over the destination rows of their shares, in a fixed order (deterministic, no float atomics): stage 1 sums
// blocks of DA_ROWS rows (one workgroup per block and 256 columns, 8 independent loads in flight per thread), stage 2 the block
// sums (64 columns per workgroup, four row groups combined through LDS in group order)
#define DA_ROWS 32
__global__ void __launch_bounds__(256) k_gat_dattn_stage1(const float* __restrict__ part, int n_rows, const int* __restrict__ n_rows_dev,
                                                          int HD, float* __restrict__ blocks) {
  int S = n_rows;
  if (n_rows_dev) { const int t = *n_rows_dev; S = t < S ? t : S; }
  const int col = blockIdx.y * 256 + threadIdx.x, r0 = blockIdx.x * DA_ROWS, r1 = min(S, r0 + DA_ROWS);
  if (col >= HD) return;
  float s = 0.f;
  int r = r0;
  for (; r + 8 <= r1; r += 8) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = part[(long long)(r + i) * HD + col];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
  }
  for (; r < r1; ++r) s += part[(long long)r * HD + col];
  blocks[(long long)blockIdx.x * HD + col] = s;          // (blocks beyond the true row count hold 0)
}
__global__ void __launch_bounds__(256) k_gat_dattn_stage2(const float* __restrict__ blocks, int nb, int HD, float* __restrict__ d_attn) {
  __shared__ float sh[4][64];
  const int c = threadIdx.x & 63, grp = threadIdx.x >> 6, col = blockIdx.x * 64 + c;
  float s = 0.f;
  if (col < HD) for (int b = grp; b < nb; b += 4) s += blocks[(long long)b * HD + col];
  sh[grp][c] = s;
  __syncthreads();
  if (grp == 0 && col < HD) d_attn[col] = ((sh[0][c] + sh[1][c]) + sh[2][c]) + sh[3][c];
}

bool gf_fill(const bliss_gat_fused_t* a, GatFused* p, bool* vec4) {
  if (!a || !a->indptr || !a->feat || !a->attn || a->heads <= 0 || a->heads > GF_MAXH || a->head_dim <= 0 || a->n_dst <= 0) return false;
  const int HD = a->heads * a->head_dim;
  const bool v4 = a->head_dim % 4 == 0 && a->feat_stride % 4 == 0 && ((uintptr_t)a->feat) % 8 == 0 && ((uintptr_t)a->attn) % 8 == 0 &&
                  (!a->g || (a->g_stride % 4 == 0 && ((uintptr_t)a->g) % 8 == 0));
  if (HD > GF_ITER * 64 * (v4 ? 4 : 1)) return false;
  p->indptr = a->indptr; p->src = a->src; p->n_dst = a->n_dst; p->n_dst_dev = a->n_dst_dev;
  p->feat = (const bf16_t*)a->feat; p->feat_stride = a->feat_stride; p->attn = (const bf16_t*)a->attn; p->H = a->heads; p->D = a->head_dim;
  p->slope = a->negative_slope;
  p->e = (bf16_t*)a->e; p->a = (bf16_t*)a->a; p->ad = (bf16_t*)a->a_drop;
  p->rst = (bf16_t*)a->rst; p->rst_stride = a->rst_stride;
  p->drop_thresh = a->drop_p > 0.f ? (unsigned)((double)a->drop_p * 4294967296.0) : 0u;
  p->drop_scale = a->drop_p > 0.f ? 1.0f / (1.0f - a->drop_p) : 1.0f;
  p->seed = a->drop_seed; p->ctr = (unsigned long long*)a->drop_ctr; p->ctr_used = (unsigned*)a->drop_ctr_used;
  p->g = (const bf16_t*)a->g; p->g_stride = a->g_stride; p->de = (bf16_t*)a->de;
  p->d_er = (bf16_t*)a->d_er; p->der_stride = a->d_er_stride; p->dattn_part = a->dattn_part;
  *vec4 = v4;
  return true;
}

}  // namespace

extern "C" {

int bliss_gat_fused_supported(int32_t heads, int32_t head_dim) {
  if (heads <= 0 || heads > GF_MAXH || head_dim <= 0) return 0;
  return heads * head_dim <= GF_ITER * 64 * (head_dim % 4 == 0 ? 4 : 1);
}

int bliss_gat_fused_fwd(const bliss_gat_fused_t* args, void* stream) {
  GatFused p;
  bool v4;
  if (!gf_fill(args, &p, &v4) || !p.src || !p.e || !p.a || !p.rst) return BLISS_EINVAL;
  if (p.drop_thresh && (!p.ad || !p.ctr)) return BLISS_EINVAL;
  if (args->drop_p < 0.f || args->drop_p >= 1.f) return BLISS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int hg = (v4 && p.D == 256 && (p.H == 1 || p.H == 2 || p.H == 4)) ? p.H : 0;
  if (hg == 4) k_gat_fwd<true, 4><<<p.n_dst, GF_TPB, 0, st>>>(p);
  else if (hg == 2) k_gat_fwd<true, 2><<<p.n_dst, GF_TPB, 0, st>>>(p);
  else if (hg == 1) k_gat_fwd<true, 1><<<p.n_dst, GF_TPB, 0, st>>>(p);
  else if (v4) k_gat_fwd<true, 0><<<p.n_dst, GF_TPB, 0, st>>>(p);
  else k_gat_fwd<false, 0><<<p.n_dst, GF_TPB, 0, st>>>(p);
  if (p.drop_thresh) k_gat_bump<<<1, 64, 0, st>>>(p.ctr);
  return (int)hipGetLastError();
}

int bliss_gat_fused_bwd_dst(const bliss_gat_fused_t* args, float* block_sums, float* d_attn, uint32_t* ticket, void* stream) {
  GatFused p;
  bool v4;
  if (!gf_fill(args, &p, &v4) || !p.src || !p.a || !p.g || !p.de || !p.d_er || !p.dattn_part || !block_sums || !d_attn || !ticket) return BLISS_EINVAL;
  if (p.drop_thresh && !p.ad) return BLISS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int hg = (v4 && p.D == 256 && (p.H == 1 || p.H == 2 || p.H == 4)) ? p.H : 0;
  if (hg == 4) k_gat_bwd_dst<true, 4><<<p.n_dst, GF_TPB, 0, st>>>(p);
  else if (hg == 2) k_gat_bwd_dst<true, 2><<<p.n_dst, GF_TPB, 0, st>>>(p);
  else if (hg == 1) k_gat_bwd_dst<true, 1><<<p.n_dst, GF_TPB, 0, st>>>(p);
  else if (v4) k_gat_bwd_dst<true, 0><<<p.n_dst, GF_TPB, 0, st>>>(p);
  else k_gat_bwd_dst<false, 0><<<p.n_dst, GF_TPB, 0, st>>>(p);
  const int nb = (p.n_dst + DA_ROWS - 1) / DA_ROWS, HD = p.H * p.D;
  k_gat_dattn_stage1<<<dim3(nb, (HD + 255) / 256), 256, 0, st>>>(p.dattn_part, p.n_dst, p.n_dst_dev, HD, block_sums);
  k_gat_dattn_stage2<<<(HD + 63) / 64, 256, 0, st>>>(block_sums, nb, HD, d_attn);
  (void)ticket;
  return (int)hipGetLastError();
}

}  // extern "C"
