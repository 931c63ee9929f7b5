// The dense transforms of a SAGE layer on the matrix cores, fused with what surrounds them.
//
// model.py:321-333 -> dglnn.SAGEConv('mean') [DGL-recalled, SURVEY.md m2-m5]:  rst = fc_self(h_dst) + fc_neigh o mean_w(h)
// with fc_neigh BEFORE the aggregation iff in > out, then relu + dropout, and model.py:318-320 takes the row norms of
// every layer's input.  The reference (and round 1 here) runs each Linear as a library GEMM with its own launch, plus a
// gather kernel in front and an element-wise epilogue behind; at these sizes (a few thousand rows) every launch costs
// more than its arithmetic.  k_tile_gemm is ONE launch for
//     out[r, :] = epilogue( A1[r, :] . W1^T  (+ A2[r, :] . W2^T)  + bias )
// where the rows of A1 may be gathered through an index (blocks[0].srcdata['features'] = features[input_nodes],
// train_lightning.py:138 -- the gather IS the A-operand load), the epilogue is bf16 rounding, optional ReLU, optional
// dropout, and the kernel also leaves the row norms of its INPUT rows (embed_norm of this layer) and of its OUTPUT rows
// (embed_norm of the next layer).  Two argument sets can share a launch (blockIdx.y): fc_neigh over all source rows and
// fc_self over the destination rows of a W-first layer.
//
// Shape of the work: a workgroup (8 waves) owns 32 rows and all N <= 256 output columns.  The 32 input rows are staged
// ONCE in LDS (32 x K bf16 <= 66 KB for K <= 1024, 39 KB for the 602-wide features: four workgroups per CU overlap each
// other's latency chains), which is also where the input norms are taken in exactly k_embed_norm's order (same bits as the
// unfused path).  Wave w computes columns 32 w .. 32 w + 31 as one 32 x 32 tile of v_mfma_f32_32x32x16_bf16: A fragments are 16-byte LDS reads (row = lane & 31, k = 8 (lane >> 5) ..+7), B fragments
// 16-byte global (L2) reads of W[n][k..k+7] -- nn.Linear keeps W as [out, in], i.e. K-contiguous, exactly the B layout
// the instruction wants, so no operand is ever transposed.  fp32 accumulation over all of K (and both products), one
// rounding to bf16 at the store.  A row's result depends on nothing but that row: capacity-padded and exact-size blocks
// give identical bits.
#include "common.cuh"
#include "bliss_gnn.h"
#include <cstdlib>

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

#define TG_TPB 512
#define TG_M 32
#define TG_WAVES (TG_TPB / 64)
#define TG_NB (256 / 32 / TG_WAVES)          // 32-column blocks per wave: 8 waves x 1 block cover the 256 columns

struct TileGemm {
  const bf16_t* a1; long long a1_stride; const int* ids;
  const bf16_t* w1; long long w1_stride; int k1;
  const bf16_t* a2; long long a2_stride; const bf16_t* w2; long long w2_stride; int k2;
  const bf16_t* bias;
  int m_bound; const int* m_dev; int n;
  bf16_t* out; long long out_stride;
  bf16_t* a_copy; long long copy_stride;
  bf16_t* in_norm; bf16_t* out_norm;
  int relu; unsigned drop_thresh; float drop_scale; unsigned seed; unsigned long long* ctr;
};

__device__ __forceinline__ int k_pad16(int k) { return (k + 15) & ~15; }
// LDS row stride (elements): K rounded to 16, plus 8: the stride is then 4 (mod 8) dwords, so the 16-byte fragment reads
// of 8 consecutive rows fall into 8 different 4-bank groups (and a 64-row tile of 602-wide rows stays below 80 KB: two
// workgroups per CU)
__device__ __host__ __forceinline__ int lds_stride(int k) {
  return ((k + 15) & ~15) + 8;
}

__device__ __forceinline__ uint32_t tg_drop_hash(uint32_t seed, uint32_t ctr, uint32_t idx) {   // = drop_hash of spmm.hip
  uint32_t x = idx * 0x9e3779b1u + seed;
  x ^= ctr * 0x85ebca77u + 0x165667b1u;
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  x += ctr; x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12;
  return x;
}

// sum of squares of one LDS row in k_embed_norm's order (spmm.hip): vec4 = the 4-columns-per-lane walk, else 1 per lane
__device__ __forceinline__ float row_sumsq(const bf16_t* sh, int dim, int vec4, int lane) {
  float s = 0.f;
  if (vec4) {
    for (int c = lane * 4; c < dim; c += 256) {
      const float a = bf2f(sh[c]), b = bf2f(sh[c + 1]), cc = bf2f(sh[c + 2]), d = bf2f(sh[c + 3]);
      s += a * a; s += b * b; s += cc * cc; s += d * d;
    }
  } else {
    for (int c = lane; c < dim; c += 64) { const float a = bf2f(sh[c]); s += a * a; }
  }
  for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
  return s;
}

// bf16 norms of this wave's rows of an LDS tile, each in k_embed_norm's summation order (row_sumsq), the rows side by side:
// independent accumulators, so the LDS reads and the butterfly shuffles of the 8 rows overlap instead of queueing
__device__ __forceinline__ void rows_norm(const bf16_t* tile, int stride, int dim, int row0, int M, int m_bound, bf16_t* __restrict__ out,
                                          int wave, int lane) {
  constexpr int RPW = TG_M / TG_WAVES;
  float s[RPW];
#pragma unroll
  for (int q = 0; q < RPW; ++q) s[q] = 0.f;
  if (dim % 4 == 0) {
    for (int c = lane * 4; c < dim; c += 256)
#pragma unroll
      for (int q = 0; q < RPW; ++q) {
        const bf16_t* sh = tile + (size_t)(q * TG_WAVES + wave) * stride + c;
        const float a = bf2f(sh[0]), b = bf2f(sh[1]), cc = bf2f(sh[2]), d = bf2f(sh[3]);
        s[q] += a * a; s[q] += b * b; s[q] += cc * cc; s[q] += d * d;
      }
  } else {
    for (int c = lane; c < dim; c += 64)
#pragma unroll
      for (int q = 0; q < RPW; ++q) { const float a = bf2f(tile[(size_t)(q * TG_WAVES + wave) * stride + c]); s[q] += a * a; }
  }
  for (int d = 32; d >= 1; d >>= 1)
#pragma unroll
    for (int q = 0; q < RPW; ++q) s[q] += __shfl_xor(s[q], d);
#pragma unroll
  for (int q = 0; q < RPW; ++q) {
    const int r = row0 + q * TG_WAVES + wave;
    if (lane == 0 && r < m_bound) out[r] = r < M ? f2bf(sqrtf(s[q])) : (bf16_t)0;
  }
}

// Stage the 64 rows of one operand in LDS.  The loads of all rows are independent and issued together: wave w takes rows
// w, w + 8, ... (8 rounds), a lane the dwords lane, lane + 64, ... of its row, so ~40 loads per lane are in flight before
// the first LDS store -- the gather is latency-bound (a random 1.2 KB row per id), not bandwidth-bound.  row_id[lr] = the
// source row of tile row lr or -1 (beyond the true row count: zeros).  Optional copy-out of the staged rows.
__device__ __forceinline__ void stage_rows(const bf16_t* __restrict__ a, long long a_stride, const int* row_id, int k,
                                           int row0, int m_bound, bf16_t* tile, int stride, bf16_t* __restrict__ copy, long long copy_stride,
                                           int wave, int lane) {
  const int kp = k_pad16(k);
  const bool even = (k % 2 == 0) && (a_stride % 2 == 0) && (((uintptr_t)a) % 4 == 0);
  const bool copy32 = copy && (copy_stride % 2 == 0) && (((uintptr_t)copy) % 4 == 0);
  if (even) {
    const int dw = k / 2;
    constexpr int MAXC = (1024 / 2 + 63) / 64;          // dwords per lane and row at the largest K
    constexpr int RPW = TG_M / TG_WAVES, HALF = RPW;      // rows per wave, all in flight together (LDS, not registers, bounds the occupancy)
#pragma unroll
    for (int hb = 0; hb < 1; ++hb) {
      uint32_t v[HALF][MAXC];
#pragma unroll
      for (int q = 0; q < HALF; ++q) {
        const int id = row_id[(hb * HALF + q) * TG_WAVES + wave];
        const uint32_t* src = reinterpret_cast<const uint32_t*>(a + (long long)(id < 0 ? 0 : id) * a_stride);
#pragma unroll
        for (int j = 0; j < MAXC; ++j) {
          // unconditional loads from clamped (always valid) addresses, masked afterwards: a load under a branch costs a full
          // wait for everything outstanding, and these are meant to be in flight together
          const int c = lane + 64 * j;
          const uint32_t x = src[c < dw ? c : dw - 1];
          v[q][j] = (id >= 0 && c < dw) ? x : 0u;
        }
      }
#pragma unroll
      for (int q = 0; q < HALF; ++q) {
        const int lr = (hb * HALF + q) * TG_WAVES + wave, r = row0 + lr;
        uint32_t* sh = reinterpret_cast<uint32_t*>(tile + (size_t)lr * stride);
#pragma unroll
        for (int j = 0; j < MAXC; ++j) {
          const int c = lane + 64 * j;
          if (c < kp / 2) sh[c] = v[q][j];               // (c >= dw: the zero padding up to the 16-multiple)
          if (copy && r < m_bound && c < dw) {
            if (copy32) reinterpret_cast<uint32_t*>(copy + (long long)r * copy_stride)[c] = v[q][j];
            else { bf16_t* cp = copy + (long long)r * copy_stride + 2 * c; cp[0] = (bf16_t)(v[q][j] & 0xffffu); cp[1] = (bf16_t)(v[q][j] >> 16); }
          }
        }
      }
    }
  } else {
    for (int rd = 0; rd < TG_M / TG_WAVES; ++rd) {
      const int lr = rd * TG_WAVES + wave, r = row0 + lr, id = row_id[lr];
      const bf16_t* src = a + (long long)(id < 0 ? 0 : id) * a_stride;
      bf16_t* sh = tile + (size_t)lr * stride;
      for (int c = lane; c < kp; c += 64) {
        const bf16_t x = (id >= 0 && c < k) ? src[c] : (bf16_t)0;
        sh[c] = x;
        if (copy && r < m_bound && c < k) copy[(long long)r * copy_stride + c] = x;
      }
    }
  }
}

// W through LDS in slabs of 64 k: an MFMA B fragment is 16 bytes of ONE row of W per lane, so loading fragments straight
// from global memory touches 32 different cache lines per wave instruction and uses a quarter of each (measured: the
// texture path, not the matrix core, bounded the first version of this kernel at ~45 us per launch).  Staged through LDS,
// 8 consecutive threads read one row's 128 contiguous bytes; the next slab is in flight (registers) while the current one
// is multiplied.
#define TG_SLAB 64
#define TG_WSTRIDE (TG_SLAB + 8)
#define TG_WROWS 256
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

struct WRegs { uint4 v[TG_WROWS * TG_SLAB / 8 / TG_TPB]; };      // 8 x 16 bytes per thread

// full slab (all 64 k inside K, rows 4-byte aligned): no branch, no mask -- rows beyond N are CLAMPED to a real row (their
// columns are computed from it and never stored), so every lane issues its eight 16-byte loads unconditionally and the
// compiler can count them (a load under a branch made it wait for ALL outstanding loads at every use)
__device__ __forceinline__ void w_slab_load_full(WRegs& g, const bf16_t* __restrict__ w, long long w_stride, int N, int slab, int tid) {
  const int part = tid & 7, k = slab * TG_SLAB + 8 * part;
#pragma unroll
  for (int i = 0; i < (int)(sizeof(g.v) / sizeof(g.v[0])); ++i) {
    int n = i * (TG_TPB / 8) + (tid >> 3);
    n = n < N ? n : N - 1;
    const u32x4_a4 q = *reinterpret_cast<const u32x4_a4*>(w + (long long)n * w_stride + k);
    g.v[i] = make_uint4(q[0], q[1], q[2], q[3]);
  }
}
// the last, partial slab (or odd alignment): element-wise with masks
__device__ __forceinline__ void w_slab_load_tail(WRegs& g, const bf16_t* __restrict__ w, long long w_stride, int N, int K, int slab, int tid) {
  const int part = tid & 7, k = slab * TG_SLAB + 8 * part;
  for (int i = 0; i < (int)(sizeof(g.v) / sizeof(g.v[0])); ++i) {
    int n = i * (TG_TPB / 8) + (tid >> 3);
    n = n < N ? n : N - 1;
    const bf16_t* p = w + (long long)n * w_stride + k;
    union { uint4 u; bf16_t e[8]; } t;
    t.u = make_uint4(0, 0, 0, 0);
    for (int j = 0; j < 8; ++j) if (k + j < K) t.e[j] = p[j];
    g.v[i] = t.u;
  }
}
// how many leading slabs of this W take the branch-free path
__device__ __forceinline__ int w_full_slabs(const bf16_t* w, long long w_stride, int K) {
  return ((w_stride % 2 == 0) && (((uintptr_t)w) % 4 == 0)) ? K / TG_SLAB : 0;
}
__device__ __forceinline__ void w_slab_store(const WRegs& g, bf16_t* wl, int tid) {
  const int part = tid & 7;
#pragma unroll
  for (int i = 0; i < (int)(sizeof(g.v) / sizeof(g.v[0])); ++i) {
    const int n = i * (TG_TPB / 8) + (tid >> 3);
    *reinterpret_cast<uint4*>(wl + (size_t)n * TG_WSTRIDE + 8 * part) = g.v[i];
  }
}

// acc += A_tile[32 x K] . W[N x K]^T for this wave's 64 columns (n0 ..); all four waves take part in the slab traffic
__device__ __forceinline__ void mma_product(WRegs& ga, WRegs& gb, const bf16_t* tile, int stride, int k, const bf16_t* __restrict__ w,
                                            long long w_stride, int n0, int N, bf16_t* wl, int tid, int lane, f32x16_t acc[TG_NB]) {
  const int kp = k_pad16(k), r = lane & 31, h = lane >> 5;
  const int nslab = (kp + TG_SLAB - 1) / TG_SLAB;
  auto multiply = [&](int sl) {
    if (n0 < N) {
#pragma unroll
      for (int st = 0; st < TG_SLAB / 16; ++st) {
        const int ks = sl * TG_SLAB + 16 * st;
        if (ks < kp) {
          const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(tile + (size_t)r * stride + ks + 8 * h);
#pragma unroll
          for (int nb = 0; nb < TG_NB; ++nb) {
            const bf16x8_t b = *reinterpret_cast<const bf16x8_t*>(wl + (size_t)(n0 + 32 * nb + r) * TG_WSTRIDE + 16 * st + 8 * h);
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[nb], 0, 0, 0);
          }
        }
      }
    }
  };
  // Two register sets (ga: even slabs, gb: odd), two slabs in flight beside the one being multiplied; the caller has
  // requested slabs 0 and 1.  Only full slabs run through this pipeline: their loads are unconditional, so the compiler's
  // wait before a set's LDS store counts exactly that set's (older) loads.
  const int nfull = w_full_slabs(w, w_stride, k);
  for (int sl = 0; sl < nfull; sl += 2) {
    __syncthreads();                                      // the previous slab has been consumed
    w_slab_store(ga, wl, tid);
    if (sl + 2 < nfull) w_slab_load_full(ga, w, w_stride, N, sl + 2, tid);
    __syncthreads();
    multiply(sl);
    if (sl + 1 < nfull) {
      __syncthreads();
      w_slab_store(gb, wl, tid);
      if (sl + 3 < nfull) w_slab_load_full(gb, w, w_stride, N, sl + 3, tid);
      __syncthreads();
      multiply(sl + 1);
    }
  }
  // the partial last slab (K not a multiple of 64) or an oddly aligned W: one slab at a time, masked element loads
  for (int sl = nfull; sl < nslab; ++sl) {
    WRegs t;
    w_slab_load_tail(t, w, w_stride, N, k, sl, tid);
    __syncthreads();
    w_slab_store(t, wl, tid);
    __syncthreads();
    multiply(sl);
  }
}

__device__ __forceinline__ void tile_gemm_body(const TileGemm& p, bf16_t* lds, int* row_id, int dbg) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  int M = p.m_bound;
  if (p.m_dev) { const int t = *p.m_dev; M = t < M ? t : M; }
  const int row0 = blockIdx.x * TG_M;
  const uint32_t ctr = p.drop_thresh ? (uint32_t)__hip_atomic_load(p.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
  const int N = p.n;
  if (row0 >= p.m_bound) return;                             // (workgroup-uniform: the other argument set has more tiles)
  if (row0 >= M) {
    // a tile of capacity padding: nothing to compute, but what later passes read of these rows must be finite (zero)
    for (int lr = wave; lr < TG_M; lr += TG_WAVES) {
      const int r = row0 + lr;
      if (r >= p.m_bound) break;
      for (int c = lane; c < N; c += 64) p.out[(long long)r * p.out_stride + c] = 0;
      if (p.a_copy) for (int c = lane; c < p.k1; c += 64) p.a_copy[(long long)r * p.copy_stride + c] = 0;
      if (lane == 0) { if (p.in_norm) p.in_norm[r] = 0; if (p.out_norm) p.out_norm[r] = 0; }
    }
    return;
  }
  const int s1 = lds_stride(p.k1), s2 = p.k2 ? lds_stride(p.k2) : 0;
  bf16_t* t1 = lds;
  bf16_t* t2 = lds + (size_t)TG_M * s1;
  const int n0 = wave * 32 * TG_NB;
  bf16_t* wl = lds + (size_t)TG_M * (s1 + s2);              // the W slab behind the staged rows
  float bias_v[TG_NB];
#pragma unroll
  for (int nb = 0; nb < TG_NB; ++nb) bias_v[nb] = 0.f;
#pragma unroll
  for (int nb = 0; nb < TG_NB; ++nb) { const int col = n0 + 32 * nb + (lane & 31); if (p.bias && col < N) bias_v[nb] = bf2f(p.bias[col]); }
  // the first two W slabs depend on nothing: requested now, they arrive while the rows are gathered
  WRegs ga, gb;
  {
    const int nf = w_full_slabs(p.w1, p.w1_stride, p.k1);
    if (nf > 0) w_slab_load_full(ga, p.w1, p.w1_stride, N, 0, tid);
    if (nf > 1) w_slab_load_full(gb, p.w1, p.w1_stride, N, 1, tid);
  }
  if (tid < TG_M) {
    const int r = row0 + tid;
    row_id[tid] = r < M ? (p.ids ? p.ids[r] : r) : -1;
    row_id[TG_M + tid] = r < M ? r : -1;
  }
  __syncthreads();
  if (!(dbg & 1)) stage_rows(p.a1, p.a1_stride, row_id, p.k1, row0, p.m_bound, t1, s1, p.a_copy, p.copy_stride, wave, lane);
  if (p.k2) stage_rows(p.a2, p.a2_stride, row_id + TG_M, p.k2, row0, p.m_bound, t2, s2, nullptr, 0, wave, lane);
  __syncthreads();
  if (p.in_norm && !(dbg & 4)) rows_norm(t1, s1, p.k1, row0, M, p.m_bound, p.in_norm, wave, lane);   // model.py:318-320 of THIS layer
  f32x16_t acc[TG_NB];
#pragma unroll
  for (int m = 0; m < TG_NB; ++m)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;
  if (!(dbg & 2)) mma_product(ga, gb, t1, s1, p.k1, p.w1, p.w1_stride, n0, N, wl, tid, lane, acc);
  if (p.k2) {
    const int nf = w_full_slabs(p.w2, p.w2_stride, p.k2);
    if (nf > 0) w_slab_load_full(ga, p.w2, p.w2_stride, N, 0, tid);
    if (nf > 1) w_slab_load_full(gb, p.w2, p.w2_stride, N, 1, tid);
    mma_product(ga, gb, t2, s2, p.k2, p.w2, p.w2_stride, n0, N, wl, tid, lane, acc);
  }
  __syncthreads();                                          // every wave is done reading the staged rows: reuse the LDS for the output tile
  const int so = ((N + 63) & ~63) + 8;
  bf16_t* ot = lds;
  if (n0 < N) {
#pragma unroll
    for (int m = 0; m < TG_NB; ++m) {                        // (m = the wave's column blocks)
      const int col = n0 + 32 * m + (lane & 31);
      const float bv = bias_v[m];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int lr = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
        float t = rbf(acc[m][i] + bv);                       // fp32 accumulator + bias, ONE rounding to bf16
        if (p.relu) t = t > 0.f ? t : 0.f;
        if (p.drop_thresh) {
          const bool keep = tg_drop_hash(p.seed, ctr, (uint32_t)((row0 + lr) * N + col)) >= p.drop_thresh;
          t = keep ? rbf(t * p.drop_scale) : 0.f;
        }
        ot[(size_t)lr * so + col] = (row0 + lr < M && col < N) ? f2bf(t) : (bf16_t)0;
      }
    }
  }
  __syncthreads();
  const bool pair_ok = (N % 2 == 0) && (p.out_stride % 2 == 0) && (((uintptr_t)p.out) % 4 == 0);
  for (int lr = wave; lr < TG_M; lr += TG_WAVES) {
    const int r = row0 + lr;
    if (r >= p.m_bound) break;
    const bf16_t* sh = ot + (size_t)lr * so;
    bf16_t* o = p.out + (long long)r * p.out_stride;
    if (pair_ok) for (int c = lane * 2; c < N; c += 128) *reinterpret_cast<uint32_t*>(o + c) = *reinterpret_cast<const uint32_t*>(sh + c);
    else for (int c = lane; c < N; c += 64) o[c] = sh[c];
  }
  if (p.out_norm) rows_norm(ot, so, N, row0, M, p.m_bound, p.out_norm, wave, lane);
}

__global__ void __launch_bounds__(TG_TPB) k_tile_gemm(TileGemm p0, TileGemm p1, int n_sets, int dbg) {
  extern __shared__ __attribute__((aligned(16))) bf16_t tg_lds[];
  __shared__ int row_id[2 * TG_M];
  if (dbg & 8) return;
  tile_gemm_body(blockIdx.y == 0 ? p0 : p1, tg_lds, row_id, dbg);
  // dropout stream: the last workgroup of the launch bumps the device-resident launch counter (everybody has read it)
  unsigned long long* ctr = p0.drop_thresh ? p0.ctr : (n_sets > 1 && p1.drop_thresh ? p1.ctr : nullptr);
  if (ctr) {
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned long long total = (unsigned long long)gridDim.x * gridDim.y;
      if (atomicAdd(ctr + 1, 1ull) == total - 1) {
        __hip_atomic_store(ctr + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicAdd(ctr, 1ull);
      }
    }
  }
}

bool convert(const bliss_tile_gemm_t* a, TileGemm* p, size_t* lds_bytes, int* tiles) {
  if (!a->a1 || !a->w1 || !a->out || a->k1 <= 0 || a->k1 > 1024 || a->n <= 0 || a->n > 256 || a->m_bound <= 0) return false;
  if (a->k2 < 0 || a->k2 > 1024 || (a->k2 > 0 && (!a->a2 || !a->w2))) return false;
  if (a->drop_p < 0.f || a->drop_p >= 1.f || (a->drop_p > 0.f && !a->drop_ctr)) return false;
  p->a1 = (const bf16_t*)a->a1; p->a1_stride = a->a1_stride; p->ids = a->ids;
  p->w1 = (const bf16_t*)a->w1; p->w1_stride = a->w1_stride; p->k1 = a->k1;
  p->a2 = (const bf16_t*)a->a2; p->a2_stride = a->a2_stride; p->w2 = (const bf16_t*)a->w2; p->w2_stride = a->w2_stride; p->k2 = a->k2;
  p->bias = (const bf16_t*)a->bias; p->m_bound = a->m_bound; p->m_dev = a->m_dev; p->n = a->n;
  p->out = (bf16_t*)a->out; p->out_stride = a->out_stride; p->a_copy = (bf16_t*)a->a_copy; p->copy_stride = a->copy_stride;
  p->in_norm = (bf16_t*)a->in_norm; p->out_norm = (bf16_t*)a->out_norm; p->relu = a->relu;
  p->drop_thresh = a->drop_p > 0.f ? (unsigned)((double)a->drop_p * 4294967296.0) : 0u;
  p->drop_scale = a->drop_p > 0.f ? 1.0f / (1.0f - a->drop_p) : 1.0f;
  p->seed = a->drop_seed; p->ctr = (unsigned long long*)a->drop_ctr;
  const size_t stage = ((size_t)TG_M * (lds_stride(a->k1) + (a->k2 ? lds_stride(a->k2) : 0)) + (size_t)TG_WROWS * TG_WSTRIDE) * sizeof(bf16_t);
  const size_t outt = (size_t)TG_M * (((a->n + 63) & ~63) + 8) * sizeof(bf16_t);
  *lds_bytes = stage > outt ? stage : outt;
  *tiles = (a->m_bound + TG_M - 1) / TG_M;
  return true;
}

}  // namespace

extern "C" int bliss_tile_gemm(const bliss_tile_gemm_t* first, const bliss_tile_gemm_t* second, void* stream) {
  if (!first) return BLISS_EINVAL;
  TileGemm p0, p1;
  size_t l0 = 0, l1 = 0;
  int t0 = 0, t1 = 0;
  if (!convert(first, &p0, &l0, &t0)) return BLISS_EINVAL;
  p1 = p0;
  if (second && !convert(second, &p1, &l1, &t1)) return BLISS_EINVAL;
  const size_t lds = l0 > l1 ? l0 : l1;
  if (lds > 160 * 1024) return BLISS_EINVAL;
  static size_t lds_set = 0;
  if (lds > lds_set) {
    if (hipFuncSetAttribute((const void*)k_tile_gemm, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return BLISS_EINVAL;
    lds_set = lds;
  }
  const dim3 grid(t0 > t1 ? t0 : t1, second ? 2 : 1);
  static const int dbg = []() { const char* e = getenv("BLISS_TG_DEBUG"); return e ? atoi(e) : 0; }();
  k_tile_gemm<<<grid, TG_TPB, lds, (hipStream_t)stream>>>(p0, p1, second ? 2 : 1, dbg);
  return (int)hipGetLastError();
}
