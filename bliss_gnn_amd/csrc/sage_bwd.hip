// The backward products of a SAGE layer's two Linears on the matrix cores (round 3; the forward ones: sage.hip).
//
// dglnn.SAGEConv (model.py:303-308, 321-329) holds fc_neigh and fc_self as nn.Linear, W = [out, in]; their autograd needs
//   (1) the input gradients    dX[r, :] = dZ[r, :] . W            ("dgrad": sum over W's ROW index -- W is the k-strided operand)
//   (2) the weight gradients   dW[n, c] = sum_r dZ[r, n] X[r, c]   ("wgrad": sum over the block's rows -- BOTH operands k-strided)
//   (3) the bias gradient      db[n]    = sum_r dZ[r, n]
// Round 2 ran them as twelve hipBLASLt launches per step (285 us of the backward stream under the profiler, the input
// layer's dW at 1.5 % of the MFMA peak: 40 workgroups).  Here:
//   k_dgrad:  32 rows x 256 columns per workgroup like k_tile_gemm, two products into one accumulator (the layer input feeds
//             fc_neigh through the aggregation and fc_self directly), W slabs in LDS AS STORED ([k][n], coalesced 16-byte
//             copies) and read as B fragments with ds_read_b64_tr_b16 -- the hardware's transposing LDS read: no operand
//             is ever transposed in memory;
//   k_wgrad:  one workgroup = all <= 256 output rows x 128 output columns x one chunk of block rows; both operand chunks
//             are copied into LDS as stored ([r][n] and [r][c]) and BOTH fragments come from transposed reads; fp32 partial
//             tiles per chunk, summed in chunk order by k_wgrad_reduce (deterministic: no float atomics), which also rounds
//             to bf16; the bias gradient rides along as one more MFMA per tile against a fragment of ones.
// LDS rows are padded to stride = 32 (mod 128) elements: the four rows a transposed read touches per 32-lane half then sit
// in four different 16-bank groups (conflict-free; MI355X_MICROARCH.md section LDS).
#include "common.cuh"
#include "bliss_gnn.h"
#include <cstdlib>

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

// element j = 0..3 of the result: image row (r0 + j) at this lane's column, where the 16 lanes of a group 16g .. 16g + 15
// hold 16 consecutive columns and lane 4q + p of the group passes the address of row r0 + q, columns 4p .. 4p + 3
// (cdna_hip_programming.md T10).  EXEC must be all ones: never call under divergence.
__device__ __forceinline__ s16x4_t lds_tr4(const bf16_t* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
}
// a 32x32x16 MFMA operand fragment (8 consecutive k of this lane's row/column) out of a [k][n] image: two transposed reads
__device__ __forceinline__ bf16x8_t tr_frag(const bf16_t* img, int stride, int k0, int c0, int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const bf16_t* a = img + (size_t)(k0 + 8 * (g >> 1) + q) * stride + c0 + 16 * (g & 1) + 4 * p;
  const s16x4_t lo = lds_tr4(a), hi = lds_tr4(a + 4 * (size_t)stride);
  s16x8_t v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bf16x8_t, v);
}

__device__ __forceinline__ uint4 ld16(const bf16_t* p) {          // 16 bytes from a 4-byte aligned address
  const u32x4_a4 q = *reinterpret_cast<const u32x4_a4*>(p);
  return make_uint4(q[0], q[1], q[2], q[3]);
}
// 8 consecutive elements of a row, zero beyond `valid` elements (valid <= 0: all zero); element-wise when the row is not
// 4-byte aligned or the tail is partial
__device__ __forceinline__ uint4 ld8_masked(const bf16_t* p, int valid, bool aligned) {
  if (valid >= 8 && aligned) return ld16(p);
  union { uint4 u; bf16_t e[8]; } t;
  t.u = make_uint4(0, 0, 0, 0);
#pragma unroll
  for (int j = 0; j < 8; ++j) if (j < valid) t.e[j] = p[j];
  return t.u;
}

// ------------------------------------------------------------------------------------------------------------------ dgrad
#define DG_TPB 512
#define DG_M 32
#define DG_SLAB 64
#define DG_WSTRIDE (256 + 32)

struct DGrad {
  const bf16_t* a1; long long a1_stride; const bf16_t* w1; long long w1_stride; int k1;
  const bf16_t* a2; long long a2_stride; const bf16_t* w2; long long w2_stride; int k2; int m2_bound; const int* m2_dev;
  int m_bound; const int* m_dev; int n;
  bf16_t* out; long long out_stride;
};

__device__ __host__ __forceinline__ int dg_astride(int k) { return ((k + 15) & ~15) + 8; }

// rows row0 .. row0+31 of A (k columns, zero rows at or beyond M, zero padding up to the 16-multiple) -> LDS [32][astride]
__device__ __forceinline__ void dg_stage_a(const bf16_t* __restrict__ a, long long a_stride, int k, int row0, int M, bf16_t* tile, int tid) {
  const int kp = (k + 15) & ~15, st = dg_astride(k), chunks = kp / 8;
  const bool al = (a_stride % 2 == 0) && (((uintptr_t)a) % 4 == 0);
  for (int idx = tid; idx < DG_M * chunks; idx += DG_TPB) {
    const int lr = idx / chunks, c8 = idx - lr * chunks, r = row0 + lr;
    const uint4 v = r < M ? ld8_masked(a + (long long)r * a_stride + 8 * c8, k - 8 * c8, al) : make_uint4(0, 0, 0, 0);
    *reinterpret_cast<uint4*>(tile + (size_t)lr * st + 8 * c8) = v;
  }
}

// one slab of W as stored: rows k = slab*64 .. +63 (zero at or beyond K), columns n0 .. n0+255 (zero at or beyond N) -> LDS
__device__ __forceinline__ void dg_stage_w(const bf16_t* __restrict__ w, long long w_stride, int K, int N, int n0, int slab, bf16_t* wl, int tid) {
  const bool al = (w_stride % 2 == 0) && (((uintptr_t)w) % 4 == 0) && (n0 % 2 == 0);
  for (int idx = tid; idx < DG_SLAB * 32; idx += DG_TPB) {
    const int kr = idx >> 5, c8 = idx & 31, k = slab * DG_SLAB + kr, n = n0 + 8 * c8;
    const uint4 v = k < K ? ld8_masked(w + (long long)k * w_stride + n, N - n, al) : make_uint4(0, 0, 0, 0);
    *reinterpret_cast<uint4*>(wl + (size_t)kr * DG_WSTRIDE + 8 * c8) = v;
  }
}

__device__ __forceinline__ void dg_product(const bf16_t* tile, int k, const bf16_t* __restrict__ w, long long w_stride, int N, int n0,
                                           bf16_t* wl, int tid, int lane, int wave, f32x16_t& acc) {
  const int st = dg_astride(k), kp = (k + 15) & ~15, r = lane & 31, h = lane >> 5;
  const int nslab = (kp + DG_SLAB - 1) / DG_SLAB;
  for (int sl = 0; sl < nslab; ++sl) {
    __syncthreads();                                   // the previous slab has been consumed (and the A tile is in place)
    dg_stage_w(w, w_stride, k, N, n0, sl, wl, tid);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < DG_SLAB / 16; ++s) {
      const int ks = sl * DG_SLAB + 16 * s;
      if (ks < kp) {                                   // (workgroup-uniform)
        const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(tile + (size_t)r * st + ks + 8 * h);
        const bf16x8_t b = tr_frag(wl, DG_WSTRIDE, 16 * s, 32 * wave, lane);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
      }
    }
  }
}

__global__ void __launch_bounds__(DG_TPB) k_dgrad(DGrad p) {
  extern __shared__ __attribute__((aligned(16))) bf16_t dg_lds[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  int M = p.m_bound;
  if (p.m_dev) { const int t = *p.m_dev; M = t < M ? t : M; }
  int M2 = p.k2 ? p.m2_bound : 0;
  if (p.k2 && p.m2_dev) { const int t = *p.m2_dev; M2 = t < M2 ? t : M2; }
  if (M2 > M) M2 = M;
  const int row0 = blockIdx.x * DG_M, n0 = blockIdx.y * 256, N = p.n;
  if (row0 >= p.m_bound) return;
  const bool pair_ok = (p.out_stride % 2 == 0) && (((uintptr_t)p.out) % 4 == 0) && (N % 2 == 0);
  if (row0 >= M) {                                     // a tile of capacity padding: finite zeros
    for (int idx = tid; idx < DG_M * 256; idx += DG_TPB) {
      const int r = row0 + (idx >> 8), c = n0 + (idx & 255);
      if (r < p.m_bound && c < N) p.out[(long long)r * p.out_stride + c] = 0;
    }
    return;
  }
  const int s1 = dg_astride(p.k1), s2 = p.k2 ? dg_astride(p.k2) : 0;
  bf16_t* t1 = dg_lds;
  bf16_t* t2 = t1 + (size_t)DG_M * s1;
  bf16_t* wl = t2 + (size_t)DG_M * s2;
  dg_stage_a(p.a1, p.a1_stride, p.k1, row0, M, t1, tid);
  const bool second = p.k2 && row0 < M2;               // (workgroup-uniform)
  if (second) dg_stage_a(p.a2, p.a2_stride, p.k2, row0, M2, t2, tid);
  f32x16_t acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  dg_product(t1, p.k1, p.w1, p.w1_stride, N, n0, wl, tid, lane, wave, acc);
  if (second) dg_product(t2, p.k2, p.w2, p.w2_stride, N, n0, wl, tid, lane, wave, acc);
  __syncthreads();                                     // every wave is done with the staged rows: reuse the LDS for the output tile
  const int so = 256 + 8;
  bf16_t* ot = dg_lds;
  {
    const int col = 32 * wave + (lane & 31);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int lr = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
      ot[(size_t)lr * so + col] = (row0 + lr < M) ? f2bf(acc[i]) : (bf16_t)0;
    }
  }
  __syncthreads();
  for (int lr = wave; lr < DG_M; lr += DG_TPB / 64) {
    const int r = row0 + lr;
    if (r >= p.m_bound) break;
    const bf16_t* sh = ot + (size_t)lr * so;
    bf16_t* o = p.out + (long long)r * p.out_stride + n0;
    const int nn = min(256, N - n0);
    if (pair_ok && (n0 % 2 == 0)) { for (int c = lane * 2; c < nn; c += 128) *reinterpret_cast<uint32_t*>(o + c) = *reinterpret_cast<const uint32_t*>(sh + c); }
    else for (int c = lane; c < nn; c += 64) o[c] = sh[c];
  }
}

// ------------------------------------------------------------------------------------------------------------------ wgrad
#define WG_TPB 512
#define WG_ROWS 32                       // block rows per pipeline step
#define WG_TN 128                        // output columns (= columns of X) per workgroup
#define WG_DSTRIDE (256 + 32)
#define WG_XSTRIDE (WG_TN + 32)
#define WG_MAX_PROBS 2

struct WGradProb {
  const bf16_t* d; long long d_stride; int n_out;
  const bf16_t* x; long long x_stride; int k_in;
  int rows_bound; const int* rows_dev;
  bf16_t* dw; long long dw_stride; bf16_t* db;
  int col_tiles, chunks, rows_per_chunk; long long part_off;      // partial tiles of this problem: [chunks][256][ldp] floats
  int ldp;
  int wg_begin;                          // first workgroup of this problem in the launch
};
struct WGradLaunch { WGradProb p[WG_MAX_PROBS]; int n; };

struct WRegsB { uint4 d[2]; uint4 x; };   // one step's share of a thread: 32 x 256 of D (two 16-byte pieces), 32 x 128 of X (one)

__device__ __forceinline__ void wg_load(WRegsB& g, const WGradProb& p, int r_lo, int r_hi, int c0, int tid, bool d_al, bool x_al) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + i * WG_TPB, lr = idx >> 5, c8 = idx & 31, r = r_lo + lr, n = 8 * c8;
    g.d[i] = (r < r_hi) ? ld8_masked(p.d + (long long)r * p.d_stride + n, p.n_out - n, d_al) : make_uint4(0, 0, 0, 0);
  }
  {
    const int lr = tid >> 4, c8 = tid & 15, r = r_lo + lr, c = c0 + 8 * c8;
    g.x = (r < r_hi) ? ld8_masked(p.x + (long long)r * p.x_stride + c, p.k_in - c, x_al) : make_uint4(0, 0, 0, 0);
  }
}
__device__ __forceinline__ void wg_store(const WRegsB& g, bf16_t* dimg, bf16_t* ximg, int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + i * WG_TPB, lr = idx >> 5, c8 = idx & 31;
    *reinterpret_cast<uint4*>(dimg + (size_t)lr * WG_DSTRIDE + 8 * c8) = g.d[i];
  }
  *reinterpret_cast<uint4*>(ximg + (size_t)(tid >> 4) * WG_XSTRIDE + 8 * (tid & 15)) = g.x;
}

__global__ void __launch_bounds__(WG_TPB) k_wgrad(WGradLaunch L, float* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) bf16_t dimg[WG_ROWS * WG_DSTRIDE];
  __shared__ __attribute__((aligned(16))) bf16_t ximg[WG_ROWS * WG_XSTRIDE];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int pi = (L.n > 1 && (int)blockIdx.x >= L.p[1].wg_begin) ? 1 : 0;
  const WGradProb& p = L.p[pi];
  const int w = (int)blockIdx.x - p.wg_begin;
  const int ct = w % p.col_tiles, ch = w / p.col_tiles;
  int R = p.rows_bound;
  if (p.rows_dev) { const int t = *p.rows_dev; R = t < R ? t : R; }
  const int r_lo = ch * p.rows_per_chunk, r_hi = min(R, r_lo + p.rows_per_chunk);
  if (r_lo >= r_hi) return;                            // a chunk of capacity padding: k_wgrad_reduce skips it too
  const int c0 = ct * WG_TN;
  const int wm = wave >> 1, wn = wave & 1;             // 4 x 2 waves: 64 output rows x 64 output columns each
  const int m0 = 64 * wm, nl0 = 64 * wn;
  const bool m_act = m0 < p.n_out;                     // (wave-uniform; n_out = 41 leaves three of the four row groups idle)
  const bool bias = p.db != nullptr && ct == 0 && wn == 0;
  f32x16_t acc[2][2], accb[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[a][0][i] = 0.f; acc[a][1][i] = 0.f; accb[a][i] = 0.f; }
  }
  s16x8_t ones_s;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones_s[j] = (short)0x3f80;
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_s);
  const bool d_al = (p.d_stride % 2 == 0) && (((uintptr_t)p.d) % 4 == 0);
  const bool x_al = (p.x_stride % 2 == 0) && (((uintptr_t)p.x) % 4 == 0);
  WRegsB g;
  wg_load(g, p, r_lo, r_hi, c0, tid, d_al, x_al);
  for (int rs = r_lo; rs < r_hi; rs += WG_ROWS) {
    __syncthreads();                                   // the previous step's images have been consumed
    wg_store(g, dimg, ximg, tid);
    if (rs + WG_ROWS < r_hi) wg_load(g, p, rs + WG_ROWS, r_hi, c0, tid, d_al, x_al);
    __syncthreads();
    if (m_act) {
#pragma unroll
      for (int ks = 0; ks < WG_ROWS; ks += 16) {
        const bf16x8_t a0 = tr_frag(dimg, WG_DSTRIDE, ks, m0, lane), a1 = tr_frag(dimg, WG_DSTRIDE, ks, m0 + 32, lane);
        const bf16x8_t b0 = tr_frag(ximg, WG_XSTRIDE, ks, nl0, lane), b1 = tr_frag(ximg, WG_XSTRIDE, ks, nl0 + 32, lane);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        if (bias) {                                    // (wave-uniform) column sums of D: the same rows against ones
          accb[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, ones, accb[0], 0, 0, 0);
          accb[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, ones, accb[1], 0, 0, 0);
        }
      }
    }
  }
  if (!m_act) return;
  // fp32 partial tile of this chunk: part[ch][n][c], row stride ldp; the bias sums in column col_tiles * WG_TN
  float* pt = part + p.part_off + (long long)ch * 256 * p.ldp;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int n = m0 + 32 * a + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
      float* row = pt + (long long)n * p.ldp;
      row[c0 + nl0 + (lane & 31)] = acc[a][0][i];
      row[c0 + nl0 + 32 + (lane & 31)] = acc[a][1][i];
      if (bias && (lane & 31) == 0) row[p.col_tiles * WG_TN] = accb[a][i];
    }
  }
}

// dW[n, c] = bf16( sum over the chunks that hold rows, in chunk order ), db[n] likewise; one thread per 4 columns
__global__ void __launch_bounds__(256) k_wgrad_reduce(WGradLaunch L, const float* __restrict__ part) {
  const int pi = blockIdx.y;
  const WGradProb& p = L.p[pi];
  int R = p.rows_bound;
  if (p.rows_dev) { const int t = *p.rows_dev; R = t < R ? t : R; }
  const int nch = R > 0 ? min(p.chunks, (R + p.rows_per_chunk - 1) / p.rows_per_chunk) : 0;
  const int q4 = (p.k_in + 3) / 4 + (p.db ? 1 : 0);    // column quads of the weight rows (+ one slot per row for the bias)
  const long long total = (long long)p.n_out * q4;
  const float* base = part + p.part_off;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int n = (int)(idx / q4), q = (int)(idx - (long long)n * q4);
    if (p.db && q == q4 - 1) {
      float s = 0.f;
      for (int c = 0; c < nch; ++c) s += base[((long long)c * 256 + n) * p.ldp + p.col_tiles * WG_TN];
      p.db[n] = f2bf(s);
      continue;
    }
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c = 0; c < nch; ++c) {
      const float4 v = *reinterpret_cast<const float4*>(base + ((long long)c * 256 + n) * p.ldp + 4 * q);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    bf16_t* o = p.dw + (long long)n * p.dw_stride + 4 * q;
    const int left = p.k_in - 4 * q;
    o[0] = f2bf(s.x);
    if (left > 1) o[1] = f2bf(s.y);
    if (left > 2) o[2] = f2bf(s.z);
    if (left > 3) o[3] = f2bf(s.w);
  }
}

// How many workgroups a launch aims for.  NOT "as many as the chip holds": the weight gradients run on the backward stream
// beside the sampler's latency-bound chain, which IS the step's critical path, and a launch that spreads over every CU slows
// that chain down more than it gains itself -- measured on the Reddit-like step (same box, 400-step windows): 16 workgroups
// 1231 steps/s (the backward stream becomes the longer chain), 32: 1467, 48: 1505, 64: 1510, 96: 1509, 128: 1493, 224: 1484,
// 448: 1454.  Fewer chunks also mean fewer fp32 partial tiles (80 workgroups: ~7 MB for the input layer instead of 20).
int wgrad_target_wgs() {
  static const int t = []() { const char* e = getenv("BLISS_WGRAD_WGS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 80; }();
  return t;
}

// chunking of one problem: as many chunks as the launch can keep busy, at least 64 rows each
bool wgrad_plan(const bliss_wgrad_t* a, int n_probs, WGradLaunch* L, long long* floats, int* total_wgs) {
  if (n_probs < 1 || n_probs > WG_MAX_PROBS) return false;
  long long off = 0;
  int wgs = 0;
  int tiles_all = 0;
  for (int i = 0; i < n_probs; ++i) {
    if (!a[i].d || !a[i].x || !a[i].dw || a[i].n_out <= 0 || a[i].n_out > 256 || a[i].k_in <= 0 || a[i].rows_bound <= 0) return false;
    tiles_all += (a[i].k_in + WG_TN - 1) / WG_TN;
  }
  for (int i = 0; i < n_probs; ++i) {
    WGradProb& p = L->p[i];
    p.d = (const bf16_t*)a[i].d; p.d_stride = a[i].d_stride; p.n_out = a[i].n_out;
    p.x = (const bf16_t*)a[i].x; p.x_stride = a[i].x_stride; p.k_in = a[i].k_in;
    p.rows_bound = a[i].rows_bound; p.rows_dev = a[i].rows_dev;
    p.dw = (bf16_t*)a[i].dw; p.dw_stride = a[i].dw_stride; p.db = (bf16_t*)a[i].db;
    p.col_tiles = (a[i].k_in + WG_TN - 1) / WG_TN;
    int chunks = wgrad_target_wgs() / tiles_all;
    const int max_chunks = (a[i].rows_bound + 63) / 64;
    if (chunks > max_chunks) chunks = max_chunks;
    if (chunks < 1) chunks = 1;
    int rpc = (a[i].rows_bound + chunks - 1) / chunks;
    rpc = (rpc + WG_ROWS - 1) / WG_ROWS * WG_ROWS;
    chunks = (a[i].rows_bound + rpc - 1) / rpc;
    p.chunks = chunks; p.rows_per_chunk = rpc;
    p.ldp = p.col_tiles * WG_TN + 4;
    p.part_off = off;
    p.wg_begin = wgs;
    off += (long long)chunks * 256 * p.ldp;
    wgs += chunks * p.col_tiles;
  }
  L->n = n_probs;
  *floats = off;
  *total_wgs = wgs;
  return true;
}

}  // namespace

extern "C" {

int bliss_sage_dgrad(const bliss_dgrad_t* a, void* stream) {
  if (!a || !a->a1 || !a->w1 || !a->out || a->k1 <= 0 || a->k1 > 256 || a->n <= 0 || a->m_bound <= 0) return BLISS_EINVAL;
  if (a->k2 < 0 || a->k2 > 256 || (a->k2 > 0 && (!a->a2 || !a->w2 || a->m2_bound <= 0))) return BLISS_EINVAL;
  DGrad p;
  p.a1 = (const bf16_t*)a->a1; p.a1_stride = a->a1_stride; p.w1 = (const bf16_t*)a->w1; p.w1_stride = a->w1_stride; p.k1 = a->k1;
  p.a2 = (const bf16_t*)a->a2; p.a2_stride = a->a2_stride; p.w2 = (const bf16_t*)a->w2; p.w2_stride = a->w2_stride; p.k2 = a->k2;
  p.m2_bound = a->m2_bound; p.m2_dev = a->m2_dev;
  p.m_bound = a->m_bound; p.m_dev = a->m_dev; p.n = a->n;
  p.out = (bf16_t*)a->out; p.out_stride = a->out_stride;
  const size_t stage = ((size_t)DG_M * (dg_astride(a->k1) + (a->k2 ? dg_astride(a->k2) : 0)) + (size_t)DG_SLAB * DG_WSTRIDE) * sizeof(bf16_t);
  const size_t outt = (size_t)DG_M * (256 + 8) * sizeof(bf16_t);
  const size_t lds = stage > outt ? stage : outt;
  static size_t lds_set = 0;
  if (lds > lds_set) {
    if (hipFuncSetAttribute((const void*)k_dgrad, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return BLISS_EINVAL;
    lds_set = lds;
  }
  const dim3 grid((a->m_bound + DG_M - 1) / DG_M, (a->n + 255) / 256);
  k_dgrad<<<grid, DG_TPB, lds, (hipStream_t)stream>>>(p);
  return (int)hipGetLastError();
}

int64_t bliss_sage_wgrad_workspace(const bliss_wgrad_t* probs, int32_t n_probs) {
  WGradLaunch L;
  long long floats = 0;
  int wgs = 0;
  if (!probs || !wgrad_plan(probs, n_probs, &L, &floats, &wgs)) return -1;
  return (int64_t)floats;
}

int bliss_sage_wgrad(const bliss_wgrad_t* probs, int32_t n_probs, float* partials, int64_t partial_floats, void* stream) {
  WGradLaunch L;
  long long floats = 0;
  int wgs = 0;
  if (!probs || !partials || !wgrad_plan(probs, n_probs, &L, &floats, &wgs)) return BLISS_EINVAL;
  if (partial_floats < floats || (((uintptr_t)partials) % 16) != 0) return BLISS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  k_wgrad<<<wgs, WG_TPB, 0, st>>>(L, partials);
  int max_elems = 0;
  for (int i = 0; i < n_probs; ++i) { const int e = L.p[i].n_out * ((L.p[i].k_in + 3) / 4 + 1); if (e > max_elems) max_elems = e; }
  int gx = (max_elems + 255) / 256;
  if (gx > 1024) gx = 1024;
  k_wgrad_reduce<<<dim3(gx, n_probs), 256, 0, st>>>(L, partials);
  return (int)hipGetLastError();
}

}  // extern "C"
