// GATv2 message passing, one (virtual) workgroup per DESTINATION row (round 3) -- custom_GATv2Conv.forward, model.py:82-99, and its
// backward by destination.  csrc/gat.hip runs the same arithmetic as five launches per direction that each re-gather the
// 2 KB feature rows of every edge (logits: source + destination row; aggregation: source row again; backward: seven row
// gathers per edge) and pass [B, H] tensors through memory in between: 760 us forward and 1.2 ms backward per step on the
// Reddit-like config (profiles/r03_d_gat_step_timeline.txt).  Here a workgroup of 8 waves owns up to 256 in-edges of one
// destination i (k_gat_segments: one 16-byte descriptor per virtual workgroup, longest first; a longer row is shared, see below):
//   forward   edges in chunks of 64 (8 per wave): all row gathers of a chunk in flight at once, the rows kept PACKED IN REGISTERS
//             (64 VGPRs); pass 1: logits e_ij[h] (stored: the reference returns them as "attention", model.py:108-110; kept in
//             LDS for the next passes) and the per-head maximum; pass 2: the edge softmax over the logits in LDS (exact
//             per-destination sum like every copy_e_sum), attention dropout (model.py:88) folded in; pass 3: sum_j a_ij el_j
//             from the rows still in registers -- a row of <= 64 in-edges (most) is gathered ONCE; earlier chunks of a longer
//             row are gathered a second time.  One launch instead of five, no partial / fix-up pass.
//   backward  (by destination, same shape) g_i = d rst_i packed in registers, er_i in LDS; pass 1: d a_ij = g_i . el_j on
//             v_dot2_f32_bf16 and t = sum a d a; pass 2: d e_ij = a (d a - t) (softmax backward); pass 3: d er_i = attn * sum_j
//             d e_ij lrelu'(el_j + er_i) and the workgroup's share of d attn.  The by-SOURCE half (d el_j: out-degrees are far
//             more skewed) stays on the merge-style kernel of gat.hip, with the aggregation's backward folded in.
// The forward rounds exactly where gat.hip's kernels round (= where the reference's bf16 tensor ops round): the golden
// fixtures of tests/golden/gat*_model_exp3.npz hold for both paths.  bf16 conversions use the hardware's
// v_cvt_pk_bf16_f32 (round to nearest even, like common.cuh:f2bf on every finite input).
// Where a workgroup's time goes is measured by the kernels themselves (bliss_gat_fused_stamps, scratch/gatbench.py).
#include "common.cuh"
#include "bliss_gnn.h"

namespace {

#define GF_TPB 512
#define GF_WAVES (GF_TPB / 64)
#define GF_MAXH 8
#define GF_ITER 4                      // a lane owns up to 4 groups of W consecutive columns: H*D <= 64 * W * 4
#define GF_EW 8                        // in-edges per wave and chunk: their gathered rows are all in flight / in registers together
#define GF_CHUNK (GF_WAVES * GF_EW)    // in-edges per chunk of a workgroup (64)
#define GF_NCH 4                       // chunks per workgroup
#define GF_SEG (GF_CHUNK * GF_NCH)     // in-edges per workgroup (256): longer rows are shared by ceil(deg / GF_SEG) workgroups
#define GF_ROWWS 32                    // words of cross-workgroup state per destination row
// rowws layout (uint32 words): [0..2] arrive counters of the three meeting points, [3] error bits, [4..11] per-head maximum
// (order-preserving encoding of the float), [12..27] per-head exact sum (uint64 each)

__device__ __forceinline__ float rbf_hw(float f) { return (float)(__bf16)f; }
__device__ __forceinline__ bf16_t f2bf_hw(float f) { const __bf16 b = (__bf16)f; return __builtin_bit_cast(unsigned short, b); }
__device__ __forceinline__ float lrelu_f(float x, float s) { return x > 0.f ? x : s * x; }
// v_max_f32 as it is (fmaxf adds a canonicalising v_max per operand); a NaN logit is caught by the exact sum's flag instead
__device__ __forceinline__ float max_raw(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

struct g4f { float v[4]; };
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// two bf16 values at a time: unpack (2 ops), v_pk_add_f32 / v_pk_mul_f32 on the pair, ONE v_cvt_pk_bf16_f32 for both roundings
__device__ __forceinline__ f32x2_t unpack2(unsigned u) { return f32x2_t{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)}; }
__device__ __forceinline__ unsigned round2(f32x2_t v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t)); }
// rbf(attn * rbf(lrelu(rbf(el + er)))) of a packed pair of el: the operations of k_gat_edge_dot<0>, element for element
__device__ __forceinline__ f32x2_t logit_terms2(unsigned el2, f32x2_t er, f32x2_t at, float slope) {
  const f32x2_t r = unpack2(round2(unpack2(el2) + er));
  const f32x2_t l = unpack2(round2(r * f32x2_t{slope, slope}));
  const f32x2_t sel = f32x2_t{r.x > 0.f ? r.x : l.x, r.y > 0.f ? r.y : l.y};
  return unpack2(round2(sel * at));
}
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

// phase stamps for scratch/gatbench.py (bliss_gat_fused_stamps): 8 x s_memrealtime (100 MHz) per virtual workgroup of the forward
__device__ long long* g_gf_stamps = nullptr;
__device__ long long* g_gf_stamps_bwd = nullptr;
#define GF_STAMP(k) do { if (stamps && threadIdx.x == 0) stamps[(long long)blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)

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
  float* dattn_part;                                   // [n_wg, H*D] out: this workgroup's share of d attn
  // heavy rows: a destination with more than GF_SEG in-edges is shared by several workgroups (segments of GF_SEG edges)
  const int4* wg_row; const int* n_wg_dev;             // virtual workgroup -> (row, first edge, end, G << 16 | segment): k_gat_segments; how many
  unsigned* rowws;                                     // [n_dst, GF_ROWWS] zero-initialised, self-cleaning: arrive counters, max, sum
  float* seg_part;                                     // [n_wg, H*D] partial rows of the segments; [n_wg, GF_MAXH] behind it for t
  int* err;
};

// A row of one edge as the lane's NG column groups, still packed (two registers per group of four bf16).
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
// A workgroup's edges come in chunks of GF_CHUNK = 64: edge k * 64 + j * 8 + w of the workgroup is the j-th edge of wave w in chunk
// k, and lane k * 8 + j of wave w holds its source id (my_s: ONE coalesced load per wave for the whole workgroup).
// ld_wave_rows fetches ALL rows of the wave's (at most GF_EW) edges of one chunk, packed (8 VGPRs per edge when VEC4): every
// gather of the chunk is in flight at the same time, and the rows of the workgroup's LAST chunk stay in registers from the first
// pass to the last -- a row of at most 64 in-edges (most rows) is gathered once.  Slots beyond the wave's share re-read its last
// edge (same cache lines, never used); the upper half of the slots is skipped altogether when the wave has at most four edges.
__device__ __forceinline__ int wave_edges_in_chunk(int n, int k, int wave) {
  const int left = min(n - k * GF_CHUNK, GF_CHUNK) - wave;
  return left <= 0 ? 0 : (left + GF_WAVES - 1) / GF_WAVES;
}
template <bool VEC4, int NG>
__device__ __forceinline__ void ld_wave_rows(RawGroup<VEC4> (&raw)[GF_EW][NG], const bf16_t* feat, long long stride, int my_s, int lane0,
                                             int n_mine, const int (&coff)[NG]) {
  const int last = n_mine > 0 ? n_mine - 1 : 0;
#pragma unroll
  for (int j = 0; j < GF_EW / 2; ++j)
    ld_edge_raw<VEC4, NG>(raw[j], feat + (long long)__builtin_amdgcn_readlane(my_s, lane0 + (j < last ? j : last)) * stride, coff);
  if (n_mine > GF_EW / 2) {                           // (wave-uniform)
#pragma unroll
    for (int j = GF_EW / 2; j < GF_EW; ++j)
      ld_edge_raw<VEC4, NG>(raw[j], feat + (long long)__builtin_amdgcn_readlane(my_s, lane0 + (j < last ? j : last)) * stride, coff);
  }
}


// ---- several workgroups on one destination row ---------------------------------------------------------------------------
// The sampled blocks hold destinations with ~1000 in-edges beside a mean of ~50 (hubs of the lognormal degree law): one
// workgroup walking such a row alone WAS the kernel's duration (scratch/gatbench.py: 250 us on the Reddit-like input layer with
// a 980-edge row, whatever the other 3,376 rows cost).  A row is therefore cut into segments of GF_SEG edges, one workgroup
// each (consecutive virtual workgroup ids: they are dispatched together).  The arithmetic does not change: the row's maximum is
// an atomicMax over the segments' maxima, its softmax sum an integer atomicAdd of their exact partial sums (order-free), the
// output row the sum of the segments' fp32 partial rows IN SEGMENT ORDER by whichever workgroup arrives last -- so the bits do
// not depend on timing.  Meeting points are counters the segments' workgroups add to and poll (bounded: a row whose partners
// never arrive sets BLISS_ERR_FLAG_TIMEOUT and goes on, the grid always drains); the last workgroup through the third one
// returns the row's words to zero for the next launch.
__device__ __forceinline__ unsigned f2ord(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

// all threads call; returns once every one of the row's G workgroups has arrived at meeting point `which` (0 or 1)
__device__ __forceinline__ void row_meet(unsigned* ws, int which, int G, int* err) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's stores / atomics have left
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    atomicAdd(ws + which, 1u);
    long long spins = 0;
    while (__hip_atomic_load(ws + which, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)G) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1ll << 21)) { if (err) atomicOr(err, BLISS_ERR_FLAG_TIMEOUT); break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
}
// the third meeting point only counts: true in the workgroup that arrived LAST (it then reads what the others left)
__device__ __forceinline__ bool row_last(unsigned* ws, int G, int* sh_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const bool last = atomicAdd(ws + 2, 1u) == (unsigned)G - 1u;
    if (last) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    *sh_flag = last ? 1 : 0;
  }
  __syncthreads();
  return *sh_flag != 0;
}

// virtual workgroup -> destination row: row r gets max(1, ceil(deg_r / GF_SEG)) consecutive ids (rows of capacity padding have
// no edges: one id each, their workgroup writes the zero row).  The shared rows get the LOWEST ids: their workgroups are the
// long ones (four chunks and three meeting points) and the dispatcher hands out workgroups in id order.  One workgroup scans the
// rows 1024 at a time, twice.
#define GF_SEG_R 16                     // rows per thread of the one-sweep path
__device__ __forceinline__ int gf_row_wgs(int deg) { return deg > GF_SEG ? (deg + GF_SEG - 1) / GF_SEG : 1; }
// what one virtual workgroup needs to know, in ONE 16-byte load: its row, its edge range, how many workgroups share the row
__device__ __forceinline__ int4 gf_desc(int row, int rbeg, int rend, int G, int seg_i) {
  const int beg = rbeg + seg_i * GF_SEG;
  return make_int4(row, beg, G > 1 ? min(rend, beg + GF_SEG) : rend, (G << 16) | seg_i);
}
__global__ void __launch_bounds__(1024) k_gat_segments(const int* __restrict__ indptr, int n_dst, int cap_wg, int4* __restrict__ wg_row,
                                                        int* __restrict__ n_wg_dev, int* err) {
  __shared__ int sh[17];
  int run = 0;
  if (n_dst <= 1024 * GF_SEG_R) {
    // one sweep: a thread owns R consecutive rows, all their degrees loaded up front (the launch sits on the forward's critical
    // path: 15 us as two scanning loops over 9 K rows, a third of that like this)
    const int R = (n_dst + 1023) / 1024, r0 = threadIdx.x * R;
    int ip[GF_SEG_R + 1];
#pragma unroll
    for (int i = 0; i <= GF_SEG_R; ++i) ip[i] = (i <= R && r0 + i <= n_dst) ? indptr[r0 + i] : 0;
    // longest first: the shared rows, then the rows of 4, 3, 2 chunks, then the single-chunk ones (a workgroup's life is
    // ~10 us per chunk; the dispatcher hands ids out in order, so the long ones must not be the last to start)
    int n_sh = 0, n_43 = 0, n_21 = 0;                  // (two 16-bit counts per word: at most 16384 rows)
#pragma unroll
    for (int i = 0; i < GF_SEG_R; ++i) {
      if (i < R && r0 + i < n_dst) {
        const int deg = ip[i + 1] - ip[i], g = gf_row_wgs(deg), ch = (deg + GF_CHUNK - 1) / GF_CHUNK;
        if (g > 1) n_sh += g;
        else if (ch >= 4) n_43 += 1 << 16;
        else if (ch == 3) n_43 += 1;
        else if (ch == 2) n_21 += 1 << 16;
        else n_21 += 1;
      }
    }
    int tot_sh, tot_43, tot_21;
    int at_sh = block_excl_scan(n_sh, sh, &tot_sh);
    __syncthreads();
    const int ex_43 = block_excl_scan(n_43, sh, &tot_43);
    __syncthreads();
    const int ex_21 = block_excl_scan(n_21, sh, &tot_21);
    const int c4 = tot_43 >> 16, c3 = tot_43 & 0xffff, c2 = tot_21 >> 16, c1 = tot_21 & 0xffff;
    int at_4 = tot_sh + (ex_43 >> 16), at_3 = tot_sh + c4 + (ex_43 & 0xffff);
    int at_2 = tot_sh + c4 + c3 + (ex_21 >> 16), at_1 = tot_sh + c4 + c3 + c2 + (ex_21 & 0xffff);
#pragma unroll
    for (int i = 0; i < GF_SEG_R; ++i) {
      if (i < R && r0 + i < n_dst) {
        const int deg = ip[i + 1] - ip[i], g = gf_row_wgs(deg), ch = (deg + GF_CHUNK - 1) / GF_CHUNK;
        if (g > 1) { for (int q = 0; q < g; ++q) if (at_sh + q < cap_wg) wg_row[at_sh + q] = gf_desc(r0 + i, ip[i], ip[i + 1], g, q); at_sh += g; }
        else {
          int& at = ch >= 4 ? at_4 : ch == 3 ? at_3 : ch == 2 ? at_2 : at_1;
          if (at < cap_wg) wg_row[at] = gf_desc(r0 + i, ip[i], ip[i + 1], 1, 0);
          ++at;
        }
      }
    }
    run = tot_sh + c4 + c3 + c2 + c1;
  } else {
    for (int pass = 0; pass < 2; ++pass) {
      for (int base = 0; base < n_dst; base += 1024) {
        const int r = base + threadIdx.x;
        int g = 0, b0 = 0, b1 = 0;
        if (r < n_dst) {
          b0 = indptr[r]; b1 = indptr[r + 1];
          g = gf_row_wgs(b1 - b0);
          if ((g > 1) != (pass == 0)) g = 0;
        }
        int tot, ex = block_excl_scan(g, sh, &tot);
        for (int i = 0; i < g; ++i) if (run + ex + i < cap_wg) wg_row[run + ex + i] = gf_desc(r, b0, b1, g, i);
        run += tot;
        __syncthreads();
      }
    }
  }
  if (threadIdx.x == 0) {
    if (run > cap_wg) { if (err) atomicOr(err, BLISS_ERR_CAP_EDGES); run = cap_wg; }
    *n_wg_dev = run;
  }
}

// per-head sums of a lane's partial values: part[h] over the wave (all lanes get the totals).  The butterfly s += s[lane ^ d],
// d = 32, 16, .. 1 (the order csrc/gat.hip's __shfl_xor loop adds in: same bits) on the VALU: gfx950's v_permlane32_swap /
// v_permlane16_swap for d = 32 / 16 (both halves of the swapped pair added: fp32 addition commutes), DPP row_ror:8 and quad_perm
// for d = 8, 2, 1, ds_swizzle for d = 4 -- no ds_bpermute address arithmetic, no LDS round trip per step
__device__ __forceinline__ float wave_butterfly_sum(float s) {
  const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
  s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(s), __float_as_uint(s), false, false);
  s = __uint_as_float(b[0]) + __uint_as_float(b[1]);
  s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x128, 0xf, 0xf, false));          // row_ror:8  == lane ^ 8
  s += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(s), 0x1f | (4 << 10)));                    // lane ^ 4
  s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x4e, 0xf, 0xf, false));           // quad_perm [2,3,0,1]
  s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0xb1, 0xf, 0xf, false));           // quad_perm [1,0,3,2]
  return s;
}
template <int NH>
__device__ __forceinline__ void wave_sum_heads(float (&part)[NH], int H) {
#pragma unroll
  for (int h = 0; h < NH; ++h)
    if (h < H) part[h] = wave_butterfly_sum(part[h]);
}

// this workgroup's share of its destination row: all of it (G == 1: at most GF_SEG edges) or segment seg_i of G
struct RowSeg { int row, beg, end, G, seg_i; };
__device__ __forceinline__ RowSeg row_segment(const GatFused& p, int vwg) {
  const int4 d = p.wg_row[vwg];
  RowSeg r;
  r.row = d.x; r.beg = d.y; r.end = d.z; r.G = d.w >> 16; r.seg_i = d.w & 0xffff;
  return r;
}

// Forward.  A workgroup holds at most GF_SEG = 256 edges (longer rows are shared), so the logits and the softmax coefficients of
// its edges live in LDS between the passes (the global copies are outputs only: nothing is re-read from memory, no wait for a
// store, no agent-scope load) and a wave's source ids are ONE coalesced load kept in registers for both gather passes.
template <bool VEC4, int HG>
__global__ void __launch_bounds__(GF_TPB, 4) k_gat_fwd(GatFused p) {
  constexpr int W = VEC4 ? 4 : 1;
  // HG > 0: "a column group is a head" (VEC4, D == 256: group c of 64 lanes x 4 columns IS head c, H == HG <= 4) -- the Reddit
  // config's 4 x 256; every per-head selection below is then a compile-time index.  HG == 0: any H <= 8, D (head by compare)
  constexpr int NG = HG ? HG : GF_ITER, NH = HG ? HG : GF_MAXH;
  __shared__ float sh_acc[GF_WAVES][NG * 64 * W];          // cross-wave reduction of the output row (32 KiB when VEC4)
  __shared__ float sh_e[GF_SEG][GF_MAXH];                  // logits of this workgroup's edges, then their softmax coefficients
  __shared__ float sh_er[NG * 64 * W], sh_at[NG * 64 * W]; // er_i, attn (zero beyond H*D)
  __shared__ float sh_max[GF_WAVES][GF_MAXH];
  __shared__ unsigned long long sh_sum[GF_MAXH];
  __shared__ int sh_bad;
  __shared__ int sh_last;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int vwg = blockIdx.x;
  long long* stamps = g_gf_stamps;
  GF_STAMP(0);
  if (vwg >= *p.n_wg_dev) return;
  int S = p.n_dst;
  if (p.n_dst_dev) { const int t = *p.n_dst_dev; S = t < S ? t : S; }
  const int H = HG ? HG : p.H, D = p.D, HD = H * D;
  const uint32_t ctr = p.drop_thresh ? (uint32_t)__hip_atomic_load(p.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
  if (vwg == 0 && tid == 0 && p.drop_thresh && p.ctr_used) *p.ctr_used = ctr;
  const RowSeg rs = row_segment(p, vwg);
  const int row = rs.row, beg = rs.beg, end = rs.end, G = rs.G, seg_i = rs.seg_i;
  GF_STAMP(1);
  if (row >= S) {                                      // capacity padding: finite zeros
    for (int c = tid; c < HD; c += GF_TPB) p.rst[(long long)row * p.rst_stride + c] = 0;
    return;
  }
  unsigned* ws = p.rowws + (long long)row * GF_ROWWS;
  const int my_e = beg + (lane >> 3) * GF_CHUNK + (lane & 7) * GF_WAVES + wave;
  const int my_s = (lane < GF_NCH * GF_EW && my_e < end) ? p.src[my_e] : 0;
  const int n_edges = end - beg, n_chunks = (n_edges + GF_CHUNK - 1) / GF_CHUNK;
  // this lane's columns: group c covers columns c*64*W + lane*W .. +W-1 (one head per group when VEC4: D % 4 == 0)
  float acc[NG][W];
  int hd[NG], coff[NG];
#pragma unroll
  for (int c = 0; c < NG; ++c) {
    const int col = c * 64 * W + lane * W;
    hd[c] = col < HD ? col / D : -1;
    coff[c] = col < HD ? col : 0;                       // (masked columns re-read the row's first W elements; attn = 0 there)
#pragma unroll
    for (int jj = 0; jj < W; ++jj) acc[c][jj] = 0.f;
  }
  // er_i and attn go through LDS (pass 1 reads them one column group at a time: 32 registers less beside the packed rows);
  // threads 0..255 fetch er_i, 256..511 attn, W columns each -- in flight together with the source ids and the row gathers
  g4f er_at = g4f{{0.f, 0.f, 0.f, 0.f}};
  {
    const int col = (tid & 255) * W;
    if (col < HD) er_at = ldrow<VEC4>((tid >> 8) ? p.attn + col : p.feat + (long long)row * p.feat_stride + col);
  }
  if (tid < GF_MAXH) sh_sum[tid] = 0ull;
  if (tid == 0) sh_bad = 0;
  // ---- pass 1: logits (model.py:82-86, op by op like k_gat_edge_dot<0>) and the per-head maximum
  float mx[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) mx[h] = -__builtin_inff();
  bool staged = false;
  auto logits_of_chunk = [&](RawGroup<VEC4> (&raw)[GF_EW][NG], int k) {
    const int n_mine = wave_edges_in_chunk(n_edges, k, wave);
    ld_wave_rows<VEC4, NG>(raw, p.feat, p.feat_stride, my_s, k * GF_EW, n_mine, coff);
    if (!staged) {                                    // (block-uniform; behind the first chunk's gathers, which stay in flight)
      float* dstp = (tid >> 8) ? sh_at : sh_er;
#pragma unroll
      for (int jj = 0; jj < W; ++jj) dstp[(tid & 255) * W + jj] = er_at.v[jj];
      __syncthreads();
      staged = true;
    }
    if (HG) {
      // a column group is a head: group by group (er_i / attn of ONE group in registers), the group's dot product of every edge
      // finished by its own butterfly -- the same operations in the same order per (edge, head) as edge by edge
      float mine[GF_EW];
#pragma unroll
      for (int j = 0; j < GF_EW; ++j) mine[j] = 0.f;
#pragma unroll
      for (int c = 0; c < NG; ++c) {
        float erc[W], atc[W];
#pragma unroll
        for (int jj = 0; jj < W; ++jj) { erc[jj] = sh_er[c * 64 * W + lane * W + jj]; atc[jj] = sh_at[c * 64 * W + lane * W + jj]; }
        // (two edges per wave-uniform branch: their butterflies interleave and fill each other's DPP wait states; the odd slot
        // beyond the wave's share works on the re-read last row, its result is dropped)
#pragma unroll
        for (int j2 = 0; j2 < GF_EW; j2 += 2) {
          if (j2 < n_mine) {                          // (wave-uniform)
            float eb[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              const int j = j2 + q;
              float v = 0.f, part = 0.f;
              if (VEC4) {
                const f32x2_t t0 = logit_terms2(raw[j][c].u.x, f32x2_t{erc[0], erc[1 % W]}, f32x2_t{atc[0], atc[1 % W]}, p.slope);
                const f32x2_t t1 = logit_terms2(raw[j][c].u.y, f32x2_t{erc[2 % W], erc[3 % W]}, f32x2_t{atc[2 % W], atc[3 % W]}, p.slope);
                v += t0.x; v += t0.y; v += t1.x; v += t1.y;
              } else v += rbf_hw(atc[0] * rbf_hw(lrelu_f(rbf_hw(__uint_as_float(raw[j][c].u.x << 16) + erc[0]), p.slope)));
              part += v;
              eb[q] = rbf_hw(wave_butterfly_sum(part));
              mine[j] = lane == c ? eb[q] : mine[j];
            }
            mx[c % NH] = max_raw(mx[c % NH], max_raw(eb[0], j2 + 1 < n_mine ? eb[1] : eb[0]));
          }
        }
      }
#pragma unroll
      for (int j = 0; j < GF_EW; ++j) {
        if (j < n_mine && lane < H) {
          const int eidx = k * GF_CHUNK + j * GF_WAVES + wave;
          p.e[(long long)(beg + eidx) * H + lane] = (bf16_t)(__float_as_uint(mine[j]) >> 16);
          sh_e[eidx][lane] = mine[j];
        }
      }
      return;
    }
    float er[NG][W], at[NG][W];
#pragma unroll
    for (int c = 0; c < NG; ++c)
#pragma unroll
      for (int jj = 0; jj < W; ++jj) { er[c][jj] = sh_er[c * 64 * W + lane * W + jj]; at[c][jj] = sh_at[c * 64 * W + lane * W + jj]; }
#pragma unroll
    for (int j = 0; j < GF_EW; ++j) {
      if (j < n_mine) {                                 // (wave-uniform)
        float x[NG][W];
        unpack_row<VEC4, W, NG>(x, raw[j]);
        const int eidx = k * GF_CHUNK + j * GF_WAVES + wave;
        float part[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) part[h] = 0.f;
#pragma unroll
        for (int c = 0; c < NG; ++c) {
          if (hd[c] >= 0) {
            float v = 0.f;
#pragma unroll
            for (int jj = 0; jj < W; ++jj) v += rbf_hw(at[c][jj] * rbf_hw(lrelu_f(rbf_hw(x[c][jj] + er[c][jj]), p.slope)));
#pragma unroll
            for (int h = 0; h < NH; ++h) if (h == hd[c]) part[h] += v;
          }
        }
        wave_sum_heads<NH>(part, H);
        float mine = 0.f;                                // lane h < H: head h's logit of this edge
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          if (h < H) {
            const float eb = rbf_hw(part[h]);
            mine = lane == h ? eb : mine;
            mx[h] = max_raw(mx[h], eb);
          }
        }
        if (lane < H) { p.e[(long long)(beg + eidx) * H + lane] = (bf16_t)(__float_as_uint(mine) >> 16); sh_e[eidx][lane] = mine; }
      }
    }
  };
  for (int k = 0; k + 1 < n_chunks; ++k) {            // (rows of more than 64 in-edges only: these chunks are gathered again in pass 3)
    RawGroup<VEC4> tmp[GF_EW][NG];
    logits_of_chunk(tmp, k);
  }
  RawGroup<VEC4> raw[GF_EW][NG];                        // the last chunk's rows: kept to pass 3
  if (n_chunks > 0) logits_of_chunk(raw, n_chunks - 1);
  if (lane < NH) {
    float m = -__builtin_inff();
#pragma unroll
    for (int h = 0; h < NH; ++h) if (h == lane) m = mx[h];
    sh_max[wave][lane] = m;
  }
  GF_STAMP(2);
  __syncthreads();
  GF_STAMP(3);
  if (G > 1) {                                         // the row's maximum over all its segments
    if (tid < H) {
      float m = sh_max[0][tid];
#pragma unroll
      for (int w2 = 1; w2 < GF_WAVES; ++w2) m = fmaxf(m, sh_max[w2][tid]);
      atomicMax(ws + 4 + tid, f2ord(m));
    }
    row_meet(ws, 0, G, p.err);
    if (tid < H) {
      const float m = ord2f(__hip_atomic_load(ws + 4 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#pragma unroll
      for (int w2 = 0; w2 < GF_WAVES; ++w2) sh_max[w2][tid] = m;
    }
    __syncthreads();
  }
  GF_STAMP(4);
  // ---- pass 2: edge softmax over the logits in LDS (model.py:88-90; [DGL-recalled] four bf16 ops, exact sum), attention dropout
  const int cnt = (end - beg) * H;
  int bad = 0;
  for (int i = tid; i < cnt; i += GF_TPB) {
    const int eidx = i / H, h = i - eidx * H;
    float m = sh_max[0][h];
#pragma unroll
    for (int w2 = 1; w2 < GF_WAVES; ++w2) m = fmaxf(m, sh_max[w2][h]);
    const bf16_t sc = f2bf((float)exp((double)rbf(sh_e[eidx][h] - m)));
    sh_e[eidx][h] = bf2f(sc);                            // (this thread reads it back below)
    const long long fx = bf_to_fixed(sc, FRAC_DST, &bad);
    if (fx) atomicAdd(&sh_sum[h], (unsigned long long)fx);
  }
  if (bad) atomicOr(&sh_bad, bad);
  __syncthreads();
  if (G > 1) {                                         // the row's exact sum: integer adds, any order
    if (tid < H) {
      const unsigned long long v = sh_sum[tid];
      if (v) atomicAdd(reinterpret_cast<unsigned long long*>(ws + 12) + tid, v);
    }
    if (tid == 0 && sh_bad) atomicOr(ws + 3, (unsigned)sh_bad);
    row_meet(ws, 1, G, p.err);
    if (tid < H) sh_sum[tid] = __hip_atomic_load(reinterpret_cast<unsigned long long*>(ws + 12) + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) sh_bad |= (int)__hip_atomic_load(ws + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
  }
  for (int i = tid; i < cnt; i += GF_TPB) {
    const int eidx = i / H, h = i - eidx * H;
    int b2 = 0;
    float ssum = bf2f(fixed_to_bf((long long)sh_sum[h], FRAC_DST, &b2));
    if (sh_bad) ssum = __builtin_nanf("");             // a non-finite logit: the reference's sum, and with it the row, is NaN
    const long long o = (long long)beg * H + i;
    const bf16_t av = f2bf(sh_e[eidx][h] / ssum);
    p.a[o] = av;
    bf16_t cv = av;
    if (p.drop_thresh) {                               // nn.Dropout on a bf16 tensor: a * mask / (1 - p), one rounding
      const bool keep = gf_drop_hash(p.seed, ctr, (uint32_t)o) >= p.drop_thresh;
      cv = keep ? f2bf(bf2f(av) * p.drop_scale) : (bf16_t)0;
      p.ad[o] = cv;
    }
    sh_e[eidx][h] = bf2f(cv);                            // what multiplies el_j in pass 3
  }
  __syncthreads();
  GF_STAMP(5);
  // ---- pass 3: out = sum_j a_ij el_j (model.py:98), fp32 products and sums, one rounding.  The last chunk first: its rows are
  // still in registers; the others (rows of more than 64 in-edges only) are gathered a second time
  auto aggregate_chunk = [&](const RawGroup<VEC4> (&rows)[GF_EW][NG], int k) {
    const int n_mine = wave_edges_in_chunk(n_edges, k, wave);
#pragma unroll
    for (int j = 0; j < GF_EW; ++j) {
      if (j < n_mine) {
        float x[NG][W];
        unpack_row<VEC4, W, NG>(x, rows[j]);
        const int eidx = k * GF_CHUNK + j * GF_WAVES + wave;
#pragma unroll
        for (int c = 0; c < NG; ++c) {
          if (HG || hd[c] >= 0) {
            const float cf = sh_e[eidx][HG ? c : hd[c]];
#pragma unroll
            for (int jj = 0; jj < W; ++jj) acc[c][jj] = __builtin_fmaf(cf, x[c][jj], acc[c][jj]);   // (bf16 x bf16 is exact in fp32: == mul, add)
          }
        }
      }
    }
  };
  if (n_chunks > 0) aggregate_chunk(raw, n_chunks - 1);
  for (int k = 0; k + 1 < n_chunks; ++k) {
    RawGroup<VEC4> tmp[GF_EW][NG];
    ld_wave_rows<VEC4, NG>(tmp, p.feat, p.feat_stride, my_s, k * GF_EW, wave_edges_in_chunk(n_edges, k, wave), coff);
    aggregate_chunk(tmp, k);
  }
#pragma unroll
  for (int c = 0; c < NG; ++c)
#pragma unroll
    for (int j = 0; j < W; ++j) sh_acc[wave][c * 64 * W + lane * W + j] = acc[c][j];
  __syncthreads();
  GF_STAMP(6);
  if (G == 1) {
    for (int col = tid; col < HD; col += GF_TPB) {
      float s = sh_acc[0][col];
#pragma unroll
      for (int w2 = 1; w2 < GF_WAVES; ++w2) s += sh_acc[w2][col];   // fixed order: bitwise reproducible
      p.rst[(long long)row * p.rst_stride + col] = f2bf(s);
    }
    GF_STAMP(7);
    return;
  }
  // a shared row: leave this segment's partial row; whoever arrives last adds the segments up in segment order
  for (int col = tid; col < HD; col += GF_TPB) {
    float s = sh_acc[0][col];
#pragma unroll
    for (int w2 = 1; w2 < GF_WAVES; ++w2) s += sh_acc[w2][col];
    p.seg_part[(long long)vwg * HD + col] = s;
  }
  if (row_last(ws, G, &sh_last)) {
    const long long v0 = vwg - seg_i;
    for (int col = tid; col < HD; col += GF_TPB) {
      float s = 0.f;
      for (int q = 0; q < G; ++q) s += __hip_atomic_load(p.seg_part + (v0 + q) * HD + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      p.rst[(long long)row * p.rst_stride + col] = f2bf(s);
    }
    for (int i = tid; i < GF_ROWWS; i += GF_TPB) ws[i] = 0u;          // the row's words back to zero for the next launch
  }
  GF_STAMP(7);
}

// the dropout stream's launch counter moves on behind the forward kernel (every workgroup has read it by then).  A ticket taken
// by each of the ~5 K workgroups on the counter's own cache line cost 40-70 us per launch (scratch/gatbench.py): every
// workgroup's first load queued behind the other workgroups' atomics
__global__ void k_gat_bump(unsigned long long* ctr) { if (threadIdx.x == 0) *ctr += 1ull; }

// backward by destination: d a, softmax backward, d er and the workgroup's share of d attn (same structure: at most GF_SEG edges
// per workgroup, d a / d e in LDS between the passes; a and a_drop are the forward's outputs: plain loads)
template <bool VEC4, int HG>
__global__ void __launch_bounds__(GF_TPB, 4) k_gat_bwd_dst(GatFused p) {
  constexpr int W = VEC4 ? 4 : 1;
  constexpr int NG = HG ? HG : GF_ITER, NH = HG ? HG : GF_MAXH;
  __shared__ float sh_acc[GF_WAVES][NG * 64 * W], sh_acc2[GF_WAVES][NG * 64 * W];   // the waves' shares of d er_i / of d attn
  __shared__ float sh_c[GF_MAXH][GF_SEG];                  // d a, then d e, of this workgroup's edges, head-major (pass 3 goes head by head)
  __shared__ float sh_t[GF_WAVES][GF_MAXH];
  __shared__ float sh_er[NG * 64 * W];                     // er_i (pass 3 reads it group by group)
  __shared__ int sh_last;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int vwg = blockIdx.x;
  long long* stamps = g_gf_stamps_bwd;
  GF_STAMP(0);
  if (vwg >= *p.n_wg_dev) return;
  int S = p.n_dst;
  if (p.n_dst_dev) { const int t = *p.n_dst_dev; S = t < S ? t : S; }
  const int H = HG ? HG : p.H, D = p.D, HD = H * D;
  const RowSeg rs = row_segment(p, vwg);
  const int row = rs.row, beg = rs.beg, end = rs.end, G = rs.G, seg_i = rs.seg_i;
  GF_STAMP(1);
  if (row >= S) {
    for (int c = tid; c < HD; c += GF_TPB) { p.d_er[(long long)row * p.der_stride + c] = 0; p.dattn_part[(long long)vwg * HD + c] = 0.f; }
    return;
  }
  unsigned* ws = p.rowws + (long long)row * GF_ROWWS;
  const int my_e = beg + (lane >> 3) * GF_CHUNK + (lane & 7) * GF_WAVES + wave;
  const int my_s = (lane < GF_NCH * GF_EW && my_e < end) ? p.src[my_e] : 0;
  const int n_edges = end - beg, n_chunks = (n_edges + GF_CHUNK - 1) / GF_CHUNK;
  int hd[NG], coff[NG];
#pragma unroll
  for (int c = 0; c < NG; ++c) {
    const int col = c * 64 * W + lane * W;
    hd[c] = col < HD ? col / D : -1;
    coff[c] = col < HD ? col : 0;
  }
  RawGroup<VEC4> graw[NG];                               // g_i, packed like the gathered rows (masked columns: hd < 0, never used)
  ld_edge_raw<VEC4, NG>(graw, p.g + (long long)row * p.g_stride, coff);
  for (int col = tid; col < NG * 64 * W; col += GF_TPB) sh_er[col] = col < HD ? bf2f(p.feat[(long long)row * p.feat_stride + col]) : 0.f;
  // ---- pass 1: d a_ij[h] = g_i[h,:] . el_j[h,:] (bf16, like k_gat_edge_dot<1>), through the dropout mask; t[h] = sum a d a
  float tp = 0.f;                                        // lane h: head h's share of t over this wave's edges
  auto dalpha_of_chunk = [&](RawGroup<VEC4> (&raw)[GF_EW][NG], int k) {
    const int n_mine = wave_edges_in_chunk(n_edges, k, wave);
    ld_wave_rows<VEC4, NG>(raw, p.feat, p.feat_stride, my_s, k * GF_EW, n_mine, coff);
    float mine[GF_EW];                                 // lane h: g_i[h,:] . el_j[h,:] of the wave's j-th edge
#pragma unroll
    for (int j = 0; j < GF_EW; ++j) mine[j] = 0.f;
    if (HG) {
#pragma unroll
      for (int c = 0; c < NG; ++c) {
#pragma unroll
        for (int j2 = 0; j2 < GF_EW; j2 += 2) {
          if (j2 < n_mine) {                           // (wave-uniform; two edges per branch: see the forward)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              const int j = j2 + q;
              // g . el on the packed pairs (v_dot2_f32_bf16: exact products, fp32 accumulation)
              float v = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, graw[c].u.x), __builtin_bit_cast(bf16x2_t, raw[j][c].u.x), 0.f, false);
              if (VEC4) v = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, graw[c].u.y), __builtin_bit_cast(bf16x2_t, raw[j][c].u.y), v, false);
              const float sum = wave_butterfly_sum(v);
              mine[j] = lane == c ? sum : mine[j];
            }
          }
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < GF_EW; ++j) {
        if (j < n_mine) {
          float part[NH];
#pragma unroll
          for (int h = 0; h < NH; ++h) part[h] = 0.f;
#pragma unroll
          for (int c = 0; c < NG; ++c) {
            if (hd[c] >= 0) {
              float v = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, graw[c].u.x), __builtin_bit_cast(bf16x2_t, raw[j][c].u.x), 0.f, false);
              if (VEC4) v = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, graw[c].u.y), __builtin_bit_cast(bf16x2_t, raw[j][c].u.y), v, false);
#pragma unroll
              for (int h = 0; h < NH; ++h) if (h == hd[c]) part[h] += v;
            }
          }
          wave_sum_heads<NH>(part, H);
#pragma unroll
          for (int h = 0; h < NH; ++h) if (h == lane) mine[j] = part[h];
        }
      }
    }
    // lane h < H: a / a_drop of head h of the wave's edges (the forward's outputs)
    bf16_t a_e[GF_EW], ad_e[GF_EW];
#pragma unroll
    for (int j = 0; j < GF_EW; ++j) {
      const bool on = j < n_mine && lane < H;
      const long long o = on ? (long long)(beg + k * GF_CHUNK + j * GF_WAVES + wave) * H + lane : 0;
      a_e[j] = p.a[o];
      ad_e[j] = p.drop_thresh ? p.ad[o] : (bf16_t)0x3f80;
    }
#pragma unroll
    for (int j = 0; j < GF_EW; ++j) {
      if (j < n_mine && lane < H) {                    // lane h finishes head h of this edge
        float da = rbf_hw(mine[j]);
        if (p.drop_thresh) da = (ad_e[j] != 0) ? rbf_hw(da * p.drop_scale) : 0.f;   // dropout backward (mask = what the forward kept)
        sh_c[lane][k * GF_CHUNK + j * GF_WAVES + wave] = da;
        tp += bf2f(a_e[j]) * da;
      }
    }
  };
  for (int k = 0; k + 1 < n_chunks; ++k) {
    RawGroup<VEC4> tmp[GF_EW][NG];
    dalpha_of_chunk(tmp, k);
  }
  RawGroup<VEC4> raw[GF_EW][NG];
  if (n_chunks > 0) dalpha_of_chunk(raw, n_chunks - 1);
  if (lane < NH) sh_t[wave][lane] = tp;
  GF_STAMP(2);
  __syncthreads();
  GF_STAMP(3);
  if (G > 1) {                                         // t over the whole row: the segments' totals added in segment order
    float* tseg = p.seg_part + (long long)(*p.n_wg_dev) * HD;         // [n_wg, GF_MAXH] behind the partial rows
    if (tid < H) {
      float t = sh_t[0][tid];
#pragma unroll
      for (int w2 = 1; w2 < GF_WAVES; ++w2) t += sh_t[w2][tid];
      tseg[(long long)vwg * GF_MAXH + tid] = t;
    }
    row_meet(ws, 0, G, p.err);
    if (tid < H) {
      const long long v0 = vwg - seg_i;
      float t = 0.f;
      for (int q = 0; q < G; ++q) t += __hip_atomic_load(tseg + (v0 + q) * GF_MAXH + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      sh_t[0][tid] = t;
#pragma unroll
      for (int w2 = 1; w2 < GF_WAVES; ++w2) sh_t[w2][tid] = 0.f;
    }
    __syncthreads();
  }
  GF_STAMP(4);
  // ---- pass 2: d e = a (d a - t)   (k_gat_softmax<true>); the by-source kernel reads it from memory, pass 3 from LDS
  const int cnt = (end - beg) * H;
  for (int i = tid; i < cnt; i += GF_TPB) {
    const int eidx = i / H, h = i - eidx * H;
    float t = sh_t[0][h];
#pragma unroll
    for (int w2 = 1; w2 < GF_WAVES; ++w2) t += sh_t[w2][h];
    const long long o = (long long)beg * H + i;
    const bf16_t de = f2bf(bf2f(p.a[o]) * (sh_c[h][eidx] - t));
    p.de[o] = de;
    sh_c[h][eidx] = bf2f(de);
  }
  __syncthreads();
  GF_STAMP(5);
  // ---- pass 3: d er_i = sum_j d e attn lrelu'(el_j + er_i);  d attn += d e lrelu(el_j + er_i)   (k_gat_rows<true, false>)
  // Chunk by chunk (the last one first: its rows are still in registers), column group by column group: one group's accumulators and
  // er_i beside the packed rows keep the kernel under 128 registers; a wave adds each chunk's share to its own slice in LDS
#pragma unroll
  for (int c = 0; c < NG; ++c)
#pragma unroll
    for (int jj = 0; jj < W; ++jj) { sh_acc[wave][c * 64 * W + lane * W + jj] = 0.f; sh_acc2[wave][c * 64 * W + lane * W + jj] = 0.f; }
  auto rows_of_chunk = [&](const RawGroup<VEC4> (&raw)[GF_EW][NG], int k) {
    const int n_mine = wave_edges_in_chunk(n_edges, k, wave);
#pragma unroll
    for (int c = 0; c < NG; ++c) {
      float erc[W], da[W], aa[W];
#pragma unroll
      for (int jj = 0; jj < W; ++jj) { erc[jj] = sh_er[c * 64 * W + lane * W + jj]; da[jj] = 0.f; aa[jj] = 0.f; }
      if (HG || hd[c] >= 0) {
#pragma unroll
        for (int j = 0; j < GF_EW; ++j) {
          if (j < n_mine) {
            float x[W];
            if (VEC4) {
              x[0] = __uint_as_float(raw[j][c].u.x << 16); x[1 % W] = __uint_as_float(raw[j][c].u.x & 0xffff0000u);
              x[2 % W] = __uint_as_float(raw[j][c].u.y << 16); x[3 % W] = __uint_as_float(raw[j][c].u.y & 0xffff0000u);
            } else x[0] = __uint_as_float(raw[j][c].u.x << 16);
            // d e times lrelu'(sx) is one of two per-edge values; lrelu(sx) = sx lrelu'(sx): add, compare, select, add, fma
            const float cf = sh_c[HG ? c : hd[c]][k * GF_CHUNK + j * GF_WAVES + wave], cfs = cf * p.slope;
#pragma unroll
            for (int jj = 0; jj < W; ++jj) {
              const float sx = x[jj] + erc[jj];
              const float t = sx > 0.f ? cf : cfs;
              da[jj] += t;
              aa[jj] = __builtin_fmaf(t, sx, aa[jj]);
            }
          }
        }
      }
#pragma unroll
      for (int jj = 0; jj < W; ++jj) { sh_acc[wave][c * 64 * W + lane * W + jj] += da[jj]; sh_acc2[wave][c * 64 * W + lane * W + jj] += aa[jj]; }
    }
  };
  if (n_chunks > 0) rows_of_chunk(raw, n_chunks - 1);
  for (int k = 0; k + 1 < n_chunks; ++k) {
    RawGroup<VEC4> tmp[GF_EW][NG];
    ld_wave_rows<VEC4, NG>(tmp, p.feat, p.feat_stride, my_s, k * GF_EW, wave_edges_in_chunk(n_edges, k, wave), coff);
    rows_of_chunk(tmp, k);
  }
  __syncthreads();
  GF_STAMP(6);
  for (int col = tid; col < HD; col += GF_TPB) {       // the waves' shares in wave order; d er carries attn as a common factor
    float s = sh_acc[0][col], s2 = sh_acc2[0][col];
#pragma unroll
    for (int w2 = 1; w2 < GF_WAVES; ++w2) { s += sh_acc[w2][col]; s2 += sh_acc2[w2][col]; }
    s *= bf2f(p.attn[col]);
    if (G == 1) p.d_er[(long long)row * p.der_stride + col] = f2bf(s);
    else p.seg_part[(long long)vwg * HD + col] = s;
    p.dattn_part[(long long)vwg * HD + col] = s2;      // (one share per workgroup: the reduction adds them in workgroup order)
  }
  if (G > 1 && row_last(ws, G, &sh_last)) {            // d er of a shared row: the segments' partial rows in segment order
    const long long v0 = vwg - seg_i;
    for (int col = tid; col < HD; col += GF_TPB) {
      float s = 0.f;
      for (int q = 0; q < G; ++q) s += __hip_atomic_load(p.seg_part + (v0 + q) * HD + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      p.d_er[(long long)row * p.der_stride + col] = f2bf(s);
    }
    for (int i = tid; i < GF_ROWWS; i += GF_TPB) ws[i] = 0u;
  }
  GF_STAMP(7);
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
  p->wg_row = reinterpret_cast<const int4*>(a->wg_row); p->n_wg_dev = a->n_wg_dev; p->rowws = (unsigned*)a->row_ws; p->seg_part = a->seg_part; p->err = a->err;
  if (!a->wg_row || !a->n_wg_dev || !a->row_ws || !a->seg_part || a->cap_wg < a->n_dst) return false;
  *vec4 = v4;
  return true;
}

}  // namespace

extern "C" {

int bliss_gat_fused_supported(int32_t heads, int32_t head_dim) {
  if (heads <= 0 || heads > GF_MAXH || head_dim <= 0) return 0;
  return heads * head_dim <= GF_ITER * 64 * (head_dim % 4 == 0 ? 4 : 1);
}

int bliss_gat_segment_edges(void) { return GF_SEG; }

int bliss_gat_fused_stamps(long long* stamps, long long* stamps_bwd) {
  const int rc = (int)hipMemcpyToSymbol(HIP_SYMBOL(g_gf_stamps), &stamps, sizeof(stamps));
  return rc ? rc : (int)hipMemcpyToSymbol(HIP_SYMBOL(g_gf_stamps_bwd), &stamps_bwd, sizeof(stamps_bwd));
}

int bliss_gat_segments(const int32_t* indptr, int32_t n_dst, int32_t cap_wg, int32_t* wg_row, int32_t* n_wg_dev, int32_t* err, void* stream) {
  if (!indptr || !wg_row || !n_wg_dev || n_dst <= 0 || cap_wg < n_dst) return BLISS_EINVAL;
  k_gat_segments<<<1, 1024, 0, (hipStream_t)stream>>>(indptr, n_dst, cap_wg, reinterpret_cast<int4*>(wg_row), n_wg_dev, err);
  return (int)hipGetLastError();
}

int bliss_gat_fused_fwd(const bliss_gat_fused_t* args, void* stream) {
  GatFused p;
  bool v4;
  if (!gf_fill(args, &p, &v4) || !p.src || !p.e || !p.a || !p.rst) return BLISS_EINVAL;
  if (p.drop_thresh && (!p.ad || !p.ctr)) return BLISS_EINVAL;
  if (args->drop_p < 0.f || args->drop_p >= 1.f) return BLISS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int hg = (v4 && p.D == 256 && (p.H == 1 || p.H == 2 || p.H == 4)) ? p.H : 0;
  const int grid = args->cap_wg;
  if (hg == 4) k_gat_fwd<true, 4><<<grid, GF_TPB, 0, st>>>(p);
  else if (hg == 2) k_gat_fwd<true, 2><<<grid, GF_TPB, 0, st>>>(p);
  else if (hg == 1) k_gat_fwd<true, 1><<<grid, GF_TPB, 0, st>>>(p);
  else if (v4) k_gat_fwd<true, 0><<<grid, GF_TPB, 0, st>>>(p);
  else k_gat_fwd<false, 0><<<grid, GF_TPB, 0, st>>>(p);
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
  const int grid = args->cap_wg;
  if (hg == 4) k_gat_bwd_dst<true, 4><<<grid, GF_TPB, 0, st>>>(p);
  else if (hg == 2) k_gat_bwd_dst<true, 2><<<grid, GF_TPB, 0, st>>>(p);
  else if (hg == 1) k_gat_bwd_dst<true, 1><<<grid, GF_TPB, 0, st>>>(p);
  else if (v4) k_gat_bwd_dst<true, 0><<<grid, GF_TPB, 0, st>>>(p);
  else k_gat_bwd_dst<false, 0><<<grid, GF_TPB, 0, st>>>(p);
  // the shares of d attn: one per (virtual) workgroup; rows of capacity padding wrote zeros
  const int nb = (grid + DA_ROWS - 1) / DA_ROWS, HD = p.H * p.D;
  k_gat_dattn_stage1<<<dim3(nb, (HD + 255) / 256), 256, 0, st>>>(p.dattn_part, grid, p.n_wg_dev, HD, block_sums);
  k_gat_dattn_stage2<<<(HD + 63) / 64, 256, 0, st>>>(block_sums, nb, HD, d_attn);
  (void)ticket;
  return (int)hipGetLastError();
}

}  // extern "C"
