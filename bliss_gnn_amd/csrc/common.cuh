// Shared device helpers for the BLISS-GNN gfx950 kernels.
//
// Arithmetic contract (mirrors oracle/numerics.py, which mirrors what torch's CPU kernels do
// for the reference's bf16 tensors):
//   * element-wise op: read bf16 -> fp32 IEEE op (correctly rounded + - * / sqrt) -> ONE
//     round-to-nearest-even to bf16;
//   * segment reduction (DGL's copy_e_sum, bandit_sampler.py:67,73,129,316): terms are added
//     EXACTLY as 64-bit fixed point (integer adds commute => any schedule, any wave order, any
//     number of GPUs gives the same bits) and the exact sum is rounded once to bf16.
// This file is compiled with -ffp-contract=off so no multiply-add is ever fused across these
// rounding points.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "bliss_gnn.h"

#define BLISS_WAVE 64

// fixed-point formats (value = integer * 2^-FRAC); see oracle/numerics.py
#define FRAC_DST 40
#define FRAC_SRC 44
#define FRAC_BLK 36

// error bits, accumulated into LayerCounts::err
#define BLISS_ERR_CAP_FRONTIER 1   // frontier larger than 2^31-1 positions
#define BLISS_ERR_CAP_CAND     2   // candidate capacity exceeded
#define BLISS_ERR_CAP_KEPT     4   // kept-node capacity exceeded
#define BLISS_ERR_CAP_EDGES    8   // block-edge capacity exceeded
#define BLISS_ERR_NONFINITE   16   // a non-finite / negative term reached an exact reduction
#define BLISS_ERR_FIXED_RANGE 32   // an exact sum left its fixed-point range
#define BLISS_ERR_CAP_SEEDS   64   // more seeds than the per-layer seed capacity
#define BLISS_ERR_RNG_STREAM 128   // the random stream ran short (capacity) or the generator did not make progress
#define BLISS_ERR_FLAG_TIMEOUT 256 // bliss_flag_wait gave up: the producer's flag never came

typedef uint16_t bf16_t;   // raw bfloat16 bits

__device__ __forceinline__ float bf2f(bf16_t b) { return __uint_as_float(((uint32_t)b) << 16); }

// fp32 -> bf16 round-to-nearest-even, NaN stays NaN (quiet), like torch's c10::BFloat16.
__device__ __forceinline__ bf16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)0x7fc0;   // NaN (c10 returns 0x7FC0)
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }   // round-trip through bf16

// F.normalize(row, p=1) element by element (bandit_sampler.py:249): x / max(norm, eps), rounded to bf16.  ONE definition
// for the pass that rewrites a row (exp3.hip) and for the readers that apply a pending pass on the fly (sampler.hip:
// BLISS_NORM_DEFER), so both produce the same bits.  pend = 0x10000 | bf16 bits of the norm, 0 = nothing pending.
// The state word of a row (bliss_norm_state_t in bliss_gnn.h): bits 0-15 norm, bit 16 "a pass is pending", bit 17 "the row's
// current values live in the ALTERNATE buffer" (the pending pass is out of place: it reads the current buffer, writes the other
// one and flips bit 17 when it is done, so it may run beside readers); 8 bytes on: the alternate buffer's distance in elements.
#define NORM_PEND_MASK 0x1ffff
#define NORM_CUR_ALT 0x20000
template <typename T>
__device__ __forceinline__ T* norm_state_row(T* w, const int* state_ptr, int* pend_out) {
  if (!state_ptr) { *pend_out = 0; return w; }
  const int st = __hip_atomic_load(state_ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  *pend_out = st & NORM_PEND_MASK;
  return (st & NORM_CUR_ALT) ? w + *reinterpret_cast<const long long*>(state_ptr + 2) : w;
}
__device__ __forceinline__ float renorm_denom(int pend) { return rbf(fmaxf(bf2f((bf16_t)(pend & 0xffff)), 1e-12f)); }
__device__ __forceinline__ bf16_t renorm_bf16(bf16_t x, float denom) { return f2bf(bf2f(x) / denom); }
__device__ __forceinline__ bf16_t renorm_pending(bf16_t x, int pend, float denom) { return pend ? renorm_bf16(x, denom) : x; }

// bf16 -> signed 64-bit fixed point with `frac` fractional bits; truncates below 2^-frac.
// Sets *bad on a non-finite input or a magnitude that does not fit.
__device__ __forceinline__ int64_t bf_to_fixed(bf16_t b, int frac, int* bad) {
  uint32_t e = (b >> 7) & 0xff, m = b & 0x7f;
  if (e == 255) { *bad |= BLISS_ERR_NONFINITE; return 0; }
  if (e == 0) e = 1; else m |= 0x80;
  int shift = (int)e - 134 + frac;
  uint64_t mag;
  if (shift >= 0) {
    if (shift > 55) { *bad |= BLISS_ERR_FIXED_RANGE; return 0; }
    mag = (uint64_t)m << shift;
  } else {
    mag = (-shift >= 8) ? 0 : ((uint64_t)m >> (-shift));
  }
  return (b & 0x8000) ? -(int64_t)mag : (int64_t)mag;
}

// signed fixed point -> bf16, exact round-to-nearest-even.
__device__ __forceinline__ bf16_t fixed_to_bf(int64_t n, int frac, int* bad) {
  if (n == 0) return 0;
  uint32_t sign = n < 0 ? 0x8000u : 0u;
  uint64_t mag = n < 0 ? (uint64_t)(-n) : (uint64_t)n;
  int msb = 63 - __clzll((long long)mag);
  uint64_t q;
  if (msb > 7) {
    int sh = msb - 7;
    q = mag >> sh;
    uint64_t rem = mag & ((1ull << sh) - 1), half = 1ull << (sh - 1);
    if (rem > half || (rem == half && (q & 1))) q += 1;
  } else {
    q = mag << (7 - msb);
  }
  int e = msb - frac + 127;
  if (msb - frac + 127 <= 0) {
    // below the normal range (only the block-floating column sums get here: frac > 126): round at the subnormal spacing
    // 2^-133 like IEEE (torch's CPU ops do not flush); q == 128 is the smallest normal number, same encoding
    const int sh = frac - 133;
    if (sh <= 0) return (bf16_t)(sign | (uint32_t)(mag << (-sh)));
    uint64_t qs = sh > 63 ? 0 : mag >> sh;
    if (sh <= 63) {
      const uint64_t rem = mag & ((1ull << sh) - 1), half = 1ull << (sh - 1);
      if (rem > half || (rem == half && (qs & 1))) qs += 1;
    }
    return (bf16_t)(sign | (uint32_t)qs);
  }
  if (q >= 256) { q >>= 1; e += 1; }
  if (e >= 255) { *bad |= BLISS_ERR_FIXED_RANGE; return 0; }
  return (bf16_t)(sign | ((uint32_t)e << 7) | ((uint32_t)q & 0x7f));
}

// ---- wide form of the two helpers above, for the one sum whose terms span bf16's whole range (sum_j w_ij of a seed column
// once the bandit has concentrated a row, bandit_sampler.py:129).  Found by the long reference run of tests/golden
// (collapse0_*): a column holding 0.4766, 0.4902 and a few weights around 1e-14 sums to a hair above a rounding tie; the
// tiny terms fell below the accumulator's last bit, the sum looked like an exact tie and rounded to even -- the reference's
// (exact) sum rounds up.  So the bits a term loses below the accumulator's unit go to a SECOND accumulator 40 bits finer,
// and whatever falls below even that sets a sticky flag that breaks ties upward (all terms are >= 0).  The common case --
// nothing lost anywhere in the column -- costs one ballot.
__device__ __forceinline__ long long bf_to_fixed_wide(bf16_t b, int frac, long long* lo, int* sticky, int* bad) {
  uint32_t e = (b >> 7) & 0xff, m = b & 0x7f;
  if (e == 255) { *bad |= BLISS_ERR_NONFINITE; return 0; }
  if (e == 0) e = 1; else m |= 0x80;
  const int shift = (int)e - 134 + frac;
  const bool neg = (b & 0x8000) != 0;
  if (shift >= 0) {
    if (shift > 55) { *bad |= BLISS_ERR_FIXED_RANGE; return 0; }
    const long long mag = (long long)((uint64_t)m << shift);
    return neg ? -mag : mag;
  }
  const int rs = -shift;
  const uint32_t top = rs >= 8 ? 0u : (m >> rs);
  const uint32_t r = rs >= 8 ? m : (m & ((1u << rs) - 1u));          // the bits below the accumulator's unit
  if (r) {
    long long l;
    if (rs <= 40) l = (long long)((uint64_t)r << (40 - rs));
    else {
      const int d = rs - 40;
      l = d >= 8 ? 0 : (long long)(r >> d);
      if (d >= 8 || (r & ((1u << d) - 1u))) *sticky = 1;
    }
    *lo += neg ? -l : l;
  }
  return neg ? -(long long)top : (long long)top;
}

// (hi * 2^40 + lo) * 2^-(frac + 40) -> bf16, round to nearest even on the EXACT value; sticky = more (positive) bits below
__device__ __forceinline__ bf16_t fixed_wide_to_bf(int64_t hi, int64_t lo, int sticky, int frac, int* bad) {
  if (lo == 0 && !sticky) return fixed_to_bf(hi, frac, bad);
  __int128 t = ((__int128)hi << 40) + (__int128)lo;
  if (t == 0) return 0;
  const uint32_t sign = t < 0 ? 0x8000u : 0u;
  const unsigned __int128 mag = t < 0 ? (unsigned __int128)(-t) : (unsigned __int128)t;
  const uint64_t mh = (uint64_t)(mag >> 64), ml = (uint64_t)mag;
  const int msb = mh ? 127 - __clzll((long long)mh) : 63 - __clzll((long long)ml);
  const int f = frac + 40;
  auto round_shift = [&](int sh) -> uint64_t {           // mag >> sh, nearest even, ties broken upward when sticky
    if (sh <= 0) return (uint64_t)(mag << (-sh));
    if (sh > 126) return 0;
    uint64_t q = (uint64_t)(mag >> sh);
    const unsigned __int128 rem = mag & ((((unsigned __int128)1) << sh) - 1), half = ((unsigned __int128)1) << (sh - 1);
    if (rem > half || (rem == half && (sticky || (q & 1)))) q += 1;
    return q;
  };
  if (msb - f + 127 <= 0) return (bf16_t)(sign | (uint32_t)round_shift(f - 133));     // subnormal result (see fixed_to_bf)
  uint64_t q = round_shift(msb - 7);
  int e = msb - f + 127;
  if (q >= 256) { q >>= 1; e += 1; }
  if (e >= 255) { *bad |= BLISS_ERR_FIXED_RANGE; return 0; }
  return (bf16_t)(sign | ((uint32_t)e << 7) | ((uint32_t)q & 0x7f));
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & (BLISS_WAVE - 1); }

__device__ __forceinline__ int64_t shfl_up_i64(int64_t v, int d) {
  int lo = __shfl_up((int)(v & 0xffffffffll), d), hi = __shfl_up((int)(v >> 32), d);
  return ((int64_t)hi << 32) | (uint32_t)lo;
}

// ---- wave reductions on the DPP data path (no LDS round trip per step, unlike __shfl = ds_bpermute) ----
// inclusive scan inside each row of 16 lanes (row_shr 1, 2, 4, 8), then row 0 -> 1 and 2 -> 3 (row_bcast:15), then rows
// 0-1 -> 2-3 (row_bcast:31): lane 63 ends up with the wave total.  Lanes without a source receive 0 (old = 0).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_mov0(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false); }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ long long dpp_mov0_i64(long long v) {
  const int lo = dpp_mov0<CTRL, ROW_MASK>((int)(v & 0xffffffffll)), hi = dpp_mov0<CTRL, ROW_MASK>((int)(v >> 32));
  return ((long long)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ int wave_total_i32(int v) {
  v += dpp_mov0<0x111, 0xf>(v); v += dpp_mov0<0x112, 0xf>(v); v += dpp_mov0<0x114, 0xf>(v); v += dpp_mov0<0x118, 0xf>(v);
  v += dpp_mov0<0x142, 0xa>(v); v += dpp_mov0<0x143, 0xc>(v);
  return __builtin_amdgcn_readlane(v, 63);
}
// wave maximum of non-negative values (lanes without a DPP source contribute 0)
__device__ __forceinline__ int wave_max_u31(int v) {
  v = max(v, dpp_mov0<0x111, 0xf>(v)); v = max(v, dpp_mov0<0x112, 0xf>(v)); v = max(v, dpp_mov0<0x114, 0xf>(v));
  v = max(v, dpp_mov0<0x118, 0xf>(v)); v = max(v, dpp_mov0<0x142, 0xa>(v)); v = max(v, dpp_mov0<0x143, 0xc>(v));
  return __builtin_amdgcn_readlane(v, 63);
}
// Block-floating exact sums (sum_j w_ij of a seed column, bandit_sampler.py:129): the EXP3 weights of a row span many
// orders of magnitude once the bandit has concentrated it, so a column's terms are scaled by 2^s, s = 126 - (largest
// biased exponent in the column), before they are added as Q.FRAC_DST integers: the sum is exact relative to the column's
// largest term (terms more than ~2^-33 below it truncate), wherever the column sits in bf16's range.  s = 0 for a column
// holding a weight >= 0.5; the oracle does the same (numerics.exact_segment_sum_rel).
__device__ __forceinline__ int bf_exp_field(bf16_t b) { const int e = (b >> 7) & 0xff; return e == 0 ? 1 : e; }
__device__ __forceinline__ int rel_frac(int frac, int emax) { const int s = 126 - emax; return frac + (s > 0 ? s : 0); }

__device__ __forceinline__ long long wave_total_i64(long long v) {
  v += dpp_mov0_i64<0x111, 0xf>(v); v += dpp_mov0_i64<0x112, 0xf>(v); v += dpp_mov0_i64<0x114, 0xf>(v);
  v += dpp_mov0_i64<0x118, 0xf>(v); v += dpp_mov0_i64<0x142, 0xa>(v); v += dpp_mov0_i64<0x143, 0xc>(v);
  const int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffll), 63), hi = __builtin_amdgcn_readlane((int)(v >> 32), 63);
  return ((long long)hi << 32) | (unsigned)lo;
}
// all lanes with key >= 0 carry the same key (the common case: a wave's 64 edges lie inside one seed column)
__device__ __forceinline__ bool wave_single_key(int key, int* k0) {
  const unsigned long long act = __ballot(key >= 0);
  if (!act) { *k0 = -1; return true; }
  *k0 = __builtin_amdgcn_readlane(key, __ffsll((long long)act) - 1);
  return __ballot(key >= 0 && key != *k0) == 0ull;
}

// Wave-level segmented sum of 64-bit terms keyed by a NON-DECREASING int key; the last lane of
// every key run adds the run's total to acc[key] with one atomic.  Lanes with key < 0 are idle.
__device__ __forceinline__ void wave_segsum_atomic_i64(int key, int64_t v, unsigned long long* acc) {
  const int lane = lane_id();
  int k0;
  if (wave_single_key(key, &k0)) {                   // one segment: a DPP reduction and one atomic
    if (k0 < 0) return;
    const long long tot = wave_total_i64(key >= 0 ? v : 0);
    if (lane == 0 && tot != 0) atomicAdd(acc + k0, (unsigned long long)tot);
    return;
  }
#pragma unroll
  for (int d = 1; d < BLISS_WAVE; d <<= 1) {
    int64_t vu = shfl_up_i64(v, d);
    int ku = __shfl_up(key, d);
    if (lane >= d && ku == key) v += vu;
  }
  int kn = __shfl_down(key, 1);
  if (key >= 0 && (lane == BLISS_WAVE - 1 || kn != key) && v != 0)
    atomicAdd(acc + key, (unsigned long long)v);
}
__device__ __forceinline__ void wave_segsum_atomic_i32(int key, int v, int* acc) {
  const int lane = lane_id();
  int k0;
  if (wave_single_key(key, &k0)) {
    if (k0 < 0) return;
    const int tot = wave_total_i32(key >= 0 ? v : 0);
    if (lane == 0 && tot != 0) atomicAdd(acc + k0, tot);
    return;
  }
#pragma unroll
  for (int d = 1; d < BLISS_WAVE; d <<= 1) {
    int vu = __shfl_up(v, d);
    int ku = __shfl_up(key, d);
    if (lane >= d && ku == key) v += vu;
  }
  int kn = __shfl_down(key, 1);
  if (key >= 0 && (lane == BLISS_WAVE - 1 || kn != key) && v != 0) atomicAdd(acc + key, v);
}

// Block-wide exclusive scan of one int per thread (blockDim.x <= 1024, multiple of 64).
// Returns the exclusive prefix; *total gets the block sum.  `sh` needs 17 ints of LDS.
__device__ __forceinline__ int block_excl_scan(int v, int* sh, int* total) {
  const int lane = lane_id(), wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  int inc = v;
#pragma unroll
  for (int d = 1; d < BLISS_WAVE; d <<= 1) {
    int t = __shfl_up(inc, d);
    if (lane >= d) inc += t;
  }
  __syncthreads();                       // protect sh from a previous use
  if (lane == BLISS_WAVE - 1) sh[wid] = inc;
  __syncthreads();
  if (wid == 0) {
    int w = lane < nw ? sh[lane] : 0, winc = w;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
      int t = __shfl_up(winc, d);
      if (lane >= d) winc += t;
    }
    if (lane < nw) sh[lane] = winc - w;  // exclusive wave offsets
    if (lane == nw - 1) sh[16] = winc;
  }
  __syncthreads();
  *total = sh[16];
  return sh[wid] + inc - v;
}

// Largest k in [0, n) with seg_ptr[k] <= pos (seg_ptr non-decreasing, seg_ptr[0] = 0).
// Empty segments are skipped correctly because we take the LAST k whose start <= pos.
__device__ __forceinline__ int find_segment(const int* __restrict__ seg_ptr, int n, int pos) {
  int lo = 0, hi = n;           // invariant: seg_ptr[lo] <= pos < seg_ptr[hi]
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (seg_ptr[mid] <= pos) lo = mid; else hi = mid;
  }
  return lo;
}

// Segment (= seed column) of 64 CONSECUTIVE frontier positions base .. base+63, one per lane, without a
// per-lane binary search: the wave loads the next 64 column boundaries once (coalesced) and every lane
// counts the boundaries at or before its position with 6 shuffles.  k_hint (wave-uniform): a segment
// known to start at or before `base` (carried from the previous 64 positions), or -1.
// Returns the lane's segment; lanes at or past seg_ptr[S] (= E) get S.  Updates k_hint for base+64.
__device__ __forceinline__ int wave_segment(const int* __restrict__ seg_ptr, int S, int base, int* k_hint) {
  const int lane = lane_id();
  int k0;
  if (*k_hint < 0) {
    k0 = find_segment(seg_ptr, S + 1, base);          // uniform address: one broadcast transaction per step
  } else {
    const int idx = *k_hint + 1 + lane;
    const int b = idx <= S ? seg_ptr[idx] : 0x7fffffff;
    const int adv = __popcll(__ballot(b <= base));
    k0 = adv < 64 ? *k_hint + adv : find_segment(seg_ptr, S + 1, base);
  }
  const int idx = k0 + 1 + lane;
  const int d = (idx <= S ? seg_ptr[idx] : 0x7fffffff) - base;   // > 0, non-decreasing over lanes
  int k;
  if (__shfl(d, 63) > 63) {                           // the window covers every boundary inside this span
    int pos = 0;
#pragma unroll
    for (int step = 32; step >= 1; step >>= 1) {
      const int v = __shfl(d, pos + step - 1);
      if (v <= lane) pos += step;
    }
    k = k0 + pos;
  } else {                                            // > 64 boundaries in 64 positions (runs of empty columns)
    k = find_segment(seg_ptr, S + 1, base + lane);
  }
  *k_hint = __shfl(k, 63);
  if (*k_hint >= S) *k_hint = S - 1 >= 0 ? S - 1 : 0;
  return k;
}

// One lane: wait until the streaming generator (rng.hip) has produced the numbers a layer needs, hand the layer its
// offset into the stream and account for the numbers it will consume.  ctl: [0] progress, [1] stop_at, [2] pos,
// [3] error, [4] base.  Bounded spin: never hangs the GPU.
__device__ __forceinline__ void rng_stream_acquire(int* ctl, int C, int* layer_off, int is_last, int cap_total) {
  // ctl[5] = "this control block belongs to the generator of THIS call": set by whoever initialised it (k_rng_ctl_init on the
  // consumer's own stream, or k_mt19937_chain on the generator's stream -- then nothing else orders that kernel before this
  // one), cleared by the call's last layer.  Bounded like the wait below.
  {
    long long spins = 0;
    while (__hip_atomic_load(ctl + 5, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > (1ll << 22)) { atomicOr(ctl + 3, 2); break; }
    }
  }
  const int pos = ctl[2];
  int need = pos + C;
  if (need > cap_total) { need = cap_total; atomicOr(ctl + 3, 1); }
  long long spins = 0;
  while (__hip_atomic_load(ctl + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
    __builtin_amdgcn_s_sleep(8);
    if (++spins > (1ll << 22)) { atomicOr(ctl + 3, 2); break; }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  *layer_off = ctl[4] + pos;
  ctl[2] = need;
  if (is_last) {
    __hip_atomic_store(ctl + 1, need, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(ctl + 5, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Per-layer sizes kept on the device (the S/E/C/K/B symbols of SURVEY.md); layout = the ABI's.
typedef bliss_layer_counts_t LayerCounts;
