// EXP3 bandit state for gfx950: reward, weight update, exact L1 renormalisation.
//
// Replaces BanditLadiesSampler.calculate_alpha (SAGE/GCN branch), calculate_rewards and
// update_exp3_weights (bandit_sampler.py:140-249) -- ~15 DGL/ATen launches per block plus an
// 18*|E_g|-byte whole-row normalise per layer per step -- with ONE edge-parallel launch per block
// and a renormalisation that is bit-exact to F.normalize(p=1) yet touches the row only when the
// bf16 norm differs from 1.0:
//   the exact sum of the row is carried as three signed 64-bit limbs (value * 2^64, 32-bit digits);
//   the update kernel adds digits(new) - digits(old) for the B edges it rewrites, so the norm
//   bf16(exact sum) is known in O(B); dividing a bf16 by 1.0 is the identity, so when the norm
//   rounds to 1.0 -- the steady state, the row sums to 1 +- 0.4% -- the 2 x |E_g| pass is skipped
//   on the device with identical bits.
#include "common.cuh"
#include "bliss_gnn.h"
#include "prof.h"
#include <cstdlib>

namespace {

#define E3_TPB 256
// the row is streamed once: non-temporal accesses keep it from evicting what the sampler works on (A/B on the Reddit-like
// loop: steps with a pass 0.92 -> 0.89 ms; alone on the chip the pass is a little slower this way, 85 -> 90 us)
#ifndef NORM_NT
#define NORM_NT 1
#endif
#ifndef NORM_UNROLL
#define NORM_UNROLL 4
#endif

// digits of (bf16 value * 2^64), truncated below 2^-64; non-negative finite input required
__device__ __forceinline__ void row_digits(bf16_t b, int64_t d[3], int* bad) {
  uint32_t e = (b >> 7) & 0xff, m = b & 0x7f;
  d[0] = d[1] = d[2] = 0;
  if (e == 255 || (b & 0x8000 && (b & 0x7fff))) { *bad |= BLISS_ERR_NONFINITE; return; }
  if (e == 0) e = 1; else m |= 0x80;
  int shift = (int)e - 134 + 64;                 // value*2^64 = m << shift
  if (shift > 80) { *bad |= BLISS_ERR_FIXED_RANGE; return; }
  if (shift < 0) { d[0] = (-shift >= 8) ? 0 : (int64_t)(m >> (-shift)); return; }
  // an 8-bit mantissa shifted into three 32-bit digits: it straddles at most two of them
  const int idx = shift >> 5, off = shift & 31;
  const uint64_t v = (uint64_t)m << off;         // < 2^39
  const int64_t lo = (int64_t)(v & 0xffffffffu), hi = (int64_t)(v >> 32);
  if (idx == 0) { d[0] = lo; d[1] = hi; }
  else if (idx == 1) { d[1] = lo; d[2] = hi; }
  else { d[2] = (int64_t)v; }                    // the top digit is not reduced mod 2^32 (as before)
}

__device__ __forceinline__ int64_t wave_sum_i64(int64_t v) { return wave_total_i64(v); }

// The exact sum lives in BLISS_ROWSUM_SLOTS replicas of three limbs (its value is the sum over the replicas): every wave
// adds its total to the replica picked by its global wave id, so that thousands of waves do not serialise on three
// addresses (a same-address memory-side atomic costs ~8 ns; 49 K of them made one renormalisation pass take 470 us).
#define ROWSUM_SLOTS BLISS_ROWSUM_SLOTS
__device__ __forceinline__ void flush_digits(int64_t d[3], int64_t* limbs) {
  const int slot = (int)((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) % ROWSUM_SLOTS);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    int64_t s = wave_sum_i64(d[i]);
    if (lane_id() == 0 && s != 0) atomicAdd((unsigned long long*)(limbs + 3 * slot + i), (unsigned long long)s);
  }
}
__device__ __forceinline__ void gather_limbs(const int64_t* limbs, int64_t out[3]) {
  out[0] = out[1] = out[2] = 0;
  for (int s = 0; s < ROWSUM_SLOTS; ++s) { out[0] += limbs[3 * s]; out[1] += limbs[3 * s + 1]; out[2] += limbs[3 * s + 2]; }
}

// exact (l0 + l1*2^32 + l2*2^64) * 2^-64 -> bf16 RNE
__device__ bf16_t limbs_to_bf16(const int64_t* replicas, int* bad) {
  int64_t limbs[3];
  gather_limbs(replicas, limbs);
  __int128 t = (__int128)limbs[0] + ((__int128)limbs[1] << 32) + ((__int128)limbs[2] << 64);
  if (t < 0) { *bad |= BLISS_ERR_FIXED_RANGE; return 0; }
  if (t == 0) return 0;
  unsigned __int128 u = (unsigned __int128)t;
  uint64_t hi = (uint64_t)(u >> 64), lo = (uint64_t)u;
  int msb = hi ? 127 - __clzll((long long)hi) : 63 - __clzll((long long)lo);
  unsigned __int128 q;
  if (msb > 7) {
    int sh = msb - 7;
    q = u >> sh;
    unsigned __int128 rem = u & ((((unsigned __int128)1) << sh) - 1), half = ((unsigned __int128)1) << (sh - 1);
    if (rem > half || (rem == half && (q & 1))) q += 1;
  } else q = u << (7 - msb);
  int e = msb - 64 + 127;
  if (q >= 256) { q >>= 1; e += 1; }
  if (e <= 0 || e >= 255) { *bad |= BLISS_ERR_FIXED_RANGE; return 0; }
  return (bf16_t)(((uint32_t)e << 7) | ((uint32_t)q & 0x7f));
}

__device__ __forceinline__ float nan_to_num_posinf0(float x) {   // torch.nan_to_num(x, posinf=0) on bf16
  if (x != x) return 0.f;
  if (x == __builtin_inff()) return 0.f;
  if (x == -__builtin_inff()) return -3.3895313892515355e38f;     // lowest bf16
  return x;
}

// one thread per block edge; wg / nwg: this workgroup's index and count among those working on THIS block (a launch may
// cover several blocks, see k_exp3_update_multi)
__device__ __forceinline__ void exp3_update_body(const int64_t* __restrict__ g_indptr, const bf16_t* __restrict__ edge_w,
                                                 bf16_t* w_row, int64_t* row_sum, const int* __restrict__ blk_indptr,
                                                 const int* __restrict__ blk_src, const int* __restrict__ blk_dst,
                                                 const int* __restrict__ blk_pos, const bf16_t* __restrict__ q_ij,
                                                 const bf16_t* __restrict__ node_prob, const bf16_t* __restrict__ embed_norm,
                                                 const bf16_t* __restrict__ alpha_in, const int* __restrict__ dst_nid,
                                                 const int* __restrict__ n_edges_dev, float delta_f,
                                                 bf16_t* __restrict__ rewards_out, bf16_t* __restrict__ factor_out,
                                                 int apply, int* err, int wg, int nwg, int bound) {
  int B = *n_edges_dev;
  int bad = 0;
  if (B > bound) { B = bound; bad |= BLISS_ERR_CAP_EDGES; }        // the caller's arrays end at `bound`
  int64_t dg[3] = {0, 0, 0};
  for (int base = wg * E3_TPB + (threadIdx.x & ~63); base < B; base += nwg * E3_TPB) {
    const int e = base + lane_id();
    if (e < B) {
      const int i = blk_dst[e], j = blk_src[e], pos = blk_pos[e];
      const float alpha = bf2f(alpha_in ? alpha_in[e] : edge_w[pos]);                  // :157 mfg.edata['w']
      const float k_i = rbf((float)(blk_indptr[i + 1] - blk_indptr[i]));               // :180 in_degrees().bfloat16()
      float adk = rbf(rbf(alpha * alpha) / k_i);                                       // :186 e_div_v(alpha**2, k_i)
      adk = nan_to_num_posinf0(adk);                                                   // :187
      const float hn = bf2f(embed_norm[j]), q = bf2f(q_ij[e]);
      const float hq = rbf(rbf(hn * hn) / rbf(q * q));                                 // :189 u_div_e
      const float r = rbf(adk * hq);                                                   // :191
      if (rewards_out) rewards_out[e] = f2bf(r);                                       // :193
      const int dn = dst_nid[i];
      const float n_i = rbf((float)(g_indptr[dn + 1] - g_indptr[dn]));                 // :223
      const float r_hat = rbf(r / bf2f(node_prob[j]));                                 // :240 e_div_u
      // :242 e_mul_v(rewards_hat, delta / n_i): `delta / n_i` is n_i.reciprocal() * delta on a bf16 tensor
      float dr = rbf(r_hat * rbf(rbf(1.0f / n_i) * delta_f));
      if (dr > 1.0f) dr = 1.0f;                                                        // :244
      const float ex = rbf((float)exp((double)dr));                                    // :246 torch.exp on bf16
      if (factor_out) factor_out[e] = f2bf(ex);
      if (!apply) continue;
      const bf16_t w_old = w_row[pos];
      const bf16_t w_new = f2bf(bf2f(w_old) * ex);                                     // :248
      if (w_new != w_old) {
        w_row[pos] = w_new;
        int64_t a[3], b[3];
        row_digits(w_new, a, &bad);
        row_digits(w_old, b, &bad);
        dg[0] += a[0] - b[0]; dg[1] += a[1] - b[1]; dg[2] += a[2] - b[2];
      }
    }
  }
  flush_digits(dg, row_sum);
  if (bad) atomicOr(err, bad);
}

__global__ void __launch_bounds__(E3_TPB) k_exp3_update(const int64_t* __restrict__ g_indptr, const bf16_t* __restrict__ edge_w,
                                                       bf16_t* w_row, int64_t* row_sum, const int* __restrict__ blk_indptr,
                                                       const int* __restrict__ blk_src, const int* __restrict__ blk_dst,
                                                       const int* __restrict__ blk_pos, const bf16_t* __restrict__ q_ij,
                                                       const bf16_t* __restrict__ node_prob, const bf16_t* __restrict__ embed_norm,
                                                       const bf16_t* __restrict__ alpha_in, const int* __restrict__ dst_nid,
                                                       const int* __restrict__ n_edges_dev, float delta_f,
                                                       bf16_t* __restrict__ rewards_out, bf16_t* __restrict__ factor_out,
                                                       int apply, int* err, int edges_bound) {
  exp3_update_body(g_indptr, edge_w, w_row, row_sum, blk_indptr, blk_src, blk_dst, blk_pos, q_ij, node_prob, embed_norm, alpha_in,
                   dst_nid, n_edges_dev, delta_f, rewards_out, factor_out, apply, err, blockIdx.x, gridDim.x, edges_bound);
}

// all blocks of a step in ONE launch (each launch costs >= 4 us inside a graph; the blocks are independent: every
// layer has its own weight row)
struct Exp3Multi {
  bliss_exp3_block_t blk[BLISS_EXP3_MAX_BLOCKS];
  int grid_begin[BLISS_EXP3_MAX_BLOCKS + 1];
  int n;
  int* done_flag;          // NORM_DECIDE: raised (bliss_flag_wait's protocol) once every row has been decided, or null
};
__global__ void __launch_bounds__(E3_TPB) k_exp3_update_multi(const int64_t* __restrict__ g_indptr, const bf16_t* __restrict__ edge_w,
                                                             const Exp3Multi m, float delta_f, int* err) {
  int b = 0;
  while (b + 1 < m.n && (int)blockIdx.x >= m.grid_begin[b + 1]) ++b;
  const bliss_exp3_block_t& k = m.blk[b];
  int pend_unused;
  bf16_t* w_row = norm_state_row((bf16_t*)k.w_pos, k.norm_pend, &pend_unused);        // (a pending pass has completed by now)
  exp3_update_body(g_indptr, edge_w, w_row, k.row_sum, k.blk_indptr, k.blk_src, k.blk_dst, k.blk_pos, (const bf16_t*)k.q_ij,
                   (const bf16_t*)k.node_prob, (const bf16_t*)k.embed_norm, (const bf16_t*)k.alpha_or_null, k.dst_nid, k.n_edges_dev,
                   delta_f, (bf16_t*)k.rewards_out, nullptr, 1, err, (int)blockIdx.x - m.grid_begin[b], m.grid_begin[b + 1] - m.grid_begin[b],
                   k.edges_bound);
}

__global__ void __launch_bounds__(E3_TPB) k_exp3_apply(bf16_t* w_row, int64_t* row_sum, const int* __restrict__ pos,
                                                      const bf16_t* __restrict__ factor, const int* __restrict__ n_dev, int* err) {
  const int n = *n_dev;
  int bad = 0;
  int64_t dg[3] = {0, 0, 0};
  for (int e = blockIdx.x * E3_TPB + threadIdx.x; e < n; e += gridDim.x * E3_TPB) {
    const int p = pos[e];
    const bf16_t w_old = w_row[p];
    const bf16_t w_new = f2bf(bf2f(w_old) * bf2f(factor[e]));                          // :248
    if (w_new != w_old) {
      w_row[p] = w_new;
      int64_t a[3], b[3];
      row_digits(w_new, a, &bad);
      row_digits(w_old, b, &bad);
      dg[0] += a[0] - b[0]; dg[1] += a[1] - b[1]; dg[2] += a[2] - b[2];
    }
  }
  flush_digits(dg, row_sum);
  if (bad) atomicOr(err, bad);
}

// ---- the update lists of ALL ranks and ALL blocks in one launch (replicas of the bandit state) ----
// Lists are applied rank after rank (the bf16 products do not commute, and every replica must end with the same bits): a
// grid barrier separates the ranks -- the launch gaps of n_ranks x n_blocks small kernels were the cost of the old path.
// A position may occur in the lists of several ranks, written by workgroups on different XCDs (whose L2s are not coherent
// for plain stores), and two neighbouring positions share a 32-bit word: every update is a compare-and-swap on that word
// at agent scope (performed at the memory side), so no cache write-back is needed at the barrier.
// Grid: <= APPLY_MAX_WGS co-resident workgroups.  bar[0] = arrivals (monotone within a launch), bar[1] = exit ticket;
// the last workgroup to leave resets both.
#define APPLY_MAX_WGS 256
struct ApplyLists {
  bf16_t* w_row[BLISS_EXP3_MAX_BLOCKS];
  int64_t* row_sum[BLISS_EXP3_MAX_BLOCKS];
  int pos_off[BLISS_EXP3_MAX_BLOCKS], fac_off[BLISS_EXP3_MAX_BLOCKS], cnt_off[BLISS_EXP3_MAX_BLOCKS], bound[BLISS_EXP3_MAX_BLOCKS];
  int n_blocks, n_ranks;
  long long rank_stride;          // int32 words between two ranks' packed buffers
};

__global__ void __launch_bounds__(E3_TPB) k_exp3_apply_ranks(ApplyLists m, const int* __restrict__ gathered, int* bar, int* err) {
  int bad = 0;
  int64_t dg[BLISS_EXP3_MAX_BLOCKS][3];
#pragma unroll
  for (int b = 0; b < BLISS_EXP3_MAX_BLOCKS; ++b) dg[b][0] = dg[b][1] = dg[b][2] = 0;
  const int G = gridDim.x;
  for (int r = 0; r < m.n_ranks; ++r) {
    const int* base = gathered + (long long)r * m.rank_stride;
#pragma unroll
    for (int b = 0; b < BLISS_EXP3_MAX_BLOCKS; ++b) {
      if (b >= m.n_blocks) break;
      int n = base[m.cnt_off[b]];
      if (n > m.bound[b]) { n = m.bound[b]; bad |= BLISS_ERR_CAP_EDGES; }     // that rank's list was cut short: flagged everywhere
      const int* pos = base + m.pos_off[b];
      const bf16_t* fac = reinterpret_cast<const bf16_t*>(base) + m.fac_off[b];
      for (int e = blockIdx.x * E3_TPB + threadIdx.x; e < n; e += G * E3_TPB) {
        const float f = bf2f(fac[e]);
        const uintptr_t a = (uintptr_t)(m.w_row[b] + pos[e]);
        unsigned* word = reinterpret_cast<unsigned*>(a & ~(uintptr_t)3);
        const int sh = (a & 2) ? 16 : 0;
        unsigned cur = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (;;) {
          const bf16_t w_old = (bf16_t)((cur >> sh) & 0xffffu);
          const bf16_t w_new = f2bf(bf2f(w_old) * f);                                   // :248
          if (w_new == w_old) break;
          const unsigned want = (cur & ~(0xffffu << sh)) | ((unsigned)w_new << sh);
          if (__hip_atomic_compare_exchange_strong(word, &cur, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            int64_t x[3], y[3];
            row_digits(w_new, x, &bad);
            row_digits(w_old, y, &bad);
            dg[b][0] += x[0] - y[0]; dg[b][1] += x[1] - y[1]; dg[b][2] += x[2] - y[2];
            break;
          }
        }
      }
    }
    if (r + 1 < m.n_ranks) {                             // every update of rank r has been performed (the CAS returned)
      __syncthreads();
      if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(bar, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        // bounded like every other device-side wait here (k_flag_wait, rng_stream_acquire): a workgroup that never arrives
        // (the grid not co-resident after all) flags the step invalid instead of hanging the GPU
        long long spins = 0;
        while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (r + 1) * G) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > (1ll << 24)) { bad |= BLISS_ERR_FLAG_TIMEOUT; break; }
        }
      }
      __syncthreads();
    }
  }
#pragma unroll
  for (int b = 0; b < BLISS_EXP3_MAX_BLOCKS; ++b)
    if (b < m.n_blocks) flush_digits(dg[b], m.row_sum[b]);
  if (bad) atomicOr(err, bad);
  __syncthreads();
  if (threadIdx.x == 0 && __hip_atomic_fetch_add(bar + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == G - 1) {
    __hip_atomic_store(bar, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(bar + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// positions + true counts of all blocks into the packed exchange buffer, ONE launch, only the entries that exist
struct PackLists {
  const int* pos[BLISS_EXP3_MAX_BLOCKS];
  const int* n_dev[BLISS_EXP3_MAX_BLOCKS];
  int pos_off[BLISS_EXP3_MAX_BLOCKS], cnt_off[BLISS_EXP3_MAX_BLOCKS], bound[BLISS_EXP3_MAX_BLOCKS];
  int n_blocks;
};
__global__ void __launch_bounds__(E3_TPB) k_pack_lists(PackLists m, int* __restrict__ buf) {
  for (int b = 0; b < m.n_blocks; ++b) {
    const int n_true = *m.n_dev[b];
    const int n = n_true < m.bound[b] ? n_true : m.bound[b];
    if (blockIdx.x == 0 && threadIdx.x == 0) buf[m.cnt_off[b]] = n_true;      // uncapped: a list cut short is flagged by the apply
    const int* __restrict__ src = m.pos[b];
    int* __restrict__ dst = buf + m.pos_off[b];
    for (int e = blockIdx.x * E3_TPB + threadIdx.x; e < n; e += gridDim.x * E3_TPB) dst[e] = src[e];
  }
}

__global__ void __launch_bounds__(E3_TPB) k_row_sum(const bf16_t* __restrict__ w, int64_t n, int64_t* limbs, int* err) {
  int bad = 0;
  int64_t dg[3] = {0, 0, 0};
  for (int64_t i = (int64_t)blockIdx.x * E3_TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * E3_TPB) {
    int64_t a[3];
    row_digits(w[i], a, &bad);
    dg[0] += a[0]; dg[1] += a[1]; dg[2] += a[2];
  }
  flush_digits(dg, limbs);
  if (bad && err) atomicOr(err, bad);
}

__global__ void k_zero_i64(int64_t* p, int n) { if ((int)threadIdx.x < n) p[threadIdx.x] = 0; }

// F.normalize(row, p=1) in ONE launch.  scratch (int64[BLISS_NORM_SCRATCH], zero between calls): [0] = norm bits |
// skip << 16 | err << 20 (for the host), [1] ticket, [2 ..] replicas of the renormalised row's exact sum.  Every workgroup
// derives the norm from the (read-only) exact row sum; if it is 1.0 nothing is touched.  Otherwise the last workgroup to
// finish installs the new exact sum -- the others have all read row_sum long before (they read it first thing).
// norm_src (optional): take the norm from THESE limbs instead of row_sum -- a destination-range shard normalises its part
// of the row by the exact sum over ALL shards (an all-reduce of the limb arrays; integer sums commute), and row_sum still
// receives the exact sum of this shard's renormalised part.
// two elements of the pass the slow way (by division; sum through the general three-digit path)
struct PairSlow { uint32_t word; int bad; int64_t d0, d1, d2; };
__device__ __attribute__((noinline)) PairSlow renorm_pair_slow(uint32_t word, float denom) {
  PairSlow r;
  r.bad = 0;
  const bf16_t lo = renorm_bf16((bf16_t)(word & 0xffffu), denom), hi = renorm_bf16((bf16_t)(word >> 16), denom);
  int64_t a[3], b[3];
  row_digits(lo, a, &r.bad);
  row_digits(hi, b, &r.bad);
  r.d0 = a[0] + b[0]; r.d1 = a[1] + b[1]; r.d2 = a[2] + b[2];
  r.word = (uint32_t)lo | ((uint32_t)hi << 16);
  return r;
}

// mode NORM_NOW: decide and rewrite in this launch.  NORM_DECIDE (one workgroup per row): only derive the norm and leave
// 0x10000 | norm bits in *pend when the row needs the pass (0 otherwise) -- the readers of the row apply the division on
// the fly (renorm_pending, common.cuh) until NORM_APPLY (any number of workgroups, any later launch) rewrites the row from
// *pend, installs the new exact sum and clears *pend.  DECIDE + APPLY leave the bits of NOW.
enum { NORM_NOW = 0, NORM_DECIDE = 1, NORM_APPLY = 2 };
__device__ __forceinline__ void normalize_row_body(bf16_t* w, int64_t n, int64_t* row_sum, int64_t* scratch, bf16_t* norm_out, int wg, int nwg,
                                                   const int64_t* norm_src = nullptr, int mode = NORM_NOW, int* pend = nullptr) {
  __shared__ int sh_norm, sh_last, sh_state;
  if (threadIdx.x == 0) {
    if (mode == NORM_APPLY) {
      const int pd = __hip_atomic_load(pend, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      sh_state = pd;
      sh_norm = (pd & 0x10000) ? (pd & 0xffff) : 0x3f80;       // nothing pending: leave like a row whose norm is 1.0
    } else {
      int bad = 0;
      const bf16_t nb = limbs_to_bf16(norm_src ? norm_src : row_sum, &bad);
      sh_norm = (int)nb | (bad << 20);
      if (wg == 0) {
        scratch[0] = (int64_t)nb | ((int64_t)(nb == 0x3f80) << 16) | ((int64_t)bad << 20);
        if (norm_out) *norm_out = nb;
        if (mode == NORM_DECIDE) {                             // (which buffer is current stays as it is)
          const int cur = __hip_atomic_load(pend, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & NORM_CUR_ALT;
          __hip_atomic_store(pend, cur | ((nb == 0x3f80) ? 0 : (0x10000 | (int)nb)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
  }
  __syncthreads();
  const bf16_t nb = (bf16_t)(sh_norm & 0xffff);
  if (nb == 0x3f80 || mode == NORM_DECIDE) return;   // x / 1.0 == x : the row is already normalised, bit for bit
  // NORM_APPLY works out of place: from the row's current buffer into the other one (the readers of the current one are
  // not disturbed; they divide on the fly), and the last workgroup makes the other one current
  const bf16_t* src = w;
  if (mode == NORM_APPLY) {
    const long long alt = *reinterpret_cast<const long long*>(pend + 2);
    if (sh_state & NORM_CUR_ALT) src = w + alt; else w = w + alt;          // w = destination from here on
  }
  const float denom = renorm_denom((int)nb);       // F.normalize: norm.clamp_min(eps)
  int bad = 0;
  int64_t dg[3] = {0, 0, 0};
  // The quotient without dividing.  Division by a fixed denominator commutes with powers of two as long as nothing leaves
  // the normal range, and so does the rounding to bf16: the quotient's BITS are x's bits plus a constant that depends only on
  // x's 7 mantissa bits -- 128 constants per pass, taken from renorm_bf16 itself on [1, 2).  (The pass was bound by its
  // arithmetic, not by HBM: 8 IEEE divisions per 16 bytes.)  Valid for a positive normal x whose quotient is normal too, with a
  // binade to spare on either side; everything else -- zero, subnormal, negative, non-finite, the edges of the range --
  // takes the division.
  __shared__ int sh_tbl[128];
  if (threadIdx.x < 128) {
    const bf16_t x0 = (bf16_t)(0x3f80u | threadIdx.x);
    sh_tbl[threadIdx.x] = ((int)renorm_bf16(x0, denom) - (int)x0) & 0xffff;      // (added mod 2^16)
  }
  const int de_lo = (int)(renorm_bf16((bf16_t)0x3f80u, denom) >> 7) - 127;      // exponent change of the smallest mantissa ...
  const int de_hi = (int)(renorm_bf16((bf16_t)0x3fffu, denom) >> 7) - 127;      // ... and of the largest (monotonic in between)
  // x's sign|exponent field must lie in [e_lo, e_hi]; both quotients above normal and finite, else no x qualifies
  const bool tbl_ok = de_lo > -126 && de_hi < 126 && de_lo <= de_hi;
  const uint32_t e_lo = (uint32_t)max(2, 2 - de_lo), e_hi = (uint32_t)min(253, 253 - de_hi);
  // ... and for the two-at-a-time path below also the QUOTIENT's exponent field in [70, 125]: its value * 2^64 is then
  // m << s with s = field - 70 in [0, 56), which that path adds up in seven per-thread LDS counters, one per 8 bits of s
  // (term = m << (s & 7) < 2^15, a thread adds far fewer than 2^17 of them: no overflow, no 64-bit shifts, no conflicts: bank = lane)
  const int f_lo = max((int)e_lo, 70 - de_lo), f_hi = min((int)e_hi, 125 - de_hi);
  const bool pair_ok = tbl_ok && e_hi >= e_lo && f_hi >= f_lo;
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  const unsigned short f_base = pair_ok ? (unsigned short)f_lo : (unsigned short)0xffffu, f_span = pair_ok ? (unsigned short)(f_hi - f_lo) : (unsigned short)0;
  const u16x2 FLO2 = {f_base, f_base}, SPAN2 = {f_span, f_span};
  __shared__ unsigned sh_bins[7][E3_TPB];
#pragma unroll
  for (int a = 0; a < 7; ++a) sh_bins[a][threadIdx.x] = 0;
  __syncthreads();
  // (everything else: one at a time, by division, sum through the general three-digit path)
  auto one = [&](bf16_t x) {
    const bf16_t v = renorm_bf16(x, denom);        // :249 input / denom
    int64_t a[3];
    row_digits(v, a, &bad);
    dg[0] += a[0]; dg[1] += a[1]; dg[2] += a[2];
    return v;
  };
  // When the pass does run it is a pure stream over the row (2 x 2 bytes per edge, 0.46 GB on the Reddit-like graph):
  // 16-byte accesses, several in flight per thread; a scalar head / tail brings the row to 16-byte alignment.
  const int64_t head = min(n, (int64_t)(((16 - ((uintptr_t)w & 15)) & 15) >> 1));
  const int64_t nvec = (n - head) / 8;
  const int64_t gtid = (int64_t)wg * E3_TPB + threadIdx.x, gsz = (int64_t)nwg * E3_TPB;
  if (gtid < head) w[gtid] = one(src[gtid]);
  for (int64_t i = head + nvec * 8 + gtid; i < n; i += gsz) w[i] = one(src[i]);
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  u32x4* wv = reinterpret_cast<u32x4*>(w + head);
  const u32x4* sv = reinterpret_cast<const u32x4*>(src + head);        // (both buffers share the 16-byte phase: alt % 8 == 0)
#pragma unroll NORM_UNROLL
  for (int64_t i = gtid; i < nvec; i += gsz) {
#if NORM_NT
    u32x4 x = __builtin_nontemporal_load(sv + i);
#else
    u32x4 x = sv[i];
#endif
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t word = x[k];
      const u16x2 W = __builtin_bit_cast(u16x2, word);
      const u16x2 D = (W >> 7) - FLO2;               // sign|exponent fields relative to the range (wrapping)
      if (__builtin_bit_cast(uint32_t, __builtin_elementwise_max(D, SPAN2)) == __builtin_bit_cast(uint32_t, SPAN2)) {
        // both in range: quotient bits = bits + constant(mantissa), two at a time
        const uint32_t t = (uint32_t)sh_tbl[word & 0x7fu] | ((uint32_t)sh_tbl[(word >> 16) & 0x7fu] << 16);
        const u16x2 V = W + __builtin_bit_cast(u16x2, t);
        const uint32_t v2 = __builtin_bit_cast(uint32_t, V);
        const u16x2 S = (V >> 7) - (u16x2){70, 70};
        const u16x2 M = __builtin_bit_cast(u16x2, (v2 & 0x007f007fu) | 0x00800080u);
        const u16x2 T = M << (S & (u16x2){7, 7});
        const u16x2 A = S >> 3;
        __hip_atomic_fetch_add(&sh_bins[A.x][threadIdx.x], (unsigned)T.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&sh_bins[A.y][threadIdx.x], (unsigned)T.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        x[k] = v2;
      } else {                                     // rare: a call, so that the loop body stays small
        const PairSlow r = renorm_pair_slow(word, denom);
        x[k] = r.word; bad |= r.bad;
        dg[0] += r.d0; dg[1] += r.d1; dg[2] += r.d2;
      }
    }
#if NORM_NT == 1
    __builtin_nontemporal_store(x, wv + i);
#else
    wv[i] = x;
#endif
  }
  {                                                // the thread's seven counters -> digits
    uint64_t a = 0, b = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) a += (uint64_t)sh_bins[k][threadIdx.x] << (8 * k);           // units of 2^-64
#pragma unroll
    for (int k = 4; k < 7; ++k) b += (uint64_t)sh_bins[k][threadIdx.x] << (8 * (k - 4));     // units of 2^-32
    dg[0] += (int64_t)(a & 0xffffffffull); dg[1] += (int64_t)(a >> 32);
    dg[1] += (int64_t)(b & 0xffffffffull); dg[2] += (int64_t)(b >> 32);
  }
  flush_digits(dg, scratch + 2);
  if (bad) atomicOr((unsigned long long*)scratch, (unsigned long long)bad << 20);
  // The last workgroup installs the new exact sum.  Everything it needs from the others went through memory-side
  // atomics, which are device-coherent by themselves: draining them (vmcnt) before the ticket is all the release this
  // hand-off needs.  (__threadfence() here cost a full L2 write-back per workgroup: 470 us per pass on a 15 M-edge row.)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) sh_last = (atomicAdd((unsigned long long*)(scratch + 1), 1ull) == (unsigned long long)nwg - 1);
  __syncthreads();
  if (sh_last) {
    for (int k = threadIdx.x; k < 3 * ROWSUM_SLOTS; k += E3_TPB) {
      row_sum[k] = (int64_t)__hip_atomic_load((unsigned long long*)(scratch + 2 + k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store((unsigned long long*)(scratch + 2 + k), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0) {
      __hip_atomic_store((unsigned long long*)(scratch + 1), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // (NORM_APPLY: the state word is flipped by k_norm_flip, a launch of its own behind this one -- the other workgroups'
      // stores are only guaranteed to have left their XCD's L2 when this kernel has ended)
    }
  }
}

__global__ void __launch_bounds__(E3_TPB) k_normalize_row(bf16_t* w, int64_t n, int64_t* row_sum, int64_t* scratch, bf16_t* norm_out,
                                                          const int64_t* norm_src) {
  normalize_row_body(w, n, row_sum, scratch, norm_out, blockIdx.x, gridDim.x, norm_src);
}
// the rows of all layers in one launch, each row's norm from its own all-reduced limbs (rows spread over several shards)
__global__ void __launch_bounds__(E3_TPB) k_normalize_rows_global(const Exp3Multi m, int64_t n, int per_row, const int64_t* __restrict__ limbs,
                                                                  int64_t limb_stride) {
  const int r = blockIdx.x / per_row;
  const bliss_exp3_block_t& k = m.blk[r];
  normalize_row_body((bf16_t*)k.w_pos, n, k.row_sum, k.scratch, (bf16_t*)k.norm_out, (int)blockIdx.x - r * per_row, per_row, limbs + r * limb_stride);
}
// the rows of all layers in one launch: gridDim.x / n_rows workgroups per row
__global__ void __launch_bounds__(E3_TPB) k_normalize_rows(const Exp3Multi m, int64_t n, int per_row, int mode) {
  const int r = blockIdx.x / per_row;
  const bliss_exp3_block_t& k = m.blk[r];
  normalize_row_body((bf16_t*)k.w_pos, n, k.row_sum, k.scratch, (bf16_t*)k.norm_out, (int)blockIdx.x - r * per_row, per_row, nullptr, mode,
                     k.norm_pend);
  if (mode == NORM_DECIDE && m.done_flag && threadIdx.x == 0) {
    // one workgroup per row here; the last one to have stored its row's state word tells the stream that runs the pass
    // (row 0's pass ticket is idle during a decide launch and doubles as this one)
    unsigned long long* ticket = (unsigned long long*)(m.blk[0].scratch + 1);
    if (atomicAdd(ticket, 1ull) == (unsigned long long)m.n - 1) {
      __hip_atomic_store(ticket, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(m.done_flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// after a NORM_APPLY launch: the rows that had a pass pending now live in their other buffer, nothing pending
__global__ void k_norm_flip(const Exp3Multi m) {
  const int r = threadIdx.x;
  if (r >= m.n) return;
  int* st = m.blk[r].norm_pend;
  const int v = __hip_atomic_load(st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (v & 0x10000) __hip_atomic_store(st, (v & NORM_CUR_ALT) ^ NORM_CUR_ALT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// w_pos[p] = bf16(1 / bf16(indeg(dst(p))))          bandit_sampler.py:20-27
__global__ void __launch_bounds__(E3_TPB) k_normalized_edata(const int64_t* __restrict__ indptr, int num_nodes, bf16_t* __restrict__ out) {
  const int lane = lane_id();
  for (int v = blockIdx.x * (E3_TPB / 64) + (threadIdx.x >> 6); v < num_nodes; v += gridDim.x * (E3_TPB / 64)) {
    const int64_t b = indptr[v], e = indptr[v + 1];
    // exact sum of (e-b) ones = the integer, rounded once to bf16 by the int->float->bf16 chain (exact below 2^24)
    const bf16_t val = f2bf(1.0f / rbf((float)(e - b)));
    for (int64_t p = b + lane; p < e; p += 64) out[p] = val;
  }
}

}  // namespace

// workgroups per row of the F.normalize pass (BLISS_NORM_WGS overrides, for measurements)
static int norm_wgs() {
  static const int v = [] { const char* e = getenv("BLISS_NORM_WGS"); int x = e ? atoi(e) : 0; return x > 0 ? x : 1024; }();
  return v;
}

extern "C" {

int bliss_exp3_update(const bliss_graph_t* g, const void* edge_w_pos, void* w_pos, int64_t* row_sum,
                      const int32_t* blk_indptr, const int32_t* blk_src, const int32_t* blk_dst,
                      const int32_t* blk_pos, const void* q_ij, const void* node_prob, const void* embed_norm,
                      const void* alpha_or_null, const int32_t* dst_nid, int32_t n_dst, const int32_t* n_edges_dev,
                      int32_t edges_bound, float delta_f, void* rewards_out, void* factor_out, int apply,
                      int32_t* err, void* stream) {
  if (!g || !w_pos || !row_sum || !blk_indptr || !blk_src || !blk_dst || !blk_pos || !q_ij || !node_prob ||
      !embed_norm || !dst_nid || !n_edges_dev || !err || (!edge_w_pos && !alpha_or_null))
    return BLISS_EINVAL;
  (void)n_dst;
  if (edges_bound <= 0) return 0;
  int grid = (edges_bound + E3_TPB - 1) / E3_TPB;
  if (grid > 2048) grid = 2048;
  hipStream_t st = (hipStream_t)stream;
  PROF_LAUNCH(BK_EXP3_UPDATE, st, k_exp3_update<<<grid, E3_TPB, 0, st>>>(g->indptr, (const bf16_t*)edge_w_pos, (bf16_t*)w_pos, row_sum, blk_indptr,
                                                          blk_src, blk_dst, blk_pos, (const bf16_t*)q_ij, (const bf16_t*)node_prob,
                                                          (const bf16_t*)embed_norm, (const bf16_t*)alpha_or_null, dst_nid,
                                                          n_edges_dev, delta_f, (bf16_t*)rewards_out, (bf16_t*)factor_out, apply, err, edges_bound));
  return (int)hipGetLastError();
}

static int exp3_step(const bliss_graph_t* g, const void* edge_w_pos, const bliss_exp3_block_t* blocks, int32_t n_blocks,
                     float delta_f, int32_t* err, void* stream, bool defer, int32_t* done_flag, bool normalize = true) {
  if (!g || !blocks || n_blocks <= 0 || n_blocks > BLISS_EXP3_MAX_BLOCKS || !err) return BLISS_EINVAL;
  Exp3Multi m;
  m.n = n_blocks;
  m.done_flag = done_flag;
  int total = 0;
  for (int i = 0; i < n_blocks; ++i) {
    const bliss_exp3_block_t& k = blocks[i];
    if (!k.w_pos || !k.row_sum || !k.scratch || !k.blk_indptr || !k.blk_src || !k.blk_dst || !k.blk_pos || !k.q_ij || !k.node_prob ||
        !k.embed_norm || !k.dst_nid || !k.n_edges_dev || (!edge_w_pos && !k.alpha_or_null) || k.edges_bound < 0 || (defer && !k.norm_pend))
      return BLISS_EINVAL;
    m.blk[i] = k;
    m.grid_begin[i] = total;
    int gb = (k.edges_bound + E3_TPB - 1) / E3_TPB;
    if (gb < 1) gb = 1;
    if (gb > 2048) gb = 2048;
    total += gb;
  }
  m.grid_begin[n_blocks] = total;
  hipStream_t st = (hipStream_t)stream;
  PROF_LAUNCH(BK_EXP3_UPDATE, st, k_exp3_update_multi<<<total, E3_TPB, 0, st>>>(g->indptr, (const bf16_t*)edge_w_pos, m, delta_f, err));
  if (!normalize) return (int)hipGetLastError();
  int64_t per_row = (g->num_edges + E3_TPB * 8 - 1) / (E3_TPB * 8);
  // usually every workgroup returns at once (norm == 1.0), but a row that does need the pass streams 4 bytes per edge and
  // wants the whole chip: 1024 workgroups per row
  if (per_row > norm_wgs()) per_row = norm_wgs();
  if (per_row < 1) per_row = 1;
  if (defer) per_row = 1;                              // only the decision: one workgroup per row
  PROF_LAUNCH(BK_NORMALIZE, st, k_normalize_rows<<<(int)(per_row * n_blocks), E3_TPB, 0, st>>>(m, g->num_edges, (int)per_row, defer ? NORM_DECIDE : NORM_NOW));
  return (int)hipGetLastError();
}

int bliss_exp3_step(const bliss_graph_t* g, const void* edge_w_pos, const bliss_exp3_block_t* blocks, int32_t n_blocks,
                    float delta_f, int32_t* err, void* stream) {
  return exp3_step(g, edge_w_pos, blocks, n_blocks, delta_f, err, stream, false, nullptr);
}

int bliss_exp3_update_blocks(const bliss_graph_t* g, const void* edge_w_pos, const bliss_exp3_block_t* blocks, int32_t n_blocks,
                             float delta_f, int32_t* err, void* stream) {
  return exp3_step(g, edge_w_pos, blocks, n_blocks, delta_f, err, stream, false, nullptr, false);
}

int bliss_exp3_step_deferred(const bliss_graph_t* g, const void* edge_w_pos, const bliss_exp3_block_t* blocks, int32_t n_blocks,
                             float delta_f, int32_t* done_flag, int32_t* err, void* stream) {
  return exp3_step(g, edge_w_pos, blocks, n_blocks, delta_f, err, stream, true, done_flag);
}

int bliss_exp3_normalize_pending(const bliss_exp3_block_t* rows, int32_t n_rows, int64_t num_edges, void* stream) {
  if (!rows || n_rows <= 0 || n_rows > BLISS_EXP3_MAX_BLOCKS || num_edges < 0) return BLISS_EINVAL;
  Exp3Multi m;
  m.n = n_rows;
  m.done_flag = nullptr;
  for (int i = 0; i < n_rows; ++i) {
    if (!rows[i].w_pos || !rows[i].row_sum || !rows[i].scratch || !rows[i].norm_pend) return BLISS_EINVAL;
    m.blk[i] = rows[i];
    m.grid_begin[i] = 0;
  }
  int64_t per_row = (num_edges + E3_TPB * 8 - 1) / (E3_TPB * 8);
  if (per_row > norm_wgs()) per_row = norm_wgs();
  if (per_row < 1) per_row = 1;
  hipStream_t st = (hipStream_t)stream;
  PROF_LAUNCH(BK_NORMALIZE, st, k_normalize_rows<<<(int)(per_row * n_rows), E3_TPB, 0, st>>>(m, num_edges, (int)per_row, NORM_APPLY));
  k_norm_flip<<<1, 64, 0, st>>>(m);
  return (int)hipGetLastError();
}

int bliss_exp3_apply(void* w_pos, int64_t* row_sum, const int32_t* pos, const void* factor, const int32_t* n_dev,
                     int32_t n_bound, int32_t* err, void* stream) {
  if (!w_pos || !row_sum || !n_dev || !err) return BLISS_EINVAL;
  if (n_bound <= 0) return 0;
  if (!pos || !factor) return BLISS_EINVAL;
  int grid = (n_bound + E3_TPB - 1) / E3_TPB;
  if (grid > 2048) grid = 2048;
  hipStream_t st = (hipStream_t)stream;
  PROF_LAUNCH(BK_EXP3_APPLY, st, k_exp3_apply<<<grid, E3_TPB, 0, st>>>((bf16_t*)w_pos, row_sum, pos, (const bf16_t*)factor, n_dev, err));
  return (int)hipGetLastError();
}

int bliss_pack_lists(const bliss_pack_lists_t* lists, int32_t* buf, void* stream) {
  if (!lists || !buf || lists->n_blocks < 1 || lists->n_blocks > BLISS_EXP3_MAX_BLOCKS) return BLISS_EINVAL;
  PackLists m;
  int64_t most = 1;
  for (int b = 0; b < lists->n_blocks; ++b) {
    if (!lists->pos[b] || !lists->n_dev[b] || lists->bound[b] < 0) return BLISS_EINVAL;
    m.pos[b] = lists->pos[b]; m.n_dev[b] = lists->n_dev[b];
    m.pos_off[b] = lists->pos_off_words[b]; m.cnt_off[b] = lists->count_off_words[b]; m.bound[b] = lists->bound[b];
    if (lists->bound[b] > most) most = lists->bound[b];
  }
  m.n_blocks = lists->n_blocks;
  int grid = (int)((most + E3_TPB * 4 - 1) / (E3_TPB * 4));
  if (grid > 512) grid = 512;
  if (grid < 1) grid = 1;
  k_pack_lists<<<grid, E3_TPB, 0, (hipStream_t)stream>>>(m, buf);
  return (int)hipGetLastError();
}

int bliss_exp3_apply_ranks(const bliss_exp3_rank_lists_t* lists, const int32_t* gathered, int32_t* barrier, int32_t* err, void* stream) {
  if (!lists || !gathered || !barrier || !err || lists->n_blocks < 1 || lists->n_blocks > BLISS_EXP3_MAX_BLOCKS || lists->n_ranks < 1 ||
      lists->rank_stride_words <= 0)
    return BLISS_EINVAL;
  ApplyLists m;
  int64_t most = 1;
  for (int b = 0; b < lists->n_blocks; ++b) {
    if (!lists->w_pos[b] || !lists->row_sum[b] || lists->bound[b] < 0) return BLISS_EINVAL;
    m.w_row[b] = (bf16_t*)lists->w_pos[b]; m.row_sum[b] = lists->row_sum[b];
    m.pos_off[b] = lists->pos_off_words[b]; m.fac_off[b] = lists->factor_off_bf16[b]; m.cnt_off[b] = lists->count_off_words[b];
    m.bound[b] = lists->bound[b];
    if (lists->bound[b] > most) most = lists->bound[b];
  }
  m.n_blocks = lists->n_blocks; m.n_ranks = lists->n_ranks; m.rank_stride = lists->rank_stride_words;
  int grid = (int)((most + E3_TPB - 1) / E3_TPB);
  // all workgroups must be resident together (grid barrier): at most a quarter of what the device can hold of this kernel
  // (the loop's other streams keep CUs busy beside it), never more than APPLY_MAX_WGS
  static int resident_cap = 0;
  if (!resident_cap) {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_exp3_apply_ranks, E3_TPB, 0) != hipSuccess || hipGetDevice(&dev) != hipSuccess ||
        hipGetDeviceProperties(&prop, dev) != hipSuccess)
      return BLISS_EINVAL;
    long long cap = (long long)per_cu * prop.multiProcessorCount / 4;
    resident_cap = (int)(cap < 1 ? 1 : (cap > APPLY_MAX_WGS ? APPLY_MAX_WGS : cap));
  }
  if (grid > resident_cap) grid = resident_cap;
  k_exp3_apply_ranks<<<grid, E3_TPB, 0, (hipStream_t)stream>>>(m, gathered, barrier, err);
  return (int)hipGetLastError();
}

int bliss_exp3_normalize(void* w_pos, int64_t num_edges, int64_t* row_sum, int64_t* scratch, void* norm_out_bf16, void* stream) {
  if (!w_pos || !row_sum || !scratch || num_edges <= 0) return BLISS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  int64_t grid = (num_edges + E3_TPB * 8 - 1) / (E3_TPB * 8);
  if (grid > norm_wgs()) grid = norm_wgs();          // usually every workgroup returns at once (norm == 1.0): keep the launch small
  if (grid < 1) grid = 1;
  PROF_LAUNCH(BK_NORMALIZE, st, k_normalize_row<<<(int)grid, E3_TPB, 0, st>>>((bf16_t*)w_pos, num_edges, row_sum, scratch, (bf16_t*)norm_out_bf16, nullptr));
  return (int)hipGetLastError();
}

int bliss_exp3_normalize_global(void* w_pos, int64_t num_edges, int64_t* row_sum, const int64_t* norm_limbs, int64_t* scratch,
                                void* norm_out_bf16, void* stream) {
  if (!w_pos || !row_sum || !norm_limbs || !scratch || num_edges <= 0) return BLISS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  int64_t grid = (num_edges + E3_TPB * 8 - 1) / (E3_TPB * 8);
  if (grid > norm_wgs()) grid = norm_wgs();
  if (grid < 1) grid = 1;
  PROF_LAUNCH(BK_NORMALIZE, st, k_normalize_row<<<(int)grid, E3_TPB, 0, st>>>((bf16_t*)w_pos, num_edges, row_sum, scratch, (bf16_t*)norm_out_bf16, norm_limbs));
  return (int)hipGetLastError();
}

int bliss_exp3_normalize_global_rows(const bliss_exp3_block_t* rows, int32_t n_rows, int64_t num_edges, const int64_t* norm_limbs,
                                     int64_t limb_stride, void* stream) {
  if (!rows || n_rows <= 0 || n_rows > BLISS_EXP3_MAX_BLOCKS || num_edges <= 0 || !norm_limbs || limb_stride < 3 * ROWSUM_SLOTS) return BLISS_EINVAL;
  Exp3Multi m;
  m.n = n_rows;
  m.done_flag = nullptr;
  for (int i = 0; i < n_rows; ++i) {
    if (!rows[i].w_pos || !rows[i].row_sum || !rows[i].scratch) return BLISS_EINVAL;
    m.blk[i] = rows[i];
    m.grid_begin[i] = 0;
  }
  int64_t per_row = (num_edges + E3_TPB * 8 - 1) / (E3_TPB * 8);
  if (per_row > norm_wgs()) per_row = norm_wgs();
  if (per_row < 1) per_row = 1;
  hipStream_t st = (hipStream_t)stream;
  PROF_LAUNCH(BK_NORMALIZE, st, k_normalize_rows_global<<<(int)(per_row * n_rows), E3_TPB, 0, st>>>(m, num_edges, (int)per_row, norm_limbs, limb_stride));
  return (int)hipGetLastError();
}

int bliss_row_sum(const void* w_pos, int64_t num_edges, int64_t* row_sum, void* stream) {
  if (!w_pos || !row_sum || num_edges <= 0) return BLISS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  k_zero_i64<<<1, 128, 0, st>>>(row_sum, 3 * ROWSUM_SLOTS);          // not hipMemsetAsync: see k_init_counts in sampler.hip
  int64_t grid = (num_edges + E3_TPB * 8 - 1) / (E3_TPB * 8);
  if (grid > 4096) grid = 4096;
  PROF_LAUNCH(BK_ROW_SUM, st, k_row_sum<<<(int)grid, E3_TPB, 0, st>>>((const bf16_t*)w_pos, num_edges, row_sum, nullptr));
  return (int)hipGetLastError();
}

int bliss_normalized_edata(const bliss_graph_t* g, void* w_pos, void* stream) {
  if (!g || !w_pos) return BLISS_EINVAL;
  int grid = (g->num_nodes + 3) / 4;
  if (grid > 4096) grid = 4096;
  if (grid < 1) grid = 1;
  k_normalized_edata<<<grid, E3_TPB, 0, (hipStream_t)stream>>>(g->indptr, g->num_nodes, (bf16_t*)w_pos);
  return (int)hipGetLastError();
}

}  // extern "C"
