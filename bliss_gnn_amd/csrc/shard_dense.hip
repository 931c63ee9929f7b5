// The STATIC-SHAPE sharded sampler's own kernels (bliss_gnn_amd/shard_static.py).
//
// bliss_gnn_amd/shard.py routes (source, partial sum) pairs to the sources' owners, all-reduces a histogram and all-gathers the
// kept lists: three exchanges per layer whose sizes are data-dependent, hence a host sync each.  Here every exchange is DENSE:
// a rank scatters the exact Q.44 partial sums of compute_prob's by-source reduction (bandit_sampler.py:67-75) over its own
// seeds' columns into an int64 [2, |V|] buffer (sum, touch / seed mark), ONE all-reduce gives every rank every source's sum
// (integer addition: the bits do not depend on the reduction order or on the number of shards), and every rank then derives the
// SAME candidate list (ascending node id: the sharded mode's order, shard.py), importances, histogram, Poisson scale, keyed draw
// and kept list with no further collective.  Every size stays on the device; the whole layer records into a HIP graph.
//
//   bliss_shard_local_seeds       the global seed list -> the seeds this rank owns (order kept) and their positions
//   bliss_shard_scatter_partials  (seed, partial) + (touched source, partial) of bliss_frontier_prob(BLISS_MODE_PARTIALS) -> dense
//   bliss_shard_candidates        dense -> candidates in node order, p_j = sqrt(bf16(sum)) (:75), histogram of p's bit patterns
//   bliss_shard_select_kept       P_j, the keyed Poisson draw (:403-406, :422-424), kept list = seeds ++ drawn non-seeds, dense map
#include "common.cuh"
#include "bliss_gnn.h"

namespace {

#define SD_TPB 1024
#define SD_SEED_MARK (1ll << 32)

// ---- seeds of this rank ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(SD_TPB) k_sd_local_seeds(const int* __restrict__ seeds_g, int n_seeds, const int* __restrict__ n_seeds_dev,
                                                           int lo, int hi, int cap_s, int* __restrict__ seeds_l, int* __restrict__ seeds_l2,
                                                           int* __restrict__ seed_pos, int* __restrict__ n_local_dev, int* err) {
  __shared__ int sh[17];
  int S = n_seeds >= 0 ? n_seeds : *n_seeds_dev;
  if (S > cap_s) { if (threadIdx.x == 0 && err) atomicOr(err, BLISS_ERR_CAP_SEEDS); S = cap_s; }
  int run = 0;
  for (int base = 0; base < S; base += SD_TPB) {
    const int i = base + threadIdx.x;
    const int v = i < S ? seeds_g[i] : -1;
    const int mine = (v >= lo && v < hi) ? 1 : 0;
    int tot, ex = block_excl_scan(mine, sh, &tot);
    if (mine) { seeds_l[run + ex] = v; if (seeds_l2) seeds_l2[run + ex] = v; seed_pos[run + ex] = i; }
    run += tot;
    __syncthreads();
  }
  // (positions beyond the count stay valid indices, the ids behind them valid ids -- the list's first entry: consumers read both
  // at capacity, e.g. the label gather of the padded destination rows)
  const int pad = S > 0 ? seeds_g[0] : lo;
  for (int i = run + threadIdx.x; i < cap_s; i += SD_TPB) { seed_pos[i] = 0; seeds_l[i] = pad; }
  if (threadIdx.x == 0) *n_local_dev = run;
}

// ---- partial sums -> dense ------------------------------------------------------------------------------------------------
// (hipMemsetAsync issued from library code into a stream that torch is capturing did not replay with the graph here -- the
// buffer kept the marks of earlier steps from the second replay on -- so the zeroing is a kernel like everything else.  Round 3,
// later: the buffer is zeroed once (bliss_shard_zero_dense) and k_sd_cand returns the entries it read to zero: one launch less per layer)
__global__ void __launch_bounds__(256) k_sd_zero(long long* __restrict__ p, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) p[i] = 0;
}
__global__ void __launch_bounds__(256) k_sd_scatter(const int* __restrict__ seeds_l, const long long* __restrict__ seed_p2,
                                                    const int* __restrict__ n_local_dev, const unsigned long long* __restrict__ tkey,
                                                    const long long* __restrict__ tsum, const int* __restrict__ n_touched_dev,
                                                    long long* __restrict__ dense, int V, int* err) {
  const int n_local = *n_local_dev, n_t = *n_touched_dev;
  const int stride = gridDim.x * 256;
  int bad = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n_local + n_t; i += stride) {
    int v; long long s, mark;
    if (i < n_local) { v = seeds_l[i]; s = seed_p2[i]; mark = SD_SEED_MARK + 1; }     // a seed is a candidate whatever its sum (shard.py)
    else { v = (int)(tkey[i - n_local] & 0xffffffffull); s = tsum[i - n_local]; mark = 1; }
    if (v < 0 || v >= V) { bad = BLISS_ERR_CAP_CAND; continue; }
    // (one writer per node on a rank: local seeds and touched sources are disjoint.  Sum and mark side by side: ONE 16-byte store,
    // one line per node -- as two arrays [2, V] the scatter moved 13 MB per launch for 1.6 MB of payload, profiles/r03_z_shards_pmc.json)
    *reinterpret_cast<longlong2*>(dense + 2ll * v) = make_longlong2(s, mark);
  }
  if (bad && err) atomicOr(err, bad);
}

// ---- ordered compaction in ONE launch: decoupled look-back ---------------------------------------------------------------------
// Block b of 1024 elements needs the number of selected elements in the blocks before it.  One 64-bit status word per block:
// tag << 62 | value, tag 1 = "my own count", tag 2 = "the inclusive prefix up to me".  A block publishes its count at once, then
// walks back over its predecessors' words -- 64 at a time, one per lane of its first wave --, adding counts until it meets an
// inclusive prefix (blocks are dispatched in index order,
// so what it waits for is running or done; the spin is bounded like bliss_flag_wait's).  Two status arrays: the candidate pass
// uses A and returns B to zero, the kept pass uses B and returns A to zero -- each is zero again before its next use.
// (Round 3 started with count / scan / write launches: 18 launches per step on the critical stream.)
#define SD_TAG_SHIFT 62
__device__ __forceinline__ int sd_lookback(unsigned long long* status, int count, int* sh_prefix, int* err) {
  if (threadIdx.x < 64) {                               // the block's first wave: lane l looks at block hi - l, 64 predecessors per round
    const int b = blockIdx.x, lane = threadIdx.x;
    int ex = 0;
    if (b == 0) {
      if (lane == 0) __hip_atomic_store(status, (2ull << SD_TAG_SHIFT) | (unsigned long long)(unsigned)count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (lane == 0) __hip_atomic_store(status + b, (1ull << SD_TAG_SHIFT) | (unsigned long long)(unsigned)count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int hi = b - 1;
      long long spins = 0;
      for (;;) {
        const int j = hi - lane;
        const unsigned long long w = j >= 0 ? __hip_atomic_load(status + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                            : (2ull << SD_TAG_SHIFT);            // (in front of block 0: an inclusive prefix of 0)
        const unsigned tag = (unsigned)(w >> SD_TAG_SHIFT);
        const unsigned long long ready = __ballot(tag != 0), incl = __ballot(tag == 2);
        const int k = incl ? __ffsll((long long)incl) - 1 : 63;                  // the nearest inclusive prefix ends the walk
        const unsigned long long need = k == 63 ? ~0ull : ((1ull << (k + 1)) - 1);
        if ((ready & need) != need) {                   // someone in between has not published yet
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1ll << 22)) { if (lane == 0 && err) atomicOr(err, BLISS_ERR_FLAG_TIMEOUT); break; }
          continue;
        }
        int v = lane <= k ? (int)(w & 0xffffffffull) : 0;
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        ex += v;
        if (incl) break;
        hi -= 64;
      }
      if (lane == 0) __hip_atomic_store(status + b, (2ull << SD_TAG_SHIFT) | (unsigned long long)(unsigned)(ex + count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) *sh_prefix = ex;
  }
  __syncthreads();
  return *sh_prefix;
}

// (the histogram is privatised in LDS like k_cand_number's: importances live in a narrow band -- sqrt of a sum of squared
// fractions -- so a 4096-bin window + one counter for p == 0 catches them; plain global atomics on the few hot bins cost 70 us)
#define SD_HWIN_LO 0x3000
#define SD_HWIN_N 4096
__global__ void __launch_bounds__(SD_TPB) k_sd_cand(long long* __restrict__ dense, int V, int uniform_nodes,
                                                    unsigned long long* __restrict__ status, unsigned long long* __restrict__ status_other, int n_other,
                                                    int* __restrict__ cand_nid, bf16_t* __restrict__ p, unsigned char* __restrict__ is_seed,
                                                    int* __restrict__ hist, LayerCounts* cnt, int cap_c, int* err) {
  __shared__ int sh[17];
  __shared__ int sh_prefix;
  __shared__ int lh[SD_HWIN_N + 1];
  for (int i = threadIdx.x; i <= SD_HWIN_N; i += SD_TPB) lh[i] = 0;
  for (int w = blockIdx.x * SD_TPB + threadIdx.x; w < n_other; w += gridDim.x * SD_TPB) status_other[w] = 0ull;   // (the kept pass's words)
  const int v = blockIdx.x * SD_TPB + threadIdx.x;
  const longlong2 sm = v < V ? *reinterpret_cast<const longlong2*>(dense + 2ll * v) : make_longlong2(0, 0);
  const long long mark = sm.y;
  int tot, ex = block_excl_scan(mark != 0 ? 1 : 0, sh, &tot);          // (its barriers also cover the zeroing of lh)
  const int base = sd_lookback(status, tot, &sh_prefix, err);
  int bad = 0;
  if (mark != 0) {
    const int at = base + ex;
    const long long raw = sm.x;
    *reinterpret_cast<longlong2*>(dense + 2ll * v) = make_longlong2(0, 0);     // back to zero for the next scatter (only marked nodes are non-zero)
    if (at < cap_c) {
      bf16_t pj;
      if (uniform_nodes) pj = raw ? (bf16_t)0x3f80 : (bf16_t)0;                       // bandit_sampler.py:79-81
      else pj = f2bf(sqrtf(bf2f(fixed_to_bf(raw, FRAC_SRC, &bad))));                  // :75 torch.sqrt(prob)
      cand_nid[at] = v; p[at] = pj; is_seed[at] = mark >= SD_SEED_MARK ? 1 : 0;
      const int bin = pj & 0x7fff;                      // the histogram bliss_poisson_scale reads (and zeroes)
      if (bin == 0) atomicAdd(&lh[SD_HWIN_N], 1);
      else if (bin >= SD_HWIN_LO && bin < SD_HWIN_LO + SD_HWIN_N) atomicAdd(&lh[bin - SD_HWIN_LO], 1);
      else atomicAdd(&hist[bin], 1);
    } else bad |= BLISS_ERR_CAP_CAND;
  }
  __syncthreads();
  for (int i = threadIdx.x; i <= SD_HWIN_N; i += SD_TPB) {
    const int c = lh[i];
    if (c) atomicAdd(&hist[i == SD_HWIN_N ? 0 : SD_HWIN_LO + i], c);
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {               // the last block knows the total
    int C = base + tot;
    if (C > cap_c) C = cap_c;
    cnt->C = C; cnt->err = 0; cnt->iters = 0; cnt->all_one = 0;
  }
  if (bad && err) atomicOr(err, bad);
}

// SplitMix64 finaliser of (seed, step, layer, node id); top 24 bits -> u = r * 2^-24   (= csrc/shard.hip, oracle keyed_uniform)
__device__ __forceinline__ unsigned long long sd_key(unsigned long long seed, unsigned long long step, int layer) {
  unsigned long long key = seed * 0x9E3779B97F4A7C15ull + step;
  key = (key ^ (key >> 30)) * 0xBF58476D1CE4E5B9ull;
  key = (key ^ (key >> 27)) * 0x94D049BB133111EBull;
  key ^= key >> 31;
  return key ^ ((unsigned long long)((unsigned)layer & 0xffu) << 56);
}
__device__ __forceinline__ float sd_u24(unsigned long long key, int nid) {
  unsigned long long z = key ^ (unsigned long long)(unsigned)nid;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (float)(unsigned)(z >> 40) * (1.0f / 16777216.0f);
}
// P_j and the draw of candidate i (csrc/shard.hip:k_keyed_select, bandit_sampler.py:403-406, :422-424)
__device__ __forceinline__ bool sd_draw(int nid, bf16_t pj, bool seed, const LayerCounts* cnt, unsigned long long key, bf16_t* P_out) {
  bf16_t Pv = 0x3f80;
  if (!cnt->all_one && !seed) {
    const float v = rbf(bf2f(pj) * (float)cnt->c);
    Pv = (v < 1.0f || v != v) ? f2bf(v) : (bf16_t)0x3f80;
  }
  *P_out = Pv;
  return sd_u24(key, nid) < bf2f(Pv);
}

__global__ void __launch_bounds__(SD_TPB) k_sd_keep(const int* __restrict__ cand_nid, const bf16_t* __restrict__ p,
                                                    const unsigned char* __restrict__ is_seed, const LayerCounts* __restrict__ cnt,
                                                    unsigned long long seed, long long* __restrict__ step_dev, int layer,
                                                    unsigned long long* __restrict__ status, unsigned long long* __restrict__ status_other, int n_other,
                                                    const int* __restrict__ seeds_g, int n_seeds, const int* __restrict__ n_seeds_dev,
                                                    bf16_t* __restrict__ P_out, int* __restrict__ kept_nid, bf16_t* __restrict__ node_prob,
                                                    int* __restrict__ kept_map, int cap_k, LayerCounts* layer_cnt,
                                                    const int* __restrict__ n_local_dev, unsigned* ticket, int bump_step, int* done_flag,
                                                    int* err) {
  __shared__ int sh[17];
  __shared__ int sh_prefix;
  for (int w = blockIdx.x * SD_TPB + threadIdx.x; w < n_other; w += gridDim.x * SD_TPB) status_other[w] = 0ull;   // (the candidate pass's words)
  const unsigned long long key = sd_key(seed, (unsigned long long)*step_dev, layer);
  const int S = n_seeds >= 0 ? n_seeds : *n_seeds_dev;
  const int C = cnt->C;
  const int i = blockIdx.x * SD_TPB + threadIdx.x;
  int keep_new = 0;
  bf16_t Pv = 0;
  if (i < C) {
    const bool s = is_seed[i] != 0;
    keep_new = (sd_draw(cand_nid[i], p[i], s, cnt, key, &Pv) && !s) ? 1 : 0;
    P_out[i] = Pv;
  }
  int tot, ex = block_excl_scan(keep_new, sh, &tot);
  const int base = sd_lookback(status, tot, &sh_prefix, err);
  int bad = 0;
  if (keep_new) {
    const int at = S + base + ex;                       // the seeds come first, in seed order (bandit_sampler.py:408-414 union)
    if (at < cap_k) { kept_nid[at] = cand_nid[i]; node_prob[at] = Pv; kept_map[cand_nid[i]] = at; }
    else bad |= BLISS_ERR_CAP_KEPT;
  }
  // the seeds themselves: P = 1 (:403-404)
  for (int j = i; j < S; j += gridDim.x * SD_TPB) {
    if (j < cap_k) { const int v = seeds_g[j]; kept_nid[j] = v; node_prob[j] = 0x3f80; kept_map[v] = j; }
    else bad |= BLISS_ERR_CAP_KEPT;
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
    int K = S + base + tot;
    if (K > cap_k) K = cap_k;
    layer_cnt->K = K;
    layer_cnt->C = *n_local_dev;                        // what bliss_build_block's clean-up walks: cand_nid[0 .. C) = this rank's seeds
  }
  if (bad && err) atomicOr(err, bad);
  if (bump_step || done_flag) {
    // the LAST workgroup to finish (a ticket, left zero): every workgroup has read the step number and written its part of the lists
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (bump_step) *step_dev += 1;
        if (done_flag) __hip_atomic_store(done_flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

// ---- block inputs, owner side: out[i] = table[nid[i] - lo] for the rows i < K this rank owns, +0 elsewhere (the halo buffer that
// is then summed over the ranks as integer words: the zeros must be +0 bits).  One workgroup per row, 32-bit words.
__global__ void __launch_bounds__(128) k_sd_pack_rows(const int* __restrict__ nid, const int* __restrict__ n_rows_dev, int lo, int hi,
                                                      const unsigned* __restrict__ table, long long table_words, int row_words,
                                                      unsigned* __restrict__ out, long long out_words) {
  const int i = blockIdx.x;
  const int v = nid[i];
  const bool mine = i < *n_rows_dev && v >= lo && v < hi;
  const unsigned* src = table + (long long)(mine ? v - lo : 0) * table_words;
  unsigned* dst = out + (long long)i * out_words;
  for (int w = threadIdx.x; w < row_words; w += 128) dst[w] = mine ? src[w] : 0u;
}

// ---- rows between a rank's own list and a block's list (the halo buffers of the layers behind the first, and the destination rows
// of every layer).  pos[j], j < *n_dev: the position of this rank's j-th row in the block's list, ASCENDING (the own rows keep the
// list's order: k_sd_local_seeds).  One wave per row, 32-bit words (two bf16).
//   place: out[r] = src[j] where pos[j] == r, +0 bits for every other row (binary search: no zero-fill launch, no atomics)
//   take:  out[j] = bf16(src[pos[j]]) for j < *n_dev, +0 bits for the rows behind (src bf16 or fp32, RNE as torch's .to(bfloat16))
__global__ void __launch_bounds__(256) k_sd_place_rows(const unsigned* __restrict__ src, long long src_words, const int* __restrict__ pos,
                                                       const int* __restrict__ n_dev, int cap_s, unsigned* __restrict__ out, long long out_words,
                                                       int n_rows, int row_words) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n_rows) return;
  const int lane = threadIdx.x & 63;
  int n = *n_dev;
  if (n > cap_s) n = cap_s;
  int lo = 0, hi = n;                                   // the first j with pos[j] >= r
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (pos[mid] < r) lo = mid + 1; else hi = mid;
  }
  const bool hit = lo < n && pos[lo] == r;
  const unsigned* s = src + (long long)(hit ? lo : 0) * src_words;
  unsigned* d = out + (long long)r * out_words;
  for (int w = lane; w < row_words; w += 64) d[w] = hit ? s[w] : 0u;
}

template <bool F32>
__global__ void __launch_bounds__(256) k_sd_take_rows(const void* __restrict__ src, long long src_stride, int n_src_rows, const int* __restrict__ pos,
                                                      const int* __restrict__ n_dev, int cap_s, unsigned* __restrict__ out, long long out_words,
                                                      int row_words) {
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= cap_s) return;
  const int lane = threadIdx.x & 63;
  int n = n_dev ? *n_dev : cap_s;
  if (n > cap_s) n = cap_s;
  int p = j < n ? pos[j] : -1;
  if (p >= n_src_rows) p = -1;
  unsigned* d = out + (long long)j * out_words;
  if (p < 0) {
    for (int w = lane; w < row_words; w += 64) d[w] = 0u;
    return;
  }
  if (F32) {
    const float2* s = reinterpret_cast<const float2*>(static_cast<const float*>(src) + (long long)p * src_stride);
    for (int w = lane; w < row_words; w += 64) {
      const float2 v = s[w];
      d[w] = (unsigned)f2bf(v.x) | ((unsigned)f2bf(v.y) << 16);
    }
  } else {
    const unsigned* s = reinterpret_cast<const unsigned*>(static_cast<const bf16_t*>(src) + (long long)p * src_stride);
    for (int w = lane; w < row_words; w += 64) d[w] = s[w];
  }
}

inline int sd_blocks(int n) { return (n + SD_TPB - 1) / SD_TPB; }

}  // namespace

extern "C" {

int bliss_shard_local_seeds(const int32_t* seeds_g, int32_t n_seeds, const int32_t* n_seeds_dev, int32_t lo, int32_t hi, int32_t cap_s,
                            int32_t* seeds_l, int32_t* seeds_l_copy, int32_t* seed_pos, int32_t* n_local_dev, int32_t* err, void* stream) {
  if (!seeds_g || !seeds_l || !seed_pos || !n_local_dev || cap_s <= 0 || (n_seeds < 0 && !n_seeds_dev) || n_seeds > cap_s) return BLISS_EINVAL;
  k_sd_local_seeds<<<1, SD_TPB, 0, (hipStream_t)stream>>>(seeds_g, n_seeds, n_seeds_dev, lo, hi, cap_s, seeds_l, seeds_l_copy, seed_pos, n_local_dev, err);
  return (int)hipGetLastError();
}

int bliss_shard_scatter_partials(const int32_t* seeds_l, const int64_t* seed_p2, const int32_t* n_local_dev, const int64_t* touched_key,
                                 const int64_t* touched_sum, const int32_t* n_touched_dev, int64_t* dense, int32_t num_nodes,
                                 int32_t* err, void* stream) {
  if (!seeds_l || !seed_p2 || !n_local_dev || !touched_key || !touched_sum || !n_touched_dev || !dense || num_nodes <= 0) return BLISS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  int grid = (num_nodes + 255) / 256;
  if (grid > 1024) grid = 1024;
  k_sd_scatter<<<grid, 256, 0, st>>>(seeds_l, (const long long*)seed_p2, n_local_dev, (const unsigned long long*)touched_key,
                                     (const long long*)touched_sum, n_touched_dev, (long long*)dense, num_nodes, err);
  return (int)hipGetLastError();
}

int bliss_shard_zero_dense(int64_t* dense, int32_t num_nodes, void* stream) {
  if (!dense || num_nodes <= 0) return BLISS_EINVAL;
  int grid = (num_nodes + 255) / 256;
  if (grid > 1024) grid = 1024;
  k_sd_zero<<<grid, 256, 0, (hipStream_t)stream>>>((long long*)dense, 2ll * num_nodes);       // (a kernel, not hipMemsetAsync: see k_sd_zero)
  return (int)hipGetLastError();
}

int bliss_shard_pack_rows(const int32_t* nid, const int32_t* n_rows_dev, int32_t cap_rows, int32_t lo, int32_t hi, const void* table_bf16,
                          int64_t table_stride, int32_t row_len, void* out_bf16, int64_t out_stride, void* stream) {
  if (!nid || !n_rows_dev || cap_rows <= 0 || hi <= lo || !table_bf16 || !out_bf16 || row_len <= 0 || (row_len & 1) || (table_stride & 1) ||
      (out_stride & 1) || ((uintptr_t)table_bf16 & 3) || ((uintptr_t)out_bf16 & 3)) return BLISS_EINVAL;
  k_sd_pack_rows<<<cap_rows, 128, 0, (hipStream_t)stream>>>(nid, n_rows_dev, lo, hi, (const unsigned*)table_bf16, table_stride / 2, row_len / 2,
                                                            (unsigned*)out_bf16, out_stride / 2);
  return (int)hipGetLastError();
}

int bliss_shard_place_rows(const void* src_bf16, int64_t src_stride, const int32_t* pos, const int32_t* n_dev, int32_t cap_s, void* out_bf16,
                           int64_t out_stride, int32_t n_rows, int32_t row_len, void* stream) {
  if (!src_bf16 || !pos || !n_dev || cap_s <= 0 || !out_bf16 || n_rows <= 0 || row_len <= 0 || (row_len & 1) || (src_stride & 1) || (out_stride & 1) ||
      ((uintptr_t)src_bf16 & 3) || ((uintptr_t)out_bf16 & 3)) return BLISS_EINVAL;
  k_sd_place_rows<<<(n_rows + 3) / 4, 256, 0, (hipStream_t)stream>>>((const unsigned*)src_bf16, src_stride / 2, pos, n_dev, cap_s, (unsigned*)out_bf16,
                                                                     out_stride / 2, n_rows, row_len / 2);
  return (int)hipGetLastError();
}

int bliss_shard_take_rows(const void* src, int32_t src_is_f32, int64_t src_stride, int32_t n_src_rows, const int32_t* pos, const int32_t* n_dev,
                          int32_t cap_s, void* out_bf16, int64_t out_stride, int32_t row_len, void* stream) {
  if (!src || !pos || cap_s <= 0 || n_src_rows <= 0 || !out_bf16 || row_len <= 0 || (row_len & 1) || (src_stride & 1) || (out_stride & 1) ||
      ((uintptr_t)src & (src_is_f32 ? 7 : 3)) || ((uintptr_t)out_bf16 & 3)) return BLISS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int nb = (cap_s + 3) / 4;
  if (src_is_f32) k_sd_take_rows<true><<<nb, 256, 0, st>>>(src, src_stride, n_src_rows, pos, n_dev, cap_s, (unsigned*)out_bf16, out_stride / 2, row_len / 2);
  else k_sd_take_rows<false><<<nb, 256, 0, st>>>(src, src_stride, n_src_rows, pos, n_dev, cap_s, (unsigned*)out_bf16, out_stride / 2, row_len / 2);
  return (int)hipGetLastError();
}

int bliss_shard_candidates(int64_t* dense, int32_t num_nodes, int32_t uniform_nodes, int32_t* cand_nid, void* p_bf16, uint8_t* is_seed,
                           int32_t* hist, void* counts, int32_t cap_c, int32_t* scratch, int32_t* err, void* stream) {
  if (!dense || num_nodes <= 0 || !cand_nid || !p_bf16 || !is_seed || !hist || !counts || cap_c <= 0 || !scratch || ((uintptr_t)scratch & 7))
    return BLISS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int nb = sd_blocks(num_nodes), nb_k = sd_blocks(cap_c);   // scratch: uint64[nb + nb_k] -- this pass's status words, then the kept pass's
  unsigned long long* st_a = reinterpret_cast<unsigned long long*>(scratch);
  k_sd_cand<<<nb, SD_TPB, 0, st>>>((long long*)dense, num_nodes, uniform_nodes, st_a, st_a + nb, nb_k, cand_nid, (bf16_t*)p_bf16, is_seed, hist,
                                   (LayerCounts*)counts, cap_c, err);
  return (int)hipGetLastError();
}

int bliss_shard_select_kept(const int32_t* cand_nid, const void* p_bf16, const uint8_t* is_seed, const void* counts, uint64_t seed,
                            int64_t* step_dev, int32_t layer, const int32_t* seeds_g, int32_t n_seeds, const int32_t* n_seeds_dev,
                            void* P_bf16, int32_t* kept_nid, void* node_prob_bf16, int32_t* kept_map, int32_t cap_k, int32_t cap_c,
                            int32_t num_nodes, void* layer_counts, const int32_t* n_local_dev, int32_t* scratch, int32_t bump_step,
                            int32_t* done_flag, int32_t* err, void* stream) {
  if (!cand_nid || !p_bf16 || !is_seed || !counts || !step_dev || !seeds_g || (n_seeds < 0 && !n_seeds_dev) || !P_bf16 || !kept_nid ||
      !node_prob_bf16 || !kept_map || cap_k <= 0 || cap_c <= 0 || num_nodes <= 0 || !layer_counts || !n_local_dev || !scratch ||
      ((uintptr_t)scratch & 7)) return BLISS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int nb = sd_blocks(cap_c), nb_c = sd_blocks(num_nodes);   // scratch: the SAME array as bliss_shard_candidates'; blocks beyond the count find nothing
  unsigned long long* st_a = reinterpret_cast<unsigned long long*>(scratch);
  k_sd_keep<<<nb, SD_TPB, 0, st>>>(cand_nid, (const bf16_t*)p_bf16, is_seed, (const LayerCounts*)counts, seed, (long long*)step_dev, layer,
                                   st_a + nb_c, st_a, nb_c, seeds_g, n_seeds, n_seeds_dev, (bf16_t*)P_bf16, kept_nid, (bf16_t*)node_prob_bf16, kept_map,
                                   cap_k, (LayerCounts*)layer_counts, n_local_dev, reinterpret_cast<unsigned*>(st_a + nb_c + nb), bump_step, done_flag, err);
  return (int)hipGetLastError();
}

}  // extern "C"
