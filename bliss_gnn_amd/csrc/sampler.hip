// Layer-wise bandit / LADIES block sampler for gfx950 -- frontier expansion, EXP3 edge
// probabilities, LADIES node importance, Poisson scale + draw, block construction.
//
// Replaces, per layer, the ~40 DGL/ATen launches and >= 8 host syncs of
//   bandit_sampler.py:101-138 (exp3_probabilities), :47-82 + :381-406 (compute_prob),
//   :408-425 (select_neighbors), :269-339 (generate_block)      and the ladies_sampler.py twins
// with 15 sync-free launches that never materialise the frontier: the passes re-read the seeds' CSC columns
// (coalesced, 4 B index + 2 B weight per edge) and keep only per-seed, per-candidate and per-kept-edge state.
// Sizes (E, C, K, B) live on the device in LayerCounts.
//
//   k_seg_scan -> [k_col_sums] -> k_bin_scatter -> k_bin_reduce -> k_bitmap_tiles -> k_cand_number     bliss_frontier_prob
//   k_poisson_scale -> k_select_fused                                                                  bliss_poisson_select
//   k_block_pass1 -> k_block_scans -> k_block_pass2 -> k_tr_sort_lists (+ map cleanup)                 bliss_build_block
// (k_frontier_pass1/2/3 + k_cand_finalize: the same candidate stage with memory-side atomics, for graphs whose node
// slots do not fit the LDS bins.)
//
// Node maps are dense |V|-sized arrays (local_id, kept_map; first_pos / acc_p2 for the atomic path) -- on a 288 GB
// part a dense map beats a hash table: one L2-resident gather per edge, no probing, reset by touched entry.
//
// Ordering contract: candidates = seeds in given order, then every other frontier
// source by FIRST APPEARANCE in the dst-major frontier; kept nodes and block edges keep that
// relative order, so the block comes out CSR-by-destination with no sort anywhere.
#include "common.cuh"
#include "bliss_gnn.h"
#include "prof.h"
#include <cstdlib>

#ifndef TPB
#define TPB 256
#endif
#define ITEMS 4
#define CHUNK (TPB * ITEMS)

namespace {

// Work decomposition of every frontier pass: a workgroup (4 waves) owns a CHUNK of 1024 consecutive
// frontier positions, a wave owns a SPAN of 256 consecutive positions inside it and walks it as 4
// items of 64 (one position per lane).  Consecutive items let the segment lookup carry its hint, and
// ordered ranks come from ballots + one 4-entry LDS exchange per chunk instead of a block scan per item.
#define SPAN (64 * ITEMS)
#define SEG_ZERO_WGS 8
#define LONG_COL 1024                     // = COL_BIG below: columns longer than this get a workgroup in k_col_sums

struct EdgeAt {
  int k;          // seed index (local destination id); valid iff e < E
  int64_t pos;    // CSC position
  int src;        // global source id
};

// lane's edge at frontier position e = base + lane (base wave-uniform)
__device__ __forceinline__ EdgeAt decode(int base, int E, int S, const int* __restrict__ seg_ptr,
                                         const long long* __restrict__ col_base, const int* __restrict__ indices, int* k_hint) {
  EdgeAt r;
  const int e = base + lane_id();
  r.k = wave_segment(seg_ptr, S, base, k_hint);
  r.pos = 0; r.src = 0;
  if (e < E) {
    r.pos = col_base[r.k] + e;                       // = indptr[seeds[k]] + (e - seg_ptr[k]), folded by k_seg_scan
    r.src = indices[r.pos];
  } else r.k = -1;
  return r;
}

// exclusive rank of this wave's span inside its chunk + chunk total, from per-wave totals (one barrier)
__device__ __forceinline__ int chunk_wave_offset(int wave_total, int* sh4, int* chunk_total) {
  const int wave = threadIdx.x >> 6;
  __syncthreads();                                   // sh4 free again
  if (lane_id() == 0) sh4[wave] = wave_total;
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < TPB / 64; ++w) { const int v = sh4[w]; tot += v; if (w < wave) off += v; }
  *chunk_total = tot;
  return off;
}

// q_ij = eta/n_i + (1-eta) * w_ij / sum_j w_ij        bandit_sampler.py:131-137
__device__ __forceinline__ bf16_t edge_q(bf16_t w, bf16_t wsum, int n, float eta_f, float ome_f) {
  float wd = rbf(bf2f(w) / bf2f(wsum));     // :131 e_div_v
  // :137 (self.eta / n_i).bfloat16(): Python `scalar / tensor` is Tensor.__rtruediv__ = reciprocal() * scalar,
  // i.e. TWO fp32 roundings on the int32 -> fp32 degree, then one to bf16
  float a = rbf((1.0f / (float)n) * eta_f);
  float b = rbf(ome_f * wd);                // :137 (1 - self.eta) * exp_weights_divided
  return f2bf(a + b);                       // :137 v_add_e
}

// the same with the per-seed part a = rbf((1/n) * eta) taken from k_col_sums' per-seed record
__device__ __forceinline__ bf16_t edge_q_pre(bf16_t w, bf16_t wsum, float a, float ome_f) {
  float wd = rbf(bf2f(w) / bf2f(wsum));
  float b = rbf(ome_f * wd);
  return f2bf(a + b);
}

// ---------------------------------------------------------------- K_a: seed columns -> seg_ptr
// also resets the counts record and zeroes the per-seed accumulators (a kernel, not hipMemsetAsync: memset nodes
// of a captured HIP graph were observed to leave this buffer stale on replay -- ROCm 7.2)
// (wg / n_wgs: this workgroup's index among those of the launch; keep_sums: k_col_sums, launched right behind, WRITES the
// first two per-seed accumulators -- leave them alone)
__device__ __forceinline__ void seg_scan_body(const int64_t* __restrict__ indptr, const int* __restrict__ seeds,
                                              LayerCounts* cnt, int S_host, const int* __restrict__ S_dev, int cap_s,
                                              unsigned long long* __restrict__ seed_acc, int* __restrict__ seg_ptr,
                                              int* __restrict__ local_id, int num_nodes, int* __restrict__ src_cnt, int cap_k,
                                              int* __restrict__ bin_cursor, int n_bins, long long* __restrict__ col_base,
                                              int* __restrict__ span_seg, long long frontier_cap, int* entry_flag,
                                              int wg, int n_wgs, bool keep_sums, int* sh, int* st_sh, int* hub_count = nullptr) {
  // This kernel runs <=> everything enqueued before this layer has completed (stream / graph order): tell a consumer on
  // another stream (bliss_flag_wait) without an event, i.e. without cutting a captured graph in two.
  if (entry_flag && wg == 0 && threadIdx.x == 0) __hip_atomic_store(entry_flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  // hub_count (optional): int[cap_s + 1] -- the columns of more than COL_BIG edges, in no particular order, count last: k_col_sums
  // gives each of them a workgroup without walking over the short ones
  __shared__ int sh_long;
  if (hub_count && threadIdx.x == 0) sh_long = 0;     // (appended to behind the first block_excl_scan's barriers)
  int S = S_host >= 0 ? S_host : *S_dev;
  int bad = 0;
  if (S > cap_s) { S = cap_s; bad |= BLISS_ERR_CAP_SEEDS; }         // clamp: results invalid but in bounds
  if (wg > 0 || n_wgs == 1) {                                         // the zeroing does not wait for the serial scan
    const int nz = n_wgs > 1 ? n_wgs - 1 : 1, bz = n_wgs > 1 ? wg - 1 : 0;
    const int t0 = bz * blockDim.x + threadIdx.x, step = nz * blockDim.x;
    // acc_w, acc_q, acc_wt (u64) + deg_blk (i32) + seed_p2 (u64)
    for (int i = (keep_sums ? 2 * cap_s : 0) + t0; i < cap_s * 5; i += step) seed_acc[i] = 0ull;
    if (src_cnt) for (int i = t0; i <= cap_k; i += step) src_cnt[i] = 0;
    if (bin_cursor) for (int i = t0; i <= n_bins; i += step) bin_cursor[i] = 0;   // [n_bins] = touched count
    // the seeds' local ids (scattered 4-byte stores): here, spread over several CUs, not in the one workgroup that scans
    for (int k = t0; k < S; k += step) { const int s = seeds[k]; if (s >= 0 && s < num_nodes) local_id[s] = k; }
    if (n_wgs > 1) return;
  }
  long long run = 0;
  constexpr int G = 4;                                  // four 1024-seed rounds have their pointer chases in flight together
  for (int base0 = 0; base0 < S; base0 += G * (int)blockDim.x) {
    int degs[G];
    long long cols[G];
#pragma unroll
    for (int i = 0; i < G; ++i) {
      const int k = base0 + i * (int)blockDim.x + threadIdx.x;
      degs[i] = 0; cols[i] = 0;
      if (k < S) {
        const int s = seeds[k];
        if (s < 0 || s >= num_nodes) { bad |= BLISS_ERR_CAP_CAND; }
        else {
          cols[i] = indptr[s];
          degs[i] = (int)(indptr[s + 1] - cols[i]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < G; ++i) {
      const int k = base0 + i * (int)blockDim.x + threadIdx.x;
      if (base0 + i * (int)blockDim.x >= S) break;      // block-uniform
      const int deg = degs[i];
      int tot, ex = block_excl_scan(deg, sh, &tot);
      const long long start = run + ex;
      if (k < S) {
        seg_ptr[k] = (int)start;
        // what every frontier pass needs to find its edges without a search and without chasing seeds -> indptr:
        col_base[k] = deg > 0 ? cols[i] - start : 0;    // CSC position = col_base[k] + frontier position
      }
      if (hub_count) {
        const bool is_long = k < S && deg > LONG_COL;
        const unsigned long long mask = __ballot(is_long);
        if (mask) {
          int base_l = 0;
          if (lane_id() == 0) base_l = atomicAdd(&sh_long, __popcll(mask));
          base_l = __shfl(base_l, 0);
          if (is_long) hub_count[base_l + __popcll(mask & ((1ull << lane_id()) - 1ull))] = k;
        }
      }
      // span_seg[sp] = the seed column holding frontier position sp * 256.  Done by the whole workgroup over the round's
      // spans (a search in the round's 1024 column starts), not by every thread over its own column: a hub column of 30 K
      // edges has > 100 spans, and a round used to wait for the thread that owned it.
      // (a frontier longer than the graph has edges means repeated seeds: flagged, nothing is written past the tables)
      if (run + tot <= frontier_cap) {
        st_sh[threadIdx.x] = (int)(k < S ? start : run + tot);
        __syncthreads();
        const long long sp_end = (run + tot + SPAN - 1) / SPAN;
        for (long long sp = (run + SPAN - 1) / SPAN + threadIdx.x; sp < sp_end; sp += blockDim.x) {
          const int p = (int)(sp * SPAN);
          int t = 0;                                    // last column of the round that starts at or before p (st_sh[0] = run <= p)
#pragma unroll
          for (int step = 512; step >= 1; step >>= 1) if (st_sh[t + step] <= p) t += step;
          span_seg[sp] = base0 + i * (int)blockDim.x + t;
        }
      }
      run += tot;
      if (run > frontier_cap) bad |= BLISS_ERR_CAP_FRONTIER;
    }
  }
  int any_bad = __syncthreads_or(bad);
  if (threadIdx.x == 0) {
    if (hub_count) hub_count[cap_s] = sh_long;
    seg_ptr[S] = (int)run;
    cnt->S = S; cnt->E = (any_bad & BLISS_ERR_CAP_FRONTIER) ? 0 : (int)run;
    cnt->C = 0; cnt->K = 0; cnt->B = 0; cnt->iters = 0; cnt->all_one = 0; cnt->c = 1.0;
    cnt->err = any_bad;
  }
}

__global__ void __launch_bounds__(1024) k_seg_scan(const int64_t* __restrict__ indptr, const int* __restrict__ seeds,
                                                   LayerCounts* cnt, int S_host, const int* __restrict__ S_dev, int cap_s,
                                                   unsigned long long* __restrict__ seed_acc, int* __restrict__ seg_ptr,
                                                   int* __restrict__ local_id, int num_nodes, int* __restrict__ src_cnt, int cap_k,
                                                   int* __restrict__ bin_cursor, int n_bins, long long* __restrict__ col_base,
                                                   int* __restrict__ span_seg, long long frontier_cap, int* entry_flag,
                                                   int keep_sums, int* hub_count) {
  __shared__ int sh[17];
  __shared__ int st_sh[1024];
  seg_scan_body(indptr, seeds, cnt, S_host, S_dev, cap_s, seed_acc, seg_ptr, local_id, num_nodes, src_cnt, cap_k, bin_cursor, n_bins,
                col_base, span_seg, frontier_cap, entry_flag, blockIdx.x, gridDim.x, keep_sums != 0, sh, st_sh, hub_count);
}

// ---------------------------------------------------------------- K_b: first appearance + sum_j w_ij
template <bool BANDIT>
__global__ void __launch_bounds__(TPB) k_frontier_pass1(const int64_t* __restrict__ indptr, const int* __restrict__ indices,
                                                        const bf16_t* __restrict__ w, const int* __restrict__ seeds,
                                                        const int* __restrict__ seg_ptr, const long long* __restrict__ col_base,
    const int* __restrict__ span_seg, LayerCounts* cnt,
                                                        const int* __restrict__ local_id, unsigned* first_pos,
                                                        unsigned long long* acc_w, const int* __restrict__ w_pend) {
  int pend;                                                 // a deferred F.normalize pass over this row (bliss_exp3_step_deferred)
  const bf16_t* __restrict__ wq = norm_state_row(w, w_pend, &pend);
  const float pdenom = renorm_denom(pend ? pend : 0x3f80);
  const int S = cnt->S, E = cnt->E;
  const int nspans = (E + SPAN - 1) / SPAN;
  int bad = 0;
  for (int sp = blockIdx.x * (TPB / 64) + (threadIdx.x >> 6); sp < nspans; sp += gridDim.x * (TPB / 64)) {
    int hint = span_seg[sp];                                 // the segment this span starts in (k_seg_scan)
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      const int base = sp * SPAN + i * 64;
      if (base >= E) break;                                 // wave-uniform
      const int e = base + lane_id();
      EdgeAt a = decode(base, E, S, seg_ptr, col_base, indices, &hint);
      int64_t term = 0;
      if (a.k >= 0) {
        if (local_id[a.src] < 0) {                          // not a seed: seeds are numbered already
          // the atomics execute at the memory side and leave nothing in L2, so a plain pre-check would keep
          // reading a stale 0xFFFFFFFF line; an agent-scope (sc1) load sees the current minimum
          if (__hip_atomic_load(first_pos + a.src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > (unsigned)e)
            atomicMin(first_pos + a.src, (unsigned)e);
        }
        if (BANDIT) term = bf_to_fixed(renorm_pending(wq[a.pos], pend, pdenom), FRAC_DST, &bad);   // :129 copy_e_sum over exp3 weights
      }
      if (BANDIT) wave_segsum_atomic_i64(a.k, term, acc_w);
    }
  }
  if (bad) atomicOr(&cnt->err, bad);
}

// ---------------------------------------------------------------- K_d: sum_k q_ik + count first appearances
template <bool BANDIT>
__global__ void __launch_bounds__(TPB) k_frontier_pass2(const int64_t* __restrict__ indptr, const int* __restrict__ indices,
                                                        const bf16_t* __restrict__ w, const int* __restrict__ seeds,
                                                        const int* __restrict__ seg_ptr, const long long* __restrict__ col_base,
    const int* __restrict__ span_seg, LayerCounts* cnt,
                                                        const unsigned* __restrict__ first_pos,
                                                        const unsigned long long* __restrict__ acc_w,
                                                        unsigned long long* acc_q, int* __restrict__ chunk_cnt,
                                                        float eta_f, float ome_f, const int* __restrict__ w_pend) {
  int pend;                                                 // a deferred F.normalize pass over this row (bliss_exp3_step_deferred)
  const bf16_t* __restrict__ wq = norm_state_row(w, w_pend, &pend);
  const float pdenom = renorm_denom(pend ? pend : 0x3f80);
  __shared__ int sh4[TPB / 64];
  const int S = cnt->S, E = cnt->E;
  const int nchunks = (E + CHUNK - 1) / CHUNK;
  int bad = 0;
  for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    int nfirst = 0, hint = span_seg[chunk * ITEMS + (threadIdx.x >> 6)];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      const int base = chunk * CHUNK + (threadIdx.x >> 6) * SPAN + i * 64;
      if (base >= E) break;
      const int e = base + lane_id();
      EdgeAt a = decode(base, E, S, seg_ptr, col_base, indices, &hint);
      int64_t term = 0;
      bool first = false;
      if (a.k >= 0) {
        first = first_pos[a.src] == (unsigned)e;
        if (BANDIT) {
          bf16_t wsum = fixed_to_bf((int64_t)acc_w[a.k], FRAC_DST, &bad);
          bf16_t q = edge_q(renorm_pending(wq[a.pos], pend, pdenom), wsum, seg_ptr[a.k + 1] - seg_ptr[a.k], eta_f, ome_f);
          term = bf_to_fixed(q, FRAC_DST, &bad);       // :67 copy_e_sum(insg, edge_prob)
        }
      }
      nfirst += __popcll(__ballot(first));
      if (BANDIT) wave_segsum_atomic_i64(a.k, term, acc_q);
    }
    int tot;
    chunk_wave_offset(nfirst, sh4, &tot);
    if (threadIdx.x == 0) chunk_cnt[chunk] = tot;
  }
  if (bad) atomicOr(&cnt->err, bad);
}

// ---------------------------------------------------------------- single-block exclusive scan of chunk counts
// which: 0 -> C = S + total (candidates), 1 -> K = total (kept nodes), 2 -> B = total (block edges)
__global__ void __launch_bounds__(1024) k_chunk_scan(int* __restrict__ chunk_cnt, LayerCounts* cnt, int which, int cap) {
  __shared__ int sh[17];
  int n_items = which == 1 ? cnt->C : cnt->E;
  const int n = (n_items + CHUNK - 1) / CHUNK;
  int run = 0;
  for (int base = 0; base < n; base += blockDim.x) {
    int i = base + threadIdx.x;
    int v = i < n ? chunk_cnt[i] : 0;
    int tot, ex = block_excl_scan(v, sh, &tot);
    if (i < n) chunk_cnt[i] = run + ex;
    run += tot;
  }
  if (threadIdx.x == 0) {
    int total = which == 0 ? cnt->S + run : run;
    int errbit = which == 0 ? BLISS_ERR_CAP_CAND : (which == 1 ? BLISS_ERR_CAP_KEPT : BLISS_ERR_CAP_EDGES);
    if (total > cap) { atomicOr(&cnt->err, errbit); total = cap; }   // clamp: results invalid but in bounds
    if (which == 0) cnt->C = total; else if (which == 1) cnt->K = total; else cnt->B = total;
  }
}

// ---------------------------------------------------------------- K_f: number candidates + scatter (q/sum q)^2 by source
template <bool BANDIT>
__global__ void __launch_bounds__(TPB) k_frontier_pass3(const int64_t* __restrict__ indptr, const int* __restrict__ indices,
                                                        const bf16_t* __restrict__ w, const int* __restrict__ seeds,
                                                        const int* __restrict__ seg_ptr, const long long* __restrict__ col_base,
    const int* __restrict__ span_seg, LayerCounts* cnt,
                                                        const unsigned* __restrict__ first_pos,
                                                        const unsigned long long* __restrict__ acc_w,
                                                        const unsigned long long* __restrict__ acc_q,
                                                        const int* __restrict__ chunk_off, int* local_id,
                                                        int* __restrict__ cand_nid, unsigned long long* acc_p2,
                                                        float eta_f, float ome_f, int cap_c, int uniform_nodes, const int* __restrict__ w_pend) {
  int pend;                                                 // a deferred F.normalize pass over this row (bliss_exp3_step_deferred)
  const bf16_t* __restrict__ wq = norm_state_row(w, w_pend, &pend);
  const float pdenom = renorm_denom(pend ? pend : 0x3f80);
  __shared__ int sh4[TPB / 64];
  const int S = cnt->S, E = cnt->E;
  const int nchunks = (E + CHUNK - 1) / CHUNK;
  int bad = 0;
  for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    int hint = span_seg[chunk * ITEMS + (threadIdx.x >> 6)], wave_total = 0;
    unsigned long long mask[ITEMS];
    int srcs[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      mask[i] = 0; srcs[i] = 0;
      const int base = chunk * CHUNK + (threadIdx.x >> 6) * SPAN + i * 64;
      if (base >= E) continue;
      const int e = base + lane_id();
      EdgeAt a = decode(base, E, S, seg_ptr, col_base, indices, &hint);
      bool first = false;
      if (a.k >= 0) {
        srcs[i] = a.src;
        first = first_pos[a.src] == (unsigned)e;
        bf16_t t = 0;
        if (uniform_nodes) {                            // importance_sampling=False (:77-81): only "has an out-edge" matters
          acc_p2[a.src] = 1ull;
        } else if (BANDIT) {
          bf16_t wsum = fixed_to_bf((int64_t)acc_w[a.k], FRAC_DST, &bad);
          bf16_t q = edge_q(renorm_pending(wq[a.pos], pend, pdenom), wsum, seg_ptr[a.k + 1] - seg_ptr[a.k], eta_f, ome_f);
          bf16_t qsum = fixed_to_bf((int64_t)acc_q[a.k], FRAC_DST, &bad);
          float r = rbf(bf2f(q) / bf2f(qsum));          // :71 e_div_u on the reversed frontier
          t = f2bf(r * r);                              // :73 edge_prob_div_sum ** 2
        } else {
          float x = bf2f(renorm_pending(wq[a.pos], pend, pdenom));                     // ladies_sampler.py:46-47  weight ** 2
          t = f2bf(x * x);
        }
        int64_t fx = uniform_nodes ? 0 : bf_to_fixed(t, FRAC_SRC, &bad);
        if (fx) atomicAdd(acc_p2 + a.src, (unsigned long long)fx);   // :73 copy_e_sum by SOURCE
      }
      mask[i] = __ballot(first);
      wave_total += __popcll(mask[i]);
    }
    int tot;
    int run = S + chunk_off[chunk] + chunk_wave_offset(wave_total, sh4, &tot);
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      if ((mask[i] >> lane_id()) & 1ull) {
        const int id = run + __popcll(mask[i] & ((1ull << lane_id()) - 1ull));
        if (id < cap_c) { local_id[srcs[i]] = id; cand_nid[id] = srcs[i]; }
      }
      run += __popcll(mask[i]);
    }
  }
  if (bad) atomicOr(&cnt->err, bad);
}

// ---------------------------------------------------------------- K_g: p_j = sqrt(sum), reset the dense maps,
// and histogram the bf16 bit patterns of p (32768 non-negative values) for the Poisson scale: the
// reference evaluates sum_j min(c p_j, 1) up to 50 times over all C candidates with a host sync each
// (bandit_sampler.py:396-401); with counts per distinct p the same sum costs 32 bins per thread.
#define HIST_BINS 32768
#define HWIN_LO 0x3000           // k_cand_number's LDS window of the histogram: bf16 patterns of [2^-31, 2)
#define HWIN_N 4096
#ifndef FIN_TPB
#define FIN_TPB 1024             // (512: step 0.790 ms, 1024: 0.772, 256: 0.828 -- same box, Reddit-like loop)
#endif
__global__ void __launch_bounds__(FIN_TPB) k_cand_finalize(const int* __restrict__ seeds, LayerCounts* cnt, int* __restrict__ cand_nid,
                                                           unsigned long long* acc_p2, unsigned* first_pos,
                                                           bf16_t* __restrict__ p, int* hist, int cap_c, int uniform_nodes) {
  __shared__ int lh[HIST_BINS];                       // 128 KiB static LDS (one workgroup per CU; gfx950 has 160 KiB)
  const int S = cnt->S;
  const int C = min(cnt->C, cap_c);
  if ((int)blockIdx.x * FIN_TPB >= C) return;         // surplus workgroups: nothing to zero, nothing to flush
  {                                                   // (16 bytes per LDS access: the 128 KiB are most of this kernel's work)
    int4* lh4 = reinterpret_cast<int4*>(lh);
    for (int b = threadIdx.x; b < HIST_BINS / 4; b += FIN_TPB) lh4[b] = make_int4(0, 0, 0, 0);
  }
  __syncthreads();
  int bad = 0;
  for (int id = blockIdx.x * FIN_TPB + threadIdx.x; id < C; id += gridDim.x * FIN_TPB) {
    int g;
    if (id < S) { g = seeds[id]; cand_nid[id] = g; } else g = cand_nid[id];
    const unsigned long long raw = acc_p2[g];
    acc_p2[g] = 0;
    first_pos[g] = 0xffffffffu;
    bf16_t pj;
    if (uniform_nodes) pj = raw ? (bf16_t)0x3f80 : (bf16_t)0;            // :79-81 ones, 0 where out_degree == 0
    else pj = f2bf(sqrtf(bf2f(fixed_to_bf((int64_t)raw, FRAC_SRC, &bad))));   // :75 torch.sqrt(prob)
    p[id] = pj;
    if (pj < HIST_BINS) atomicAdd(&lh[pj], 1); else bad |= BLISS_ERR_NONFINITE;   // sign bit set = negative / -0 cannot occur
  }
  __syncthreads();
  {
    const int4* lh4 = reinterpret_cast<const int4*>(lh);
    for (int b = threadIdx.x; b < HIST_BINS / 4; b += FIN_TPB) {
      const int4 v = lh4[b];
      if (v.x | v.y | v.z | v.w) {
        if (v.x) atomicAdd(hist + 4 * b, v.x);
        if (v.y) atomicAdd(hist + 4 * b + 1, v.y);
        if (v.z) atomicAdd(hist + 4 * b + 2, v.z);
        if (v.w) atomicAdd(hist + 4 * b + 3, v.w);
      }
    }
  }
  if (bad) atomicOr(&cnt->err, bad);
}

// ================================================================ binned candidate pipeline
// The passes above spend their time in scattered memory-side atomics: one atomicMin (first appearance) and one 64-bit
// atomicAdd (sum of squares) per frontier edge, ~14 edges per candidate on a Reddit-like layer, at the ~20 G atomics/s
// the memory side sustains.  When |V| / n_bins node slots fit in LDS the same two reductions are done there instead:
//   k_col_sums     one workgroup per seed column: sum_j w_ij, then sum_k q_ik with the column held in registers
//   k_bin_scatter  every edge becomes one 64-bit record (position, source, (q/sum q)^2), multisplit in LDS by source % n_bins and
//                  appended to its bin with ONE global atomic per (workgroup, bin)
//   k_bin_reduce   one workgroup per bin: min position and exact fixed-point sum per source with LDS atomics; sets one
//                  bit per first appearance in a bitmap over frontier positions and lists the touched sources
//   k_bitmap_tiles prefix popcount of the bitmap => rank of every first appearance (the candidate order)
//   k_cand_number  numbers the candidates, p_j = sqrt(sum), histogram for the Poisson scale
// Integer sums and minima are order-free, so the results are bit-identical to the atomic passes.
#define BIN_TPB 1024
#ifndef BIN_ITEMS
#define BIN_ITEMS 4
#endif
#define BIN_BATCH (BIN_TPB * BIN_ITEMS)      // frontier positions per workgroup step: 4096 (16 waves keep the loads in flight)
#define MAX_BINS 1024
#ifndef COL_TPB
#define COL_TPB 1024
#endif
#ifndef COL_R
#define COL_R 16
#endif
#ifndef COL_RB
#define COL_RB 8
#endif
#define COL_BIG (COL_R * 64)              // a wave keeps a whole column of up to 1024 edges in registers
static_assert(LONG_COL == COL_BIG, "k_seg_scan lists the columns k_col_sums treats as long");
#ifndef BINRED_TPB
#define BINRED_TPB 1024          // one workgroup per bin, 256 bins: 16 waves per CU (512: 23.1 us per launch, 1024: 21.0, 256: 28.1)
#endif

template <int NT>
__device__ __forceinline__ int block_max_u31(int v, long long* sh) {
  v = wave_max_u31(v);
  __syncthreads();                                    // sh free again
  if (lane_id() == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  int t = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) t = max(t, (int)sh[w]);
  return t;
}

// two sums through one pair of barriers (sh2: 2 * NT / 64 entries)
template <int NT>
__device__ __forceinline__ long long block_sum2_i64(long long a, long long b, long long* sh2, long long* b_out) {
  a = wave_total_i64(a);
  b = wave_total_i64(b);
  __syncthreads();
  if (lane_id() == 0) { sh2[threadIdx.x >> 6] = a; sh2[NT / 64 + (threadIdx.x >> 6)] = b; }
  __syncthreads();
  long long ta = 0, tb = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) { ta += sh2[w]; tb += sh2[NT / 64 + w]; }
  *b_out = tb;
  return ta;
}

template <int NT>
__device__ __forceinline__ long long block_sum_i64(long long v, long long* sh) {
  v = wave_total_i64(v);
  __syncthreads();                                    // sh free again
  if (lane_id() == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  long long t = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) t += sh[w];
  return t;
}

// one WAVE per seed column (typical columns hold a few hundred to a few thousand edges: several columns in flight per
// workgroup hide the pointer-chasing latency, and the two sums are DPP wave reductions with no barrier); columns longer
// than COL_BIG are left to a second loop in which the whole workgroup shares one column
__device__ __forceinline__ void col_store(int k, long long ws_fixed, long long qs_fixed, bf16_t wsum, int n, float eta_f,
                                          unsigned long long* acc_w, unsigned long long* acc_q, uint2* seed_coef, int* bad) {
  acc_w[k] = (unsigned long long)ws_fixed; acc_q[k] = (unsigned long long)qs_fixed;
  // what the per-edge passes need of this seed, converted once: bf16 sums and the eta / n_i term of q_ij (:137)
  const bf16_t qsum = fixed_to_bf(qs_fixed, FRAC_DST, bad);
  seed_coef[k] = make_uint2((unsigned)wsum | ((unsigned)qsum << 16), __float_as_uint(rbf((1.0f / (float)n) * eta_f)));
}

// (Round 2 tried three other shapes for this kernel -- scan and column sums in one 1024-thread launch; four columns per
// 256-thread workgroup with the long ones taken by the same workgroup; the latter plus one fat workgroup per hub column --
// and measured all of them SLOWER on the Reddit-like step (34 / 78 / 73 us per layer against 16 / 27 / 55 for this one,
// profiles/r02_e and r02_f step timelines): every dependent global access costs microseconds here, and this shape has the
// fewest per workgroup -- one column per wave, one long column per workgroup, all of them side by side.  What does help the
// long columns is MORE threads each: 1024-thread workgroups holding 8 edges per thread in registers took the launch from
// 47 to 34 us on average over the three layers (512 x 32: 47, 256 x 32: 56, 1024 x 16: 35, 1024 x 32: 43), the step from
// 0.816 to 0.789 ms, same box.)
__global__ void __launch_bounds__(COL_TPB) k_col_sums(const int64_t* __restrict__ indptr, const bf16_t* __restrict__ w,
                                                      const int* __restrict__ seeds, const int* __restrict__ seg_ptr,
                                                      const long long* __restrict__ col_base, const int* __restrict__ span_seg,
                                                      LayerCounts* cnt, unsigned long long* __restrict__ acc_w,
                                                      unsigned long long* __restrict__ acc_q, float eta_f, float ome_f,
                                                      uint2* __restrict__ seed_coef, int n_wave_wgs, const int* __restrict__ w_pend,
                                                      const int* __restrict__ long_list, int cap_s) {
  int pend;                                                 // a deferred F.normalize pass over this row (bliss_exp3_step_deferred)
  const bf16_t* __restrict__ wq = norm_state_row(w, w_pend, &pend);
  const float pdenom = renorm_denom(pend ? pend : 0x3f80);
  __shared__ long long sh[COL_TPB / 64];
  __shared__ long long sh2[2 * COL_TPB / 64];
  const int S = cnt->S, tid = threadIdx.x, lane = lane_id();
  if (cnt->E == 0) return;
  int bad = 0;
  // the first n_wave_wgs workgroups take the short columns (one per wave), the others the long ones (one per workgroup):
  // both kinds are latency chains, so they run side by side instead of one after the other
  const int n_block_wgs = (int)gridDim.x - n_wave_wgs;
  // ---- columns up to COL_BIG edges: one per wave
  if ((int)blockIdx.x < n_wave_wgs)
  for (int k = blockIdx.x * (COL_TPB / 64) + (tid >> 6); k < S; k += n_wave_wgs * (COL_TPB / 64)) {
    const int s0 = seg_ptr[k], n = seg_ptr[k + 1] - s0;
    if (n == 0 || n > COL_BIG) continue;              // wave-uniform
    const long long p0 = col_base[k] + s0;
    bf16_t wr[COL_R];
    long long part = 0;
    int emax = 1;
#pragma unroll
    for (int r = 0; r < COL_R; ++r) {
      const int i = lane + r * 64;
      wr[r] = 0;
      if (i < n) wr[r] = wq[p0 + i];
    }
    if (pend) {                                       // (uniform; after ALL the loads have been issued)
#pragma unroll
      for (int r = 0; r < COL_R; ++r) wr[r] = renorm_bf16(wr[r], pdenom);
    }
#pragma unroll
    for (int r = 0; r < COL_R; ++r)
      if (lane + r * 64 < n) emax = max(emax, bf_exp_field(wr[r]));
    const int wfrac = rel_frac(FRAC_DST, wave_max_u31(emax));      // block-floating: exact relative to the column's largest weight
    long long part_lo = 0;
    int sticky = 0;
#pragma unroll
    for (int r = 0; r < COL_R; ++r)
      if (lane + r * 64 < n) part += bf_to_fixed_wide(wr[r], wfrac, &part_lo, &sticky, &bad);  // :129 copy_e_sum over exp3 weights
    const long long ws_fixed = wave_total_i64(part);
    bf16_t wsum;
    if (__ballot(part_lo != 0 || sticky) == 0ull) wsum = fixed_to_bf(ws_fixed, wfrac, &bad);   // nothing lost below the last bit
    else wsum = fixed_wide_to_bf(ws_fixed, wave_total_i64(part_lo), __ballot(sticky) != 0ull, wfrac, &bad);
    const float a = rbf((1.0f / (float)n) * eta_f);
    part = 0;
#pragma unroll
    for (int r = 0; r < COL_R; ++r) {
      const int i = lane + r * 64;
      if (i < n) part += bf_to_fixed(edge_q_pre(wr[r], wsum, a, ome_f), FRAC_DST, &bad);     // :67 copy_e_sum(insg, edge_prob)
    }
    const long long qs_fixed = wave_total_i64(part);
    if (lane == 0) col_store(k, ws_fixed, qs_fixed, wsum, n, eta_f, acc_w, acc_q, seed_coef, &bad);
  }
  // ---- the long columns: one per workgroup, the first COL_RB * COL_TPB edges held in registers between the two sums
  if ((int)blockIdx.x >= n_wave_wgs)
  for (int li = (int)blockIdx.x - n_wave_wgs, n_long = long_list[cap_s]; li < n_long; li += n_block_wgs) {
    const int k = long_list[li];                      // (k_seg_scan's list: no workgroup walks over short columns)
    const int s0 = seg_ptr[k], n = seg_ptr[k + 1] - s0;
    if (n <= COL_BIG) continue;                       // block-uniform (cannot happen: the list holds exactly the long ones)
    const long long p0 = col_base[k] + s0;
    bf16_t wr[COL_RB];
    long long part = 0;
    int emax = 1;
#pragma unroll
    for (int r = 0; r < COL_RB; ++r) {
      const int i = tid + r * COL_TPB;
      wr[r] = 0;
      if (i < n) wr[r] = wq[p0 + i];
    }
    if (pend) {
#pragma unroll
      for (int r = 0; r < COL_RB; ++r) wr[r] = renorm_bf16(wr[r], pdenom);
    }
#pragma unroll
    for (int r = 0; r < COL_RB; ++r)
      if (tid + r * COL_TPB < n) emax = max(emax, bf_exp_field(wr[r]));
#pragma unroll 8
    for (int i = tid + COL_RB * COL_TPB; i < n; i += COL_TPB) emax = max(emax, bf_exp_field(renorm_pending(wq[p0 + i], pend, pdenom)));
    const int wfrac = rel_frac(FRAC_DST, block_max_u31<COL_TPB>(emax, sh));
    long long part_lo = 0;
    int sticky = 0;
#pragma unroll
    for (int r = 0; r < COL_RB; ++r)
      if (tid + r * COL_TPB < n) part += bf_to_fixed_wide(wr[r], wfrac, &part_lo, &sticky, &bad);
#pragma unroll 8
    for (int i = tid + COL_RB * COL_TPB; i < n; i += COL_TPB)
      part += bf_to_fixed_wide(renorm_pending(wq[p0 + i], pend, pdenom), wfrac, &part_lo, &sticky, &bad);
    // (lost bits anywhere in the column? the count rides through the SAME pair of barriers as the sum)
    long long lost;
    const long long ws_fixed = block_sum2_i64<COL_TPB>(part, (part_lo != 0 || sticky) ? 1 : 0, sh2, &lost);
    bf16_t wsum;
    if (lost == 0) wsum = fixed_to_bf(ws_fixed, wfrac, &bad);
    else {
      const long long lo_tot = block_sum_i64<COL_TPB>(part_lo, sh);
      const long long st_tot = block_sum_i64<COL_TPB>(sticky, sh);
      wsum = fixed_wide_to_bf(ws_fixed, lo_tot, st_tot != 0, wfrac, &bad);
    }
    const float a = rbf((1.0f / (float)n) * eta_f);
    part = 0;
#pragma unroll
    for (int r = 0; r < COL_RB; ++r) {
      const int i = tid + r * COL_TPB;
      if (i < n) part += bf_to_fixed(edge_q_pre(wr[r], wsum, a, ome_f), FRAC_DST, &bad);
    }
#pragma unroll 8
    for (int i = tid + COL_RB * COL_TPB; i < n; i += COL_TPB) part += bf_to_fixed(edge_q_pre(renorm_pending(wq[p0 + i], pend, pdenom), wsum, a, ome_f), FRAC_DST, &bad);
    const long long qs_fixed = block_sum_i64<COL_TPB>(part, sh);
    if (tid == 0) col_store(k, ws_fixed, qs_fixed, wsum, n, eta_f, acc_w, acc_q, seed_coef, &bad);
  }
  if (bad) atomicOr(&cnt->err, bad);
}

template <bool BANDIT>
__global__ void __launch_bounds__(BIN_TPB) k_bin_scatter(const int64_t* __restrict__ indptr, const int* __restrict__ indices,
                                                     const bf16_t* __restrict__ w, const int* __restrict__ seeds,
                                                     const int* __restrict__ seg_ptr, const long long* __restrict__ col_base,
    const int* __restrict__ span_seg, LayerCounts* cnt,
                                                     const unsigned long long* __restrict__ acc_w,
                                                     const unsigned long long* __restrict__ acc_q, float eta_f, float ome_f,
                                                     int uniform_nodes, int n_bins, int log2_bins, long long bin_cap, int* bin_cursor,
                                                     unsigned long long* __restrict__ bin_rec, unsigned* __restrict__ bitmap,
                                                     const uint2* __restrict__ seed_coef, const int* __restrict__ w_pend) {
  int pend;                                                 // a deferred F.normalize pass over this row (bliss_exp3_step_deferred)
  const bf16_t* __restrict__ wq = norm_state_row(w, w_pend, &pend);
  const float pdenom = renorm_denom(pend ? pend : 0x3f80);
  __shared__ int hist[MAX_BINS];
  __shared__ int gbase[MAX_BINS];
  const int S = cnt->S, E = cnt->E, tid = threadIdx.x;
  const int nb = (E + BIN_BATCH - 1) / BIN_BATCH;
  const int bmask = n_bins - 1;
  int bad = 0;
  for (int batch = blockIdx.x; batch < nb; batch += gridDim.x) {
    for (int b = tid; b < n_bins; b += BIN_TPB) hist[b] = 0;
    if (tid < BIN_BATCH / 32) bitmap[batch * (BIN_BATCH / 32) + tid] = 0u;      // this batch's slice of the first-appearance bitmap
    __syncthreads();
    int srcs[BIN_ITEMS], ranks[BIN_ITEMS];
    bf16_t ts[BIN_ITEMS];
    const int wbase = batch * BIN_BATCH + (tid >> 6) * (64 * BIN_ITEMS);
    int hint = span_seg[wbase >> 8];       // a wave owns 256 consecutive positions
#pragma unroll
    for (int j = 0; j < BIN_ITEMS; ++j) {
      ranks[j] = -1; srcs[j] = 0; ts[j] = 0;
      const int base = wbase + j * 64;
      if (base >= E) continue;                                                  // wave-uniform
      EdgeAt a = decode(base, E, S, seg_ptr, col_base, indices, &hint);
      if (a.k >= 0) {
        bf16_t t;
        if (uniform_nodes) {
          t = (bf16_t)0x3f80;                           // importance_sampling=False (:77-81): only "has an out-edge" matters
        } else if (BANDIT) {
          const uint2 cf = seed_coef[a.k];              // per-seed: bf16 sum_j w_ij | bf16 sum_k q_ik, eta / n_i
          bf16_t q = edge_q_pre(renorm_pending(wq[a.pos], pend, pdenom), (bf16_t)(cf.x & 0xffffu), __uint_as_float(cf.y), ome_f);
          float r = rbf(bf2f(q) / bf2f((bf16_t)(cf.x >> 16)));   // :71 e_div_u on the reversed frontier
          t = f2bf(r * r);                              // :73 edge_prob_div_sum ** 2
        } else {
          float x = bf2f(renorm_pending(wq[a.pos], pend, pdenom));                     // ladies_sampler.py:46-47  weight ** 2
          t = f2bf(x * x);
        }
        srcs[j] = a.src; ts[j] = t;
        ranks[j] = atomicAdd(&hist[a.src & bmask], 1);
      }
    }
    __syncthreads();
    for (int b = tid; b < n_bins; b += BIN_TPB) {
      const int h = hist[b];
      int g0 = 0;
      if (h) {
        g0 = atomicAdd(bin_cursor + b, h);
        if ((long long)g0 + h > bin_cap) bad |= BLISS_ERR_CAP_CAND;
      }
      gbase[b] = g0;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < BIN_ITEMS; ++j) {
      if (ranks[j] >= 0) {
        const int b = srcs[j] & bmask;
        const long long idx = (long long)gbase[b] + ranks[j];
        if (idx < bin_cap) {
          const size_t o = (size_t)b * (size_t)bin_cap + (size_t)idx;
          const unsigned e = (unsigned)(wbase + j * 64 + lane_id());
          // position (32) | slot = source / n_bins (17) | term (15: squares are non-negative) -- one 8-byte store per edge
          bin_rec[o] = ((unsigned long long)e << 32) | ((unsigned long long)((unsigned)srcs[j] >> log2_bins) << 15) | (ts[j] & 0x7fffu);
        }
      }
    }
    __syncthreads();                                    // hist / gbase are reused by the next batch
  }
  if (bad) atomicOr(&cnt->err, bad);
}

__global__ void __launch_bounds__(BINRED_TPB) k_bin_reduce(LayerCounts* cnt, int n_bins, int log2_bins, long long bin_cap,
                                                           int* bin_cursor, const unsigned long long* __restrict__ bin_rec,
                                                           int num_nodes, int slots,
                                                           const int* __restrict__ local_id, unsigned long long* __restrict__ seed_p2,
                                                           unsigned long long* __restrict__ touched_key,
                                                           unsigned long long* __restrict__ touched_sum, unsigned* bitmap, int cap_c) {
  extern __shared__ unsigned long long dyn_lds[];
  unsigned long long* sm = dyn_lds;                    // [slots] exact sum of (q / sum q)^2 per source of this bin
  unsigned* mn = (unsigned*)(dyn_lds + slots);         // [slots] first frontier position per source
  const int b = blockIdx.x, tid = threadIdx.x;
  if (cnt->E == 0) return;
  const int ne = b < num_nodes ? ((num_nodes - 1 - b) >> log2_bins) + 1 : 0;   // nodes v with v % n_bins == b
  for (int i = tid; i < ne; i += BINRED_TPB) { sm[i] = 0ull; mn[i] = 0xffffffffu; }
  __syncthreads();
  long long n = bin_cursor[b];
  if (n > bin_cap) n = bin_cap;
  const unsigned long long* rec = bin_rec + (size_t)b * (size_t)bin_cap;
  int bad = 0;
  for (long long i = tid; i < n; i += BINRED_TPB) {
    const unsigned long long r = rec[i];
    const int li = (int)(((unsigned)r) >> 15);
    atomicMin(&mn[li], (unsigned)(r >> 32));
    const int64_t fx = bf_to_fixed((bf16_t)(r & 0x7fffu), FRAC_SRC, &bad);
    if (fx) atomicAdd(&sm[li], (unsigned long long)fx);                         // :73 copy_e_sum by SOURCE
  }
  __syncthreads();
  // seeds are numbered already: only their sums are needed; what stays marked in mn[] are the new candidates
  __shared__ int wg_cnt, wg_base;
  if (tid == 0) wg_cnt = 0;
  __syncthreads();
  for (int li = tid; li < ne; li += BINRED_TPB) {
    bool is_cand = false;
    if (mn[li] != 0xffffffffu) {
      const int lid = local_id[((unsigned)li << log2_bins) | (unsigned)b];
      if (lid >= 0) { seed_p2[lid] = sm[li]; mn[li] = 0xffffffffu; }
      else is_cand = true;
    }
    const unsigned long long mask = __ballot(is_cand);
    if (mask && lane_id() == 0) atomicAdd(&wg_cnt, __popcll(mask));
  }
  __syncthreads();
  if (tid == 0) { wg_base = wg_cnt ? atomicAdd(bin_cursor + n_bins, wg_cnt) : 0; wg_cnt = 0; }   // ONE global atomic per bin
  __syncthreads();
  for (int li = tid; li < ne; li += BINRED_TPB) {
    const unsigned fp = mn[li];
    if (fp != 0xffffffffu) {
      const int j = wg_base + atomicAdd(&wg_cnt, 1);
      if (j < cap_c) { touched_key[j] = ((unsigned long long)fp << 32) | (((unsigned)li << log2_bins) | (unsigned)b); touched_sum[j] = sm[li]; }
      atomicOr(bitmap + (fp >> 5), 1u << (fp & 31u));
    }
  }
  if (bad) atomicOr(&cnt->err, bad);
}

// rank of a first appearance = bits set before it: exclusive popcount prefix per bitmap word inside a tile of 4096 words
// (one workgroup per tile, no serial loop) + per-tile totals that k_cand_number prefixes itself
#define BTILE_W 4                                       // bitmap words per thread
#define BTILE (1024 * BTILE_W)
#define MAX_TILES 2048
__global__ void __launch_bounds__(1024) k_bitmap_tiles(const unsigned* __restrict__ bitmap, int* __restrict__ word_prefix,
                                                       int* __restrict__ tile_sum, LayerCounts* cnt) {
  __shared__ int sh[17];
  const int nw = (cnt->E + 31) >> 5;                   // the bitmap is allocated (and zeroed) in whole 128-word batches
  const int ntiles = (nw + BTILE - 1) / BTILE;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int i = tile * BTILE + threadIdx.x * BTILE_W;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (i < nw) v = *reinterpret_cast<const uint4*>(bitmap + i);
    const int c0 = __popc(v.x), c1 = __popc(v.y), c2 = __popc(v.z), c3 = __popc(v.w);
    int total, ex = block_excl_scan(c0 + c1 + c2 + c3, sh, &total);
    if (i < nw) *reinterpret_cast<int4*>(word_prefix + i) = make_int4(ex, ex + c0, ex + c0 + c1, ex + c0 + c1 + c2);
    if (threadIdx.x == 0) tile_sum[tile] = total;
  }
}

// ---------------------------------------------------------------- K_h: Poisson scale c  (bandit_sampler.py:391-401)
__device__ __forceinline__ double block_sum_f64(double v, double* shd) {
  for (int d = 32; d >= 1; d >>= 1) {
    long long b = __double_as_longlong(v);
    int lo = __shfl_down((int)(b & 0xffffffffll), d), hi = __shfl_down((int)(b >> 32), d);
    v += __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
  }
  __syncthreads();
  if (lane_id() == 0) shd[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += shd[i];   // same order in every thread
  return t;
}

// (1024 threads.  COHERENT: the caller is the last workgroup of the launch that filled `hist` with memory-side atomics -- read it
// past this XCD's L2)
template <bool COHERENT>
__device__ __forceinline__ void poisson_scale_body(int* hist, LayerCounts* cnt, int C, int num, double eps, int* rng_ctl,
                                                   int* layer_off, int is_last, int rng_cap_total, int* __restrict__ sel_state,
                                                   double* shd) {
  // ticket + one status word per 1024-candidate chunk for k_select_fused's look-back
  for (int i = threadIdx.x; i < (C + CHUNK - 1) / CHUNK + 2; i += 1024) sel_state[i] = 0;
  // the random numbers of this layer come from the streaming generator: one lane waits for them while the others work
  if (rng_ctl && threadIdx.x == 1023) rng_stream_acquire(rng_ctl, C, layer_off, is_last, rng_cap_total);
  // every thread owns 32 bins; load the counts and leave the histogram zero for the next layer
  int n[HIST_BINS / 1024];
  bool any = false;
#pragma unroll
  for (int i = 0; i < HIST_BINS / 1024; ++i) {
    const int b = i * 1024 + threadIdx.x;
    n[i] = COHERENT ? __hip_atomic_load(hist + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : hist[b];
  }
#pragma unroll
  for (int i = 0; i < HIST_BINS / 1024; ++i) {
    const int b = i * 1024 + threadIdx.x;
    if (n[i]) {
      if (COHERENT) __hip_atomic_store(hist + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else hist[b] = 0;
      any = true;
    }
  }
  if (C <= num) {                                     // :392-393 everything is kept
    if (threadIdx.x == 0) { cnt->c = 1.0; cnt->all_one = 1; cnt->iters = 0; }
    return;
  }
  double c = 1.0;
  int it = 0;
  for (; it < 50; ++it) {                             // :396
    const float c32 = (float)c;                       // torch multiplies a bf16 tensor by a Python float in fp32
    double loc = 0;
    if (any) {
#pragma unroll
      for (int i = 0; i < HIST_BINS / 1024; ++i) {
        if (n[i]) {
          float v = rbf(bf2f((bf16_t)(i * 1024 + threadIdx.x)) * c32);
          v = v < 1.0f ? v : (v != v ? v : 1.0f);     // torch.minimum propagates NaN
          loc += (double)n[i] * (double)v;            // count * bf16 value: exact in fp64
        }
      }
    }
    double Ssum = block_sum_f64(loc, shd);            // :397, exact in fp64 (bf16 terms, < 2^24 of them)
    double lo = Ssum < (double)num ? Ssum : (double)num, hi = Ssum < (double)num ? (double)num : Ssum;
    if (lo / hi >= eps) { ++it; break; }              // :398
    c *= (double)num / Ssum;                          // :401
  }
  if (threadIdx.x == 0) { cnt->c = c; cnt->all_one = 0; cnt->iters = it > 50 ? 50 : it; }
}

__global__ void __launch_bounds__(1024) k_poisson_scale(int* hist, LayerCounts* cnt, int num, double eps, int* rng_ctl,
                                                        int* layer_off, int is_last, int rng_cap_total, int* __restrict__ sel_state) {
  __shared__ double shd[16];
  poisson_scale_body<false>(hist, cnt, cnt->C, num, eps, rng_ctl, layer_off, is_last, rng_cap_total, sel_state, shd);
}

// the Poisson scale in k_cand_number's launch: the last workgroup to have flushed its histogram runs it (one launch less on
// the sampler's critical chain, ~4.5 us per layer)
struct FusedScale { int* ticket; int num; double eps; int* rng_ctl; int* layer_off; int is_last; int rng_cap_total; int* sel_state; };
__global__ void __launch_bounds__(FIN_TPB) k_cand_number(const int* __restrict__ seeds, LayerCounts* cnt, int* __restrict__ cand_nid,
                                                         int* local_id, const unsigned long long* __restrict__ seed_p2,
                                                         const unsigned long long* __restrict__ touched_key,
                                                         const unsigned long long* __restrict__ touched_sum,
                                                         const unsigned* __restrict__ bitmap, const int* __restrict__ word_prefix,
                                                         const int* __restrict__ tile_sum, bf16_t* __restrict__ p, int* hist,
                                                         int cap_c, int uniform_nodes, const FusedScale fs) {
  // Histogram of p's bit patterns for the Poisson scale, privatised in LDS.  Round 2 kept all 32768 bins per workgroup
  // (128 KiB: zeroing and flushing them was most of this kernel's 20-30 us); the importances of a frontier live in a narrow
  // band -- sqrt of a sum of squared fractions: (2^-31, 2) covers them -- so the LDS copy is the 4096-bin window
  // [HWIN_LO, HWIN_LO + HWIN_N) plus one counter for p == 0; anything else (never seen on the tested graphs) goes straight to
  // the global histogram.  Same counts in `hist`, whatever the route.
  __shared__ int lh[HWIN_N + 4];
  __shared__ int tile_off[MAX_TILES];
  __shared__ int sh[17];
  const int S = cnt->S;
  const int ntiles = (((cnt->E + 31) >> 5) + BTILE - 1) / BTILE;
  int run = 0;                                        // every workgroup prefixes the (few) tile totals for itself
  for (int base = 0; base < ntiles; base += FIN_TPB) {
    const int t = base + threadIdx.x;
    int total, ex = block_excl_scan(t < ntiles ? tile_sum[t] : 0, sh, &total);
    if (t < ntiles) tile_off[t] = run + ex;
    run += total;
  }
  int C = S + run;
  if (C > cap_c) { C = cap_c; if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&cnt->err, BLISS_ERR_CAP_CAND); }
  if (blockIdx.x == 0 && threadIdx.x == 0) cnt->C = C;
  if (blockIdx.x != 0 && (int)blockIdx.x * FIN_TPB >= C) return;      // surplus workgroups (workgroup 0 always takes part)
  {
    int4* lh4 = reinterpret_cast<int4*>(lh);
    for (int b = threadIdx.x; b < (HWIN_N + 4) / 4; b += FIN_TPB) lh4[b] = make_int4(0, 0, 0, 0);
  }
  __syncthreads();
  int bad = 0;
  for (int i = blockIdx.x * FIN_TPB + threadIdx.x; i < C; i += gridDim.x * FIN_TPB) {
    int id;
    unsigned long long raw;
    if (i < S) {
      id = i; raw = seed_p2[i];
      cand_nid[i] = seeds[i];
    } else {
      const unsigned long long key = touched_key[i - S];
      raw = touched_sum[i - S];
      const unsigned fp = (unsigned)(key >> 32), g = (unsigned)key;
      id = S + tile_off[(fp >> 5) / BTILE] + word_prefix[fp >> 5] + __popc(bitmap[fp >> 5] & ((1u << (fp & 31u)) - 1u));   // rank of the first appearance
      if (id >= cap_c) continue;
      local_id[g] = id; cand_nid[id] = (int)g;
    }
    bf16_t pj;
    if (uniform_nodes) pj = raw ? (bf16_t)0x3f80 : (bf16_t)0;            // :79-81 ones, 0 where out_degree == 0
    else pj = f2bf(sqrtf(bf2f(fixed_to_bf((int64_t)raw, FRAC_SRC, &bad))));   // :75 torch.sqrt(prob)
    p[id] = pj;
    if (pj >= HIST_BINS) bad |= BLISS_ERR_NONFINITE;                     // sign bit set = negative / -0 cannot occur
    else if ((unsigned)(pj - HWIN_LO) < (unsigned)HWIN_N) atomicAdd(&lh[pj - HWIN_LO], 1);
    else if (pj == 0) atomicAdd(&lh[HWIN_N], 1);
    else atomicAdd(hist + pj, 1);
  }
  __syncthreads();
  {
    const int4* lh4 = reinterpret_cast<const int4*>(lh);
    for (int b = threadIdx.x; b < HWIN_N / 4; b += FIN_TPB) {
      const int4 v = lh4[b];
      if (v.x | v.y | v.z | v.w) {
        if (v.x) atomicAdd(hist + HWIN_LO + 4 * b, v.x);
        if (v.y) atomicAdd(hist + HWIN_LO + 4 * b + 1, v.y);
        if (v.z) atomicAdd(hist + HWIN_LO + 4 * b + 2, v.z);
        if (v.w) atomicAdd(hist + HWIN_LO + 4 * b + 3, v.w);
      }
    }
    if (threadIdx.x == 0 && lh[HWIN_N]) atomicAdd(hist, lh[HWIN_N]);
  }
  if (bad) atomicOr(&cnt->err, bad);
  if (fs.ticket) {
    // the histogram is complete when every participating workgroup has passed here: atomics drained, then a ticket
    static_assert(FIN_TPB == 1024, "the fused Poisson scale runs in a 1024-thread workgroup");
    __shared__ int sh_is_last;
    __shared__ double shd[16];
    int active = (C + FIN_TPB - 1) / FIN_TPB;
    if (active > (int)gridDim.x) active = (int)gridDim.x;
    if (active < 1) active = 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      sh_is_last = atomicAdd(fs.ticket, 1) == active - 1;
      if (sh_is_last) __hip_atomic_store(fs.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (sh_is_last)
      poisson_scale_body<true>(hist, cnt, C, fs.num, fs.eps, fs.rng_ctl, fs.layer_off, fs.is_last, fs.rng_cap_total, fs.sel_state, shd);
  }
}

__device__ __forceinline__ bf16_t incl_prob(const bf16_t* __restrict__ p, int j, int S, int all_one, float c32) {
  if (all_one || j < S) return 0x3f80;                // 1.0: early-out (:393) or a seed (:403-404, inf*c -> min -> 1)
  float v = rbf(bf2f(p[j]) * c32);                    // :406
  return (v < 1.0f || v != v) ? f2bf(v) : (bf16_t)0x3f80;
}

// ---------------------------------------------------------------- K_i + scan + K_k in ONE launch
// Bernoulli compare and ORDERED compaction in a single pass: every workgroup takes the next 1024-candidate chunk from a
// ticket counter, publishes its kept count and looks back over its predecessors' status words (aggregate / inclusive
// prefix) for its base rank.  Tickets are handed out in execution order, so a predecessor is always running or done.
// Saves two launches (>= 4 us each inside a graph) per layer over pass1 + chunk scan + pass2.
#define SEL_AGG 1
#define SEL_PFX 2
__global__ void __launch_bounds__(TPB) k_select_fused(const bf16_t* __restrict__ p, const float* __restrict__ uniforms_base,
                                                      const int* __restrict__ u_off, LayerCounts* cnt, bf16_t* __restrict__ P,
                                                      int* state, const int* __restrict__ cand_nid, int* __restrict__ new_id,
                                                      int* __restrict__ kept_nid, bf16_t* __restrict__ node_prob, int cap_c, int cap_k,
                                                      int* __restrict__ kept_map) {
  __shared__ int sh4[TPB / 64];
  __shared__ int sh_chunk, sh_base, sh_K;
  const float* __restrict__ uniforms = uniforms_base + (u_off ? *u_off : 0);
  const int S = cnt->S, C = min(cnt->C, cap_c), all_one = cnt->all_one;
  const float c32 = (float)cnt->c;
  const int nchunks = (C + CHUNK - 1) / CHUNK;
  if (threadIdx.x == 0) sh_chunk = atomicAdd(state, 1);
  __syncthreads();
  const int chunk = sh_chunk;
  if (chunk >= nchunks) return;
  unsigned long long mask[ITEMS];
  bf16_t Pv[ITEMS];
  int wave_total = 0;
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int j = chunk * CHUNK + (threadIdx.x >> 6) * SPAN + i * 64 + lane_id();
    bool keep = false;
    Pv[i] = 0;
    if (j < C) {
      Pv[i] = incl_prob(p, j, S, all_one, c32);
      P[j] = Pv[i];
      // the numbers may come from a generator kernel that is still running: agent-scope (sc1) load
      keep = __hip_atomic_load(uniforms + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < bf2f(Pv[i]);   // :422-424  u24 < float(P)
    }
    mask[i] = __ballot(keep);
    wave_total += __popcll(mask[i]);
  }
  int tot;
  const int woff = chunk_wave_offset(wave_total, sh4, &tot);
  if (threadIdx.x < 64) {                              // wave 0 looks back 64 predecessors at a time
    const int lane = lane_id();
    int excl = 0, bad = 0;
    if (chunk > 0) {
      if (lane == 0) __hip_atomic_store(state + 1 + chunk, (tot << 2) | SEL_AGG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      long long spins = 0;
      for (int top = chunk - 1; top >= 0;) {
        const int j = top - lane;                       // lane 0 = nearest predecessor
        const int v = j >= 0 ? __hip_atomic_load(state + 1 + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : SEL_PFX;
        const unsigned long long ready = __ballot((v & 3) != 0), pfx = __ballot((v & 3) == SEL_PFX);
        // usable lanes: 0 .. first prefix (inclusive), all of them published
        const int stop = pfx ? __ffsll((long long)pfx) - 1 : 63;
        const unsigned long long need = stop == 63 ? ~0ull : ((1ull << (stop + 1)) - 1ull);
        if ((ready & need) != need) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1ll << 22)) { bad = BLISS_ERR_CAP_KEPT; break; }     // never hang the GPU; the step is flagged invalid
          continue;
        }
        excl += wave_total_i32(lane <= stop && j >= 0 ? (v >> 2) : 0);
        if (pfx) break;
        top -= 64;
      }
    }
    if (lane == 0) {
      __hip_atomic_store(state + 1 + chunk, ((excl + tot) << 2) | SEL_PFX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      sh_base = excl;
      if (chunk == nchunks - 1) {
        int K = excl + tot;
        if (K > cap_k) { bad |= BLISS_ERR_CAP_KEPT; K = cap_k; }                 // clamp: results invalid but in bounds
        cnt->K = K;
        sh_K = K;
      }
      if (bad) atomicOr(&cnt->err, bad);
    }
  }
  __syncthreads();
  int run = sh_base + woff;
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int j = chunk * CHUNK + (threadIdx.x >> 6) * SPAN + i * 64 + lane_id();
    if (j < C) {
      const bool keep = (mask[i] >> lane_id()) & 1ull;
      const int r = run + __popcll(mask[i] & ((1ull << lane_id()) - 1ull));
      if (keep && r < cap_k) {
        const int g = cand_nid[j];
        kept_nid[r] = g; node_prob[r] = Pv[i]; new_id[j] = r;   // :306,:309
        if (kept_map) kept_map[g] = r;                 // dense: the block passes find a kept source with ONE gather
      } else new_id[j] = -1;
    }
    run += __popcll(mask[i]);
  }
  // capacity padding (by the workgroup that knows K): ids past K stay valid node ids (0) so that padded feature gathers are harmless
  if (chunk == nchunks - 1)
    for (int r = sh_K + threadIdx.x; r < cap_k; r += TPB) { kept_nid[r] = 0; node_prob[r] = 0x3f80; }
}

// ---------------------------------------------------------------- multinomial variants (BanditLadiesSampler / LadiesSampler)
// select_neighbors draws `chosen` with torch.multinomial (bandit_sampler.py:98); generate_block then keeps
// u_nodes = union(chosen, seeds) as block sources but only edges whose SOURCE was drawn (:287-298), and uses the
// unscaled importance as P (:309).  new_id: rank in u_nodes for drawn nodes, -2 - rank for undrawn seeds, -1 otherwise.
__global__ void __launch_bounds__(TPB) k_mn_prepare(LayerCounts* cnt, const bf16_t* __restrict__ p, bf16_t* __restrict__ P,
                                                    int* __restrict__ new_id, int cap_c) {
  const int C = min(cnt->C, cap_c);
  for (int j = blockIdx.x * TPB + threadIdx.x; j < C; j += gridDim.x * TPB) { P[j] = p[j]; new_id[j] = 0; }
}
__global__ void __launch_bounds__(TPB) k_mn_mark(LayerCounts* cnt, const int* __restrict__ chosen, int n_chosen, int* __restrict__ new_id, int cap_c) {
  const int C = min(cnt->C, cap_c);
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n_chosen; i += gridDim.x * TPB) {
    const int j = chosen[i];
    if (j >= 0 && j < C) new_id[j] = 1; else atomicOr(&cnt->err, BLISS_ERR_CAP_CAND);
  }
}
template <bool EMIT>
__global__ void __launch_bounds__(TPB) k_mn_select(LayerCounts* cnt, const bf16_t* __restrict__ P, int* __restrict__ chunk_io,
                                                   const int* __restrict__ cand_nid, int* __restrict__ new_id,
                                                   int* __restrict__ kept_nid, bf16_t* __restrict__ node_prob, int cap_c, int cap_k,
                                                   int* __restrict__ kept_map) {
  __shared__ int sh4[TPB / 64];
  const int S = cnt->S, C = min(cnt->C, cap_c);
  const int nchunks = (C + CHUNK - 1) / CHUNK;
  for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    unsigned long long mask[ITEMS];
    int drawn[ITEMS];
    int wave_total = 0;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      const int j = chunk * CHUNK + (threadIdx.x >> 6) * SPAN + i * 64 + lane_id();
      drawn[i] = j < C ? (EMIT ? new_id[j] : new_id[j] == 1) : 0;
      mask[i] = __ballot(j < C && (drawn[i] == 1 || j < S));              // member of u_nodes
      wave_total += __popcll(mask[i]);
    }
    int tot;
    const int woff = chunk_wave_offset(wave_total, sh4, &tot);
    if (!EMIT) { if (threadIdx.x == 0) chunk_io[chunk] = tot; continue; }
    int run = chunk_io[chunk] + woff;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      const int j = chunk * CHUNK + (threadIdx.x >> 6) * SPAN + i * 64 + lane_id();
      if (j < C) {
        const bool in_u = (mask[i] >> lane_id()) & 1ull;
        const int r = run + __popcll(mask[i] & ((1ull << lane_id()) - 1ull));
        if (in_u && r < cap_k) {
          const int g = cand_nid[j];
          kept_nid[r] = g; node_prob[r] = P[j]; new_id[j] = drawn[i] == 1 ? r : -2 - r;
          if (kept_map && drawn[i] == 1) kept_map[g] = r;
        } else new_id[j] = -1;
      }
      run += __popcll(mask[i]);
    }
  }
  if (EMIT) {
    const int K = min(cnt->K, cap_k);
    for (int r = K + blockIdx.x * TPB + threadIdx.x; r < cap_k; r += gridDim.x * TPB) { kept_nid[r] = 0; node_prob[r] = 0x3f80; }
  }
}

// ---------------------------------------------------------------- kept edges of a wave's span, compacted
// Only ~B/E (6 % on a Reddit-like layer) of the frontier edges have a kept source, but with 64 lanes almost every wave has
// one, so per-edge work guarded by `if (kept)` is paid by everybody.  Both block passes therefore first compact the kept
// edges of the wave's 256 positions into LDS (frontier order preserved) and run the arithmetic on the dense list.
struct KeptRec { int k, pos, nid, lid; };               // seed index, CSC position, block-local source id, candidate id
__device__ __forceinline__ int span_collect(int chunk, int E, int S, const int* __restrict__ seg_ptr,
                                            const long long* __restrict__ col_base, const int* __restrict__ span_seg,
                                            const int* __restrict__ indices, const int* __restrict__ kept_map,
                                            const int* __restrict__ local_id, const int* __restrict__ new_id, KeptRec* buf) {
  int n = 0, hint = span_seg[chunk * ITEMS + (threadIdx.x >> 6)];
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int base = chunk * CHUNK + (threadIdx.x >> 6) * SPAN + i * 64;
    if (base >= E) break;                               // wave-uniform
    EdgeAt a = decode(base, E, S, seg_ptr, col_base, indices, &hint);
    int lid = 0, nid = -1;
    if (a.k >= 0) {
      if (kept_map) nid = kept_map[a.src];              // one gather
      else { lid = local_id[a.src]; nid = new_id[lid]; }
    }
    const unsigned long long mask = __ballot(nid >= 0); // :289-298 source was drawn (seeds always are)
    if (nid >= 0) {
      KeptRec r; r.k = a.k; r.pos = (int)a.pos; r.nid = nid; r.lid = lid;
      buf[n + __popcll(mask & ((1ull << lane_id()) - 1ull))] = r;
    }
    n += __popcll(mask);
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
  return n;
}

// ---------------------------------------------------------------- K_l: kept in-degree, sum of q/P per destination
template <bool BANDIT>
__global__ void __launch_bounds__(TPB) k_block_pass1(const int64_t* __restrict__ indptr, const int* __restrict__ indices,
                                                     const bf16_t* __restrict__ w, const int* __restrict__ seeds,
                                                     const int* __restrict__ seg_ptr, const long long* __restrict__ col_base,
    const int* __restrict__ span_seg, LayerCounts* cnt,
                                                     const unsigned long long* __restrict__ acc_w,
                                                     const int* __restrict__ local_id, const int* __restrict__ new_id,
                                                     const bf16_t* __restrict__ P, int* deg_blk, unsigned long long* acc_wt,
                                                     int* __restrict__ chunk_cnt, int* src_cnt, float eta_f, float ome_f,
                                                     const int* __restrict__ kept_map, const bf16_t* __restrict__ node_prob,
                                                     const uint2* __restrict__ seed_coef, KeptRec* __restrict__ kept_rec,
                                                     int* __restrict__ span_cnt, long long kept_rec_positions, const int* __restrict__ w_pend) {
  int pend;                                                 // a deferred F.normalize pass over this row (bliss_exp3_step_deferred)
  const bf16_t* __restrict__ wq = norm_state_row(w, w_pend, &pend);
  const float pdenom = renorm_denom(pend ? pend : 0x3f80);
  __shared__ int sh4[TPB / 64];
  __shared__ KeptRec sh_kept[TPB / 64][SPAN];
  const int S = cnt->S, E = cnt->E;
  if (E > kept_rec_positions) kept_rec = nullptr;       // frontier_bound is only a hint: never spill past the buffer
  const int nchunks = (E + CHUNK - 1) / CHUNK;
  KeptRec* buf = sh_kept[threadIdx.x >> 6];
  int bad = 0;
  for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const int n = span_collect(chunk, E, S, seg_ptr, col_base, span_seg, indices, kept_map, local_id, new_id, buf);
    if (kept_rec) {                                     // leave the compacted list for pass 2: it will not walk the frontier again
      const int span = chunk * ITEMS + (threadIdx.x >> 6);
      if (lane_id() == 0) span_cnt[span] = n;
      for (int j = lane_id(); j < n; j += 64) kept_rec[(size_t)span * SPAN + j] = buf[j];
    }
    for (int j0 = 0; j0 < n; j0 += 64) {                // ~B/E of the span's 256 edges: usually one trip
      const int j = j0 + lane_id();
      int key = -1;
      int64_t term = 0;
      if (j < n) {
        const KeptRec r = buf[j];
        key = r.k;
        if (src_cnt) atomicAdd(src_cnt + r.nid, 1);     // out-degree inside the block: sizes the by-source index
        if (BANDIT) {
          bf16_t q;
          const bf16_t wv = renorm_pending(wq[r.pos], pend, pdenom);
          if (seed_coef) { const uint2 cf = seed_coef[r.k]; q = edge_q_pre(wv, (bf16_t)(cf.x & 0xffffu), __uint_as_float(cf.y), ome_f); }
          else q = edge_q(wv, fixed_to_bf((int64_t)acc_w[r.k], FRAC_DST, &bad), seg_ptr[r.k + 1] - seg_ptr[r.k], eta_f, ome_f);
          bf16_t wt = f2bf(bf2f(q) / bf2f(kept_map ? node_prob[r.nid] : P[r.lid]));    // :314 e_div_u(sg, W, P)
          term = bf_to_fixed(wt, FRAC_BLK, &bad);      // :316 copy_e_sum
        }
      }
      wave_segsum_atomic_i32(key, key >= 0, deg_blk);  // :318 sg.in_degrees()
      if (BANDIT) wave_segsum_atomic_i64(key, term, acc_wt);
    }
    int tot;
    chunk_wave_offset(n, sh4, &tot);                    // (its barriers also fence buf for the next chunk)
    if (threadIdx.x == 0) chunk_cnt[chunk] = tot;
  }
  if (bad) atomicOr(&cnt->err, bad);
}

// ---------------------------------------------------------------- block CSR indptr from kept in-degrees
// block 0: exclusive scan of the kept-edge chunk counts (-> B); block 1: block CSR indptr from the kept in-degrees
__global__ void __launch_bounds__(1024) k_block_scans(int* __restrict__ chunk_cnt, const int* __restrict__ deg_blk, LayerCounts* cnt,
                                                      int* __restrict__ blk_indptr, int cap_s, int cap_b,
                                                      int* __restrict__ src_cnt, int* __restrict__ t_indptr, int cap_k) {
  __shared__ int sh[17];
  if (blockIdx.x == 2) {                               // by-source list starts; src_cnt becomes the fill cursor
    int run = 0;
    for (int base = 0; base <= cap_k; base += blockDim.x) {
      int j = base + threadIdx.x;
      int v = j < cap_k ? src_cnt[j] : 0;
      int tot, ex = block_excl_scan(v, sh, &tot);
      if (j <= cap_k) { t_indptr[j] = run + ex; if (j < cap_k) src_cnt[j] = run + ex; }
      run += tot;
    }
    return;
  }
  if (blockIdx.x == 0) {
    const int n = (cnt->E + CHUNK - 1) / CHUNK;
    int run = 0;
    for (int base = 0; base < n; base += blockDim.x) {
      int i = base + threadIdx.x;
      int v = i < n ? chunk_cnt[i] : 0;
      int tot, ex = block_excl_scan(v, sh, &tot);
      if (i < n) chunk_cnt[i] = run + ex;
      run += tot;
    }
    if (threadIdx.x == 0) {
      if (run > cap_b) { atomicOr(&cnt->err, BLISS_ERR_CAP_EDGES); run = cap_b; }   // clamp: results invalid but in bounds
      cnt->B = run;
    }
    return;
  }
  const int S = cnt->S;
  int run = 0;
  for (int base = 0; base < S; base += blockDim.x) {
    int k = base + threadIdx.x;
    int v = k < S ? deg_blk[k] : 0;
    int tot, ex = block_excl_scan(v, sh, &tot);
    if (k < S) blk_indptr[k] = run + ex;
    run += tot;
  }
  // rows S .. cap_s are empty: capacity-padded consumers (static shapes, HIP-graph replay) may walk them
  for (int k = S + threadIdx.x; k <= cap_s; k += blockDim.x) blk_indptr[k] = run;
}

// ---------------------------------------------------------------- K_n: emit the block's edges in frontier order
template <bool BANDIT>
__global__ void __launch_bounds__(TPB) k_block_pass2(const int64_t* __restrict__ indptr, const int* __restrict__ indices,
                                                     const int* __restrict__ eid_map, const bf16_t* __restrict__ w,
                                                     const int* __restrict__ seeds, const int* __restrict__ seg_ptr, const long long* __restrict__ col_base,
    const int* __restrict__ span_seg, LayerCounts* cnt, const unsigned long long* __restrict__ acc_w,
                                                     const int* __restrict__ local_id, const int* __restrict__ new_id,
                                                     const bf16_t* __restrict__ P, const int* __restrict__ deg_blk,
                                                     const unsigned long long* __restrict__ acc_wt,
                                                     const int* __restrict__ chunk_off, int* __restrict__ out_src,
                                                     int* __restrict__ out_dst, int* __restrict__ out_pos,
                                                     int* __restrict__ out_eid, bf16_t* __restrict__ out_w,
                                                     bf16_t* __restrict__ out_q, int* src_cursor, int* __restrict__ t_unsorted,
                                                     float eta_f, float ome_f, int cap_b, const int* __restrict__ kept_map,
                                                     const bf16_t* __restrict__ node_prob, const uint2* __restrict__ seed_coef,
                                                     const KeptRec* __restrict__ kept_rec, const int* __restrict__ span_cnt,
                                                     long long kept_rec_positions, const int* __restrict__ w_pend) {
  int pend;                                                 // a deferred F.normalize pass over this row (bliss_exp3_step_deferred)
  const bf16_t* __restrict__ wq = norm_state_row(w, w_pend, &pend);
  const float pdenom = renorm_denom(pend ? pend : 0x3f80);
  __shared__ int sh4[TPB / 64];
  __shared__ KeptRec sh_kept[TPB / 64][SPAN];
  const int S = cnt->S, E = cnt->E;
  if (E > kept_rec_positions) kept_rec = nullptr;       // the same decision as pass 1 (same E)
  const int nchunks = (E + CHUNK - 1) / CHUNK;
  KeptRec* buf = sh_kept[threadIdx.x >> 6];
  int bad = 0;
  for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const int span = chunk * ITEMS + (threadIdx.x >> 6);
    int n;
    const KeptRec* list = buf;
    if (kept_rec) {                                     // pass 1 left this span's kept edges, compacted and in order
      n = (span * SPAN < E) ? span_cnt[span] : 0;
      list = kept_rec + (size_t)span * SPAN;
    } else n = span_collect(chunk, E, S, seg_ptr, col_base, span_seg, indices, kept_map, local_id, new_id, buf);
    int tot;
    const int run = chunk_off[chunk] + chunk_wave_offset(n, sh4, &tot);
    for (int j = lane_id(); j < n; j += 64) {
      const int idx = run + j;
      if (idx >= cap_b) break;
      const KeptRec r = list[j];
      const int k = r.k;
      bf16_t q;
      const bf16_t wv = renorm_pending(wq[r.pos], pend, pdenom);
      if (BANDIT) {
        if (seed_coef) { const uint2 cf = seed_coef[k]; q = edge_q_pre(wv, (bf16_t)(cf.x & 0xffffu), __uint_as_float(cf.y), ome_f); }
        else q = edge_q(wv, fixed_to_bf((int64_t)acc_w[k], FRAC_DST, &bad), seg_ptr[k + 1] - seg_ptr[k], eta_f, ome_f);
      } else q = wv;
      float wt = rbf(bf2f(q) / bf2f(kept_map ? node_prob[r.nid] : P[r.lid]));   // :314
      float d = rbf((float)deg_blk[k]);                              // int -> bf16 promotion of `d`
      float out;
      if (BANDIT) {
        bf16_t wts = fixed_to_bf((int64_t)acc_wt[k], FRAC_BLK, &bad);
        float ratio = rbf(d / bf2f(wts));                            // :320  d / W_tilde_sum
        out = wt * ratio;                                            // :320  e_mul_v
      } else {
        out = wt * d;                                                // ladies_sampler.py:97
      }
      out_src[idx] = r.nid;
      if (src_cursor) t_unsorted[atomicAdd(src_cursor + r.nid, 1)] = idx;   // its source's list, arbitrary order for now
      out_dst[idx] = k;
      out_pos[idx] = r.pos;
      out_eid[idx] = eid_map ? eid_map[r.pos] : r.pos;               // :335-337
      out_w[idx] = f2bf(out);                                        // :324 edge_weights
      out_q[idx] = q;                                                // :326 q_ij
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();   // buf is rewritten by the next chunk
  }
  if (bad) atomicOr(&cnt->err, bad);
}

// leave the dense node maps clean (every touched entry back to -1)
__device__ __forceinline__ void maps_cleanup(LayerCounts* cnt, const int* __restrict__ cand_nid, int* local_id, int cap_c,
                                             const int* __restrict__ kept_nid, int* kept_map, int cap_k) {
  const int C = min(cnt->C, cap_c), K = min(cnt->K, cap_k);
  const int t0 = blockIdx.x * blockDim.x + threadIdx.x, step = gridDim.x * blockDim.x;
  for (int id = t0; id < C; id += step) local_id[cand_nid[id]] = -1;
  if (kept_map) for (int r = t0; r < K; r += step) kept_map[kept_nid[r]] = -1;
}

// ---------------------------------------------------------------- by-source lists into ascending edge order
// One wave per source.  A source has at most one edge per destination, so an edge's rank inside its list is the
// number of list members with a smaller destination: short lists (<= 64) rank in registers, longer ones through a
// per-wave LDS bitmap over the destinations (popcount prefix) -- O(len + S/32) per list, no comparison sort.
#define TSORT_MAX_S 32768
__global__ void __launch_bounds__(TPB) k_tr_sort_lists(const int* __restrict__ t_indptr, const int* __restrict__ t_unsorted,
                                                       const int* __restrict__ dst, LayerCounts* cnt, int cap_k, int cap_s,
                                                       int* __restrict__ t_edge, const int* __restrict__ cand_nid, int* local_id,
                                                       int cap_c, const int* __restrict__ kept_nid, int* kept_map, int do_cleanup) {
  extern __shared__ unsigned bm_all[];
  if (do_cleanup) maps_cleanup(cnt, cand_nid, local_id, cap_c, kept_nid, kept_map, cap_k);   // independent of the lists below
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  const int words = (cap_s + 31) / 32;
  unsigned* bm = bm_all + (size_t)wave * (words + 1);
  const int K = min(cnt->K, cap_k);
  for (int j = blockIdx.x * (TPB / 64) + wave; j < K; j += gridDim.x * (TPB / 64)) {
    const int beg = t_indptr[j], len = t_indptr[j + 1] - beg;
    if (len <= 0) continue;
    if (len <= 64) {
      const int v = lane < len ? t_unsorted[beg + lane] : 0x7fffffff;
      int rank = 0;
      for (int k = 0; k < len; ++k) rank += (__shfl(v, k) < v);
      if (lane < len) t_edge[beg + rank] = v;
      continue;
    }
    for (int w = lane; w <= words; w += 64) bm[w] = 0;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < len; i += 64) { const int d = dst[t_unsorted[beg + i]]; atomicOr(bm + (d >> 5), 1u << (d & 31)); }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
    // exclusive prefix of popcounts, kept in place in a second word array would double LDS: do it in registers
    int carry = 0;
    for (int base = 0; base < words; base += 64) {
      const int w = base + lane;
      const unsigned bits = w < words ? bm[w] : 0u;
      int c = __popc(bits), inc = c;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { int t = __shfl_up(inc, d); if (lane >= d) inc += t; }
      const int excl = carry + inc - c;
      carry += __shfl(inc, 63);
      // every list member whose destination falls in this word group finds its rank here
      for (int i = 0; i < len; i += 64) {
        const int idx = i + lane;
        int e = 0, d = -1;
        if (idx < len) { e = t_unsorted[beg + idx]; d = dst[e]; }
        const int dw = d >> 5;
        const bool mine = idx < len && dw >= base && dw < base + 64;
        const int src_lane = mine ? dw - base : 0;
        const int ex_w = __shfl(excl, src_lane);
        const unsigned bits_w = __shfl((int)bits, src_lane);
        if (mine) t_edge[beg + ex_w + __popc(bits_w & ((1u << (d & 31)) - 1u))] = e;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------- K_o: leave the dense id map clean
__global__ void __launch_bounds__(TPB) k_cleanup(LayerCounts* cnt, const int* __restrict__ cand_nid, int* local_id, int cap_c,
                                                 const int* __restrict__ kept_nid, int* kept_map, int cap_k) {
  maps_cleanup(cnt, cand_nid, local_id, cap_c, kept_nid, kept_map, cap_k);
}

inline int grid_for(int64_t n, int per_block, int max_blocks = 8192) {
  int64_t g = (n + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > max_blocks) g = max_blocks;
  return (int)g;
}

}  // namespace

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_; } while (0)

// one wave: wait until a producer on another stream has raised *flag (k_seg_scan's entry_flag), then lower it again.
// Bounded: a flag that never comes (~1 s by default, bliss_flag_set_spin_bound) sets *err and lets the stream continue, so the
// grid always drains.
static long long g_flag_spin_bound = 1ll << 22;
__global__ void __launch_bounds__(64) k_flag_wait(int* flag, int* err, long long bound) {
  if (threadIdx.x != 0) return;
  long long spins = 0;
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
    __builtin_amdgcn_s_sleep(8);
    if (++spins > bound) { if (err) atomicOr(err, BLISS_ERR_FLAG_TIMEOUT); break; }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  __hip_atomic_store(flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void k_flag_raise(int* flag) {
  if (threadIdx.x == 0) __hip_atomic_store(flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

extern "C" {

int bliss_layer_counts_bytes(void) { return (int)sizeof(LayerCounts); }

int bliss_flag_raise(int32_t* flag, void* stream) {
  if (!flag) return BLISS_EINVAL;
  k_flag_raise<<<1, 64, 0, (hipStream_t)stream>>>(flag);
  return (int)hipGetLastError();
}

int bliss_flag_wait(int32_t* flag, int32_t* err_word, void* stream) {
  if (!flag) return BLISS_EINVAL;
  k_flag_wait<<<1, 64, 0, (hipStream_t)stream>>>(flag, err_word, g_flag_spin_bound);
  return (int)hipGetLastError();
}

int bliss_flag_set_spin_bound(int64_t spins) {
  if (spins < 1) return BLISS_EINVAL;
  g_flag_spin_bound = spins;
  return 0;
}

int bliss_frontier_prob(const bliss_graph_t* g, const bliss_node_maps_t* m, const void* w_pos, const int32_t* seeds,
                        int32_t n_seeds, const int32_t* n_seeds_dev, int32_t cap_s, int mode, float eta_f,
                        float one_minus_eta_f, int64_t frontier_bound, const bliss_layer_ws_t* ws, void* stream_) {
  if (!g || !m || !seeds || !ws || !w_pos || cap_s <= 0 || !ws->hist || !ws->span_seg) return BLISS_EINVAL;
  if (n_seeds < 0 && !n_seeds_dev) return BLISS_EINVAL;
  if (n_seeds > cap_s) return BLISS_EINVAL;
  const int uniform_nodes = (mode & BLISS_MODE_UNIFORM_NODES) ? 1 : 0;
  const bool partials_only = (mode & BLISS_MODE_PARTIALS) != 0;
  mode &= ~(BLISS_MODE_UNIFORM_NODES | BLISS_MODE_PARTIALS);
  if (mode != BLISS_MODE_BANDIT && mode != BLISS_MODE_LADIES) return BLISS_EINVAL;
  if (partials_only && ws->n_bins <= 0) return BLISS_EINVAL;                 // the per-source sums come from the binned pipeline
  hipStream_t st = (hipStream_t)stream_;
  LayerCounts* cnt = (LayerCounts*)ws->counts;
  const bf16_t* w = (const bf16_t*)w_pos;
  unsigned long long* acc_w = (unsigned long long*)ws->seed_acc;            // [cap_s]
  unsigned long long* acc_q = acc_w + cap_s;                                // [cap_s]
  const long long* col_base = (const long long*)(acc_w + 5 * (size_t)cap_s);   // [cap_s], written by k_seg_scan
  if (frontier_bound < 1) frontier_bound = 1;
  const int ge = grid_for(frontier_bound, TPB), gc = grid_for(frontier_bound, CHUNK);
  const bool binned = ws->n_bins > 0;
  int log2_bins = 0, slots = 0;
  if (binned) {
    while ((1 << log2_bins) < ws->n_bins) ++log2_bins;
    slots = (g->num_nodes + ws->n_bins - 1) >> log2_bins;
    if ((1 << log2_bins) != ws->n_bins || ws->n_bins > MAX_BINS || (size_t)slots * 12 > 64 * 1024 || ws->bin_cap <= 0 ||
        g->num_edges > (long long)MAX_TILES * BTILE * 32 || !ws->bin_cursor || !ws->bin_rec || !ws->bitmap || !ws->word_prefix || !ws->touched_key || !ws->touched_sum)
      return BLISS_EINVAL;
  }
  const long long fcap = (long long)(g->num_edges < 0x7fffffffll ? g->num_edges : 0x7fffffffll);
  // the bandit's column sums write the first two per-seed accumulators themselves: the scan's zeroing leaves them alone
  const bool col_sums = binned && mode == BLISS_MODE_BANDIT;
  int* long_list = (int*)(acc_w + 7 * (size_t)cap_s);                      // [cap_s + 1], written by k_seg_scan
  PROF_LAUNCH(BK_SEG_SCAN, st, k_seg_scan<<<1 + SEG_ZERO_WGS, 1024, 0, st>>>(g->indptr, seeds, cnt, n_seeds, n_seeds_dev, cap_s, acc_w, ws->seg_ptr,
                                                            m->local_id, g->num_nodes, ws->src_cnt, ws->cap_k,
                                                            binned ? ws->bin_cursor : nullptr, ws->n_bins, (long long*)col_base, ws->span_seg,
                                                            fcap, ws->entry_flag, col_sums ? 1 : 0, col_sums ? long_list : nullptr));
  if (binned) {
    unsigned long long* seed_p2 = acc_w + 4 * (size_t)cap_s;
    uint2* seed_coef = (uint2*)(acc_w + 6 * (size_t)cap_s);               // [cap_s], written by k_col_sums
    unsigned long long* bin_rec = (unsigned long long*)ws->bin_rec;
    const int gb = grid_for(frontier_bound, BIN_BATCH);
    const int n_wave_wgs = grid_for(cap_s, COL_TPB / 64, 2048);
    if (col_sums)                                        // (the block passes need sum_j w_ij even when p_j does not)
      PROF_LAUNCH(BK_COL_SUMS, st, k_col_sums<<<n_wave_wgs + (cap_s < 2048 ? cap_s : 2048), COL_TPB, 0, st>>>(g->indptr, w, seeds, ws->seg_ptr, col_base, ws->span_seg, cnt, acc_w, acc_q, eta_f, one_minus_eta_f, seed_coef, n_wave_wgs, ws->w_pend, long_list, cap_s));
    if (mode == BLISS_MODE_BANDIT)
      PROF_LAUNCH(BK_BIN_SCATTER, st, k_bin_scatter<true><<<gb, BIN_TPB, 0, st>>>(g->indptr, g->indices, w, seeds, ws->seg_ptr, col_base, ws->span_seg, cnt, acc_w, acc_q, eta_f, one_minus_eta_f, uniform_nodes, ws->n_bins, log2_bins, ws->bin_cap, ws->bin_cursor, bin_rec, ws->bitmap, seed_coef, ws->w_pend));
    else
      PROF_LAUNCH(BK_BIN_SCATTER, st, k_bin_scatter<false><<<gb, BIN_TPB, 0, st>>>(g->indptr, g->indices, w, seeds, ws->seg_ptr, col_base, ws->span_seg, cnt, acc_w, acc_q, eta_f, one_minus_eta_f, uniform_nodes, ws->n_bins, log2_bins, ws->bin_cap, ws->bin_cursor, bin_rec, ws->bitmap, seed_coef, ws->w_pend));
    PROF_LAUNCH(BK_BIN_REDUCE, st, k_bin_reduce<<<ws->n_bins, BINRED_TPB, (size_t)slots * 12, st>>>(
        cnt, ws->n_bins, log2_bins, ws->bin_cap, ws->bin_cursor, bin_rec, g->num_nodes, slots, m->local_id, seed_p2,
        (unsigned long long*)ws->touched_key, (unsigned long long*)ws->touched_sum, ws->bitmap, ws->cap_c));
    if (partials_only) return (int)hipGetLastError();    // a shard stops here: (source, partial sum) go to the source owners
    int* tile_sum = ws->word_prefix;                     // [MAX_TILES] tile totals, then the word prefixes
    int* word_prefix = ws->word_prefix + MAX_TILES;
    PROF_LAUNCH(BK_BITMAP_SCAN, st, k_bitmap_tiles<<<grid_for(frontier_bound, (int64_t)BTILE * 32, MAX_TILES), 1024, 0, st>>>(ws->bitmap, word_prefix, tile_sum, cnt));
    int gf = (ws->cap_c + FIN_TPB - 1) / FIN_TPB;                    // (24 KiB of LDS per workgroup since round 3: several per CU)
    if (gf < 1) gf = 1;
    if (gf > 512) gf = 512;
    PROF_LAUNCH(BK_CAND_NUMBER, st, k_cand_number<<<gf, FIN_TPB, 0, st>>>(
        seeds, cnt, ws->cand_nid, m->local_id, seed_p2, (const unsigned long long*)ws->touched_key,
        (const unsigned long long*)ws->touched_sum, ws->bitmap, word_prefix, tile_sum, (bf16_t*)ws->p, ws->hist, ws->cap_c, uniform_nodes,
        FusedScale{ws->fs_ticket, ws->fs_fanout, ws->fs_eps, ws->fs_rng_ctl, ws->fs_layer_off, ws->fs_is_last, ws->fs_rng_cap, ws->chunk_cnt}));
    return (int)hipGetLastError();
  }
  if (mode == BLISS_MODE_BANDIT) {
    PROF_LAUNCH(BK_PASS1, st, k_frontier_pass1<true><<<ge, TPB, 0, st>>>(g->indptr, g->indices, w, seeds, ws->seg_ptr, col_base, ws->span_seg, cnt, m->local_id, m->first_pos, acc_w, ws->w_pend));
    PROF_LAUNCH(BK_PASS2, st, k_frontier_pass2<true><<<gc, TPB, 0, st>>>(g->indptr, g->indices, w, seeds, ws->seg_ptr, col_base, ws->span_seg, cnt, m->first_pos, acc_w, acc_q, ws->chunk_cnt, eta_f, one_minus_eta_f, ws->w_pend));
  } else {
    PROF_LAUNCH(BK_PASS1, st, k_frontier_pass1<false><<<ge, TPB, 0, st>>>(g->indptr, g->indices, w, seeds, ws->seg_ptr, col_base, ws->span_seg, cnt, m->local_id, m->first_pos, acc_w, ws->w_pend));
    PROF_LAUNCH(BK_PASS2, st, k_frontier_pass2<false><<<gc, TPB, 0, st>>>(g->indptr, g->indices, w, seeds, ws->seg_ptr, col_base, ws->span_seg, cnt, m->first_pos, acc_w, acc_q, ws->chunk_cnt, eta_f, one_minus_eta_f, ws->w_pend));
  }
  PROF_LAUNCH(BK_CHUNK_SCAN, st, k_chunk_scan<<<1, 1024, 0, st>>>(ws->chunk_cnt, cnt, 0, ws->cap_c));
  if (mode == BLISS_MODE_BANDIT)
    PROF_LAUNCH(BK_PASS3, st, k_frontier_pass3<true><<<gc, TPB, 0, st>>>(g->indptr, g->indices, w, seeds, ws->seg_ptr, col_base, ws->span_seg, cnt, m->first_pos, acc_w, acc_q, ws->chunk_cnt, m->local_id, ws->cand_nid, (unsigned long long*)m->acc_p2, eta_f, one_minus_eta_f, ws->cap_c, uniform_nodes, ws->w_pend));
  else
    PROF_LAUNCH(BK_PASS3, st, k_frontier_pass3<false><<<gc, TPB, 0, st>>>(g->indptr, g->indices, w, seeds, ws->seg_ptr, col_base, ws->span_seg, cnt, m->first_pos, acc_w, acc_q, ws->chunk_cnt, m->local_id, ws->cand_nid, (unsigned long long*)m->acc_p2, eta_f, one_minus_eta_f, ws->cap_c, uniform_nodes, ws->w_pend));
  {
    int gf = (ws->cap_c + FIN_TPB * 8 - 1) / (FIN_TPB * 8);          // ~8 candidates per thread: amortise the 128 KiB LDS zero/flush
    if (gf < 1) gf = 1;
    if (gf > 256) gf = 256;
    PROF_LAUNCH(BK_CAND_FINALIZE, st, k_cand_finalize<<<gf, FIN_TPB, 0, st>>>(
        seeds, cnt, ws->cand_nid, (unsigned long long*)m->acc_p2, m->first_pos, (bf16_t*)ws->p, ws->hist, ws->cap_c, uniform_nodes));
  }
  return (int)hipGetLastError();
}

int bliss_poisson_select(const bliss_layer_ws_t* ws, int32_t fanout, double eps, const float* uniforms,
                         int32_t* uniforms_offset_dev, int32_t* rng_ctl, int is_last, int32_t rng_cap_total,
                         int64_t cand_bound, void* stream_) {
  if (!ws || !uniforms || fanout < 0 || (rng_ctl && !uniforms_offset_dev)) return BLISS_EINVAL;
  hipStream_t st = (hipStream_t)stream_;
  LayerCounts* cnt = (LayerCounts*)ws->counts;
  if (cand_bound < 1) cand_bound = 1;
  if (cand_bound > ws->cap_c) cand_bound = ws->cap_c;
  if (!ws->fs_ticket)                                  // (else k_cand_number's last workgroup has computed the scale)
    PROF_LAUNCH(BK_POISSON_SCALE, st, k_poisson_scale<<<1, 1024, 0, st>>>(ws->hist, cnt, fanout, eps, rng_ctl, uniforms_offset_dev, is_last, rng_cap_total, ws->chunk_cnt));
  PROF_LAUNCH(BK_SELECT2, st, k_select_fused<<<grid_for(cand_bound, CHUNK, 1 << 20), TPB, 0, st>>>(
      (const bf16_t*)ws->p, uniforms, uniforms_offset_dev, cnt, (bf16_t*)ws->P, ws->chunk_cnt, ws->cand_nid, ws->new_id, ws->kept_nid,
      (bf16_t*)ws->node_prob, ws->cap_c, ws->cap_k, ws->kept_map));
  return (int)hipGetLastError();
}

int bliss_poisson_scale(int32_t* hist, void* counts, int32_t fanout, double eps, int32_t* scratch, void* stream_) {
  if (!hist || !counts || !scratch || fanout < 0) return BLISS_EINVAL;
  hipStream_t st = (hipStream_t)stream_;
  PROF_LAUNCH(BK_POISSON_SCALE, st, k_poisson_scale<<<1, 1024, 0, st>>>(hist, (LayerCounts*)counts, fanout, eps, nullptr, nullptr, 0, 0, scratch));
  return (int)hipGetLastError();
}

int bliss_multinomial_select(const bliss_layer_ws_t* ws, const int32_t* chosen, int32_t n_chosen, void* stream_) {
  if (!ws || n_chosen < 0 || (n_chosen > 0 && !chosen)) return BLISS_EINVAL;
  hipStream_t st = (hipStream_t)stream_;
  LayerCounts* cnt = (LayerCounts*)ws->counts;
  const int gc = grid_for(ws->cap_c, CHUNK), ge = grid_for(ws->cap_c, TPB);
  k_mn_prepare<<<ge, TPB, 0, st>>>(cnt, (const bf16_t*)ws->p, (bf16_t*)ws->P, ws->new_id, ws->cap_c);
  if (n_chosen > 0) k_mn_mark<<<grid_for(n_chosen, TPB), TPB, 0, st>>>(cnt, chosen, n_chosen, ws->new_id, ws->cap_c);
  k_mn_select<false><<<gc, TPB, 0, st>>>(cnt, (const bf16_t*)ws->P, ws->chunk_cnt, ws->cand_nid, ws->new_id, ws->kept_nid, (bf16_t*)ws->node_prob, ws->cap_c, ws->cap_k, ws->kept_map);
  k_chunk_scan<<<1, 1024, 0, st>>>(ws->chunk_cnt, cnt, 1, ws->cap_k);
  k_mn_select<true><<<gc, TPB, 0, st>>>(cnt, (const bf16_t*)ws->P, ws->chunk_cnt, ws->cand_nid, ws->new_id, ws->kept_nid, (bf16_t*)ws->node_prob, ws->cap_c, ws->cap_k, ws->kept_map);
  return (int)hipGetLastError();
}

int bliss_build_block(const bliss_graph_t* g, const bliss_node_maps_t* m, const void* w_pos, const int32_t* seeds,
                      int32_t cap_s, int mode, float eta_f, float one_minus_eta_f, int64_t frontier_bound,
                      const bliss_layer_ws_t* ws, const bliss_block_out_t* out, void* stream_) {
  if (!g || !m || !seeds || !ws || !out || !w_pos || cap_s <= 0 || !ws->span_seg) return BLISS_EINVAL;
  mode &= ~BLISS_MODE_UNIFORM_NODES;
  if (mode != BLISS_MODE_BANDIT && mode != BLISS_MODE_LADIES) return BLISS_EINVAL;
  hipStream_t st = (hipStream_t)stream_;
  LayerCounts* cnt = (LayerCounts*)ws->counts;
  const bf16_t* w = (const bf16_t*)w_pos;
  unsigned long long* acc_w = (unsigned long long*)ws->seed_acc;
  unsigned long long* acc_wt = acc_w + 2 * (size_t)cap_s;
  int* deg_blk = (int*)(acc_w + 3 * (size_t)cap_s);
  const long long* col_base = (const long long*)(acc_w + 5 * (size_t)cap_s);
  // per-seed coefficients exist only where k_col_sums ran (binned pipeline, bandit mode)
  const uint2* seed_coef = (ws->n_bins > 0 && mode == BLISS_MODE_BANDIT) ? (const uint2*)(acc_w + 6 * (size_t)cap_s) : nullptr;
  if (frontier_bound < 1) frontier_bound = 1;
  const int gc = grid_for(frontier_bound, CHUNK);
  // optional spill of pass 1's compacted kept-edge lists (one slot of 256 records per 256 frontier positions)
  KeptRec* kept_rec = (ws->kept_rec && ws->span_cnt && frontier_bound <= ws->kept_rec_positions) ? (KeptRec*)ws->kept_rec : nullptr;
  int* src_cnt = ws->src_cnt;
  const bool want_t = src_cnt && out->t_indptr && out->t_edge && out->t_scratch;
  if (want_t && cap_s > TSORT_MAX_S) return BLISS_EINVAL;                    // caller falls back to bliss_block_transpose
  int* sc = want_t ? src_cnt : nullptr;
  if (mode == BLISS_MODE_BANDIT)
    PROF_LAUNCH(BK_BLOCK1, st, k_block_pass1<true><<<gc, TPB, 0, st>>>(g->indptr, g->indices, w, seeds, ws->seg_ptr, col_base, ws->span_seg, cnt, acc_w, m->local_id, ws->new_id, (const bf16_t*)ws->P, deg_blk, acc_wt, ws->chunk_cnt, sc, eta_f, one_minus_eta_f, ws->kept_map, (const bf16_t*)ws->node_prob, seed_coef, kept_rec, ws->span_cnt, (long long)ws->kept_rec_positions, ws->w_pend));
  else
    PROF_LAUNCH(BK_BLOCK1, st, k_block_pass1<false><<<gc, TPB, 0, st>>>(g->indptr, g->indices, w, seeds, ws->seg_ptr, col_base, ws->span_seg, cnt, acc_w, m->local_id, ws->new_id, (const bf16_t*)ws->P, deg_blk, acc_wt, ws->chunk_cnt, sc, eta_f, one_minus_eta_f, ws->kept_map, (const bf16_t*)ws->node_prob, seed_coef, kept_rec, ws->span_cnt, (long long)ws->kept_rec_positions, ws->w_pend));
  PROF_LAUNCH(BK_INDPTR_SCAN, st, k_block_scans<<<want_t ? 3 : 2, 1024, 0, st>>>(ws->chunk_cnt, deg_blk, cnt, out->indptr, cap_s, out->cap_b, src_cnt, out->t_indptr, ws->cap_k));
  if (mode == BLISS_MODE_BANDIT)
    PROF_LAUNCH(BK_BLOCK2, st, k_block_pass2<true><<<gc, TPB, 0, st>>>(g->indptr, g->indices, g->eid, w, seeds, ws->seg_ptr, col_base, ws->span_seg, cnt, acc_w, m->local_id, ws->new_id, (const bf16_t*)ws->P, deg_blk, acc_wt, ws->chunk_cnt, out->src, out->dst, out->pos, out->eid, (bf16_t*)out->edge_weights, (bf16_t*)out->q_ij, sc, out->t_scratch, eta_f, one_minus_eta_f, out->cap_b, ws->kept_map, (const bf16_t*)ws->node_prob, seed_coef, kept_rec, ws->span_cnt, (long long)ws->kept_rec_positions, ws->w_pend));
  else
    PROF_LAUNCH(BK_BLOCK2, st, k_block_pass2<false><<<gc, TPB, 0, st>>>(g->indptr, g->indices, g->eid, w, seeds, ws->seg_ptr, col_base, ws->span_seg, cnt, acc_w, m->local_id, ws->new_id, (const bf16_t*)ws->P, deg_blk, acc_wt, ws->chunk_cnt, out->src, out->dst, out->pos, out->eid, (bf16_t*)out->edge_weights, (bf16_t*)out->q_ij, sc, out->t_scratch, eta_f, one_minus_eta_f, out->cap_b, ws->kept_map, (const bf16_t*)ws->node_prob, seed_coef, kept_rec, ws->span_cnt, (long long)ws->kept_rec_positions, ws->w_pend));
  if (want_t && ws->block_ready_flag) {
    // a consumer on another stream waits for the block itself (the forward pass): tell it before the by-source lists, which
    // only the backward pass reads, are sorted; the dense maps go back to -1 first (the next sampler's layers reuse them)
    PROF_LAUNCH(BK_CLEANUP, st, k_cleanup<<<grid_for(ws->cap_c, TPB), TPB, 0, st>>>(cnt, ws->cand_nid, m->local_id, ws->cap_c, ws->kept_nid, ws->kept_map, ws->cap_k));
    k_flag_raise<<<1, 64, 0, st>>>(ws->block_ready_flag);
    const size_t lds = (size_t)(TPB / 64) * ((cap_s + 31) / 32 + 1) * sizeof(unsigned);
    PROF_LAUNCH(BK_TRANSPOSE, st, k_tr_sort_lists<<<grid_for(ws->cap_k, TPB / 64), TPB, lds, st>>>(out->t_indptr, out->t_scratch, out->dst, cnt, ws->cap_k, cap_s, out->t_edge, ws->cand_nid, m->local_id, ws->cap_c, ws->kept_nid, ws->kept_map, 0));
  } else if (want_t) {        // the by-source lists, and (same launch) the dense maps back to -1
    const size_t lds = (size_t)(TPB / 64) * ((cap_s + 31) / 32 + 1) * sizeof(unsigned);
    PROF_LAUNCH(BK_TRANSPOSE, st, k_tr_sort_lists<<<grid_for(ws->cap_k, TPB / 64), TPB, lds, st>>>(out->t_indptr, out->t_scratch, out->dst, cnt, ws->cap_k, cap_s, out->t_edge, ws->cand_nid, m->local_id, ws->cap_c, ws->kept_nid, ws->kept_map, 1));
    if (ws->block_ready_flag) k_flag_raise<<<1, 64, 0, st>>>(ws->block_ready_flag);
  } else {
    PROF_LAUNCH(BK_CLEANUP, st, k_cleanup<<<grid_for(ws->cap_c, TPB), TPB, 0, st>>>(cnt, ws->cand_nid, m->local_id, ws->cap_c, ws->kept_nid, ws->kept_map, ws->cap_k));
    if (ws->block_ready_flag) k_flag_raise<<<1, 64, 0, st>>>(ws->block_ready_flag);
  }
  return (int)hipGetLastError();
}

}  // extern "C"
