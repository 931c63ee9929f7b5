/* bliss_gnn.h -- C ABI of libbliss_gnn.so, the MI355X (gfx950) implementation of the BLISS-GNN
 * hot path: layer-wise bandit / LADIES block sampling, per-block weighted SpMM, EXP3 update.
 *
 * The reference (linhthi/BLISS-GNN) has no FFI of its own: its boundary is the Python protocol
 * between dgl.dataloading.DataLoader / Lightning and the sampler + model classes.
 * Each entry point below therefore names the reference METHOD whose body it replaces; the
 * ctypes binding a maintainer would add is shown in INTEGRATION.md and is what
 * bliss_gnn_amd/_lib.py does.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless it points to one of the small descriptor structs
 *     below, which live on the host and only carry device pointers and sizes;
 *   - "bf16" = raw bfloat16 bits (uint16_t), passed as void*;
 *   - node / edge ids are int32 (train_lightning.py:340-342 casts the graph to int32), CSC
 *     indptr is int64;
 *   - `stream` is a hipStream_t cast to void* (NULL = default stream); nothing here
 *     synchronises, allocates or frees: all entry points are graph-capturable;
 *   - return value: 0 on success, a hipError_t (> 0) from the runtime, or BLISS_E* (< 0).
 *     Data-dependent failures (capacity, non-finite weights) are reported through the `err`
 *     word of the per-layer counts record, read by the caller together with the sizes.
 */
#ifndef BLISS_GNN_H
#define BLISS_GNN_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BLISS_EINVAL (-1)

#define BLISS_MODE_BANDIT 0   /* EXP3 edge probabilities  (bandit_sampler.py) */
#define BLISS_MODE_LADIES 1   /* static edge weights      (ladies_sampler.py) */
#define BLISS_ROWSUM_SLOTS 32
#define BLISS_NORM_SCRATCH (2 + 3 * BLISS_ROWSUM_SLOTS)
#define BLISS_MODE_UNIFORM_NODES 4   /* OR-ed in: importance_sampling=False, p_j = [j has an out-edge] (bandit_sampler.py:77-81) */
#define BLISS_MODE_PARTIALS 8        /* OR-ed in (bliss_frontier_prob, binned pipeline only): stop after the per-source reduction --
                                      * a destination-range shard owns only some columns, so its sums over sources are PARTIAL:
                                      * seeds' sums in seed_acc (fifth int64 array), the others as (first position << 32 | node,
                                      * sum) in touched_key / touched_sum, their number in bin_cursor[n_bins]; exact Q.44 integers
                                      * that the source owners add up (bliss_gnn_amd/shard.py) */

/* The message graph g as the sampler sees it (train_lightning.py:373: CSC only). */
typedef struct {
  const int64_t* indptr;    /* [num_nodes + 1] */
  const int32_t* indices;   /* [num_edges] source of every in-edge, grouped by destination */
  const int32_t* eid;       /* [num_edges] edge id of every CSC position, or NULL = identity */
  int32_t num_nodes;
  int64_t num_edges;
} bliss_graph_t;

/* Dense per-node scratch, owned by one sampler, clean between calls
 * (local_id = -1, first_pos = 0xFFFFFFFF, acc_p2 = 0). */
typedef struct {
  int32_t* local_id;        /* [num_nodes] */
  uint32_t* first_pos;      /* [num_nodes] */
  uint64_t* acc_p2;         /* [num_nodes] */
} bliss_node_maps_t;

/* Per-layer counts record (device).  Layout fixed: 8 x int32 then one double. */
typedef struct {
  int32_t S, E, C, K, B;    /* seeds, frontier edges, candidates, kept nodes, block edges */
  int32_t err;              /* BLISS_ERR_* bits, 0 = ok */
  int32_t iters;            /* evaluations of the Poisson fixed point (bandit_sampler.py:396) */
  int32_t all_one;          /* 1 if C <= fanout (bandit_sampler.py:392) */
  double c;                 /* Poisson scale */
} bliss_layer_counts_t;

/* Per-layer workspace (all device memory, caller allocated). */
typedef struct {
  void* counts;             /* bliss_layer_counts_t */
  int32_t* seg_ptr;         /* [cap_s + 1] start of every seed's column in the frontier */
  void* seed_acc;           /* [56 * cap_s bytes] exact per-seed accumulators, column bases, per-seed coefficients */
  int32_t* chunk_cnt;       /* [max(frontier_bound, cap_c) / 1024 + 2] */
  int32_t* cand_nid;        /* [cap_c] global id of every candidate, seeds first (ndata[NID]) */
  void* p;                  /* bf16 [cap_c] LADIES importance p_j */
  void* P;                  /* bf16 [cap_c] inclusion probability P_j */
  int32_t* new_id;          /* [cap_c] block-local id of a kept candidate, -1 otherwise */
  int32_t* kept_nid;        /* [cap_k] global id of every kept node (block srcdata[NID]) */
  void* node_prob;          /* bf16 [cap_k] srcdata['node_prob'] */
  int32_t* hist;            /* [32768] counts per bf16 bit pattern of p; zero on entry, left zero */
  int32_t* src_cnt;         /* [cap_k + 1] edges per block source / fill cursor of the by-source index, or NULL */
  int32_t cap_c, cap_k;
  /* Binned candidate pipeline (optional; n_bins = 0 selects the atomic passes, which need no extra memory).
   * n_bins: power of two <= 1024 with ceil(num_nodes / n_bins) * 12 bytes <= 64 KiB of LDS.  The buffers may be shared
   * by all layers of a call (they are dead once bliss_frontier_prob's kernels have run). */
  int32_t n_bins;
  int64_t bin_cap;          /* records per bin; a bin that overflows raises BLISS_ERR_CAP_CAND */
  int32_t* bin_cursor;      /* [n_bins + 1] */
  uint64_t* bin_rec;        /* [n_bins * bin_cap] frontier position (32) | source / n_bins (17) | bf16 term of p_j^2 (15) */
  uint32_t* bitmap;         /* [frontier_bound / 32 rounded up to 128 words, + 4] first appearances by frontier position */
  int32_t* word_prefix;     /* [2048 + same length] 2048 tile totals, then the exclusive popcount prefix of the bitmap words (per 4096-word tile) */
  uint64_t* touched_key;    /* [cap_c] (first position << 32 | source id) of every non-seed frontier source */
  uint64_t* touched_sum;    /* [cap_c] its exact sum */
  int32_t* span_seg;        /* [frontier_bound / 256 + 2] seed column in which every 256th frontier position lies */
  void* kept_rec;           /* [kept_rec_positions * 16 bytes] optional: bliss_build_block's first pass leaves its per-span kept-edge */
  int32_t* span_cnt;        /* [kept_rec_positions / 256 + 4]   lists here so that the second pass need not walk the frontier again; */
  int64_t kept_rec_positions; /* used when frontier_bound <= kept_rec_positions (a multiple of 1024), else ignored (NULL = never) */
  int32_t* kept_map;        /* [num_nodes] block-local id of a kept node, -1 everywhere on entry and on exit of
                               bliss_build_block; NULL = look kept sources up through local_id + new_id (two gathers) */
  int32_t* entry_flag;      /* optional: set to 1 by the first kernel of bliss_frontier_prob, i.e. once everything enqueued before
                               this layer has completed -- the hand-off bliss_flag_wait consumes on another stream */
  const int32_t* w_pend;    /* optional: the bliss_norm_state_t of the w_pos row handed to bliss_frontier_prob / bliss_build_block
                               (bliss_exp3_step_deferred): the weights are read from the buffer it names, and while a pass is
                               pending every weight read is divided by that norm on the fly, exactly as
                               bliss_exp3_normalize_pending will write it */
  /* optional (binned pipeline only): fs_ticket != NULL (int32 device word, zero, left zero) makes the last workgroup of
     bliss_frontier_prob's final kernel compute the Poisson scale, and bliss_poisson_select skips its own scale launch; the
     other fs_* fields are then the fanout / eps / rng_ctl / uniforms_offset_dev / is_last / rng_cap_total arguments that
     bliss_poisson_select will be called with */
  int32_t* fs_ticket; int32_t fs_fanout, fs_is_last, fs_rng_cap, fs_reserved; double fs_eps; int32_t* fs_rng_ctl; int32_t* fs_layer_off;
  int32_t* block_ready_flag; /* optional: bliss_build_block raises it (bliss_flag_wait's protocol) as soon as the block's forward
                               arrays (indptr, src, dst, pos, eid, edge_weights, q_ij, counts) and the dense maps are final --
                               before the by-source index, which only a backward pass reads, is sorted */
} bliss_layer_ws_t;

/* The block (MFG) of one layer, CSR by destination, edges in frontier order. */
typedef struct {
  int32_t* indptr;          /* [cap_s + 1] */
  int32_t* src;             /* [cap_b] block-local source id */
  int32_t* dst;             /* [cap_b] block-local destination id (= seed index) */
  int32_t* pos;             /* [cap_b] CSC position in g of every block edge */
  int32_t* eid;             /* [cap_b] edata[dgl.EID] */
  void* edge_weights;       /* bf16 [cap_b] edata['edge_weights'] */
  void* q_ij;               /* bf16 [cap_b] edata['q_ij'] (bandit) / the static weight (ladies) */
  int32_t* t_indptr;        /* [cap_k + 1] by-SOURCE index of the same edges (the SpMM backward gathers through it), */
  int32_t* t_edge;          /* [cap_b]     ascending edge index inside a source; NULL = do not build it              */
  int32_t* t_scratch;       /* [cap_b]     scratch for building it (needs ws->src_cnt and cap_s <= 32768)            */
  int32_t cap_b;
} bliss_block_out_t;

int bliss_layer_counts_bytes(void);

/* Cross-stream ordering without an event (no reference counterpart).  An event between two kernels of one stream cuts a
 * captured HIP graph in two, and every graph launch costs ~20 us on the stream it is launched on; a flag does not.
 * Enqueues a one-wave kernel on `stream` that waits until *flag != 0 (raised through bliss_layer_ws_t.entry_flag), then
 * resets it to 0.  One producer and one consumer per flag and round.  The wait is bounded (~1 s): on a timeout bit 256
 * is OR-ed into *err_word (optional) and the stream continues. */
int bliss_flag_wait(int32_t* flag, int32_t* err_word, void* stream);
/* The bound of the waits enqueued from now on, in polls (~0.25 us each; default 2^22).  A loop whose flags wait on work that
 * contains collectives raises it: a peer rank's hiccup must not read as "the flag never came". */
int bliss_flag_set_spin_bound(int64_t spins);
/* Raise *flag from `stream` (a one-thread kernel).  The hot path raises its flags from kernels it runs anyway
 * (entry_flag); this entry point exists so that a caller can PROBE, with harmless kernels, whether a wait enqueued on one
 * stream and a raise enqueued later on another really run side by side (they do not when the two streams share a
 * hardware queue, or under a profiler that serialises kernels) before it relies on flags for ordering. */
int bliss_flag_raise(int32_t* flag, void* stream);

/* exp3_probabilities + BanditLadiesSampler.compute_prob      bandit_sampler.py:101-138, :47-82
 * LadiesSampler.compute_prob                                 ladies_sampler.py:34-52
 * In: seeds (unique); their count is n_seeds if >= 0, else *n_seeds_dev (e.g. the K of the previous
 * layer's counts record, so that consecutive layers need no host round trip); cap_s >= count sizes
 * seg_ptr / seed_acc.  w_pos: bf16 [num_edges] in CSC-position order -- the layer's
 * exp3_weights row (BANDIT) or g.edata['w'] (LADIES).  eta_f = (float)eta,
 * one_minus_eta_f = (float)(1.0 - eta).  frontier_bound >= number of in-edges of the seeds
 * (num_edges is always valid).  Out (in ws): counts{S,E,C}, seg_ptr, cand_nid, p; the node maps
 * hold local ids of all candidates until bliss_build_block cleans them. */
int bliss_frontier_prob(const bliss_graph_t* g, const bliss_node_maps_t* maps, const void* w_pos,
                        const int32_t* seeds, int32_t n_seeds, const int32_t* n_seeds_dev, int32_t cap_s, int mode,
                        float eta_f, float one_minus_eta_f, int64_t frontier_bound, const bliss_layer_ws_t* ws,
                        void* stream);

/* PoissonBanditLadiesSampler.compute_prob (scale c, :391-406) + select_neighbors (:408-425).
 * uniforms: fp32 [>= C], the values torch.rand(C) draws from the CPU generator (ATen's serial
 * Bernoulli kernel consumes the same 24-bit stream); the C numbers start at uniforms[*uniforms_offset_dev]
 * (NULL = 0).  With rng_ctl (bliss_rng_stream_begin) the call itself waits for the streaming generator and
 * writes the offset, exactly like bliss_rng_stream_wait.  Out: counts{K,c,iters,all_one}, P, new_id,
 * kept_nid, node_prob. */
int bliss_poisson_select(const bliss_layer_ws_t* ws, int32_t fanout, double eps, const float* uniforms,
                         int32_t* uniforms_offset_dev, int32_t* rng_ctl, int is_last, int32_t rng_cap_total,
                         int64_t cand_bound, void* stream);

/* The non-Poisson samplers: `chosen` [n_chosen] = candidate-local ids drawn by select_neighbors
 * (torch.multinomial(prob, min(num, C)), bandit_sampler.py:98 / ladies_sampler.py:68 -- its random stream is MKL's
 * inside ATen's CPU kernel, so the draw itself stays with torch on the host, on the device-computed ws->p).  Marks
 * union(chosen, seeds) as block sources, only drawn nodes as edge sources (bandit_sampler.py:287-298), P = p (:309). */
int bliss_multinomial_select(const bliss_layer_ws_t* ws, const int32_t* chosen, int32_t n_chosen, void* stream);

/* generate_block      bandit_sampler.py:269-339 (BANDIT: Hajek weights) / ladies_sampler.py:71-107.
 * Same g, maps, w_pos, seeds, eta as the matching bliss_frontier_prob call.  Out: counts{B}, the block;
 * leaves the node maps clean. */
int bliss_build_block(const bliss_graph_t* g, const bliss_node_maps_t* maps, const void* w_pos,
                      const int32_t* seeds, int32_t cap_s, int mode, float eta_f, float one_minus_eta_f,
                      int64_t frontier_bound, const bliss_layer_ws_t* ws, const bliss_block_out_t* out, void* stream);

/* at::mt19937 on the device: n = n_dev[n_word_offset] uniforms u = (r & 0xFFFFFF) * 2^-24 from the
 * generator state (uint32[626]: 624 state words, left, next -- the fields torch.get_rng_state()
 * exposes), advanced in place.  Same numbers as torch.rand(n) / torch.bernoulli draw on the CPU. */
int bliss_mt19937_uniform(void* state, const int32_t* n_dev, int32_t n_word_offset, float* out, int32_t cap, void* stream);

/* The same generator as ONE streaming kernel per sample_blocks call, on a library-owned side stream, so that the
 * serial MT19937 recurrence overlaps the sampling kernels (eagerly and inside a captured HIP graph):
 *   begin: fork the generator from `stream`; it writes uniforms into out[cap_total + 1248] in stream order, keeps the raw
 *          state blocks in raw[624 * (cap_total / 624 + 3)] and publishes its progress in ctl (int32[8]; [5] = "initialised for the current call");
 *   wait:  (per layer, on `stream`) block `stream` until the next C numbers exist, C = the layer's counts record;
 *          *layer_off = where they start in `out`; is_last tells the generator where to stop;
 *   end:   join, then advance `state` by exactly the number of draws the layers consumed; a stream that ran
 *          short or a generator that made no progress sets bit 128 in *err_word (a counts record's err). */
int bliss_rng_stream_begin(const void* state, int32_t* ctl, float* out, uint32_t* raw, int32_t cap_total, void* stream);
/* Optional, once per cap_total, outside any stream capture (allocates, copies): lets `begin` / `chain` generate the stream
 * of a call with SEVERAL workgroups -- MT19937 jump-ahead: the state n words ahead is a fixed GF(2) convolution of the word
 * sequence with t^(n-1) mod the generator's characteristic polynomial (csrc/mt_jump.hip) -- instead of one wave walking
 * the recurrence: same numbers, same out / raw layout.  plan4 = {first parallel block, blocks per stretch, stretches (0 =
 * serial), total blocks}; `out` and `raw` must then hold 624 * (total blocks + 2) elements.  BLISS_RNG_SERIAL=1 keeps the
 * serial generator.  bliss_mt_jump_poly (host only, no GPU): the 19937 coefficients of t^n_words mod phi as 624 uint32. */
int bliss_rng_prepare(int32_t cap_total, int32_t* plan4);
int bliss_mt_jump_poly(int64_t n_words, uint32_t* poly_words);
int bliss_rng_stream_wait(int32_t* ctl, const void* counts, int32_t* layer_off, int is_last, int32_t cap_total, void* stream);
int bliss_rng_stream_end(void* state, const int32_t* ctl, const uint32_t* raw, int32_t cap_total, int32_t* err_word,
                         void* stream);
/* For a loop that samples batch after batch without a host round trip (no reference counterpart: the reference syncs
 * at every draw): `end` of the generator started last and `begin` of the next one in ONE kernel on the generator's own
 * stream, ordered after everything enqueued on `stream` so far (the sampler that consumed the numbers).  That kernel also
 * copies the finished call's counts records (counts_dev, n_count_words int32, LayerCounts::err of the first record
 * receives bit 128 as in `end`) to counts_host, which must be pinned, device-visible host memory.  Nothing is enqueued on
 * `stream` except an event record.  The sampler that consumes the new generator needs no ordering after this call: its
 * first wait (bliss_poisson_select / bliss_rng_stream_wait) also waits for the control block to be the new one (ctl[5]).
 * `ready` (optional, stricter): make `stream` itself wait until that control block is initialised. */
int bliss_rng_stream_chain(void* state, int32_t* ctl, float* out, uint32_t* raw, int32_t cap_total, int32_t* counts_dev,
                           int32_t n_count_words, int32_t* counts_host, void* stream);
int bliss_rng_stream_ready(void* stream);
/* The library-owned generator stream as an integer (hipStream_t; created on first use), so that a caller can record an
 * event behind the kernels `chain` enqueued there -- e.g. to know when counts_host has been written. */
int64_t bliss_rng_stream_handle(void);

/* ---- destination-range shards (no reference counterpart: the reference is single-device) ----------
 * The by-source reduction of compute_prob (bandit_sampler.py:67-75) crosses shards; everything per candidate then happens at
 * the candidate's owner on plain lists:
 *   bliss_cand_importance: p_j = sqrt(bf16(exact sum)) (:75) from the summed Q.44 partials (uniform_nodes: [sum != 0], :79-81);
 *   bliss_poisson_scale:   the fixed point c of :391-401 from the GLOBAL histogram of p's bit patterns (hist int32[32768],
 *                          left zeroed) and the global candidate count counts->C; writes counts->{c, iters, all_one};
 *                          scratch: int32[counts->C / 1024 + 2];
 *   bliss_keyed_select:    P_j = seed ? 1 : min(c p_j, 1) (:403-406) and the Poisson draw u < P (:422-424) with the
 *                          counter-based uniform of (seed, step, layer, node id) -- SplitMix64 finaliser, top 24 bits -- since
 *                          shards cannot share torch's serial stream; keep[j] = 1 iff drawn. */
int bliss_cand_importance(const int64_t* sums, int32_t n, int uniform_nodes, void* p_bf16, int32_t* err, void* stream);
int bliss_poisson_scale(int32_t* hist, void* counts, int32_t fanout, double eps, int32_t* scratch, void* stream);
int bliss_keyed_select(const int32_t* nid, const void* p_bf16, const uint8_t* is_seed, int32_t n, const void* counts,
                       uint64_t seed, uint64_t step, int32_t layer, void* P_bf16, uint8_t* keep, void* stream);
/* The STATIC-SHAPE sharded sampler (csrc/shard_dense.hip, bliss_gnn_amd/shard_static.py): the same split as above with every
 * exchange made dense, so that a layer has ONE collective of a fixed shape and no size ever reaches the host:
 *   bliss_shard_local_seeds:      the seeds of the global list (n_seeds, or *n_seeds_dev when n_seeds < 0) that lie in [lo, hi),
 *                                 order kept -> seeds_l (padded to cap_s with the list's first entry: a block's destination ids at capacity;
 *                                 and a second, unpadded copy: what bliss_build_block's clean-up walks),
 *                                 seed_pos[i] = position in the global list (0 beyond the count), *n_local_dev;
 *   bliss_shard_zero_dense:       dense (int64 [2 * num_nodes]) to zero, by a kernel (once, after allocation: the calls below keep it zero
 *                                 between uses);
 *   bliss_shard_scatter_partials: into a ZERO dense buffer, scatters the per-source partial sums that
 *                                 bliss_frontier_prob(BLISS_MODE_PARTIALS) left (seeds: seed_p2; others: touched_key / touched_sum,
 *                                 their count in *n_touched_dev) into dense[2 v], with dense[2 v + 1] = 1 (+ 2^32 for a seed): sum and mark side by side.
 *                                 The caller all-reduces dense (integer sums: exact for any shard count, any order);
 *   bliss_shard_candidates:       candidates = nodes with a non-zero mark, ascending id: cand_nid, p_j = sqrt(bf16(sum)) (:75); every
 *                                 entry of dense it read goes back to zero (the next bliss_shard_scatter_partials finds it clean);
 *                                 is_seed, the histogram of p's bit patterns, counts->C -- one launch, the ordered compaction by decoupled
 *                                 look-back over one 64-bit status word per 1024 nodes.  scratch: 8-byte aligned, uint64[ceil(num_nodes /
 *                                 1024) + ceil(cap_c / 1024) + 1], zero-initialised once and SHARED with bliss_shard_select_kept (each of
 *                                 the two calls returns the other's words to zero, so they alternate: candidates, select, candidates ...);
 *   (bliss_poisson_scale on that histogram and counts)
 *   bliss_shard_select_kept:      P_j and the keyed draw (as bliss_keyed_select, the step read from *step_dev), kept list =
 *                                 the seeds in seed order, then the drawn non-seeds in node order: kept_nid, node_prob, kept_map
 *                                 [kept_nid[i]] = i, layer_counts->K, layer_counts->C = *n_local_dev (bliss_build_block's clean-up
 *                                 bound); scratch: the array bliss_shard_candidates was given (same num_nodes and cap_c; ONE more 64-bit word behind the
 *                                 status words: a ticket).  When the launch's last workgroup has finished: bump_step != 0 adds 1 to *step_dev
 *                                 (a sampler's last layer), done_flag (optional) is raised (bliss_flag_wait's protocol).
 * (source: bandit_sampler.py:47-82, 381-425; no reference counterpart for the split itself -- SURVEY.md section 8e) */
int bliss_shard_local_seeds(const int32_t* seeds_g, int32_t n_seeds, const int32_t* n_seeds_dev, int32_t lo, int32_t hi, int32_t cap_s,
                            int32_t* seeds_l, int32_t* seeds_l_copy, int32_t* seed_pos, int32_t* n_local_dev, int32_t* err, void* stream);
int bliss_shard_scatter_partials(const int32_t* seeds_l, const int64_t* seed_p2, const int32_t* n_local_dev, const int64_t* touched_key,
                                 const int64_t* touched_sum, const int32_t* n_touched_dev, int64_t* dense, int32_t num_nodes,
                                 int32_t* err, void* stream);
/* block inputs, owner side (train_lightning.py:138 over shards): out[i, :] = table[nid[i] - lo, :] for the rows i < *n_rows_dev whose
 * node this rank owns (lo <= nid[i] < hi), +0 bits elsewhere -- the zero-padded buffer the ranks then sum as integer words.  bf16 rows
 * of even length, 4-byte aligned. */
int bliss_shard_pack_rows(const int32_t* nid, const int32_t* n_rows_dev, int32_t cap_rows, int32_t lo, int32_t hi, const void* table_bf16,
                          int64_t table_stride, int32_t row_len, void* out_bf16, int64_t out_stride, void* stream);
/* rows between a rank's own list and a block's list (model.py:318-332 over shards: the inputs of the layers behind the first, the
 * destination rows h[:num_dst] of every layer, and the gradients on the way back).  pos[j], j < *n_dev (<= cap_s): the position of this
 * rank's j-th row in the block's list, ASCENDING.  bf16 rows of even length, 4-byte aligned (fp32 source: 8-byte).
 *   bliss_shard_place_rows: out[r, :] = src[j, :] where pos[j] == r, +0 bits for every other row r < n_rows -- the zero-padded buffer
 *                           that is then summed over the ranks as integer words, in one launch;
 *   bliss_shard_take_rows:  out[j, :] = bf16(src[pos[j], :]) for j < *n_dev (n_dev NULL: all cap_s rows), +0 bits behind; src is bf16,
 *                           or fp32 rounded to nearest even (the all-reduced gradient buffer). */
int bliss_shard_place_rows(const void* src_bf16, int64_t src_stride, const int32_t* pos, const int32_t* n_dev, int32_t cap_s, void* out_bf16,
                           int64_t out_stride, int32_t n_rows, int32_t row_len, void* stream);
int bliss_shard_take_rows(const void* src, int32_t src_is_f32, int64_t src_stride, int32_t n_src_rows, const int32_t* pos, const int32_t* n_dev,
                          int32_t cap_s, void* out_bf16, int64_t out_stride, int32_t row_len, void* stream);
int bliss_shard_zero_dense(int64_t* dense, int32_t num_nodes, void* stream);
int bliss_shard_candidates(int64_t* dense, int32_t num_nodes, int32_t uniform_nodes, int32_t* cand_nid, void* p_bf16, uint8_t* is_seed,
                           int32_t* hist, void* counts, int32_t cap_c, int32_t* scratch, int32_t* err, void* stream);
int bliss_shard_select_kept(const int32_t* cand_nid, const void* p_bf16, const uint8_t* is_seed, const void* counts, uint64_t seed,
                            int64_t* step_dev, int32_t layer, const int32_t* seeds_g, int32_t n_seeds, const int32_t* n_seeds_dev,
                            void* P_bf16, int32_t* kept_nid, void* node_prob_bf16, int32_t* kept_map, int32_t cap_k, int32_t cap_c,
                            int32_t num_nodes, void* layer_counts, const int32_t* n_local_dev, int32_t* scratch, int32_t bump_step,
                            int32_t* done_flag, int32_t* err, void* stream);
/* bliss_exp3_normalize with the norm taken from norm_limbs (int64[3 * BLISS_ROWSUM_SLOTS], e.g. the all-reduced row sums of
 * all shards) instead of row_sum; row_sum receives the exact sum of THIS row part after the division. */
int bliss_exp3_normalize_global(void* w_pos, int64_t num_edges, int64_t* row_sum, const int64_t* norm_limbs, int64_t* scratch,
                                void* norm_out_bf16, void* stream);

/* The Linear layers of dglnn.SAGEConv (model.py:303-308, 321-329; fc_neigh / fc_self) on the matrix cores, one launch:
 *   out[r, :] = epilogue( A1[r, :] . W1^T (+ A2[r, :] . W2^T) + bias ),   r < min(m_bound, *m_dev)
 * bf16 in, fp32 accumulation (v_mfma_f32_32x32x16_bf16), one rounding to bf16; W as nn.Linear stores it ([n, k], row stride
 * w_stride elements).  ids != NULL: row r of A1 is a1[ids[r]] (the feature gather of train_lightning.py:138 as the operand
 * load); a_copy (optional) receives those rows (the weight gradients need them).  in_norm / out_norm (optional): bf16 row
 * norms of the A1 rows / of the stored output rows (model.py:318-320), in bliss_embed_norm's summation order.  relu,
 * drop_p > 0 (drop_ctr: device uint64[2] launch counter + ticket, zero-initialised; drop_seed): model.py:330-332.  Rows at or
 * beyond *m_dev (capacity padding) are written as zeros.  k1, k2 <= 1024, n <= 256.  Two argument sets may share one launch. */
typedef struct {
  const void* a1; int64_t a1_stride; const int32_t* ids;
  const void* w1; int64_t w1_stride; int32_t k1;
  const void* a2; int64_t a2_stride; const void* w2; int64_t w2_stride; int32_t k2;
  const void* bias;
  int32_t m_bound; const int32_t* m_dev; int32_t n;
  void* out; int64_t out_stride;
  void* a_copy; int64_t copy_stride;
  void* in_norm; void* out_norm;
  int32_t relu; float drop_p; uint32_t drop_seed; void* drop_ctr;
} bliss_tile_gemm_t;
int bliss_tile_gemm(const bliss_tile_gemm_t* first, const bliss_tile_gemm_t* second_or_null, void* stream);

/* The BACKWARD products of those Linear layers on the matrix cores (csrc/sage_bwd.hip) -- what autograd derives for
 * fc_neigh / fc_self of dglnn.SAGEConv (model.py:303-308, 321-329); W = [out, in] as nn.Linear stores it.
 *   bliss_sage_dgrad:  out[r, :] = A1[r, :] . W1 (+ A2[r, :] . W2 for r < min(m2_bound, *m2_dev)),  r < min(m_bound, *m_dev):
 *                      the gradient w.r.t. a layer's input rows -- A = gradient rows [m, k], W = [k, n] (k = the layer's out
 *                      features <= 256, n = its in features), fp32 accumulation over both products, one rounding to bf16;
 *                      rows at or beyond *m_dev are written as zeros.
 *   bliss_sage_wgrad:  for up to two problems in one launch pair:  dW[n, c] = sum_r D[r, n] X[r, c]  (bf16 [n_out, k_in]) and,
 *                      with db != NULL, db[n] = sum_r D[r, n]; r < min(rows_bound, *rows_dev); n_out <= 256.  fp32 partial
 *                      tiles per chunk of rows in `partials` (bliss_sage_wgrad_workspace floats, 16-byte aligned), summed in
 *                      chunk order and rounded once: bitwise reproducible. */
typedef struct {
  const void* a1; int64_t a1_stride; const void* w1; int64_t w1_stride; int32_t k1;
  const void* a2; int64_t a2_stride; const void* w2; int64_t w2_stride; int32_t k2; int32_t m2_bound; const int32_t* m2_dev;
  int32_t m_bound; const int32_t* m_dev; int32_t n;
  void* out; int64_t out_stride;
} bliss_dgrad_t;
int bliss_sage_dgrad(const bliss_dgrad_t* args, void* stream);
typedef struct {
  const void* d; int64_t d_stride; int32_t n_out;
  const void* x; int64_t x_stride; int32_t k_in;
  int32_t rows_bound; const int32_t* rows_dev;
  void* dw; int64_t dw_stride; void* db;
} bliss_wgrad_t;
int64_t bliss_sage_wgrad_workspace(const bliss_wgrad_t* probs, int32_t n_probs);
int bliss_sage_wgrad(const bliss_wgrad_t* probs, int32_t n_probs, float* partials, int64_t partial_floats, void* stream);

/* nn.CrossEntropyLoss() (mean) on bf16 logits [n_rows, n_cls] and int64 labels (train_lightning.py:77-79, :142), forward and
 * gradient in one launch: *loss_out = mean_r (logsumexp(x_r) - x_r[y_r]) in fp32, dlogits = (softmax(x) - onehot(y)) / n_rows in
 * bf16.  row_loss: float[n_rows] scratch; ticket: zero-initialised uint32 (left zero); a label outside [0, n_cls) sets bit 2 in *err. */
int bliss_cross_entropy(const void* logits, int64_t stride, const int64_t* labels, int32_t n_rows, int32_t n_cls,
                        float* row_loss, void* dlogits, int64_t d_stride, float* loss_out, uint32_t* ticket, int32_t* err,
                        void* stream);
/* The same with the two launches in front of it taken in: logits2 != NULL: the logits are bf16(logits + logits2) -- the output
 * layer's `rst = fc_self + h_neigh` (model.py:321-329); dlogits is then the gradient of both addends.  label_ids != NULL: row r's
 * label is label_table[label_ids[r]] -- mfgs[-1].dstdata['labels'] (train_lightning.py:139) without the gather.  At least one
 * of the two must be given. */
int bliss_cross_entropy_sum(const void* logits, int64_t stride, const void* logits2, int64_t stride2, const int64_t* label_table,
                            const int32_t* label_ids, int32_t n_rows, int32_t n_cls, float* row_loss, void* dlogits, int64_t d_stride,
                            float* loss_out, uint32_t* ticket, int32_t* err, void* stream);
/* The same for a rank of a sharded step (train_lightning.py:139-142 over destination-range shards): only the first *n_rows_dev of
 * the n_rows (capacity) rows are this rank's output seeds -- the others get a zero gradient row and no loss --, the labels are
 * label_table[label_ids[r] - id_off] (global node ids into the owner's table of n_table rows), and the divisor is `denom` (the
 * GLOBAL batch): *loss_out = sum_r loss_r / denom, so that the ranks' losses and gradients ADD to the global mean.  logits2 may be
 * NULL. */
int bliss_cross_entropy_masked(const void* logits, int64_t stride, const void* logits2, int64_t stride2, const int64_t* label_table,
                               int32_t n_table, const int32_t* label_ids, int32_t id_off, int32_t n_rows, const int32_t* n_rows_dev,
                               float denom, int32_t n_cls, float* row_loss, void* dlogits, int64_t d_stride, float* loss_out,
                               uint32_t* ticket, int32_t* err, void* stream);

/* th.optim.Adam(self.parameters(), lr) (train_lightning.py:205-206) for a bf16 module: parameters, gradients and both moment
 * buffers bf16, one launch over all tensors, math in fp32, one rounding per stored value.  state: float[4] on the device --
 * [0] step count (incremented by the launch), [1] learning rate (the caller rewrites it when its scheduler does,
 * train_lightning.py:208: a captured graph keeps working), [2] internal ticket (zero-initialised). */
#define BLISS_ADAM_MAX_TENSORS 32
typedef struct {
  void* param[BLISS_ADAM_MAX_TENSORS];
  const void* grad[BLISS_ADAM_MAX_TENSORS];
  void* exp_avg[BLISS_ADAM_MAX_TENSORS];
  void* exp_avg_sq[BLISS_ADAM_MAX_TENSORS];
  int64_t numel[BLISS_ADAM_MAX_TENSORS];
  int32_t count;
} bliss_adam_t;
int bliss_adam_step(const bliss_adam_t* tensors, float* state, float beta1, float beta2, float eps, float weight_decay, void* stream);

/* normalized_edata    bandit_sampler.py:20-27: w_pos[p] = bf16(1 / bf16(indeg(dst(p)))). */
int bliss_normalized_edata(const bliss_graph_t* g, void* w_pos, void* stream);

/* Feature gather      train_lightning.py:138 (blocks[0].srcdata['features'], DGL's lazy slice of g.ndata by ndata[NID]):
 * out[r, :] = feat[ids[r], :] for r < n_rows, bf16 rows of `dim` elements (strides in elements).  norm_out (optional, bf16
 * [n_rows]) receives the row norms of the gathered rows with exactly the bits bliss_embed_norm(out, ...) would produce --
 * the embed_norm model.py:318-320 takes of the same rows right afterwards.  ids must be valid row numbers (padded rows of
 * a static-shape block point at row 0).  BLISS_EINVAL if dim exceeds what one wave stages in LDS (6144 elements). */
int bliss_gather_rows(const void* feat, int64_t feat_stride, const int32_t* ids, int32_t n_rows, int32_t dim, void* out,
                      int64_t out_stride, void* norm_out, void* stream);

/* th.norm(h, dim=1)   model.py:318-320: embed_norm[j] = ||h_j||_2, fp32 accumulation, bf16 out. */
int bliss_embed_norm(const void* h, int32_t n_rows, int32_t dim, int64_t row_stride, void* out, void* stream);

/* SAGEConv 'mean' message passing with edge weights [DGL-recalled: update_all(u_mul_e, mean)],
 * model.py:321-329.  out[i,:] = (1/max(deg_i,1)) * sum_{e in row i} w_e * h[src_e,:]  (mean != 0)
 * or the plain weighted sum (mean == 0).  h bf16 [n_src, dim] (row stride in elements), w bf16
 * [nnz] or NULL (= 1), out bf16 (out_fp32 == 0) or fp32 [n_dst, dim]; src/dst [nnz] = the block's edges
 * (CSR order: dst non-decreasing).  nnz_dev (optional): the true edge count on the device, with nnz an
 * upper bound (capacity-padded arrays; rows past the true n_dst must be empty in indptr). */
int bliss_spmm_chunk_edges(int32_t nnz_bound);
int bliss_spmm_fwd(const int32_t* indptr, const int32_t* src, const int32_t* dst, const void* w, const void* h,
                   int64_t h_stride, int32_t n_dst, const int32_t* nnz_dev, int32_t nnz, int32_t dim, int mean, void* out,
                   int64_t out_stride, int out_fp32, float* partials, void* stream);

/* Backward of the above w.r.t. h: gh[j,:] = sum_{e: src_e = j} (w_e / max(deg_dst(e),1)) * gout[dst_e,:].
 * t_indptr [n_src+1], t_edge [nnz]: the block's edges grouped by SOURCE in ascending edge order.
 * partials (both calls): fp32 scratch [2 * ceil(nnz / EC) * dim] for rows cut by the EC-edge work chunks,
 * EC = bliss_spmm_chunk_edges(nnz) with the same nnz (bound) as passed to the call. */
int bliss_spmm_bwd(const int32_t* t_indptr, const int32_t* t_edge, const int32_t* src, const int32_t* dst,
                   const int32_t* indptr, const void* w, const void* gout, int64_t gout_stride, int32_t n_src,
                   const int32_t* nnz_dev, int32_t nnz, int32_t dim, int mean, void* gh, int64_t gh_stride, int out_fp32,
                   float* partials, void* stream);

/* The element-wise tail of a hidden SAGE layer in one pass (model.py:321-333 and :318-320 of the next layer):
 * out = dropout_p(relu(a + b)) row by row, norm_out[row] = ||out[row,:]||_2 (bf16, may be NULL).  a, b, out: bf16
 * [n_rows, dim].  p_drop = 0 (evaluation) makes it deterministic.  ctr: device uint64[66], zero-initialised once: the
 * dropout stream's launch counter + tickets (bumped by every call with p_drop > 0) -- counter-based bits, not torch's generator.
 * _bwd: din = dout / (1 - p) where out > 0, else 0 (the gradient w.r.t. both a and b). */
int bliss_sage_epilogue_fwd(const void* a, int64_t a_stride, const void* b, int64_t b_stride, int32_t n_rows, int32_t dim,
                            float p_drop, uint32_t seed, uint64_t* ctr, void* out, int64_t out_stride, void* norm_out, void* stream);
int bliss_sage_epilogue_bwd(const void* dout, int64_t dout_stride, const void* out, int64_t out_stride, int32_t n_rows, int32_t dim,
                            float p_drop, void* din, int64_t din_stride, void* stream);

/* The by-source index the backward needs: t_edge [cap_b] = edge indices grouped by source, ascending
 * inside a source; t_indptr [n_src_cap + 1].  The edge count is *nnz_dev if given (arrays padded to
 * cap_b; how static-shape / HIP-graph callers work), else nnz.  temp: bliss_block_transpose_temp_bytes. */
int64_t bliss_block_transpose_temp_bytes(int32_t cap_b, int32_t n_src_cap);
int bliss_block_transpose(const int32_t* src, const int32_t* nnz_dev, int32_t nnz, int32_t cap_b, int32_t n_src_cap,
                          int32_t* t_indptr, int32_t* t_edge, void* temp, int64_t temp_bytes, void* stream);

/* Graph preparation: edge list -> the int32 CSC graph the samplers read.  Replaces
 * dgl.remove_self_loop + dgl.add_self_loop (+ g.add_edges(dst, src) when undirected) + g.int() + g.formats(['csc'])
 * of train_lightning.py:334-341, 373.  Edge ids follow DGL: surviving edges renumbered in order, then the V self loops,
 * then (undirected) the reverse of every edge; columns hold their edges in ascending edge id (self loop last).
 * Outputs are sized bliss_graph_prepare_capacity() = (n_edges + V) * (undirected ? 2 : 1) (must stay < 2^31); the true
 * edge count is written to the device word n_out; err (device int32, caller-zeroed) gets bit 1 if an endpoint is outside
 * [0, V).  No host synchronisation. */
int64_t bliss_graph_prepare_capacity(int64_t n_edges, int32_t num_nodes, int undirected);
int64_t bliss_graph_prepare_temp_bytes(int64_t n_edges, int32_t num_nodes, int undirected);
int bliss_graph_prepare(const int32_t* coo_src, const int32_t* coo_dst, int64_t n_edges, int32_t num_nodes, int undirected,
                        int64_t* indptr, int32_t* indices, int32_t* eid, int64_t* n_out, int32_t* err, void* temp,
                        int64_t temp_bytes, void* stream);

/* calculate_alpha (SAGE/GCN) + calculate_rewards + update_exp3_weights up to the scatter,
 * bandit_sampler.py:157, :180-193, :221-248.  One launch per block.
 * edge_w_pos: g.edata['w'] by CSC position.  w_pos: the layer's exp3 row (updated in place).
 * row_sum: int64[3 * BLISS_ROWSUM_SLOTS] exact running sum of the row (32-bit limbs, value * 2^64; the sum is the total
 * over BLISS_ROWSUM_SLOTS replicas of three limbs, which keeps concurrent waves off a single address), updated.
 * rewards_out: bf16 [n_edges] edata['rewards'] or NULL.  factor_out: bf16 [n_edges] exp(min(1, delta r/(P n)))
 * or NULL.  apply == 0 computes rewards/factors only and leaves w_pos and row_sum untouched. */
int bliss_exp3_update(const bliss_graph_t* g, const void* edge_w_pos, void* w_pos, int64_t* row_sum,
                      const int32_t* blk_indptr, const int32_t* blk_src, const int32_t* blk_dst,
                      const int32_t* blk_pos, const void* q_ij, const void* node_prob, const void* embed_norm,
                      const void* alpha_or_null, const int32_t* dst_nid, int32_t n_dst, const int32_t* n_edges_dev,
                      int32_t edges_bound, float delta_f, void* rewards_out, void* factor_out, int apply,
                      int32_t* err, void* stream);

/* sampler.exp3(mfgs, g) (bandit_sampler.py:251-267) for ALL blocks of a step in two launches: one bliss_exp3_update
 * (apply = 1) over every block and one bliss_exp3_normalize over every layer's row.  Fields as the arguments of those
 * two calls; every block has its own w_pos row, row_sum and scratch. */
#define BLISS_EXP3_MAX_BLOCKS 8
typedef struct {
  void* w_pos; int64_t* row_sum; int64_t* scratch; void* norm_out;
  const int32_t *blk_indptr, *blk_src, *blk_dst, *blk_pos;
  const void *q_ij, *node_prob, *embed_norm, *alpha_or_null;
  const int32_t* dst_nid; const int32_t* n_edges_dev;
  void* rewards_out;
  int32_t edges_bound;
  int32_t* norm_pend;          /* bliss_exp3_step_deferred / bliss_exp3_normalize_pending only: the row's bliss_norm_state_t */
} bliss_exp3_block_t;
int bliss_exp3_step(const bliss_graph_t* g, const void* edge_w_pos, const bliss_exp3_block_t* blocks, int32_t n_blocks,
                    float delta_f, int32_t* err, void* stream);
/* Only the first of the two launches: the updates of all blocks, rows and row sums left un-normalised -- for rows that are
 * spread over several shards, whose norm is the all-reduced sum (bliss_exp3_normalize_global). */
int bliss_exp3_update_blocks(const bliss_graph_t* g, const void* edge_w_pos, const bliss_exp3_block_t* blocks, int32_t n_blocks,
                             float delta_f, int32_t* err, void* stream);
/* bliss_exp3_normalize_global for several rows in ONE launch (fields used: w_pos, row_sum, scratch, norm_out); row r's all-reduced
 * limbs at norm_limbs + r * limb_stride. */
int bliss_exp3_normalize_global_rows(const bliss_exp3_block_t* rows, int32_t n_rows, int64_t num_edges, const int64_t* norm_limbs,
                                     int64_t limb_stride, void* stream);

/* The same with F.normalize's pass over the rows taken off the caller's critical path.  The pass (4 bytes of HBM traffic per
 * edge of the graph, whenever the bf16 norm of a row is not exactly 1.0) is the only part of exp3() that scales with |E|, and
 * the next batch's sampler waits for exp3().  Every row has TWO buffers and a state record (bliss_norm_state_t, device memory,
 * zero-initialised except `alt`): bits 0-15 of `state` the bf16 norm, bit 16 "a pass is pending", bit 17 "the row currently
 * lives in the alternate buffer"; `alt` = distance of the alternate buffer from w_pos in ELEMENTS (a multiple of 8).
 * bliss_exp3_step_deferred updates the rows (in whichever buffer is current) and, per row, only DECIDES: it leaves
 * 0x10000 | norm in the state's low bits when the row needs the pass, 0 when it does not; done_flag (optional) is raised when
 * all rows are decided (bliss_flag_wait's protocol).  bliss_exp3_normalize_pending (fields used: w_pos, row_sum, scratch,
 * norm_pend) then runs the pass OUT OF PLACE -- current buffer -> other buffer, exact sums installed -- and a second small
 * launch makes the other buffer current and clears the pending bits.  It may overlap READERS of the rows that go through the
 * state record: the sampler does when bliss_layer_ws_t::w_pend points at it (a reader that started before the flip reads the
 * old buffer and applies x -> bf16(x / max(norm, 1e-12)) on the fly; one that starts after it reads the new buffer).  It must
 * not overlap a WRITER (the next update), nor the pass before it.  Deferred step + pending pass leave the bits of
 * bliss_exp3_step in the current buffer.  The immediate entry points (bliss_exp3_step, bliss_exp3_normalize, ...) know nothing of
 * the second buffer: callers switch back by copying a row that lives in the alternate buffer to w_pos and clearing bit 17. */
typedef struct { int32_t state; int32_t reserved; int64_t alt; } bliss_norm_state_t;
int bliss_exp3_step_deferred(const bliss_graph_t* g, const void* edge_w_pos, const bliss_exp3_block_t* blocks, int32_t n_blocks,
                             float delta_f, int32_t* done_flag, int32_t* err, void* stream);
int bliss_exp3_normalize_pending(const bliss_exp3_block_t* rows, int32_t n_rows, int64_t num_edges, void* stream);

/* The scatter half of update_exp3_weights (bandit_sampler.py:248) for factors computed elsewhere -- used
 * when several ranks keep replicas of the bandit state: w_pos[pos[e]] *= factor[e] for e < *n_dev, and
 * row_sum follows.  pos must be unique within one call; calls apply in stream order. */
int bliss_exp3_apply(void* w_pos, int64_t* row_sum, const int32_t* pos, const void* factor, const int32_t* n_dev,
                     int32_t n_bound, int32_t* err, void* stream);

/* The same for the lists of ALL ranks and ALL blocks of a step in ONE launch (replicas).  `gathered`
 * holds n_ranks packed buffers of rank_stride_words int32 words each (what an all-gather of every rank's buffer leaves);
 * inside a rank's buffer block b has its positions at pos_off_words[b] (int32 [bound[b]]), its bf16 factors at element
 * factor_off_bf16[b] of the buffer viewed as bf16, and its true list length at count_off_words[b].  Ranks are applied in
 * rank order (grid barrier between them: the bf16 products do not commute), blocks of one rank side by side; a position may
 * occur in several ranks' lists.  barrier: int32[2], zero on first use, left zero.  Bits identical to n_ranks x n_blocks
 * bliss_exp3_apply calls in rank order. */
typedef struct {
  void* w_pos[BLISS_EXP3_MAX_BLOCKS];
  int64_t* row_sum[BLISS_EXP3_MAX_BLOCKS];
  int32_t pos_off_words[BLISS_EXP3_MAX_BLOCKS], factor_off_bf16[BLISS_EXP3_MAX_BLOCKS], count_off_words[BLISS_EXP3_MAX_BLOCKS],
      bound[BLISS_EXP3_MAX_BLOCKS];
  int32_t n_blocks, n_ranks;
  int64_t rank_stride_words;
} bliss_exp3_rank_lists_t;
int bliss_exp3_apply_ranks(const bliss_exp3_rank_lists_t* lists, const int32_t* gathered, int32_t* barrier, int32_t* err, void* stream);

/* Fill one rank's packed buffer for that exchange in ONE launch: for every block the first min(*n_dev[b], bound[b])
 * positions (blk_pos) go to buf + pos_off_words[b], the true count *n_dev[b] to buf[count_off_words[b]] (the factors are
 * written in place by bliss_exp3_update's factor_out). */
typedef struct {
  const int32_t* pos[BLISS_EXP3_MAX_BLOCKS];
  const int32_t* n_dev[BLISS_EXP3_MAX_BLOCKS];
  int32_t pos_off_words[BLISS_EXP3_MAX_BLOCKS], count_off_words[BLISS_EXP3_MAX_BLOCKS], bound[BLISS_EXP3_MAX_BLOCKS];
  int32_t n_blocks;
} bliss_pack_lists_t;
int bliss_pack_lists(const bliss_pack_lists_t* lists, int32_t* buf, void* stream);

/* F.normalize(row, p=1, dim=0), bandit_sampler.py:249, bit-exact: norm = bf16(exact sum).  The pass
 * over the row is skipped on the device when norm == 1.0 (x / 1.0 == x).  scratch: int64[BLISS_NORM_SCRATCH], zero-initialised
 * once by the caller and left zero ([0] afterwards holds norm bits | skip << 16 | err << 20). */
int bliss_exp3_normalize(void* w_pos, int64_t num_edges, int64_t* row_sum, int64_t* scratch, void* norm_out_bf16,
                         void* stream);

/* Exact row sum from scratch (initialisation / verification): row_sum int64[3 * BLISS_ROWSUM_SLOTS]. */
int bliss_row_sum(const void* w_pos, int64_t num_edges, int64_t* row_sum, void* stream);

/* ---- GATv2 attention path (custom_GATv2Conv.forward, model.py:48-112) ------------------------------------------
 * feat: bf16 [n_src, heads*head_dim] = fc_src(h) (shared weights: the destinations are its first n_dst rows);
 * edge arrays src/dst [nnz] of the block (CSR order); nnz_dev optional as for bliss_spmm_fwd.
 *   bliss_gat_logits        e[e,h] = attn[h,:] . leaky_relu(feat[src_e,h,:] + feat[dst_e,h,:])   -> bf16 [nnz, heads]  (:82-86)
 *   bliss_gat_edge_softmax  a = softmax of x over the in-edges of every destination (:88-90), or with backward != 0
 *                           de = a * (x - sum_{e'} a x) where x = d_a
 *   bliss_gat_rows          row sums of per-edge vectors, never materialised.  which bit 0: rows are sources (through
 *                           t_edge, the by-source index) instead of destinations; bit 1: the vector is the logits
 *                           backward  coef[e,h]*attn[col]*lrelu'(feat[src]+feat[dst])  instead of  coef[e,h]*feat[nbr,col]
 *                           (which = 0 is the forward aggregation of :98, 1 its backward w.r.t. feat, 2 / 3 the two halves of
 *                           the logits backward; with which = 2 and d_attn != NULL also d_attn[col] += sum_e coef*lrelu(x)).
 *                           partials: fp32 [2 * ceil(nnz / bliss_gat_chunk_edges()) * heads*head_dim].
 *   bliss_gat_edge_dot      out[e,h] = g[dst_e,h,:] . feat[src_e,h,:]   (d_a of the aggregation)
 *   bliss_gat_alpha         calculate_alpha, model == 'gat' (bandit_sampler.py:146-154), exact sums, bf16 [nnz]. */
/* edges per work chunk of bliss_gat_rows: its partials buffer holds 2 * ceil(nnz / chunk) * heads * head_dim floats */
int bliss_gat_chunk_edges(void);
int bliss_gat_logits(const int32_t* src, const int32_t* dst, const int32_t* nnz_dev, int32_t nnz, const void* feat,
                     int64_t feat_stride, const void* attn, int32_t heads, int32_t head_dim, float negative_slope, void* e_out,
                     void* stream);
/* The same message passing with ONE workgroup per destination row (csrc/gat_fused.hip): bliss_gat_fused_fwd = logits + edge
 * softmax (+ attention dropout, model.py:88) + aggregation in one launch -- e, a, a_drop [nnz, heads] bf16 and rst [n_dst,
 * heads*head_dim] bf16 come out with the bits of the three separate kernels above; bliss_gat_fused_bwd_dst = d a, the softmax
 * backward (de [nnz, heads]), d er [n_dst, heads*head_dim] and d attn (float [heads*head_dim], summed over the rows in row order:
 * dattn_part float [n_dst, heads*head_dim], block_sums float [ceil(n_dst / 32), heads*head_dim], ticket unused) in
 * three launches.  The by-source half of the backward is bliss_gat_rows(which = 3) with g / a_drop given (see below).
 * n_dst_dev (optional) = the true row count of a capacity-padded block; drop_p > 0: drop_ctr = device uint64[2] (launch
 * counter + ticket, zero-initialised), drop_ctr_used (optional uint32) receives the counter value this launch used.
 * Supported when heads <= 8 and heads*head_dim <= 1024 (head_dim % 4 == 0) or <= 256: bliss_gat_fused_supported. */
typedef struct {
  const int32_t* indptr; const int32_t* src; int32_t n_dst; const int32_t* n_dst_dev;
  const void* feat; int64_t feat_stride; const void* attn; int32_t heads; int32_t head_dim; float negative_slope;
  void* e; void* a; void* a_drop;
  void* rst; int64_t rst_stride;
  float drop_p; uint32_t drop_seed; void* drop_ctr; void* drop_ctr_used;
  const void* g; int64_t g_stride; void* de; void* d_er; int64_t d_er_stride; float* dattn_part;
  /* virtual workgroups (required): a destination with more than bliss_gat_segment_edges() in-edges is shared by several
   * workgroups.  wg_row int32 [cap_wg, 4], 16-byte aligned (row, first edge, end, G << 16 | segment per virtual workgroup; the
   * shared rows first) + n_wg_dev from bliss_gat_segments; row_ws uint32 [n_dst * 32],
   * zero-initialised once (the kernels return it to zero); seg_part float [cap_wg * (heads*head_dim + 8)] scratch; dattn_part
   * then holds cap_wg rows and block_sums ceil(cap_wg / 32) rows; err receives BLISS_ERR_FLAG_TIMEOUT if a row's workgroups
   * fail to meet. */
  const int32_t* wg_row; const int32_t* n_wg_dev; int32_t cap_wg; void* row_ws; float* seg_part; int32_t* err;
} bliss_gat_fused_t;
int bliss_gat_segment_edges(void);
/* measurement hook (scratch/gatbench.py): while `stamps` / `stamps_bwd` is not NULL every virtual workgroup of bliss_gat_fused_fwd /
 * bliss_gat_fused_bwd_dst writes eight 100 MHz device timestamps (start, row resolved, pass 1 done, after its barrier, after the
 * exchange, pass 2 done, pass 3 done, end) to stamps[8 * workgroup ..]; NULL switches it off again.  Not part of the data path. */
int bliss_gat_fused_stamps(long long* stamps, long long* stamps_bwd);
/* the virtual workgroups' descriptors for the two kernels below: row r gets max(1, ceil(deg_r / bliss_gat_segment_edges()))
 * consecutive ids; cap_wg >= n_dst + nnz_bound / bliss_gat_segment_edges() suffices; wg_row holds 4 * cap_wg int32. */
int bliss_gat_segments(const int32_t* indptr, int32_t n_dst, int32_t cap_wg, int32_t* wg_row, int32_t* n_wg_dev, int32_t* err, void* stream);
int bliss_gat_fused_supported(int32_t heads, int32_t head_dim);
/* d el[j, :] = sum over the out-edges e = (j -> i) of  de[e, h] attn lrelu'(feat[j] + feat[i])  +  a_drop[e, h] g[i, :]  -- the
 * by-source half of the GATv2 backward (logits + aggregation) in one merge-style pass; partials as for bliss_gat_rows. */
int bliss_gat_rows_src_fused(const int32_t* t_indptr, int32_t n_src, const int32_t* t_edge, const int32_t* src, const int32_t* dst,
                             const int32_t* nnz_dev, int32_t nnz, const void* de, const void* a_drop, const void* feat, int64_t feat_stride,
                             const void* g, int64_t g_stride, const void* attn, int32_t heads, int32_t head_dim, float negative_slope,
                             void* out, int64_t out_stride, float* partials, void* stream);
int bliss_gat_fused_fwd(const bliss_gat_fused_t* args, void* stream);
int bliss_gat_fused_bwd_dst(const bliss_gat_fused_t* args, float* block_sums, float* d_attn, uint32_t* ticket, void* stream);
/* bliss_gat_logits without the intermediate bf16 roundings, float result (the north star's 1e-4 check against fp32 math);
 * likewise bliss_gat_edge_softmax with backward == 2 (float logits in, float out) and bliss_gat_rows with which == 4 (the
 * forward aggregation with float coefficients and float output rows, out_stride in floats). */
int bliss_gat_logits_f32(const int32_t* src, const int32_t* dst, const int32_t* nnz_dev, int32_t nnz, const void* feat,
                         int64_t feat_stride, const void* attn, int32_t heads, int32_t head_dim, float negative_slope, float* e_out,
                         void* stream);
int bliss_gat_edge_dot(const int32_t* src, const int32_t* dst, const int32_t* nnz_dev, int32_t nnz, const void* feat,
                       int64_t feat_stride, const void* g, int64_t g_stride, int32_t heads, int32_t head_dim, void* out, void* stream);
int bliss_gat_edge_softmax(const int32_t* indptr, int32_t n_dst, const void* x, const void* a_or_null, int32_t heads, int backward,
                           void* out, void* stream);
int bliss_gat_rows(int which, const int32_t* row_ptr, int32_t n_rows, const int32_t* t_edge, const int32_t* src, const int32_t* dst,
                   const int32_t* nnz_dev, int32_t nnz, const void* coef, const void* feat, int64_t feat_stride, const void* attn,
                   int32_t heads, int32_t head_dim, float negative_slope, void* out, int64_t out_stride, float* partials,
                   float* d_attn, void* stream);
int bliss_gat_alpha(const int32_t* indptr, int32_t n_dst, const void* q_ij, const void* a_ij, void* alpha_out, int32_t* err, void* stream);

/* Per-kernel timing with HIP events recorded on the launching stream (bench.py's roofline object).
 * bliss_prof_enable(id): -2 off (default), -1 every kernel, >= 0 one kernel id; names via
 * bliss_prof_kernel_name(id), id < bliss_prof_kernel_count().  bliss_prof_read synchronises on the
 * recorded events and returns the summed duration and the number of launches since the last reset. */
int bliss_prof_enable(int kernel_id);
int bliss_prof_reset(void);
int bliss_prof_read(int kernel_id, double* total_ms, int64_t* launches);
int bliss_prof_kernel_count(void);
const char* bliss_prof_kernel_name(int kernel_id);

#ifdef __cplusplus
}
#endif
#endif
