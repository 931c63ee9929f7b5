// torch's CPU random stream on the GPU: MT19937 exactly as at::mt19937 (ATen/core/MT19937RNGEngine.h)
// steps it, so that the Poisson draw consumes the SAME numbers torch.bernoulli(P) would take from
// the global CPU generator (select_neighbors, bandit_sampler.py:422-424):
// one 32-bit output per candidate, in candidate order, u = (r & 0xFFFFFF) * 2^-24.
//
// The reference pays ~1.8 ms of host time per layer for 225 K draws plus a host->device copy and a
// sync to learn the count; here the 2.5 KB generator state is uploaded once per sample_blocks call,
// the count is read from the device, and the advanced state is handed back with the step's sizes.
//
// The 624-word recurrence is a serial chain.  The one-shot kernel (k_mt19937_uniform) walks it with a 256-thread
// workgroup in its three dependency stages; the streaming generator (k_mt19937_stream) gives the chain to ONE wave,
// which needs no barriers inside a block, and lets the other three waves temper and store behind it.
#include "common.cuh"
#include "bliss_gnn.h"
#include "prof.h"
#include "mt_jump.h"
#include <cstdlib>
#include <mutex>
#include <vector>

namespace {

#define MT_N 624
#define MT_M 397

__device__ __forceinline__ uint32_t mt_twist(uint32_t u, uint32_t v) {
  return (((u & 0x80000000u) | (v & 0x7fffffffu)) >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

// out-of-place next_state(): nw[] from od[] (at::mt19937::next_state) in its three dependency stages
//   A: k in [0,227)    nw[k] = od[k+397] ^ twist(od[k], od[k+1])
//   B: k in [227,454)  nw[k] = nw[k-227] ^ twist(od[k], od[k+1])      (needs stage A)
//   C: k in [454,624)  nw[k] = nw[k-227] ^ twist(od[k], od[k+1]),  od[624] := nw[0]   (needs stage B)
// one word per thread per stage (256 threads), one barrier between stages; out of place, so no
// read-before-write hazards inside a stage.
__device__ __forceinline__ void mt_generate(const uint32_t* od, uint32_t* nw, int tid) {
  constexpr int H = MT_N - MT_M;   // 227
  if (tid < H) nw[tid] = od[tid + MT_M] ^ mt_twist(od[tid], od[tid + 1]);
  __syncthreads();
  if (tid < H) { const int k = tid + H; nw[k] = nw[k - H] ^ mt_twist(od[k], od[k + 1]); }
  __syncthreads();
  { const int k = tid + 2 * H; if (k < MT_N) nw[k] = nw[k - H] ^ mt_twist(od[k], k == MT_N - 1 ? nw[0] : od[k + 1]); }
  __syncthreads();
}

__device__ __forceinline__ float mt_uniform(uint32_t raw) {
  return (float)(mt_temper(raw) & 0xffffffu) * (1.0f / 16777216.0f);   // at::uniform_real_distribution<float>: 24 bits
}

// state layout (device, int32[626]): s[0..623], left, next   -- the at::mt19937 fields.
// The tempering + store of block i is issued right after block i is complete and overlaps (no barrier
// in between) with stage A of block i+1, which only reads block i.
__global__ void __launch_bounds__(256) k_mt19937_uniform(uint32_t* state, const int* __restrict__ n_dev, int n_off_words,
                                                         float* __restrict__ out, int cap) {
  __shared__ uint32_t buf[2][MT_N];
  const int tid = threadIdx.x;
  for (int i = tid; i < MT_N; i += 256) buf[0][i] = state[i];
  int left = (int)state[MT_N], next = (int)state[MT_N + 1];
  __syncthreads();
  int n = n_dev[n_off_words];
  if (n > cap) n = cap;
  int avail = left - 1;                               // values still unread in the current block
  if (avail < 0) avail = 0;
  int t = avail < n ? avail : n;
  for (int i = tid; i < t; i += 256) out[i] = mt_uniform(buf[0][next + i]);
  int done = t;
  next += t; left -= t;
  int cur = 0;
  while (done < n) {
    mt_generate(buf[cur], buf[cur ^ 1], tid);         // ends with a barrier: buf[cur^1] complete, buf[cur] free
    cur ^= 1;
    t = (n - done) < MT_N ? (n - done) : MT_N;
    for (int i = tid; i < t; i += 256) out[done + i] = mt_uniform(buf[cur][i]);
    done += t; next = t; left = MT_N + 1 - t;
  }
  __syncthreads();
  for (int i = tid; i < MT_N; i += 256) state[i] = buf[cur][i];
  if (tid == 0) { state[MT_N] = (uint32_t)left; state[MT_N + 1] = (uint32_t)next; }
}

// ---- streaming variant: ONE generator per sample_blocks call, on a side stream, beside everything else ------------
// The counts C_n of the layers become known one after the other while the layers are being sampled, but the random
// stream itself does not depend on them: layer n simply consumes the next C_n numbers.  So one 256-thread generator
// walks the recurrence for the whole call, publishing its progress; the main stream only waits (k_rng_wait, one
// wave) until the numbers a layer needs exist, and the exact generator state after sum(C_n) draws is read off the
// kept raw blocks at the end (k_mt19937_commit).  The generator never waits for anyone: it stops when it has
// covered `stop_at` (published with the last layer) or at `cap_total`, so it terminates under any scheduling.
// ctl (device int32[8]): [0] progress = numbers generated so far, [1] stop_at (-1 = unknown), [2] pos = numbers
// consumed by finished layers, [3] error flag, [4] base (offset of stream position 0 in `out`).
// raw layout: block 0 = the state at entry, block i >= 1 = the i-th regenerated state.
__device__ __forceinline__ void store_sc1_x4(float* p, float a, float b, float c, float d) {
  // 16-byte write-through store (the consumer may read while this kernel is still running); drained by the caller
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 v = {a, b, c, d};
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

// out layout: the numbers still unread in the entry block sit right below out[624] (base = 624 - avail), the k-th
// regenerated block occupies out[624 k .. 624 (k+1)) -- so every regenerated block is stored with aligned 16-byte
// stores.  Stream position p lives at out[base + p]; ctl[4] = base (set by k_rng_ctl_init).
#define MT_PUBLISH_EVERY 4

// One wave walks next_state() out of place, 64 words per step in index order.  Word k needs od[k], od[k+1] and either
// od[k+397] (k < 227) or nw[k-227] -- written at least three steps (227 words) earlier by this same wave, so LDS
// program order is all the synchronisation the recurrence needs: no barrier inside a block.
// The cross-lane dependencies through LDS are invisible to the compiler (a lane never re-reads its own addresses), so a
// compiler barrier after every step pins the program order the hardware then honours (a wave's DS operations execute in
// order).
__device__ __forceinline__ void mt_fetch(const uint32_t* od, const uint32_t* nw, int k, uint32_t& u, uint32_t& v, uint32_t& m) {
  constexpr int H = MT_N - MT_M;   // 227
  if (k < MT_N) {
    u = od[k];
    v = (k == MT_N - 1) ? nw[0] : od[k + 1];
    m = (k < H) ? od[k + MT_M] : nw[k - H];
  }
}
__device__ __forceinline__ void mt_produce_wave(const uint32_t* od, uint32_t* nw, int lane) {
  // word k of step c+2 reads nw[k-227], written by steps <= c-1: its operands can be fetched before the store of step c,
  // which keeps two LDS round trips in flight (the chain is latency-bound, not issue-bound)
  constexpr int STEPS = (MT_N + 63) / 64;
  uint32_t u = 0, v = 0, m = 0, u1 = 0, v1 = 0, m1 = 0, u2 = 0, v2 = 0, m2 = 0;
  mt_fetch(od, nw, lane, u, v, m);
  mt_fetch(od, nw, lane + 64, u1, v1, m1);
#pragma unroll
  for (int c = 0; c < STEPS; ++c) {
    const int k = c * 64 + lane;
    if (c + 2 < STEPS) mt_fetch(od, nw, k + 128, u2, v2, m2);
    if (k < MT_N) nw[k] = m ^ mt_twist(u, v);
    asm volatile("" ::: "memory");
    u = u1; v = v1; m = m1;
    u1 = u2; v1 = v2; m1 = m2;
  }
}

// The serial chain (wave 0) is kept free of everything else: waves 1..3 temper and store block i while wave 0 makes
// block i+1 (one workgroup barrier per block); progress is published every MT_PUBLISH_EVERY blocks, one period late, and
// the stop word is polled one period ahead, so that neither the write-through stores nor that load are ever waited for.
// max_blocks >= 0: stop after that many regenerated blocks (the head of the parallel generator below: the stretches take
// over from there); zero_words / zero_n: scratch of the jump kernels, cleared here (a kernel, not a memset node).
__global__ void __launch_bounds__(256) k_mt19937_stream(const uint32_t* __restrict__ state, int* ctl, float* __restrict__ out,
                                                        uint32_t* __restrict__ raw, int cap_total, int max_blocks,
                                                        uint32_t* __restrict__ zero_words, int zero_n) {
  __shared__ __attribute__((aligned(16))) uint32_t buf[2][MT_N];
  __shared__ int stop_sh;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  __builtin_amdgcn_s_setprio(3);      // the chain shares its SIMD with the sampler's waves: let the arbiter pick it first
  for (int i = tid; i < zero_n; i += 256) zero_words[i] = 0u;
  for (int i = tid; i < MT_N; i += 256) { const uint32_t v = state[i]; buf[0][i] = v; raw[i] = v; }
  const int left = (int)state[MT_N], next = (int)state[MT_N + 1];
  if (tid == 0) stop_sh = -1;
  __syncthreads();
  int avail = left - 1;
  if (avail < 0) avail = 0;
  if (avail > MT_N) avail = MT_N;
  const int base = MT_N - avail;
  for (int i = tid; i < avail; i += 256)
    __hip_atomic_store(out + base + i, mt_uniform(buf[0][next + i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int cur = 0, stop_pending = -1, published_next = avail, covered = avail;
  // top of iteration i: blocks 1..i-1 are stored, block i sits in buf[cur] (block 0 = the entry state)
  for (int i = 0;; ++i) {
    covered = avail + (i > 0 ? i - 1 : 0) * MT_N;
    const bool more = covered < cap_total && (max_blocks < 0 || i - 1 < max_blocks);
    if ((i % MT_PUBLISH_EVERY) == 0 || !more) {
      // Progress is published one period late so that nobody waits for write-through stores on the critical path: a
      // storing wave issues exactly two stores per block and nothing else, its stores retire in order, so vmcnt(2 *
      // period) means "everything up to the previous publish point has reached memory".
      if (i == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * MT_PUBLISH_EVERY) : "memory");
      if (tid == 0) stop_sh = stop_pending;
      __syncthreads();
      if (tid == 0) {
        __hip_atomic_store(ctl + 0, published_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        stop_pending = __hip_atomic_load(ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // used one period later
      }
      published_next = covered;
      const int stop = stop_sh;
      if (!more || (stop >= 0 && covered >= stop)) break;
    }
    if (wave == 0) {
      mt_produce_wave(buf[cur], buf[cur ^ 1], lane);
    } else if (i >= 1) {
      const int g = tid - 64;                             // 156 groups of four words
      if (g < MT_N / 4) {
        const uint4 v = *reinterpret_cast<const uint4*>(&buf[cur][g * 4]);
        *reinterpret_cast<uint4*>(raw + i * MT_N + g * 4) = v;          // read only after this kernel ended: plain store
        store_sc1_x4(out + i * MT_N + g * 4, mt_uniform(v.x), mt_uniform(v.y), mt_uniform(v.z), mt_uniform(v.w));
      }
    }
    __syncthreads();
    cur ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // final drain: everything generated is visible ...
  __syncthreads();
  if (tid == 0) __hip_atomic_store(ctl + 0, covered, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... and published
}


// ---- parallel variant: jump-ahead (mt_jump.hip) ---------------------------------------------------------------------
// One wave cannot walk the recurrence faster than ~0.5 us per 624 numbers, and a Reddit-like call consumes ~800 blocks.
// MT19937 is linear over GF(2): the state `n` words ahead is a fixed binary convolution of the word sequence x_1 ..
// x_{19936+624} with the coefficients of t^(n-1) mod phi (mt_jump.hip).  So:
//   1. k_mt19937_stream (above, max_blocks = b0 = 32) generates blocks 1 .. b0 as before -- which ARE x_1 .. (raw blocks);
//   2. k_mt_jump: for every stretch s >= 1 the state of block b0 + s J, the convolution split over JUMP_SLICES workgroups
//      per stretch (each takes 1024 coefficients, window of the sequence in LDS, partial states XOR-ed together);
//   3. k_mt_stretch: one workgroup per stretch generates its J blocks exactly like the serial kernel (one wave on the
//      chain, three tempering and storing); the last one to finish publishes the whole stream.
// Same numbers, same `out` / `raw` layout, same state hand-back as the serial generator.
#define JUMP_SLICE_BITS 1024
#define JUMP_DEG 19937
#define JUMP_SLICES ((JUMP_DEG + JUMP_SLICE_BITS - 1) / JUMP_SLICE_BITS)
__global__ void __launch_bounds__(256) k_mt_jump(const uint32_t* __restrict__ raw, const uint32_t* __restrict__ poly,
                                                 uint32_t* __restrict__ start) {
  __shared__ uint32_t win[JUMP_SLICE_BITS + 768];
  const int tid = threadIdx.x, s = blockIdx.y + 1;
  const int i0 = blockIdx.x * JUMP_SLICE_BITS;
  const int nbits = min(JUMP_SLICE_BITS, JUMP_DEG - i0);
  // win[j] = x_{1 + i0 + j}; new_state[k] ^= win[(i - i0) + k] for every coefficient i of this slice
  for (int j = tid; j < JUMP_SLICE_BITS + 768; j += 256) win[j] = j < nbits + MT_N ? raw[1 + i0 + j] : 0u;
  __syncthreads();
  const uint32_t* pw = poly + (size_t)s * MT_N + (i0 >> 5);
  uint32_t a0 = 0, a1 = 0, a2 = 0;
  const uint32_t* w0 = win + tid;
  for (int w = 0; w < (nbits + 31) / 32; ++w) {
    uint32_t bits = __builtin_amdgcn_readfirstlane(pw[w]);
    const int left = nbits - w * 32;
    if (left < 32) bits &= (1u << left) - 1u;
    while (bits) {                                       // wave-uniform: scalar loop over the set coefficients
      const int j = w * 32 + __builtin_ctz(bits);
      bits &= bits - 1u;
      a0 ^= w0[j]; a1 ^= w0[j + 256]; a2 ^= w0[j + 512];
    }
  }
  uint32_t* dst = start + (size_t)s * MT_N;
  if (a0) atomicXor(dst + tid, a0);
  if (a1) atomicXor(dst + tid + 256, a1);
  if (tid + 512 < MT_N && a2) atomicXor(dst + tid + 512, a2);
}

__global__ void __launch_bounds__(256) k_mt_stretch(const uint32_t* __restrict__ start, int b0, int J, int n_stretch, int* ctl,
                                                    float* __restrict__ out, uint32_t* __restrict__ raw, int* done) {
  __shared__ __attribute__((aligned(16))) uint32_t buf[2][MT_N];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, s = blockIdx.x;
  const int first = b0 + s * J;                          // block index of this stretch's start state
  const uint32_t* src = s == 0 ? raw + (size_t)b0 * MT_N : start + (size_t)s * MT_N;
  for (int i = tid; i < MT_N; i += 256) buf[0][i] = src[i];
  __syncthreads();
  int cur = 0;
  for (int i = 0; i <= J; ++i) {                         // top: block first + i sits in buf[cur], not stored yet (i >= 1)
    if (wave == 0) {
      if (i < J) mt_produce_wave(buf[cur], buf[cur ^ 1], lane);
    } else if (i >= 1) {
      const int g = tid - 64;
      if (g < MT_N / 4) {
        const uint4 v = *reinterpret_cast<const uint4*>(&buf[cur][g * 4]);
        const size_t o = (size_t)(first + i) * MT_N + g * 4;
        *reinterpret_cast<uint4*>(raw + o) = v;
        store_sc1_x4(out + o, mt_uniform(v.x), mt_uniform(v.y), mt_uniform(v.z), mt_uniform(v.w));
      }
    }
    __syncthreads();
    cur ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this stretch's numbers have reached memory ...
  __syncthreads();
  if (tid == 0) {
    const int prev = __hip_atomic_fetch_add(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == n_stretch - 1) {                         // ... and so have everybody else's: publish the whole stream
      const int avail = MT_N - __hip_atomic_load(ctl + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(ctl + 0, avail + (b0 + n_stretch * J) * MT_N, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// main stream, one wave: wait until the generator has produced the numbers layer `n` needs; hand the layer its offset
__global__ void __launch_bounds__(64) k_rng_wait(int* ctl, const int* __restrict__ cnt_words, int* __restrict__ layer_off,
                                                 int is_last, int cap_total) {
  if (threadIdx.x == 0) rng_stream_acquire(ctl, cnt_words[2], layer_off, is_last, cap_total);
}

__global__ void k_rng_ctl_init(int* ctl, const uint32_t* __restrict__ state) {
  if (threadIdx.x == 0) {
    int avail = (int)state[MT_N] - 1;
    if (avail < 0) avail = 0;
    if (avail > MT_N) avail = MT_N;
    ctl[0] = 0; ctl[1] = -1; ctl[2] = 0; ctl[3] = 0; ctl[4] = MT_N - avail;
    __hip_atomic_store(ctl + 5, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);     // see rng_stream_acquire
  }
}

// generator state after exactly n = ctl[2] draws, from the raw blocks
__global__ void __launch_bounds__(256) k_mt19937_commit(uint32_t* state, const int* __restrict__ ctl,
                                                        const uint32_t* __restrict__ raw, int cap_total, int* err_word) {
  const int tid = threadIdx.x;
  if (tid == 0 && err_word && ctl[3]) atomicOr(err_word, BLISS_ERR_RNG_STREAM);
  int n = ctl[2];
  if (n > cap_total) n = cap_total;
  const int left = (int)state[MT_N], next = (int)state[MT_N + 1];
  int avail = left - 1;
  if (avail < 0) avail = 0;
  if (avail > MT_N) avail = MT_N;
  __syncthreads();                                      // everyone has read the old fields
  if (n <= avail) {
    if (tid == 0) { state[MT_N] = (uint32_t)(left - n); state[MT_N + 1] = (uint32_t)(next + n); }
    return;
  }
  const int jp = n - avail - 1;                         // last consumed index among the regenerated numbers
  const int b = 1 + jp / MT_N, t = jp % MT_N + 1;
  for (int i = tid; i < MT_N; i += 256) state[i] = raw[b * MT_N + i];
  if (tid == 0) { state[MT_N] = (uint32_t)(MT_N + 1 - t); state[MT_N + 1] = (uint32_t)t; }
}

// k_mt19937_commit for the generator that just finished + hand-back of the finished call's counts record + k_rng_ctl_init
// for the next generator, in ONE launch on the generator's own stream (bliss_rng_stream_chain): a free-running loop keeps
// all of this off the stream its sampler and model run on.  counts_host is pinned host memory (device-visible).
__global__ void __launch_bounds__(256) k_mt19937_chain(uint32_t* state, int* ctl, const uint32_t* __restrict__ raw, int cap_total,
                                                       int* counts_dev, int n_count_words, int* counts_host) {
  const int tid = threadIdx.x;
  if (tid == 0 && ctl[3]) atomicOr(counts_dev + 5, BLISS_ERR_RNG_STREAM);     // LayerCounts::err of the first layer
  int n = ctl[2];
  if (n > cap_total) n = cap_total;
  const int left = (int)state[MT_N], next = (int)state[MT_N + 1];
  int avail = left - 1;
  if (avail < 0) avail = 0;
  if (avail > MT_N) avail = MT_N;
  __syncthreads();                                      // everyone has read the old fields
  int new_left, new_next;
  if (n <= avail) {
    new_left = left - n; new_next = next + n;
  } else {
    const int jp = n - avail - 1;                       // last consumed index among the regenerated numbers
    const int b = 1 + jp / MT_N, t = jp % MT_N + 1;
    for (int i = tid; i < MT_N; i += 256) state[i] = raw[b * MT_N + i];
    new_left = MT_N + 1 - t; new_next = t;
  }
  for (int i = tid; i < n_count_words; i += 256)
    counts_host[i] = __hip_atomic_load(counts_dev + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tid == 0) {
    state[MT_N] = (uint32_t)new_left; state[MT_N + 1] = (uint32_t)new_next;
    int a = new_left - 1;
    if (a < 0) a = 0;
    if (a > MT_N) a = MT_N;
    ctl[0] = 0; ctl[1] = -1; ctl[2] = 0; ctl[3] = 0; ctl[4] = MT_N - a;
    // the consumer (next sampler, another stream) starts reading the block once it sees this: no event needed in between
    __hip_atomic_store(ctl + 5, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

constexpr int N_EV = 48;
hipStream_t g_side = nullptr;
hipEvent_t g_ev[N_EV];
int g_ev_next = 0;
bool g_init = false;
hipEvent_t g_join = nullptr;
hipEvent_t g_ready = nullptr;     // control block of the generator started last by bliss_rng_stream_chain is initialised

int rng_init() {
  if (g_init) return 0;
  if (hipStreamCreateWithFlags(&g_side, hipStreamNonBlocking) != hipSuccess) return BLISS_EINVAL;
  for (int i = 0; i < N_EV; ++i) if (hipEventCreateWithFlags(&g_ev[i], hipEventDisableTiming) != hipSuccess) return BLISS_EINVAL;
  g_init = true;
  return 0;
}
hipEvent_t next_event() {
  hipEvent_t e = g_ev[g_ev_next];
  g_ev_next = (g_ev_next + 1) % N_EV;
  return e;
}

// the parallel generator's plan for one stream capacity (bliss_rng_prepare); device buffers live as long as the process
struct RngPlan {
  int cap_total, b0, J, n_stretch, total_blocks;
  uint32_t* d_poly;      // [n_stretch][624] coefficients of t^((b0 + s J) 624 - 1) mod phi   (s = 0 unused)
  uint32_t* d_start;     // [n_stretch][624] start states (XOR-accumulated by k_mt_jump) + the done counter behind them
};
std::vector<RngPlan> g_plans;
std::mutex g_plan_mu;

const RngPlan* find_plan(int cap_total) {
  std::lock_guard<std::mutex> lk(g_plan_mu);
  for (const auto& p : g_plans) if (p.cap_total == cap_total) return &p;
  return nullptr;
}

void plan_shape(int cap_total, RngPlan* p) {
  p->cap_total = cap_total;
  p->b0 = 33;                                            // x_1 .. x_{19936 + 624} live in raw blocks 0 .. 32
  const int need = (cap_total + MT_N - 1) / MT_N + 1;    // blocks that cover cap_total numbers whatever the entry block still holds
  const char* serial = getenv("BLISS_RNG_SERIAL");
  if ((serial && serial[0] == '1') || need <= p->b0 + 16) { p->n_stretch = 0; p->J = 0; p->total_blocks = need; return; }
  const int rest = need - p->b0;
  int n = (rest + 7) / 8;                                // >= 8 blocks per stretch; at most 32 stretches
  if (n > 32) n = 32;
  p->n_stretch = n;
  p->J = (rest + n - 1) / n;
  p->total_blocks = p->b0 + n * p->J;
}

// the generator of one call on the library's stream: serial kernel, or head + jumps + stretches
int launch_generator(const uint32_t* state, int* ctl, float* out, uint32_t* raw, int cap_total) {
  const RngPlan* p = find_plan(cap_total);
  if (!p || p->n_stretch == 0) {
    PROF_LAUNCH(BK_MT19937, g_side, k_mt19937_stream<<<1, 256, 0, g_side>>>(state, ctl, out, raw, cap_total, -1, nullptr, 0));
    return (int)hipGetLastError();
  }
  int* done = (int*)(p->d_start + (size_t)p->n_stretch * MT_N);
  PROF_LAUNCH(BK_MT19937, g_side, {
    k_mt19937_stream<<<1, 256, 0, g_side>>>(state, ctl, out, raw, cap_total, p->b0, p->d_start, p->n_stretch * MT_N + 4);
    if (p->n_stretch > 1) k_mt_jump<<<dim3(JUMP_SLICES, p->n_stretch - 1), 256, 0, g_side>>>(raw, p->d_poly, p->d_start);
    k_mt_stretch<<<p->n_stretch, 256, 0, g_side>>>(p->d_start, p->b0, p->J, p->n_stretch, ctl, out, raw, done);
  });
  return (int)hipGetLastError();
}

}  // namespace

extern "C" {

int bliss_mt19937_uniform(void* state, const int32_t* n_dev, int32_t n_word_offset, float* out, int32_t cap, void* stream) {
  if (!state || !n_dev || !out || cap < 0) return BLISS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  PROF_LAUNCH(BK_MT19937, st, k_mt19937_uniform<<<1, 256, 0, st>>>((uint32_t*)state, n_dev, n_word_offset, out, cap));
  return (int)hipGetLastError();
}

int bliss_rng_stream_begin(const void* state, int32_t* ctl, float* out, uint32_t* raw, int32_t cap_total, void* stream) {
  if (!state || !ctl || !out || !raw || cap_total <= 0) return BLISS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (rng_init()) return BLISS_EINVAL;
  hipEvent_t fork = next_event(), jn = next_event();
  hipError_t e;
  g_ready = nullptr;
  k_rng_ctl_init<<<1, 64, 0, st>>>(ctl, (const uint32_t*)state);
  if ((e = hipEventRecord(fork, st)) != hipSuccess) return (int)e;
  if ((e = hipStreamWaitEvent(g_side, fork, 0)) != hipSuccess) return (int)e;
  if (int rc = launch_generator((const uint32_t*)state, ctl, out, raw, cap_total)) return rc;
  if ((e = hipEventRecord(jn, g_side)) != hipSuccess) return (int)e;
  g_join = jn;
  return (int)hipGetLastError();
}

int bliss_rng_prepare(int32_t cap_total, int32_t* plan4) {
  if (cap_total <= 0 || !plan4) return BLISS_EINVAL;
  const RngPlan* have = find_plan(cap_total);
  RngPlan p;
  if (have) p = *have;
  else {
    plan_shape(cap_total, &p);
    p.d_poly = p.d_start = nullptr;
    if (p.n_stretch > 0) {
      std::vector<uint32_t> polys((size_t)p.n_stretch * MT_N, 0u);
      // stretch s starts at block b0 + s J: jump of (b0 + s J) 624 words from x_0, i.e. the polynomial of that minus one
      if (p.n_stretch > 1 && !mt_jump_polys((int64_t)(p.b0 + p.J) * MT_N - 1, (int64_t)p.J * MT_N, p.n_stretch - 1, polys.data() + MT_N))
        return BLISS_EINVAL;
      const size_t bytes = polys.size() * sizeof(uint32_t);
      if (hipMalloc((void**)&p.d_poly, bytes) != hipSuccess || hipMalloc((void**)&p.d_start, bytes + 64) != hipSuccess) return BLISS_EINVAL;
      if (hipMemcpy(p.d_poly, polys.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return BLISS_EINVAL;
    }
    std::lock_guard<std::mutex> lk(g_plan_mu);
    g_plans.push_back(p);
  }
  plan4[0] = p.b0; plan4[1] = p.J; plan4[2] = p.n_stretch; plan4[3] = p.total_blocks;
  return 0;
}

int bliss_rng_stream_chain(void* state, int32_t* ctl, float* out, uint32_t* raw, int32_t cap_total, int32_t* counts_dev,
                           int32_t n_count_words, int32_t* counts_host, void* stream) {
  if (!state || !ctl || !out || !raw || cap_total <= 0 || !counts_dev || !counts_host || n_count_words < 6 || !g_join) return BLISS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipEvent_t done = next_event(), ready = next_event(), jn = next_event();
  hipError_t e;
  // the generator's stream is in order: the previous generator has finished when the kernel below starts
  if ((e = hipEventRecord(done, st)) != hipSuccess) return (int)e;                   // the sampler that consumed it has finished
  if ((e = hipStreamWaitEvent(g_side, done, 0)) != hipSuccess) return (int)e;
  k_mt19937_chain<<<1, 256, 0, g_side>>>((uint32_t*)state, ctl, raw, cap_total, counts_dev, n_count_words, counts_host);
  if ((e = hipEventRecord(ready, g_side)) != hipSuccess) return (int)e;
  g_ready = ready;
  if (int rc = launch_generator((const uint32_t*)state, ctl, out, raw, cap_total)) return rc;
  if ((e = hipEventRecord(jn, g_side)) != hipSuccess) return (int)e;
  g_join = jn;
  return (int)hipGetLastError();
}

int64_t bliss_rng_stream_handle(void) {
  if (rng_init()) return 0;
  return (int64_t)(uintptr_t)g_side;
}

int bliss_rng_stream_ready(void* stream) {
  if (!g_ready) return 0;
  hipError_t e = hipStreamWaitEvent((hipStream_t)stream, g_ready, 0);
  g_ready = nullptr;
  return (int)e;
}

int bliss_rng_stream_wait(int32_t* ctl, const void* counts, int32_t* layer_off, int is_last, int32_t cap_total, void* stream) {
  if (!ctl || !counts || !layer_off) return BLISS_EINVAL;
  k_rng_wait<<<1, 64, 0, (hipStream_t)stream>>>(ctl, (const int*)counts, layer_off, is_last, cap_total);
  return (int)hipGetLastError();
}

int bliss_rng_stream_end(void* state, const int32_t* ctl, const uint32_t* raw, int32_t cap_total, int32_t* err_word, void* stream) {
  if (!state || !ctl || !raw || !g_join) return BLISS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e;
  if ((e = hipStreamWaitEvent(st, g_join, 0)) != hipSuccess) return (int)e;
  g_join = nullptr;
  k_mt19937_commit<<<1, 256, 0, st>>>((uint32_t*)state, ctl, raw, cap_total, err_word);
  return (int)hipGetLastError();
}

}  // extern "C"
