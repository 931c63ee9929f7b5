// torch's CPU random stream on the GPU: MT19937 exactly as at::mt19937 (ATen/core/MT19937RNGEngine.h)
// steps it, so that the Poisson draw consumes the SAME numbers torch.bernoulli(P) would take from
// the global CPU generator (select_neighbors, bandit_sampler.py:422-424):
// one 32-bit output per candidate, in candidate order, u = (r & 0xFFFFFF) * 2^-24.
//
// The reference pays ~1.8 ms of host time per layer for 225 K draws plus a host->device copy and a
// sync to learn the count; here the 2.5 KB generator state is uploaded once per sample_blocks call,
// the count is read from the device, and the advanced state is handed back with the step's sizes.
//
// The 624-word recurrence is a serial chain from block to block with three dependency stages per block;
// a 256-thread workgroup walks it out of place between two LDS buffers (one word per thread per stage,
// three barriers per 624 numbers) and tempers / stores block i while stage A of block i+1 runs.
#include "common.cuh"
#include "bliss_gnn.h"
#include "prof.h"

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

// ---- speculative variant, run on a side stream BESIDE the frontier passes -----------------------------------------
// The count C is produced by those passes, so it is unknown when this kernel starts.  It generates ahead, polls C
// (agent-scope load of the counts record) after every 624-number block, and stops as soon as it has covered C -- or
// at `cap` if C never shows up (no waiting anywhere: it cannot deadlock whatever the scheduler does).  Every raw
// state block is kept so that k_mt19937_commit can hand over the generator state after EXACTLY C draws.
// raw layout: block 0 = the state at entry, block i >= 1 = the i-th regenerated state.
__global__ void __launch_bounds__(256) k_mt19937_speculate(const uint32_t* __restrict__ state, const int* n_dev, int n_off_words,
                                                           float* __restrict__ out, uint32_t* __restrict__ raw, int cap) {
  __shared__ uint32_t buf[2][MT_N];
  __shared__ int n_sh;
  const int tid = threadIdx.x;
  for (int i = tid; i < MT_N; i += 256) { const uint32_t v = state[i]; buf[0][i] = v; raw[i] = v; }
  const int left = (int)state[MT_N], next = (int)state[MT_N + 1];
  __syncthreads();
  int avail = left - 1;
  if (avail < 0) avail = 0;
  if (avail > cap) avail = cap;
  for (int i = tid; i < avail; i += 256) out[i] = mt_uniform(buf[0][next + i]);
  int done = avail, cur = 0, blk = 0, n = -1;
  while (done < cap) {
    if (tid == 0) n_sh = __hip_atomic_load(n_dev + n_off_words, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    n = n_sh;
    if (n >= 0 && done >= n) break;                    // covered
    mt_generate(buf[cur], buf[cur ^ 1], tid);          // ends with a barrier
    cur ^= 1; ++blk;
    int t = (cap - done) < MT_N ? (cap - done) : MT_N;
    for (int i = tid; i < MT_N; i += 256) {
      const uint32_t v = buf[cur][i];
      raw[blk * MT_N + i] = v;
      if (i < t) out[done + i] = mt_uniform(v);
    }
    done += t;
  }
}

// generator state after exactly C = n_dev[n_off_words] draws (C <= cap), from the raw blocks
__global__ void __launch_bounds__(256) k_mt19937_commit(uint32_t* state, const int* __restrict__ n_dev, int n_off_words,
                                                        const uint32_t* __restrict__ raw, int cap) {
  const int tid = threadIdx.x;
  int n = n_dev[n_off_words];
  if (n > cap) n = cap;
  const int left = (int)state[MT_N], next = (int)state[MT_N + 1];
  int avail = left - 1;
  if (avail < 0) avail = 0;
  if (avail > cap) avail = cap;
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

}  // namespace

extern "C" int bliss_mt19937_uniform(void* state, const int32_t* n_dev, int32_t n_word_offset, float* out, int32_t cap,
                                     void* stream) {
  if (!state || !n_dev || !out || cap < 0) return BLISS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  PROF_LAUNCH(BK_MT19937, st, k_mt19937_uniform<<<1, 256, 0, st>>>((uint32_t*)state, n_dev, n_word_offset, out, cap));
  return (int)hipGetLastError();
}

// Fork / join helpers used by bliss_frontier_prob (sampler.hip): the speculative generator runs on a library-owned
// side stream between two events, so it overlaps the frontier passes both eagerly and inside a captured HIP graph.
namespace {
hipStream_t g_side = nullptr;
hipEvent_t g_ev[16];
int g_ev_next = 0;
bool g_init = false;
}
int bliss_rng_fork(void* state, const int32_t* cnt_words, float* out, uint32_t* raw, int cap, hipStream_t st, hipEvent_t* join) {
  if (!g_init) {
    if (hipStreamCreateWithFlags(&g_side, hipStreamNonBlocking) != hipSuccess) return BLISS_EINVAL;
    for (int i = 0; i < 16; ++i) if (hipEventCreateWithFlags(&g_ev[i], hipEventDisableTiming) != hipSuccess) return BLISS_EINVAL;
    g_init = true;
  }
  hipEvent_t fork = g_ev[g_ev_next], jn = g_ev[g_ev_next + 1];
  g_ev_next = (g_ev_next + 2) % 16;
  hipError_t e;
  if ((e = hipEventRecord(fork, st)) != hipSuccess) return (int)e;
  if ((e = hipStreamWaitEvent(g_side, fork, 0)) != hipSuccess) return (int)e;
  PROF_LAUNCH(BK_MT19937, g_side, k_mt19937_speculate<<<1, 256, 0, g_side>>>((const uint32_t*)state, cnt_words, 2, out, raw, cap));
  if ((e = hipEventRecord(jn, g_side)) != hipSuccess) return (int)e;
  *join = jn;
  return 0;
}
int bliss_rng_join(void* state, const int32_t* cnt_words, const uint32_t* raw, int cap, hipStream_t st, hipEvent_t join) {
  hipError_t e;
  if ((e = hipStreamWaitEvent(st, join, 0)) != hipSuccess) return (int)e;
  k_mt19937_commit<<<1, 256, 0, st>>>((uint32_t*)state, cnt_words, 2, raw, cap);
  return (int)hipGetLastError();
}
