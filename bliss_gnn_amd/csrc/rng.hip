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

}  // namespace

extern "C" int bliss_mt19937_uniform(void* state, const int32_t* n_dev, int32_t n_word_offset, float* out, int32_t cap,
                                     void* stream) {
  if (!state || !n_dev || !out || cap < 0) return BLISS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  PROF_LAUNCH(BK_MT19937, st, k_mt19937_uniform<<<1, 256, 0, st>>>((uint32_t*)state, n_dev, n_word_offset, out, cap));
  return (int)hipGetLastError();
}
