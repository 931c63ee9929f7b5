// torch's CPU random stream on the GPU: MT19937 exactly as at::mt19937 (ATen/core/MT19937RNGEngine.h)
// steps it, so that the Poisson draw consumes the SAME numbers torch.bernoulli(P) would take from
// the global CPU generator (select_neighbors, bandit_sampler.py:422-424):
// one 32-bit output per candidate, in candidate order, u = (r & 0xFFFFFF) * 2^-24.
//
// The reference pays ~1.8 ms of host time per layer for 225 K draws plus a host->device copy and a
// sync to learn the count; here the 2.5 KB generator state is uploaded once per sample_blocks call,
// the count is read from the device, and the advanced state is handed back with the step's sizes.
//
// One 256-thread workgroup regenerates the 624-word state in its three dependency phases
// (k < 227, 227 <= k < 454, 454 <= k < 624: each reads only words the previous phases finished),
// one word per thread per phase, LDS resident; tempering and the coalesced store use all lanes.
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

// state layout (device, int32[626]): s[0..623], left, next   -- the at::mt19937 fields
__global__ void __launch_bounds__(256) k_mt19937_uniform(uint32_t* state, const int* __restrict__ n_dev, int n_off_words,
                                                         float* __restrict__ out, int cap) {
  __shared__ uint32_t s[MT_N];
  const int tid = threadIdx.x;
  for (int i = tid; i < MT_N; i += 256) s[i] = state[i];
  int left = (int)state[MT_N], next = (int)state[MT_N + 1];
  __syncthreads();
  int n = n_dev[n_off_words];
  if (n > cap) n = cap;
  int done = 0;
  // values still unread in the current block: positions next .. next + (left-1) - 1
  int avail = left - 1;
  if (avail < 0) avail = 0;
  int t = avail < n ? avail : n;
  for (int i = tid; i < t; i += 256) out[i] = (float)(mt_temper(s[next + i]) & 0xffffffu) * (1.0f / 16777216.0f);
  done = t; next += t; left -= t;
  while (done < n) {
    // next_state(): three phases, old values read into registers before anyone overwrites them
    uint32_t a, b, c;
    __syncthreads();
    if (tid < MT_N - MT_M) { a = s[tid]; b = s[tid + 1]; c = s[tid + MT_M]; }
    __syncthreads();
    if (tid < MT_N - MT_M) s[tid] = c ^ mt_twist(a, b);                               // k in [0, 227)
    __syncthreads();
    if (tid < MT_N - MT_M) { int k = tid + (MT_N - MT_M); a = s[k]; b = s[k + 1]; c = s[k - (MT_N - MT_M)]; }
    __syncthreads();
    if (tid < MT_N - MT_M) s[tid + (MT_N - MT_M)] = c ^ mt_twist(a, b);               // k in [227, 454)
    __syncthreads();
    const int k3 = tid + 2 * (MT_N - MT_M);
    if (k3 < MT_N) { a = s[k3]; b = (k3 == MT_N - 1) ? s[0] : s[k3 + 1]; c = s[k3 - (MT_N - MT_M)]; }
    __syncthreads();
    if (k3 < MT_N) s[k3] = c ^ mt_twist(a, b);                                        // k in [454, 624)
    __syncthreads();
    t = (n - done) < MT_N ? (n - done) : MT_N;
    for (int i = tid; i < t; i += 256) out[done + i] = (float)(mt_temper(s[i]) & 0xffffffu) * (1.0f / 16777216.0f);
    done += t; next = t; left = MT_N + 1 - t;
  }
  __syncthreads();
  for (int i = tid; i < MT_N; i += 256) state[i] = s[i];
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
