// Per-candidate steps of the destination-range-sharded sampler (bliss_gnn_amd/shard.py): the candidate's OWNER holds the
// summed partials of compute_prob's by-source reduction (bandit_sampler.py:67-75) as a plain list and turns them into
// importances, inclusion probabilities and the draw.  The single-GPU sampler does the same inside k_cand_number /
// k_select_fused (sampler.hip) on its first-appearance-ordered candidate arrays; here the lists are in node-id order and
// the uniform of a candidate is a function of its node id (shards cannot share a serial stream).
#include "common.cuh"
#include "bliss_gnn.h"

namespace {

#define SH_TPB 256

__global__ void __launch_bounds__(SH_TPB) k_cand_importance(const int64_t* __restrict__ sums, int n, int uniform_nodes,
                                                            bf16_t* __restrict__ p, int* err) {
  int bad = 0;
  for (int i = blockIdx.x * SH_TPB + threadIdx.x; i < n; i += gridDim.x * SH_TPB) {
    const int64_t raw = sums[i];
    bf16_t pj;
    if (uniform_nodes) pj = raw ? (bf16_t)0x3f80 : (bf16_t)0;                       // :79-81 ones, 0 where out_degree == 0
    else pj = f2bf(sqrtf(bf2f(fixed_to_bf(raw, FRAC_SRC, &bad))));                  // :75 torch.sqrt(prob)
    p[i] = pj;
  }
  if (bad) atomicOr(err, bad);
}

// SplitMix64 finaliser of (seed, step, layer, node id); top 24 bits -> u = r * 2^-24   (oracle: bliss_oracle.keyed_uniform)
__device__ __forceinline__ float keyed_u24(unsigned long long key, int nid) {
  unsigned long long z = key ^ (unsigned long long)(unsigned)nid;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (float)(unsigned)(z >> 40) * (1.0f / 16777216.0f);
}

__global__ void __launch_bounds__(SH_TPB) k_keyed_select(const int* __restrict__ nid, const bf16_t* __restrict__ p,
                                                         const unsigned char* __restrict__ is_seed, int n,
                                                         const LayerCounts* __restrict__ cnt, unsigned long long key,
                                                         bf16_t* __restrict__ P, unsigned char* __restrict__ keep) {
  const int all_one = cnt->all_one;
  const float c32 = (float)cnt->c;                      // torch multiplies a bf16 tensor by a Python float in fp32
  for (int i = blockIdx.x * SH_TPB + threadIdx.x; i < n; i += gridDim.x * SH_TPB) {
    bf16_t Pv = 0x3f80;                                 // 1.0: early-out (:393) or a seed (:403-404, inf * c -> min -> 1)
    if (!all_one && !is_seed[i]) {
      const float v = rbf(bf2f(p[i]) * c32);            // :406
      Pv = (v < 1.0f || v != v) ? f2bf(v) : (bf16_t)0x3f80;
    }
    P[i] = Pv;
    keep[i] = keyed_u24(key, nid[i]) < bf2f(Pv) ? 1 : 0;   // :422-424  u24 < float(P)
  }
}

}  // namespace

extern "C" {

int bliss_cand_importance(const int64_t* sums, int32_t n, int uniform_nodes, void* p_bf16, int32_t* err, void* stream) {
  if (n < 0 || !err || (n > 0 && (!sums || !p_bf16))) return BLISS_EINVAL;
  if (n == 0) return 0;
  int grid = (n + SH_TPB - 1) / SH_TPB;
  if (grid > 1024) grid = 1024;
  k_cand_importance<<<grid, SH_TPB, 0, (hipStream_t)stream>>>(sums, n, uniform_nodes, (bf16_t*)p_bf16, err);
  return (int)hipGetLastError();
}

int bliss_keyed_select(const int32_t* nid, const void* p_bf16, const uint8_t* is_seed, int32_t n, const void* counts,
                       uint64_t seed, uint64_t step, int32_t layer, void* P_bf16, uint8_t* keep, void* stream) {
  if (n < 0 || !counts || (n > 0 && (!nid || !p_bf16 || !is_seed || !P_bf16 || !keep))) return BLISS_EINVAL;
  if (n == 0) return 0;
  // the key is FINALISED before the node id is mixed in (round-2 advice): with key = seed*G + step the step sat in the low
  // bits next to the node id, and an aligned block of 2^k ids saw the same 2^k uniforms, permuted, on every step that shares
  // (K + step) >> k -- kept counts per block constant instead of binomial.  splitmix64(key) spreads the step over all 64 bits.
  unsigned long long key = (unsigned long long)seed * 0x9E3779B97F4A7C15ull + (unsigned long long)step;
  key = (key ^ (key >> 30)) * 0xBF58476D1CE4E5B9ull;
  key = (key ^ (key >> 27)) * 0x94D049BB133111EBull;
  key ^= key >> 31;
  key ^= (unsigned long long)((unsigned)layer & 0xffu) << 56;
  int grid = (n + SH_TPB - 1) / SH_TPB;
  if (grid > 1024) grid = 1024;
  k_keyed_select<<<grid, SH_TPB, 0, (hipStream_t)stream>>>(nid, (const bf16_t*)p_bf16, is_seed, n, (const LayerCounts*)counts, key,
                                                           (bf16_t*)P_bf16, keep);
  return (int)hipGetLastError();
}

}  // extern "C"
