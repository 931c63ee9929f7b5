// Adam for the bf16 model (train_lightning.py:205-216: th.optim.Adam(self.parameters(), lr) on a module cast to bf16, so
// parameters, gradients and both moment buffers are bf16 -- no fp32 master copy, SURVEY.md section 5 "Precision").
// torch's fused multi-tensor Adam needs ~50 us for the six small tensors of the 3-layer SAGE (0.45 M parameters) inside a
// replayed HIP graph; this is ONE launch over all tensors: every value is read once, updated in fp32 and rounded once to
// bf16 (3.6 MB of traffic).  The step count and the learning rate live on the device, so a captured graph keeps working
// when StepLR changes the rate (train_lightning.py:208) and the bias corrections follow the true step.
#include "common.cuh"
#include "bliss_gnn.h"
#include <cmath>

namespace {

#define AD_TPB 256
#define AD_PER_WG (AD_TPB * 8)

struct AdamTensors {
  bf16_t* p[BLISS_ADAM_MAX_TENSORS];
  const bf16_t* g[BLISS_ADAM_MAX_TENSORS];
  bf16_t* m[BLISS_ADAM_MAX_TENSORS];
  bf16_t* v[BLISS_ADAM_MAX_TENSORS];
  long long n[BLISS_ADAM_MAX_TENSORS];
  int wg_begin[BLISS_ADAM_MAX_TENSORS + 1];
  int count;
};

// state[0] = step count (float, incremented here by the last workgroup), state[1] = learning rate, state[2] = ticket
__global__ void __launch_bounds__(AD_TPB) k_adam(AdamTensors t, float* state, float beta1, float beta2, float eps, float weight_decay) {
  int i = 0;
  while (i + 1 < t.count && (int)blockIdx.x >= t.wg_begin[i + 1]) ++i;
  const float step = state[0] + 1.0f, lr = state[1];
  // (fp32 like torch's capturable path: 1 - beta^step)
  const float bc1 = 1.0f - powf(beta1, step), bc2 = 1.0f - powf(beta2, step);
  const float step_size = lr / bc1, bc2_sqrt = sqrtf(bc2);
  bf16_t* __restrict__ p = t.p[i];
  const bf16_t* __restrict__ g = t.g[i];
  bf16_t* __restrict__ m = t.m[i];
  bf16_t* __restrict__ v = t.v[i];
  const long long n = t.n[i];
  const long long base = (long long)((int)blockIdx.x - t.wg_begin[i]) * AD_PER_WG;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const long long j = base + k * AD_TPB + threadIdx.x;
    if (j < n) {
      float gj = bf2f(g[j]);
      const float pj = bf2f(p[j]);
      if (weight_decay != 0.f) gj += weight_decay * pj;
      const float mj = bf2f(m[j]) + (gj - bf2f(m[j])) * (1.0f - beta1);           // exp_avg.lerp_(grad, 1 - beta1)
      const float vj = beta2 * bf2f(v[j]) + (1.0f - beta2) * gj * gj;             // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1 - beta2)
      const float denom = sqrtf(vj) / bc2_sqrt + eps;
      p[j] = f2bf(pj - step_size * (mj / denom));
      m[j] = f2bf(mj);
      v[j] = f2bf(vj);
    }
  }
  // every workgroup has read state[0] before the last one to finish bumps it
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned* ticket = reinterpret_cast<unsigned*>(state + 2);
    if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      state[0] = step;
    }
  }
}

}  // namespace

extern "C" int bliss_adam_step(const bliss_adam_t* t, float* state, float beta1, float beta2, float eps, float weight_decay, void* stream) {
  if (!t || !state || t->count < 1 || t->count > BLISS_ADAM_MAX_TENSORS) return BLISS_EINVAL;
  AdamTensors a;
  int total = 0;
  for (int i = 0; i < t->count; ++i) {
    if (!t->param[i] || !t->grad[i] || !t->exp_avg[i] || !t->exp_avg_sq[i] || t->numel[i] < 0) return BLISS_EINVAL;
    a.p[i] = (bf16_t*)t->param[i]; a.g[i] = (const bf16_t*)t->grad[i]; a.m[i] = (bf16_t*)t->exp_avg[i]; a.v[i] = (bf16_t*)t->exp_avg_sq[i];
    a.n[i] = t->numel[i];
    a.wg_begin[i] = total;
    total += (int)((t->numel[i] + AD_PER_WG - 1) / AD_PER_WG);
  }
  a.wg_begin[t->count] = total;
  a.count = t->count;
  if (total == 0) return 0;
  k_adam<<<total, AD_TPB, 0, (hipStream_t)stream>>>(a, state, beta1, beta2, eps, weight_decay);
  return (int)hipGetLastError();
}
