// nn.CrossEntropyLoss() on the batch's logits (train_lightning.py:77-79, :142) in ONE launch, forward and gradient together:
// torch runs log_softmax + nll_loss as four ~5 us kernels at the end of the forward pass (gather of the labels, softmax,
// nll reduce, a cast) and four more in the backward pass.  Here one wave takes one row: max, sum of exponentials and the
// row's loss in fp32 from the bf16 logits, and -- since d loss / d logits = (softmax - onehot) / N needs nothing else --
// the gradient row is written in the same pass (scaled by the incoming gradient in the backward, which is a single
// element-wise multiply only if that gradient is not 1).  mean reduction; the per-row losses are summed in row order by
// the last workgroup, so the result does not depend on the launch's scheduling.
#include "common.cuh"
#include "bliss_gnn.h"

namespace {

#define CE_TPB 256

// logits2 (optional): the logits are bf16(logits + logits2), the output layer's `rst = fc_self + h_neigh` (model.py:321-329) taken
// in here instead of an element-wise launch; label_ids (optional): row r's label is labels[label_ids[r]], the gather of
// train_lightning.py:139 (mfgs[-1].dstdata['labels']) likewise
__global__ void __launch_bounds__(CE_TPB) k_cross_entropy(const bf16_t* __restrict__ logits, long long stride, const bf16_t* __restrict__ logits2,
                                                          long long stride2, const long long* __restrict__ labels,
                                                          const int* __restrict__ label_ids,
                                                          int n_rows, int n_cls, float* __restrict__ row_loss, bf16_t* __restrict__ dlogits,
                                                          long long d_stride, float* __restrict__ loss_out, unsigned* ticket, int* err,
                                                          const int* __restrict__ n_rows_dev, float denom, int id_off, int n_table) {
  // masked form (bliss_cross_entropy_masked): only the first *n_rows_dev rows count (the others get a zero gradient row and no
  // loss), the divisor is `denom` (a global batch, not this rank's rows), label ids are node ids minus id_off into a table of
  // n_table rows
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  const float inv_n = 1.0f / (denom > 0.f ? denom : (float)n_rows);
  int n_valid = n_rows;
  if (n_rows_dev) { const int v = *n_rows_dev; n_valid = v < n_rows ? (v < 0 ? 0 : v) : n_rows; }
  for (int r = blockIdx.x * (CE_TPB / 64) + wave; r < n_rows; r += gridDim.x * (CE_TPB / 64)) {
    if (r >= n_valid) {                                 // (wave-uniform)
      bf16_t* g0 = dlogits + (long long)r * d_stride;
      for (int c = lane; c < n_cls; c += 64) g0[c] = 0;
      if (lane == 0) row_loss[r] = 0.f;
      continue;
    }
    const bf16_t* x1 = logits + (long long)r * stride;
    const bf16_t* x2 = logits2 ? logits2 + (long long)r * stride2 : nullptr;
    auto X = [&](int c) { return x2 ? rbf(bf2f(x1[c]) + bf2f(x2[c])) : bf2f(x1[c]); };
    float m = -__builtin_inff();
    for (int c = lane; c < n_cls; c += 64) m = fmaxf(m, X(c));
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    float s = 0.f;
    for (int c = lane; c < n_cls; c += 64) s += __expf(X(c) - m);
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    long long li = label_ids ? (long long)label_ids[r] - id_off : (long long)r;
    const bool in_table = n_table <= 0 || (li >= 0 && li < n_table);
    if (!in_table) li = 0;
    const long long y = labels[li];
    const bool ok = in_table && y >= 0 && y < n_cls;
    if (!ok && lane == 0) atomicOr(err, BLISS_ERR_CAP_CAND);            // label out of range (torch raises a device assert)
    const float lse = m + __logf(s);
    if (lane == 0) row_loss[r] = ok ? lse - X((int)y) : 0.f;
    const float inv_s = 1.0f / s;
    bf16_t* g = dlogits + (long long)r * d_stride;
    for (int c = lane; c < n_cls; c += 64) {
      const float p = __expf(X(c) - m) * inv_s;
      g[c] = f2bf((p - ((ok && c == (int)y) ? 1.0f : 0.0f)) * inv_n);
    }
  }
  // the last workgroup sums the per-row losses in row order (deterministic) and writes the mean
  __shared__ float part[CE_TPB / 64];
  __shared__ int last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    last = atomicAdd(ticket, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  float acc = 0.f;
  for (int r = threadIdx.x; r < n_rows; r += CE_TPB) acc += __hip_atomic_load(row_loss + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
  if (lane == 0) part[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < CE_TPB / 64; ++w) t += part[w];
    *loss_out = t * inv_n;
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

}  // namespace

static int ce_launch(const void* logits, int64_t stride, const void* logits2, int64_t stride2, const int64_t* labels, const int32_t* label_ids,
                     int32_t n_rows, int32_t n_cls, float* row_loss, void* dlogits, int64_t d_stride, float* loss_out, uint32_t* ticket,
                     int32_t* err, void* stream, const int32_t* n_rows_dev = nullptr, float denom = 0.f, int32_t id_off = 0,
                     int32_t n_table = 0) {
  if (!logits || !labels || !row_loss || !dlogits || !loss_out || !ticket || !err || n_rows <= 0 || n_cls <= 0) return BLISS_EINVAL;
  int grid = (n_rows + CE_TPB / 64 - 1) / (CE_TPB / 64);
  if (grid > 1024) grid = 1024;
  k_cross_entropy<<<grid, CE_TPB, 0, (hipStream_t)stream>>>((const bf16_t*)logits, stride, (const bf16_t*)logits2, stride2, (const long long*)labels,
                                                            label_ids, n_rows, n_cls, row_loss, (bf16_t*)dlogits, d_stride, loss_out, ticket, err,
                                                            n_rows_dev, denom, id_off, n_table);
  return (int)hipGetLastError();
}

extern "C" int bliss_cross_entropy(const void* logits, int64_t stride, const int64_t* labels, int32_t n_rows, int32_t n_cls,
                                   float* row_loss, void* dlogits, int64_t d_stride, float* loss_out, uint32_t* ticket, int32_t* err,
                                   void* stream) {
  return ce_launch(logits, stride, nullptr, 0, labels, nullptr, n_rows, n_cls, row_loss, dlogits, d_stride, loss_out, ticket, err, stream);
}

extern "C" int bliss_cross_entropy_sum(const void* logits, int64_t stride, const void* logits2, int64_t stride2, const int64_t* label_table,
                                       const int32_t* label_ids, int32_t n_rows, int32_t n_cls, float* row_loss, void* dlogits,
                                       int64_t d_stride, float* loss_out, uint32_t* ticket, int32_t* err, void* stream) {
  if (!logits2 && !label_ids) return BLISS_EINVAL;
  return ce_launch(logits, stride, logits2, stride2, label_table, label_ids, n_rows, n_cls, row_loss, dlogits, d_stride, loss_out, ticket, err,
                   stream);
}

extern "C" int bliss_cross_entropy_masked(const void* logits, int64_t stride, const void* logits2, int64_t stride2, const int64_t* label_table,
                                          int32_t n_table, const int32_t* label_ids, int32_t id_off, int32_t n_rows, const int32_t* n_rows_dev,
                                          float denom, int32_t n_cls, float* row_loss, void* dlogits, int64_t d_stride, float* loss_out,
                                          uint32_t* ticket, int32_t* err, void* stream) {
  if (!label_ids || !n_rows_dev || n_table <= 0 || !(denom > 0.f)) return BLISS_EINVAL;
  return ce_launch(logits, stride, logits2, stride2, label_table, label_ids, n_rows, n_cls, row_loss, dlogits, d_stride, loss_out, ticket, err,
                   stream, n_rows_dev, denom, id_off, n_table);
}
