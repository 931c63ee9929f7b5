// HIP-event bracketing of individual kernel launches, on the stream the kernel is launched on.
#include "prof.h"
#include "bliss_gnn.h"
#include <mutex>
#include <vector>

int g_bliss_prof_sel = -2;

namespace {
struct Pair { hipEvent_t a, b; int id; };
std::mutex g_mu;
std::vector<Pair> g_open, g_free;
double g_ms[BK_COUNT];
long long g_n[BK_COUNT];
const char* kNames[BK_COUNT] = {
  "k_seg_scan", "k_frontier_pass1", "k_frontier_pass2", "k_chunk_scan", "k_frontier_pass3", "k_cand_finalize", "k_poisson_scale",
  "k_select_pass1", "k_select_pass2", "k_block_pass1", "k_indptr_scan", "k_block_pass2", "k_cleanup", "k_mt19937_uniform",
  "k_spmm_fwd", "k_spmm_bwd", "k_embed_norm", "k_exp3_update", "k_exp3_apply", "k_normalize_row", "k_row_sum",
  "k_normalized_edata", "k_block_transpose", "k_spmm_fixup", "k_col_sums", "k_bin_scatter", "k_bin_reduce", "k_bitmap_scan",
  "k_cand_number"};
thread_local Pair t_cur;

void drain_locked() {
  for (auto& p : g_open) {
    (void)hipEventSynchronize(p.b);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { g_ms[p.id] += ms; g_n[p.id] += 1; }
    g_free.push_back(p);
  }
  g_open.clear();
}
}  // namespace

void bliss_prof_begin(int id, hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_open.size() >= 8192) drain_locked();
  Pair p;
  if (!g_free.empty()) { p = g_free.back(); g_free.pop_back(); }
  else { (void)hipEventCreate(&p.a); (void)hipEventCreate(&p.b); }
  p.id = id;
  (void)hipEventRecord(p.a, st);
  t_cur = p;
}

void bliss_prof_end(int id, hipStream_t st) {
  (void)id;
  (void)hipEventRecord(t_cur.b, st);
  std::lock_guard<std::mutex> lk(g_mu);
  g_open.push_back(t_cur);
}

extern "C" {

int bliss_prof_enable(int kernel_id) {
  if (kernel_id < -2 || kernel_id >= BK_COUNT) return BLISS_EINVAL;
  g_bliss_prof_sel = kernel_id;
  return 0;
}

int bliss_prof_reset(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  drain_locked();
  for (int i = 0; i < BK_COUNT; ++i) { g_ms[i] = 0; g_n[i] = 0; }
  return 0;
}

int bliss_prof_read(int kernel_id, double* total_ms, int64_t* launches) {
  if (kernel_id < 0 || kernel_id >= BK_COUNT || !total_ms || !launches) return BLISS_EINVAL;
  std::lock_guard<std::mutex> lk(g_mu);
  drain_locked();
  *total_ms = g_ms[kernel_id];
  *launches = g_n[kernel_id];
  return 0;
}

int bliss_prof_kernel_count(void) { return BK_COUNT; }

const char* bliss_prof_kernel_name(int kernel_id) { return (kernel_id >= 0 && kernel_id < BK_COUNT) ? kNames[kernel_id] : ""; }

}  // extern "C"
