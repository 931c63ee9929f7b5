// Internal: per-kernel HIP-event timing used by bench.py's `roofline` object (bliss_prof_* in the ABI).
// Disabled (one predictable branch per launch) unless bliss_prof_enable() selects a kernel id.
#pragma once
#include <hip/hip_runtime.h>

enum BlissKernelId {
  BK_SEG_SCAN = 0, BK_PASS1, BK_PASS2, BK_CHUNK_SCAN, BK_PASS3, BK_CAND_FINALIZE, BK_POISSON_SCALE, BK_SELECT1, BK_SELECT2,
  BK_BLOCK1, BK_INDPTR_SCAN, BK_BLOCK2, BK_CLEANUP, BK_MT19937, BK_SPMM_FWD, BK_SPMM_BWD, BK_EMBED_NORM, BK_EXP3_UPDATE,
  BK_EXP3_APPLY, BK_NORMALIZE, BK_ROW_SUM, BK_NORM_EDATA, BK_TRANSPOSE, BK_SPMM_FIXUP, BK_COL_SUMS, BK_BIN_SCATTER, BK_BIN_REDUCE,
  BK_BITMAP_SCAN, BK_CAND_NUMBER, BK_COUNT
};

extern int g_bliss_prof_sel;                       // -2 off, -1 all, >= 0 one kernel id
void bliss_prof_begin(int id, hipStream_t st);
void bliss_prof_end(int id, hipStream_t st);

#define PROF_LAUNCH(id, st, ...)                                              \
  do {                                                                        \
    const bool p_ = (g_bliss_prof_sel == -1 || g_bliss_prof_sel == (id));    \
    if (p_) bliss_prof_begin((id), (st));                                     \
    __VA_ARGS__;                                                              \
    if (p_) bliss_prof_end((id), (st));                                       \
  } while (0)
