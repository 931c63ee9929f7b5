// Internal: MT19937 jump-ahead polynomials (mt_jump.hip, host code) used by the parallel generator in rng.hip.
#pragma once
#include <cstdint>

#define MT_JUMP_WORDS 624          // uint32 words per polynomial (19937 coefficients, padded)

bool mt_jump_ready();
// out[s * MT_JUMP_WORDS ..] = coefficients of t^(first + s * stride) mod phi, s = 0 .. count-1
bool mt_jump_polys(int64_t first, int64_t stride, int count, uint32_t* out);
