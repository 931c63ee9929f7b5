// MT19937 jump-ahead polynomials (host code): what lets several workgroups generate disjoint stretches of torch's CPU
// random stream at once (rng.hip) instead of one wave walking the 624-word recurrence for the whole call.
//
// The generator is linear over GF(2): with T = "advance the 19937-bit state by ONE word", every bit of every state word
// x_k (k >= 1; the low 31 bits of x_0 are not part of the state) is a linear functional of the state, so the word
// sequence obeys the recurrence of T's characteristic polynomial phi (degree 19937):
//     x_{k+n} = XOR over { i : c_i = 1 } of x_{k+i},      sum_i c_i t^i = t^n mod phi(t),     for every k >= 1.
// The state after a jump of n words is x_n .. x_{n+623}; taking k = j + 1 and the polynomial of n - 1 gives all 624 words
// (word 0 included, bit for bit) from x_1 .. x_{19936+624}: one binary convolution per jump, done on the device.
//
// phi is computed once per process with Berlekamp-Massey from 2 * 19937 bits of one bit-plane of the sequence (any
// non-zero state gives the same minimal polynomial; checked: degree 19937), the powers by stepping p <- t * p mod phi.
// No reference counterpart: the reference draws on the host (bandit_sampler.py:422-424) and ATen walks the recurrence
// serially (ATen/core/MT19937RNGEngine.h).
#include <cstdint>
#include <cstring>
#include <mutex>
#include <vector>

#include "bliss_gnn.h"
#include "mt_jump.h"

namespace {

constexpr int DEG = 19937;
constexpr int PW = (DEG + 1 + 63) / 64;                  // 64-bit words of a polynomial of degree <= 19937

std::once_flag g_phi_once;
uint64_t g_phi[PW];                                     // bit i = coefficient of t^i; bit 19937 set
bool g_phi_ok = false;

// at::mt19937 / std::mt19937 seeding and one-word stepping, host side (only to feed Berlekamp-Massey)
void lsb_sequence(std::vector<uint8_t>& bits, int n) {
  constexpr int N = 624, M = 397;
  std::vector<uint32_t> x(N + n + 1);
  x[0] = 5489u;
  for (int i = 1; i < N; ++i) x[i] = 1812433253u * (x[i - 1] ^ (x[i - 1] >> 30)) + (uint32_t)i;
  for (int k = 0; k < n + 1; ++k) {
    const uint32_t y = (x[k] & 0x80000000u) | (x[k + 1] & 0x7fffffffu);
    x[k + N] = x[k + M] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
  }
  bits.resize(n);
  for (int k = 0; k < n; ++k) bits[k] = (uint8_t)(x[k + 1] & 1u);          // bit 0 of x_1, x_2, ...
}

inline int parity_and(const uint64_t* a, const uint64_t* b, int words) {
  uint64_t acc = 0;
  for (int i = 0; i < words; ++i) acc ^= a[i] & b[i];
  return __builtin_parityll(acc);
}

// c ^= b << m   (polynomials of W words)
void xor_shifted(uint64_t* c, const uint64_t* b, int m, int W) {
  const int ws = m >> 6, bs = m & 63;
  for (int i = W - 1; i >= ws; --i) {
    uint64_t v = b[i - ws] << bs;
    if (bs && i - ws - 1 >= 0) v |= b[i - ws - 1] >> (64 - bs);
    c[i] ^= v;
  }
}

void compute_phi() {
  const int n = 2 * DEG + 64;
  std::vector<uint8_t> s;
  lsb_sequence(s, n);
  const int W = (n + 64) / 64 + 1;
  std::vector<uint64_t> C(W, 0), B(W, 0), T(W, 0), win(W, 0), Csh(W, 0);
  C[0] = 1; B[0] = 1;
  int L = 0, m = 1;
  for (int i = 0; i < n; ++i) {
    // discrepancy d = s[i] + sum_{j=1..L} C_j s[i-j];  win bit (j-1) = s[i-j]
    for (int w = 0; w < W - 1; ++w) Csh[w] = (C[w] >> 1) | (C[w + 1] << 63);
    Csh[W - 1] = C[W - 1] >> 1;
    const int used = (L + 63) / 64 + 1 < W ? (L + 63) / 64 + 1 : W;
    const int d = s[i] ^ parity_and(Csh.data(), win.data(), used);
    if (d) {
      T = C;
      xor_shifted(C.data(), B.data(), m, W);
      if (2 * L <= i) { L = i + 1 - L; B = T; m = 1; } else ++m;
    } else ++m;
    for (int w = W - 1; w > 0; --w) win[w] = (win[w] << 1) | (win[w - 1] >> 63);
    win[0] = (win[0] << 1) | s[i];
  }
  if (L != DEG) return;
  // connection polynomial C (s_i = sum_j C_j s_{i-j})  ->  characteristic polynomial phi(t) = t^L C(1/t)
  std::memset(g_phi, 0, sizeof(g_phi));
  for (int j = 0; j <= L; ++j)
    if ((C[j >> 6] >> (j & 63)) & 1ull) { const int k = L - j; g_phi[k >> 6] |= 1ull << (k & 63); }
  g_phi_ok = ((g_phi[DEG >> 6] >> (DEG & 63)) & 1ull) != 0;
}

// p <- t * p mod phi
inline void step(uint64_t* p) {
  uint64_t carry = 0;
  for (int w = 0; w < PW; ++w) { const uint64_t nc = p[w] >> 63; p[w] = (p[w] << 1) | carry; carry = nc; }
  if ((p[DEG >> 6] >> (DEG & 63)) & 1ull)
    for (int w = 0; w < PW; ++w) p[w] ^= g_phi[w];
}

}  // namespace

bool mt_jump_ready() {
  std::call_once(g_phi_once, compute_phi);
  return g_phi_ok;
}

// polys[s] (s = 0 .. count-1) = t^(first + s * stride) mod phi as 624 uint32 words (bit i of the 19937 = coefficient of t^i)
bool mt_jump_polys(int64_t first, int64_t stride, int count, uint32_t* out) {
  if (!mt_jump_ready() || first < 0 || stride < 0 || count < 0) return false;
  uint64_t p[PW];
  std::memset(p, 0, sizeof(p));
  p[0] = 1;
  int64_t at = 0;
  for (int s = 0; s < count; ++s) {
    const int64_t want = first + (int64_t)s * stride;
    for (; at < want; ++at) step(p);
    uint32_t* o = out + (size_t)s * MT_JUMP_WORDS;
    for (int w = 0; w < MT_JUMP_WORDS; ++w) {
      const int q = w >> 1;
      o[w] = q < PW ? (uint32_t)(p[q] >> ((w & 1) * 32)) : 0u;
    }
  }
  return true;
}

extern "C" int bliss_mt_jump_poly(int64_t n_words, uint32_t* poly_words) {
  if (n_words < 0 || !poly_words) return BLISS_EINVAL;
  return mt_jump_polys(n_words, 0, 1, poly_words) ? 0 : BLISS_EINVAL;
}
