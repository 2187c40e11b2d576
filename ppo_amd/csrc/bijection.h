// Keyed bijection on [0, M): the permutation of a shuffled pass evaluated IN REGISTERS, no index array.
//
// The reference draws `indices.shuffle(&mut thread_rng())` (ppo/src/model/data.rs:91-96): a fresh uniform
// permutation per pass from an OS-seeded generator nobody can reproduce.  What a pass needs from it is
// perm(j) for the output positions j it gathers, and (for the source-side statistics) perm^-1(i).  Both are
// computed here from (seed, M) alone:
//
//   domain   2^n with n = max(8, ceil(log2 M)); x = (a << l) | b with l = n / 2 low bits, h = n - l high bits
//   F        six alternating Feistel steps  r even: a ^= f(b, key_r) & (2^h - 1)
//                                           r odd : b ^= f(a, key_r) & (2^l - 1)
//            every step is an involution, so F^-1 is the same steps in reverse order
//   f(v, k)  t = v ^ (k & 0xFFFF); t = mul24(t, 0xE995); t ^= t >> 15; t = mul24(t, 0x2C1B3); t ^= t >> 13; t += k >> 16
//            mul24 = low 32 bits of the product of the operands' low 24 bits (v_mul_u32_u24, full rate on gfx950)
//   key_r    high 32 bits of the r-th output of SplitMix64 started at `seed`
//   perm(j)  cycle walking: x = j; do x = F(x) while x >= M   (2^n < 2M for M >= 128: under two steps on average)
//
// The minimum domain of 256 keeps the halves at >= 4 bits: below that the family of six Feistel steps is visibly
// non-uniform (chi-square over 200 000 seeds), with it every M from 2 to 1000 tested at chi2/dof ~ 1.
// CPU twins (C and numpy) live with the oracle; indices are compared bit for bit in tests/.  A uniform-looking keyed permutation like the reference's shuffle, not rand's
// sequence; the host Fisher-Yates of ppo_xp_pass_begin stays the reference-faithful mode.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#include <hip/hip_runtime.h>
#define PPOXP_HD __host__ __device__ __forceinline__
#else
#define PPOXP_HD inline
#endif

namespace ppoxp {

constexpr int kBijRounds = 6;

struct Bijection {
  uint32_t M;              // domain [0, M)
  uint32_t l, hm, lm;      // low-half width, masks of the high / low half
  uint32_t key[kBijRounds];
};

inline uint64_t bij_splitmix64_at(uint64_t seed, uint64_t i) {
  uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// M in [1, 2^32).
inline Bijection bij_make(uint64_t seed, uint64_t M) {
  Bijection b{};
  unsigned n = 8;
  while (n < 32 && (uint64_t(1) << n) < M) ++n;
  b.M = (uint32_t)M;
  b.l = n / 2;
  const unsigned h = n - b.l;
  b.hm = (uint32_t)((uint64_t(1) << h) - 1);
  b.lm = (uint32_t)((uint64_t(1) << b.l) - 1);
  for (int r = 0; r < kBijRounds; ++r) b.key[r] = (uint32_t)(bij_splitmix64_at(seed, (uint64_t)r) >> 32);
  return b;
}

PPOXP_HD uint32_t bij_mul24(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul24(a, b);
#else
  return (uint32_t)((uint64_t)(a & 0xFFFFFFu) * (uint64_t)(b & 0xFFFFFFu));
#endif
}

PPOXP_HD uint32_t bij_f(uint32_t v, uint32_t k) {
  uint32_t t = v ^ (k & 0xFFFFu);
  t = bij_mul24(t, 0xE995u);
  t ^= t >> 15;
  t = bij_mul24(t, 0x2C1B3u);
  t ^= t >> 13;
  return t + (k >> 16);
}

PPOXP_HD uint32_t bij_F(const Bijection &b, uint32_t x) {
  uint32_t a = x >> b.l, c = x & b.lm;
#pragma unroll
  for (int r = 0; r < kBijRounds; ++r) {
    if (!(r & 1)) a ^= bij_f(c, b.key[r]) & b.hm;
    else c ^= bij_f(a, b.key[r]) & b.lm;
  }
  return (a << b.l) | c;
}

PPOXP_HD uint32_t bij_Finv(const Bijection &b, uint32_t x) {
  uint32_t a = x >> b.l, c = x & b.lm;
#pragma unroll
  for (int r = kBijRounds - 1; r >= 0; --r) {
    if (!(r & 1)) a ^= bij_f(c, b.key[r]) & b.hm;
    else c ^= bij_f(a, b.key[r]) & b.lm;
  }
  return (a << b.l) | c;
}

// perm(j), j < M.  The walk stays inside the cycle of j under F, which contains j < M: it terminates.
PPOXP_HD uint32_t bij_perm(const Bijection &b, uint32_t j) {
  uint32_t x = j;
  do x = bij_F(b, x); while (x >= b.M);
  return x;
}

// perm^-1(i), i < M.
PPOXP_HD uint32_t bij_perm_inv(const Bijection &b, uint32_t i) {
  uint32_t x = i;
  do x = bij_Finv(b, x); while (x >= b.M);
  return x;
}

}  // namespace ppoxp
