// Host-side permutation of the minibatch iterator: `indices.shuffle(&mut rng)`
// (ppo/src/model/data.rs:93-95).
//
// The reference draws from rand 0.8.5's OS-seeded ThreadRng, which nobody can reproduce.  This
// is the same algorithm made seedable, so that a Rust caller holding
// `ChaCha12Rng::seed_from_u64(seed)` (rand_chacha 0.3.1, the generator behind ThreadRng) and
// calling `indices.shuffle(&mut rng)` gets the same permutation:
//   * key   = rand_core 0.6 `seed_from_u64`: eight PCG32 (XSH-RR) outputs, little endian;
//   * stream = ChaCha with 12 rounds, 64-bit block counter from 0, stream id 0, words consumed
//     in order (rand_chacha refills four blocks at a time, which does not change the order);
//   * shuffle = `for i in (1..len).rev() { swap(i, gen_index(rng, i + 1)) }` with
//     gen_index = UniformInt<u32>::sample_single: widening multiply, reject while the low half
//     exceeds `(range << range.leading_zeros()) - 1`.
// len must be < 2^32 (rand switches to 64-bit sampling above that; the device indices are u32).
#pragma once
#include <cstddef>
#include <cstdint>

namespace ppoxp {

class ChaCha12Stream {
 public:
  explicit ChaCha12Stream(uint64_t seed) {
    uint64_t s = seed;
    for (uint32_t &k : key_) {
      s = s * 6364136223846793005ULL + 11634580027462260723ULL;
      const uint32_t x = (uint32_t)(((s >> 18) ^ s) >> 27);
      const unsigned r = (unsigned)(s >> 59);
      k = rotr(x, r);
    }
  }

  uint32_t next_u32() {
    if (pos_ == kWords) refill();
    return buf_[pos_++];
  }

  // UniformInt<u32>::sample_single(0, range), range >= 1
  uint32_t below(uint32_t range) {
    const uint32_t zone = (range << __builtin_clz(range)) - 1u;
    for (;;) {
      const uint64_t wide = (uint64_t)next_u32() * range;
      if ((uint32_t)wide <= zone) return (uint32_t)(wide >> 32);
    }
  }

 private:
  static constexpr int kBlocks = 4, kWords = 16 * kBlocks;
  static uint32_t rotl(uint32_t v, unsigned n) { return (v << n) | (v >> (32 - n)); }
  static uint32_t rotr(uint32_t v, unsigned n) { n &= 31; return n ? (v >> n) | (v << (32 - n)) : v; }
  static void quarter(uint32_t *w, int a, int b, int c, int d) {
    w[a] += w[b]; w[d] = rotl(w[d] ^ w[a], 16);
    w[c] += w[d]; w[b] = rotl(w[b] ^ w[c], 12);
    w[a] += w[b]; w[d] = rotl(w[d] ^ w[a], 8);
    w[c] += w[d]; w[b] = rotl(w[b] ^ w[c], 7);
  }
  void refill() {
    for (int blk = 0; blk < kBlocks; ++blk, ++counter_) {
      uint32_t in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key_[0], key_[1], key_[2], key_[3],
                         key_[4], key_[5], key_[6], key_[7], (uint32_t)counter_, (uint32_t)(counter_ >> 32), 0u, 0u};
      uint32_t w[16];
      for (int i = 0; i < 16; ++i) w[i] = in[i];
      for (int dr = 0; dr < 6; ++dr) {  // 6 double rounds = ChaCha12
        quarter(w, 0, 4, 8, 12); quarter(w, 1, 5, 9, 13); quarter(w, 2, 6, 10, 14); quarter(w, 3, 7, 11, 15);
        quarter(w, 0, 5, 10, 15); quarter(w, 1, 6, 11, 12); quarter(w, 2, 7, 8, 13); quarter(w, 3, 4, 9, 14);
      }
      for (int i = 0; i < 16; ++i) buf_[16 * blk + i] = w[i] + in[i];
    }
    pos_ = 0;
  }
  uint32_t key_[8];
  uint64_t counter_ = 0;
  uint32_t buf_[kWords];
  int pos_ = kWords;
};

inline void shuffle_indices(uint64_t seed, uint32_t *idx, size_t len) {
  for (size_t i = 0; i < len; ++i) idx[i] = (uint32_t)i;
  ChaCha12Stream rng(seed);
  for (size_t i = len; i-- > 1;) {
    const uint32_t j = rng.below((uint32_t)(i + 1));
    const uint32_t t = idx[i]; idx[i] = idx[j]; idx[j] = t;
  }
}

}  // namespace ppoxp
