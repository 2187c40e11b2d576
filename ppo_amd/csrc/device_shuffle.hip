// Device-side permutation generator: removes the host Fisher-Yates
// (70 ms at M = 8.4 M, seconds at M = 134 M) and its H2D copy from the wall clock of a pass.
//
// The reference's contract (ppo/src/model/data.rs:91-96) is "a fresh uniform permutation of 0..M
// per pass"; the specific permutation of its OS-seeded ThreadRng cannot be reproduced by anyone.
// Here:  key_i = splitmix64(seed, i)  (the i-th output of the SplitMix64 stream started at `seed`),
//        perm  = stable argsort(key)   (ties, probability ~M^2/2^65, keep index order).
// The CPU twin (oracle: orc_keyed_shuffle, numpy: argsort(kind="stable")) gives the same indices
// bit for bit; tests/test_gpu_parity.py checks that.  The host Fisher-Yates of ppo_xp_pass_begin
// stays the reference-faithful mode.
//
// Sorting is rocPRIM's LSD radix sort (stable) on (u64 key, u32 index) pairs: this step is outside
// the measured path, so a library sort is used rather than a hand-written one.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include "device_shuffle.h"

namespace ppoxp {

__host__ __device__ inline uint64_t splitmix64_at(uint64_t seed, uint64_t i) {
  uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void shuffle_keys_kernel(uint64_t seed, size_t M, uint64_t *__restrict__ keys,
                                                           uint32_t *__restrict__ idx) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < M) { keys[i] = splitmix64_at(seed, i); idx[i] = (uint32_t)i; }
}

size_t device_shuffle_workspace_bytes(size_t M) {
  size_t tmp = 0;
  (void)rocprim::radix_sort_pairs(nullptr, tmp, (const uint64_t *)nullptr, (uint64_t *)nullptr, (const uint32_t *)nullptr,
                                  (uint32_t *)nullptr, M, 0, 64, (hipStream_t) nullptr);
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  // [keys_in | keys_out | idx_in | rocprim temp]
  return up(M * 8) + up(M * 8) + up(M * 4) + up(tmp);
}

int device_shuffle(uint64_t seed, size_t M, uint32_t *perm_out, void *workspace, size_t workspace_bytes,
                   hipStream_t stream) {
  if (M == 0) return 0;
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  char *w = (char *)workspace;
  uint64_t *keys_in = (uint64_t *)w; w += up(M * 8);
  uint64_t *keys_out = (uint64_t *)w; w += up(M * 8);
  uint32_t *idx_in = (uint32_t *)w; w += up(M * 4);
  size_t tmp = workspace_bytes - (size_t)(w - (char *)workspace);
  hipLaunchKernelGGL(shuffle_keys_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, stream, seed, M, keys_in, idx_in);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  e = rocprim::radix_sort_pairs((void *)w, tmp, (const uint64_t *)keys_in, keys_out, (const uint32_t *)idx_in, perm_out, M, 0,
                                64, stream);
  return (int)e;
}

}  // namespace ppoxp
