// membench8 — what would a SOURCE-ORDER statistics pre-pass cost?  (second half.)
// With an invertible keyed permutation the minibatch of source transition i is k = perm^-1(i) / B, so the per-minibatch
// (sum, sum of squares) of the advantages could be taken by streaming over the scan halves of the records in order
// and the gather could then write standardised advantages directly (no reduce_partials / normalize launches).
// This measures that pre-pass alone at TGT size: 8 B read per transition from the 128-byte record lines, perm^-1 in
// registers (six Feistel steps), fp64 (sum, sumsq) binned per lane in LDS (K x 64 lanes x 16 B), fixed-order folds.
// Compare with what it would replace: reduce_partials_kernel + normalize_kernel + one dependency gap.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../ppo_amd/csrc/bijection.h"
using namespace ppoxp;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// one wave per strip of `per_wave` consecutive transitions; lane l takes i = base + s*64 + l
template <int K, int ROUNDS_ON>
__global__ __launch_bounds__(64) void prepass(const u32x2 *__restrict__ rec, Bijection b, unsigned shiftB, size_t per_wave,
                                              double2 *__restrict__ partials) {
  __shared__ double2 bins[K][64];
  const unsigned lane = threadIdx.x;
  for (int k = 0; k < K; ++k) bins[k][lane] = make_double2(0.0, 0.0);
  const size_t base = (size_t)blockIdx.x * per_wave;
  for (size_t s = 0; s < per_wave; s += 64 * 8) {
    u32x2 q[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const size_t i = base + s + u * 64 + lane;
      q[u] = rec[((i >> 3) << 4) + (i & 7) + 8];          // scan half {v, ret}
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const size_t i = base + s + u * 64 + lane;
      const float a = __uint_as_float(q[u].y) - __uint_as_float(q[u].x);
      const unsigned j = ROUNDS_ON ? bij_perm_inv(b, (uint32_t)i) : (uint32_t)((i * 2654435761u) & (b.M - 1));
      const unsigned k = j >> shiftB;
      double2 t = bins[k][lane];
      t.x += (double)a; t.y += (double)a * (double)a;
      bins[k][lane] = t;
    }
  }
  for (int k = 0; k < K; ++k) {
    double2 t = bins[k][lane];
    for (int off = 32; off > 0; off >>= 1) { t.x += __shfl_down(t.x, off, 64); t.y += __shfl_down(t.y, off, 64); }
    if (lane == 0) partials[(size_t)blockIdx.x * K + k] = t;
  }
}

int main() {
  const size_t M = size_t(1) << 23, B = size_t(1) << 18;
  const int K = 32;
  u32x2 *rec; double2 *partials;
  CK(hipMalloc(&rec, M * 16)); CK(hipMemset(rec, 0x3c, M * 16));
  const size_t waves[] = {1024, 2048, 4096};
  CK(hipMalloc(&partials, 4096 * K * sizeof(double2)));
  Bijection b = bij_make(1234, M);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (size_t W : waves) for (int rounds = 1; rounds >= 0; --rounds) {
    std::vector<float> ms;
    for (int it = 0; it < 8; ++it) {
      CK(hipEventRecord(e0));
      if (rounds) hipLaunchKernelGGL((prepass<32, 1>), dim3((unsigned)W), dim3(64), 0, 0, rec, b, 18u, M / W, partials);
      else hipLaunchKernelGGL((prepass<32, 0>), dim3((unsigned)W), dim3(64), 0, 0, rec, b, 18u, M / W, partials);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float t; CK(hipEventElapsedTime(&t, e0, e1)); if (it > 1) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    printf("prepass M=2^23 K=32: %4zu waves, %s: median %.1f us (min %.1f)\n", W, rounds ? "perm^-1 six steps" : "cheap hash instead of perm^-1 (memory + LDS only)",
           ms[ms.size() / 2] * 1e3, ms[0] * 1e3);
  }
  CK(hipGetLastError());
  printf("replaces: reduce_partials_kernel 6.1 us + normalize_kernel 17.0 us + one dependency gap ~4 us (profiles/r3/ab_bijection_TGT.log); a fold launch of its own (~5 us + gap) would still follow it\n");
  return 0;
}
