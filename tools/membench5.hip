// membench5 — 256-byte rows: random-read gather (out[j] = src[idx[j]]) vs random-write scatter
// (out[idx[i]] = src[i]); thread = one float4, 64-thread workgroups, non-temporal accesses.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool SCATTER, bool NTS>
__global__ __launch_bounds__(64) void rows(const f32x4 *__restrict__ src, f32x4 *__restrict__ out,
                                           const uint32_t *__restrict__ idx, size_t total) {
  const size_t e = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (e >= total) return;
  const size_t row = e >> 4; const unsigned col = e & 15;
  const uint32_t p = idx[row];
  if (SCATTER) {
    const f32x4 q = __builtin_nontemporal_load(src + e);
    if (NTS) __builtin_nontemporal_store(q, out + ((size_t)p << 4) + col); else out[((size_t)p << 4) + col] = q;
  } else {
    const f32x4 q = __builtin_nontemporal_load(src + ((size_t)p << 4) + col);
    if (NTS) __builtin_nontemporal_store(q, out + e); else out[e] = q;
  }
}

template <typename F>
static void timeit(const char *name, F launch, double gb) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  std::vector<float> ms;
  for (int i = 0; i < 8; ++i) { CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float t; CK(hipEventElapsedTime(&t, a, b)); if (i) ms.push_back(t); }
  std::sort(ms.begin(), ms.end());
  printf("%-30s median %8.1f us  %6.0f GB/s\n", name, ms[ms.size() / 2] * 1e3, gb / (ms[ms.size() / 2] * 1e-3)); fflush(stdout);
}

int main() {
  const size_t n = 8388608, total = n * 16;
  f32x4 *src, *out; uint32_t *idx;
  CK(hipMalloc(&src, n * 256)); CK(hipMalloc(&out, n * 256)); CK(hipMalloc(&idx, n * 4));
  CK(hipMemset(src, 1, n * 256));
  std::vector<uint32_t> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (uint32_t)i;
  std::mt19937_64 rng(7);
  for (size_t i = n; i-- > 1;) std::swap(h[i], h[rng() % (i + 1)]);
  CK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice));
  const unsigned blocks = (unsigned)(total / 64);
  const double gb = 2.0 * n * 256 / 1e9;
  timeit("gather rows, nt store", [&] { hipLaunchKernelGGL((rows<false, true>), dim3(blocks), dim3(64), 0, 0, src, out, idx, total); }, gb);
  timeit("scatter rows, nt store", [&] { hipLaunchKernelGGL((rows<true, true>), dim3(blocks), dim3(64), 0, 0, src, out, idx, total); }, gb);
  timeit("scatter rows, plain store", [&] { hipLaunchKernelGGL((rows<true, false>), dim3(blocks), dim3(64), 0, 0, src, out, idx, total); }, gb);
  return 0;
}
