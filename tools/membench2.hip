// membench2 — obs-gather structure study at 256-B and 1-KiB rows (measurement tool only).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// A: one float4 per thread, one-shot grid.  row = gid >> qshift.
template <bool NT, int BS>
__global__ __launch_bounds__(BS) void g_thread(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst,
                                               const uint32_t *__restrict__ idx, size_t n4, unsigned qshift, unsigned qmask) {
  const size_t g = (size_t)blockIdx.x * BS + threadIdx.x;
  if (g >= n4) return;
  const uint32_t p = idx[g >> qshift];
  const f32x4 *s = src + (((size_t)p << qshift) + (g & qmask));
  f32x4 v = NT ? __builtin_nontemporal_load(s) : *s;
  if (NT) __builtin_nontemporal_store(v, dst + g); else dst[g] = v;
}

// B: R float4 per thread (strided by the block), all loads first; one-shot grid.
template <bool NT, int BS, int R>
__global__ __launch_bounds__(BS) void g_multi(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst,
                                              const uint32_t *__restrict__ idx, size_t n4, unsigned qshift, unsigned qmask) {
  const size_t g0 = (size_t)blockIdx.x * BS * R + threadIdx.x;
  uint32_t p[R]; f32x4 v[R];
#pragma unroll
  for (int r = 0; r < R; ++r) { size_t g = g0 + (size_t)r * BS; p[r] = g < n4 ? idx[g >> qshift] : 0; }
#pragma unroll
  for (int r = 0; r < R; ++r) { size_t g = g0 + (size_t)r * BS; if (g < n4) { const f32x4 *s = src + (((size_t)p[r] << qshift) + (g & qmask)); v[r] = NT ? __builtin_nontemporal_load(s) : *s; } }
#pragma unroll
  for (int r = 0; r < R; ++r) { size_t g = g0 + (size_t)r * BS; if (g < n4) { if (NT) __builtin_nontemporal_store(v[r], dst + g); else dst[g] = v[r]; } }
}

// C: nt loads only / nt stores only mixes on the thread kernel
template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void g_mix(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst,
                                             const uint32_t *__restrict__ idx, size_t n4, unsigned qshift, unsigned qmask) {
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= n4) return;
  const uint32_t p = idx[g >> qshift];
  const f32x4 *s = src + (((size_t)p << qshift) + (g & qmask));
  f32x4 v = NTL ? __builtin_nontemporal_load(s) : *s;
  if (NTS) __builtin_nontemporal_store(v, dst + g); else dst[g] = v;
}

template <typename F>
static void timeit(const char *name, double bytes, F launch) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  std::vector<float> ms;
  for (int i = 0; i < 8; ++i) { CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float t; CK(hipEventElapsedTime(&t, a, b)); if (i) ms.push_back(t); }
  std::sort(ms.begin(), ms.end());
  printf("%-46s median %8.1f us  %7.0f GB/s   best %7.0f GB/s\n", name, ms[ms.size() / 2] * 1e3, bytes / (ms[ms.size() / 2] * 1e-3) / 1e9, bytes / (ms[0] * 1e-3) / 1e9);
  fflush(stdout);
}

int main() {
  const size_t n4 = (size_t)2147483648ull / 16;
  f32x4 *src, *dst; CK(hipMalloc(&src, n4 * 16)); CK(hipMalloc(&dst, n4 * 16));
  CK(hipMemset(src, 1, n4 * 16)); CK(hipMemset(dst, 0, n4 * 16));
  for (unsigned rowbytes : {256u, 1024u}) {
    const unsigned Q = rowbytes / 16; unsigned qs = 0; while ((1u << qs) < Q) ++qs;
    const size_t rows = n4 / Q;
    std::vector<uint32_t> h(rows);
    for (size_t i = 0; i < rows; ++i) h[i] = (uint32_t)i;
    std::mt19937_64 rng(7);
    for (size_t i = rows; i-- > 1;) std::swap(h[i], h[rng() % (i + 1)]);
    uint32_t *idx; CK(hipMalloc(&idx, rows * 4)); CK(hipMemcpy(idx, h.data(), rows * 4, hipMemcpyHostToDevice));
    const double by = 2.0 * rows * rowbytes + rows * 4.0;
    char nm[128];
#define RUN(label, kern, bs, per)  snprintf(nm, sizeof nm, "%4u-B rows %s", rowbytes, label); \
    timeit(nm, by, [&] { hipLaunchKernelGGL(kern, dim3((unsigned)((n4 + (size_t)(bs) * (per) - 1) / ((size_t)(bs) * (per)))), dim3(bs), 0, 0, src, dst, idx, n4, qs, Q - 1); });
    RUN("thread nt  bs256", (g_thread<true, 256>), 256, 1)
    RUN("thread nt  bs64", (g_thread<true, 64>), 64, 1)
    RUN("thread nt  bs512", (g_thread<true, 512>), 512, 1)
    RUN("thread nt  bs1024", (g_thread<true, 1024>), 1024, 1)
    RUN("thread plain bs256", (g_thread<false, 256>), 256, 1)
    RUN("mix ntLoad only", (g_mix<true, false>), 256, 1)
    RUN("mix ntStore only", (g_mix<false, true>), 256, 1)
    RUN("multi nt bs256 R2", (g_multi<true, 256, 2>), 256, 2)
    RUN("multi nt bs256 R4", (g_multi<true, 256, 4>), 256, 4)
    RUN("multi nt bs64 R4", (g_multi<true, 64, 4>), 64, 4)
    RUN("multi nt bs256 R8", (g_multi<true, 256, 8>), 256, 8)
    CK(hipFree(idx));
  }
  return 0;
}
