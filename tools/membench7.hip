// membench7 — can the SCALAR memory path fetch a random 16-byte record for less than a 128-byte line?
// Vector loads always cost a 128-B fabric request (membench3, membench6).  Kernel-argument loads (s_load) show up in the
// counters as 64-B requests, so this fetches the records with uniform (per-wave) scalar loads: each wave walks its 64 rows,
// takes row r's index with readlane, loads the 16-byte record into SGPRs and hands it to lane r.  LOADS ONLY: nothing is
// ever stored through the scalar cache.  Run plain for times and under
//   rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void gather_vector(const u32x4 *__restrict__ tab, u32x4 *__restrict__ out,
                                                     const uint32_t *__restrict__ idx, size_t n) {
  const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  out[j] = tab[idx[j]];
}

// U records in flight per wave
template <int U>
__global__ __launch_bounds__(256) void gather_scalar(const u32x4 *__restrict__ tab, u32x4 *__restrict__ out,
                                                     const uint32_t *__restrict__ idx, size_t n) {
  const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;      // n is a multiple of 256 here
  const uint32_t p = idx[j];
  const unsigned lane = threadIdx.x & 63u;
  u32x4 mine = {0u, 0u, 0u, 0u};
#pragma unroll 1
  for (int r0 = 0; r0 < 64; r0 += U) {
    u32x4 q[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t pr = __builtin_amdgcn_readlane(p, r0 + u);             // uniform
      const u32x4 *src = tab + pr;
      asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=s"(q[u]) : "s"(src));   // scalar LOAD of the 16-byte record
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < U; ++u) {
      // tie the SGPR values to the wait above, then give them to lane r0+u
      asm volatile("" : "+s"(q[u]));
      if (lane == (unsigned)(r0 + u)) mine = q[u];
    }
  }
  out[j] = mine;
}

template <typename F>
static float timeit(const char *name, F launch) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  std::vector<float> ms;
  for (int i = 0; i < 6; ++i) { CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float t; CK(hipEventElapsedTime(&t, a, b)); if (i) ms.push_back(t); }
  std::sort(ms.begin(), ms.end());
  printf("%-36s median %8.1f us\n", name, ms[ms.size() / 2] * 1e3); fflush(stdout);
  return ms[ms.size() / 2];
}

int main() {
  const size_t n = 8388608;
  u32x4 *tab, *out, *ref; uint32_t *idx;
  CK(hipMalloc(&tab, n * 16)); CK(hipMalloc(&out, n * 16)); CK(hipMalloc(&ref, n * 16)); CK(hipMalloc(&idx, n * 4));
  std::vector<uint32_t> h(n), t4(n * 4);
  for (size_t i = 0; i < n; ++i) { h[i] = (uint32_t)i; t4[4 * i] = (uint32_t)i; t4[4 * i + 1] = ~(uint32_t)i; t4[4 * i + 2] = 7; t4[4 * i + 3] = (uint32_t)(i * 2654435761u); }
  CK(hipMemcpy(tab, t4.data(), n * 16, hipMemcpyHostToDevice));
  std::mt19937_64 rng(7);
  for (size_t i = n; i-- > 1;) std::swap(h[i], h[rng() % (i + 1)]);
  CK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice));
  const unsigned blocks = (unsigned)(n / 256);
  timeit("vector gather (global_load_dwordx4)", [&] { hipLaunchKernelGGL(gather_vector, dim3(blocks), dim3(256), 0, 0, tab, ref, idx, n); });
  timeit("scalar gather, 4 in flight per wave", [&] { hipLaunchKernelGGL((gather_scalar<4>), dim3(blocks), dim3(256), 0, 0, tab, out, idx, n); });
  timeit("scalar gather, 8 in flight per wave", [&] { hipLaunchKernelGGL((gather_scalar<8>), dim3(blocks), dim3(256), 0, 0, tab, out, idx, n); });
  timeit("scalar gather, 16 in flight per wave", [&] { hipLaunchKernelGGL((gather_scalar<16>), dim3(blocks), dim3(256), 0, 0, tab, out, idx, n); });
  std::vector<uint32_t> a(n * 4), b(n * 4);
  CK(hipMemcpy(a.data(), out, n * 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), ref, n * 16, hipMemcpyDeviceToHost));
  printf("scalar gather == vector gather: %s\n", a == b ? "yes" : "NO");
  return 0;
}
