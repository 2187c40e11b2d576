// membench3 — does any load flavour make the L2 fetch less than a 128-B line for a random 16-B read?
// 8.4M random 16-B records out of a 134 MB table, sequential 16-B writes.  Run under
// rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__device__ __forceinline__ u32x4 ld(const u32x4 *p) {
  u32x4 q;
  if (MODE == 0) return *p;
  if (MODE == 1) return __builtin_nontemporal_load(p);
  if (MODE == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(q) : "v"(p) : "memory");
  if (MODE == 3) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(q) : "v"(p) : "memory");
  if (MODE == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(q) : "v"(p) : "memory");
  if (MODE == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_waitcnt vmcnt(0)" : "=v"(q) : "v"(p) : "memory");
  if (MODE == 6) asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(q) : "v"(p) : "memory");
  return q;
}

template <int MODE>
__global__ __launch_bounds__(256) void rec_gather(const u32x4 *__restrict__ tab, u32x4 *__restrict__ out,
                                                  const uint32_t *__restrict__ idx, size_t n) {
  const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const uint32_t p = __builtin_nontemporal_load(idx + j);
  u32x4 q = ld<MODE>(tab + p);
  __builtin_nontemporal_store(q, out + j);
}

template <typename F>
static void timeit(const char *name, F launch) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  std::vector<float> ms;
  for (int i = 0; i < 6; ++i) { CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float t; CK(hipEventElapsedTime(&t, a, b)); if (i) ms.push_back(t); }
  std::sort(ms.begin(), ms.end());
  printf("%-28s median %8.1f us\n", name, ms[ms.size() / 2] * 1e3); fflush(stdout);
}

int main() {
  const size_t n = 8388608;
  u32x4 *tab, *out; uint32_t *idx;
  CK(hipMalloc(&tab, n * 16)); CK(hipMalloc(&out, n * 16)); CK(hipMalloc(&idx, n * 4));
  CK(hipMemset(tab, 1, n * 16));
  std::vector<uint32_t> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (uint32_t)i;
  std::mt19937_64 rng(7);
  for (size_t i = n; i-- > 1;) std::swap(h[i], h[rng() % (i + 1)]);
  CK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice));
  const unsigned blocks = (unsigned)(n / 256);
  timeit("plain", [&] { hipLaunchKernelGGL((rec_gather<0>), dim3(blocks), dim3(256), 0, 0, tab, out, idx, n); });
  timeit("nt", [&] { hipLaunchKernelGGL((rec_gather<1>), dim3(blocks), dim3(256), 0, 0, tab, out, idx, n); });
  timeit("asm sc0", [&] { hipLaunchKernelGGL((rec_gather<2>), dim3(blocks), dim3(256), 0, 0, tab, out, idx, n); });
  timeit("asm sc1", [&] { hipLaunchKernelGGL((rec_gather<3>), dim3(blocks), dim3(256), 0, 0, tab, out, idx, n); });
  timeit("asm sc0 sc1", [&] { hipLaunchKernelGGL((rec_gather<4>), dim3(blocks), dim3(256), 0, 0, tab, out, idx, n); });
  timeit("asm sc0 sc1 nt", [&] { hipLaunchKernelGGL((rec_gather<5>), dim3(blocks), dim3(256), 0, 0, tab, out, idx, n); });
  timeit("asm plain", [&] { hipLaunchKernelGGL((rec_gather<6>), dim3(blocks), dim3(256), 0, 0, tab, out, idx, n); });
  return 0;
}
