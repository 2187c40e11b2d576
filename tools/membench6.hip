// membench6 — does the ALLOCATION flavour of the record table change what a random 16-B read costs?
// Default hipMalloc memory is cached by L2 and fetched as 128-B lines whatever the load flavour (membench3).
// Here the 134 MB table is allocated (a) hipMalloc, (b) hipExtMallocWithFlags(hipDeviceMallocUncached),
// (c) hipExtMallocWithFlags(hipDeviceMallocFinegrained) and 8.4M random 16-B records are gathered from it.
// Run plain for times, and under
//   rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
// for the request sizes (kernel names carry the variant).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int VARIANT, bool NT>
__global__ __launch_bounds__(256) void rec_gather(const u32x4 *__restrict__ tab, u32x4 *__restrict__ out,
                                                  const uint32_t *__restrict__ idx, size_t n) {
  const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const uint32_t p = __builtin_nontemporal_load(idx + j);
  u32x4 q = NT ? __builtin_nontemporal_load(tab + p) : tab[p];
  __builtin_nontemporal_store(q, out + j);
}

// the producer side: coalesced 16-B stores into the table (what the GAE scan does)
template <int VARIANT>
__global__ __launch_bounds__(256) void rec_fill(u32x4 *__restrict__ tab, size_t n) {
  const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  u32x4 q = {(unsigned)j, 1u, 2u, 3u};
  tab[j] = q;
}

template <typename F>
static void timeit(const char *name, F launch) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  std::vector<float> ms;
  for (int i = 0; i < 6; ++i) { CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float t; CK(hipEventElapsedTime(&t, a, b)); if (i) ms.push_back(t); }
  std::sort(ms.begin(), ms.end());
  printf("%-40s median %8.1f us\n", name, ms[ms.size() / 2] * 1e3); fflush(stdout);
}

template <int V>
static void run(const char *what, u32x4 *tab, u32x4 *out, uint32_t *idx, size_t n) {
  const unsigned blocks = (unsigned)(n / 256);
  char name[96];
  snprintf(name, sizeof name, "%s fill (16-B coalesced stores)", what);
  timeit(name, [&] { hipLaunchKernelGGL((rec_fill<V>), dim3(blocks), dim3(256), 0, 0, tab, n); });
  snprintf(name, sizeof name, "%s gather plain", what);
  timeit(name, [&] { hipLaunchKernelGGL((rec_gather<V, false>), dim3(blocks), dim3(256), 0, 0, tab, out, idx, n); });
  snprintf(name, sizeof name, "%s gather nt", what);
  timeit(name, [&] { hipLaunchKernelGGL((rec_gather<V, true>), dim3(blocks), dim3(256), 0, 0, tab, out, idx, n); });
}

int main() {
  const size_t n = 8388608;
  u32x4 *t0, *t1 = nullptr, *t2 = nullptr, *out; uint32_t *idx;
  CK(hipMalloc(&t0, n * 16)); CK(hipMalloc(&out, n * 16)); CK(hipMalloc(&idx, n * 4));
  hipError_t e1 = hipExtMallocWithFlags((void **)&t1, n * 16, hipDeviceMallocUncached);
  hipError_t e2 = hipExtMallocWithFlags((void **)&t2, n * 16, hipDeviceMallocFinegrained);
  printf("uncached alloc: %s, fine-grained alloc: %s\n", hipGetErrorString(e1), hipGetErrorString(e2));
  std::vector<uint32_t> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (uint32_t)i;
  std::mt19937_64 rng(7);
  for (size_t i = n; i-- > 1;) std::swap(h[i], h[rng() % (i + 1)]);
  CK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice));
  run<0>("hipMalloc      ", t0, out, idx, n);
  if (e1 == hipSuccess) run<1>("uncached       ", t1, out, idx, n);
  if (e2 == hipSuccess) run<2>("fine-grained   ", t2, out, idx, n);
  return 0;
}
