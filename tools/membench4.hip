// membench4 — is a random 16-byte SCATTER cheaper than the random 16-byte GATHER (which costs a
// 128-B line per record, membench3)?  8.4M records, 134 MB table.  Also 32-byte records.
// gather:  out[j] = tab[idx[j]]   (random read, sequential write)
// scatter: out[idx[i]] = tab[i]   (sequential read, random write), plain and non-temporal stores
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void rec_gather(const u32x4 *__restrict__ tab, u32x4 *__restrict__ out,
                                                  const uint32_t *__restrict__ idx, size_t n) {
  const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const uint32_t p = __builtin_nontemporal_load(idx + j);
  __builtin_nontemporal_store(tab[p], out + j);
}

template <bool NT>
__global__ __launch_bounds__(256) void rec_scatter(const u32x4 *__restrict__ tab, u32x4 *__restrict__ out,
                                                   const uint32_t *__restrict__ idx, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t p = __builtin_nontemporal_load(idx + i);
  const u32x4 q = __builtin_nontemporal_load(tab + i);
  if (NT) __builtin_nontemporal_store(q, out + p); else out[p] = q;
}

// 32-byte records (two u32x4), lane pair per record would be better; here one lane moves both halves
template <bool NT>
__global__ __launch_bounds__(256) void rec_scatter32(const u32x4 *__restrict__ tab, u32x4 *__restrict__ out,
                                                     const uint32_t *__restrict__ idx, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t p = __builtin_nontemporal_load(idx + i);
  const u32x4 a = __builtin_nontemporal_load(tab + 2 * i), b = __builtin_nontemporal_load(tab + 2 * i + 1);
  if (NT) { __builtin_nontemporal_store(a, out + 2 * (size_t)p); __builtin_nontemporal_store(b, out + 2 * (size_t)p + 1); }
  else { out[2 * (size_t)p] = a; out[2 * (size_t)p + 1] = b; }
}

__global__ __launch_bounds__(256) void unpack(const u32x4 *__restrict__ aos, float *__restrict__ a, float *__restrict__ b,
                                              float *__restrict__ c, long long *__restrict__ d, float *__restrict__ e, size_t n) {
  const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const u32x4 q = __builtin_nontemporal_load(aos + j);
  __builtin_nontemporal_store(__uint_as_float(q.x), a + j);
  __builtin_nontemporal_store(__uint_as_float(q.y), b + j);
  __builtin_nontemporal_store(__uint_as_float(q.z), c + j);
  __builtin_nontemporal_store((long long)q.w, d + j);
  e[j] = __uint_as_float(q.z) - __uint_as_float(q.x);
}

template <typename F>
static void timeit(const char *name, F launch) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  std::vector<float> ms;
  for (int i = 0; i < 8; ++i) { CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float t; CK(hipEventElapsedTime(&t, a, b)); if (i) ms.push_back(t); }
  std::sort(ms.begin(), ms.end());
  printf("%-34s median %8.1f us\n", name, ms[ms.size() / 2] * 1e3); fflush(stdout);
}

int main() {
  const size_t n = 8388608;
  u32x4 *tab, *out; uint32_t *idx; float *s[4]; long long *s64;
  CK(hipMalloc(&tab, n * 32)); CK(hipMalloc(&out, n * 32)); CK(hipMalloc(&idx, n * 4));
  for (auto &p : s) CK(hipMalloc(&p, n * 4));
  CK(hipMalloc(&s64, n * 8));
  CK(hipMemset(tab, 1, n * 32));
  std::vector<uint32_t> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (uint32_t)i;
  std::mt19937_64 rng(7);
  for (size_t i = n; i-- > 1;) std::swap(h[i], h[rng() % (i + 1)]);
  CK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice));
  const unsigned blocks = (unsigned)(n / 256);
  timeit("gather 16 B", [&] { hipLaunchKernelGGL(rec_gather, dim3(blocks), dim3(256), 0, 0, tab, out, idx, n); });
  timeit("scatter 16 B plain store", [&] { hipLaunchKernelGGL((rec_scatter<false>), dim3(blocks), dim3(256), 0, 0, tab, out, idx, n); });
  timeit("scatter 16 B nt store", [&] { hipLaunchKernelGGL((rec_scatter<true>), dim3(blocks), dim3(256), 0, 0, tab, out, idx, n); });
  timeit("scatter 32 B plain store", [&] { hipLaunchKernelGGL((rec_scatter32<false>), dim3(blocks), dim3(256), 0, 0, tab, out, idx, n); });
  timeit("scatter 32 B nt store", [&] { hipLaunchKernelGGL((rec_scatter32<true>), dim3(blocks), dim3(256), 0, 0, tab, out, idx, n); });
  timeit("unpack AoS -> 5 streams", [&] { hipLaunchKernelGGL(unpack, dim3(blocks), dim3(256), 0, 0, out, s[0], s[1], s[2], s64, s[3], n); });
  return 0;
}
