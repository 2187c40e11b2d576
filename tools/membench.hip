// membench — what the HBM of this box gives to the access shapes of the experience path:
// streaming copy, read-only, write-only, random-row gather (random read, sequential write) and
// random-row scatter (sequential read, random write) at several row sizes.  Measurement tool
// only (DESIGN.md quotes its numbers); not part of libppo_xp.so.
//   hipcc --offload-arch=gfx950 -O3 -o membench tools/membench.hip && ./membench
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool NT, int U>
__global__ __launch_bounds__(256) void copy_k(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, size_t n) {
  size_t base = ((size_t)blockIdx.x * U) * 256 + threadIdx.x;
  for (; base < n; base += (size_t)gridDim.x * U * 256) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { size_t i = base + (size_t)u * 256; if (i < n) v[u] = NT ? __builtin_nontemporal_load(src + i) : src[i]; }
#pragma unroll
    for (int u = 0; u < U; ++u) { size_t i = base + (size_t)u * 256; if (i < n) { if (NT) __builtin_nontemporal_store(v[u], dst + i); else dst[i] = v[u]; } }
  }
}

template <bool NT>
__global__ __launch_bounds__(256) void read_k(const f32x4 *__restrict__ src, float *out, size_t n) {
  f32x4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    f32x4 v = NT ? __builtin_nontemporal_load(src + i) : src[i];
    acc += v;
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.f;
}

template <bool NT>
__global__ __launch_bounds__(256) void write_k(f32x4 *__restrict__ dst, size_t n) {
  f32x4 v = {1, 2, 3, 4};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
  }
}

// rows of Q float4; each wave handles 64 consecutive destination (GATHER) or source (SCATTER) rows.
template <bool SCATTER, bool NT, int U>
__global__ __launch_bounds__(256) void rows_k(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst,
                                             const uint32_t *__restrict__ idx, size_t rows, unsigned Q, unsigned qshift) {
  const unsigned lane = threadIdx.x & 63u;
  const size_t r0 = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
  if (r0 >= rows) return;
  const unsigned nr = (unsigned)min((size_t)64, rows - r0);
  const uint32_t p = lane < nr ? idx[r0 + lane] : 0u;
  const unsigned total = nr * Q;
  for (unsigned q0 = 0; q0 < total; q0 += 64u * U) {
    f32x4 buf[U];
    uint32_t ix[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned q = q0 + u * 64u + lane;
      const unsigned row = q >> qshift, col = q & (Q - 1);
      ix[u] = __shfl(p, (int)(row & 63u), 64);
      if (q < total) {
        const f32x4 *s = SCATTER ? src + (r0 * Q + q) : src + ((size_t)ix[u] * Q + col);
        buf[u] = NT ? __builtin_nontemporal_load(s) : *s;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned q = q0 + u * 64u + lane;
      const unsigned col = q & (Q - 1);
      if (q < total) {
        f32x4 *d = SCATTER ? dst + ((size_t)ix[u] * Q + col) : dst + (r0 * Q + q);
        if (NT) __builtin_nontemporal_store(buf[u], d); else *d = buf[u];
      }
    }
  }
}

template <typename F>
static void timeit(const char *name, double bytes, F launch) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  std::vector<float> ms;
  for (int i = 0; i < 7; ++i) {
    CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float t; CK(hipEventElapsedTime(&t, a, b)); if (i) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  printf("%-44s median %8.1f us  %7.0f GB/s   best %7.0f GB/s\n", name, ms[ms.size() / 2] * 1e3,
         bytes / (ms[ms.size() / 2] * 1e-3) / 1e9, bytes / (ms[0] * 1e-3) / 1e9);
  fflush(stdout);
}

int main(int argc, char **argv) {
  const size_t bytes = (argc > 1 ? atof(argv[1]) : 2.147483648) * 1e9;   // per array
  const size_t n4 = bytes / 16;
  f32x4 *src, *dst; float *out;
  CK(hipMalloc(&src, n4 * 16)); CK(hipMalloc(&dst, n4 * 16)); CK(hipMalloc(&out, 4));
  CK(hipMemset(src, 1, n4 * 16)); CK(hipMemset(dst, 0, n4 * 16));
  printf("array %.3f GB each\n", n4 * 16 / 1e9);
  const unsigned full = (unsigned)((n4 + 255) / 256);
  for (unsigned grid : {2048u, 8192u, 0u}) {
    char nm[128];
    unsigned g1 = grid ? grid : full;
    snprintf(nm, sizeof nm, "copy plain U1 grid %u", g1); timeit(nm, 2.0 * n4 * 16, [&] { hipLaunchKernelGGL((copy_k<false, 1>), dim3(g1), dim3(256), 0, 0, src, dst, n4); });
    snprintf(nm, sizeof nm, "copy nt    U1 grid %u", g1); timeit(nm, 2.0 * n4 * 16, [&] { hipLaunchKernelGGL((copy_k<true, 1>), dim3(g1), dim3(256), 0, 0, src, dst, n4); });
    unsigned g4 = grid ? grid : (full + 3) / 4;
    snprintf(nm, sizeof nm, "copy plain U4 grid %u", g4); timeit(nm, 2.0 * n4 * 16, [&] { hipLaunchKernelGGL((copy_k<false, 4>), dim3(g4), dim3(256), 0, 0, src, dst, n4); });
    snprintf(nm, sizeof nm, "copy nt    U4 grid %u", g4); timeit(nm, 2.0 * n4 * 16, [&] { hipLaunchKernelGGL((copy_k<true, 4>), dim3(g4), dim3(256), 0, 0, src, dst, n4); });
    unsigned g8 = grid ? grid : (full + 7) / 8;
    snprintf(nm, sizeof nm, "copy nt    U8 grid %u", g8); timeit(nm, 2.0 * n4 * 16, [&] { hipLaunchKernelGGL((copy_k<true, 8>), dim3(g8), dim3(256), 0, 0, src, dst, n4); });
  }
  timeit("read plain grid 8192", 1.0 * n4 * 16, [&] { hipLaunchKernelGGL((read_k<false>), dim3(8192), dim3(256), 0, 0, src, out, n4); });
  timeit("read nt    grid 8192", 1.0 * n4 * 16, [&] { hipLaunchKernelGGL((read_k<true>), dim3(8192), dim3(256), 0, 0, src, out, n4); });
  timeit("write plain grid 8192", 1.0 * n4 * 16, [&] { hipLaunchKernelGGL((write_k<false>), dim3(8192), dim3(256), 0, 0, dst, n4); });
  timeit("write nt    grid 8192", 1.0 * n4 * 16, [&] { hipLaunchKernelGGL((write_k<true>), dim3(8192), dim3(256), 0, 0, dst, n4); });
  timeit("hipMemcpyDtoD", 2.0 * n4 * 16, [&] { CK(hipMemcpyAsync(dst, src, n4 * 16, hipMemcpyDeviceToDevice, 0)); });

  for (unsigned rowbytes : {64u, 256u, 1024u, 4096u}) {
    const unsigned Q = rowbytes / 16;
    unsigned qshift = 0; while ((1u << qshift) < Q) ++qshift;
    const size_t rows = n4 / Q;
    std::vector<uint32_t> h(rows);
    for (size_t i = 0; i < rows; ++i) h[i] = (uint32_t)i;
    std::mt19937_64 rng(7);
    for (size_t i = rows; i-- > 1;) std::swap(h[i], h[rng() % (i + 1)]);
    uint32_t *idx; CK(hipMalloc(&idx, rows * 4)); CK(hipMemcpy(idx, h.data(), rows * 4, hipMemcpyHostToDevice));
    const unsigned blocks = (unsigned)((rows + 255) / 256);
    const double by = 2.0 * rows * rowbytes + rows * 4.0;
    char nm[128];
    snprintf(nm, sizeof nm, "gather  %4u-B rows nt U4", rowbytes); timeit(nm, by, [&] { hipLaunchKernelGGL((rows_k<false, true, 4>), dim3(blocks), dim3(256), 0, 0, src, dst, idx, rows, Q, qshift); });
    snprintf(nm, sizeof nm, "gather  %4u-B rows nt U8", rowbytes); timeit(nm, by, [&] { hipLaunchKernelGGL((rows_k<false, true, 8>), dim3(blocks), dim3(256), 0, 0, src, dst, idx, rows, Q, qshift); });
    snprintf(nm, sizeof nm, "gather  %4u-B rows plain U4", rowbytes); timeit(nm, by, [&] { hipLaunchKernelGGL((rows_k<false, false, 4>), dim3(blocks), dim3(256), 0, 0, src, dst, idx, rows, Q, qshift); });
    snprintf(nm, sizeof nm, "scatter %4u-B rows nt U4", rowbytes); timeit(nm, by, [&] { hipLaunchKernelGGL((rows_k<true, true, 4>), dim3(blocks), dim3(256), 0, 0, src, dst, idx, rows, Q, qshift); });
    snprintf(nm, sizeof nm, "scatter %4u-B rows plain U4", rowbytes); timeit(nm, by, [&] { hipLaunchKernelGGL((rows_k<true, false, 4>), dim3(blocks), dim3(256), 0, 0, src, dst, idx, rows, Q, qshift); });
    // identity order: the same kernel with sequential indices = its streaming ceiling
    for (size_t i = 0; i < rows; ++i) h[i] = (uint32_t)i;
    CK(hipMemcpy(idx, h.data(), rows * 4, hipMemcpyHostToDevice));
    snprintf(nm, sizeof nm, "gather  %4u-B rows nt U4 IDENTITY order", rowbytes); timeit(nm, by, [&] { hipLaunchKernelGGL((rows_k<false, true, 4>), dim3(blocks), dim3(256), 0, 0, src, dst, idx, rows, Q, qshift); });
    CK(hipFree(idx));
  }
  return 0;
}
