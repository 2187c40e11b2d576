// Minibatch gather + advantage statistics + standardisation kernels for gfx950 (MI355X).
// HBM-bound byte movement: no MFMA, no LDS tiling (nothing is reused); what matters is whole
// 16-byte-per-lane accesses, many rows in flight per wave and one sector per random read.
//
// Replaces Iterator::next of ExperienceBatcherIterator (ppo/src/model/data.rs:115-192):
//   5 x gather (data.rs:137-156), the advantage gather + standardisation (data.rs:157-164,
//   correct_std data.rs:196-200) and the six uploads into the device cache (data.rs:166-182,
//   not needed here: the kernel writes the device minibatch directly).
//
// One launch gathers ALL K minibatches of a pass: minibatch k is rows [k*B, (k+1)*B) of the
// output slabs and takes its indices from perm[start_k .. start_k+B), start_k = min(k*B, M-B)
// (the overlapping tail window of data.rs:121-125).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels_gae.h"

namespace ppoxp {

struct GatherArgs {
  const float *obs;        // [M, D]
  const Rec *rec;          // [M]
  const uint32_t *perm;    // [M]
  float *obs_out;          // [K*B, D]
  int64_t *act_out;        // [K*B]
  float *val_out, *nlp_out, *ret_out, *adv_out;  // [K*B]; adv_out holds RAW adv' until normalize
  double2 *partials;       // [K * tiles_per_mb] (sum, sum of squares) per 256-row tile
  size_t M, B;
  unsigned D, Q, qshift;   // Q = D / V vector elements per row; qshift = log2(Q) if POW2
  unsigned tiles_per_mb;   // ceil(B / 256)
};

// native clang vectors (the nontemporal builtins do not take HIP's float4 struct)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int V> struct VecT;
template <> struct VecT<1> { using type = float; };
template <> struct VecT<2> { using type = f32x2; };
template <> struct VecT<4> { using type = f32x4; };

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
  return x;
}

// grid = K * tiles_per_mb blocks of 256 threads; one wave = 64 consecutive output rows.
// V = floats per lane access (4 when D % 4 == 0), U = vector loads in flight per lane,
// NT = non-temporal obs loads/stores (each obs byte is touched once), POW2 = Q is a power of 2.
template <int V, int U, bool NT, bool POW2>
__global__ __launch_bounds__(256) void gather_kernel(GatherArgs g) {
  using vec = typename VecT<V>::type;
  __shared__ double red[4][2];
  const unsigned tile = blockIdx.x;
  const unsigned k = tile / g.tiles_per_mb;
  const unsigned c = tile - k * g.tiles_per_mb;
  const unsigned w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const size_t r0 = (size_t)c * 256u + w * 64u;               // first row of this wave inside minibatch k
  size_t start = (size_t)k * g.B;
  if (start + g.B > g.M) start = g.M - g.B;                   // data.rs:121-122
  const unsigned rows = r0 < g.B ? (unsigned)min((size_t)64, g.B - r0) : 0u;   // wave-uniform
  const size_t o0 = (size_t)k * g.B + r0;                     // first output row of this wave

  // ---- scalar streams: lane = row -------------------------------------------------------
  const bool valid = lane < rows;
  uint32_t p = 0;
  float4 lo = make_float4(0.f, 0.f, 0.f, 0.f);
  longlong2 hi; hi.x = 0; hi.y = 0;
  if (valid) {
    p = g.perm[start + r0 + lane];
    p = min(p, (uint32_t)(g.M - 1));                          // never read out of bounds
    const float4 *src = reinterpret_cast<const float4 *>(g.rec + p);
    lo = src[0];
    hi = *reinterpret_cast<const longlong2 *>(src + 1);
    const size_t o = o0 + lane;
    g.val_out[o] = lo.x;
    g.nlp_out[o] = lo.y;
    g.ret_out[o] = lo.z;
    g.adv_out[o] = lo.w;                                      // raw; standardised by normalize_kernel
    g.act_out[o] = hi.x;
  }
  // ---- advantage statistics: fp64 partial sums per tile ---------------------------------
  {
    double a = valid ? (double)lo.w : 0.0;
    double s1 = wave_sum(a), s2 = wave_sum(a * a);
    if (lane == 0) { red[w][0] = s1; red[w][1] = s2; }
  }

  // ---- observations: the wave's `rows` x D floats are contiguous in the output ----------
  const unsigned total = rows * g.Q;                          // vector elements of this wave
  const vec *obs = reinterpret_cast<const vec *>(g.obs);
  vec *out = reinterpret_cast<vec *>(g.obs_out) + o0 * g.Q;
  for (unsigned q0 = 0; q0 < total; q0 += 64u * U) {
    vec buf[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned q = q0 + u * 64u + lane;
      const unsigned row = POW2 ? (q >> g.qshift) : (q / g.Q);
      const unsigned col = q - row * g.Q;
      const uint32_t idx = __shfl(p, (int)(row & 63u), 64);   // all lanes take part
      if (q < total) {
        const vec *s = obs + (size_t)idx * g.Q + col;
        buf[u] = NT ? __builtin_nontemporal_load(s) : *s;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned q = q0 + u * 64u + lane;
      if (q < total) {
        if (NT) __builtin_nontemporal_store(buf[u], out + q); else out[q] = buf[u];
      }
    }
  }

  __syncthreads();
  if (threadIdx.x == 0) {
    double s1 = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
    double s2 = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
    g.partials[tile] = make_double2(s1, s2);
  }
}

// Fixed-order reduction of the tile partials of minibatch k -> sums[k] = (count, sum, sumsq).
// grid = K, block = 256.  Deterministic: the tree shape depends only on tiles_per_mb.
__global__ __launch_bounds__(256) void finalize_stats_kernel(const double2 *__restrict__ partials,
                                                             double *__restrict__ sums, unsigned tiles_per_mb,
                                                             size_t B) {
  __shared__ double s1s[256], s2s[256];
  const unsigned k = blockIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (unsigned i = threadIdx.x; i < tiles_per_mb; i += 256u) {
    double2 p = partials[(size_t)k * tiles_per_mb + i];
    s1 += p.x; s2 += p.y;
  }
  s1s[threadIdx.x] = s1; s2s[threadIdx.x] = s2;
  __syncthreads();
  for (unsigned off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) { s1s[threadIdx.x] += s1s[threadIdx.x + off]; s2s[threadIdx.x] += s2s[threadIdx.x + off]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    sums[3 * k + 0] = (double)B;
    sums[3 * k + 1] = s1s[0];
    sums[3 * k + 2] = s2s[0];
  }
}

// mean / unbiased std from (count, sum, sumsq) in fp64, then the reference's fp32 tail:
//   std = sqrt(var * (n / (n - 1)))  (data.rs:197-199),  out = (a - mean) / (std + 1e-8)  (data.rs:163)
__device__ __forceinline__ void stats_from_sums(const double *s, float &mean, float &stdv) {
  const double n = s[0];
  const double m = s[1] / n;
  double var = (s[2] - s[1] * m) / n;                         // population variance (dfdx var)
  var = var > 0.0 ? var : 0.0;
  mean = (float)m;
  const float nf = (float)n;
  const float corrected = (float)var * (nf / (nf - 1.0f));    // n == 1: 0 * inf = NaN, as the reference
  stdv = sqrtf(corrected);
}

// grid = (blocks_x, K), block = 256: in-place standardisation of adv_out[k*B .. (k+1)*B).
__global__ __launch_bounds__(256) void normalize_kernel(float *__restrict__ adv_out, const double *__restrict__ sums,
                                                        float *__restrict__ stats, size_t B) {
  const unsigned k = blockIdx.y;
  float mean, stdv;
  stats_from_sums(sums + 3 * k, mean, stdv);
  const float denom = stdv + 1e-8f;
  float *a = adv_out + (size_t)k * B;
  for (size_t j = (size_t)blockIdx.x * 256u + threadIdx.x; j < B; j += (size_t)gridDim.x * 256u)
    a[j] = (a[j] - mean) / denom;
  if (blockIdx.x == 0 && threadIdx.x == 0) { stats[2 * k] = mean; stats[2 * k + 1] = stdv; }
}

}  // namespace ppoxp
