// Partitioned gather of the scalar streams (tunable "scalar_path" = 1): the same outputs as
// scalar_gather_kernel, bit for bit, without fetching a 128-byte record line per 16-byte record.
//
// Replaces the four scalar gathers of Iterator::next (ppo/src/model/data.rs:141-160) for passes whose
// records do not fit the Infinity Cache.  The direct gather reads 128 + 4 B and writes 24 B per row
// (156 B); here every stage streams:
//   A0  histogram of the rows by SOURCE window (2^17 records = 2 MiB of record lines)            4 R
//   A   (row, p) pairs binned by source window, LDS-staged runs                                   4 R +  8 W
//   B   per source window, on ONE XCD so that its L2 holds the window's lines: read the pairs,
//       fetch the records (each line leaves HBM once), emit {v, nlp, ret, act32, row} binned by
//       DESTINATION window (2^16 rows)                                                        8 + 16 R + 20 W
//   C   per destination window, on one XCD: scatter the five streams inside the window's 1.8 MB
//       of outputs, which its L2 writes back as whole lines                                      20 R + 28 W
//   S   fp64 (sum, sum of squares) per 256-row tile of the raw advantages, row order (the very
//       partials scalar_gather_kernel produces: same bits)                                         4 R
// = 112 B per row.  Order inside a bin depends on atomics and is not reproducible; the OUTPUTS are
// (every row is written once, with values that do not depend on the order).
//
// Workgroup -> XCD: hardware deals workgroups round-robin over the 8 XCDs (blockIdx % 8); the maps
// below only use that for SPEED (which L2 sees a window), never for correctness.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../ppo_amd/csrc/kernels_gae.h"
#include "../../ppo_amd/csrc/kernels_gather.h"

namespace ppoxp {

constexpr unsigned kSrcShift = 17, kDstShift = 16;          // 2^17 records / 2^16 rows per window
constexpr unsigned kPartChunk = 4096;                       // rows per workgroup in the binning stages
constexpr unsigned kMaxBins = 1024;

struct PartArgs {
  const uint32_t *perm;      // [M]
  size_t M;                  // rows of the pass (K*B == M: no overlapping tail window)
  const Rec *rec;
  const int64_t *actions;
  uint32_t *src_count;       // [nS] histogram, then (exclusive scan) running cursors of stage A
  uint32_t *dst_cursor;      // [nD] rows already placed in each destination bin
  uint2 *pairs;              // [M] (row, p) by source window
  uint32_t *entries;         // [M][5] {v, nlp, ret, act32, row} by destination window; bin d starts at d << kDstShift
  int64_t *act_out;
  float *val_out, *nlp_out, *ret_out, *adv_out;
  unsigned nS, nD;
};

// chunk handled by workgroup b when the chunks of one window (per_window of them) should all land on one XCD and
// an XCD should finish a window before it starts the next
__device__ __forceinline__ size_t xcd_window_chunk(unsigned b, unsigned per_window) {
  const unsigned x = b & 7u, seq = b >> 3;
  const size_t window = (size_t)(seq / per_window) * 8u + x;
  return window * per_window + seq % per_window;
}

// A0: rows per source window.  grid = ceil(M / 4096), block = 256.
__global__ __launch_bounds__(256) void part_histogram_kernel(PartArgs a) {
  __shared__ unsigned cnt[kMaxBins];
  for (unsigned i = threadIdx.x; i < a.nS; i += 256u) cnt[i] = 0;
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * kPartChunk;
#pragma unroll 4
  for (unsigned i = threadIdx.x; i < kPartChunk; i += 256u) {
    const size_t row = base + i;
    if (row < a.M) atomicAdd(&cnt[min(a.perm[row], (uint32_t)(a.M - 1)) >> kSrcShift], 1u);
  }
  __syncthreads();
  for (unsigned i = threadIdx.x; i < a.nS; i += 256u)
    if (cnt[i]) atomicAdd(&a.src_count[i], cnt[i]);
}

// counts -> exclusive prefix (the first free slot of every source bin); also clears the destination cursors.
// One workgroup of 1024 threads.
__global__ __launch_bounds__(1024) void part_scan_kernel(PartArgs a) {
  __shared__ unsigned s[kMaxBins];
  const unsigned t = threadIdx.x;
  const unsigned v = t < a.nS ? a.src_count[t] : 0u;
  s[t] = v;
  __syncthreads();
  for (unsigned off = 1; off < kMaxBins; off <<= 1) {
    const unsigned x = t >= off ? s[t - off] : 0u;
    __syncthreads();
    s[t] += x;
    __syncthreads();
  }
  if (t < a.nS) a.src_count[t] = s[t] - v;
  for (unsigned i = t; i < a.nD; i += 1024u) a.dst_cursor[i] = 0u;
}

// Places the items of one workgroup (keys in `key`, n_items of them, bins < nbins) so that items of one bin are
// adjacent in LDS order: returns for every item its slot in [0, n_items) and fills off[] (first slot of each bin in
// the workgroup) and gbase[] (first global slot the workgroup reserved in each bin).
template <int BS, int IPT>
__device__ __forceinline__ void bin_place(const unsigned (&key)[IPT], const bool (&live)[IPT], unsigned nbins,
                                          unsigned *cnt, unsigned *off, unsigned *gbase, uint32_t *cursor,
                                          unsigned (&slot)[IPT]) {
  for (unsigned i = threadIdx.x; i < nbins; i += BS) cnt[i] = 0;
  __syncthreads();
  unsigned rank[IPT];
#pragma unroll
  for (int k = 0; k < IPT; ++k) rank[k] = live[k] ? atomicAdd(&cnt[key[k]], 1u) : 0u;
  __syncthreads();
  // exclusive scan of cnt over nbins (<= 1024) by the first wave-fulls of threads: Hillis-Steele in place on off[]
  for (unsigned i = threadIdx.x; i < kMaxBins; i += BS) off[i] = i < nbins ? cnt[i] : 0u;
  __syncthreads();
  for (unsigned d = 1; d < kMaxBins; d <<= 1) {
    unsigned x[kMaxBins / BS > 0 ? kMaxBins / BS : 1];
    int n = 0;
    for (unsigned i = threadIdx.x; i < kMaxBins; i += BS) x[n++] = i >= d ? off[i - d] : 0u;
    __syncthreads();
    n = 0;
    for (unsigned i = threadIdx.x; i < kMaxBins; i += BS) off[i] += x[n++];
    __syncthreads();
  }
  for (unsigned i = threadIdx.x; i < nbins; i += BS) {
    const unsigned c = cnt[i];
    off[i] -= c;                                             // inclusive -> exclusive
    gbase[i] = c ? atomicAdd(&cursor[i], c) : 0u;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < IPT; ++k) slot[k] = live[k] ? off[key[k]] + rank[k] : 0u;
}

// A: (row, p) pairs by source window.  grid = ceil(M / 4096), block = 256, 16 rows per thread.
__global__ __launch_bounds__(256) void part_pairs_kernel(PartArgs a) {
  __shared__ unsigned cnt[kMaxBins], off[kMaxBins], gbase[kMaxBins];
  __shared__ uint2 stage[kPartChunk];
  constexpr int IPT = kPartChunk / 256;
  const size_t base = (size_t)blockIdx.x * kPartChunk;
  unsigned key[IPT], slot[IPT], p[IPT];
  bool live[IPT];
#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    const size_t row = base + threadIdx.x + 256u * k;
    live[k] = row < a.M;
    p[k] = live[k] ? min(a.perm[row], (uint32_t)(a.M - 1)) : 0u;
    key[k] = p[k] >> kSrcShift;
  }
  bin_place<256, IPT>(key, live, a.nS, cnt, off, gbase, a.src_count, slot);
#pragma unroll
  for (int k = 0; k < IPT; ++k)
    if (live[k]) stage[slot[k]] = make_uint2((unsigned)(base + threadIdx.x + 256u * k), p[k]);
  __syncthreads();
  const unsigned n = (unsigned)min((size_t)kPartChunk, a.M - base);
  for (unsigned i = threadIdx.x; i < n; i += 256u) {
    const uint2 q = stage[i];
    const unsigned s = q.y >> kSrcShift;
    a.pairs[(size_t)gbase[s] + (i - off[s])] = q;            // runs of one bin: consecutive lanes, consecutive slots
  }
}

// B: records of one source window (XCD-resident) -> entries by destination window.
// grid = ceil(nS/8)*8*32 workgroups of 1024 threads (4 pairs each), 1 workgroup per CU (LDS), so that one XCD works
// on about one window (32 CUs x 4096 pairs = 2^17) at a time.
__global__ __launch_bounds__(1024) void part_records_kernel(PartArgs a) {
  __shared__ unsigned cnt[kMaxBins], off[kMaxBins], gbase[kMaxBins];
  __shared__ uint32_t stage[kPartChunk * 5];                 // 80 KB: one workgroup per CU
  constexpr int IPT = kPartChunk / 1024;
  const size_t chunk = xcd_window_chunk(blockIdx.x, (1u << kSrcShift) / kPartChunk);
  const size_t base = chunk * kPartChunk;
  if (base >= a.M) return;                                   // workgroup-uniform
  unsigned key[IPT], slot[IPT];
  bool live[IPT];
  uint32_t e[IPT][5];
#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    const size_t i = base + threadIdx.x + 1024u * k;
    live[k] = i < a.M;
    key[k] = 0;
    if (live[k]) {
      const uint2 q = a.pairs[i];                            // (row, p)
      const u32x2_t *rp = reinterpret_cast<const u32x2_t *>(a.rec + rec_static_idx(q.y));
      const u32x2_t qs = *rp, qd = *(rp + 8);                // {nlp, act32}, {v, ret}: one line, resident in this XCD's L2
      if (qs.y == kRecActEscape) a.act_out[q.x] = a.actions[q.y];   // rare: the full 64-bit action goes out directly
      e[k][0] = qd.x; e[k][1] = qs.x; e[k][2] = qd.y; e[k][3] = qs.y; e[k][4] = q.x;
      key[k] = q.x >> kDstShift;
    }
  }
  bin_place<1024, IPT>(key, live, a.nD, cnt, off, gbase, a.dst_cursor, slot);
#pragma unroll
  for (int k = 0; k < IPT; ++k)
    if (live[k]) {
#pragma unroll
      for (int j = 0; j < 5; ++j) stage[slot[k] * 5u + j] = e[k][j];
    }
  __syncthreads();
  const unsigned n = (unsigned)min((size_t)kPartChunk, a.M - base);
  for (unsigned w = threadIdx.x; w < n * 5u; w += 1024u) {   // the staged entries as one dword stream
    const unsigned i = w / 5u, j = w - i * 5u;
    const unsigned d = stage[i * 5u + 4u] >> kDstShift;
    const size_t dst = ((size_t)d << kDstShift) + gbase[d] + (i - off[d]);
    a.entries[dst * 5u + j] = stage[w];
  }
}

// C: entries of one destination window (XCD-resident outputs) -> the five streams.  Thread = entry.
// grid = ceil(nD/8)*8*256 workgroups of 256 threads.
__global__ __launch_bounds__(256) void part_scatter_kernel(PartArgs a) {
  const size_t chunk = xcd_window_chunk(blockIdx.x, (1u << kDstShift) / 256u);
  const size_t i = chunk * 256u + threadIdx.x;
  if (i >= a.M) return;
  const uint32_t *e = a.entries + i * 5u;
  const uint32_t v = e[0], nlp = e[1], ret = e[2], act = e[3], row = e[4];
  a.val_out[row] = __uint_as_float(v);
  a.nlp_out[row] = __uint_as_float(nlp);
  a.ret_out[row] = __uint_as_float(ret);
  if (act != kRecActEscape) a.act_out[row] = (int64_t)act;
  a.adv_out[row] = __uint_as_float(ret) - __uint_as_float(v);   // data.rs:62
}

// S: the tile partials scalar_gather_kernel would have written, from the raw advantages in row order.
// grid = K * tiles_per_mb, block = 256 (tile = 256 rows of one minibatch).
__global__ __launch_bounds__(256) void part_tile_stats_kernel(const float *__restrict__ adv_out, double2 *__restrict__ partials,
                                                              size_t B, unsigned tiles_per_mb) {
  __shared__ double red[4][2];
  const unsigned tile = blockIdx.x, k = tile / tiles_per_mb, c = tile - k * tiles_per_mb;
  const size_t r = (size_t)c * 256u + threadIdx.x;
  const double x = r < B ? (double)adv_out[(size_t)k * B + r] : 0.0;
  const double s1 = wave_sum(x), s2 = wave_sum(x * x);
  const unsigned w = threadIdx.x >> 6;
  if ((threadIdx.x & 63u) == 0) { red[w][0] = s1; red[w][1] = s2; }
  __syncthreads();
  if (threadIdx.x == 0)
    partials[tile] = make_double2((red[0][0] + red[1][0]) + (red[2][0] + red[3][0]),
                                  (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]));
}

}  // namespace ppoxp
