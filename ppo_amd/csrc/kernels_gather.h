// Minibatch gather + advantage statistics + standardisation kernels for gfx950 (MI355X).
// HBM-bound byte movement: no MFMA, no LDS tiling (nothing is reused).
//
// Replaces Iterator::next of ExperienceBatcherIterator (ppo/src/model/data.rs:115-192):
//   5 x gather (data.rs:137-156), the advantage gather + standardisation (data.rs:157-164,
//   correct_std data.rs:196-200) and the six uploads into the device cache (data.rs:166-182,
//   not needed here: the kernels write the device minibatch directly).
//
// One pass gathers ALL K minibatches: minibatch k is rows [k*B, (k+1)*B) of the output slabs and
// takes its indices from perm[start_k .. start_k+B), start_k = min(k*B, M-B) (the overlapping
// tail window of data.rs:121-125).
//
// Shape of the kernels (measured with tools/membench*.hip on MI355X, profiles/r1/membench*.log):
//   * a one-shot grid with ONE 16-byte access per thread and 64-thread workgroups moves random
//     256-B rows at 6.08 TB/s (95 % of the best streaming copy, 6.4 TB/s); a wave that loops over
//     64 rows with 4-8 loads in flight per lane reaches only 5.5, and persistent grids are slower
//     still.  So the observation gather is "thread = one float4 of one output row".
//   * non-temporal loads AND stores are worth 12 % (each observation byte is touched once).
//   * the five scalar streams travel as ONE 16-byte record per transition (kernels_gae.h), so a
//     gathered row costs one random sector, not four.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bijection.h"
#include "kernels_gae.h"

namespace ppoxp {

// native clang vectors (the nontemporal builtins do not take HIP's float4 struct)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int V> struct VecT;
template <> struct VecT<1> { using type = float; };
template <> struct VecT<2> { using type = f32x2; };
template <> struct VecT<4> { using type = f32x4; };

// ---- device-resident insert (buffer.rs:84-121 for rows that are already on the GPU) ----------------------
// ONE launch per inserted row instead of six copies and two pack kernels: the first obs_blocks workgroups copy
// the observation row, the rest copy the five scalar rows and keep the records' static halves and the done
// bits in step (what pack_static_kernel / pack_done_bits_kernel do for the host-side insert).
struct InsertArgs {
  const float *obs, *rewards, *values, *dones, *neglogps;   // source rows: [N, D], [N] x 4
  const uint8_t *dones_u8;                                   // when set: the done flags as Rust bools instead of `dones`
                                                             // (`dones[i] as u8 as f32`, buffer.rs:95)
  const int64_t *actions;                                    // [N]
  float *obs_dst, *rew_dst, *val_dst, *done_dst, *nlp_dst;  // row `idx` of the buffer's streams
  int64_t *act_dst;
  Rec *rec;                                                  // whole record array
  uint32_t *bits_row;                                        // done bits of row idx: [W]
  size_t i0;                                                 // flat index of the row's first transition (idx * N)
  size_t obs_elems;                                          // elements (of V floats) in the observation row
  unsigned N, W, obs_blocks;
};

template <int V>
__global__ __launch_bounds__(256) void insert_row_kernel(const InsertArgs a) {
  using T = typename VecT<V>::type;
  if (blockIdx.x < a.obs_blocks) {
    const size_t j = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (j < a.obs_elems)
      __builtin_nontemporal_store(__builtin_nontemporal_load(reinterpret_cast<const T *>(a.obs) + j), reinterpret_cast<T *>(a.obs_dst) + j);
    return;
  }
  const unsigned e = (blockIdx.x - a.obs_blocks) * 256u + threadIdx.x, e0 = e & ~63u;
  bool d = false;
  if (e < a.N) {
    const float df = a.dones_u8 ? (a.dones_u8[e] != 0 ? 1.0f : 0.0f) : a.dones[e], nlp = a.neglogps[e];
    const int64_t act = a.actions[e];
    d = df != 0.0f;
    a.rew_dst[e] = a.rewards[e];
    a.val_dst[e] = a.values[e];
    a.done_dst[e] = df;
    a.nlp_dst[e] = nlp;
    a.act_dst[e] = act;
    store_rec_static(a.rec, a.i0 + e, nlp, act);
  }
  const unsigned long long m = __ballot(d);
  if ((threadIdx.x & 63u) == 0 && e0 < a.N) {
    a.bits_row[e0 >> 5] = (uint32_t)m;
    if ((e0 >> 5) + 1 < a.W) a.bits_row[(e0 >> 5) + 1] = (uint32_t)(m >> 32);
  }
}

struct ObsGatherArgs {
  const float *obs;        // [M, D]
  const uint32_t *perm;    // [M]
  float *obs_out;          // [K*B, D]
  size_t M, B, total;      // total = Kn*B*Q vector elements to move (Kn minibatches in this launch)
  size_t row_base;         // first output row of the launch in the pass (k0*B; 0 unless minibatches are streamed one by one)
  size_t tail_from, tail_shift;   // TAIL: positions >= tail_from = (K-1)*B belong to the overlapping last window, which
                                  // starts at M - B (data.rs:121-122): they read position pos - tail_shift, tail_shift = K*B - M
  unsigned Q, qshift;      // Q = D / V vector elements per row; qshift = log2(Q) when POW2
  unsigned nblocks, xcd;   // grid size; xcd: 0 = identity, 1 = 8x8 chunk remap, 2 = one contiguous eighth per XCD
};

// Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 labels the L2 a block uses) and
// one 128-byte line of the permutation feeds 32 rows = several 64-thread workgroups.  Unmapped,
// those workgroups sit on 8 different L2s and each fetches the line again (measured: 2.1 M extra
// 128-B reads per pass, profiles/r1/pmc_TGT_v2_split_gather.txt).  The remaps make workgroups
// that share an XCD work on consecutive tiles.  Both are bijections on [0, nblocks).
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned nblocks, unsigned mode) {
  if (mode == 1) {                       // inside every aligned group of 64 workgroups: XCD x takes tiles 8x..8x+7
    const unsigned base = b & ~63u;
    if (base + 64u > nblocks) return b;  // ragged last group stays as it is
    return base + ((b & 7u) << 3) + ((b >> 3) & 7u);
  }
  if (mode == 2) {                       // XCD x streams through its own contiguous eighth of the tiles
    const unsigned q = nblocks >> 3, r = nblocks & 7u, x = b & 7u;
    return (x < r ? x * (q + 1u) : r * (q + 1u) + (x - r) * q) + (b >> 3);
  }
  return b;
}

// Observation rows.  grid = ceil(K*B*Q / BS) workgroups of BS threads, one vector element each.
// V = floats per access (4 when D % 4 == 0, else 2 or 1), POW2 = Q is a power of two,
// TAIL = the last window overlaps (B does not divide M), NT = non-temporal accesses.
template <int V, int BS, bool NT, bool POW2, bool TAIL>
__device__ __forceinline__ void obs_gather_body(const ObsGatherArgs &g, size_t block) {
  using vec = typename VecT<V>::type;
  const size_t e = block * BS + threadIdx.x;                  // output vector element
  if (e >= g.total) return;
  size_t row; unsigned col;                                   // output row (k*B + r), vector column
  if (POW2) { row = e >> g.qshift; col = (unsigned)e & (g.Q - 1); }
  else { row = e / g.Q; col = (unsigned)(e - row * g.Q); }
  size_t pos = g.row_base + row;                              // position in the permutation
  if (TAIL && pos >= g.tail_from) pos -= g.tail_shift;        // data.rs:121-122
  uint32_t p = g.perm[pos];
  p = min(p, (uint32_t)(g.M - 1));                            // a bad index can never fault
  const vec *s = reinterpret_cast<const vec *>(g.obs) + ((size_t)p * g.Q + col);
  vec *d = reinterpret_cast<vec *>(g.obs_out) + e;
  if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(s), d);
  else *d = *s;
}

template <int V, int BS, bool NT, bool POW2, bool TAIL>
__global__ __launch_bounds__(BS) void obs_gather_kernel(ObsGatherArgs g) {
  obs_gather_body<V, BS, NT, POW2, TAIL>(g, xcd_remap(blockIdx.x, g.nblocks, g.xcd));
}

struct ScalarGatherArgs {
  const Rec *rec;          // record lines of the M transitions (kernels_gae.h)
  const int64_t *actions;  // [M] (only read for actions that do not fit the record)
  const uint32_t *perm;    // [M]
  int64_t *act_out;        // [K*B]
  float *val_out, *nlp_out, *ret_out, *adv_out;  // [K*B]; adv_out holds RAW adv' until normalize
  double2 *partials;       // [K * tiles_per_mb] (sum, sum of squares) per 256-row tile
  size_t M, B;
  unsigned tiles_per_mb;   // ceil(B / rows per tile); a tile = the rows of one workgroup = one (sum, sumsq) partial
  unsigned k0;             // first minibatch of the launch (0 unless minibatches are streamed one by one)
  int rec_nt, out_nt;      // cache policy of the record reads / of the output stores
};

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
  return x;
}

// Scalar streams + advantage statistics of one tile of BS rows (BS = 256: scalar_gather_kernel, BS = 64: the
// scalar workgroups of mixed_gather_kernel).  Lane = one output row: one random 16-byte record read, five
// coalesced stores, and the (sum, sum of squares) of the tile's raw advantages in fp64.
template <int BS>
__device__ __forceinline__ void scalar_gather_body(const ScalarGatherArgs &g, unsigned tile) {
  __shared__ double red[BS / 64][2];
  const unsigned k = tile / g.tiles_per_mb;
  const unsigned c = tile - k * g.tiles_per_mb;
  const unsigned w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const size_t r = (size_t)c * BS + threadIdx.x;              // row inside minibatch k
  size_t start = (size_t)(g.k0 + k) * g.B;
  if (start + g.B > g.M) start = g.M - g.B;                   // data.rs:121-122
  const bool valid = r < g.B;
  float adv = 0.f;
  if (valid) {
    uint32_t p = __builtin_nontemporal_load(g.perm + start + r);
    p = min(p, (uint32_t)(g.M - 1));
    // the two halves of the record sit in one 128-byte line: one fetch, the second load hits
    const u32x2_t *rp = reinterpret_cast<const u32x2_t *>(g.rec + rec_static_idx(p));
    const u32x2_t qs = g.rec_nt ? __builtin_nontemporal_load(rp) : *rp;
    const u32x2_t qd = g.rec_nt ? __builtin_nontemporal_load(rp + 8) : *(rp + 8);
    const float nlp = __uint_as_float(qs.x), v = __uint_as_float(qd.x), ret = __uint_as_float(qd.y);
    int64_t act = (int64_t)qs.y;
    if (qs.y == kRecActEscape) act = g.actions[p];             // rare: action does not fit 32 bits
    adv = ret - v;                                            // data.rs:62 (same op as the GAE kernel's adv')
    const size_t o = (size_t)k * g.B + r;
    if (g.out_nt) {
      __builtin_nontemporal_store(v, g.val_out + o);
      __builtin_nontemporal_store(nlp, g.nlp_out + o);
      __builtin_nontemporal_store(ret, g.ret_out + o);
      __builtin_nontemporal_store(act, g.act_out + o);
    } else {
      g.val_out[o] = v; g.nlp_out[o] = nlp; g.ret_out[o] = ret; g.act_out[o] = act;
    }
    g.adv_out[o] = adv;                                       // raw; standardised (re-read) by normalize_kernel
  }
  const double a = valid ? (double)adv : 0.0;
  const double s1 = wave_sum(a), s2 = wave_sum(a * a);
  if (BS == 64) {
    if (lane == 0) g.partials[tile] = make_double2(s1, s2);
    return;
  }
  if (lane == 0) { red[w][0] = s1; red[w][1] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t1 = 0.0, t2 = 0.0;
    if (BS == 256) {                                          // the tree of round 1: same bits
      t1 = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
      t2 = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
    } else {
      for (int i = 0; i < BS / 64; ++i) { t1 += red[i][0]; t2 += red[i][1]; }
    }
    g.partials[tile] = make_double2(t1, t2);
  }
}

// grid = K * tiles_per_mb workgroups of 256 threads.
__global__ __launch_bounds__(256) void scalar_gather_kernel(ScalarGatherArgs g) { scalar_gather_body<256>(g, blockIdx.x); }

// Observation rows AND the scalar streams of RPW = BS/Q output rows in one workgroup (D % 4 == 0, Q = D/4 a
// power of two, RPW <= 64, B % RPW == 0).  Thread = one 16-byte element of one row, exactly as in
// obs_gather_kernel; the first RPW lanes of wave 0 also fetch the rows' record lines and store the five
// scalar streams (RPW*4 contiguous bytes per stream) and the tile's fp64 (sum, sum of squares).
// Why: with BS = 512 and D = 64 a workgroup covers 32 rows = exactly one 128-byte line of the permutation,
// which is then fetched once instead of by 8 workgroups on 8 XCDs (0.27 GB -> 0.03 GB at TGT), and the record
// lines travel with the rows' stream instead of in a launch of their own: 839 us against 713 + 188 at TGT.
// It only pays while the records (16 B x M) fit the Infinity Cache beside the stream; with 0.5 - 2 GB of
// records (C3, C5 on one GPU) the random line fetches disturb the row stream and the split kernels win
// (profiles/r2/ab_fused_shapes.log), so the host picks it by size.
struct FusedGatherArgs {
  ObsGatherArgs o;
  ScalarGatherArgs s;
  unsigned rpw;            // rows per workgroup = BS / Q  (<= 64) = rows per statistics tile
  Bijection bij;           // BIJ != 0: perm(j) is evaluated from this, o.perm / s.perm are not read
};

// BIJ = 0: the permutation is an array (o.perm).  BIJ = 1: keyed bijection (bijection.h), evaluated ONCE per output row
// by the first rpw lanes of wave 0 (the lanes that also gather the row's scalar streams), handed to the row's threads
// through LDS: no index array is written, uploaded or read, and no thread waits for an index load before it can issue
// its row load.  BIJ = 2: every thread evaluates its own row's index (no LDS, no barrier; 8x the integer work).
template <int BS, bool NT, bool TAIL, int BIJ = 0>
__global__ __launch_bounds__(BS) void fused_gather_kernel(FusedGatherArgs g) {
  __shared__ uint32_t src_row[64];
  const unsigned Q = g.o.Q, rpw = g.rpw;
  const size_t row0 = (size_t)blockIdx.x * rpw;
  const size_t KB = g.o.total >> g.o.qshift;                  // K*B output rows
  uint32_t p_lane = 0;                                        // BIJ = 1: source row of output row row0 + lane (wave 0)
  if (BIJ == 1) {
    if (threadIdx.x < rpw && row0 + threadIdx.x < KB) {
      size_t pos = g.o.row_base + row0 + threadIdx.x;
      if (TAIL && pos >= g.o.tail_from) pos -= g.o.tail_shift;   // data.rs:121-122
      p_lane = bij_perm(g.bij, (uint32_t)pos);
      src_row[threadIdx.x] = p_lane;
    }
    __syncthreads();
  }
  // ---- observation element of this thread
  {
    const size_t row = row0 + (threadIdx.x >> g.o.qshift);
    const unsigned col = threadIdx.x & (Q - 1);
    if (row < KB) {
      size_t pos = g.o.row_base + row;
      if (TAIL && BIJ != 1 && pos >= g.o.tail_from) pos -= g.o.tail_shift;
      uint32_t p;
      if (BIJ == 1) p = src_row[threadIdx.x >> g.o.qshift];
      else if (BIJ == 2) p = bij_perm(g.bij, (uint32_t)pos);
      else p = g.o.perm[pos];
      p = min(p, (uint32_t)(g.o.M - 1));
      const f32x4 *s = reinterpret_cast<const f32x4 *>(g.o.obs) + ((size_t)p * Q + col);
      f32x4 *d = reinterpret_cast<f32x4 *>(g.o.obs_out) + (row * Q + col);
      if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(s), d);
      else *d = *s;
    }
  }
  // ---- scalar streams of the workgroup's rows: the first rpw lanes of wave 0
  if (threadIdx.x < 64 && row0 < KB) {
    const size_t row = row0 + threadIdx.x;
    const bool valid = threadIdx.x < rpw && row < KB;
    float adv = 0.f;
    if (valid) {
      size_t pos = g.o.row_base + row;
      if (TAIL && pos >= g.o.tail_from) pos -= g.o.tail_shift;   // data.rs:121-122
      uint32_t p;
      if (BIJ == 1) p = p_lane;
      else if (BIJ == 2) p = bij_perm(g.bij, (uint32_t)pos);
      else p = g.s.perm[pos];
      p = min(p, (uint32_t)(g.s.M - 1));
      const u32x2_t *rp = reinterpret_cast<const u32x2_t *>(g.s.rec + rec_static_idx(p));
      const u32x2_t qs = *rp, qd = *(rp + 8);
      const float nlp = __uint_as_float(qs.x), v = __uint_as_float(qd.x), ret = __uint_as_float(qd.y);
      int64_t act = (int64_t)qs.y;
      if (qs.y == kRecActEscape) act = g.s.actions[p];
      adv = ret - v;                                          // data.rs:62
      __builtin_nontemporal_store(v, g.s.val_out + row);
      __builtin_nontemporal_store(nlp, g.s.nlp_out + row);
      __builtin_nontemporal_store(ret, g.s.ret_out + row);
      __builtin_nontemporal_store(act, g.s.act_out + row);
      g.s.adv_out[row] = adv;
    }
    const double a = valid ? (double)adv : 0.0;
    const double s1 = wave_sum(a), s2 = wave_sum(a * a);
    if (threadIdx.x == 0) g.s.partials[blockIdx.x] = make_double2(s1, s2);   // tile = the workgroup's rows (B % rpw == 0)
  }
}

// The fused gather with ITER row groups per workgroup (keyed-bijection passes only): a workgroup of BS threads covers
// RPW = ITER * BS/Q output rows - every thread loads its 16-byte element of each of the ITER groups, then stores them; the
// first RPW threads evaluate perm(j) for the workgroup's rows, gather the rows' scalar streams and add up ONE (sum, sum of
// squares) partial for all RPW rows: a quarter of the workgroups, barriers and partials at ITER = 4.
// As compiled, each guarded load is waited for before the next one is issued (one row load in flight per thread; checked in
// the ISA).  That is the shape that wins: the same kernel with the ITER loads issued back to back LOSES 3 % at TGT
// (882 against 857 us per step with one row group; this kernel: 850), like every deeper-than-one-load shape measured in
// round 1; a plain load-store-next loop ties, 256 rows per workgroup lose 4 % (profiles/r3/ab_fused_row_groups_isa.log).
template <int BS, int ITER, bool NT, bool TAIL>
__global__ __launch_bounds__(BS) void fused_gather_loop_kernel(FusedGatherArgs g) {
  __shared__ uint32_t src_row[256];
  __shared__ double red[4][2];
  const unsigned Q = g.o.Q, rpi = (unsigned)BS >> g.o.qshift, rpw = g.rpw;      // rpw = ITER * rpi <= 256
  const size_t row0 = (size_t)blockIdx.x * rpw;
  const size_t KB = g.o.total >> g.o.qshift;
  if (threadIdx.x < rpw && row0 + threadIdx.x < KB) {
    size_t pos = g.o.row_base + row0 + threadIdx.x;
    if (TAIL && pos >= g.o.tail_from) pos -= g.o.tail_shift;   // data.rs:121-122
    src_row[threadIdx.x] = min(bij_perm(g.bij, (uint32_t)pos), (uint32_t)(g.o.M - 1));
  }
  __syncthreads();
  const unsigned col = threadIdx.x & (Q - 1), rsub = threadIdx.x >> g.o.qshift;
  f32x4 x[ITER];
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const unsigned r = it * rpi + rsub;
    if (row0 + r < KB) {
      const f32x4 *s = reinterpret_cast<const f32x4 *>(g.o.obs) + ((size_t)src_row[r] * Q + col);
      x[it] = NT ? __builtin_nontemporal_load(s) : *s;
    }
  }
  // scalar streams of row row0 + threadIdx.x: the first rpw threads (whole waves take part in the sums)
  const unsigned nw = (rpw + 63u) >> 6;
  if (threadIdx.x < nw * 64u) {
    const size_t row = row0 + threadIdx.x;
    const bool valid = threadIdx.x < rpw && row < KB;
    float adv = 0.f;
    if (valid) {
      const uint32_t p = src_row[threadIdx.x];
      const u32x2_t *rp = reinterpret_cast<const u32x2_t *>(g.s.rec + rec_static_idx(p));
      const u32x2_t qs = *rp, qd = *(rp + 8);
      const float nlp = __uint_as_float(qs.x), v = __uint_as_float(qd.x), ret = __uint_as_float(qd.y);
      int64_t act = (int64_t)qs.y;
      if (qs.y == kRecActEscape) act = g.s.actions[p];
      adv = ret - v;                                          // data.rs:62
      __builtin_nontemporal_store(v, g.s.val_out + row);
      __builtin_nontemporal_store(nlp, g.s.nlp_out + row);
      __builtin_nontemporal_store(ret, g.s.ret_out + row);
      __builtin_nontemporal_store(act, g.s.act_out + row);
      g.s.adv_out[row] = adv;
    }
    const double a = valid ? (double)adv : 0.0;
    const double s1 = wave_sum(a), s2 = wave_sum(a * a);
    if ((threadIdx.x & 63u) == 0) { red[threadIdx.x >> 6][0] = s1; red[threadIdx.x >> 6][1] = s2; }
  }
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const unsigned r = it * rpi + rsub;
    if (row0 + r < KB) {
      f32x4 *d = reinterpret_cast<f32x4 *>(g.o.obs_out) + ((row0 + r) * Q + col);
      if (NT) __builtin_nontemporal_store(x[it], d);
      else *d = x[it];
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t1 = 0.0, t2 = 0.0;
    for (unsigned i = 0; i < nw; ++i) { t1 += red[i][0]; t2 += red[i][1]; }   // fixed order
    g.s.partials[blockIdx.x] = make_double2(t1, t2);
  }
}

// perm[j] = bij_perm(j) for j < M: the keyed bijection written out as an index array, for the gather kernels that
// read one (rows wider than a workgroup, records beyond the Infinity Cache) and for ppo_xp_pass_perm_device / _host.
__global__ __launch_bounds__(256) void bijection_perm_kernel(Bijection b, uint32_t *__restrict__ perm) {
  const size_t j = (size_t)blockIdx.x * 256u + threadIdx.x;
  if (j < b.M) perm[j] = bij_perm(b, (uint32_t)j);
}

// Fixed-order reduction of the tile partials of one minibatch by a 256-thread workgroup; the result is
// valid in every thread after the call.  Deterministic: the tree shape depends only on tiles_per_mb, so
// every workgroup that runs it (finalize_stats_kernel, or each normalize workgroup when it folds the
// partials itself) gets the same bits.
__device__ __forceinline__ void reduce_partials(const double2 *__restrict__ partials, unsigned tiles_per_mb,
                                                double (*red)[2], double &t1, double &t2) {
  double s1 = 0.0, s2 = 0.0;
  for (unsigned i = threadIdx.x; i < tiles_per_mb; i += 256u) {
    double2 p = partials[i];
    s1 += p.x; s2 += p.y;
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2);                       // lanes, then the four waves: one barrier
  if ((threadIdx.x & 63u) == 0) { red[threadIdx.x >> 6][0] = s1; red[threadIdx.x >> 6][1] = s2; }
  __syncthreads();
  t1 = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
  t2 = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
}

// sums[k] = (count, sum, sumsq) of minibatch k.  grid = K, block = 256.  Launched when the sums
// themselves are needed (sharded passes: they cross the ranks) or when a minibatch has too many tiles
// for every normalize workgroup to fold them again.
__global__ __launch_bounds__(256) void finalize_stats_kernel(const double2 *__restrict__ partials,
                                                             double *__restrict__ sums, unsigned tiles_per_mb,
                                                             size_t B) {
  __shared__ double red[4][2];
  const unsigned k = blockIdx.x;
  double t1, t2;
  reduce_partials(partials + (size_t)k * tiles_per_mb, tiles_per_mb, red, t1, t2);
  if (threadIdx.x == 0) {
    sums[3 * k + 0] = (double)B;
    sums[3 * k + 1] = t1;
    sums[3 * k + 2] = t2;
  }
}

// First level of a two-level fold for minibatches with many small tiles (the fused gather writes one partial
// per 32 rows): workgroup (g, k) adds tiles [256 g, 256 g + 256) of minibatch k in a fixed order and stores
// partials2[k * G + g]; normalize_kernel<FOLD> then folds the G second-level partials.  grid = (G, K).
__global__ __launch_bounds__(256) void reduce_partials_kernel(const double2 *__restrict__ partials,
                                                              double2 *__restrict__ partials2, unsigned tiles_per_mb) {
  __shared__ double red[4][2];
  const unsigned g = blockIdx.x, k = blockIdx.y, G = gridDim.x;
  const unsigned i = g * 256u + threadIdx.x;
  double2 p = make_double2(0.0, 0.0);
  if (i < tiles_per_mb) p = partials[(size_t)k * tiles_per_mb + i];
  const double s1 = wave_sum(p.x), s2 = wave_sum(p.y);
  const unsigned w = threadIdx.x >> 6;
  if ((threadIdx.x & 63u) == 0) { red[w][0] = s1; red[w][1] = s2; }
  __syncthreads();
  if (threadIdx.x == 0)
    partials2[(size_t)k * G + g] = make_double2((red[0][0] + red[1][0]) + (red[2][0] + red[3][0]),
                                                (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]));
}

// mean / unbiased std from (count, sum, sumsq) in fp64, then the reference's fp32 tail:
//   std = sqrt(var * (n / (n - 1)))  (data.rs:197-199),  out = (a - mean) / (std + 1e-8)  (data.rs:163)
__device__ __forceinline__ void stats_from_sums(const double *s, float &mean, float &stdv) {
  const double n = s[0];
  const double m = s[1] / n;
  double var = (s[2] - s[1] * m) / n;                         // population variance (dfdx var)
  var = var < 0.0 ? 0.0 : var;                                // cancellation only; a NaN (inf / NaN advantages) stays NaN as in the reference
  mean = (float)m;
  const float nf = (float)n;
  const float corrected = (float)var * (nf / (nf - 1.0f));    // n == 1: 0 * inf = NaN, as the reference
  stdv = sqrtf(corrected);
}

// In-place standardisation of adv_out[k*B .. (k+1)*B).  grid = K * tiles_per_mb blocks (1-D, so K
// is not limited by gridDim.y), tiles_per_mb = ceil(B / 1024), block = 256, four consecutive floats
// per thread (16-byte accesses when the minibatch base allows it).
// FOLD: no finalize_stats launch ran; every workgroup reduces the minibatch's `stat_tiles` partials
// itself (same tree, same bits) and workgroup 0 of the minibatch also stores sums[k].
// VPT: 16-byte vectors per thread (a workgroup covers 1024 * VPT rows): the fold is paid once per workgroup, so
// big minibatches use 4.
template <bool FOLD, int VPT = 1>
__global__ __launch_bounds__(256) void normalize_kernel(float *__restrict__ adv_out, double *__restrict__ sums,
                                                        const double2 *__restrict__ partials, unsigned stat_tiles,
                                                        float *__restrict__ stats, size_t B, unsigned tiles_per_mb) {
  const unsigned k = blockIdx.x / tiles_per_mb;
  const unsigned bx = blockIdx.x - k * tiles_per_mb;
  float *a = adv_out + (size_t)k * B;
  size_t j[VPT];
  bool vec[VPT];
  f32x4 x[VPT];
#pragma unroll
  for (int v = 0; v < VPT; ++v) {
    j[v] = (((size_t)bx * VPT + v) * 256u + threadIdx.x) * 4u;
    vec[v] = j[v] + 4 <= B && ((reinterpret_cast<uintptr_t>(a + j[v]) & 15u) == 0);
    x[v] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (FOLD && vec[v]) x[v] = *reinterpret_cast<f32x4 *>(a + j[v]);   // in flight while the partials are folded
  }
  double s[3];
  if (FOLD) {
    __shared__ double red[4][2];
    reduce_partials(partials + (size_t)k * stat_tiles, stat_tiles, red, s[1], s[2]);
    s[0] = (double)B;
    if (bx == 0 && threadIdx.x == 0) { sums[3 * (size_t)k] = s[0]; sums[3 * (size_t)k + 1] = s[1]; sums[3 * (size_t)k + 2] = s[2]; }
  } else {
    s[0] = sums[3 * (size_t)k]; s[1] = sums[3 * (size_t)k + 1]; s[2] = sums[3 * (size_t)k + 2];
  }
  float mean, stdv;
  stats_from_sums(s, mean, stdv);
  const float denom = stdv + 1e-8f;
#pragma unroll
  for (int v = 0; v < VPT; ++v) {
    if (vec[v]) {
      if (!FOLD) x[v] = *reinterpret_cast<f32x4 *>(a + j[v]);   // measured: loading first costs 2 us at TGT
      f32x4 y = x[v];
      y.x = (y.x - mean) / denom; y.y = (y.y - mean) / denom; y.z = (y.z - mean) / denom; y.w = (y.w - mean) / denom;
      *reinterpret_cast<f32x4 *>(a + j[v]) = y;
    } else {
      for (size_t i = j[v]; i < B && i < j[v] + 4; ++i) a[i] = (a[i] - mean) / denom;
    }
  }
  if (bx == 0 && threadIdx.x == 0) { stats[2 * (size_t)k] = mean; stats[2 * (size_t)k + 1] = stdv; }
}

// explained_variance (ppo/src/training.rs:88-99) needs var(returns) and var(returns - values), both
// with ddof 0.  One pass: fp64 (sum, sum of squares) of both streams per block -> partials[block]
// = (S_ret, Q_ret, S_adv, Q_adv); the host adds the few thousand partials in block order.
// rec != nullptr: the flat arrays were not written by the scan; ret and adv' = fl(ret - v) come from the records.
__global__ __launch_bounds__(256) void moments_kernel(const float *__restrict__ ret, const float *__restrict__ adv,
                                                      const Rec *__restrict__ rec, size_t M,
                                                      double *__restrict__ partials) {
  __shared__ double red[4][4];
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < M; i += (size_t)gridDim.x * 256u) {
    double r, a;
    if (rec) {                                                  // kernel-uniform
      float vf, rf;
      load_rec_scan(rec, i, vf, rf);
      r = rf; a = rf - vf;                                      // data.rs:62 in fp32, then widened
    } else {
      r = ret[i]; a = adv[i];
    }
    s[0] += r; s[1] += r * r; s[2] += a; s[3] += a * a;
  }
  const unsigned w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
#pragma unroll
  for (int k = 0; k < 4; ++k) { s[k] = wave_sum(s[k]); if (lane == 0) red[w][k] = s[k]; }
  __syncthreads();
  if (threadIdx.x < 4) {
    const unsigned k = threadIdx.x;
    partials[4 * (size_t)blockIdx.x + k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
  }
}

}  // namespace ppoxp
