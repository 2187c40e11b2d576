// GAE(lambda) / returns kernels for gfx950 (MI355X).  HBM-bound scans: no MFMA.
//
// Replaces ExperienceBuffer::returns (ppo/src/common/buffer.rs:187-225) fused with
// `advantages = returns - values` of ExperienceBatcher::new (ppo/src/model/data.rs:62).
//
// Layout: every scalar stream is [T, N] row-major (env index fastest), exactly the reference's
// Array2 layout (buffer.rs:32-36), so one wave64 = 64 consecutive envs reads 256 contiguous
// bytes per stream per time step.  Time runs in registers.
//
// Arithmetic: fp32, every operation rounded on its own, evaluation order of buffer.rs:217-222.
// This translation unit is compiled with -ffp-contract=off so that no FMA is formed.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ppoxp {

// Packed per-transition records read by the scalar gather: ONE 128-byte line fetch per gathered row instead
// of four scattered 4/8-byte reads (values, neglogps, returns, actions).  adv' is not stored: the gather
// recomputes fl(ret - v), the very operation of data.rs:62.
//
// A record line holds 8 consecutive transitions (flat index i = t*N + e, line i/8, slot i%8):
//   bytes [ 0,  64): eight STATIC halves {nlp, act32}  — known when the row is inserted (buffer.rs:93-113),
//                    written by pack_static_kernel at insert time, outside the timed path;
//   bytes [64, 128): eight SCAN halves   {v, ret}      — written by the GAE scan, 64 contiguous aligned
//                    bytes per 8 lanes, i.e. full 64-byte write requests.
// So the scan moves exactly its algorithmic 20 B per transition (12 read + 8 written) and never touches
// neglogps / actions, and a gathered row still costs one line.
// Actions are usize in the reference (buffer.rs:33).  The record keeps the low 32 bits; an action
// that does not fit (or equals the escape value itself) is stored as kRecActEscape and the gather
// fetches the full 64-bit value from the actions stream for that one row.
struct __attribute__((aligned(8))) RecHalf { uint32_t x, y; };
typedef RecHalf Rec;
static_assert(sizeof(RecHalf) == 8, "RecHalf must be 8 bytes");
constexpr uint32_t kRecActEscape = 0xFFFFFFFFu;
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

// index, in RecHalf units, of the static half of transition i; its scan half is 8 further
__host__ __device__ __forceinline__ size_t rec_static_idx(size_t i) { return ((i >> 3) << 4) + (i & 7); }
__host__ __device__ __forceinline__ size_t rec_halves(size_t transitions) { return ((transitions + 7) >> 3) << 4; }

__device__ __forceinline__ void store_rec_scan(Rec *rec, size_t i, float v, float ret) {
  u32x2_t q;
  q.x = __float_as_uint(v); q.y = __float_as_uint(ret);
  *reinterpret_cast<u32x2_t *>(rec + rec_static_idx(i) + 8) = q;
}
__device__ __forceinline__ void store_rec_static(Rec *rec, size_t i, float nlp, int64_t act) {
  const uint64_t a = (uint64_t)act;
  u32x2_t q;
  q.x = __float_as_uint(nlp);
  q.y = a >= (uint64_t)kRecActEscape ? kRecActEscape : (uint32_t)a;
  *reinterpret_cast<u32x2_t *>(rec + rec_static_idx(i)) = q;
}
__device__ __forceinline__ void load_rec_scan(const Rec *rec, size_t i, float &v, float &ret) {
  const u32x2_t q = *reinterpret_cast<const u32x2_t *>(rec + rec_static_idx(i) + 8);
  v = __uint_as_float(q.x); ret = __uint_as_float(q.y);
}

struct GaeArgs {
  const float *rewards, *values;                     // [T, N]
  const uint32_t *done_bits;                         // [T, W] W = ceil(N/32): bit e%32 of word e/32 = dones[t, e] != 0,
                                                     // kept in step with the f32 dones stream by the insert entry points
                                                     // (pack_done_bits_kernel): the scan reads 8 bytes per wave and step
                                                     // instead of 256; 1 - d is exactly 1.0f / 0.0f for d in {0, 1}
  unsigned W;
  const uint8_t *last_dones;                         // [N] Rust bools
  float *returns, *adv;                              // [T*N]
  Rec *rec;                                          // record lines of [T*N] transitions (scan halves written here)
  const float *carry;                                // [C, N] chunk carry-in (assoc) or nullptr
  float *sumP, *sumQ;                                // [C, N] chunk summaries (assoc)
  unsigned N, len, lvi, chunk;                       // chunk = time steps per blockIdx.y
  float gamma, gl;                                   // gl = fl(gamma*lambda)  (buffer.rs:219)
  int nt;                                            // non-temporal inputs / returns / adv (records keep the default policy)
  int flat;                                          // store the flat returns / adv arrays (always when no records are built);
                                                     // a pass reads only the records, so the packing scan skips these 8 B per
                                                     // transition unless a caller asked for the arrays (ppo_xp_views & co.)
};

// nonterminal factor 1 - dones[t, e] (buffer.rs:199) from the packed done bits
__device__ __forceinline__ uint32_t done_word(const GaeArgs &a, size_t t, unsigned e) { return a.done_bits[t * a.W + (e >> 5)]; }
__device__ __forceinline__ float nnt_of(uint32_t word, unsigned e) { return ((word >> (e & 31u)) & 1u) ? 0.0f : 1.0f; }

// One time step of buffer.rs:217-222, in the reference's rounding order.
__device__ __forceinline__ void gae_step(float r, float v, float v_next, float nnt_next, float gamma,
                                         float gl, float &A, float &ret, float &adv) {
  float x1 = v_next * nnt_next;
  float x2 = x1 * gamma;
  float x3 = r + x2;
  float delta = x3 - v;       // (rewards + nextvalues*nextnonterminal*gamma) - values
  float y1 = nnt_next * A;
  float y2 = y1 * gl;
  A = delta + y2;             // lastgaelam = delta + nextnonterminal*lastgaelam*(gamma*lambda)
  ret = A + v;                // buffer.rs:222
  adv = ret - v;              // data.rs:62
}

// K1 (and the fix-up phase of K2): sequential reverse-time scan, one lane per env.
// grid = (ceil(N/64), C), block = 64.  blockIdx.y = time chunk (C = 1 for the pure sequential
// algorithm).  Loads of the next U time steps are issued before the current U are consumed.
// SUMMARY = true turns the same loop into phase 1 of K2: chunk index = blockIdx.y + 1 (chunk 0 feeds
// nobody), carry-in 0, nothing stored per step, and at the end the chunk's affine map
//   A_{t_lo} = Q + P * A_{t_hi+1},   P = prod fl(nnt_{t+1} * gl),  Q = the scan started from 0
// goes to sumP / sumQ.  Q is computed with exactly the sequential kernel's operations.
template <bool PACK, int U, bool SUMMARY = false>
__global__ __launch_bounds__(64) void gae_scan_kernel(GaeArgs a) {
  const unsigned e = blockIdx.x * 64u + threadIdx.x;
  if (e >= a.N) return;
  const unsigned c = SUMMARY ? blockIdx.y + 1 : blockIdx.y;
  const int t_lo = (int)(c * a.chunk);
  int t_hi = (int)min((c + 1) * a.chunk, a.len) - 1;
  if (t_hi < t_lo) return;
  const size_t N = a.N;

  float v_next, nnt_next, A;
  if ((unsigned)t_hi == a.len - 1) {           // buffer.rs:209-210, 201-204
    v_next = a.values[(size_t)a.lvi * N + e];
    nnt_next = a.last_dones[e] ? 0.0f : 1.0f;
    A = 0.0f;                                  // buffer.rs:196
  } else {                                     // buffer.rs:212-213, 199
    v_next = a.values[(size_t)(t_hi + 1) * N + e];
    nnt_next = nnt_of(done_word(a, (size_t)(t_hi + 1), e), e);
    A = SUMMARY ? 0.0f : a.carry[(size_t)c * N + e];
  }
  float P = 1.0f;

  float r[U], v[U];
  uint32_t d[U];                               // done words (wave-uniform pairs: 8 bytes per wave and step)
  auto load = [&](int t_top) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int t = max(t_top - u, t_lo);            // clamped: tail re-reads row t_lo (cached)
      size_t i = (size_t)t * N + e;
      if (a.nt) {   // wave-uniform: streams read once must not push the records out of the Infinity Cache
        r[u] = __builtin_nontemporal_load(a.rewards + i);
        v[u] = __builtin_nontemporal_load(a.values + i);
      } else {
        r[u] = a.rewards[i];
        v[u] = a.values[i];
      }
      d[u] = done_word(a, (size_t)t, e);
    }
  };

  load(t_hi);
  for (int t_top = t_hi; t_top >= t_lo; t_top -= U) {
    float rc[U], vc[U];
    uint32_t dc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { rc[u] = r[u]; vc[u] = v[u]; dc[u] = d[u]; }
    if (t_top - U >= t_lo) load(t_top - U);    // prefetch the next U steps (wave-uniform)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int t = t_top - u;
      if (t >= t_lo) {
        float ret, adv;
        if (SUMMARY) P = P * (nnt_next * a.gl);
        gae_step(rc[u], vc[u], v_next, nnt_next, a.gamma, a.gl, A, ret, adv);
        if (!SUMMARY) {
          size_t i = (size_t)t * N + e;
          if (!PACK || a.flat) {                 // kernel-uniform
            if (a.nt) { __builtin_nontemporal_store(ret, a.returns + i); __builtin_nontemporal_store(adv, a.adv + i); }
            else { a.returns[i] = ret; a.adv[i] = adv; }
          }
          if (PACK) store_rec_scan(a.rec, i, vc[u], ret);
        }
        v_next = vc[u];
        nnt_next = nnt_of(dc[u], e);
      }
    }
  }
  if (SUMMARY) {
    a.sumP[(size_t)c * N + e] = P;
    a.sumQ[(size_t)c * N + e] = A;
  }
}

// K1w: the same sequential scan with W waves per 64 envs, for launches with too few env groups to
// cover the memory latency with one wave each (C2: 64 groups, C3: 256).  The scan is cut into
// super-chunks of W*S time steps; wave w of the workgroup owns S consecutive steps of the
// super-chunk and loads them (and the next super-chunk's) into registers, so W*S steps per env are
// in flight instead of U.  Everything that does not depend on the running A (delta and the
// non-terminal factors) is computed by all waves at once; only the three dependent operations per
// step  y1 = nnt*A, y2 = y1*gl, A = delta + y2  run wave after wave, the carry handed over through
// LDS.  Operations and their order per (t, e) are exactly gae_step's: results are bit-identical to
// gae_scan_kernel.  grid = ceil(N/64), block = 64*W.  Lanes past N work on a clamped env and store
// nothing, so every wave reaches every barrier.
template <bool PACK, int W, int S>
__global__ __launch_bounds__(64 * W) void gae_wide_kernel(GaeArgs a) {
  __shared__ float carry[64];
  const unsigned lane = threadIdx.x & 63u;
  const int w = (int)(threadIdx.x >> 6);
  const unsigned e_raw = blockIdx.x * 64u + lane;
  const bool live = e_raw < a.N;
  const size_t N = a.N;
  const size_t e = live ? e_raw : N - 1;
  const int len = (int)a.len;
  constexpr int span = W * S;

  float r[S], v[S], vb, db;                   // vb, db: value / done flag of the row above the slice
  uint32_t d[S];                              // done words
  auto load = [&](int top) {                  // top = latest time step of this wave's slice (may be < 0: empty)
#pragma unroll
    for (int u = 0; u < S; ++u) {
      const size_t i = (size_t)max(top - u, 0) * N + e;
      if (a.nt) {
        r[u] = __builtin_nontemporal_load(a.rewards + i);
        v[u] = __builtin_nontemporal_load(a.values + i);
      } else {
        r[u] = a.rewards[i]; v[u] = a.values[i];
      }
      d[u] = done_word(a, (size_t)max(top - u, 0), (unsigned)e);
    }
    if (top >= len - 1) {                     // buffer.rs:209-210, 201-204 (1 - 1 = 0, 1 - 0 = 1: same factors)
      vb = a.values[(size_t)a.lvi * N + e];
      db = a.last_dones[e] ? 1.0f : 0.0f;
    } else {                                  // buffer.rs:212-213, 199
      const size_t i = (size_t)max(top + 1, 0) * N + e;
      vb = a.values[i];
      db = 1.0f - nnt_of(done_word(a, (size_t)max(top + 1, 0), (unsigned)e), (unsigned)e);
    }
  };

  load(len - 1 - w * S);
  for (int sc_top = len - 1; sc_top >= 0; sc_top -= span) {
    const int top = sc_top - w * S;
    float rc[S], vc[S];
    uint32_t dc[S];
#pragma unroll
    for (int u = 0; u < S; ++u) { rc[u] = r[u]; vc[u] = v[u]; dc[u] = d[u]; }
    float v_next = vb, nnt_next = 1.0f - db;
    if (sc_top - span >= 0) load(sc_top - span - w * S);   // next super-chunk (workgroup-uniform)

    float delta[S], nnt[S];                   // the part of gae_step that does not depend on A
#pragma unroll
    for (int u = 0; u < S; ++u) {
      float x1 = v_next * nnt_next;
      float x2 = x1 * a.gamma;
      float x3 = rc[u] + x2;
      delta[u] = x3 - vc[u];
      nnt[u] = nnt_next;
      v_next = vc[u];
      nnt_next = nnt_of(dc[u], (unsigned)e);
    }
#pragma unroll 1
    for (int p = 0; p < W; ++p) {
      if (w == p) {                           // wave-uniform
        float A = (p == 0 && sc_top == len - 1) ? 0.0f : carry[lane];   // buffer.rs:196
#pragma unroll
        for (int u = 0; u < S; ++u) {
          if (top - u >= 0) {
            float y1 = nnt[u] * A;
            float y2 = y1 * a.gl;
            A = delta[u] + y2;
            delta[u] = A;                     // keep A_t for the stores below
          }
        }
        carry[lane] = A;
      }
      if (W > 1) __syncthreads();
      if (w == p && live) {                   // stores overlap the next wave's dependent chain
#pragma unroll
        for (int u = 0; u < S; ++u) {
          const int t = top - u;
          if (t >= 0) {
            const float ret = delta[u] + vc[u];          // buffer.rs:222
            const float adv = ret - vc[u];               // data.rs:62
            const size_t i = (size_t)t * N + e;
            if (!PACK || a.flat) {               // kernel-uniform
              if (a.nt) { __builtin_nontemporal_store(ret, a.returns + i); __builtin_nontemporal_store(adv, a.adv + i); }
              else { a.returns[i] = ret; a.adv[i] = adv; }
            }
            if (PACK) store_rec_scan(a.rec, i, vc[u], ret);
          }
        }
      }
    }
  }
}

// K2 phase 2: LDS-staged associative scan over the chunk summaries.
// carry[c] = A entering chunk c from the future:  carry[C-1] = 0,
//            carry[c] = Q[c+1] + P[c+1] * carry[c+1].
// One workgroup = 64 envs x W waves.  The C-1 maps are cut into W consecutive segments; each
// wave composes its segment (registers), the W segment maps are combined by a Hillis-Steele
// scan in LDS with the associative operator
//     (P_l, Q_l) o (P_e, Q_e) = (P_l * P_e, Q_l + P_l * Q_e)      (l applied after e)
// and each wave then replays its segment from its carry-in.  grid = ceil(N/64), block = 64*W.
template <int W>
__global__ __launch_bounds__(64 * W) void gae_carry_scan_kernel(const float *__restrict__ sumP,
                                                                const float *__restrict__ sumQ,
                                                                float *__restrict__ carry, unsigned N,
                                                                unsigned C) {
  __shared__ float sP[2][W][64];
  __shared__ float sQ[2][W][64];
  const unsigned lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
  const unsigned e = blockIdx.x * 64u + lane;
  const bool live = e < N;
  // processing order j = 0 .. J-1 applies the map of chunk c = C-1-j;  Y_0 = 0, Y_{j+1} = f_j(Y_j),
  // carry[C-1-j] = Y_j.
  const unsigned J = C - 1;
  const unsigned S = (J + W - 1) / W;
  const unsigned j0 = min(w * S, J), j1 = min(j0 + S, J);
  float P = 1.0f, Q = 0.0f;
  if (live) {
    for (unsigned j = j0; j < j1; ++j) {
      size_t i = (size_t)(C - 1 - j) * N + e;
      float p = sumP[i], q = sumQ[i];
      Q = q + p * Q;
      P = p * P;
    }
  }
  int cur = 0;
  sP[0][w][lane] = P; sQ[0][w][lane] = Q;
  __syncthreads();
#pragma unroll
  for (unsigned off = 1; off < (unsigned)W; off <<= 1) {
    float p = sP[cur][w][lane], q = sQ[cur][w][lane];
    if (w >= off) {
      float pe = sP[cur][w - off][lane], qe = sQ[cur][w - off][lane];
      q = q + p * qe;
      p = p * pe;
    }
    sP[cur ^ 1][w][lane] = p; sQ[cur ^ 1][w][lane] = q;
    cur ^= 1;
    __syncthreads();
  }
  if (!live) return;
  float Y = (w == 0) ? 0.0f : sQ[cur][w - 1][lane];  // inclusive prefix of earlier segments applied to 0
  if (w == 0) carry[(size_t)(C - 1) * N + e] = 0.0f;
  for (unsigned j = j0; j < j1; ++j) {
    size_t i = (size_t)(C - 1 - j) * N + e;
    Y = sumQ[i] + sumP[i] * Y;
    carry[(size_t)(C - 2 - j) * N + e] = Y;
  }
}

// K2 in ONE launch ("one-pass" associative scan).  Same decomposition as the three kernels above - chunk maps (P, Q),
// carries by map application, exact replay of the recurrence from the carry - but every chunk is loaded from HBM ONCE
// and kept in registers between the two phases, and the carries are exchanged inside the running grid:
//   workgroup = W waves = W consecutive chunks of L time steps of one group of 64 envs (a "super-chunk"); work is handed
//   out by a ticket counter in reverse-time order, so a workgroup only ever waits for workgroups that already run.
//   phase A  wave w loads its L steps (r, v: L x 2 registers per lane; the done flags of the L rows as two words per lane,
//            broadcast with readlane), computes delta_t exactly as gae_step does, keeps delta_t, v_t and the
//            non-terminal bits, and runs the recurrence from A = 0 to get its map  A_out = Q + P * A_in.
//   publish  the W maps are composed in a fixed order (LDS) into the super-chunk's map, which wave 0 publishes as two
//            64-bit words {serial tag, float} per env with relaxed agent-scope atomic stores: the tag travels in the
//            same word as the value, so no fence is needed and nothing has to be cleared between calls.
//   look-back wave 0 reads the maps of ALL earlier super-chunks (they depend on nothing, so they are published as soon as
//            their phase A ends: no chain of waits) and applies them one after the other to 0: the carry entering this
//            super-chunk.  Applying maps in a fixed order makes the result independent of timing: deterministic.
//   phase B  the carry is pushed through the W maps (LDS), every wave replays its L steps from its carry with the
//            sequential kernel's three dependent operations per step and stores ret / adv' (or the record halves).
// HBM traffic: 9 B read + 8 B written per transition, one launch (the three-kernel path reads the inputs twice).
// The spin on a tag is bounded: a wave that gives up sets *err and finishes, so the grid always drains.
struct OnePassArgs {
  GaeArgs a;
  unsigned long long *words;     // [S][N][2] tagged (P, Q) of every super-chunk
  unsigned long long *ticket;    // work counter; the workgroup that draws the last ticket resets it to 0
  unsigned *err;                 // set to 1 when a look-back gave up
  unsigned serial, S, groups;    // tag of this call (never 0); super-chunks per env group; env groups
};

constexpr unsigned kOnePassMaxSupers = 64;   // super-chunks per env group the kernel can look back over (LDS: 2 x 16 KB)

__device__ __forceinline__ unsigned long long op_pack(unsigned tag, float x) {
  return ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(x);
}

template <bool PACK, int L, int W>
__global__ __launch_bounds__(64 * W) void gae_onepass_kernel(OnePassArgs o) {
  static_assert(L <= 64, "the non-terminal bits of a chunk live in one 64-bit mask");
  __shared__ unsigned s_ticket;
  __shared__ float sP[W][64], sQ[W][64], sY[W][64];
  __shared__ float lbP[kOnePassMaxSupers][64], lbQ[kOnePassMaxSupers][64];   // maps of the earlier super-chunks
  const GaeArgs &a = o.a;
  if (threadIdx.x == 0) {
    const unsigned long long t = atomicAdd(o.ticket, 1ull);
    if (t + 1 == (unsigned long long)o.S * o.groups) __hip_atomic_store(o.ticket, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_ticket = (unsigned)t;
  }
  __syncthreads();
  const unsigned s = s_ticket / o.groups, g = s_ticket - s * o.groups;   // reverse-time order: s = 0 holds the last steps
  const unsigned lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
  const unsigned e_raw = g * 64u + lane;
  const bool live = e_raw < a.N;
  const size_t N = a.N;
  const size_t e = live ? e_raw : N - 1;
  const int len = (int)a.len;
  const int t_top = len - 1 - (int)((s * W + w) * L);          // latest step of this wave's chunk; < 0: no chunk
  const int steps = t_top < 0 ? 0 : (t_top + 1 < L ? t_top + 1 : L);

  // ---- phase A: the chunk into registers
  float dl[L], vv[L];
  float v_next, nnt_next;
  unsigned long long mask = 0;                                 // bit u: non-terminal factor used by step u (row t_top - u)
  if (steps > 0) {                                             // wave-uniform
#pragma unroll
    for (int u = 0; u < L; ++u) {
      const size_t i = (size_t)(t_top - u > 0 ? t_top - u : 0) * N + e;
      if (a.nt) {                                              // every input byte is read exactly once
        dl[u] = __builtin_nontemporal_load(a.rewards + i);
        vv[u] = __builtin_nontemporal_load(a.values + i);
      } else {
        dl[u] = a.rewards[i];
        vv[u] = a.values[i];
      }
    }
    // done flags of row t_top - lane, both words of this env group, in lane `lane`
    const size_t drow = (size_t)(t_top - (int)lane > 0 ? t_top - (int)lane : 0) * a.W + (size_t)((g * 64u) >> 5);
    const uint32_t dw0 = a.done_bits[drow];
    const uint32_t dw1 = ((g * 64u) >> 5) + 1 < a.W ? a.done_bits[drow + 1] : 0u;
    if (t_top == len - 1) {                                    // buffer.rs:209-210, 201-204
      v_next = a.values[(size_t)a.lvi * N + e];
      nnt_next = a.last_dones[e] ? 0.0f : 1.0f;
    } else {                                                   // buffer.rs:212-213, 199
      v_next = a.values[(size_t)(t_top + 1) * N + e];
      nnt_next = nnt_of(done_word(a, (size_t)(t_top + 1), (unsigned)e), (unsigned)e);
    }
    float A = 0.0f, P = 1.0f;
#pragma unroll
    for (int u = 0; u < L; ++u) {
      if (u < steps) {
        const float x1 = v_next * nnt_next;
        const float x2 = x1 * a.gamma;
        const float x3 = dl[u] + x2;
        const float delta = x3 - vv[u];                        // (rewards + nextvalues*nextnonterminal*gamma) - values
        dl[u] = delta;
        if (nnt_next != 0.0f) mask |= 1ull << u;
        P = P * (nnt_next * a.gl);
        const float y1 = nnt_next * A;
        const float y2 = y1 * a.gl;
        A = delta + y2;
        v_next = vv[u];
        const uint32_t word = lane < 32 ? (uint32_t)__builtin_amdgcn_readlane((int)dw0, u) : (uint32_t)__builtin_amdgcn_readlane((int)dw1, u);
        nnt_next = ((word >> (lane & 31u)) & 1u) ? 0.0f : 1.0f;
      }
    }
    sP[w][lane] = P; sQ[w][lane] = A;
  } else {
    sP[w][lane] = 1.0f; sQ[w][lane] = 0.0f;                    // beyond t = 0: the identity map, feeds nobody
  }
  __syncthreads();

  // ---- carries: wave 0 publishes the super-chunk's map; ALL waves fetch the maps of the earlier super-chunks (entry i by
  // wave i mod W: one round trip for the whole list instead of one per batch); wave 0 applies them in order
  if (w == 0) {
    float Pa = sP[0][lane], Qa = sQ[0][lane];
#pragma unroll
    for (int k = 1; k < W; ++k) {                              // f_k after f_{k-1} ... after f_0, fixed order
      const float p = sP[k][lane], q = sQ[k][lane];
      Qa = q + p * Qa;
      Pa = p * Pa;
    }
    unsigned long long *mine = o.words + ((size_t)s * N + e) * 2;
    if (live) {
      __hip_atomic_store(mine, op_pack(o.serial, Pa), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(mine + 1, op_pack(o.serial, Qa), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  {
    bool gave_up = false;
    for (unsigned i = w; i < s; i += W) {
      const unsigned long long *src = o.words + ((size_t)i * N + e) * 2;
      unsigned long long wp, wq;
      unsigned spins = 0;
      for (;;) {
        wp = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        wq = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool ready = (unsigned)(wp >> 32) == o.serial && (unsigned)(wq >> 32) == o.serial;
        if (__all(ready)) break;                               // lanes past N read env N - 1's words: published too
        if (++spins > (1u << 20)) { gave_up = true; break; }   // seconds: something is badly wrong; drain the grid
        __builtin_amdgcn_s_sleep(8);
      }
      if (gave_up) break;
      lbP[i][lane] = __uint_as_float((unsigned)wp);
      lbQ[i][lane] = __uint_as_float((unsigned)wq);
    }
    if (gave_up && lane == 0) atomicExch(o.err, 1u);
  }
  __syncthreads();
  if (w == 0) {
    float Y = 0.0f;                                            // buffer.rs:196 for the super-chunk that holds the last steps
    for (unsigned i = 0; i < s; ++i) Y = lbQ[i][lane] + lbP[i][lane] * Y;
    sY[0][lane] = Y;
#pragma unroll
    for (int k = 0; k + 1 < W; ++k) {
      Y = sQ[k][lane] + sP[k][lane] * Y;
      sY[k + 1][lane] = Y;
    }
  }
  __syncthreads();

  // ---- phase B: exact replay from the carry (the sequential kernel's operations, in its order)
  if (steps > 0 && live) {
    float A = sY[w][lane];
#pragma unroll
    for (int u = 0; u < L; ++u) {
      if (u < steps) {
        const float nnt = ((mask >> u) & 1ull) ? 1.0f : 0.0f;
        const float y1 = nnt * A;
        const float y2 = y1 * a.gl;
        A = dl[u] + y2;                                        // lastgaelam = delta + nextnonterminal*lastgaelam*(gamma*lambda)
        const float ret = A + vv[u];                           // buffer.rs:222
        const float adv = ret - vv[u];                         // data.rs:62
        const size_t i = (size_t)(t_top - u) * N + e;
        if (!PACK || a.flat) {
          if (a.nt) { __builtin_nontemporal_store(ret, a.returns + i); __builtin_nontemporal_store(adv, a.adv + i); }
          else { a.returns[i] = ret; a.adv[i] = adv; }
        }
        if (PACK) store_rec_scan(a.rec, i, vv[u], ret);
      }
    }
  }
}

// Scan halves from the flat streams (only needed after a PPO_XP_GAE_NO_PACK returns()).
__global__ __launch_bounds__(256) void pack_records_kernel(const float *__restrict__ values,
                                                           const float *__restrict__ returns,
                                                           Rec *__restrict__ rec, size_t M) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (size_t)gridDim.x * blockDim.x)
    store_rec_scan(rec, i, values[i], returns[i]);
}

// Done bits of rows [row0, row0 + gridDim.y) from the f32 dones stream (0.0 / 1.0, buffer.rs:95): one wave per 64
// envs, ballot -> two words.  grid = (ceil(N / 256), rows), block = 256.
__global__ __launch_bounds__(256) void pack_done_bits_kernel(const float *__restrict__ dones, uint32_t *__restrict__ bits,
                                                             unsigned N, unsigned W, size_t row0) {
  const size_t t = row0 + blockIdx.y;
  const unsigned e = blockIdx.x * 256u + threadIdx.x, e0 = e & ~63u;
  const bool d = e < N && dones[t * N + e] != 0.0f;
  const unsigned long long m = __ballot(d);
  if ((threadIdx.x & 63u) == 0 && e0 < N) {
    bits[t * W + (e0 >> 5)] = (uint32_t)m;
    if ((e0 >> 5) + 1 < W) bits[t * W + (e0 >> 5) + 1] = (uint32_t)(m >> 32);
  }
}

// Static halves {nlp, act32} of transitions [i0, i0 + count): launched by the insert entry points right
// after the rows have been copied into the buffer (buffer.rs:93-113), so they are always in step with it.
__global__ __launch_bounds__(256) void pack_static_kernel(const float *__restrict__ neglogps,
                                                          const int64_t *__restrict__ actions,
                                                          Rec *__restrict__ rec, size_t i0, size_t count) {
  for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < count; j += (size_t)gridDim.x * blockDim.x)
    store_rec_static(rec, i0 + j, neglogps[i0 + j], actions[i0 + j]);
}

// The flat returns / adv' arrays from the records, for callers that ask for them after a packing
// scan that skipped them (ppo_xp_views, ppo_xp_returns with a host buffer): ret is stored in the
// record as it was computed, adv' = fl(ret - v) is the very operation of data.rs:62.
__global__ __launch_bounds__(256) void unpack_flat_kernel(const Rec *__restrict__ rec, float *__restrict__ returns,
                                                          float *__restrict__ adv, size_t M) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (size_t)gridDim.x * blockDim.x) {
    float v, ret;
    load_rec_scan(rec, i, v, ret);
    returns[i] = ret;
    adv[i] = ret - v;
  }
}

}  // namespace ppoxp
