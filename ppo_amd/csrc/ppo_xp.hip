// libppo_xp.so — C ABI (include/ppo_xp.h) over the HIP kernels of the experience path.
// gfx950 only; there is no CPU fallback anywhere in this file.
//
// Host-side state mirrors struct ExperienceBuffer (ppo/src/common/buffer.rs:23-39) and
// ExperienceBatcher / ExperienceBatcherIterator (ppo/src/model/data.rs:14-19,105-109).
#include "../../include/ppo_xp.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "kernels_gae.h"
#include "kernels_gather.h"
#ifdef PPO_XP_EXPERIMENTS   // losing experiments kept as evidence: built only into libppo_xp_experiments.so (make experiments)
#include "kernels_partition.h"
#endif
#include "shuffle.h"
#include "device_shuffle.h"
#include "errors.h"

using namespace ppoxp;

// ---------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------
static thread_local std::string g_err;

int ppoxp_fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
#define fail ppoxp_fail

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess)                                                                      \
      return fail(_e == hipErrorOutOfMemory ? PPO_XP_ENOMEM : PPO_XP_EIO, "%s failed: %s (%s:%d)", \
                  #expr, hipGetErrorString(_e), __FILE__, __LINE__);                           \
  } while (0)

// ---------------------------------------------------------------------------------------
// RCCL, bound lazily (the library has no link-time dependency on librccl)
// ---------------------------------------------------------------------------------------
namespace {
struct RcclUniqueId { char internal[128]; };
struct RcclApi {
  void *lib = nullptr;
  int (*GetUniqueId)(RcclUniqueId *) = nullptr;
  int (*CommInitRank)(void **, int, RcclUniqueId, int) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  int (*CommCount)(void *, int *) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;

int rccl_load() {
  if (g_rccl.lib) return 0;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void *lib = nullptr;
  for (const char *n : names) { lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (lib) break; }
  if (!lib) return fail(PPO_XP_EIO, "dlopen(librccl) failed: %s", dlerror());
  RcclApi a;
  a.lib = lib;
  a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
  a.CommInitRank = (decltype(a.CommInitRank))dlsym(lib, "ncclCommInitRank");
  a.CommDestroy = (decltype(a.CommDestroy))dlsym(lib, "ncclCommDestroy");
  a.CommCount = (decltype(a.CommCount))dlsym(lib, "ncclCommCount");
  a.AllReduce = (decltype(a.AllReduce))dlsym(lib, "ncclAllReduce");
  a.GetErrorString = (decltype(a.GetErrorString))dlsym(lib, "ncclGetErrorString");
  if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.CommCount || !a.AllReduce || !a.GetErrorString)
    return fail(PPO_XP_EIO, "librccl lacks an expected ncclXxx symbol");
  g_rccl = a;
  return 0;
}
constexpr int kNcclFloat64 = 8, kNcclSum = 0;  // rccl.h: ncclFloat64 = 8, ncclSum = 0
}  // namespace

// ---------------------------------------------------------------------------------------
// Static scratch of ppo_xp_returns_noalloc: the chunk maps (P, Q) and the chunk carries of the
// associative scan for callers that cannot allocate (HIP graph capture, serving loops).  One copy of
// the array lives on every GPU the library runs on, for the life of the process (12 MiB each).
// In host code the NAME `scratch` is only a stand-in for the symbol: the device address is asked
// for once per device with hipGetSymbolAddress (that device current) and kept in g_scratch_ptr;
// every kernel argument below is that pointer, never the name.
// ---------------------------------------------------------------------------------------
constexpr size_t kScratchPairs = size_t(1) << 20;           // (chunk, env) pairs one call may use
constexpr size_t kScratchOnePassWords = size_t(1) << 19;    // tagged 64-bit words of the one-pass scan: (super-chunk, env) pairs x 2
// [0]: P  [1]: Q  [2]: carry  (three-kernel scan, floats) | one-pass scan: tagged words, then its ticket counter and error flag
static __device__ __align__(256) float scratch[3 * kScratchPairs + 2 * kScratchOnePassWords + 64];
namespace {
constexpr int kMaxDevices = 64;
float *g_scratch_ptr[kMaxDevices] = {};
std::mutex g_scratch_mu;

// Call with `device` current.
int scratch_for_device(int device, float **out) {
  if (device < 0 || device >= kMaxDevices) return ppoxp_fail(PPO_XP_EINVAL, "device %d is beyond the %d devices the static scratch is tracked for", device, kMaxDevices);
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (!g_scratch_ptr[device]) {
    float *ptr = nullptr;
    hipError_t e = hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr)
      return ppoxp_fail(PPO_XP_EIO, "hipGetSymbolAddress(scratch) failed on device %d: %s", device, hipGetErrorString(e));
    g_scratch_ptr[device] = ptr;
  }
  *out = g_scratch_ptr[device];
  return 0;
}
}  // namespace

// ---------------------------------------------------------------------------------------
// handle
// ---------------------------------------------------------------------------------------
constexpr size_t kStageSlots = 8, kStageSlotBytes = size_t(256) << 10;   // staging of small host rows (ppo_xp_add)

struct ProfRec { int kid; hipEvent_t a, b; };

struct ppo_xp {
  int device = 0;
  size_t capacity = 0, N = 0, D = 0;
  float gamma = 0.99f, lambda = 0.95f;
  // ring state — buffer.rs:24-25,38
  size_t counter = 0, last_values_idx = 0;
  // device streams — buffer.rs:31-36
  float *obs = nullptr, *rewards = nullptr, *values = nullptr, *dones = nullptr, *neglogps = nullptr;
  int64_t *actions = nullptr;
  // GAE outputs
  float *returns = nullptr, *adv = nullptr;
  Rec *rec = nullptr;
  uint32_t *done_bits = nullptr;   // [capacity, ceil(N/32)] the dones stream as bits, for the scans
  // last_dones of ppo_xp_returns: a ring of 8 page-locked host slots of N bytes that the scan kernels read IN
  // PLACE (mapped host memory: 64 bytes per wave, once): no copy command and its two queue gaps per call.  A slot
  // is reused 8 calls later; one event per 4 calls (two alternate) tells when those scans have run.
  uint8_t *last_dones_pin = nullptr, *last_dones_pin_dev = nullptr;
  hipEvent_t pin_ev[2] = {nullptr, nullptr}; bool pin_used[2] = {false, false};
  // small host rows (ppo_xp_add) are staged in page-locked, device-visible slots and inserted by ONE kernel that reads
  // them over the bus: no copy commands, no synchronisation.  Slots of block b (4 slots) are reused after its event.
  uint8_t *stage_pin = nullptr, *stage_pin_dev = nullptr; unsigned stage_next = 0;
  hipEvent_t stage_ev[2] = {nullptr, nullptr}; bool stage_used[2] = {false, false};
  hipEvent_t perm_ev = nullptr; bool perm_ev_used = false;       // marks the end of the last upload from perm_host
  // split path with overlap: the observation rows are gathered on a side stream while the handle's stream runs the
  // scalar streams, the statistics (and the exchange) and the standardisation; joined before the pass is handed out
  hipStream_t side = nullptr; hipEvent_t side_fork = nullptr, side_join = nullptr; bool side_pending = false;
  unsigned pin_next = 0;
  // overlapped host hand-off (ppo_xp_pass_next_host_pipelined): minibatch k+1 travels to the caller's second slot on a
  // copy stream of the handle while the caller consumes k
  hipStream_t xfer = nullptr; hipEvent_t xfer_fork = nullptr, xfer_done = nullptr; bool xfer_used = false;
  bool hp_valid = false; size_t hp_k = 0; const void *hp_obs = nullptr; uint64_t hp_pass = 0;   // the prefetch in flight
  uint64_t pass_serial = 0;
  const uint8_t *last_dones_cur = nullptr;   // device address of the slot of the current call
  float *carry = nullptr, *sumP = nullptr, *sumQ = nullptr;
  size_t scan_cap = 0;       // elements allocated in carry/sumP/sumQ
  unsigned long long *op_words = nullptr; size_t op_cap = 0;   // one-pass scan: tagged maps [S][N][2] + ticket + error flag behind them
  unsigned *op_err_pending = nullptr;                           // error flag of a one-pass launch nobody has looked at yet
  bool have_returns = false, have_rec = false;
  bool have_flat = false;    // the flat returns / adv arrays hold the current GAE results (else only the records do)
  size_t ret_len = 0;        // len at the time of returns()
  // pass state — data.rs:105-109
  uint32_t *perm = nullptr;  // [capacity*N] owned copy (host-drawn permutations)
  const uint32_t *perm_cur = nullptr;  // permutation of the current pass: perm, or a borrowed device array
  float *obs_out = nullptr, *val_out = nullptr, *nlp_out = nullptr, *ret_out = nullptr, *adv_out = nullptr;
  int64_t *act_out = nullptr;
  size_t slab_rows = 0;
  double2 *partials = nullptr; size_t partials_cap = 0;
  double *sums = nullptr; float *stats = nullptr; size_t sums_cap = 0;
  size_t pass_B = 0, pass_K = 0, pass_M = 0, pass_iter = 0;
  bool pass_active = false, pass_pending_finish = false;
  int pass_kind = 0;         // 0: scalar_gather_kernel + obs_gather_kernel, 1: fused_gather_kernel, 2: fused_gather_loop_kernel
  bool streaming = false;    // minibatches are gathered one at a time by pass_next into two alternating B-row slabs
  bool pass_streamed = false, pass_fusable = false;   // how the current pass was begun
  unsigned pass_iter_rows = 1;                        // row groups per workgroup of the fused gather of the current pass
  // keyed-bijection pass (ppo_xp_pass_begin_bijection): perm(j) comes from `bij`; the index array h->perm is only
  // written when a kernel that reads one runs (split path) or when the caller asks for it
  Bijection bij{}; bool pass_bij = false, bij_inline = false, bij_perm_written = false;
  std::vector<uint32_t> perm_host;
  // Fisher-Yates of a coming pass drawn by a background thread (ppo_xp_shuffle_prefetch)
  std::thread pf_thread; std::vector<uint32_t> pf_perm; uint64_t pf_seed = 0; size_t pf_M = 0; bool pf_valid = false;
  void *shuffle_ws = nullptr; size_t shuffle_ws_bytes = 0;   // device permutation generator workspace
#ifdef PPO_XP_EXPERIMENTS
  void *part_ws = nullptr; size_t part_ws_bytes = 0;         // partitioned scalar path: counters, pairs, entries
#endif
  double *moments = nullptr;                                  // [kMomentBlocks][4] partials of explained_variance
  // exchange
  int xmode = 0; void *comm = nullptr;
  // stream / profiling
  hipStream_t stream = nullptr; bool own_stream = false;
  unsigned prof = 0;   // bit k set: record an event pair around every launch of kernel id k
  bool no_alloc_call = false;   // inside ppo_xp_returns_noalloc: nothing may allocate, events included
  std::vector<ProfRec> prof_recs;
  std::vector<hipEvent_t> ev_pool;
  // tunables
  long gather_nt = 1, gather_block = 64, gae_unroll = 0, carry_waves = 16, gae_nt = 1, rec_nt = 0, out_nt = 1, gather_xcd = 0, summary_nt = 0, gae_waves = 0, fold_tiles = 256, gae_flat = 0, fused = -1, fused_block = 512, scalar_path = 0, overlap = 0, bij_mode = 1, gae_onepass = -1, op_l = 32, op_w = 16, fused_iter = 0;
};

namespace {

size_t buf_len(const ppo_xp *h) { return h->counter >= h->capacity ? h->capacity : h->counter; }  // buffer.rs:123-129

struct ProfScope {
  ppo_xp *h; int kid; hipEvent_t a = nullptr, b = nullptr;
  ProfScope(ppo_xp *h_, int kid_) : h(h_), kid(kid_) {
    if (!((h->prof >> kid) & 1u)) return;
    a = take(); b = take();
    if (a && b) (void)hipEventRecord(a, h->stream);
  }
  ~ProfScope() {
    if (!a || !b) return;
    (void)hipEventRecord(b, h->stream);
    h->prof_recs.push_back({kid, a, b});
  }
  hipEvent_t take() {
    if (!h->ev_pool.empty()) { hipEvent_t e = h->ev_pool.back(); h->ev_pool.pop_back(); return e; }
    if (h->no_alloc_call) return nullptr;   // no pooled event left: this launch goes untimed
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
  }
};

template <typename T>
int dev_alloc(T **p, size_t count, bool zero) {
  *p = nullptr;
  size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
  HIP_TRY(hipMalloc((void **)p, bytes));
  if (zero) {
    // hipMemset runs on the NULL stream, which a non-blocking handle stream does not wait for:
    // finish it here, or it could land after a kernel of the handle has already written the buffer
    HIP_TRY(hipMemset(*p, 0, bytes));
    HIP_TRY(hipDeviceSynchronize());
  }
  return 0;
}

template <typename T>
void dev_free(T *&p) { if (p) { (void)hipFree(p); p = nullptr; } }

int launch_check(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(PPO_XP_EIO, "launch of %s failed: %s", what, hipGetErrorString(e));
  return 0;
}

// ---- GAE launches --------------------------------------------------------------------------
template <bool PACK>
int launch_gae_scan(ppo_xp *h, const GaeArgs &a, unsigned chunks) {
  dim3 grid((unsigned)((h->N + 63) / 64), chunks), block(64);
  ProfScope ps(h, PPO_XP_K_GAE);
  // time steps in flight per wave.  Round 1 (the scan also carried neglogps / actions into the records): 8 at TGT,
  // 16 for launches of at most one wave per CU.  Since the scan reads only r, v, d, 16 is never slower
  // (TGT 38 us vs 44 with 8; C2 20.7 vs 24.6; C4 fix-up equal) — profiles/r2/tune_gae_r2.log
  const long unroll = h->gae_unroll ? h->gae_unroll : 16;
  switch (unroll) {
    case 4: hipLaunchKernelGGL((gae_scan_kernel<PACK, 4>), grid, block, 0, h->stream, a); break;
    case 16: hipLaunchKernelGGL((gae_scan_kernel<PACK, 16>), grid, block, 0, h->stream, a); break;
    case 32: hipLaunchKernelGGL((gae_scan_kernel<PACK, 32>), grid, block, 0, h->stream, a); break;
    default: hipLaunchKernelGGL((gae_scan_kernel<PACK, 8>), grid, block, 0, h->stream, a); break;
  }
  return launch_check("gae_scan_kernel");
}

// K1w: W waves per env group, S = 8 steps per wave and super-chunk.  W = 0: one wave (gae_scan_kernel).
unsigned pick_gae_waves(const ppo_xp *h, size_t len) {
  if (h->gae_waves > 0) return (unsigned)h->gae_waves == 1 ? 0u : (unsigned)h->gae_waves;
  // measured (profiles/r1/tune_gae_waves.log): C2 (64 env groups) 23.2 us with one wave per group,
  // 16.9 / 13.5 / 12.5 / 13.8 us with 2 / 4 / 8 / 16; TGT (1024 groups) 70.5 us with one wave, slower with more
  const size_t groups = (h->N + 63) / 64;
  unsigned w = 1;
  // r2, scan without the record's static half: C3 (256 groups) 394 / 280 / 176 us with 1 / 2 / 4 waves, TGT (1024
  // groups) 38 us with one wave and 16 steps in flight, 38 with two: aim at >= 1024 waves
  while (w < 8 && groups * w < 1024) w <<= 1;
  while (w > 1 && (size_t)(w / 2) * 8 >= len) w >>= 1;   // no more waves than slices of 8 steps
  return w > 1 ? w : 0;
}

template <bool PACK>
int launch_gae_wide(ppo_xp *h, const GaeArgs &a, unsigned W) {
  dim3 grid((unsigned)((h->N + 63) / 64));
  ProfScope ps(h, PPO_XP_K_GAE);
  switch (W) {
    case 2: hipLaunchKernelGGL((gae_wide_kernel<PACK, 2, 8>), grid, dim3(128), 0, h->stream, a); break;
    case 4: hipLaunchKernelGGL((gae_wide_kernel<PACK, 4, 8>), grid, dim3(256), 0, h->stream, a); break;
    case 8: hipLaunchKernelGGL((gae_wide_kernel<PACK, 8, 8>), grid, dim3(512), 0, h->stream, a); break;
    default:   // 1024-thread workgroups have 128 VGPRs per lane: the packing variant holds 4 steps per wave there
      if (PACK) hipLaunchKernelGGL((gae_wide_kernel<PACK, 16, 4>), grid, dim3(1024), 0, h->stream, a);
      else hipLaunchKernelGGL((gae_wide_kernel<PACK, 16, 8>), grid, dim3(1024), 0, h->stream, a);
      break;
  }
  return launch_check("gae_wide_kernel");
}

// One-pass associative scan (gae_onepass_kernel): super-chunks of kOpW chunks of kOpL steps.
constexpr unsigned kOpMaxSupers = kOnePassMaxSupers;   // every workgroup applies all earlier super-chunk maps itself: the list lives in LDS
std::atomic<unsigned> g_onepass_serial{0};

unsigned onepass_supers(const ppo_xp *h, size_t len) {
  const size_t L = (size_t)h->op_l, W = (size_t)h->op_w;
  return (unsigned)((((len + L - 1) / L) + W - 1) / W);
}

template <bool PACK>
bool launch_onepass_shape(ppo_xp *h, const OnePassArgs &o, unsigned blocks) {
  const long L = h->op_l, W = h->op_w;
#define PPOXP_OP(LL, WW) if (L == LL && W == WW) { hipLaunchKernelGGL((gae_onepass_kernel<PACK, LL, WW>), dim3(blocks), dim3(64 * WW), 0, h->stream, o); return true; }
  PPOXP_OP(64, 8) PPOXP_OP(64, 4) PPOXP_OP(64, 2) PPOXP_OP(32, 8) PPOXP_OP(32, 4) PPOXP_OP(32, 16) PPOXP_OP(16, 16) PPOXP_OP(16, 8)
#undef PPOXP_OP
  return false;
}

// PPO_XP_GAE_ASSOC without a forced chunk length takes the one-pass kernel when the list of earlier super-chunks every
// workgroup walks stays short (len <= 64 * 8 * 64 = 32 768 steps); "gae_onepass" 0 / 1 forces the choice.
bool onepass_wanted(const ppo_xp *h, int algo, size_t len) {
  if (h->gae_onepass == 0) return false;
  if ((((unsigned)algo >> 16) & 0xffffu) != 0) return false;          // the caller chose a chunk length: three-kernel scan
  return onepass_supers(h, len) <= kOpMaxSupers;
}

int launch_gae_onepass(ppo_xp *h, GaeArgs a, bool pack, float *static_scratch) {
  const size_t N = h->N;
  const unsigned S = onepass_supers(h, a.len), groups = (unsigned)((N + 63) / 64);
  const size_t words = (size_t)S * N * 2;
  OnePassArgs o{};
  if (static_scratch) {
    if (words > kScratchOnePassWords)
      return fail(PPO_XP_EINVAL, "returns_noalloc: %u super-chunks x %zu envs need %zu tagged words, the static scratch holds %zu",
                  S, N, words, kScratchOnePassWords);
    o.words = reinterpret_cast<unsigned long long *>(static_scratch + 3 * kScratchPairs);
  } else {
    if (words > h->op_cap) {
      if (h->op_words) { HIP_TRY(hipStreamSynchronize(h->stream)); dev_free(h->op_words); h->op_cap = 0; }
      if (int rc = dev_alloc(&h->op_words, words + 8, true)) return rc;     // zeroed: tag 0 is never used
      h->op_cap = words;
    }
    o.words = h->op_words;
  }
  const size_t cap = static_scratch ? kScratchOnePassWords : h->op_cap;
  o.ticket = o.words + cap;
  o.err = reinterpret_cast<unsigned *>(o.words + cap + 1);
  unsigned serial = ++g_onepass_serial;
  if (serial == 0) serial = ++g_onepass_serial;
  o.a = a; o.serial = serial; o.S = S; o.groups = groups;
  o.a.chunk = (unsigned)h->op_l;
  if ((size_t)S * groups > 0x7fffffffull) return fail(PPO_XP_EINVAL, "returns: too many workgroups for the one-pass scan");
  ProfScope ps(h, PPO_XP_K_GAE);
  if (!(pack ? launch_onepass_shape<true>(h, o, S * groups) : launch_onepass_shape<false>(h, o, S * groups)))
    return fail(PPO_XP_EINVAL, "one-pass scan: no kernel for op_l = %ld, op_w = %ld", h->op_l, h->op_w);
  if (int rc = launch_check("gae_onepass_kernel")) return rc;
  h->op_err_pending = o.err;
  return 0;
}

// A look-back of the one-pass scan that gave up (never seen; the bound exists so that a bug cannot hang the GPU) is
// reported by the next call that waits for the stream anyway.
int onepass_check(ppo_xp *h) {
  if (!h->op_err_pending) return 0;
  unsigned flag = 0;
  unsigned *p = h->op_err_pending;
  h->op_err_pending = nullptr;
  HIP_TRY(hipMemcpyAsync(&flag, p, sizeof(flag), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (flag) {
    (void)hipMemsetAsync(p, 0, sizeof(flag), h->stream);
    return fail(PPO_XP_EIO, "one-pass GAE scan: a workgroup gave up waiting for an earlier chunk's carry map; results are invalid");
  }
  return 0;
}

// Chunk length and count of the associative scan for `len` time steps (bits 16..31 of algo force the length).
void gae_chunking(const ppo_xp *h, int algo, size_t len, unsigned &L_out, unsigned &chunks_out) {
  unsigned L = ((unsigned)algo >> 16) & 0xffffu;
  const unsigned groups = (unsigned)((h->N + 63) / 64);
  if (L == 0) {
    // enough (chunk, env-group) waves to fill 256 CUs several times over, chunks >= 32 steps
    // (C4, 16 env groups: 128 chunks of 128 steps = 114 us, 256 x 64 = 122 us, 64 x 256 = 123 us)
    unsigned want = std::max(1u, 2048u / groups);
    unsigned c = (unsigned)std::min<size_t>(want, (len + 31) / 32);
    c = std::max(c, 1u);
    L = (unsigned)((len + c - 1) / c);
  }
  L = std::max(1u, std::min<unsigned>(L, (unsigned)len));
  unsigned chunks = (unsigned)((len + L - 1) / L);
  if (chunks > 65535u) { L = (unsigned)((len + 65534) / 65535); chunks = (unsigned)((len + L - 1) / L); }
  L_out = L; chunks_out = chunks;
}

// static_scratch != nullptr: the chunk maps and carries of the associative scan live there (3 x kScratchPairs
// floats) and nothing is allocated; else in the handle's own buffers, grown on demand.
int run_gae(ppo_xp *h, int algo, bool want_flat, float *static_scratch = nullptr) {
  const size_t len = buf_len(h);
  const size_t N = h->N;
  if (len == 0 || N == 0) return 0;
  const bool pack = !(algo & PPO_XP_GAE_NO_PACK);
  const int kind = algo & 0xff;
  GaeArgs a{};
  a.rewards = h->rewards; a.values = h->values; a.done_bits = h->done_bits; a.W = (unsigned)((N + 31) / 32);
  a.last_dones = h->last_dones_cur; a.returns = h->returns; a.adv = h->adv;
  a.rec = h->rec; a.N = (unsigned)N; a.len = (unsigned)len; a.lvi = (unsigned)h->last_values_idx;
  a.gamma = h->gamma; a.nt = (int)h->gae_nt;
  a.flat = (!pack || want_flat || h->gae_flat) ? 1 : 0;
  a.gl = h->gamma * h->lambda;  // one fp32 product, computed first (buffer.rs:219)
  unsigned chunks = 1;
  a.chunk = (unsigned)len;
  if (kind == PPO_XP_GAE_ASSOC && onepass_wanted(h, algo, len)) {
    if (int rc = launch_gae_onepass(h, a, pack, static_scratch)) return rc;
    h->have_returns = true; h->have_rec = pack; h->have_flat = a.flat != 0; h->ret_len = len;
    return 0;
  }
  if (kind == PPO_XP_GAE_ASSOC) gae_chunking(h, algo, len, a.chunk, chunks);
  if (chunks > 1) {
    const size_t need = (size_t)chunks * N;
    if (static_scratch) {
      if (need > kScratchPairs)
        return fail(PPO_XP_EINVAL, "returns_noalloc: %u chunks x %zu envs = %zu chunk maps, the static scratch holds %zu "
                    "(use fewer, longer chunks with PPO_XP_GAE_CHUNK, or ppo_xp_returns)", chunks, N, need, kScratchPairs);
    } else if (need > h->scan_cap) {
      dev_free(h->carry); dev_free(h->sumP); dev_free(h->sumQ); h->scan_cap = 0;
      if (int rc = dev_alloc(&h->carry, need, false)) return rc;
      if (int rc = dev_alloc(&h->sumP, need, false)) return rc;
      if (int rc = dev_alloc(&h->sumQ, need, false)) return rc;
      h->scan_cap = need;
    }
    a.carry = h->carry; a.sumP = h->sumP; a.sumQ = h->sumQ;
    if (static_scratch) { a.sumP = static_scratch; a.sumQ = static_scratch + kScratchPairs; a.carry = static_scratch + 2 * kScratchPairs; }
    {
      dim3 grid((unsigned)((N + 63) / 64), chunks - 1), block(64);
      ProfScope ps(h, PPO_XP_K_GAE_SUMMARY);
      GaeArgs sa = a;
      sa.nt = (int)h->summary_nt;   // default policy: the fix-up pass reads the same bytes again
      hipLaunchKernelGGL((gae_scan_kernel<false, 8, true>), grid, block, 0, h->stream, sa);
      if (int rc = launch_check("gae_scan_kernel<summary>")) return rc;
    }
    {
      dim3 grid((unsigned)((N + 63) / 64));
      ProfScope ps(h, PPO_XP_K_GAE_CARRY);
      if (h->carry_waves == 4)
        hipLaunchKernelGGL((gae_carry_scan_kernel<4>), grid, dim3(256), 0, h->stream, a.sumP, a.sumQ, const_cast<float *>(a.carry), (unsigned)N, chunks);
      else if (h->carry_waves == 8)
        hipLaunchKernelGGL((gae_carry_scan_kernel<8>), grid, dim3(512), 0, h->stream, a.sumP, a.sumQ, const_cast<float *>(a.carry), (unsigned)N, chunks);
      else
        hipLaunchKernelGGL((gae_carry_scan_kernel<16>), grid, dim3(1024), 0, h->stream, a.sumP, a.sumQ, const_cast<float *>(a.carry), (unsigned)N, chunks);
      if (int rc = launch_check("gae_carry_scan_kernel")) return rc;
    }
  }
  const unsigned W = chunks == 1 ? pick_gae_waves(h, len) : 0u;
  int rc;
  if (W) rc = pack ? launch_gae_wide<true>(h, a, W) : launch_gae_wide<false>(h, a, W);
  else rc = pack ? launch_gae_scan<true>(h, a, chunks) : launch_gae_scan<false>(h, a, chunks);
  if (rc) return rc;
  h->have_returns = true;
  h->have_rec = pack;
  h->have_flat = a.flat != 0;
  h->ret_len = len;
  return 0;
}

// returns / adv as flat arrays, for the callers that read them (views, host copies): a packing scan
// leaves them only inside the records.
int ensure_flat(ppo_xp *h) {
  if (!h->have_returns || h->have_flat) return 0;
  const size_t M = h->ret_len * h->N;
  if (M) {
    HIP_TRY(hipSetDevice(h->device));
    const unsigned blocks = (unsigned)std::min<size_t>((M + 255) / 256, 1u << 20);
    hipLaunchKernelGGL(unpack_flat_kernel, dim3(blocks), dim3(256), 0, h->stream, h->rec, h->returns, h->adv, M);
    if (int rc = launch_check("unpack_flat_kernel")) return rc;
  }
  h->have_flat = true;
  return 0;
}

// ---- gather launches -----------------------------------------------------------------------
template <int V, int BS, bool NT>
void launch_obs_v(ppo_xp *h, const ObsGatherArgs &g, unsigned blocks, bool pow2, bool tail) {
  dim3 grid(blocks), block(BS);
  if (pow2) {
    if (tail) hipLaunchKernelGGL((obs_gather_kernel<V, BS, NT, true, true>), grid, block, 0, h->stream, g);
    else hipLaunchKernelGGL((obs_gather_kernel<V, BS, NT, true, false>), grid, block, 0, h->stream, g);
  } else {
    if (tail) hipLaunchKernelGGL((obs_gather_kernel<V, BS, NT, false, true>), grid, block, 0, h->stream, g);
    else hipLaunchKernelGGL((obs_gather_kernel<V, BS, NT, false, false>), grid, block, 0, h->stream, g);
  }
}

// Kn minibatches starting with minibatch k0, written to obs_out (rows [0, Kn*B) of it).
int launch_obs_gather(ppo_xp *h, size_t K, size_t B, size_t M, size_t k0, size_t Kn, float *obs_out) {
  const unsigned D = (unsigned)h->D;
  const int V = (D % 4 == 0) ? 4 : (D % 2 == 0) ? 2 : 1;
  ObsGatherArgs g{};
  g.obs = h->obs; g.perm = h->perm_cur; g.obs_out = obs_out; g.M = M; g.B = B; g.row_base = k0 * B;
  g.tail_from = (K - 1) * B; g.tail_shift = K * B - M;
  g.Q = D / V;
  g.total = Kn * B * g.Q;
  const bool pow2 = (g.Q & (g.Q - 1)) == 0;
  g.qshift = 0;
  if (pow2) while ((1u << g.qshift) < g.Q) ++g.qshift;
  const bool tail = K * B != M;
  const unsigned bs = h->gather_block == 256 ? 256u : 64u;
  const size_t blocks = (g.total + bs - 1) / bs;
  if (blocks > 0x7fffffffull) return fail(PPO_XP_EINVAL, "pass too large for one launch (%zu workgroups)", blocks);
  g.nblocks = (unsigned)blocks; g.xcd = (unsigned)h->gather_xcd;
  ProfScope ps(h, PPO_XP_K_GATHER);
  const bool nt = h->gather_nt != 0;
  if (V == 4) {
    if (bs == 64) nt ? launch_obs_v<4, 64, true>(h, g, (unsigned)blocks, pow2, tail) : launch_obs_v<4, 64, false>(h, g, (unsigned)blocks, pow2, tail);
    else nt ? launch_obs_v<4, 256, true>(h, g, (unsigned)blocks, pow2, tail) : launch_obs_v<4, 256, false>(h, g, (unsigned)blocks, pow2, tail);
  } else if (V == 2) {
    if (bs == 64) launch_obs_v<2, 64, true>(h, g, (unsigned)blocks, pow2, tail); else launch_obs_v<2, 256, true>(h, g, (unsigned)blocks, pow2, tail);
  } else {
    if (bs == 64) launch_obs_v<1, 64, true>(h, g, (unsigned)blocks, pow2, tail); else launch_obs_v<1, 256, true>(h, g, (unsigned)blocks, pow2, tail);
  }
  return launch_check("obs_gather_kernel");
}

#ifdef PPO_XP_EXPERIMENTS
// The scalar streams through the partitioned path (kernels_partition.h): six streaming launches instead of one
// kernel that fetches a 128-byte line per record.  Same outputs, same tile partials.
int launch_partitioned_scalars(ppo_xp *h, const ScalarGatherArgs &g, size_t K, size_t B, size_t M) {
  const unsigned nS = (unsigned)((M + (size_t(1) << kSrcShift) - 1) >> kSrcShift);
  const unsigned nD = (unsigned)((M + (size_t(1) << kDstShift) - 1) >> kDstShift);
  const size_t counters = (size_t)(nS + nD) * sizeof(uint32_t);
  const size_t off_pairs = (counters + 255) & ~size_t(255);
  const size_t off_entries = off_pairs + ((M * sizeof(uint2) + 255) & ~size_t(255));
  const size_t need = off_entries + M * 5 * sizeof(uint32_t);
  if (need > h->part_ws_bytes) {
    if (h->part_ws) { HIP_TRY(hipStreamSynchronize(h->stream)); (void)hipFree(h->part_ws); h->part_ws = nullptr; h->part_ws_bytes = 0; }
    HIP_TRY(hipMalloc(&h->part_ws, need));
    h->part_ws_bytes = need;
  }
  char *ws = (char *)h->part_ws;
  PartArgs a{};
  a.perm = g.perm; a.M = M; a.rec = g.rec; a.actions = g.actions;
  a.src_count = (uint32_t *)ws; a.dst_cursor = a.src_count + nS;
  a.pairs = (uint2 *)(ws + off_pairs); a.entries = (uint32_t *)(ws + off_entries);
  a.act_out = g.act_out; a.val_out = g.val_out; a.nlp_out = g.nlp_out; a.ret_out = g.ret_out; a.adv_out = g.adv_out;
  a.nS = nS; a.nD = nD;
  ProfScope ps(h, PPO_XP_K_SCALARS);
  HIP_TRY(hipMemsetAsync(a.src_count, 0, (size_t)nS * sizeof(uint32_t), h->stream));
  const unsigned chunks = (unsigned)((M + kPartChunk - 1) / kPartChunk);
  hipLaunchKernelGGL(part_histogram_kernel, dim3(chunks), dim3(256), 0, h->stream, a);
  hipLaunchKernelGGL(part_scan_kernel, dim3(1), dim3(1024), 0, h->stream, a);
  hipLaunchKernelGGL(part_pairs_kernel, dim3(chunks), dim3(256), 0, h->stream, a);
  {
    const unsigned per_window = (1u << kSrcShift) / kPartChunk;
    const unsigned windows = (chunks + per_window - 1) / per_window;
    hipLaunchKernelGGL(part_records_kernel, dim3(((windows + 7) / 8) * 8 * per_window), dim3(1024), 0, h->stream, a);
  }
  {
    const size_t c256 = (M + 255) / 256;
    const unsigned per_window = (1u << kDstShift) / 256u;
    const size_t windows = (c256 + per_window - 1) / per_window;
    hipLaunchKernelGGL(part_scatter_kernel, dim3((unsigned)(((windows + 7) / 8) * 8 * per_window)), dim3(256), 0, h->stream, a);
  }
  hipLaunchKernelGGL(part_tile_stats_kernel, dim3((unsigned)(K * g.tiles_per_mb)), dim3(256), 0, h->stream, g.adv_out, g.partials, B,
                     g.tiles_per_mb);
  return launch_check("partitioned scalar path");
}
#endif

int ensure_pass_buffers(ppo_xp *h, size_t K, size_t B, size_t tile_rows) {
  // sized for this batch size at FULL capacity, so that a pass over a partly filled ring followed by
  // one over the full ring does not move the slabs (views of the earlier pass stay valid memory);
  // a larger batch size later still re-allocates: views die with the next pass_begin (include/ppo_xp.h)
  const size_t Mcap = h->capacity * h->N;
  // streaming: two alternating B-row slots instead of the K*B-row slab (no second copy of the observations)
  const size_t rows = h->streaming ? 2 * B : std::max(K * B, B <= Mcap ? ((Mcap + B - 1) / B) * B : (size_t)0);
  if (rows > h->slab_rows) {
    HIP_TRY(hipStreamSynchronize(h->stream));   // kernels of the previous pass may still write the old slabs
    if (h->xfer) HIP_TRY(hipStreamSynchronize(h->xfer));   // ... and host copies may still read them
    dev_free(h->obs_out); dev_free(h->act_out); dev_free(h->val_out); dev_free(h->nlp_out);
    dev_free(h->ret_out); dev_free(h->adv_out); h->slab_rows = 0;
    if (int rc = dev_alloc(&h->obs_out, rows * h->D, false)) return rc;
    if (int rc = dev_alloc(&h->act_out, rows, false)) return rc;
    if (int rc = dev_alloc(&h->val_out, rows, false)) return rc;
    if (int rc = dev_alloc(&h->nlp_out, rows, false)) return rc;
    if (int rc = dev_alloc(&h->ret_out, rows, false)) return rc;
    if (int rc = dev_alloc(&h->adv_out, rows, false)) return rc;
    h->slab_rows = rows;
  }
  const size_t tiles = std::max(K, B <= Mcap ? (Mcap + B - 1) / B : (size_t)0) * ((B + tile_rows - 1) / tile_rows);
  // + the second-level partials of reduce_partials_kernel (one per 256 tiles, rounded up per minibatch)
  const size_t need = tiles + (tiles + 255) / 256 + std::max(K, (size_t)1);
  if (need > h->partials_cap) {
    dev_free(h->partials); h->partials_cap = 0;
    if (int rc = dev_alloc(&h->partials, need, false)) return rc;
    h->partials_cap = need;
  }
  if (K > h->sums_cap) {
    dev_free(h->sums); dev_free(h->stats); h->sums_cap = 0;
    if (int rc = dev_alloc(&h->sums, 3 * K, true)) return rc;
    if (int rc = dev_alloc(&h->stats, 2 * K, true)) return rc;
    h->sums_cap = K;
  }
  return 0;
}

// Where a launch range puts its results: the slabs of the whole pass, or (streaming) one of the two B-row slots.
struct PassOut {
  float *obs, *val, *nlp, *ret, *adv;
  int64_t *act;
  double *sums;    // [Kn][3]
  float *stats;    // [Kn][2]
};

PassOut pass_out_at(const ppo_xp *h, size_t row, size_t k) {
  PassOut o;
  o.obs = h->obs_out + row * h->D; o.val = h->val_out + row; o.nlp = h->nlp_out + row; o.ret = h->ret_out + row;
  o.adv = h->adv_out + row; o.act = h->act_out + row; o.sums = h->sums + 3 * k; o.stats = h->stats + 2 * k;
  return o;
}

// Standardises Kn minibatches in place.  fold = true: no finalize_stats launch preceded; the normalize workgroups
// reduce the partials themselves.
int normalize_range(ppo_xp *h, size_t Kn, const PassOut &o, bool fold = false, unsigned fold_stat_tiles = 0,
                    const double2 *fold_partials = nullptr) {
  const size_t B = h->pass_B;
  if (Kn == 0) return 0;
  const unsigned vpt = (fold && B >= 65536) ? 4u : 1u;      // the fold is paid per workgroup: fewer, fatter ones for big minibatches
  const size_t nbx = (B + 1024 * vpt - 1) / (1024 * vpt);
  if (Kn * nbx > 0x7fffffffull) return fail(PPO_XP_EINVAL, "pass too large for the normalize launch");
  const unsigned stat_tiles = fold_stat_tiles ? fold_stat_tiles : (unsigned)((B + 255) / 256);
  ProfScope ps(h, PPO_XP_K_NORMALIZE);
  const double2 *fp = fold_partials ? fold_partials : h->partials;
  const dim3 grid((unsigned)(Kn * nbx));
  if (fold && vpt == 4)
    hipLaunchKernelGGL((normalize_kernel<true, 4>), grid, dim3(256), 0, h->stream, o.adv, o.sums, fp, stat_tiles, o.stats, B, (unsigned)nbx);
  else if (fold)
    hipLaunchKernelGGL((normalize_kernel<true, 1>), grid, dim3(256), 0, h->stream, o.adv, o.sums, fp, stat_tiles, o.stats, B, (unsigned)nbx);
  else
    hipLaunchKernelGGL((normalize_kernel<false, 1>), grid, dim3(256), 0, h->stream, o.adv, o.sums, fp, stat_tiles, o.stats, B, (unsigned)nbx);
  return launch_check("normalize_kernel");
}

// ppo_xp_pass_finish and the tail of a sharded pass: the whole pass, statistics already in h->sums.
// Observation rows on the side stream: ordered after everything the handle's stream holds now, joined by side_wait.
int launch_obs_side(ppo_xp *h, size_t K, size_t B, size_t M, size_t k0, size_t Kn, float *obs_out) {
  if (!h->side) {
    HIP_TRY(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&h->side_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&h->side_join, hipEventDisableTiming));
  }
  HIP_TRY(hipEventRecord(h->side_fork, h->stream));
  HIP_TRY(hipStreamWaitEvent(h->side, h->side_fork, 0));
  hipStream_t mainstream = h->stream;
  h->stream = h->side;
  const int rc = launch_obs_gather(h, K, B, M, k0, Kn, obs_out);
  h->stream = mainstream;
  if (rc) return rc;
  HIP_TRY(hipEventRecord(h->side_join, h->side));
  h->side_pending = true;
  return 0;
}

int side_wait(ppo_xp *h) {
  if (!h->side_pending) return 0;
  h->side_pending = false;
  HIP_TRY(hipStreamWaitEvent(h->stream, h->side_join, 0));
  return 0;
}

int pass_finish_impl(ppo_xp *h) {
  h->pass_pending_finish = false;
  if (int rc = normalize_range(h, h->pass_K, pass_out_at(h, 0, 0))) return rc;
  return side_wait(h);
}

template <int BS>
void launch_fused(ppo_xp *h, const FusedGatherArgs &f, dim3 grid, bool tail, int bij) {
  if (bij == 1) {
    if (tail) hipLaunchKernelGGL((fused_gather_kernel<BS, true, true, 1>), grid, dim3(BS), 0, h->stream, f);
    else hipLaunchKernelGGL((fused_gather_kernel<BS, true, false, 1>), grid, dim3(BS), 0, h->stream, f);
  } else if (bij == 2) {
    if (tail) hipLaunchKernelGGL((fused_gather_kernel<BS, true, true, 2>), grid, dim3(BS), 0, h->stream, f);
    else hipLaunchKernelGGL((fused_gather_kernel<BS, true, false, 2>), grid, dim3(BS), 0, h->stream, f);
  } else {
    if (tail) hipLaunchKernelGGL((fused_gather_kernel<BS, true, true, 0>), grid, dim3(BS), 0, h->stream, f);
    else hipLaunchKernelGGL((fused_gather_kernel<BS, true, false, 0>), grid, dim3(BS), 0, h->stream, f);
  }
}

void launch_fused_loop(ppo_xp *h, const FusedGatherArgs &f, dim3 grid, bool tail, unsigned iter) {
#define PPOXP_FL(IT) if (iter == IT) { if (tail) hipLaunchKernelGGL((fused_gather_loop_kernel<512, IT, true, true>), grid, dim3(512), 0, h->stream, f); \
                                       else hipLaunchKernelGGL((fused_gather_loop_kernel<512, IT, true, false>), grid, dim3(512), 0, h->stream, f); return; }
  PPOXP_FL(2) PPOXP_FL(4)
#undef PPOXP_FL
}

// The keyed bijection of the current pass written out into h->perm (once per pass, on the handle's stream).
int materialize_bijection(ppo_xp *h) {
  if (!h->pass_bij || h->bij_perm_written) return 0;
  const size_t M = h->bij.M;
  if (M) {
    hipLaunchKernelGGL(bijection_perm_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, h->stream, h->bij, h->perm);
    if (int rc = launch_check("bijection_perm_kernel")) return rc;
  }
  h->bij_perm_written = true;
  return 0;
}

// The one cross-GPU step of the path: all-reduce(sum) of the K (count, sum, sum of squares) triples, in place, on the
// handle's stream (exchange mode 1: the library's own RCCL call).
int exchange_sums(ppo_xp *h, size_t K) {
  ProfScope ps(h, PPO_XP_K_EXCHANGE);
  const int e = g_rccl.AllReduce(h->sums, h->sums, 3 * K, kNcclFloat64, kNcclSum, h->comm, h->stream);
  if (e != 0) return fail(PPO_XP_EIO, "ncclAllReduce failed: %s", g_rccl.GetErrorString(e));
  return 0;
}

// Gathers and standardises minibatches [k0, k0 + Kn) of the current pass into `o`.  Whole pass: k0 = 0, Kn = K;
// streaming: one minibatch at a time.  Exchange modes (sharded ranks) only occur with the whole pass.
int gather_range(ppo_xp *h, size_t k0, size_t Kn, const PassOut &o) {
  const size_t K = h->pass_K, B = h->pass_B, M = h->pass_M;
  const bool whole = k0 == 0 && Kn == K;
  const unsigned Dq = (unsigned)h->D / 4;
  const unsigned FBq = (unsigned)h->fused_block;
  const unsigned rpi = (Dq && FBq % Dq == 0) ? FBq / Dq : 0u;
  const unsigned iter = h->pass_iter_rows;                     // row groups per workgroup (1 unless the looping kernel runs)
  const unsigned rpw = rpi * iter;
  if (h->pass_fusable) {
    FusedGatherArgs f{};
    f.rpw = rpw;
    f.o.obs = h->obs; f.o.perm = h->perm_cur; f.o.obs_out = o.obs; f.o.M = M; f.o.B = B; f.o.Q = Dq; f.o.row_base = k0 * B;
    f.o.total = Kn * B * Dq; f.o.qshift = 0;
    f.o.tail_from = (K - 1) * B; f.o.tail_shift = K * B - M;
    while ((1u << f.o.qshift) < Dq) ++f.o.qshift;
    f.s.rec = h->rec; f.s.actions = h->actions; f.s.perm = h->perm_cur; f.s.act_out = o.act;
    f.s.val_out = o.val; f.s.nlp_out = o.nlp; f.s.ret_out = o.ret; f.s.adv_out = o.adv;
    f.s.partials = h->partials; f.s.M = M; f.s.B = B; f.s.tiles_per_mb = (unsigned)(B / rpw); f.s.k0 = (unsigned)k0;
    const size_t blocks = Kn * (B / rpw);
    if (blocks > 0x7fffffffull) return fail(PPO_XP_EINVAL, "pass too large for one launch");
    {
      ProfScope ps(h, PPO_XP_K_GATHER);
      const bool tail = K * B != M;
      const dim3 grid((unsigned)blocks);
      const int bij = h->bij_inline ? (int)h->bij_mode : 0;
      f.bij = h->bij;
      if (iter > 1) launch_fused_loop(h, f, grid, tail, iter);
      else if (FBq == 256) launch_fused<256>(h, f, grid, tail, bij);
      else launch_fused<512>(h, f, grid, tail, bij);
      if (int rc = launch_check("fused_gather_kernel")) return rc;
    }
    // statistics: few tiles per minibatch -> every standardise workgroup folds them itself; many -> one
    // launch folds groups of 256 first (fixed order either way: deterministic)
    const unsigned tiles = f.s.tiles_per_mb;
    if (h->xmode == 0 && tiles <= 1024u) return normalize_range(h, Kn, o, true, tiles);
    const unsigned G = (tiles + 255u) / 256u;
    double2 *partials2 = h->partials + Kn * (size_t)tiles;
    const bool two_level = tiles > 1024u;
    if (two_level) {
      ProfScope ps(h, PPO_XP_K_FINALIZE);
      hipLaunchKernelGGL(reduce_partials_kernel, dim3(G, (unsigned)Kn), dim3(256), 0, h->stream, h->partials, partials2, tiles);
      if (int rc = launch_check("reduce_partials_kernel")) return rc;
    }
    if (h->xmode == 0) return normalize_range(h, Kn, o, true, G, partials2);
    // sharded: this rank's (count, sum, sumsq) per minibatch must exist as numbers, they cross the ranks
    {
      ProfScope ps(h, PPO_XP_K_FINALIZE);
      hipLaunchKernelGGL(finalize_stats_kernel, dim3((unsigned)Kn), dim3(256), 0, h->stream, two_level ? partials2 : h->partials,
                         o.sums, two_level ? G : tiles, B);
      if (int rc = launch_check("finalize_stats_kernel")) return rc;
    }
    if (h->xmode == 2) { h->pass_pending_finish = true; return 0; }
    if (int rc = exchange_sums(h, K)) return rc;
    return pass_finish_impl(h);
  }
  // scalar streams + statistics first (the records were just written by the GAE kernel and are
  // the most likely bytes to still sit in the Infinity Cache), observation rows after
  ScalarGatherArgs g{};
  g.rec = h->rec; g.actions = h->actions; g.perm = h->perm_cur; g.act_out = o.act;
  g.val_out = o.val; g.nlp_out = o.nlp; g.ret_out = o.ret; g.adv_out = o.adv;
  g.partials = h->partials; g.M = M; g.B = B; g.tiles_per_mb = (unsigned)((B + 255) / 256); g.k0 = (unsigned)k0;
  g.rec_nt = (int)h->rec_nt; g.out_nt = (int)h->out_nt;
#ifdef PPO_XP_EXPERIMENTS
  const bool partitioned = whole && h->scalar_path == 1 && K * B == M && M >= kPartChunk &&
                           ((M + (size_t(1) << kSrcShift) - 1) >> kSrcShift) <= kMaxBins && ((M + (size_t(1) << kDstShift) - 1) >> kDstShift) <= kMaxBins;
#else
  const bool partitioned = false;
  (void)whole;
#endif
  const bool overlap = h->overlap != 0;
  if (overlap)
    if (int rc = launch_obs_side(h, K, B, M, k0, Kn, o.obs)) return rc;
  if (partitioned) {
#ifdef PPO_XP_EXPERIMENTS
    if (int rc = launch_partitioned_scalars(h, g, K, B, M)) return rc;
#endif
  } else {
    ProfScope ps(h, PPO_XP_K_SCALARS);
    hipLaunchKernelGGL(scalar_gather_kernel, dim3((unsigned)(Kn * g.tiles_per_mb)), dim3(256), 0, h->stream, g);
    if (int rc = launch_check("scalar_gather_kernel")) return rc;
  }
  // single GPU and few tiles per minibatch (C2: 64): the normalize workgroups fold the partials
  // themselves and the finalize launch (4.5 us, a tenth of such a pass) is skipped
  const bool fold = h->xmode == 0 && g.tiles_per_mb <= (unsigned)h->fold_tiles;
  if (!fold) {
    ProfScope ps(h, PPO_XP_K_FINALIZE);
    hipLaunchKernelGGL(finalize_stats_kernel, dim3((unsigned)Kn), dim3(256), 0, h->stream, h->partials, o.sums,
                       g.tiles_per_mb, B);
    if (int rc = launch_check("finalize_stats_kernel")) return rc;
  }
  if (h->xmode == 0) {
    // single GPU: standardise while the raw advantages are still in L2 / Infinity Cache, rows last
    if (int rc = normalize_range(h, Kn, o, fold)) return rc;
    return overlap ? side_wait(h) : launch_obs_gather(h, K, B, M, k0, Kn, o.obs);
  }
  // sharded: the observation rows do not depend on the statistics; enqueue them before the
  // exchange so that the all-reduce's host-side latency hides behind ~0.7 ms of GPU work
  if (!overlap)
    if (int rc = launch_obs_gather(h, K, B, M, k0, Kn, o.obs)) return rc;
  if (h->xmode == 2) { h->pass_pending_finish = true; return 0; }
  if (int rc = exchange_sums(h, K)) return rc;
  return pass_finish_impl(h);
}

// perm already in h->perm_cur (device).
int pass_begin_body(ppo_xp *h, size_t B) {
  const size_t len = buf_len(h), M = len * h->N;
  if (!h->have_returns || h->ret_len != len)
    return fail(PPO_XP_ESTATE, "pass_begin: call ppo_xp_returns() first (ExperienceBatcher::new takes the returns, data.rs:27)");
  if (int rc = side_wait(h)) return rc;                      // rows of an abandoned pass (begun, never finished) on the side stream
  if (h->xfer_used) HIP_TRY(hipStreamWaitEvent(h->stream, h->xfer_done, 0));   // host copies of the previous pass still read the slabs
  h->hp_valid = false; ++h->pass_serial;
  h->pass_B = B; h->pass_M = M; h->pass_iter = 0; h->pass_active = true; h->pass_pending_finish = false;
  h->pass_K = (B > M) ? 0 : (M + B - 1) / B;                 // data.rs:117-126
  h->pass_streamed = h->streaming;
  const size_t K = h->pass_K;
  if (K == 0) return 0;
  if (K * ((B + 255) / 256) > 0x7fffffffull) return fail(PPO_XP_EINVAL, "pass too large for one launch");
  // rows per statistics tile (= per workgroup of whichever kernel gathers the scalar streams).
  // The fused kernel (rows + scalar streams in one workgroup) is taken when a workgroup covers 8..64
  // whole rows and the records fit the Infinity Cache beside the row stream (measured: TGT 134 MB wins 75 us,
  // C3 0.5 GB and C5-on-one-GPU 2.1 GB lose); "fused" = 0 / 1 forces the choice, -1 decides by size.
  const unsigned Dq = (unsigned)h->D / 4;
  const unsigned FBq = (unsigned)h->fused_block;
  const unsigned rpw = (Dq && FBq % Dq == 0) ? FBq / Dq : 0u;
  const bool fits_mall = rec_halves(M) * sizeof(Rec) <= (size_t(160) << 20);
  // A keyed-bijection pass has no index stream: with 256-byte rows (32+ rows per workgroup) the fused kernel then wins or
  // ties at every size measured - per-GPU shard of C5 at 8 ranks (268 MB of records) 1 793 us against 1 909, at 4 / 2 / 1
  // ranks +0.8 / +0.3 / 0 % - while with 1-KiB rows (C3) it still loses 9 % (profiles/r3/ab_bijection_sizes.log).
  const bool bij_fused = h->pass_bij && h->bij_mode != 0 && rpw >= 32;
  const bool fusable = h->fused != 0 && h->D % 4 == 0 && (Dq & (Dq - 1)) == 0 && rpw >= 8 && rpw <= 64 &&
                       B % rpw == 0 && K <= 65535 && (h->fused == 1 || fits_mall || bij_fused);
  // looping fused kernel (bijection passes, 512-thread workgroups): ITER row groups per workgroup, when they tile the minibatch
  // Measured (profiles/r3/ab_fused_row_groups.log): 4 row groups per workgroup win 1.1 - 1.6 % while the records sit in the
  // Infinity Cache and the pass is long enough to matter (M = 2 M ... 8 M transitions: 222 against 225 us, 430 against 436,
  // 847 against 857 at TGT), lose 1 us at C2 (0.5 M) and 3.4 % on the per-GPU shard of C5 (268 MB of records); 8 groups need
  // 208 registers and lose 50 %.  "fused_iter" 0 = that rule, 1 / 2 / 4 force the choice.
  unsigned iter = 1;
  const long want_iter = h->fused_iter ? h->fused_iter : ((fits_mall && M >= (size_t(1) << 20)) ? 4 : 1);
  if (fusable && h->pass_bij && h->bij_mode == 1 && FBq == 512 && want_iter > 1 && rpw * want_iter <= 256 &&
      B % (rpw * (size_t)want_iter) == 0)
    iter = (unsigned)want_iter;
  h->pass_iter_rows = iter;
  const size_t tile_rows = fusable ? (size_t)rpw * iter : 256;
  if (int rc = ensure_pass_buffers(h, K, B, tile_rows)) return rc;
  if (!h->have_rec) {
    ProfScope ps(h, PPO_XP_K_PACK);
    unsigned blocks = (unsigned)std::min<size_t>((M + 255) / 256, 0x7fffffffull);
    hipLaunchKernelGGL(pack_records_kernel, dim3(blocks), dim3(256), 0, h->stream, h->values, h->returns, h->rec, M);
    if (int rc = launch_check("pack_records_kernel")) return rc;
    h->have_rec = true;
  }
  h->pass_kind = fusable ? (iter > 1 ? 2 : 1) : 0;
  h->pass_fusable = fusable;
  // keyed bijection: the fused kernel evaluates it in registers; the kernels that read an index array get it written
  // out first (one small launch, inside the pass)
  h->bij_inline = h->pass_bij && fusable && h->bij_mode != 0;
  if (h->pass_bij && !h->bij_inline) {
    if (int rc = materialize_bijection(h)) return rc;
    h->perm_cur = h->perm;
  }
  if (h->pass_streamed) return 0;            // ppo_xp_pass_next gathers minibatch k when it is asked for
  return gather_range(h, 0, K, pass_out_at(h, 0, 0));
}

// A pass that failed half-way (allocation, launch, all-reduce) must not hand out minibatches.
int pass_begin_impl(ppo_xp *h, size_t B) {
  h->pass_active = false; h->pass_pending_finish = false;
  const int rc = pass_begin_body(h, B);
  if (rc) { h->pass_active = false; h->pass_pending_finish = false; }
  return rc;
}

}  // namespace

// ---------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------
extern "C" {

const char *ppo_xp_last_error(void) { return g_err.c_str(); }
#ifdef PPO_XP_EXPERIMENTS
const char *ppo_xp_version(void) { return "ppo_xp 0.1 (gfx950) +experiments"; }
#else
const char *ppo_xp_version(void) { return "ppo_xp 0.1 (gfx950)"; }
#endif

const char *ppo_xp_kernel_name(int k) {
  static const char *names[PPO_XP_K_COUNT] = {"gae_scan", "gae_chunk_summary", "gae_carry_scan", "pack_records",
                                              "gather", "finalize_stats", "normalize", "scalar_gather", "exchange"};
  return (k >= 0 && k < PPO_XP_K_COUNT) ? names[k] : "?";
}

int ppo_xp_create(ppo_xp **out, int device, size_t capacity, size_t num_envs, size_t obs_size, float gamma,
                  float lambda) {
  if (!out) return fail(PPO_XP_EINVAL, "create: out is NULL");
  *out = nullptr;
  if (capacity == 0 || num_envs == 0 || obs_size == 0)
    return fail(PPO_XP_EINVAL, "create: capacity, num_envs and obs_size must be > 0");
  if (capacity * num_envs > 0xFFFFFFFFull)
    return fail(PPO_XP_EINVAL, "create: capacity*num_envs must be < 2^32 (u32 permutation indices)");
  if (num_envs > 0x7FFFFFFFull || obs_size > 0x00FFFFFFull || capacity > 0x7FFFFFFFull)
    return fail(PPO_XP_EINVAL, "create: dimension too large");
  int ndev = 0;
  hipError_t de = hipGetDeviceCount(&ndev);
  if (de != hipSuccess || ndev <= 0)
    return fail(PPO_XP_ENODEV, "create: no HIP device (hipGetDeviceCount: %s, %d devices); this library has no CPU path",
                hipGetErrorString(de), ndev);
  if (device < 0 || device >= ndev) return fail(PPO_XP_ENODEV, "create: device %d of %d", device, ndev);
  HIP_TRY(hipSetDevice(device));
  ppo_xp *h = new (std::nothrow) ppo_xp();
  if (!h) return fail(PPO_XP_ENOMEM, "create: host allocation failed");
  h->device = device; h->capacity = capacity; h->N = num_envs; h->D = obs_size;
  h->gamma = gamma; h->lambda = lambda;
  const size_t cn = capacity * num_envs;
  int rc = 0;
  // Array::zeros — buffer.rs:60-65
  if (!rc) rc = dev_alloc(&h->obs, cn * obs_size, true);
  if (!rc) rc = dev_alloc(&h->rewards, cn, true);
  if (!rc) rc = dev_alloc(&h->actions, cn, true);
  if (!rc) rc = dev_alloc(&h->values, cn, true);
  if (!rc) rc = dev_alloc(&h->dones, cn, true);
  if (!rc) rc = dev_alloc(&h->neglogps, cn, true);
  if (!rc) rc = dev_alloc(&h->returns, cn, true);
  if (!rc) rc = dev_alloc(&h->adv, cn, true);
  if (!rc) rc = dev_alloc(&h->rec, rec_halves(cn), true);   // zeros = the static halves of the zero-filled streams
  if (!rc) rc = dev_alloc(&h->done_bits, capacity * ((num_envs + 31) / 32), true);
  if (!rc) rc = dev_alloc(&h->perm, cn, true);
  if (!rc && hipHostMalloc((void **)&h->last_dones_pin, 8 * std::max<size_t>(num_envs, 1), hipHostMallocMapped) != hipSuccess)
    rc = fail(PPO_XP_ENOMEM, "create: hipHostMalloc failed");
  if (!rc && hipHostGetDevicePointer((void **)&h->last_dones_pin_dev, h->last_dones_pin, 0) != hipSuccess)
    rc = fail(PPO_XP_EIO, "create: hipHostGetDevicePointer failed");
  if (!rc && hipHostMalloc((void **)&h->stage_pin, kStageSlots * kStageSlotBytes, hipHostMallocMapped) != hipSuccess)
    rc = fail(PPO_XP_ENOMEM, "create: hipHostMalloc failed");
  if (!rc && hipHostGetDevicePointer((void **)&h->stage_pin_dev, h->stage_pin, 0) != hipSuccess)
    rc = fail(PPO_XP_EIO, "create: hipHostGetDevicePointer failed");
  if (!rc && hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess)
    rc = fail(PPO_XP_EIO, "create: hipStreamCreate failed");
  if (!rc) h->own_stream = true;
  for (int i = 0; i < 2 && !rc; ++i)
    if (hipEventCreateWithFlags(&h->pin_ev[i], hipEventDisableTiming) != hipSuccess) rc = fail(PPO_XP_EIO, "create: hipEventCreate failed");
  for (int i = 0; i < 2 && !rc; ++i)
    if (hipEventCreateWithFlags(&h->stage_ev[i], hipEventDisableTiming) != hipSuccess) rc = fail(PPO_XP_EIO, "create: hipEventCreate failed");
  if (!rc && hipEventCreateWithFlags(&h->perm_ev, hipEventDisableTiming) != hipSuccess) rc = fail(PPO_XP_EIO, "create: hipEventCreate failed");
  if (!rc && hipDeviceSynchronize() != hipSuccess) rc = fail(PPO_XP_EIO, "create: device sync failed");
  // set-up of the allocation-free entry point: this device's address of the static scratch
  if (!rc) { float *sp = nullptr; rc = scratch_for_device(device, &sp); }
  if (rc) { ppo_xp_destroy(h); return rc; }
  *out = h;
  return 0;
}

void ppo_xp_destroy(ppo_xp *h) {
  if (!h) return;
  if (h->pf_thread.joinable()) h->pf_thread.join();
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->side) (void)hipStreamSynchronize(h->side);
  if (h->xfer) (void)hipStreamSynchronize(h->xfer);
  if (h->xfer_fork) (void)hipEventDestroy(h->xfer_fork);
  if (h->xfer_done) (void)hipEventDestroy(h->xfer_done);
  if (h->xfer) (void)hipStreamDestroy(h->xfer);
  for (auto &r : h->prof_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  for (auto e : h->ev_pool) (void)hipEventDestroy(e);
  for (auto e : h->pin_ev) if (e) (void)hipEventDestroy(e);
  for (auto e : h->stage_ev) if (e) (void)hipEventDestroy(e);
  if (h->perm_ev) (void)hipEventDestroy(h->perm_ev);
  if (h->side_fork) (void)hipEventDestroy(h->side_fork);
  if (h->side_join) (void)hipEventDestroy(h->side_join);
  if (h->side) (void)hipStreamDestroy(h->side);
  dev_free(h->obs); dev_free(h->rewards); dev_free(h->actions); dev_free(h->values); dev_free(h->dones);
  dev_free(h->neglogps); dev_free(h->returns); dev_free(h->adv); dev_free(h->rec); dev_free(h->done_bits); dev_free(h->perm);
  dev_free(h->carry); dev_free(h->sumP); dev_free(h->sumQ); dev_free(h->op_words);
  dev_free(h->obs_out); dev_free(h->act_out); dev_free(h->val_out); dev_free(h->nlp_out); dev_free(h->ret_out);
  dev_free(h->adv_out); dev_free(h->partials); dev_free(h->sums); dev_free(h->stats);
  dev_free(h->moments);
  if (h->shuffle_ws) { (void)hipFree(h->shuffle_ws); h->shuffle_ws = nullptr; }
#ifdef PPO_XP_EXPERIMENTS
  if (h->part_ws) { (void)hipFree(h->part_ws); h->part_ws = nullptr; }
#endif
  if (h->last_dones_pin) (void)hipHostFree(h->last_dones_pin);
  if (h->stage_pin) (void)hipHostFree(h->stage_pin);
  if (h->stream && h->own_stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

// The static halves {nlp, act32} of the record lines of rows [row, row + rows), from the rows just copied in.
static int pack_static_rows(ppo_xp *h, size_t row, size_t rows) {
  const size_t count = rows * h->N;
  if (!count) return 0;
  const unsigned blocks = (unsigned)std::min<size_t>((count + 255) / 256, 1u << 16);
  hipLaunchKernelGGL(pack_static_kernel, dim3(blocks), dim3(256), 0, h->stream, h->neglogps, h->actions, h->rec, row * h->N, count);
  if (int rc = launch_check("pack_static_kernel")) return rc;
  const unsigned W = (unsigned)((h->N + 31) / 32);
  for (size_t r0 = 0; r0 < rows; r0 += 65535) {     // gridDim.y limit
    const unsigned ny = (unsigned)std::min<size_t>(rows - r0, 65535);
    hipLaunchKernelGGL(pack_done_bits_kernel, dim3((unsigned)((h->N + 255) / 256), ny), dim3(256), 0, h->stream, h->dones,
                       h->done_bits, (unsigned)h->N, W, row + r0);
    if (int rc = launch_check("pack_done_bits_kernel")) return rc;
  }
  return 0;
}

static void advance_counter(ppo_xp *h, size_t idx) {
  h->last_values_idx = idx;                                     // buffer.rs:115
  h->counter += 1;                                              // buffer.rs:117
  if (h->counter >= h->capacity) h->counter = (h->counter % h->capacity) + h->capacity;  // buffer.rs:118-120
  h->have_returns = false; h->have_rec = false; h->have_flat = false; h->pass_active = false;
}

int ppo_xp_add_many(ppo_xp *h, const float *obs, const float *rewards, const size_t *actions, const float *vals,
                    const uint8_t *dones, const float *neglogps, size_t n, size_t steps) {
  if (!h) return fail(PPO_XP_EINVAL, "add: handle is NULL");
  if (n != h->N)
    return fail(PPO_XP_EINVAL, "add_experience: %zu envs given, buffer has %zu (reference panics, buffer.rs:97-112)", n, h->N);
  if (steps == 0) return 0;
  if (!obs || !rewards || !actions || !vals || !dones || !neglogps) return fail(PPO_XP_EINVAL, "add: NULL input");
  HIP_TRY(hipSetDevice(h->device));
  const size_t N = h->N, D = h->D, cap = h->capacity;
  // steps older than the last `cap` are overwritten before anyone can see them: only the ring
  // state they leave behind matters.
  size_t first = steps > cap ? steps - cap : 0;
  for (size_t s = 0; s < first; ++s) advance_counter(h, h->counter % cap);
  std::vector<float> donesf;
  size_t s = first;
  while (s < steps) {
    const size_t idx = h->counter % cap;                         // buffer.rs:93
    const size_t run = std::min(steps - s, cap - idx);           // contiguous rows idx .. idx+run
    donesf.resize(run * N);
    for (size_t i = 0; i < run * N; ++i) donesf[i] = (float)(uint8_t)(dones[s * N + i] != 0);  // buffer.rs:95
    HIP_TRY(hipMemcpyAsync(h->obs + idx * N * D, obs + s * N * D, run * N * D * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->rewards + idx * N, rewards + s * N, run * N * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->actions + idx * N, actions + s * N, run * N * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->values + idx * N, vals + s * N, run * N * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->dones + idx * N, donesf.data(), run * N * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->neglogps + idx * N, neglogps + s * N, run * N * sizeof(float), hipMemcpyHostToDevice, h->stream));
    if (int rc = pack_static_rows(h, idx, run)) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));                    // host buffers are the caller's again
    for (size_t r = 0; r < run; ++r) advance_counter(h, idx + r);
    s += run;
  }
  return 0;
}

// One inserted row by one kernel: sources are device-visible pointers (device memory or staged host slots).
static int launch_insert_row(ppo_xp *h, const float *obs, const float *rewards, const int64_t *actions, const float *vals,
                             const float *dones_f32, const uint8_t *dones_u8, const float *neglogps) {
  const size_t N = h->N, D = h->D, idx = h->counter % h->capacity;
  if (N > 0xffffff00ull) return fail(PPO_XP_EINVAL, "add: %zu envs do not fit one launch", N);
  InsertArgs a{};
  a.obs = obs; a.rewards = rewards; a.values = vals; a.dones = dones_f32; a.dones_u8 = dones_u8; a.neglogps = neglogps; a.actions = actions;
  a.obs_dst = h->obs + idx * N * D; a.rew_dst = h->rewards + idx * N; a.val_dst = h->values + idx * N;
  a.done_dst = h->dones + idx * N; a.nlp_dst = h->neglogps + idx * N; a.act_dst = h->actions + idx * N;
  a.rec = h->rec; a.W = (unsigned)((N + 31) / 32); a.bits_row = h->done_bits + idx * a.W;
  a.i0 = idx * N; a.N = (unsigned)N;
  // 16-byte elements when the row and both of its ends allow it (hipMalloc'ed bases are 256-byte aligned; the
  // buffer's row starts at idx * N * D floats, the caller's pointer is whatever it is)
  const size_t row = N * D;
  const bool v4 = row % 4 == 0 && ((uintptr_t)obs % 16) == 0 && ((uintptr_t)a.obs_dst % 16) == 0;
  a.obs_elems = v4 ? row / 4 : row;
  const size_t ob = (a.obs_elems + 255) / 256, sb = (N + 255) / 256;
  if (ob + sb > 0x7fffffffull) return fail(PPO_XP_EINVAL, "add: row too large for one launch (%zu workgroups)", ob + sb);
  a.obs_blocks = (unsigned)ob;
  if (ob + sb) {
    if (v4) hipLaunchKernelGGL((insert_row_kernel<4>), dim3((unsigned)(ob + sb)), dim3(256), 0, h->stream, a);
    else hipLaunchKernelGGL((insert_row_kernel<1>), dim3((unsigned)(ob + sb)), dim3(256), 0, h->stream, a);
    if (int rc = launch_check("insert_row_kernel")) return rc;
  }
  advance_counter(h, idx);
  return 0;
}

static size_t align16(size_t x) { return (x + 15) & ~size_t(15); }

int ppo_xp_add(ppo_xp *h, const float *obs, const float *rewards, const size_t *actions, const float *vals,
               const uint8_t *dones, const float *neglogps, size_t n) {
  if (!h) return fail(PPO_XP_EINVAL, "add: handle is NULL");
  const size_t N = h->N, D = h->D;
  // layout of one staged row: obs | rewards | values | neglogps | actions | dones, every part 16-byte aligned
  const size_t o_rew = align16(N * D * 4), o_val = o_rew + align16(N * 4), o_nlp = o_val + align16(N * 4),
               o_act = o_nlp + align16(N * 4), o_don = o_act + align16(N * 8), total = o_don + align16(N);
  if (n != N || total > kStageSlotBytes || !obs || !rewards || !actions || !vals || !dones || !neglogps)
    return ppo_xp_add_many(h, obs, rewards, actions, vals, dones, neglogps, n, 1);   // large rows (copy commands) and every error message
  HIP_TRY(hipSetDevice(h->device));
  const unsigned call = h->stage_next++;
  const unsigned slot = call % kStageSlots, blk = (call >> 2) & 1u;
  if ((call & 3u) == 0 && h->stage_used[blk]) HIP_TRY(hipEventSynchronize(h->stage_ev[blk]));
  uint8_t *s = h->stage_pin + slot * kStageSlotBytes;
  const uint8_t *d = h->stage_pin_dev + slot * kStageSlotBytes;
  std::memcpy(s, obs, N * D * 4);
  std::memcpy(s + o_rew, rewards, N * 4);
  std::memcpy(s + o_val, vals, N * 4);
  std::memcpy(s + o_nlp, neglogps, N * 4);
  std::memcpy(s + o_act, actions, N * 8);
  std::memcpy(s + o_don, dones, N);
  const int rc = launch_insert_row(h, (const float *)d, (const float *)(d + o_rew), (const int64_t *)(d + o_act), (const float *)(d + o_val),
                                   nullptr, d + o_don, (const float *)(d + o_nlp));
  if ((call & 3u) == 3u) {          // also after a failed launch: the slots of this block may still be read
    HIP_TRY(hipEventRecord(h->stage_ev[blk], h->stream));
    h->stage_used[blk] = true;
  }
  return rc;
}

int ppo_xp_add_device(ppo_xp *h, const float *obs, const float *rewards, const int64_t *actions, const float *vals,
                      const float *dones_f32, const float *neglogps, size_t n) {
  if (!h) return fail(PPO_XP_EINVAL, "add_device: handle is NULL");
  if (n != h->N) return fail(PPO_XP_EINVAL, "add_experience: %zu envs given, buffer has %zu", n, h->N);
  if (!obs || !rewards || !actions || !vals || !dones_f32 || !neglogps) return fail(PPO_XP_EINVAL, "add_device: NULL input");
  HIP_TRY(hipSetDevice(h->device));
  return launch_insert_row(h, obs, rewards, actions, vals, dones_f32, nullptr, neglogps);
}

void *ppo_xp_host_alloc(size_t bytes) {
  void *p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
    (void)fail(PPO_XP_ENOMEM, "host_alloc: hipHostMalloc(%zu) failed", bytes);
    return nullptr;
  }
  return p;
}

void ppo_xp_host_free(void *p) { if (p) (void)hipHostFree(p); }

void ppo_xp_reset_counter(ppo_xp *h) {                          // buffer.rs:131-133
  if (!h) return;
  h->counter = 0; h->have_returns = false; h->have_rec = false; h->have_flat = false; h->pass_active = false;
}
size_t ppo_xp_len(const ppo_xp *h) { return h ? buf_len(h) : 0; }
size_t ppo_xp_counter(const ppo_xp *h) { return h ? h->counter : 0; }
size_t ppo_xp_last_values_idx(const ppo_xp *h) { return h ? h->last_values_idx : 0; }

static int returns_impl(ppo_xp *h, const uint8_t *last_dones, size_t n, int algo, float *host_out, float *static_scratch);

int ppo_xp_returns(ppo_xp *h, const uint8_t *last_dones, size_t n, int algo, float *host_out) {
  return returns_impl(h, last_dones, n, algo, host_out, nullptr);
}

int ppo_xp_returns_noalloc(ppo_xp *h, const uint8_t *last_dones, size_t n, int algo, float *host_out) {
  if (!h) return fail(PPO_XP_EINVAL, "returns: handle is NULL");
  HIP_TRY(hipSetDevice(h->device));
  float *sp = nullptr;
  if (int rc = scratch_for_device(h->device, &sp)) return rc;   // looked up at create(); no allocation either way
  h->no_alloc_call = true;
  const int rc = returns_impl(h, last_dones, n, algo, host_out, sp);
  h->no_alloc_call = false;
  return rc;
}

static int returns_impl(ppo_xp *h, const uint8_t *last_dones, size_t n, int algo, float *host_out, float *static_scratch) {
  if (!h) return fail(PPO_XP_EINVAL, "returns: handle is NULL");
  if (n != h->N)
    return fail(PPO_XP_EINVAL, "returns: last_dones has %zu entries, buffer has %zu envs (reference panics, buffer.rs:194)", n, h->N);
  if (!last_dones) return fail(PPO_XP_EINVAL, "returns: last_dones is NULL");
  const int kind = algo & 0xff;
  if (kind != PPO_XP_GAE_SEQUENTIAL && kind != PPO_XP_GAE_ASSOC) return fail(PPO_XP_EINVAL, "returns: unknown algo %d", algo);
  HIP_TRY(hipSetDevice(h->device));
  if (static_scratch && kind == PPO_XP_GAE_ASSOC && buf_len(h)) {   // refuse BEFORE anything changes: the call then does nothing
    unsigned L = 0, chunks = 1;
    gae_chunking(h, algo, buf_len(h), L, chunks);
    if (onepass_wanted(h, algo, buf_len(h))) {
      const size_t words = (size_t)onepass_supers(h, buf_len(h)) * h->N * 2;
      if (words > kScratchOnePassWords)
        return fail(PPO_XP_EINVAL, "returns_noalloc: %u super-chunks x %zu envs need %zu tagged words, the static scratch holds %zu "
                    "(choose a chunk length with PPO_XP_GAE_CHUNK for the three-kernel scan, or ppo_xp_returns)",
                    onepass_supers(h, buf_len(h)), h->N, words, kScratchOnePassWords);
    } else if (chunks > 1 && (size_t)chunks * h->N > kScratchPairs)
      return fail(PPO_XP_EINVAL, "returns_noalloc: %u chunks x %zu envs = %zu chunk maps, the static scratch holds %zu "
                  "(use fewer, longer chunks with PPO_XP_GAE_CHUNK, or ppo_xp_returns)", chunks, h->N, (size_t)chunks * h->N, kScratchPairs);
  }
  // slot ring (see the handle): calls 4b .. 4b+3 form block b; the event of block b is recorded after its last
  // scan and waited for when block b+2 (which reuses its slots) starts
  const unsigned call = h->pin_next++;
  const unsigned slot = call & 7u, blk = (call >> 2) & 1u;
  if ((call & 3u) == 0 && h->pin_used[blk]) HIP_TRY(hipEventSynchronize(h->pin_ev[blk]));
  std::memcpy(h->last_dones_pin + (size_t)slot * n, last_dones, n);
  h->last_dones_cur = h->last_dones_pin_dev + (size_t)slot * n;
  h->pass_active = false;
  const int grc = run_gae(h, algo, host_out != nullptr, static_scratch);
  if ((call & 3u) == 3u) {          // also after a failed launch: the slots of this block may still be read
    HIP_TRY(hipEventRecord(h->pin_ev[blk], h->stream));
    h->pin_used[blk] = true;
  }
  if (grc) return grc;
  h->have_returns = true; h->ret_len = buf_len(h);
  if (buf_len(h) == 0) { h->have_rec = true; h->have_flat = true; }
  if (host_out) {
    const size_t M = buf_len(h) * h->N;
    if (M) HIP_TRY(hipMemcpyAsync(host_out, h->returns, M * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (int rc = onepass_check(h)) return rc;
  }
  return 0;
}

int ppo_xp_views(const ppo_xp *h, ppo_xp_views_t *out) {
  if (!h || !out) return fail(PPO_XP_EINVAL, "views: NULL argument");
  // the handle is opaque: handing out returns / advantages may first have to unpack them from the records
  if (int rc = ensure_flat(const_cast<ppo_xp *>(h))) return rc;
  out->num_steps = buf_len(h) * h->N; out->obs_size = h->D;
  out->observations = h->obs; out->actions = h->actions; out->values = h->values; out->neglogps = h->neglogps;
  out->rewards = h->rewards; out->dones = h->dones; out->returns = h->returns; out->advantages = h->adv;
  return 0;
}

int ppo_xp_pass_begin(ppo_xp *h, size_t batch_size, const uint32_t *perm, uint64_t seed) {
  if (!h) return fail(PPO_XP_EINVAL, "pass_begin: handle is NULL");
  if (batch_size == 0) return fail(PPO_XP_EINVAL, "pass_begin: batch_size 0 (the reference would iterate forever)");
  HIP_TRY(hipSetDevice(h->device));
  const size_t M = buf_len(h) * h->N;
  if (M) {
    // the previous upload may still be reading perm_host; the rest of the stream need not drain
    if (h->perm_ev_used) HIP_TRY(hipEventSynchronize(h->perm_ev));
    if (h->pf_thread.joinable()) h->pf_thread.join();
    const bool prefetched = !perm && h->pf_valid && h->pf_seed == seed && h->pf_M == M;
    h->pf_valid = false;
    if (prefetched) {
      h->perm_host.swap(h->pf_perm);                             // drawn while the previous pass ran
    } else {
      h->perm_host.resize(M);
      if (perm) {
        for (size_t i = 0; i < M; ++i)
          if (perm[i] >= M) return fail(PPO_XP_EINVAL, "pass_begin: perm[%zu] = %u is out of range (M = %zu)", i, perm[i], M);
        std::memcpy(h->perm_host.data(), perm, M * sizeof(uint32_t));
      } else {
        shuffle_indices(seed, h->perm_host.data(), M);           // data.rs:93-95
      }
    }
    HIP_TRY(hipMemcpyAsync(h->perm, h->perm_host.data(), M * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipEventRecord(h->perm_ev, h->stream));
    h->perm_ev_used = true;
  }
  h->perm_cur = h->perm; h->pass_bij = false;
  return pass_begin_impl(h, batch_size);
}

int ppo_xp_pass_begin_device(ppo_xp *h, size_t batch_size, const uint32_t *perm_dev) {
  if (!h) return fail(PPO_XP_EINVAL, "pass_begin: handle is NULL");
  if (batch_size == 0) return fail(PPO_XP_EINVAL, "pass_begin: batch_size 0 (the reference would iterate forever)");
  if (!perm_dev) return fail(PPO_XP_EINVAL, "pass_begin_device: perm is NULL");
  HIP_TRY(hipSetDevice(h->device));
  h->perm_cur = perm_dev;  // borrowed, not copied: see include/ppo_xp.h
  h->pass_bij = false;
  return pass_begin_impl(h, batch_size);
}

int ppo_xp_pass_begin_device_shuffle(ppo_xp *h, size_t batch_size, uint64_t seed) {
  if (!h) return fail(PPO_XP_EINVAL, "pass_begin: handle is NULL");
  if (batch_size == 0) return fail(PPO_XP_EINVAL, "pass_begin: batch_size 0 (the reference would iterate forever)");
  HIP_TRY(hipSetDevice(h->device));
  const size_t M = buf_len(h) * h->N;
  if (M) {
    const size_t need = device_shuffle_workspace_bytes(M);
    if (need > h->shuffle_ws_bytes) {
      if (h->shuffle_ws) { HIP_TRY(hipStreamSynchronize(h->stream)); (void)hipFree(h->shuffle_ws); h->shuffle_ws = nullptr; h->shuffle_ws_bytes = 0; }
      HIP_TRY(hipMalloc(&h->shuffle_ws, need));
      h->shuffle_ws_bytes = need;
    }
    int e = device_shuffle(seed, M, h->perm, h->shuffle_ws, h->shuffle_ws_bytes, h->stream);
    if (e != 0) return fail(PPO_XP_EIO, "device shuffle failed: %s", hipGetErrorString((hipError_t)e));
  }
  h->perm_cur = h->perm; h->pass_bij = false;
  return pass_begin_impl(h, batch_size);
}

int ppo_xp_pass_begin_bijection(ppo_xp *h, size_t batch_size, uint64_t seed) {
  if (!h) return fail(PPO_XP_EINVAL, "pass_begin: handle is NULL");
  if (batch_size == 0) return fail(PPO_XP_EINVAL, "pass_begin: batch_size 0 (the reference would iterate forever)");
  HIP_TRY(hipSetDevice(h->device));
  const size_t M = buf_len(h) * h->N;
  h->bij = bij_make(seed, M ? M : 1);
  h->pass_bij = true; h->bij_perm_written = false;
  h->perm_cur = nullptr;       // set to h->perm once the index array has been written
  return pass_begin_impl(h, batch_size);
}

// explained_variance (training.rs:88-99): 0 - var(returns - values) / var(returns), ddof 0; the
// reference's guard compares the target variance with -1.0 (never true) and is kept as it is.
int ppo_xp_explained_variance_sums(ppo_xp *h, double *sums5) {
  if (!h || !sums5) return fail(PPO_XP_EINVAL, "explained_variance: NULL argument");
  const size_t len = buf_len(h), M = len * h->N;
  if (!h->have_returns || h->ret_len != len) return fail(PPO_XP_ESTATE, "explained_variance: call ppo_xp_returns() first");
  HIP_TRY(hipSetDevice(h->device));
  constexpr unsigned kBlocks = 2048;
  if (!h->moments) { if (int rc = dev_alloc(&h->moments, 4 * (size_t)kBlocks, true)) return rc; }
  for (int k = 0; k < 5; ++k) sums5[k] = 0.0;
  if (M == 0) return 0;
  hipLaunchKernelGGL(moments_kernel, dim3(kBlocks), dim3(256), 0, h->stream, h->returns, h->adv,
                     h->have_flat ? (const Rec *)nullptr : h->rec, M, h->moments);
  if (int rc = launch_check("moments_kernel")) return rc;
  std::vector<double> part(4 * (size_t)kBlocks);
  HIP_TRY(hipMemcpyAsync(part.data(), h->moments, part.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  sums5[0] = (double)M;
  for (unsigned b = 0; b < kBlocks; ++b) for (int k = 0; k < 4; ++k) sums5[1 + k] += part[4 * (size_t)b + k];
  return 0;
}

int ppo_xp_explained_variance_from_sums(const double *s, float *out) {
  if (!s || !out) return fail(PPO_XP_EINVAL, "explained_variance: NULL argument");
  const double n = s[0];
  if (!(n > 0)) { *out = __builtin_nanf(""); return 0; }         // var of nothing = 0/0
  const double mr = s[1] / n, ma = s[3] / n;
  double vr = (s[2] - s[1] * mr) / n, va = (s[4] - s[3] * ma) / n;
  if (vr < 0) vr = 0;
  if (va < 0) va = 0;
  const float target_variance = (float)vr, diff_variance = (float)va;
  if (target_variance == -1.0f) *out = __builtin_nanf("");        // training.rs:92 (sic)
  else *out = 0.0f - (diff_variance / target_variance);           // training.rs:97
  return 0;
}

int ppo_xp_explained_variance(ppo_xp *h, float *out) {
  if (!h || !out) return fail(PPO_XP_EINVAL, "explained_variance: NULL argument");
  double sums5[5];
  if (int rc = ppo_xp_explained_variance_sums(h, sums5)) return rc;
  return ppo_xp_explained_variance_from_sums(sums5, out);
}

int ppo_xp_pass_finish(ppo_xp *h) {
  if (!h) return fail(PPO_XP_EINVAL, "pass_finish: handle is NULL");
  if (!h->pass_active || !h->pass_pending_finish) return fail(PPO_XP_ESTATE, "pass_finish: no pass is waiting for its statistics exchange");
  HIP_TRY(hipSetDevice(h->device));
  return pass_finish_impl(h);
}

int ppo_xp_pass_next(ppo_xp *h, ppo_batch_view *out) {
  if (!h || !out) return fail(PPO_XP_EINVAL, "pass_next: NULL argument");
  if (!h->pass_active) return fail(PPO_XP_ESTATE, "pass_next: no pass (call ppo_xp_pass_begin)");
  if (h->pass_pending_finish) return fail(PPO_XP_ESTATE, "pass_next: statistics exchange pending (call ppo_xp_pass_finish)");
  const size_t B = h->pass_B, M = h->pass_M;
  if (B > M || h->pass_iter >= M) return 1;                      // data.rs:117-119 -> None
  const size_t k = h->pass_iter / B;
  // whole-pass mode: minibatch k is rows [k*B, (k+1)*B) of the slabs, gathered by pass_begin.  Streaming: it is
  // gathered NOW into slot k % 2 (the view handed out two calls ago is overwritten)
  const size_t row = h->pass_streamed ? (k & 1) * B : k * B;
  if (h->pass_streamed) {
    HIP_TRY(hipSetDevice(h->device));
    if (int rc = gather_range(h, k, 1, pass_out_at(h, row, k))) return rc;
  }
  h->pass_iter += B;                                             // data.rs:126
  out->index = k; out->batch_size = B; out->obs_size = h->D;
  out->observations = h->obs_out + row * h->D;
  out->actions = h->act_out + row;
  out->values = h->val_out + row;
  out->neglogps = h->nlp_out + row;
  out->returns = h->ret_out + row;
  out->advantages = h->adv_out + row;
  out->adv_mean = 0.f; out->adv_std = 0.f;
  return 0;
}

int ppo_xp_shuffle_prefetch(ppo_xp *h, uint64_t seed) {
  if (!h) return fail(PPO_XP_EINVAL, "shuffle_prefetch: handle is NULL");
  if (h->pf_thread.joinable()) h->pf_thread.join();
  const size_t M = buf_len(h) * h->N;
  h->pf_valid = false;
  if (M == 0) return 0;
  h->pf_seed = seed; h->pf_M = M;
  try {
    h->pf_perm.resize(M);
    uint32_t *dst = h->pf_perm.data();
    h->pf_thread = std::thread([dst, seed, M] { shuffle_indices(seed, dst, M); });   // data.rs:93-95, off the caller's thread
  } catch (...) {
    return fail(PPO_XP_ENOMEM, "shuffle_prefetch: could not start the worker thread");
  }
  h->pf_valid = true;
  return 0;
}

int ppo_xp_pass_next_host(ppo_xp *h, ppo_batch_host *io) {
  if (!h || !io) return fail(PPO_XP_EINVAL, "pass_next_host: NULL argument");
  if (!io->observations || !io->actions || !io->values || !io->neglogps || !io->returns || !io->advantages)
    return fail(PPO_XP_EINVAL, "pass_next_host: the six destination arrays must be set by the caller");
  ppo_batch_view v;
  const int rc = ppo_xp_pass_next(h, &v);
  if (rc != 0) return rc;
  if (io->capacity < v.batch_size) {
    h->pass_iter -= h->pass_B;                                   // nothing was consumed
    return fail(PPO_XP_EINVAL, "pass_next_host: destination holds %zu rows, the minibatch has %zu", io->capacity, v.batch_size);
  }
  HIP_TRY(hipSetDevice(h->device));
  const size_t B = v.batch_size, D = v.obs_size;
  io->index = v.index; io->batch_size = B; io->obs_size = D;
  static_assert(sizeof(size_t) == sizeof(int64_t), "usize actions travel as 64-bit words");
  // data.rs:166-182 uploads six host Vecs; here the six device arrays of minibatch k come down instead
  HIP_TRY(hipMemcpyAsync(io->observations, v.observations, B * D * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipMemcpyAsync(io->actions, v.actions, B * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipMemcpyAsync(io->values, v.values, B * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipMemcpyAsync(io->neglogps, v.neglogps, B * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipMemcpyAsync(io->returns, v.returns, B * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipMemcpyAsync(io->advantages, v.advantages, B * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return 0;
}

// Six device-to-host copies of minibatch k (rows [row, row + B) of the slabs) into `io`, on the copy stream, ordered
// after everything the handle's stream holds now; xfer_done is recorded behind them.
static int enqueue_host_copies(ppo_xp *h, size_t row, const ppo_batch_host *io) {
  if (!h->xfer) {
    HIP_TRY(hipStreamCreateWithFlags(&h->xfer, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&h->xfer_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&h->xfer_done, hipEventDisableTiming));
  }
  const size_t B = h->pass_B, D = h->D;
  HIP_TRY(hipEventRecord(h->xfer_fork, h->stream));
  HIP_TRY(hipStreamWaitEvent(h->xfer, h->xfer_fork, 0));
  HIP_TRY(hipMemcpyAsync(io->observations, h->obs_out + row * D, B * D * sizeof(float), hipMemcpyDeviceToHost, h->xfer));
  HIP_TRY(hipMemcpyAsync(io->actions, h->act_out + row, B * sizeof(int64_t), hipMemcpyDeviceToHost, h->xfer));
  HIP_TRY(hipMemcpyAsync(io->values, h->val_out + row, B * sizeof(float), hipMemcpyDeviceToHost, h->xfer));
  HIP_TRY(hipMemcpyAsync(io->neglogps, h->nlp_out + row, B * sizeof(float), hipMemcpyDeviceToHost, h->xfer));
  HIP_TRY(hipMemcpyAsync(io->returns, h->ret_out + row, B * sizeof(float), hipMemcpyDeviceToHost, h->xfer));
  HIP_TRY(hipMemcpyAsync(io->advantages, h->adv_out + row, B * sizeof(float), hipMemcpyDeviceToHost, h->xfer));
  HIP_TRY(hipEventRecord(h->xfer_done, h->xfer));
  h->xfer_used = true;
  return 0;
}

static bool host_slot_ok(const ppo_batch_host *io) {
  return io->observations && io->actions && io->values && io->neglogps && io->returns && io->advantages;
}

int ppo_xp_pass_next_host_pipelined(ppo_xp *h, ppo_batch_host *cur, ppo_batch_host *next_slot) {
  if (!h || !cur) return fail(PPO_XP_EINVAL, "pass_next_host_pipelined: NULL argument");
  if (!host_slot_ok(cur) || (next_slot && !host_slot_ok(next_slot)))
    return fail(PPO_XP_EINVAL, "pass_next_host_pipelined: the six destination arrays of a slot must be set by the caller");
  if (next_slot && next_slot->observations == cur->observations)
    return fail(PPO_XP_EINVAL, "pass_next_host_pipelined: the two slots must be different arrays");
  if (!h->pass_active) return fail(PPO_XP_ESTATE, "pass_next: no pass (call ppo_xp_pass_begin)");
  if (h->pass_pending_finish) return fail(PPO_XP_ESTATE, "pass_next: statistics exchange pending (call ppo_xp_pass_finish)");
  if (h->pass_streamed)
    return fail(PPO_XP_ESTATE, "pass_next_host_pipelined: needs the whole-pass slabs (not with ppo_xp_set_streaming); use ppo_xp_pass_next_host");
  const size_t B = h->pass_B, M = h->pass_M, K = h->pass_K;
  if (B > M || h->pass_iter >= M) { h->hp_valid = false; return 1; }          // data.rs:117-119 -> None
  if (cur->capacity < B || (next_slot && next_slot->capacity < B))
    return fail(PPO_XP_EINVAL, "pass_next_host_pipelined: a slot holds fewer rows than the minibatch (%zu)", B);
  HIP_TRY(hipSetDevice(h->device));
  const size_t k = h->pass_iter / B;
  const bool hit = h->hp_valid && h->hp_pass == h->pass_serial && h->hp_k == k && h->hp_obs == cur->observations;
  if (!hit) {
    // nothing (or something else) was prefetched: first minibatch of a pass, or the caller changed slots
    if (h->hp_valid) HIP_TRY(hipEventSynchronize(h->xfer_done));   // a stale prefetch must not land in a slot the caller reuses
    if (int rc = enqueue_host_copies(h, k * B, cur)) return rc;
  }
  h->hp_valid = false;
  HIP_TRY(hipEventSynchronize(h->xfer_done));                       // minibatch k has arrived (the only wait of the call)
  h->pass_iter += B;                                                // data.rs:126
  cur->index = k; cur->batch_size = B; cur->obs_size = h->D;
  if (next_slot && k + 1 < K) {
    // minibatch k+1 starts travelling now, while the caller consumes k (copy_from + learner.step, training.rs:205-207)
    if (int rc = enqueue_host_copies(h, (k + 1) * B, next_slot)) return rc;
    h->hp_valid = true; h->hp_k = k + 1; h->hp_obs = next_slot->observations; h->hp_pass = h->pass_serial;
  }
  return 0;
}

int ppo_xp_pass_gather_kind(const ppo_xp *h) { return (h && h->pass_active) ? h->pass_kind : -1; }
size_t ppo_xp_pass_num_batches(const ppo_xp *h) { return (h && h->pass_active) ? h->pass_K : 0; }
const uint32_t *ppo_xp_pass_perm_device(const ppo_xp *hc) {
  ppo_xp *h = const_cast<ppo_xp *>(hc);
  if (!h) return nullptr;
  if (h->pass_bij && h->pass_active && !h->bij_perm_written) {     // inline pass: nobody has needed the array so far
    if (hipSetDevice(h->device) != hipSuccess || materialize_bijection(h) != 0) return nullptr;
    h->perm_cur = h->perm;
  }
  return h->perm_cur;
}

int ppo_xp_pass_perm_host(ppo_xp *h, uint32_t *out, size_t n) {
  if (!h || !out) return fail(PPO_XP_EINVAL, "pass_perm_host: NULL argument");
  if (!h->pass_active) return fail(PPO_XP_ESTATE, "pass_perm_host: no pass");
  if (n != h->pass_M) return fail(PPO_XP_EINVAL, "pass_perm_host: n = %zu, M = %zu", n, h->pass_M);
  HIP_TRY(hipSetDevice(h->device));
  if (n && !ppo_xp_pass_perm_device(h)) return fail(PPO_XP_EIO, "pass_perm_host: the permutation could not be written out");
  if (n) HIP_TRY(hipMemcpyAsync(out, h->perm_cur, n * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return 0;
}

int ppo_xp_pass_stats_host(ppo_xp *h, float *mean_std_out, size_t k) {
  if (!h || !mean_std_out) return fail(PPO_XP_EINVAL, "pass_stats_host: NULL argument");
  if (!h->pass_active || h->pass_pending_finish) return fail(PPO_XP_ESTATE, "pass_stats_host: no finished pass");
  if (k != h->pass_K) return fail(PPO_XP_EINVAL, "pass_stats_host: k = %zu, pass has %zu minibatches", k, h->pass_K);
  if (h->pass_streamed && h->pass_K && h->pass_iter < h->pass_K * h->pass_B)
    return fail(PPO_XP_ESTATE, "pass_stats_host: a streamed pass knows minibatch k's statistics only after ppo_xp_pass_next has gathered it; "
                "%zu of %zu minibatches so far", h->pass_iter / h->pass_B, h->pass_K);
  HIP_TRY(hipSetDevice(h->device));
  if (k) HIP_TRY(hipMemcpyAsync(mean_std_out, h->stats, 2 * k * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return 0;
}

int ppo_xp_set_streaming(ppo_xp *h, int on) {
  if (!h) return fail(PPO_XP_EINVAL, "set_streaming: handle is NULL");
  if (on && h->xmode != 0) return fail(PPO_XP_ESTATE, "set_streaming: not with a statistics exchange (one all-reduce per pass needs the whole pass)");
  if ((on != 0) != h->streaming) {
    // the slabs change size: drop them (after the stream has finished with them); the next pass re-allocates
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->xfer) HIP_TRY(hipStreamSynchronize(h->xfer));
    dev_free(h->obs_out); dev_free(h->act_out); dev_free(h->val_out); dev_free(h->nlp_out); dev_free(h->ret_out); dev_free(h->adv_out);
    h->slab_rows = 0;
    h->pass_active = false;
  }
  h->streaming = on != 0;
  return 0;
}

int ppo_xp_set_exchange(ppo_xp *h, int mode, void *nccl_comm) {
  if (!h) return fail(PPO_XP_EINVAL, "set_exchange: handle is NULL");
  if (mode < 0 || mode > 2) return fail(PPO_XP_EINVAL, "set_exchange: mode %d", mode);
  if (mode != 0 && h->streaming) return fail(PPO_XP_ESTATE, "set_exchange: not while minibatches are streamed (ppo_xp_set_streaming)");
  if (mode == 1) {
    if (!nccl_comm) return fail(PPO_XP_EINVAL, "set_exchange: mode 1 needs a ncclComm_t");
    if (int rc = rccl_load()) return rc;
  }
  h->xmode = mode; h->comm = (mode == 1) ? nccl_comm : nullptr;
  return 0;
}

int ppo_xp_attach_comm(ppo_xp *h, void *nccl_comm, int rank, int world) {
  if (!h) return fail(PPO_XP_EINVAL, "attach_comm: handle is NULL");
  if (!nccl_comm || world <= 1) return ppo_xp_set_exchange(h, 0, nullptr);
  if (rank < 0 || rank >= world) return fail(PPO_XP_EINVAL, "attach_comm: rank %d of %d", rank, world);
  return ppo_xp_set_exchange(h, 1, nccl_comm);
}

double *ppo_xp_pass_sums_device(ppo_xp *h) { return h ? h->sums : nullptr; }

int ppo_xp_rccl_unique_id(void *out128) {
  if (!out128) return fail(PPO_XP_EINVAL, "rccl_unique_id: NULL");
  if (int rc = rccl_load()) return rc;
  RcclUniqueId id;
  int e = g_rccl.GetUniqueId(&id);
  if (e != 0) return fail(PPO_XP_EIO, "ncclGetUniqueId failed: %s", g_rccl.GetErrorString(e));
  std::memcpy(out128, &id, sizeof(id));
  return 0;
}

int ppo_xp_rccl_comm_init(void **comm_out, const void *unique_id128, int rank, int world) {
  if (!comm_out || !unique_id128) return fail(PPO_XP_EINVAL, "rccl_comm_init: NULL");
  if (int rc = rccl_load()) return rc;
  RcclUniqueId id;
  std::memcpy(&id, unique_id128, sizeof(id));
  void *comm = nullptr;
  int e = g_rccl.CommInitRank(&comm, world, id, rank);
  if (e != 0) return fail(PPO_XP_EIO, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(e));
  *comm_out = comm;
  return 0;
}

int ppo_xp_rccl_comm_count(void *comm, int *ranks_out) {
  if (!comm || !ranks_out) return fail(PPO_XP_EINVAL, "rccl_comm_count: NULL");
  if (int rc = rccl_load()) return rc;
  int e = g_rccl.CommCount(comm, ranks_out);
  if (e != 0) return fail(PPO_XP_EIO, "ncclCommCount failed: %s", g_rccl.GetErrorString(e));
  return 0;
}

int ppo_xp_rccl_comm_destroy(void *comm) {
  if (!comm) return 0;
  if (int rc = rccl_load()) return rc;
  int e = g_rccl.CommDestroy(comm);
  if (e != 0) return fail(PPO_XP_EIO, "ncclCommDestroy failed: %s", g_rccl.GetErrorString(e));
  return 0;
}

void *ppo_xp_stream(const ppo_xp *h) { return h ? (void *)h->stream : nullptr; }

int ppo_xp_set_stream(ppo_xp *h, void *stream) {
  if (!h) return fail(PPO_XP_EINVAL, "set_stream: handle is NULL");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (h->side_pending) HIP_TRY(hipStreamSynchronize(h->side));   // its join event stays valid for the new stream
  if (h->xfer) HIP_TRY(hipStreamSynchronize(h->xfer));
  if ((hipStream_t)stream == h->stream) return 0;               // already there (the handle's own stream stays its own)
  if (h->own_stream) { (void)hipStreamDestroy(h->stream); h->own_stream = false; }
  h->stream = (hipStream_t)stream;
  return 0;
}

int ppo_xp_sync(ppo_xp *h) {
  if (!h) return fail(PPO_XP_EINVAL, "sync: handle is NULL");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (h->xfer) HIP_TRY(hipStreamSynchronize(h->xfer));     // a prefetched host minibatch still travelling
  return onepass_check(h);
}

int ppo_xp_device(const ppo_xp *h) { return h ? h->device : -1; }

int ppo_xp_prof_enable(ppo_xp *h, int enable) {
  if (!h) return fail(PPO_XP_EINVAL, "prof_enable: handle is NULL");
  h->prof = enable < 0 ? 0xffffffffu : (unsigned)enable;
  return 0;
}

int ppo_xp_prof_collect(ppo_xp *h, double *ms_out, uint64_t *launches_out) {
  if (!h || !ms_out || !launches_out) return fail(PPO_XP_EINVAL, "prof_collect: NULL argument");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (int i = 0; i < PPO_XP_K_COUNT; ++i) { ms_out[i] = 0.0; launches_out[i] = 0; }
  for (auto &r : h->prof_recs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { ms_out[r.kid] += ms; launches_out[r.kid] += 1; }
    h->ev_pool.push_back(r.a); h->ev_pool.push_back(r.b);
  }
  h->prof_recs.clear();
  return 0;
}

int ppo_xp_set_tunable(ppo_xp *h, const char *name, long value) {
  if (!h || !name) return fail(PPO_XP_EINVAL, "set_tunable: NULL argument");
  std::string n(name);
  // a streamed pass launches per minibatch with shapes fixed at pass_begin (partials, tiles): a tunable changed in the
  // middle would mix them, so it ends the pass
  if (h->pass_active && h->pass_streamed) h->pass_active = false;
  if (n == "gather_nt" && (value == 0 || value == 1)) h->gather_nt = value;
  else if (n == "gather_block" && (value == 64 || value == 256)) h->gather_block = value;
  else if (n == "fold_tiles" && value >= 0 && value <= 65536) h->fold_tiles = value;
  else if (n == "gather_xcd" && value >= 0 && value <= 2) h->gather_xcd = value;
  else if (n == "gae_waves" && (value == 0 || value == 1 || value == 2 || value == 4 || value == 8 || value == 16)) h->gae_waves = value;
  else if (n == "summary_nt" && (value == 0 || value == 1)) h->summary_nt = value;
  else if (n == "gae_nt" && (value == 0 || value == 1)) h->gae_nt = value;
  else if (n == "overlap" && (value == 0 || value == 1)) h->overlap = value;
  else if (n == "gae_flat" && (value == 0 || value == 1)) h->gae_flat = value;
  else if (n == "fused" && value >= -1 && value <= 1) h->fused = value;
  else if (n == "fused_block" && (value == 256 || value == 512)) h->fused_block = value;
#ifdef PPO_XP_EXPERIMENTS
  else if (n == "scalar_path" && (value == 0 || value == 1)) h->scalar_path = value;
#endif
  else if (n == "bij_mode" && value >= 0 && value <= 2) h->bij_mode = value;
  else if (n == "gae_onepass" && value >= -1 && value <= 1) h->gae_onepass = value;
  else if (n == "fused_iter" && (value == 0 || value == 1 || value == 2 || value == 4)) h->fused_iter = value;
  else if (n == "op_l" && (value == 16 || value == 32 || value == 64)) h->op_l = value;
  else if (n == "op_w" && (value == 2 || value == 4 || value == 8 || value == 16)) h->op_w = value;
  else if (n == "rec_nt" && (value == 0 || value == 1)) h->rec_nt = value;
  else if (n == "out_nt" && (value == 0 || value == 1)) h->out_nt = value;
  else if (n == "gae_unroll" && (value == 0 || value == 4 || value == 8 || value == 16 || value == 32)) h->gae_unroll = value;
  else if (n == "carry_waves" && (value == 4 || value == 8 || value == 16)) h->carry_waves = value;
  else return fail(PPO_XP_EINVAL, "set_tunable: unknown name or bad value: %s = %ld", name, value);
  return 0;
}

}  // extern "C"
