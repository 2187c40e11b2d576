/*
 * ppo_xp.h — C ABI of the MI355X-native experience path (rollout buffer -> GAE/returns ->
 * advantage standardisation -> shuffled minibatch gather) of flawedmatrix/ppo.
 *
 * The reference has no FFI: the path is a set of crate-private Rust methods
 * (ppo/src/common/buffer.rs, ppo/src/model/data.rs) called only from train()
 * (ppo/src/training.rs:114,163,172,186,187,191,205).  Each entry point below names the Rust
 * method it replaces (paths relative to the reference checkout).  The Rust-side binding a
 * maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every function returning int: 0 = ok, < 0 = -errno style failure (PPO_XP_E*), and the
 *     text of the failure is available from ppo_xp_last_error() (thread local).  Nothing
 *     throws across the ABI.  Where the reference panics (size mismatches) the call fails
 *     with PPO_XP_EINVAL and changes nothing; a Rust shim turns that back into a panic.
 *   - a handle is owned by one thread at a time (the reference takes &mut self); distinct
 *     handles (one per GPU / rank) are independent.
 *   - all device work of a handle is enqueued on ONE HIP stream (ppo_xp_stream()).  Calls
 *     that return host data synchronise that stream; all others only enqueue.
 *   - device pointers handed out stay valid until the next ppo_xp_pass_begin(), a
 *     ppo_xp_returns() (for returns/advantages) or ppo_xp_destroy().
 *   - the library is HIP only: there is no CPU fallback.  Creating a handle without a usable
 *     GPU fails with PPO_XP_ENODEV.
 */
#ifndef PPO_XP_H
#define PPO_XP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PPO_XP_OK 0
#define PPO_XP_EINVAL (-22) /* size / argument mismatch: the reference would panic      */
#define PPO_XP_ENOMEM (-12) /* device or host allocation failed                         */
#define PPO_XP_EIO (-5)     /* a HIP / RCCL call failed (text in ppo_xp_last_error())   */
#define PPO_XP_ENODEV (-19) /* no usable HIP device                                     */
#define PPO_XP_ESTATE (-1)  /* call out of order (e.g. pass_begin before returns)       */

/* GAE algorithm selector for ppo_xp_returns(). */
#define PPO_XP_GAE_SEQUENTIAL 0 /* K1: one lane per env, strict reverse-time order; bit-exact
                                   with the reference's fp32 op order (buffer.rs:206-222)   */
#define PPO_XP_GAE_ASSOC 1      /* K2: time-chunked associative linear-recurrence scan; for long
                                   horizons / few envs; NOT bit-exact (bound stated in DESIGN.md).
                                   Up to 32 768 steps it runs as ONE launch (chunks kept in registers,
                                   chunk maps exchanged inside the running grid); longer buffers, or a
                                   forced chunk length, take three launches (chunk maps, LDS carry
                                   scan, replay).  Deterministic either way.                        */
/* bits 16..31 of `algo`, when non-zero, force the chunk length of PPO_XP_GAE_ASSOC (three-launch form). */
#define PPO_XP_GAE_CHUNK(len) ((int)((unsigned)(len) << 16))
/* do not build the packed per-transition records the gather reads (GAE-only measurements;
 * ppo_xp_pass_begin() then builds them itself on first use). */
#define PPO_XP_GAE_NO_PACK 0x100

typedef struct ppo_xp ppo_xp; /* opaque: replaces ExperienceBuffer + ExperienceBatcher */

/* Device views of the flattened buffer: replaces training_views() (buffer.rs:142-175) and the
 * master `set` of ExperienceBatcher (data.rs:14-19,63-70).  Flat index i = t*num_envs + e over
 * rows 0..len in STORAGE order (after a ring wrap this is not chronological, buffer.rs:386-405).
 * returns/advantages are valid after ppo_xp_returns(). */
typedef struct ppo_xp_views_t {
  size_t num_steps;         /* M = len * num_envs                              */
  size_t obs_size;          /* D                                               */
  const float *observations;/* [M, D]  f32 device                              */
  const int64_t *actions;   /* [M]     i64 device (usize in the reference)     */
  const float *values;      /* [M]                                             */
  const float *neglogps;    /* [M]                                             */
  const float *rewards;     /* [M]                                             */
  const float *dones;       /* [M]  0.0 / 1.0 (buffer.rs:95)                   */
  const float *returns;     /* [M]  fl(A + v)              (buffer.rs:222)     */
  const float *advantages;  /* [M]  fl(fl(A + v) - v)      (data.rs:62)        */
} ppo_xp_views_t;

/* One minibatch on the device: replaces struct ExperienceBatch (data.rs:4-12). */
typedef struct ppo_batch_view {
  size_t index;             /* k, 0-based position in the pass                 */
  size_t batch_size;        /* B                                               */
  size_t obs_size;          /* D                                               */
  const float *observations;/* [B, D]                                          */
  const int64_t *actions;   /* [B]                                             */
  const float *values;      /* [B]                                             */
  const float *neglogps;    /* [B]                                             */
  const float *returns;     /* [B]                                             */
  const float *advantages;  /* [B] standardised per minibatch (data.rs:157-164)*/
  float adv_mean;           /* filled only if the caller asked for host stats  */
  float adv_std;
} ppo_batch_view;

/* One minibatch in CALLER host memory: the six Vecs that the reference's next() hands to
 * Tensor::copy_from (data.rs:166-182).  The caller sets the six pointers (each with room for
 * `capacity` rows; page-locked memory from ppo_xp_host_alloc makes the copies DMA transfers) and
 * `capacity`; ppo_xp_pass_next_host fills index / batch_size / obs_size and the arrays. */
typedef struct ppo_batch_host {
  size_t capacity;          /* in: rows each destination array can hold (>= batch_size)    */
  size_t index;             /* out: k                                                      */
  size_t batch_size;        /* out: B                                                      */
  size_t obs_size;          /* out: D                                                      */
  float *observations;      /* in: [capacity, D] host                                      */
  size_t *actions;          /* in: [capacity] host, usize like the reference (data.rs:7)   */
  float *values;            /* in: [capacity] host                                         */
  float *neglogps;          /* in: [capacity] host                                         */
  float *returns;           /* in: [capacity] host                                         */
  float *advantages;        /* in: [capacity] host; standardised (data.rs:157-164)         */
} ppo_batch_host;

/* ---- lifetime -------------------------------------------------------------------------- */

/* ExperienceBuffer::new_with_params (buffer.rs:52-69); `new` (buffer.rs:45-47) is gamma=0.99,
 * lambda=0.95.  NUM_ENVS / OBS_SIZE are run-time here (const generics in the reference).
 * Allocates the six zero-filled streams on HIP device `device`. */
int ppo_xp_create(ppo_xp **out, int device, size_t capacity, size_t num_envs, size_t obs_size,
                  float gamma, float lambda);
/* Drop of ExperienceBuffer / ExperienceBatcher. */
void ppo_xp_destroy(ppo_xp *h);

/* ---- insert ---------------------------------------------------------------------------- */

/* ExperienceBuffer::add_experience (buffer.rs:84-121).  Host pointers; obs is [n, obs_size],
 * the rest [n]; dones are Rust bools (1 byte, 0/1).  n != num_envs => PPO_XP_EINVAL (the
 * reference panics, buffer.rs:97-112).  The host arrays are the caller's again when the call returns.
 * Rows of up to 256 KiB (all six arrays) are copied into a page-locked slot of the handle and inserted by one
 * kernel that reads the slot over the bus: the call does not wait for the GPU (later calls on the handle are
 * ordered after it); larger rows take one copy command per array and the call waits for them. */
int ppo_xp_add(ppo_xp *h, const float *obs, const float *rewards, const size_t *actions,
               const float *vals, const uint8_t *dones, const float *neglogps, size_t n);
/* `steps` consecutive add_experience calls from [steps, n, ...] host arrays (same ring
 * semantics as calling ppo_xp_add in a loop; one upload per stream per contiguous run). */
int ppo_xp_add_many(ppo_xp *h, const float *obs, const float *rewards, const size_t *actions,
                    const float *vals, const uint8_t *dones, const float *neglogps, size_t n,
                    size_t steps);
/* add_experience with DEVICE source rows (device-resident simulators / policies);
 * dones_f32 exactly 0.0 or 1.0 as the reference stores them (buffer.rs:95; the scans read the flags as
 * bits, any non-zero value counts as done).  The insert (one kernel per row) is ENQUEUED on the handle's stream and the call
 * returns without waiting: the source arrays must stay valid and unchanged until that stream has
 * passed them (ppo_xp_sync(), or an event recorded on ppo_xp_stream() after the call), and whatever
 * produced them must be ordered before the call (same stream, or synchronised).  The Python mirror
 * orders both ways against torch's current stream (ppo_amd/buffer.py add_experience_device). */
int ppo_xp_add_device(ppo_xp *h, const float *obs, const float *rewards, const int64_t *actions,
                      const float *vals, const float *dones_f32, const float *neglogps, size_t n);

/* Page-locked host memory for the rows handed to ppo_xp_add / ppo_xp_add_many (hipHostMalloc /
 * hipHostFree).  Rows that live in such memory are DMA-ed by the copy engines at PCIe rate; rows in
 * ordinary (pageable) memory are staged by the HIP runtime at a fraction of it ("PCIe-
 * inclusive rate").  No counterpart in the reference, whose buffer is host memory. */
void *ppo_xp_host_alloc(size_t bytes);
void ppo_xp_host_free(void *p);

/* ExperienceBuffer::reset_counter (buffer.rs:131-133): counter = 0, data untouched. */
void ppo_xp_reset_counter(ppo_xp *h);
/* ExperienceBuffer::len (buffer.rs:123-129): min(counter, capacity). */
size_t ppo_xp_len(const ppo_xp *h);
/* Internal ring state, for tests of buffer.rs:93,115-120. */
size_t ppo_xp_counter(const ppo_xp *h);
size_t ppo_xp_last_values_idx(const ppo_xp *h);

/* ---- GAE / returns --------------------------------------------------------------------- */

/* ExperienceBuffer::returns (buffer.rs:187-225) fused with `advantages = returns - values`
 * of ExperienceBatcher::new (data.rs:62).  last_dones: n Rust bools on the host.  Leaves
 * returns / advantages resident on the device; if host_out != NULL also copies the len*n
 * returns to it (synchronises).  n != num_envs => PPO_XP_EINVAL (buffer.rs:194). */
int ppo_xp_returns(ppo_xp *h, const uint8_t *last_dones, size_t n, int algo, float *host_out);

/* ppo_xp_returns for callers that cannot allocate (tight serving loops, harnesses that forbid hipMalloc, code that
 * runs between the allocations of a captured region).  NO COUNTERPART IN THE REFERENCE.  Same arguments, results and
 * bits as ppo_xp_returns; the difference is where the associative scan (PPO_XP_GAE_ASSOC) keeps its chunk maps and
 * carries: ppo_xp_returns grows three device arrays on demand (hipMalloc), this entry point uses a file-scope
 * __device__ array of 3 x 2^20 floats (12 MiB, held by every GPU the library runs on until the process ends) and
 * calls no hipMalloc / hipFree / allocating routine; per-kernel profiling events are taken from the handle's pool only.
 * Limit: chunks x num_envs <= 2^20 (chunks = ceil(len / chunk length); choose longer chunks with PPO_XP_GAE_CHUNK) -
 * beyond it the call fails with PPO_XP_EINVAL BEFORE anything changes (ring state, pass state, last_dones slots).
 * Every call on one GPU shares that GPU's array: nothing in it has to start at zero (each call writes every entry it
 * later reads), but calls that overlap on one GPU - two handles on two streams - race on it.
 * NOT a capturable call: like ppo_xp_returns it consumes `last_dones` AT CALL TIME (host copy into one of eight
 * page-locked slots the kernels read in place; every fourth call waits on / records an event for slot reuse), so a
 * HIP graph captured around it would replay with whatever that slot holds later.  Allocation-free, not capture-safe. */
int ppo_xp_returns_noalloc(ppo_xp *h, const uint8_t *last_dones, size_t n, int algo, float *host_out);

/* explained_variance(values, returns) of train() (ppo/src/training.rs:88-99, called at :189):
 * 0 - var(returns - values) / var(returns), population variances (fp64 sums on the device; the
 * reference runs ndarray's sequential fp32 Welford).  Needs ppo_xp_returns() first; synchronises. */
int ppo_xp_explained_variance(ppo_xp *h, float *out);
/* The same in two halves for env-sharded jobs: sums5 = (M, sum ret, sum ret^2, sum adv, sum adv^2) of
 * this rank, f64, to the host; the caller adds the ranks' vectors (all-reduce of 5 doubles) and
 * ppo_xp_explained_variance_from_sums finishes with the arithmetic of ppo_xp_explained_variance. */
int ppo_xp_explained_variance_sums(ppo_xp *h, double *sums5);
int ppo_xp_explained_variance_from_sums(const double *sums5, float *out);

/* ExperienceBuffer::training_views (buffer.rs:142-175) + the batcher's master set.  A pass reads the
 * GAE results from packed records, so ppo_xp_returns(host_out = NULL) does not store the flat returns /
 * advantages arrays; the first call here after such a ppo_xp_returns enqueues the small kernel that
 * unpacks them on the handle's stream (same bits: ret as stored, adv' = fl(ret - v), data.rs:62).
 * Inserts and scans are only enqueued on that stream: read the views on it (ppo_xp_stream) or after ppo_xp_sync(). */
int ppo_xp_views(const ppo_xp *h, ppo_xp_views_t *out);

/* ---- one shuffled pass over the buffer (IntoIterator + Iterator, data.rs:85-193) ------- */

/* (&batcher).into_iter() (data.rs:91-102) plus the work of every next() of the pass
 * (data.rs:115-192): takes / draws the permutation, gathers all K = ceil(M/B) minibatches into
 * contiguous device slabs and standardises each minibatch's advantages.
 *   perm != NULL : host array of M u32 indices, used as given (must be a permutation of 0..M
 *                  for reference semantics; every entry is range-checked).
 *   perm == NULL : the library draws it on the host: rand 0.8.5 Fisher-Yates + u32 gen_range
 *                  over ChaCha12 seeded like ChaCha12Rng::seed_from_u64(seed) (the reference's
 *                  thread_rng() is OS-seeded and cannot be reproduced, data.rs:94).
 * batch_size > M is not an error: the pass is empty (data.rs:117-119).  batch_size == 0 is
 * PPO_XP_EINVAL (the reference would loop forever on empty batches). */
int ppo_xp_pass_begin(ppo_xp *h, size_t batch_size, const uint32_t *perm, uint64_t seed);
/* Same, with the permutation already on the device (u32[M]).  The array is BORROWED, not copied:
 * it must stay valid and unchanged until the pass's kernels have run (ppo_xp_sync(), or the next
 * pass_begin).  Entries are not range-checked; the kernel clamps them to M-1 so that a bad index
 * can never fault the GPU. */
int ppo_xp_pass_begin_device(ppo_xp *h, size_t batch_size, const uint32_t *perm_dev);
/* Same, with the permutation drawn ON THE DEVICE (no host Fisher-Yates, no upload): perm = stable
 * argsort of key_i = SplitMix64(seed, i).  A uniform permutation like the reference's, but NOT the
 * sequence rand's shuffle would give for this seed; reproducible, with a CPU twin in the oracle. */
int ppo_xp_pass_begin_device_shuffle(ppo_xp *h, size_t batch_size, uint64_t seed);
/* Same, with the permutation given by a KEYED BIJECTION of (seed, M) that the gather kernels evaluate in registers
 * (ppo_amd/csrc/bijection.h: six alternating Feistel steps on max(8, ceil(log2 M)) bits, cycle-walked into [0, M)):
 * into_iter() (data.rs:91-102) without an index array - nothing is drawn, sorted, uploaded or read, so permutation
 * generation is part of the pass itself.  When the fused gather runs (rows and scalar streams in one kernel) perm(j) is
 * computed once per output row; the kernels that read an index array (wide rows, records beyond the Infinity Cache)
 * get it written out first by one small launch.  ppo_xp_pass_perm_device / _host write it out on demand.  A keyed
 * pseudo-random permutation like the reference's shuffle (uniformity checked in tests/), not rand's sequence;
 * reproducible, with CPU twins in the oracle (C and numpy), indices bit-equal. */
int ppo_xp_pass_begin_bijection(ppo_xp *h, size_t batch_size, uint64_t seed);
/* Iterator::next (data.rs:115-192): 0 = *out filled, 1 = exhausted (`None`), < 0 error.
 * The windows follow data.rs:121-126 exactly, including the overlapping tail batch.
 * Whole-pass mode (default): ppo_xp_pass_begin* gathered every minibatch, this only hands out slices of the slabs.
 * Streaming mode (ppo_xp_set_streaming): THIS call enqueues the gather of minibatch k into one of two alternating
 * B-row slots; the view stays valid until the call after the next one. */
int ppo_xp_pass_next(ppo_xp *h, ppo_batch_view *out);
/* Iterator::next (data.rs:115-192) as the reference's caller sees it: the six arrays of minibatch k in
 * HOST memory (what data.rs:166-182 feeds to copy_from), so a Rust shim can keep returning
 * ExperienceBatch<OBS_SIZE, D> on whatever device its learner uses.  Copies on the handle's stream
 * and synchronises.  Same return values as ppo_xp_pass_next; io->capacity < B => PPO_XP_EINVAL and
 * the minibatch is not consumed. */
int ppo_xp_pass_next_host(ppo_xp *h, ppo_batch_host *io);
/* The same hand-off with the download of minibatch k+1 OVERLAPPING the caller's work on k (the reference's loop is
 * `for batch in exp_batcher.into_iter() { learner.step(batch) }`, training.rs:205-207: copy_from + step of k take time).
 * The caller owns TWO slots (page-locked: ppo_xp_host_alloc) and alternates them:
 *     ppo_xp_pass_next_host_pipelined(h, &A, &B);  consume A;  ..._pipelined(h, &B, &A);  consume B;  ...
 * The call returns minibatch k in `cur` - already on its way if the previous call named `cur` as its `next_slot`, so
 * the only wait is one event - and, before returning, enqueues the six copies of minibatch k+1 into `next_slot`
 * (may be NULL: no prefetch) on a copy stream of the handle.  `next_slot` must not be touched until the call that
 * passes it as `cur` returns.  Whole-pass mode only (PPO_XP_ESTATE in streaming mode).  Same bits as
 * ppo_xp_pass_next_host; same return values.  No counterpart in the reference. */
int ppo_xp_pass_next_host_pipelined(ppo_xp *h, ppo_batch_host *cur, ppo_batch_host *next_slot);
/* Starts drawing the host Fisher-Yates permutation for `seed` (and the current len) on a worker
 * thread; a later ppo_xp_pass_begin(h, B, NULL, seed) with the same seed and M takes it instead of
 * shuffling on the caller's thread (the reference draws a fresh permutation per pass, training.rs:204-205:
 * prefetch pass i+1's while pass i trains).  Any other pass_begin simply discards it. */
int ppo_xp_shuffle_prefetch(ppo_xp *h, uint64_t seed);
/* Which kernels gathered the current pass: 0 = scalar_gather_kernel + obs_gather_kernel, 1 = fused_gather_kernel
 * (rows and scalar streams in one launch; chosen by size, see the "fused" tunable), 2 = fused_gather_loop_kernel (the same
 * with several row groups per workgroup: keyed-bijection passes, see "fused_iter"), -1 = no pass.  For
 * measurement harnesses that attribute bytes to kernels. */
int ppo_xp_pass_gather_kind(const ppo_xp *h);
/* Streaming mode: a pass keeps two B-row minibatch slots instead of the K*B-row slabs (which are a second copy of
 * the observations: 34 GB at T=2048, N=16384, D=256) and gathers minibatch k when ppo_xp_pass_next asks for it, so
 * a consumer can work on minibatch k while k+1 is gathered (the double-buffered ownership of SURVEY.md 8(b)).  Same
 * kernels, same bits as the whole-pass mode; K launches sets instead of one, so a pass takes a few percent longer.
 * Not together with a statistics exchange (the single all-reduce per pass needs every minibatch's sums at once):
 * PPO_XP_ESTATE.  Switching drops the slabs (synchronises) and ends the current pass.  A permutation borrowed by
 * ppo_xp_pass_begin_device must then stay valid until the pass is exhausted.  No counterpart in the reference. */
int ppo_xp_set_streaming(ppo_xp *h, int on);
/* Number of minibatches of the current pass (ExperienceBatcher::len()/B rounded up). */
size_t ppo_xp_pass_num_batches(const ppo_xp *h);
/* The permutation of the current pass on the device (u32[M]) / copied to the host. */
const uint32_t *ppo_xp_pass_perm_device(const ppo_xp *h);
int ppo_xp_pass_perm_host(ppo_xp *h, uint32_t *out, size_t n);
/* Per-minibatch (mean, unbiased std) of the current pass, K pairs of f32 to the host
 * (synchronises).  correct_std: data.rs:196-200.  Streaming mode: minibatch k's pair exists only once ppo_xp_pass_next
 * has gathered it, so the call fails with PPO_XP_ESTATE until the pass is exhausted. */
int ppo_xp_pass_stats_host(ppo_xp *h, float *mean_std_out, size_t k);

/* ---- env-sharded multi-GPU (no counterpart in the reference, which is single device) --- */

/* Per-minibatch advantage sums cross ranks.  mode 0: none (single GPU).  mode 2 "external":
 * ppo_xp_pass_begin() stops after writing this rank's K triples (count, sum, sum of squares;
 * f64) to ppo_xp_pass_sums_device(); the caller all-reduces (sum) that buffer over the ranks
 * on the handle's stream and then calls ppo_xp_pass_finish().  mode 1 "rccl": the library
 * itself calls ncclAllReduce on `nccl_comm` (a ncclComm_t created by the caller) between
 * the two halves. */
int ppo_xp_set_exchange(ppo_xp *h, int mode, void *nccl_comm);
/* The name SURVEY.md 8(b) gives this step: attach a ncclComm_t of `world` ranks (this process is
 * `rank`) = ppo_xp_set_exchange(h, 1, nccl_comm); a NULL communicator or world <= 1 detaches
 * (mode 0).  rank outside [0, world) => PPO_XP_EINVAL. */
int ppo_xp_attach_comm(ppo_xp *h, void *nccl_comm, int rank, int world);
double *ppo_xp_pass_sums_device(ppo_xp *h); /* [K][3] f64 */
int ppo_xp_pass_finish(ppo_xp *h);
/* RCCL bootstrap helpers for mode 1 (thin wrappers over ncclGetUniqueId / ncclCommInitRank /
 * ncclCommDestroy so that a host without RCCL bindings can still build the communicator). */
int ppo_xp_rccl_unique_id(void *out128);
int ppo_xp_rccl_comm_init(void **comm_out, const void *unique_id128, int rank, int world);
int ppo_xp_rccl_comm_count(void *comm, int *ranks_out); /* ncclCommCount: ranks the communicator really has */
int ppo_xp_rccl_comm_destroy(void *comm);

/* ---- reward normalisation, the step before add_experience (next row) ---- */

/* VecRunner's reward side (ppo/src/common/runner.rs:43-66: gamma 0.99, epsilon 1e-8, cliprew 10)
 * with its RunningMeanStd<Ix0> (ppo/src/common/running_mean_std.rs:16-34: mean 0, var 1, count 1e-4),
 * resident on the device. */
typedef struct ppo_xp_rnorm ppo_xp_rnorm;
int ppo_xp_rnorm_create(ppo_xp_rnorm **out, int device, size_t num_envs, float gamma, float epsilon, float cliprew);
void ppo_xp_rnorm_destroy(ppo_xp_rnorm *r);
/* VecRunner::normalized_rewards (runner.rs:114-127) on host slices (dones: Rust bools). */
int ppo_xp_rnorm_step(ppo_xp_rnorm *r, const float *rewards, const uint8_t *dones, float *out, size_t n);
/* The same on device arrays (dones already 0.0/1.0); enqueued on `stream` (NULL = the object's own). */
int ppo_xp_rnorm_step_device(ppo_xp_rnorm *r, const float *rewards_dev, const float *dones_f32_dev, float *out_dev,
                             size_t n, void *stream);
/* Running statistics (ret_rms.mean / .var / count) and, if ret_out != NULL, the n discounted returns. */
/* Env-sharded jobs: the batch moments of running_mean_std.rs:35-43 are taken over ALL ranks' envs.
 * step_begin updates the local returns (runner.rs:119) and leaves this rank's (count, sum, sum of
 * squares), f64, at ppo_xp_rnorm_sums_device(); the caller all-reduces (sum) those 3 doubles over the
 * ranks on the same stream; step_finish merges them into the running statistics (identical on every
 * rank), normalises and masks (runner.rs:120-124).  With one rank the result is bit-identical to
 * ppo_xp_rnorm_step_device.  No counterpart in the reference (single process). */
int ppo_xp_rnorm_step_begin_device(ppo_xp_rnorm *r, const float *rewards_dev, size_t n, void *stream);
double *ppo_xp_rnorm_sums_device(ppo_xp_rnorm *r);
int ppo_xp_rnorm_step_finish_device(ppo_xp_rnorm *r, const float *rewards_dev, const float *dones_f32_dev,
                                    float *out_dev, size_t n, void *stream);
int ppo_xp_rnorm_state(ppo_xp_rnorm *r, float *mean, float *var, float *count, float *ret_out, size_t n);

/* ---- plumbing --------------------------------------------------------------------------- */

void *ppo_xp_stream(const ppo_xp *h);            /* hipStream_t of the handle                */
int ppo_xp_set_stream(ppo_xp *h, void *stream);  /* borrow a caller-owned stream instead     */
int ppo_xp_sync(ppo_xp *h);                      /* hipStreamSynchronize of the handle's stream (and of the copy stream of a
                                                    pipelined host hand-off still in flight); also reports a one-launch
                                                    associative scan whose bounded wait gave up (PPO_XP_EIO)            */
int ppo_xp_device(const ppo_xp *h);

/* Per-kernel HIP-event timing on the handle's stream (bench.py's roofline numbers).
 * enable: 0 = off, -1 = every kernel, otherwise a bit mask (bit k = kernel id k).  While on, an
 * event pair is recorded around every launch of the selected kernels; collect synchronises, adds
 * up elapsed ms and launch counts per kernel id and resets. */
enum { PPO_XP_K_GAE = 0, PPO_XP_K_GAE_SUMMARY, PPO_XP_K_GAE_CARRY, PPO_XP_K_PACK, PPO_XP_K_GATHER,
       PPO_XP_K_FINALIZE, PPO_XP_K_NORMALIZE, PPO_XP_K_SCALARS, PPO_XP_K_EXCHANGE /* the library's own ncclAllReduce */,
       PPO_XP_K_COUNT };
int ppo_xp_prof_enable(ppo_xp *h, int enable);
int ppo_xp_prof_collect(ppo_xp *h, double *ms_out /*[K_COUNT]*/, uint64_t *launches_out /*[K_COUNT]*/);
const char *ppo_xp_kernel_name(int kernel_id);

/* Tunables (A/B measurements in one process).  Rows, scalar streams, returns and the sequential scan's results never
 * depend on them; standardised advantages are bit-identical for a fixed gather route and agree within 1 ulp of the
 * fp32 mean / std across routes (the fused and the split gathers tile the fp64 sums differently: 32-row against 256-row
 * tiles - identical bits were observed in every test, they are not guaranteed); PPO_XP_GAE_ASSOC has its documented
 * tolerance.  Changing a tunable ends a streamed pass in progress (its launch shapes were fixed at pass_begin). "gather_nt" 0/1, "gather_block" 64/256, "gather_xcd" 0/1/2, "fold_tiles" 0..65536 (single-GPU passes with at most this many 256-row tiles per
 * minibatch skip the finalize launch: the standardise workgroups add the tile sums themselves),
 * "gae_waves" 0 (auto)/1/2/4/8/16 (waves per 64 envs of the sequential scan), "gae_unroll" 0 (auto)/4/8/16/32,
 * "carry_waves" 4/8/16, "gae_nt", "summary_nt", "rec_nt", "out_nt" 0/1 (cache policy of the streams),
 * "gae_flat" 0/1 (1: the packing scan always stores the flat returns / advantages arrays too),
 * "fused" -1 (by size) / 0 / 1 (observation rows and scalar streams gathered by ONE kernel instead of two),
 * "fused_block" 256/512 (its workgroup size), "bij_mode" 0/1/2 (keyed-bijection passes: 0 = always write the index
 * array out first, 1 = one evaluation per output row handed to the row's threads through LDS, 2 = every thread evaluates
 * its own), "fused_iter" 0 (by size) / 1 / 2 / 4 (keyed-bijection passes: row groups per workgroup of the fused gather; with 4 a
 * workgroup covers 128 rows, copies four row groups per thread (one load in flight at a time) and adds up one statistics partial), "gae_onepass" -1 (by length) / 0 / 1 (PPO_XP_GAE_ASSOC as one launch), "op_l" 16/32/64 and "op_w" 2/4/8/16 (its
 * chunk length and chunks per workgroup; only some pairs are built),
 * "overlap" 0/1 (1: with the two-kernel path, the observation rows are gathered on a side stream of the handle while
 * the scalar streams, statistics and standardisation run on its own; joined before the pass is handed out).
 * Unknown names or values => PPO_XP_EINVAL. */
int ppo_xp_set_tunable(ppo_xp *h, const char *name, long value);

const char *ppo_xp_last_error(void);
const char *ppo_xp_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PPO_XP_H */
