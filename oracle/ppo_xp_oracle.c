/*
 * ppo_xp_oracle.c — CPU ORACLE for the experience path of flawedmatrix/ppo.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke check in
 * __graft_entry__.py and the `cpu_baseline` leg of bench.py may load it, and only as the
 * checker / the reported CPU baseline.  Nothing under ppo_amd/ imports, links or calls it.
 *
 * It is a plain-C, op-for-op fp32 restatement of the reference's Rust code (the reference
 * cannot be built here: no cargo/rustc, dependencies un-vendored).  Every
 * function cites the reference file:line it follows (paths relative to /root/reference).
 *
 * Pinning status:
 *   - buffer insert / ring wrap / flatten order / GAE: pinned by the reference's own test
 *     vectors (ppo/src/common/buffer.rs:233-458), see tests/test_oracle_golden.py.
 *   - batch windowing (count, coverage): pinned by ppo/src/model/data.rs:210-270.
 *   - advantage standardisation reduction ORDER: PARITY UNPINNED (lives in the un-vendored
 *     dfdx fork; no reference test checks a normalised value).  Restated from upstream
 *     dfdx 0.13 semantics: sequential fp32 sum, mean = sum * (1/n), var = mean((mean-x)^2).
 *   - shuffle: the reference uses rand 0.8.5 `SliceRandom::shuffle` on an OS-seeded
 *     ThreadRng (data.rs:94-95) => no reproducible permutation exists: PARITY UNPINNED for
 *     the specific permutation.  Restated here as rand 0.8.5's Fisher-Yates + its u32
 *     `gen_range` rejection sampler over a ChaCha12 stream seeded the way
 *     `ChaCha12Rng::seed_from_u64` does.  The ChaCha core is pinned at 20 rounds by the
 *     well-known all-zero key/nonce keystream vector (tests/test_oracle_golden.py).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC (oracle/Makefile).
 * -ffp-contract=off matters: ndarray rounds every elementwise op, so no FMA may be formed.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------ */
/* ExperienceBuffer — ppo/src/common/buffer.rs:23-39 (struct), :52-69 (new_with_params)  */
/* ------------------------------------------------------------------------------------ */
typedef struct orc_buffer {
  size_t counter, capacity, num_envs, obs_size;
  float gamma, lambda;
  float *obs;       /* [capacity, num_envs, obs_size]  buffer.rs:31 */
  float *rewards;   /* [capacity, num_envs]            buffer.rs:32 */
  uint64_t *actions;/* [capacity, num_envs] (usize)    buffer.rs:33 */
  float *values;    /* buffer.rs:34 */
  float *dones;     /* buffer.rs:35 */
  float *neglogps;  /* buffer.rs:36 */
  size_t last_values_idx; /* buffer.rs:38 */
} orc_buffer;

orc_buffer *orc_buffer_new(size_t capacity, size_t num_envs, size_t obs_size, float gamma,
                           float lambda) {
  orc_buffer *b = (orc_buffer *)calloc(1, sizeof(*b));
  if (!b) return NULL;
  b->capacity = capacity; b->num_envs = num_envs; b->obs_size = obs_size;
  b->gamma = gamma; b->lambda = lambda;
  size_t cn = capacity * num_envs;
  /* Array::zeros — buffer.rs:60-65 */
  b->obs = (float *)calloc((cn * obs_size) != 0 ? cn * obs_size : 1, sizeof(float));
  b->rewards = (float *)calloc(cn ? cn : 1, sizeof(float));
  b->actions = (uint64_t *)calloc(cn ? cn : 1, sizeof(uint64_t));
  b->values = (float *)calloc(cn ? cn : 1, sizeof(float));
  b->dones = (float *)calloc(cn ? cn : 1, sizeof(float));
  b->neglogps = (float *)calloc(cn ? cn : 1, sizeof(float));
  return b;
}

void orc_buffer_free(orc_buffer *b) {
  if (!b) return;
  free(b->obs); free(b->rewards); free(b->actions); free(b->values); free(b->dones);
  free(b->neglogps); free(b);
}

/* add_experience — buffer.rs:84-121.  Returns 0, or -1 where the reference panics on a
 * length mismatch (buffer.rs:97-112).  `n` is the length of every per-env slice. */
int orc_buffer_add(orc_buffer *b, const float *obs, const float *rewards,
                   const uint64_t *actions, const float *vals, const uint8_t *dones,
                   const float *neglogps, size_t n) {
  if (n != b->num_envs) return -1;
  size_t idx = b->counter % b->capacity;                          /* buffer.rs:93 */
  size_t N = b->num_envs, D = b->obs_size;
  memcpy(b->obs + idx * N * D, obs, N * D * sizeof(float));       /* buffer.rs:98 */
  memcpy(b->rewards + idx * N, rewards, N * sizeof(float));       /* buffer.rs:101 */
  memcpy(b->actions + idx * N, actions, N * sizeof(uint64_t));    /* buffer.rs:104 */
  memcpy(b->values + idx * N, vals, N * sizeof(float));           /* buffer.rs:107 */
  for (size_t e = 0; e < N; ++e)                                  /* buffer.rs:95,110 */
    b->dones[idx * N + e] = (float)(uint8_t)(dones[e] != 0);
  memcpy(b->neglogps + idx * N, neglogps, N * sizeof(float));     /* buffer.rs:113 */
  b->last_values_idx = idx;                                       /* buffer.rs:115 */
  b->counter += 1;                                                /* buffer.rs:117 */
  if (b->counter >= b->capacity)                                  /* buffer.rs:118-120 */
    b->counter = (b->counter % b->capacity) + b->capacity;
  return 0;
}

/* len — buffer.rs:123-129 */
size_t orc_buffer_len(const orc_buffer *b) {
  return b->counter >= b->capacity ? b->capacity : b->counter;
}

/* reset_counter — buffer.rs:131-133 (data and last_values_idx are left alone) */
void orc_buffer_reset_counter(orc_buffer *b) { b->counter = 0; }

size_t orc_buffer_last_values_idx(const orc_buffer *b) { return b->last_values_idx; }

/* training_views — buffer.rs:142-175: zero-copy flatten of rows 0..len in STORAGE order.
 * Here: raw pointers to the first element of each stream; flat index i = t*N + e. */
void orc_buffer_views(const orc_buffer *b, const float **obs, const uint64_t **actions,
                      const float **values, const float **neglogps, size_t *num_steps) {
  *obs = b->obs; *actions = b->actions; *values = b->values; *neglogps = b->neglogps;
  *num_steps = orc_buffer_len(b) * b->num_envs;
}

/* Raw stream access for tests (rewards / dones are not part of training_views). */
const float *orc_buffer_rewards(const orc_buffer *b) { return b->rewards; }
const float *orc_buffer_dones(const orc_buffer *b) { return b->dones; }

/* returns — buffer.rs:187-225.  GAE(lambda) reverse-time scan, every fp32 op rounded on its
 * own, evaluation order exactly as Rust parses buffer.rs:217-219:
 *   delta      = (rewards[t] + ((nextvalues * nextnonterminal) * gamma)) - values[t]
 *   lastgaelam = delta + ((nextnonterminal * lastgaelam) * (gamma * lambda))
 *   res        = advs + values                                   (buffer.rs:222)
 * The bootstrap is values.row(last_values_idx) (buffer.rs:204), not a fresh critic value.
 * Returns -1 where the reference panics (last_dones.len() != NUM_ENVS, buffer.rs:194). */
int orc_gae_rows(const float *rewards, const float *values, const float *dones, size_t len,
                 size_t N, size_t last_values_idx, const uint8_t *last_dones, float gamma,
                 float lambda, float *out_returns, int nthreads) {
  const float gl = gamma * lambda;                                /* buffer.rs:219 */
  (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static)
#endif
  for (long e = 0; e < (long)N; ++e) {
    float lastgaelam = 0.0f;                                      /* buffer.rs:196 */
    for (size_t tt = len; tt-- > 0;) {                            /* buffer.rs:206 */
      float nextvalues, nextnonterminal;
      if (tt == len - 1) {                                        /* buffer.rs:209-210 */
        nextvalues = values[last_values_idx * N + e];             /* buffer.rs:204 */
        nextnonterminal = (float)(uint8_t)(!last_dones[e]);       /* buffer.rs:201 */
      } else {                                                    /* buffer.rs:212-213 */
        nextvalues = values[(tt + 1) * N + e];
        nextnonterminal = 1.0f - dones[(tt + 1) * N + e];         /* buffer.rs:199 */
      }
      float x1 = nextvalues * nextnonterminal;
      float x2 = x1 * gamma;
      float x3 = rewards[tt * N + e] + x2;
      float delta = x3 - values[tt * N + e];                      /* buffer.rs:217-218 */
      float y1 = nextnonterminal * lastgaelam;
      float y2 = y1 * gl;
      lastgaelam = delta + y2;                                    /* buffer.rs:219 */
      out_returns[tt * N + e] = lastgaelam + values[tt * N + e];  /* buffer.rs:220,222 */
    }
  }
  return 0;
}

int orc_buffer_returns(const orc_buffer *b, const uint8_t *last_dones, size_t n,
                       float *out_returns) {
  if (n != b->num_envs) return -1;                                /* buffer.rs:194 */
  return orc_gae_rows(b->rewards, b->values, b->dones, orc_buffer_len(b), b->num_envs,
                      b->last_values_idx, last_dones, b->gamma, b->lambda, out_returns, 1);
}

/* fp64 companion of the same recurrence (NOT the reference; used only to state error bounds
 * of the associative-scan kernel against exact arithmetic). */
void orc_gae_rows_f64(const float *rewards, const float *values, const float *dones,
                      size_t len, size_t N, size_t last_values_idx,
                      const uint8_t *last_dones, float gamma, float lambda,
                      double *out_returns) {
  const double g = gamma, gl = (double)gamma * (double)lambda;
  for (size_t e = 0; e < N; ++e) {
    double A = 0.0;
    for (size_t tt = len; tt-- > 0;) {
      double nv, nnt;
      if (tt == len - 1) { nv = values[last_values_idx * N + e]; nnt = last_dones[e] ? 0.0 : 1.0; }
      else { nv = values[(tt + 1) * N + e]; nnt = 1.0 - (double)dones[(tt + 1) * N + e]; }
      double delta = rewards[tt * N + e] + nv * nnt * g - values[tt * N + e];
      A = delta + nnt * A * gl;
      out_returns[tt * N + e] = A + values[tt * N + e];
    }
  }
}

/* ------------------------------------------------------------------------------------ */
/* ExperienceBatcher — ppo/src/model/data.rs                                              */
/* ------------------------------------------------------------------------------------ */

/* advantages = returns - values over the whole buffer — data.rs:62 (so adv' is the doubly
 * rounded fl(fl(A+v)-v), not A). */
void orc_advantages(const float *returns, const float *values, size_t M, float *out,
                    int nthreads) {
  (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static)
#endif
  for (long i = 0; i < (long)M; ++i) out[i] = returns[i] - values[i];
}

/* Number of batches one pass of the iterator yields — data.rs:115-126:
 * None when batch_size > buf_size, else ceil(buf_size / batch_size). */
size_t orc_num_batches(size_t buf_size, size_t batch_size) {
  if (batch_size == 0 || batch_size > buf_size) return 0;
  return (buf_size + batch_size - 1) / batch_size;
}

/* The window of the permutation consumed by one `next()` — data.rs:117-126.
 * Returns 1 and writes [start,end) when a batch is produced, 0 for `None`.
 * *iter_idx is advanced exactly as data.rs:126 does. */
int orc_batch_window(size_t *iter_idx, size_t batch_size, size_t buf_size, size_t *start,
                     size_t *end) {
  if (batch_size > buf_size || *iter_idx >= buf_size) return 0;   /* data.rs:117-119 */
  if (*iter_idx + batch_size > buf_size) {                        /* data.rs:121-122 */
    *start = buf_size - batch_size; *end = buf_size;
  } else {                                                        /* data.rs:124 */
    *start = *iter_idx; *end = *iter_idx + batch_size;
  }
  *iter_idx += batch_size;                                        /* data.rs:126 */
  return 1;
}

/* gather — data.rs:137-156: out[j,:] = src[idx[j],:] (dfdx `gather` along axis 0). */
void orc_gather_f32(const float *src, size_t row_len, const uint32_t *idx, size_t n,
                    float *out, int nthreads) {
  (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static)
#endif
  for (long j = 0; j < (long)n; ++j)
    memcpy(out + (size_t)j * row_len, src + (size_t)idx[j] * row_len, row_len * sizeof(float));
}

void orc_gather_u64(const uint64_t *src, const uint32_t *idx, size_t n, uint64_t *out,
                    int nthreads) {
  (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static)
#endif
  for (long j = 0; j < (long)n; ++j) out[j] = src[idx[j]];
}

/* Advantage standardisation of one gathered minibatch — data.rs:157-164 + correct_std
 * data.rs:196-200.  dfdx semantics restated (upstream 0.13; the pinned fork is un-vendored,
 * so the reduction order is PARITY UNPINNED):
 *   mean = (sequential fp32 sum) * (1/n)              [dfdx mean = sum().mul(1/n)]
 *   var  = mean((mean - x)^2)  (population, two-pass) [dfdx var]
 *   std  = sqrt(var * (n / (n - 1)))  with n as f32   [data.rs:197-199]
 *   out  = (x - mean) / (std + 1e-8)                  [data.rs:163]
 * n == 1 gives 0 * inf = NaN exactly as the reference would. */
void orc_standardize(const float *a, size_t n, float *out, float *mean_out, float *std_out) {
  const float inv_n = (float)(1.0 / (double)n);
  float sum = 0.0f;
  for (size_t j = 0; j < n; ++j) sum += a[j];
  const float mean = sum * inv_n;
  float sq = 0.0f;
  for (size_t j = 0; j < n; ++j) { float d = mean - a[j]; sq += d * d; }
  const float var = sq * inv_n;
  const float nf = (float)n;
  const float corrected = var * (nf / (nf - 1.0f));               /* data.rs:198 */
  const float std = sqrtf(corrected);                             /* data.rs:199 */
  const float denom = std + 1e-8f;                                /* data.rs:163 */
  for (size_t j = 0; j < n; ++j) out[j] = (a[j] - mean) / denom;
  if (mean_out) *mean_out = mean;
  if (std_out) *std_out = std;
}

/* fp64 companion (NOT the reference): exact-ish statistics, used to state how far both the
 * fp32-sequential restatement and the GPU kernel are from exact arithmetic. */
void orc_standardize_f64(const float *a, size_t n, double *out, double *mean_out,
                         double *std_out) {
  long double s = 0.0L;
  for (size_t j = 0; j < n; ++j) s += a[j];
  const double mean = (double)(s / (long double)n);
  long double q = 0.0L;
  for (size_t j = 0; j < n; ++j) { long double d = (long double)a[j] - mean; q += d * d; }
  const double var = (double)(q / (long double)n);
  const double std = sqrt(var * ((double)n / ((double)n - 1.0)));
  for (size_t j = 0; j < n; ++j) out[j] = ((double)a[j] - mean) / (std + 1e-8);
  if (mean_out) *mean_out = mean;
  if (std_out) *std_out = std;
}

/* Sharded (multi-rank) standardisation companion: statistics over the CONCATENATION of
 * `parts` local minibatches, applied to part `which` (SURVEY.md 8(e) parity definition). */
void orc_standardize_concat(const float *const *parts, const size_t *lens, size_t nparts,
                            size_t which, float *out) {
  size_t n = 0;
  for (size_t p = 0; p < nparts; ++p) n += lens[p];
  float *cat = (float *)malloc((n ? n : 1) * sizeof(float));
  float *tmp = (float *)malloc((n ? n : 1) * sizeof(float));
  size_t o = 0, off = 0;
  for (size_t p = 0; p < nparts; ++p) {
    if (p == which) off = o;
    memcpy(cat + o, parts[p], lens[p] * sizeof(float)); o += lens[p];
  }
  orc_standardize(cat, n, tmp, NULL, NULL);
  memcpy(out, tmp + off, lens[which] * sizeof(float));
  free(cat); free(tmp);
}

/* ------------------------------------------------------------------------------------ */
/* shuffle — data.rs:91-96 (`indices.shuffle(&mut thread_rng())`), rand 0.8.5 semantics    */
/* ------------------------------------------------------------------------------------ */

#define ORC_ROTL32(x, n) (((x) << (n)) | ((x) >> (32 - (n))))
#define ORC_QR(a, b, c, d)                                                                \
  a += b; d ^= a; d = ORC_ROTL32(d, 16); c += d; b ^= c; b = ORC_ROTL32(b, 12);           \
  a += b; d ^= a; d = ORC_ROTL32(d, 8);  c += d; b ^= c; b = ORC_ROTL32(b, 7);

/* One 64-byte ChaCha block (djb variant: 64-bit block counter in words 12-13, 64-bit stream
 * id in words 14-15), `rounds` = 8/12/20.  rand_chacha 0.3.1 emits these blocks in counter
 * order, 16 little-endian u32 words each. */
void orc_chacha_block(const uint32_t key[8], uint64_t counter, uint64_t stream, int rounds,
                      uint32_t out[16]) {
  uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u,
                    key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                    (uint32_t)counter, (uint32_t)(counter >> 32),
                    (uint32_t)stream, (uint32_t)(stream >> 32)};
  uint32_t x[16];
  memcpy(x, s, sizeof(x));
  for (int r = 0; r < rounds; r += 2) {
    ORC_QR(x[0], x[4], x[8], x[12]) ORC_QR(x[1], x[5], x[9], x[13])
    ORC_QR(x[2], x[6], x[10], x[14]) ORC_QR(x[3], x[7], x[11], x[15])
    ORC_QR(x[0], x[5], x[10], x[15]) ORC_QR(x[1], x[6], x[11], x[12])
    ORC_QR(x[2], x[7], x[8], x[13]) ORC_QR(x[3], x[4], x[9], x[14])
  }
  for (int i = 0; i < 16; ++i) out[i] = x[i] + s[i];
}

/* rand_core 0.6 `SeedableRng::seed_from_u64`: PCG32 expands the u64 into the 32-byte key. */
void orc_seed_from_u64(uint64_t state, uint32_t key[8]) {
  for (int i = 0; i < 8; ++i) {
    state = state * 6364136223846793005ULL + 11634580027462260723ULL;
    uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
    uint32_t rot = (uint32_t)(state >> 59);
    key[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
  }
}

typedef struct { uint32_t key[8]; uint64_t counter; uint32_t buf[16]; int pos; } orc_rng;

static void orc_rng_init(orc_rng *g, uint64_t seed) {
  orc_seed_from_u64(seed, g->key); g->counter = 0; g->pos = 16;
}
static uint32_t orc_rng_next_u32(orc_rng *g) {
  if (g->pos == 16) { orc_chacha_block(g->key, g->counter++, 0, 12, g->buf); g->pos = 0; }
  return g->buf[g->pos++];
}

/* First `n` u32 outputs of the seeded stream (for tests). */
void orc_rng_stream(uint64_t seed, size_t n, uint32_t *out) {
  orc_rng g; orc_rng_init(&g, seed);
  for (size_t i = 0; i < n; ++i) out[i] = orc_rng_next_u32(&g);
}

/* rand 0.8.5 UniformInt<u32>::sample_single(0, range): widening-multiply with the
 * "conservative zone" rejection test. */
static uint32_t orc_gen_range_u32(orc_rng *g, uint32_t range) {
  uint32_t zone = (range << __builtin_clz(range)) - 1u;
  for (;;) {
    uint32_t v = orc_rng_next_u32(g);
    uint64_t m = (uint64_t)v * (uint64_t)range;
    if ((uint32_t)m <= zone) return (uint32_t)(m >> 32);
  }
}

/* indices = 0..M; indices.shuffle(rng) — data.rs:93-95; rand 0.8.5 SliceRandom::shuffle:
 *   for i in (1..len).rev() { swap(i, gen_index(rng, i + 1)) }.   Requires M < 2^32. */
int orc_shuffle(uint64_t seed, size_t M, uint32_t *indices) {
  if (M > 0xFFFFFFFFull) return -1;
  for (size_t i = 0; i < M; ++i) indices[i] = (uint32_t)i;
  orc_rng g; orc_rng_init(&g, seed);
  for (size_t i = M; i-- > 1;) {
    uint32_t j = orc_gen_range_u32(&g, (uint32_t)(i + 1));
    uint32_t tmp = indices[i]; indices[i] = indices[j]; indices[j] = tmp;
  }
  return 0;
}

/* ------------------------------------------------------------------------------------ */
/* Whole path, for the CPU baseline of bench.py: GAE + adv' + K x (6 gathers + standardise) */
/* following training.rs:186-205's call pattern for ONE shuffled pass.  Outputs go to      */
/* caller-provided slabs of K*B rows.  nthreads = 1 is faithful to the single-threaded     */
/* reference; > 1 parallelises over envs / rows (generous baseline).                        */
/* ------------------------------------------------------------------------------------ */
int orc_run_path(const float *obs, const float *rewards, const uint64_t *actions,
                 const float *values, const float *dones, const float *neglogps, size_t len,
                 size_t N, size_t D, size_t last_values_idx, const uint8_t *last_dones,
                 float gamma, float lambda, const uint32_t *perm, size_t B,
                 float *returns /*[M]*/, float *adv /*[M]*/, float *obs_out /*[K*B,D]*/,
                 uint64_t *act_out, float *val_out, float *nlp_out, float *ret_out,
                 float *adv_out, int nthreads) {
  const size_t M = len * N;
  orc_gae_rows(rewards, values, dones, len, N, last_values_idx, last_dones, gamma, lambda,
               returns, nthreads);
  orc_advantages(returns, values, M, adv, nthreads);
  size_t iter = 0, start, end, k = 0;
  float *tmp = (float *)malloc((B ? B : 1) * sizeof(float));
  while (orc_batch_window(&iter, B, M, &start, &end)) {
    const uint32_t *idx = perm + start;
    orc_gather_f32(obs, D, idx, B, obs_out + k * B * D, nthreads);
    orc_gather_u64(actions, idx, B, act_out + k * B, nthreads);
    orc_gather_f32(values, 1, idx, B, val_out + k * B, nthreads);
    orc_gather_f32(neglogps, 1, idx, B, nlp_out + k * B, nthreads);
    orc_gather_f32(returns, 1, idx, B, ret_out + k * B, nthreads);
    orc_gather_f32(adv, 1, idx, B, tmp, nthreads);
    orc_standardize(tmp, B, adv_out + k * B, NULL, NULL);
    ++k;
  }
  free(tmp);
  return (int)k;
}

int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------------------------ */
/* Next rows                                                             */
/* ------------------------------------------------------------------------------------ */

/* CPU twin of the DEVICE permutation generator (not in the reference, whose shuffle is the host
 * Fisher-Yates above): key_i = i-th output of SplitMix64 started at `seed`; perm = stable argsort.
 * Implemented as an LSD radix sort on (key, index) so that ties keep index order. */
static uint64_t orc_splitmix64_at(uint64_t seed, uint64_t i) {
  uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

int orc_keyed_shuffle(uint64_t seed, size_t M, uint32_t *indices) {
  if (M > 0xFFFFFFFFull) return -1;
  uint64_t *k0 = (uint64_t *)malloc((M ? M : 1) * 8), *k1 = (uint64_t *)malloc((M ? M : 1) * 8);
  uint32_t *v0 = (uint32_t *)malloc((M ? M : 1) * 4), *v1 = (uint32_t *)malloc((M ? M : 1) * 4);
  for (size_t i = 0; i < M; ++i) { k0[i] = orc_splitmix64_at(seed, i); v0[i] = (uint32_t)i; }
  for (int pass = 0; pass < 4; ++pass) {
    size_t *cnt = (size_t *)calloc(65537, sizeof(size_t));
    const int sh = 16 * pass;
    for (size_t i = 0; i < M; ++i) cnt[((k0[i] >> sh) & 0xFFFF) + 1]++;
    for (size_t b = 0; b < 65536; ++b) cnt[b + 1] += cnt[b];
    for (size_t i = 0; i < M; ++i) { size_t d = cnt[(k0[i] >> sh) & 0xFFFF]++; k1[d] = k0[i]; v1[d] = v0[i]; }
    free(cnt);
    uint64_t *tk = k0; k0 = k1; k1 = tk;
    uint32_t *tv = v0; v0 = v1; v1 = tv;
  }
  memcpy(indices, v0, M * 4);
  free(k0); free(k1); free(v0); free(v1);
  return 0;
}

/* CPU twin of the KEYED BIJECTION pass (ppo_xp_pass_begin_bijection; not in the reference, whose
 * shuffle is the host Fisher-Yates above, data.rs:91-96).  perm(j) is evaluated from (seed, M) alone:
 * six alternating Feistel steps on n = max(8, ceil(log2 M)) bits, x = (a << l) | b with l = n / 2,
 *   step r even: a ^= f(b, key_r) & hmask      step r odd: b ^= f(a, key_r) & lmask
 *   f(v, k) = mix of two 24-bit multiplies (below),  key_r = high half of the r-th SplitMix64 output,
 * cycle-walked into [0, M).  Each step is its own inverse, so perm^-1 runs them backwards.
 * Written from that specification, independently of the device header. */
#define ORC_BIJ_ROUNDS 6
typedef struct { uint32_t M, l, hmask, lmask, key[ORC_BIJ_ROUNDS]; } orc_bij;

static uint32_t orc_mul24(uint32_t a, uint32_t b) {
  return (uint32_t)(((uint64_t)(a & 0xFFFFFFu) * (uint64_t)(b & 0xFFFFFFu)) & 0xFFFFFFFFull);
}

static uint32_t orc_bij_round_fn(uint32_t v, uint32_t key) {
  uint32_t t = v ^ (key & 0xFFFFu);
  t = orc_mul24(t, 0xE995u);
  t = t ^ (t >> 15);
  t = orc_mul24(t, 0x2C1B3u);
  t = t ^ (t >> 13);
  return t + (key >> 16);
}

static void orc_bij_init(orc_bij *b, uint64_t seed, size_t M) {
  unsigned n = 8;
  while (n < 32 && ((uint64_t)1 << n) < (uint64_t)M) n++;
  const unsigned l = n / 2, h = n - l;
  b->M = (uint32_t)M; b->l = l;
  b->hmask = (uint32_t)(((uint64_t)1 << h) - 1);
  b->lmask = (uint32_t)(((uint64_t)1 << l) - 1);
  for (unsigned r = 0; r < ORC_BIJ_ROUNDS; ++r) b->key[r] = (uint32_t)(orc_splitmix64_at(seed, r) >> 32);
}

static uint32_t orc_bij_network(const orc_bij *b, uint32_t x, int backwards) {
  uint32_t hi = x >> b->l, lo = x & b->lmask;
  for (int s = 0; s < ORC_BIJ_ROUNDS; ++s) {
    const int r = backwards ? ORC_BIJ_ROUNDS - 1 - s : s;
    if (r % 2 == 0) hi ^= orc_bij_round_fn(lo, b->key[r]) & b->hmask;
    else lo ^= orc_bij_round_fn(hi, b->key[r]) & b->lmask;
  }
  return (hi << b->l) | lo;
}

/* out[j] = perm(j) (inverse = 0) or perm^-1(j) (inverse = 1) for j in [first, first + count). */
int orc_bijection_range(uint64_t seed, size_t M, size_t first, size_t count, int inverse, uint32_t *out) {
  if (M == 0 || M > 0xFFFFFFFFull || first + count > M) return -1;
  orc_bij b; orc_bij_init(&b, seed, M);
  for (size_t j = 0; j < count; ++j) {
    uint32_t x = (uint32_t)(first + j);
    do x = orc_bij_network(&b, x, inverse); while (x >= b.M);
    out[j] = x;
  }
  return 0;
}

int orc_bijection_perm(uint64_t seed, size_t M, uint32_t *indices) {
  if (M == 0) return 0;
  return orc_bijection_range(seed, M, 0, M, 0, indices);
}

/* ndarray 0.15.6 `var_axis(Axis(0), ddof)` on a 1-D f32 array (used by explained_variance,
 * ppo/src/training.rs:91,96): sequential Welford with a fused multiply-add,
 *   delta = x - mean; mean += delta / count; sum_sq = (x - mean).mul_add(delta, sum_sq); -> sum_sq / (n - ddof).
 * ndarray is not on disk: restated from the published crate source, PARITY UNPINNED. */
float orc_var_welford_f32(const float *x, size_t n, float ddof) {
  float mean = 0.0f, sum_sq = 0.0f;
  for (size_t i = 0; i < n; ++i) {
    const float count = (float)(i + 1);
    const float delta = x[i] - mean;
    mean = mean + delta / count;
    sum_sq = fmaf(x[i] - mean, delta, sum_sq);
  }
  return sum_sq / ((float)n - ddof);
}

/* explained_variance — ppo/src/training.rs:88-99 (predictions = values, targets = returns).
 * Note `0.0 -`, not `1.0 -`, and the guard against -1.0: reproduced as written. */
float orc_explained_variance(const float *values, const float *returns, size_t n) {
  const float target_variance = orc_var_welford_f32(returns, n, 0.0f);     /* training.rs:91 */
  if (target_variance == -1.0f) return NAN;                                 /* training.rs:92 */
  float *diff = (float *)malloc((n ? n : 1) * sizeof(float));
  for (size_t i = 0; i < n; ++i) diff[i] = returns[i] - values[i];         /* training.rs:95 */
  const float diff_variance = orc_var_welford_f32(diff, n, 0.0f);          /* training.rs:96 */
  free(diff);
  return 0.0f - (diff_variance / target_variance);                          /* training.rs:97 */
}

/* fp64 companion (NOT the reference). */
double orc_explained_variance_f64(const float *values, const float *returns, size_t n) {
  long double sr = 0, sd = 0;
  for (size_t i = 0; i < n; ++i) { sr += returns[i]; sd += (double)(returns[i] - values[i]); }
  const long double mr = sr / n, md = sd / n;
  long double qr = 0, qd = 0;
  for (size_t i = 0; i < n; ++i) {
    long double a = returns[i] - mr, b = (double)(returns[i] - values[i]) - md;
    qr += a * a; qd += b * b;
  }
  return (double)(0.0L - (qd / n) / (qr / n));
}

/* ------------------------------------------------------------------------------------ */
/* Reward normalisation (next row): ppo/src/common/runner.rs:114-127 +                      */
/* ppo/src/common/running_mean_std.rs:16-67 (RunningMeanStd<Ix0>)                           */
/* ------------------------------------------------------------------------------------ */
typedef struct orc_rnorm {
  size_t N;
  float gamma, epsilon, cliprew;
  float *ret;                 /* runner.rs:63 zeros */
  float mean, var, count;     /* running_mean_std.rs:29-33: 0, 1, epsilon = 1e-4 */
} orc_rnorm;

orc_rnorm *orc_rnorm_new(size_t N, float gamma, float epsilon, float cliprew) {
  orc_rnorm *r = (orc_rnorm *)calloc(1, sizeof(*r));
  r->N = N; r->gamma = gamma; r->epsilon = epsilon; r->cliprew = cliprew;
  r->ret = (float *)calloc(N ? N : 1, sizeof(float));
  r->mean = 0.0f; r->var = 1.0f; r->count = 1e-4f;
  return r;
}
void orc_rnorm_free(orc_rnorm *r) { if (r) { free(r->ret); free(r); } }

/* ndarray 0.15.6 numeric_util::unrolled_fold as used by `sum()` of a contiguous lane (restated from
 * the published crate source; ndarray is not on disk: PARITY UNPINNED). */
static float orc_ndarray_sum(const float *x, size_t n) {
  float acc = 0.0f, p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  size_t i = 0;
  for (; i + 8 <= n; i += 8) for (int k = 0; k < 8; ++k) p[k] = p[k] + x[i + k];
  acc = acc + (p[0] + p[4]); acc = acc + (p[1] + p[5]); acc = acc + (p[2] + p[6]); acc = acc + (p[3] + p[7]);
  for (; i < n; ++i) acc = acc + x[i];
  return acc;
}

/* normalized_rewards — runner.rs:114-127.  Returns -1 on a length mismatch. */
int orc_rnorm_step(orc_rnorm *r, const float *rews, const uint8_t *dones, float *out, size_t n) {
  if (n != r->N) return -1;
  const size_t N = r->N;
  for (size_t e = 0; e < N; ++e) r->ret[e] = r->ret[e] * r->gamma + rews[e];      /* runner.rs:119 */
  /* ret_rms.update(&ret) — running_mean_std.rs:35-43 */
  const float batch_mean = orc_ndarray_sum(r->ret, N) / (float)N;                 /* mean_axis */
  const float batch_var = orc_var_welford_f32(r->ret, N, 0.0f);                   /* var_axis(_, 0.) */
  const float b_count = (float)N;
  /* update_from_moments — running_mean_std.rs:45-67 */
  const float delta = batch_mean - r->mean;
  const float tot_count = r->count + b_count;
  const float new_mean = r->mean + delta * b_count / tot_count;
  const float m_a = r->var * r->count;
  const float m_b = batch_var * b_count;
  const float m2 = m_a + m_b + (delta * delta) * r->count * b_count / tot_count;
  const float new_var = m2 / tot_count;
  r->mean = new_mean; r->var = new_var; r->count = tot_count;
  const float denom = sqrtf(r->var + r->epsilon);                                 /* runner.rs:121 */
  for (size_t e = 0; e < N; ++e) {
    float x = rews[e] / denom;
    x = x < -r->cliprew ? -r->cliprew : (x > r->cliprew ? r->cliprew : x);        /* runner.rs:122 */
    out[e] = x;
    r->ret[e] = r->ret[e] * (float)(uint8_t)(dones[e] != 0);                      /* runner.rs:124 (sic) */
  }
  return 0;
}
void orc_rnorm_state(const orc_rnorm *r, float *mean, float *var, float *count, float *ret) {
  *mean = r->mean; *var = r->var; *count = r->count;
  if (ret) memcpy(ret, r->ret, r->N * sizeof(float));
}
