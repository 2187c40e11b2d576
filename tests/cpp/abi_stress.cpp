// Walks the whole C ABI of libppo_xp.so from compiled host code: several handles, ring wrap,
// every GAE algorithm, every way to start a pass, the error returns where the reference panics
// (buffer.rs:97-112,194; data.rs:117,130), profiling, tunables, the reward normaliser.  Checks are
// self-consistency checks (gathered rows == source[perm], standardised mean 0 / std 1); bit parity
// with the oracle is the Python suite's job.  Also the program that is built with host
// AddressSanitizer / LeakSanitizer (DESIGN.md "Verification"): every host allocation the library
// makes on these paths is exercised and released.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

#include "ppo_xp.h"

#define REQUIRE(c)                                                                         \
  do {                                                                                     \
    if (!(c)) {                                                                            \
      std::fprintf(stderr, "%s:%d: REQUIRE(%s) failed; last error: %s\n", __FILE__, __LINE__, #c, \
                   ppo_xp_last_error());                                                   \
      std::exit(1);                                                                        \
    }                                                                                      \
  } while (0)

template <typename T>
static std::vector<T> from_device(const T *p, size_t n) {
  std::vector<T> v(n);
  REQUIRE(hipMemcpy(v.data(), p, n * sizeof(T), hipMemcpyDeviceToHost) == hipSuccess);
  return v;
}

struct Rollout {
  size_t steps, N, D;
  std::vector<float> obs, rewards, values, neglogps;
  std::vector<size_t> actions;
  std::vector<uint8_t> dones, last_dones;
  Rollout(size_t steps_, size_t N_, size_t D_, unsigned seed) : steps(steps_), N(N_), D(D_) {
    std::mt19937 g(seed);
    std::normal_distribution<float> n01(0.f, 1.f);
    std::uniform_real_distribution<float> u(-1.f, 1.f);
    obs.resize(steps * N * D); rewards.resize(steps * N); values.resize(steps * N); neglogps.resize(steps * N);
    actions.resize(steps * N); dones.resize(steps * N); last_dones.resize(N);
    for (auto &x : obs) x = u(g);
    for (auto &x : rewards) x = n01(g);
    for (auto &x : values) x = n01(g);
    for (auto &x : neglogps) x = std::fabs(n01(g));
    for (auto &x : actions) x = g() % 18;
    for (auto &x : dones) x = (g() % 50) == 0;
    for (auto &x : last_dones) x = (g() % 50) == 0;
  }
};

// one pass: every minibatch row equals the source row the permutation names; advantages standardised
static void check_pass(ppo_xp *h, size_t B, size_t expect_K) {
  ppo_xp_views_t v;
  REQUIRE(ppo_xp_views(h, &v) == 0);
  const size_t M = v.num_steps, D = v.obs_size;
  REQUIRE(ppo_xp_pass_num_batches(h) == expect_K);
  std::vector<uint32_t> perm(M);
  REQUIRE(ppo_xp_pass_perm_host(h, perm.data(), M) == 0);
  std::vector<uint8_t> seen(M, 0);
  for (uint32_t p : perm) { REQUIRE(p < M); seen[p] = 1; }
  REQUIRE(std::accumulate(seen.begin(), seen.end(), size_t(0)) == M);       // a bijection (data.rs:93-95)
  REQUIRE(ppo_xp_sync(h) == 0);
  auto obs = from_device(v.observations, M * D);
  auto act = from_device(v.actions, M);
  auto ret = from_device(v.returns, M);
  std::vector<float> stats(2 * expect_K);
  if (expect_K) REQUIRE(ppo_xp_pass_stats_host(h, stats.data(), expect_K) == 0);
  ppo_batch_view b;
  size_t k = 0;
  int rc;
  while ((rc = ppo_xp_pass_next(h, &b)) == 0) {
    REQUIRE(b.index == k && b.batch_size == B && b.obs_size == D);
    const size_t start = (k * B + B > M) ? M - B : k * B;                    // data.rs:121-122
    auto o = from_device(b.observations, B * D);
    auto a = from_device(b.actions, B);
    auto r = from_device(b.returns, B);
    auto adv = from_device(b.advantages, B);
    double s = 0, q = 0;
    for (size_t j = 0; j < B; ++j) {
      const uint32_t p = perm[start + j];
      for (size_t d = 0; d < D; ++d) REQUIRE(o[j * D + d] == obs[(size_t)p * D + d]);
      REQUIRE(a[j] == act[p]);
      REQUIRE(r[j] == ret[p]);
      s += adv[j]; q += (double)adv[j] * adv[j];
    }
    if (B > 1 && stats[2 * k + 1] > 1e-3f) {
      const double mean = s / B, var = (q - s * mean) / (B - 1);
      REQUIRE(std::fabs(mean) < 1e-4 && std::fabs(std::sqrt(var) - 1.0) < 1e-3);
    }
    ++k;
  }
  REQUIRE(rc == 1 && k == expect_K);
  REQUIRE(ppo_xp_pass_next(h, &b) == 1);                                     // stays exhausted
}

int main() {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { std::fprintf(stderr, "no GPU\n"); return 2; }

  // ---- argument errors change nothing -------------------------------------------------------
  ppo_xp *bad = nullptr;
  REQUIRE(ppo_xp_create(&bad, 0, 0, 4, 4, 0.99f, 0.95f) == PPO_XP_EINVAL && bad == nullptr);
  REQUIRE(ppo_xp_create(&bad, 9999, 4, 4, 4, 0.99f, 0.95f) == PPO_XP_ENODEV && bad == nullptr);
  REQUIRE(ppo_xp_create(nullptr, 0, 4, 4, 4, 0.99f, 0.95f) == PPO_XP_EINVAL);
  ppo_xp_destroy(nullptr);

  const size_t N = 70, D = 12, cap = 16;
  ppo_xp *h = nullptr, *h2 = nullptr;
  REQUIRE(ppo_xp_create(&h, 0, cap, N, D, 0.99f, 0.95f) == 0);
  REQUIRE(ppo_xp_create(&h2, 0, 5, 3, 7, 0.9f, 0.8f) == 0);
  REQUIRE(ppo_xp_device(h) == 0 && ppo_xp_stream(h) != nullptr && ppo_xp_stream(h) != ppo_xp_stream(h2));

  Rollout r(37, N, D, 1);     // 37 steps into 16 rows: wraps twice
  REQUIRE(ppo_xp_add(h, r.obs.data(), r.rewards.data(), r.actions.data(), r.values.data(), r.dones.data(),
                     r.neglogps.data(), N - 1) == PPO_XP_EINVAL);            // buffer.rs:97
  REQUIRE(ppo_xp_len(h) == 0);
  ppo_batch_view bv;
  REQUIRE(ppo_xp_pass_begin(h, 8, nullptr, 1) == PPO_XP_ESTATE);             // no returns yet
  REQUIRE(ppo_xp_pass_next(h, &bv) < 0);
  for (size_t t = 0; t < 5; ++t)
    REQUIRE(ppo_xp_add(h, &r.obs[t * N * D], &r.rewards[t * N], &r.actions[t * N], &r.values[t * N], &r.dones[t * N],
                       &r.neglogps[t * N], N) == 0);
  REQUIRE(ppo_xp_len(h) == 5 && ppo_xp_counter(h) == 5 && ppo_xp_last_values_idx(h) == 4);
  REQUIRE(ppo_xp_add_many(h, &r.obs[5 * N * D], &r.rewards[5 * N], &r.actions[5 * N], &r.values[5 * N], &r.dones[5 * N],
                          &r.neglogps[5 * N], N, 32) == 0);
  REQUIRE(ppo_xp_len(h) == cap && ppo_xp_counter(h) == cap + 37 % cap && ppo_xp_last_values_idx(h) == (37 - 1) % cap);

  std::vector<float> ret(cap * N), ret2(cap * N);
  REQUIRE(ppo_xp_returns(h, r.last_dones.data(), N + 1, 0, nullptr) == PPO_XP_EINVAL);   // buffer.rs:194
  REQUIRE(ppo_xp_returns(h, r.last_dones.data(), N, PPO_XP_GAE_SEQUENTIAL, ret.data()) == 0);
  for (long waves : {1L, 2L, 4L, 8L, 16L}) {                                 // every scan shape, same bits
    REQUIRE(ppo_xp_set_tunable(h, "gae_waves", waves) == 0);
    REQUIRE(ppo_xp_returns(h, r.last_dones.data(), N, 0, ret2.data()) == 0);
    for (size_t i = 0; i < ret.size(); ++i) REQUIRE(ret[i] == ret2[i] || (ret[i] != ret[i] && ret2[i] != ret2[i]));
  }
  REQUIRE(ppo_xp_set_tunable(h, "gae_waves", 0) == 0);
  REQUIRE(ppo_xp_set_tunable(h, "gae_waves", 3) == PPO_XP_EINVAL);
  REQUIRE(ppo_xp_set_tunable(h, "no_such_knob", 1) == PPO_XP_EINVAL);
  REQUIRE(ppo_xp_returns(h, r.last_dones.data(), N, PPO_XP_GAE_ASSOC | PPO_XP_GAE_CHUNK(3), ret2.data()) == 0);
  for (size_t i = 0; i < ret.size(); ++i) REQUIRE(std::fabs(ret[i] - ret2[i]) <= 1e-5f * std::fmax(1.f, std::fabs(ret[i])));
  REQUIRE(ppo_xp_returns(h, r.last_dones.data(), N, PPO_XP_GAE_NO_PACK, nullptr) == 0);  // records built lazily by the pass
  float ev = 0.f;
  REQUIRE(ppo_xp_explained_variance(h, &ev) == 0 && ev <= 0.f);              // training.rs:97: 0 - ratio
  double s5[5]; float ev2 = 1.f;
  REQUIRE(ppo_xp_explained_variance_sums(h, s5) == 0 && s5[0] == (double)(cap * N));
  REQUIRE(ppo_xp_explained_variance_from_sums(s5, &ev2) == 0 && ev2 == ev);  // the two halves of the same call

  // ---- passes: profiling on, every way to start one ------------------------------------------
  const size_t M = cap * N;
  REQUIRE(ppo_xp_prof_enable(h, -1) == 0);
  REQUIRE(ppo_xp_pass_begin(h, 0, nullptr, 1) == PPO_XP_EINVAL);
  REQUIRE(ppo_xp_pass_begin(h, 100, nullptr, 42) == 0);                      // library-drawn Fisher-Yates
  check_pass(h, 100, (M + 99) / 100);                                        // 1120 = 11 * 100 + overlapping tail
  REQUIRE(ppo_xp_pass_begin(h, M + 1, nullptr, 42) == 0);                    // data.rs:117: empty pass
  check_pass(h, M + 1, 0);
  std::vector<uint32_t> perm(M);
  std::iota(perm.begin(), perm.end(), 0u);
  std::shuffle(perm.begin(), perm.end(), std::mt19937(7));
  REQUIRE(ppo_xp_pass_begin(h, M, perm.data(), 0) == 0);                     // caller's permutation, one batch
  check_pass(h, M, 1);
  perm[3] = (uint32_t)M;
  REQUIRE(ppo_xp_pass_begin(h, 64, perm.data(), 0) == PPO_XP_EINVAL);        // range-checked
  perm[3] = 0;
  uint32_t *dperm = nullptr;
  REQUIRE(hipMalloc((void **)&dperm, M * sizeof(uint32_t)) == hipSuccess);
  REQUIRE(hipMemcpy(dperm, perm.data(), M * sizeof(uint32_t), hipMemcpyHostToDevice) == hipSuccess);
  REQUIRE(ppo_xp_pass_begin_device(h, 1, dperm) == 0);                       // K = M minibatches of one row
  REQUIRE(ppo_xp_pass_num_batches(h) == M);
  REQUIRE(ppo_xp_sync(h) == 0);
  REQUIRE(ppo_xp_pass_begin_device_shuffle(h, 333, 99) == 0);
  check_pass(h, 333, (M + 332) / 333);
  REQUIRE(hipFree(dperm) == hipSuccess);
  double ms[PPO_XP_K_COUNT]; uint64_t launches[PPO_XP_K_COUNT];
  REQUIRE(ppo_xp_prof_collect(h, ms, launches) == 0);
  REQUIRE(launches[PPO_XP_K_GATHER] == 4 && launches[PPO_XP_K_SCALARS] == 4 && launches[PPO_XP_K_PACK] == 1);
  REQUIRE(ppo_xp_prof_enable(h, 0) == 0);
  for (int k = 0; k < PPO_XP_K_COUNT; ++k) REQUIRE(ppo_xp_kernel_name(k)[0] != '?');

  // ---- external exchange mode: pass_next refuses until pass_finish ----------------------------
  REQUIRE(ppo_xp_returns(h, r.last_dones.data(), N, 0, nullptr) == 0);
  REQUIRE(ppo_xp_set_exchange(h, 2, nullptr) == 0);
  REQUIRE(ppo_xp_pass_begin(h, 160, nullptr, 5) == 0);
  REQUIRE(ppo_xp_pass_next(h, &bv) < 0);
  REQUIRE(ppo_xp_pass_sums_device(h) != nullptr);
  REQUIRE(ppo_xp_pass_finish(h) == 0);
  check_pass(h, 160, M / 160);
  REQUIRE(ppo_xp_set_exchange(h, 1, nullptr) == PPO_XP_EINVAL);
  REQUIRE(ppo_xp_attach_comm(h, nullptr, 0, 1) == 0);                        // detaches: mode 0

  // ---- round 2 entry points: allocation-free returns, worker-thread shuffle, host minibatches, streaming ---
  {
    std::vector<float> ra(cap * N), rb(cap * N);
    const int assoc = PPO_XP_GAE_ASSOC | PPO_XP_GAE_CHUNK(3);
    REQUIRE(ppo_xp_returns(h, r.last_dones.data(), N, assoc, ra.data()) == 0);
    REQUIRE(ppo_xp_returns_noalloc(h, r.last_dones.data(), N, assoc, rb.data()) == 0);
    REQUIRE(std::memcmp(ra.data(), rb.data(), ra.size() * sizeof(float)) == 0);
    REQUIRE(ppo_xp_returns_noalloc(h, r.last_dones.data(), N + 1, assoc, rb.data()) == PPO_XP_EINVAL);
    REQUIRE(ppo_xp_returns(h, r.last_dones.data(), N, 0, nullptr) == 0);
    REQUIRE(ppo_xp_pass_gather_kind(h) == -1);                                // no pass yet
    REQUIRE(ppo_xp_shuffle_prefetch(h, 77) == 0);
    REQUIRE(ppo_xp_pass_begin(h, 224, nullptr, 77) == 0);                     // takes the prefetched permutation
    REQUIRE(ppo_xp_pass_gather_kind(h) == 0);                                 // D = 12: the split kernels
    std::vector<uint32_t> p1(M), p2(M);
    REQUIRE(ppo_xp_pass_perm_host(h, p1.data(), M) == 0);
    ppo_xp_views_t vw;
    REQUIRE(ppo_xp_views(h, &vw) == 0 && vw.num_steps == M);
    std::vector<float> store_obs(M * D), store_val(M);
    std::vector<int64_t> store_act(M);
    REQUIRE(ppo_xp_sync(h) == 0);
    REQUIRE(hipMemcpy(store_obs.data(), vw.observations, M * D * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess);
    REQUIRE(hipMemcpy(store_val.data(), vw.values, M * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess);
    REQUIRE(hipMemcpy(store_act.data(), vw.actions, M * sizeof(int64_t), hipMemcpyDeviceToHost) == hipSuccess);
    std::vector<float> hobs(224 * D), hv(224), hn(224), hr(224), ha(224), adv_whole;
    std::vector<size_t> hact(224);
    ppo_batch_host io{};
    io.capacity = 224; io.observations = hobs.data(); io.actions = hact.data(); io.values = hv.data();
    io.neglogps = hn.data(); io.returns = hr.data(); io.advantages = ha.data();
    size_t nb = 0;
    while (true) {
      const int rc = ppo_xp_pass_next_host(h, &io);
      REQUIRE(rc == 0 || rc == 1);
      if (rc == 1) break;
      REQUIRE(io.index == nb && io.batch_size == 224 && io.obs_size == D);
      const size_t start = std::min(nb * 224, M - 224);                       // data.rs:121-125
      for (size_t j = 0; j < 224; ++j) {                                     // rows = storage[perm[...]], bit for bit
        const uint32_t src = p1[start + j];
        REQUIRE(std::memcmp(&hobs[j * D], &store_obs[src * D], D * sizeof(float)) == 0);
        REQUIRE((int64_t)hact[j] == store_act[src] && hv[j] == store_val[src]);
      }
      adv_whole.insert(adv_whole.end(), ha.begin(), ha.end());
      ++nb;
    }
    REQUIRE(nb == M / 224);
    io.capacity = 10;
    REQUIRE(ppo_xp_pass_begin(h, 224, nullptr, 77) == 0);                     // same seed, drawn on this thread now
    REQUIRE(ppo_xp_pass_perm_host(h, p2.data(), M) == 0 && p1 == p2);
    REQUIRE(ppo_xp_pass_next_host(h, &io) == PPO_XP_EINVAL);                  // destination too small: not consumed
    io.capacity = 224;
    REQUIRE(ppo_xp_pass_next_host(h, &io) == 0 && io.index == 0);
    // overlapped hand-off: two page-locked slots alternate; minibatch k+1 travels while k is "consumed".  Every host
    // batch must hold the bits of the synchronous hand-off above (adv_whole) and the rows of storage[perm]
    {
      ppo_batch_host sl[2]{};
      for (auto &q : sl) {
        q.capacity = 224;
        q.observations = (float *)ppo_xp_host_alloc(224 * D * sizeof(float)); q.actions = (size_t *)ppo_xp_host_alloc(224 * sizeof(size_t));
        q.values = (float *)ppo_xp_host_alloc(224 * 4); q.neglogps = (float *)ppo_xp_host_alloc(224 * 4);
        q.returns = (float *)ppo_xp_host_alloc(224 * 4); q.advantages = (float *)ppo_xp_host_alloc(224 * 4);
        REQUIRE(q.observations && q.actions && q.values && q.neglogps && q.returns && q.advantages);
      }
      REQUIRE(ppo_xp_pass_begin(h, 224, nullptr, 77) == 0);
      REQUIRE(ppo_xp_pass_next_host_pipelined(h, &sl[0], &sl[0]) == PPO_XP_EINVAL);       // one slot twice
      size_t k = 0;
      while (true) {
        ppo_batch_host *cur = &sl[k & 1], *nxt = &sl[(k + 1) & 1];
        const int rc = ppo_xp_pass_next_host_pipelined(h, cur, k == 3 ? nullptr : nxt);   // one call without a prefetch: the next one copies on demand
        REQUIRE(rc == 0 || rc == 1);
        if (rc == 1) break;
        REQUIRE(cur->index == k && cur->batch_size == 224 && cur->obs_size == D);
        REQUIRE(std::memcmp(cur->advantages, &adv_whole[k * 224], 224 * sizeof(float)) == 0);
        const size_t start = std::min(k * 224, M - 224);
        for (size_t j = 0; j < 224; ++j) {
          const uint32_t src = p1[start + j];
          REQUIRE(std::memcmp(&cur->observations[j * D], &store_obs[src * D], D * sizeof(float)) == 0);
          REQUIRE((int64_t)cur->actions[j] == store_act[src] && cur->values[j] == store_val[src]);
        }
        std::memset(cur->observations, 0xff, 224 * D * sizeof(float));                    // "consumed": a late copy into it would show
        ++k;
      }
      REQUIRE(k == nb);
      // a new pass while a prefetch could still be in flight, then the synchronous hand-off: ordered, same bits
      REQUIRE(ppo_xp_pass_begin(h, 224, nullptr, 77) == 0);
      REQUIRE(ppo_xp_pass_next_host_pipelined(h, &sl[0], &sl[1]) == 0);
      REQUIRE(ppo_xp_pass_begin(h, 224, nullptr, 77) == 0);
      REQUIRE(ppo_xp_pass_next_host(h, &io) == 0 && io.index == 0);
      REQUIRE(std::memcmp(ha.data(), &adv_whole[0], 224 * sizeof(float)) == 0);
      REQUIRE(ppo_xp_sync(h) == 0);
      for (auto &q : sl) {
        ppo_xp_host_free(q.observations); ppo_xp_host_free(q.actions); ppo_xp_host_free(q.values);
        ppo_xp_host_free(q.neglogps); ppo_xp_host_free(q.returns); ppo_xp_host_free(q.advantages);
      }
    }
    // streaming: two slots, same bits
    REQUIRE(ppo_xp_set_streaming(h, 1) == 0);
    REQUIRE(ppo_xp_set_exchange(h, 2, nullptr) == PPO_XP_ESTATE);
    REQUIRE(ppo_xp_pass_begin(h, 224, nullptr, 77) == 0);
    const float *slot[2] = {nullptr, nullptr};
    for (size_t k = 0; k < nb; ++k) {
      REQUIRE(ppo_xp_pass_next(h, &bv) == 0 && bv.index == k);
      if (k < 2) slot[k] = bv.observations; else REQUIRE(bv.observations == slot[k & 1]);
      REQUIRE(ppo_xp_sync(h) == 0);
      REQUIRE(hipMemcpy(ha.data(), bv.advantages, 224 * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess);
      REQUIRE(std::memcmp(ha.data(), &adv_whole[k * 224], 224 * sizeof(float)) == 0);
    }
    REQUIRE(ppo_xp_pass_next(h, &bv) == 1 && slot[0] != slot[1]);
    REQUIRE(ppo_xp_set_streaming(h, 0) == 0);
  }

  // ---- rows from page-locked host memory ------------------------------------------------------
  {
    float *pin = (float *)ppo_xp_host_alloc(N * D * sizeof(float));
    REQUIRE(pin != nullptr);
    for (size_t i = 0; i < N * D; ++i) pin[i] = r.obs[i];
    REQUIRE(ppo_xp_add(h, pin, r.rewards.data(), r.actions.data(), r.values.data(), r.dones.data(), r.neglogps.data(), N) == 0);
    ppo_xp_host_free(pin);
    ppo_xp_host_free(nullptr);
    REQUIRE(ppo_xp_pass_begin(h, 8, nullptr, 1) == PPO_XP_ESTATE);           // an insert invalidates the returns
  }

  // ---- the second handle lives through all of the above ----------------------------------------
  Rollout r2(3, 3, 7, 2);
  REQUIRE(ppo_xp_add_many(h2, r2.obs.data(), r2.rewards.data(), r2.actions.data(), r2.values.data(), r2.dones.data(),
                          r2.neglogps.data(), 3, 3) == 0);
  REQUIRE(ppo_xp_returns(h2, r2.last_dones.data(), 3, 0, nullptr) == 0);
  REQUIRE(ppo_xp_pass_begin(h2, 4, nullptr, 3) == 0);
  check_pass(h2, 4, 3);                                                      // 9 rows: 4, 4, overlapping 4
  ppo_xp_reset_counter(h2);
  REQUIRE(ppo_xp_len(h2) == 0 && ppo_xp_last_values_idx(h2) == 2);           // buffer.rs:131-133: data untouched

  // ---- reward normaliser (runner.rs:114-127) -----------------------------------------------------
  ppo_xp_rnorm *rn = nullptr;
  REQUIRE(ppo_xp_rnorm_create(&rn, 0, N, 0.99f, 1e-8f, 10.f) == 0);
  std::vector<float> out(N), st(N);
  for (int t = 0; t < 4; ++t) {
    REQUIRE(ppo_xp_rnorm_step(rn, &r.rewards[t * N], &r.dones[t * N], out.data(), N) == 0);
    for (float x : out) REQUIRE(std::fabs(x) <= 10.f);
  }
  REQUIRE(ppo_xp_rnorm_step(rn, r.rewards.data(), r.dones.data(), out.data(), N - 1) == PPO_XP_EINVAL);
  float *d_rew = nullptr, *d_done = nullptr, *d_out = nullptr;              // the sharded halves (here: one rank)
  REQUIRE(hipMalloc((void **)&d_rew, N * 4) == hipSuccess && hipMalloc((void **)&d_done, N * 4) == hipSuccess &&
          hipMalloc((void **)&d_out, N * 4) == hipSuccess);
  REQUIRE(hipMemcpy(d_rew, r.rewards.data(), N * 4, hipMemcpyHostToDevice) == hipSuccess);
  REQUIRE(hipMemset(d_done, 0, N * 4) == hipSuccess);
  REQUIRE(ppo_xp_rnorm_step_finish_device(rn, d_rew, d_done, d_out, N, nullptr) == PPO_XP_ESTATE);
  REQUIRE(ppo_xp_rnorm_step_begin_device(rn, d_rew, N, nullptr) == 0);
  REQUIRE(ppo_xp_rnorm_step_device(rn, d_rew, d_done, d_out, N, nullptr) == PPO_XP_ESTATE);
  REQUIRE(ppo_xp_rnorm_step_begin_device(rn, d_rew, N, nullptr) == PPO_XP_ESTATE);
  REQUIRE(ppo_xp_rnorm_sums_device(rn) != nullptr);
  REQUIRE(ppo_xp_rnorm_step_finish_device(rn, d_rew, d_done, d_out, N, nullptr) == 0);
  float mean, var, count;
  REQUIRE(ppo_xp_rnorm_state(rn, &mean, &var, &count, st.data(), N) == 0 && count > 5.f * N - 1.f);
  REQUIRE(hipFree(d_rew) == hipSuccess && hipFree(d_done) == hipSuccess && hipFree(d_out) == hipSuccess);
  ppo_xp_rnorm_destroy(rn);

  ppo_xp_destroy(h2);
  ppo_xp_destroy(h);
  std::printf("abi stress passed (%s)\n", ppo_xp_version());
  return 0;
}
