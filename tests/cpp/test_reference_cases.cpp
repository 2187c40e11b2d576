// The reference's own unit tests (ppo/src/common/buffer.rs:233-458, ppo/src/model/data.rs:210-270)
// re-stated against the C++ mirror (include/ppo_xp.hpp) of the C ABI: compiled host code calling
// libppo_xp.so, as a Rust shim would.  Exit code 0 = all passed.  Needs a GPU.
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <set>

#include "ppo_xp.hpp"

using namespace ppo_xp_cpp;
using Obs = std::vector<std::array<float, 3>>;

static int failures = 0;
#define EXPECT(c) do { if (!(c)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); ++failures; } } while (0)

template <typename T>
static std::vector<T> to_host(const T *dev, size_t n) {
  std::vector<T> h(n);
  if (n && hipMemcpy(h.data(), dev, n * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) { std::printf("hipMemcpy failed\n"); ++failures; }
  return h;
}

struct Step { Obs obs; std::vector<float> rewards; std::vector<size_t> actions; std::vector<float> vals; std::vector<uint8_t> dones; std::vector<float> neglogps; };
static const Step s1{{{0, 1, 2}, {1, 2, 3}}, {0.1f, 1.1f}, {1, 2}, {3, 6}, {0, 0}, {20, 21}};
static const Step s2{{{2, 3, 4}, {3, 4, 5}}, {1.1f, 2.1f}, {2, 3}, {6, 9}, {0, 0}, {21, 22}};
static const Step s3{{{4, 5, 6}, {5, 6, 7}}, {2.1f, 3.1f}, {3, 4}, {9, 12}, {0, 0}, {22, 23}};
static const Step s4{{{5, 6, 7}, {6, 7, 8}}, {3.1f, 4.1f}, {4, 5}, {12, 15}, {0, 0}, {23, 24}};
static const Step s5{{{6, 7, 8}, {7, 8, 9}}, {4.1f, 5.1f}, {5, 6}, {15, 18}, {0, 1}, {24, 25}};

template <typename B> static void add(B &b, const Step &s) { b.add_experience(s.obs, s.rewards, s.actions, s.vals, s.dones, s.neglogps); }

static void exp_buffer_below_capacity() {   // buffer.rs:233-262
  auto buf = ExperienceBuffer<2, 3>::make(3);
  add(buf, s1);
  auto v = buf.training_views();
  EXPECT(v.num_steps == 2);
  ppo_xp_sync(buf.handle());
  EXPECT((to_host(v.observations, 6) == std::vector<float>{0, 1, 2, 1, 2, 3}));
  EXPECT((to_host(v.actions, 2) == std::vector<int64_t>{1, 2}));
  EXPECT((to_host(v.values, 2) == std::vector<float>{3, 6}));
  EXPECT((to_host(v.neglogps, 2) == std::vector<float>{20, 21}));
  EXPECT(buf.returns({1, 1}).size() == 2);
}

static void exp_buffer_over_capacity() {    // buffer.rs:326-409
  auto buf = ExperienceBuffer<2, 3>::make(3);
  add(buf, s1);
  for (int i = 0; i < 3 * 456 - 2; ++i) add(buf, s2);
  add(buf, s3); add(buf, s4); add(buf, s5);
  auto v = buf.training_views();
  ppo_xp_sync(buf.handle());
  EXPECT((to_host(v.observations, 18) == std::vector<float>{5, 6, 7, 6, 7, 8, 6, 7, 8, 7, 8, 9, 4, 5, 6, 5, 6, 7}));
  EXPECT((to_host(v.actions, 6) == std::vector<int64_t>{4, 5, 5, 6, 3, 4}));
  EXPECT((to_host(v.values, 6) == std::vector<float>{12, 15, 15, 18, 9, 12}));
  EXPECT((to_host(v.neglogps, 6) == std::vector<float>{23, 24, 24, 25, 22, 23}));
  EXPECT(buf.returns({1, 1}).size() == 6);
}

static void exp_buffer_returns_sanity() {   // buffer.rs:412-458
  auto buf = ExperienceBuffer<2, 3>::make(3);
  add(buf, s1); add(buf, s2); add(buf, s3);
  auto ret = buf.returns({1, 1});
  EXPECT(ret[0] > 3.708f && ret[0] < 3.7081f);
  EXPECT(ret[1] > 6.821f && ret[1] < 6.822f);
  EXPECT(ret[2] > 3.52f && ret[2] < 3.521f);
  EXPECT(ret[3] > 5.609f && ret[3] < 5.61f);
  EXPECT(ret[4] > 2.09f && ret[4] < 2.11f);
  EXPECT(ret[5] > 3.09f && ret[5] < 3.11f);
  EXPECT(buf.training_views().num_steps == ret.size());
}

static void panics() {                      // buffer.rs:97-112, 194
  auto buf = ExperienceBuffer<2, 3>::make(3);
  bool threw = false;
  try { buf.add_experience({{0, 1, 2}}, {0.1f}, {1}, {3}, {0}, {20}); } catch (const Panic &) { threw = true; }
  EXPECT(threw && buf.len() == 0);
  add(buf, s1);
  threw = false;
  try { buf.returns({1}); } catch (const Panic &) { threw = true; }
  EXPECT(threw);
}

static void batcher_sanity() {              // data.rs:210-270
  auto buf = ExperienceBuffer<2, 3>::make(10);
  for (int i = 0; i < 5; ++i)
    buf.add_experience({{0.0f + i, 1, 2}, {100.0f + i, 2, 3}}, {0.1f, 1.1f}, {1, 2}, {3, 6}, {0, 0}, {20, 21});
  buf.returns({1, 1});
  ExperienceBatcher<2, 3> batcher(buf, 6);
  EXPECT(batcher.len() == 10);
  std::set<long> seen;
  int n = 0;
  auto it = batcher.into_iter(42);
  while (const ExperienceBatch *b = it.next()) {
    ppo_xp_sync(buf.handle());
    auto obs = to_host(b->observations, b->batch_size * 3);
    for (size_t j = 0; j < obs.size(); j += 3) seen.insert(std::lround(obs[j]));
    auto adv = to_host(b->advantages, b->batch_size);
    double m = 0; for (float a : adv) m += a;
    EXPECT(std::fabs(m / adv.size()) < 1e-5 || std::isnan(m));
    ++n;
  }
  EXPECT(n == 2);
  for (long v : {0, 1, 2, 3, 4, 100, 101, 102, 103, 104}) EXPECT(seen.count(v) == 1);
}

// The shim's next() keeps the reference's type: ExperienceBatch in host vectors (data.rs:166-182 hands
// exactly these to copy_from).  Rows must be src[perm[j]] bit for bit; the standardised advantages must be
// the bits the device minibatch holds.  Also: a permutation of the wrong length panics.
static void batcher_host_batches() {
  auto buf = ExperienceBuffer<2, 3>::make(10);
  std::vector<float> obs0;
  for (int i = 0; i < 5; ++i) {
    buf.add_experience({{0.5f + i, 1, 2}, {100.25f + i, 2, 3}}, {0.1f * i, 1.1f}, {size_t(1 + i), size_t(1) << 40}, {3.0f + i, 6}, {0, 0}, {20, 21.5f + i});
    for (float x : {0.5f + i, 1.f, 2.f, 100.25f + i, 2.f, 3.f}) obs0.push_back(x);
  }
  auto ret = buf.returns({0, 1});
  ExperienceBatcher<2, 3> batcher(buf, 6);
  bool threw = false;
  try { batcher.into_iter(std::vector<uint32_t>{0, 1, 2}); } catch (const Panic &) { threw = true; }
  EXPECT(threw);
  // device batches of the pass with seed 7
  std::vector<std::vector<float>> dev_adv, dev_val;
  {
    auto it = batcher.into_iter(7);
    while (const ExperienceBatch *b = it.next()) {
      ppo_xp_sync(buf.handle());
      dev_adv.push_back(to_host(b->advantages, b->batch_size));
      dev_val.push_back(to_host(b->values, b->batch_size));
    }
  }
  batcher.prefetch_shuffle(7);                 // the same permutation, drawn by the worker thread
  auto it = batcher.into_iter(7);
  auto perm = it.perm();
  EXPECT(perm.size() == 10);
  HostExperienceBatch<3> hb;
  size_t k = 0;
  while (it.next_host(hb)) {
    EXPECT(hb.index == k && hb.observations.size() == 6);
    const size_t start = k == 0 ? 0 : 4;       // data.rs:121-125: the tail window is [M-B, M)
    for (size_t j = 0; j < 6; ++j) {
      const uint32_t p = perm[start + j];
      for (int c = 0; c < 3; ++c) EXPECT(hb.observations[j][c] == obs0[3 * p + c]);
      EXPECT(hb.returns[j] == ret[p]);
      EXPECT(hb.actions[j] == ((p & 1) ? (size_t(1) << 40) : size_t(1 + p / 2)));
      EXPECT(hb.values[j] == dev_val[k][j]);
      EXPECT(std::memcmp(&hb.advantages[j], &dev_adv[k][j], 4) == 0);
    }
    ++k;
  }
  EXPECT(k == 2 && dev_adv.size() == 2);
  // a destination that is too small is refused and the minibatch is not consumed
  auto it2 = batcher.into_iter(8);
  ppo_batch_host io{};
  std::vector<float> f(6 * 3); std::vector<size_t> a(6);
  io.capacity = 5; io.observations = f.data(); io.actions = a.data(); io.values = f.data(); io.neglogps = f.data();
  io.returns = f.data(); io.advantages = f.data();
  EXPECT(ppo_xp_pass_next_host(buf.handle(), &io) == PPO_XP_EINVAL);
  EXPECT(it2.next() != nullptr && it2.next() != nullptr && it2.next() == nullptr);
  // data.rs:91-102 with the permutation as a keyed bijection evaluated inside the gather, and the hand-off of
  // training.rs:205-207 overlapped: two host batches alternate, k+1 downloads while k is looked at
  auto it3 = batcher.into_iter_bijection(99);
  auto perm3 = it3.perm();
  std::vector<int> seen(10, 0);
  for (uint32_t p : perm3) { EXPECT(p < 10); if (p < 10) seen[p]++; }
  for (int c : seen) EXPECT(c == 1);                       // a permutation of 0..10
  HostExperienceBatch<3> slots[2];
  k = 0;
  while (it3.next_host_pipelined(slots[k & 1], slots[(k + 1) & 1])) {
    const HostExperienceBatch<3> &b = slots[k & 1];
    EXPECT(b.index == k);
    const size_t start = k == 0 ? 0 : 4;
    for (size_t j = 0; j < 6; ++j) {
      const uint32_t p = perm3[start + j];
      for (int c = 0; c < 3; ++c) EXPECT(b.observations[j][c] == obs0[3 * p + c]);
      EXPECT(b.returns[j] == ret[p]);
      EXPECT(b.actions[j] == ((p & 1) ? (size_t(1) << 40) : size_t(1 + p / 2)));
    }
    ++k;
  }
  EXPECT(k == 2);
}

int main() {
  try {
    exp_buffer_below_capacity();
    exp_buffer_over_capacity();
    exp_buffer_returns_sanity();
    panics();
    batcher_sanity();
    batcher_host_batches();
  } catch (const std::exception &e) { std::printf("exception: %s\n", e.what()); return 2; }
  std::printf(failures ? "%d FAILURES\n" : "all reference cases passed (%d failures)\n", failures);
  return failures ? 1 : 0;
}
