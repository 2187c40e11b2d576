// ppo_xp.hpp — header-only C++ mirror of the reference's Rust interface over the C ABI (ppo_xp.h).
//
// Same names, argument meaning and failure behaviour as
//   ExperienceBuffer<NUM_ENVS, OBS_SIZE>   ppo/src/common/buffer.rs:23-226
//   ExperienceBatcher / ExperienceBatch    ppo/src/model/data.rs:4-193
// Where the reference panics (assert_eq! on sizes) these throw ppo_xp::Panic.  Host code only:
// no HIP headers needed; device pointers are handed out as raw pointers.
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "ppo_xp.h"

namespace ppo_xp_cpp {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string &m) : std::runtime_error("ppo_xp error " + std::to_string(c) + ": " + m), code(c) {}
};
struct Panic : Error { using Error::Error; };  // PPO_XP_EINVAL: the reference would have panicked

inline int check(int rc) {
  if (rc < 0) {
    if (rc == PPO_XP_EINVAL) throw Panic(rc, ppo_xp_last_error());
    throw Error(rc, ppo_xp_last_error());
  }
  return rc;
}

// data.rs:4-12 — device pointers of one minibatch
using ExperienceBatch = ppo_batch_view;

// data.rs:4-12 with the arrays in host memory: what the reference's next() hands to copy_from (data.rs:166-182)
template <size_t OBS_SIZE>
struct HostExperienceBatch {
  size_t index = 0;
  std::vector<std::array<float, OBS_SIZE>> observations;
  std::vector<size_t> actions;
  std::vector<float> values, neglogps, returns, advantages;
};

template <size_t NUM_ENVS, size_t OBS_SIZE>
class ExperienceBuffer {
 public:
  // buffer.rs:45-47
  static ExperienceBuffer make(size_t capacity, int device = 0) { return ExperienceBuffer(capacity, 0.99f, 0.95f, device); }
  // buffer.rs:52-69
  static ExperienceBuffer new_with_params(size_t capacity, float gamma, float lambda, int device = 0) {
    return ExperienceBuffer(capacity, gamma, lambda, device);
  }
  ExperienceBuffer(ExperienceBuffer &&o) noexcept : h_(o.h_) { o.h_ = nullptr; }
  ExperienceBuffer(const ExperienceBuffer &) = delete;
  ExperienceBuffer &operator=(const ExperienceBuffer &) = delete;
  ~ExperienceBuffer() { ppo_xp_destroy(h_); }

  // buffer.rs:84-121.  dones: one byte per env (Rust bool).
  void add_experience(const std::vector<std::array<float, OBS_SIZE>> &obs, const std::vector<float> &rewards,
                      const std::vector<size_t> &actions, const std::vector<float> &vals,
                      const std::vector<uint8_t> &dones, const std::vector<float> &neglogps) {
    const size_t n = obs.size();
    if (rewards.size() != n || actions.size() != n || vals.size() != n || dones.size() != n || neglogps.size() != n)
      throw Panic(PPO_XP_EINVAL, "add_experience: slices of different length (buffer.rs:97-112)");
    check(ppo_xp_add(h_, n ? obs[0].data() : nullptr, rewards.data(), actions.data(), vals.data(), dones.data(),
                     neglogps.data(), n));
  }
  void reset_counter() { ppo_xp_reset_counter(h_); }   // buffer.rs:131-133
  size_t len() const { return ppo_xp_len(h_); }        // buffer.rs:123-129

  // buffer.rs:142-175 (device pointers; storage order)
  ppo_xp_views_t training_views() const { ppo_xp_views_t v; check(ppo_xp_views(h_, &v)); return v; }

  // buffer.rs:187-225: the returns, copied to the host (they also stay resident for the batcher)
  std::vector<float> returns(const std::vector<uint8_t> &last_dones, int algo = PPO_XP_GAE_SEQUENTIAL) {
    std::vector<float> out(len() * NUM_ENVS);
    check(ppo_xp_returns(h_, last_dones.data(), last_dones.size(), algo, out.data()));
    return out;
  }
  ppo_xp *handle() const { return h_; }

 private:
  ExperienceBuffer(size_t capacity, float gamma, float lambda, int device) {
    check(ppo_xp_create(&h_, device, capacity, NUM_ENVS, OBS_SIZE, gamma, lambda));
  }
  ppo_xp *h_ = nullptr;
};

// data.rs:14-193.  `for (auto it = batcher.into_iter(seed); auto *b = it.next();) ...`
template <size_t NUM_ENVS, size_t OBS_SIZE>
class ExperienceBatcher {
 public:
  ExperienceBatcher(ExperienceBuffer<NUM_ENVS, OBS_SIZE> &buf, size_t batch_size) : buf_(buf), batch_size_(batch_size) {}
  size_t len() const { return buf_.len() * NUM_ENVS; }   // data.rs:80-82

  class Iter {
   public:
    // data.rs:115-192: nullptr == None
    const ExperienceBatch *next() {
      int rc = check(ppo_xp_pass_next(h_, &cur_));
      return rc == 0 ? &cur_ : nullptr;
    }
    // the same minibatch in host vectors (ppo_xp_pass_next_host); false == None
    bool next_host(HostExperienceBatch<OBS_SIZE> &b) {
      b.observations.resize(batch_size_); b.actions.resize(batch_size_); b.values.resize(batch_size_);
      b.neglogps.resize(batch_size_); b.returns.resize(batch_size_); b.advantages.resize(batch_size_);
      ppo_batch_host io{};
      io.capacity = batch_size_;
      io.observations = batch_size_ ? b.observations[0].data() : nullptr; io.actions = b.actions.data();
      io.values = b.values.data(); io.neglogps = b.neglogps.data(); io.returns = b.returns.data();
      io.advantages = b.advantages.data();
      if (check(ppo_xp_pass_next_host(h_, &io)) != 0) return false;
      b.index = io.index;
      return true;
    }
    // the same with the download of minibatch k+1 overlapping the caller's work on k (ppo_xp_pass_next_host_pipelined):
    // `cur` receives minibatch k, `next_slot` must stay untouched until it is passed as `cur`; alternate two batches
    bool next_host_pipelined(HostExperienceBatch<OBS_SIZE> &cur, HostExperienceBatch<OBS_SIZE> &next_slot) {
      ppo_batch_host a = slot(cur), b = slot(next_slot);
      if (check(ppo_xp_pass_next_host_pipelined(h_, &a, &b)) != 0) return false;
      cur.index = a.index;
      return true;
    }
    std::vector<uint32_t> perm() const {
      ppo_xp_views_t v; check(ppo_xp_views(h_, &v));
      std::vector<uint32_t> p(v.num_steps);
      check(ppo_xp_pass_perm_host(h_, p.data(), p.size()));
      return p;
    }
   private:
    friend class ExperienceBatcher;
    Iter(ppo_xp *h, size_t batch_size) : h_(h), batch_size_(batch_size) {}
    ppo_batch_host slot(HostExperienceBatch<OBS_SIZE> &b) const {
      if (b.values.size() != batch_size_) {      // sized once: the arrays must not move while a download is in flight
        b.observations.resize(batch_size_); b.actions.resize(batch_size_); b.values.resize(batch_size_);
        b.neglogps.resize(batch_size_); b.returns.resize(batch_size_); b.advantages.resize(batch_size_);
      }
      ppo_batch_host io{};
      io.capacity = batch_size_;
      io.observations = batch_size_ ? b.observations[0].data() : nullptr; io.actions = b.actions.data();
      io.values = b.values.data(); io.neglogps = b.neglogps.data(); io.returns = b.returns.data();
      io.advantages = b.advantages.data();
      return io;
    }
    ppo_xp *h_;
    size_t batch_size_;
    ExperienceBatch cur_{};
  };

  // data.rs:91-102: a fresh (seeded) permutation per pass
  Iter into_iter(uint64_t seed) { check(ppo_xp_pass_begin(buf_.handle(), batch_size_, nullptr, seed)); return Iter(buf_.handle(), batch_size_); }
  // a caller-supplied permutation: the ABI reads exactly len() entries, so a vector of another length is a
  // panic here (the reference's indices vector always has buf_size entries, data.rs:93)
  Iter into_iter(const std::vector<uint32_t> &perm) {
    if (perm.size() != len()) throw Panic(PPO_XP_EINVAL, "into_iter: the permutation must have len() entries");
    check(ppo_xp_pass_begin(buf_.handle(), batch_size_, perm.data(), 0));
    return Iter(buf_.handle(), batch_size_);
  }
  // the permutation drawn on the device (no host Fisher-Yates on the critical path)
  Iter into_iter_device_shuffle(uint64_t seed) {
    check(ppo_xp_pass_begin_device_shuffle(buf_.handle(), batch_size_, seed));
    return Iter(buf_.handle(), batch_size_);
  }
  // the permutation as a keyed bijection of (seed, len()) evaluated inside the gather: no index array at all
  Iter into_iter_bijection(uint64_t seed) {
    check(ppo_xp_pass_begin_bijection(buf_.handle(), batch_size_, seed));
    return Iter(buf_.handle(), batch_size_);
  }
  // start drawing the Fisher-Yates of a coming into_iter(seed) on a worker thread
  void prefetch_shuffle(uint64_t seed) { check(ppo_xp_shuffle_prefetch(buf_.handle(), seed)); }

 private:
  ExperienceBuffer<NUM_ENVS, OBS_SIZE> &buf_;
  size_t batch_size_;
};

}  // namespace ppo_xp_cpp
