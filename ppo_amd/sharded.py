"""Env-sharded experience path: one process per GPU, each owning an env slice and its buffer.

The reference is single device (ppo/src/training.rs:116 `Dev::default()`); this is the multi-GPU
extension SURVEY.md 8(e) defines.  GAE is independent per env (ppo/src/common/buffer.rs:206-221 has
no cross-env term), so insert, GAE and the gathers are local.  The only exchange is the
per-minibatch advantage statistics of the standardisation (ppo/src/model/data.rs:157-164): global
minibatch k is the union of every rank's local minibatch k, so each rank contributes
(count, sum, sum of squares) in fp64 for its K local minibatches and ONE all-reduce(sum) of 3K
doubles per pass makes the global statistics known everywhere.  Observations never cross xGMI.

The permutation of a sharded pass is stratified by rank (every rank shuffles its own transitions):
a deliberate, documented difference from the single uniform permutation of data.rs:93-95.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import PpoXpPanic, check
from .buffer import ExperienceBuffer
from .data import ExperienceBatcher, ExperienceBatcherIterator


def env_slice(num_envs: int, rank: int, world: int) -> slice:
    """Contiguous env slice of `rank`; num_envs must divide evenly (every rank runs the same K)."""
    if num_envs % world:
        raise PpoXpPanic(_lib.EINVAL, f"num_envs {num_envs} is not a multiple of the world size {world}")
    n = num_envs // world
    return slice(rank * n, (rank + 1) * n)


def local_batch_size(global_batch: int, world: int) -> int:
    if global_batch % world:
        raise PpoXpPanic(_lib.EINVAL, f"batch_size {global_batch} is not a multiple of the world size {world}")
    return global_batch // world


def stats_from_sums(count: float, s1: float, s2: float):
    """Host twin of the device function of the same name (csrc/kernels_gather.h): fp64 mean and
    population variance from the sums, then the reference's fp32 tail (data.rs:163,196-200).
    Returns (mean f32, unbiased std f32)."""
    n = np.float64(count)
    m = np.float64(s1) / n
    var = max((np.float64(s2) - np.float64(s1) * m) / n, 0.0)
    nf = np.float32(n)
    with np.errstate(divide="ignore", invalid="ignore"):
        corrected = np.float32(var) * (nf / (nf - np.float32(1.0)))
        std = np.sqrt(corrected, dtype=np.float32)
    return np.float32(m), np.float32(std)


def standardize_with(a: np.ndarray, mean: np.float32, std: np.float32) -> np.ndarray:
    """(a - mean) / (std + 1e-8) in fp32 (data.rs:163)."""
    a = np.asarray(a, np.float32)
    return (a - np.float32(mean)) / (np.float32(std) + np.float32(1e-8))


def rank_seed(seed: int, rank: int) -> int:
    """Seed of rank `rank`'s local permutation in a sharded pass (one shuffle stream per rank)."""
    return (int(seed) * 0x9E3779B97F4A7C15 + int(rank)) & 0xFFFFFFFFFFFFFFFF


def local_sums(adv: np.ndarray) -> np.ndarray:
    """(count, sum, sum of squares) in fp64 of one local minibatch (what a rank contributes)."""
    a = np.asarray(adv, np.float64)
    return np.array([a.size, a.sum(), (a * a).sum()], np.float64)


class ShardedBatcher:
    """ExperienceBatcher for one rank of an env-sharded job.

    `group` is a torch.distributed process group (None = the default group).  With the nccl backend
    the all-reduce runs on the sums in device memory on torch's current stream, ordered against the
    buffer's stream by event waits (none needed after buffer.use_torch_stream()); with gloo the 3K
    doubles take a round trip through the host (tests)."""

    def __init__(self, buffer: ExperienceBuffer, global_batch_size: int, group=None, seed: int = 0x5EED):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.buffer = buffer
        self.local_batch = local_batch_size(global_batch_size, self.world)
        self._batcher = ExperienceBatcher(buffer, self.local_batch)
        self._seed = seed
        self._L = _lib.load()
        buffer.set_exchange(2)

    def __len__(self):
        return len(self._batcher)

    def num_batches(self):
        return self._batcher.num_batches()

    def exchange(self):
        """All-reduce(sum) of this pass's [K, 3] fp64 sums, then the standardise kernel."""
        import torch
        K = self._L.ppo_xp_pass_num_batches(self.buffer.handle)
        if not K:
            return        # batch_size > M: the pass is empty (data.rs:117), nothing is pending
        if K:
            sums = self.buffer.pass_sums(K)
            if self.dist.get_backend(self.group) == "nccl":
                # RCCL orders the all-reduce against torch's CURRENT stream; the sums are written, and the
                # standardise kernel launched, on the handle's stream.  When the two differ (no
                # buffer.use_torch_stream()), bracket the collective with stream-to-stream waits.
                cur, hs = self.buffer._order_streams()
                if hs is not None:
                    cur.wait_stream(hs)
                self.dist.all_reduce(sums, group=self.group)
                if hs is not None:
                    hs.wait_stream(cur)
            else:
                self.buffer.sync()
                host = sums.cpu()
                self.dist.all_reduce(host, group=self.group)
                sums.copy_(host)
                torch.cuda.current_stream(self.buffer.device).synchronize()
        self.buffer.pass_finish()

    def into_iter(self, seed=None, perm=None, bijection: bool = False) -> ExperienceBatcherIterator:
        """One sharded pass: local gather + local sums, the exchange, local standardisation.  bijection=True: every rank's
        local permutation is the keyed bijection of its rank seed, evaluated inside the gather (no index array)."""
        if seed is None and perm is None:
            seed = self._seed
            self._seed += 1
        if seed is not None:
            seed = rank_seed(seed, self.rank)
        it = self._batcher.into_iter(seed=seed, perm=perm, bijection=bijection)
        self.exchange()
        return it

    def __iter__(self):
        return self.into_iter()


def _all_reduce_f64(dist, t, group):
    """Sum a small f64 tensor over the ranks in place; gloo (tests) goes through the host."""
    import torch
    if t.is_cuda and dist.get_backend(group) != "nccl":
        torch.cuda.current_stream(t.device).synchronize()
        host = t.cpu()
        dist.all_reduce(host, group=group)
        t.copy_(host)
        torch.cuda.current_stream(t.device).synchronize()
    else:
        dist.all_reduce(t, group=group)


def explained_variance(buffer: ExperienceBuffer, group=None) -> float:
    """explained_variance(values, returns) (ppo/src/training.rs:88-99) over ALL ranks' transitions: every
    rank contributes five f64 sums, one all-reduce, the same closing arithmetic as on one GPU."""
    import torch
    import torch.distributed as dist
    sums = torch.from_numpy(buffer.explained_variance_sums())
    if dist.get_backend(group) == "nccl":
        dev = sums.to(f"cuda:{buffer.device}")
        dist.all_reduce(dev, group=group)
        sums = dev.cpu()
    else:
        dist.all_reduce(sums, group=group)
    s = np.ascontiguousarray(sums.numpy(), np.float64)
    out = C.c_float()
    check(_lib.load().ppo_xp_explained_variance_from_sums(s.ctypes.data_as(C.POINTER(C.c_double)), C.byref(out)))
    return out.value


class ShardedRewardNormalizer:
    """RewardNormalizer (ppo/src/common/runner.rs:114-127) for one rank of an env-sharded job: local
    discounted returns, ONE all-reduce of (count, sum, sum of squares) per step, identical running
    statistics on every rank."""

    def __init__(self, num_local_envs: int, group=None, **kw):
        import torch.distributed as dist
        from .runner import RewardNormalizer
        self.dist, self.group = dist, group
        self.inner = RewardNormalizer(num_local_envs, **kw)

    def normalized_rewards(self, rewards, dones):
        return self.inner.normalized_rewards_sharded(rewards, dones, lambda t: _all_reduce_f64(self.dist, t, self.group))

    def state(self):
        return self.inner.state()

    def close(self):
        self.inner.close()
