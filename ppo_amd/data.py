"""ExperienceBatch / ExperienceBatcher — host-side mirror of ppo/src/model/data.rs over the C ABI.

  ExperienceBatch            data.rs:4-12      ExperienceBatcher::new   data.rs:22-78
  ExperienceBatcher::len     data.rs:80-82     into_iter (shuffle)      data.rs:85-103
  Iterator::next             data.rs:115-192

Differences from the reference that a caller can see, all deliberate:
  * the master copy (`set`, data.rs:37-70) is never made: the buffer is already on the device and
    `advantages = returns - values` (data.rs:62) was fused into the GAE kernel;
  * one `iter()` gathers all K minibatches in one launch; `next` hands out slices of the slabs;
  * the shuffle is seedable (`seed=` / `perm=`); the reference's thread_rng() is not reproducible.
"""
from __future__ import annotations

import ctypes as C
import itertools
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import PpoXpPanic, check
from .buffer import ExperienceBuffer, device_tensor


@dataclass
class ExperienceBatch:
    """data.rs:4-12 — all fields are device tensors."""
    observations: "torch.Tensor"  # [batch_size, obs_size] f32
    actions: "torch.Tensor"       # [batch_size] i64 (usize in the reference)
    values: "torch.Tensor"        # [batch_size]
    neglogps: "torch.Tensor"      # [batch_size]
    returns: "torch.Tensor"       # [batch_size]
    advantages: "torch.Tensor"    # [batch_size], standardised per minibatch (data.rs:157-164)


class HostBatch:
    """Host-memory ExperienceBatch (ppo_batch_host): the six arrays data.rs:166-182 hands to copy_from."""

    def __init__(self, arrays, capacity):
        self.observations, self.actions, self.values, self.neglogps, self.returns, self.advantages = arrays
        self.index = -1
        self._c = _lib.BatchHost()
        self._c.capacity = capacity
        for name, a in zip(("observations", "actions", "values", "neglogps", "returns", "advantages"), arrays):
            setattr(self._c, name, a.ctypes.data)

    @classmethod
    def pinned(cls, capacity, obs_size):
        from .buffer import pinned_empty
        f = np.float32
        return cls([pinned_empty((capacity, obs_size), f), pinned_empty((capacity,), np.uint64), pinned_empty((capacity,), f),
                    pinned_empty((capacity,), f), pinned_empty((capacity,), f), pinned_empty((capacity,), f)], capacity)

    @classmethod
    def pageable(cls, capacity, obs_size):
        f = np.float32
        return cls([np.empty((capacity, obs_size), f), np.empty((capacity,), np.uint64), np.empty((capacity,), f),
                    np.empty((capacity,), f), np.empty((capacity,), f), np.empty((capacity,), f)], capacity)

    def view(self, rows, index):
        """The first `rows` rows (the arrays are sized for `capacity`)."""
        self.index = index
        if rows == self._c.capacity:
            return self
        b = HostBatch.__new__(HostBatch)
        b._c, b.index = self._c, index
        b.observations, b.actions, b.values = self.observations[:rows], self.actions[:rows], self.values[:rows]
        b.neglogps, b.returns, b.advantages = self.neglogps[:rows], self.returns[:rows], self.advantages[:rows]
        b._owner = self
        return b


class ExperienceBatcher:
    """data.rs:14-83.  Holds the buffer whose views it was built from and a batch size."""

    _seed_counter = itertools.count(0x5EED)

    def __init__(self, buffer: ExperienceBuffer, batch_size: int, seed: int | None = None):
        self.buffer, self.batch_size = buffer, int(batch_size)
        self._seed = seed
        self._L = _lib.load()

    @classmethod
    def new(cls, observations, actions, values, neglogps, returns, batch_size, cache=None, seed=None):
        """ExperienceBatcher::new (data.rs:22-30).  The five views must be the ones handed out by ONE
        ExperienceBuffer (training_views() + returns()); `cache` is accepted for signature parity and
        ignored (the kernels write device minibatches directly, data.rs:166-182 has no counterpart)."""
        owners = {id(getattr(t, "_ppo_owner", None)) for t in (observations, actions, values, neglogps, returns)}
        owner = getattr(observations, "_ppo_owner", None)
        if owner is None or len(owners) != 1:
            raise PpoXpPanic(_lib.EINVAL, "ExperienceBatcher.new: pass the views of one ExperienceBuffer")
        return cls(owner, batch_size, seed)

    def __len__(self):
        """data.rs:80-82: number of transitions M."""
        return len(self.buffer) * self.buffer.num_envs

    def num_batches(self):
        M, B = len(self), self.batch_size
        return 0 if (B > M or B <= 0) else (M + B - 1) // B

    def __iter__(self):
        return self.into_iter()

    def prefetch_shuffle(self, seed: int):
        """Draw the host Fisher-Yates of a coming pass on a worker thread of the library; the
        into_iter(seed=seed) that follows takes it (training.rs:204-205: one fresh permutation per pass)."""
        check(self._L.ppo_xp_shuffle_prefetch(self.buffer.handle, int(seed) & 0xFFFFFFFFFFFFFFFF))

    def into_iter(self, seed: int | None = None, perm=None, device_shuffle: bool = False, bijection: bool = False):
        """data.rs:91-102: a fresh permutation per pass.  perm: optional host u32 array or CUDA
        int32 tensor of M indices.  device_shuffle=True draws the permutation on the GPU (stable
        argsort of SplitMix64 keys) instead of the host Fisher-Yates.  bijection=True: the permutation is a
        keyed bijection of (seed, M) that the gather evaluates in registers (ppo_xp_pass_begin_bijection):
        no index array is drawn, uploaded or read."""
        return ExperienceBatcherIterator(self, seed, perm, device_shuffle, bijection)


class ExperienceBatcherIterator:
    """data.rs:105-193."""

    def __init__(self, batcher: ExperienceBatcher, seed=None, perm=None, device_shuffle=False, bijection=False):
        self.batcher = batcher
        buf, L = batcher.buffer, batcher._L
        h = buf.handle
        B = batcher.batch_size
        if getattr(buf, "_pass_perm", None) is not None:
            # the previous pass borrowed a torch tensor: order torch's stream behind the kernels that read
            # it before the allocator may hand its memory to someone else
            cur, hs = buf._order_streams()
            if hs is not None:
                cur.wait_stream(hs)
            buf._pass_perm = None
        if perm is not None and hasattr(perm, "is_cuda") and perm.is_cuda:
            import torch
            if perm.numel() != len(batcher) or perm.dtype not in (torch.int32,) or not perm.is_contiguous():
                raise PpoXpPanic(_lib.EINVAL, "perm: contiguous CUDA int32 tensor of M indices expected")
            # the library BORROWS the array until the pass's kernels have run: the buffer keeps it alive until
            # its next pass (released there only after torch's stream has been ordered behind the handle's),
            # or a temporary such as randperm(M).int() would go back to torch's allocator while the gathers
            # are still pending on the handle's stream
            buf._pass_perm = perm
            cur, hs = buf._order_streams()
            if hs is not None:
                hs.wait_stream(cur)                 # whatever produced perm ran on torch's stream
            check(L.ppo_xp_pass_begin_device(h, B, C.c_void_p(perm.data_ptr())))
        elif perm is not None:
            p = np.ascontiguousarray(perm, dtype=np.uint32)
            if p.shape != (len(batcher),):
                raise PpoXpPanic(_lib.EINVAL, "perm: M indices expected")
            check(L.ppo_xp_pass_begin(h, B, p.ctypes.data_as(C.c_void_p), 0))
        else:
            if seed is None:
                seed = batcher._seed if batcher._seed is not None else next(ExperienceBatcher._seed_counter)
                if batcher._seed is not None:
                    batcher._seed += 1  # a new permutation per pass (training.rs:204-205)
            if bijection:
                check(L.ppo_xp_pass_begin_bijection(h, B, int(seed) & 0xFFFFFFFFFFFFFFFF))
            elif device_shuffle:
                check(L.ppo_xp_pass_begin_device_shuffle(h, B, int(seed) & 0xFFFFFFFFFFFFFFFF))
            else:
                check(L.ppo_xp_pass_begin(h, B, None, int(seed) & 0xFFFFFFFFFFFFFFFF))

    def __iter__(self):
        return self

    def __next__(self) -> ExperienceBatch:
        """Streaming mode (buffer.set_streaming): the tensors are zero-copy views of slot k % 2 and are OVERWRITTEN by the
        call after the next one - consume or clone a minibatch before asking for the one after next; `list(iterator)`
        would keep K aliases of two slots."""
        import torch
        buf = self.batcher.buffer
        v = _lib.BatchView()
        hs = None
        if buf.streaming:
            # the call enqueues minibatch k's gather into slot k % 2 on the handle's stream: it must run after whatever
            # torch's current stream still does with that slot (the consumer's work on minibatch k - 2) ...
            cur, hs = buf._order_streams()
            if hs is not None:
                hs.wait_stream(cur)
        rc = check(self.batcher._L.ppo_xp_pass_next(buf.handle, C.byref(v)))
        if rc == 1:
            raise StopIteration
        if buf.streaming:
            # ... and torch's stream must see the minibatch: stream-to-stream, no host wait, so the consumer's work
            # on k overlaps the gather of k + 1
            if hs is not None:
                cur.wait_stream(hs)
        elif v.index == 0:
            buf.sync()
        B, D, d = v.batch_size, v.obs_size, buf.device
        f32, i64 = torch.float32, torch.int64
        return ExperienceBatch(
            device_tensor(v.observations, (B, D), f32, d, buf), device_tensor(v.actions, (B,), i64, d, buf),
            device_tensor(v.values, (B,), f32, d, buf), device_tensor(v.neglogps, (B,), f32, d, buf),
            device_tensor(v.returns, (B,), f32, d, buf), device_tensor(v.advantages, (B,), f32, d, buf))

    def next_host(self, out: "HostBatch | None" = None) -> "HostBatch | None":
        """Iterator::next as the reference's caller sees it (data.rs:166-182 feeds six host Vecs to
        copy_from): minibatch k copied into host arrays (page-locked when `out` comes from
        HostBatch.pinned).  Returns None when the pass is exhausted.  With out=None every call fills the SAME page-locked
        set owned by the iterator: copy what must outlive the next call."""
        buf = self.batcher.buffer
        if out is None:      # one page-locked set per iterator, reused by every call (a fresh hipHostMalloc per minibatch costs more than the copy)
            if getattr(self, "_host_batch", None) is None:
                self._host_batch = HostBatch.pinned(self.batcher.batch_size, buf.obs_size)
            out = self._host_batch
        rc = check(self.batcher._L.ppo_xp_pass_next_host(buf.handle, C.byref(out._c)))
        if rc == 1:
            return None
        return out.view(out._c.batch_size, out._c.index)

    def host_batches(self, slots=None):
        """Generator over the pass as HOST minibatches with the download of k+1 overlapping the consumer's work on k
        (ppo_xp_pass_next_host_pipelined): two page-locked slots alternate, so a yielded batch is valid until the one
        after next is asked for.  `slots`: two HostBatch objects (default: a page-locked pair owned by the iterator)."""
        buf = self.batcher.buffer
        if slots is None:
            if getattr(self, "_host_pair", None) is None:
                self._host_pair = (HostBatch.pinned(self.batcher.batch_size, buf.obs_size),
                                   HostBatch.pinned(self.batcher.batch_size, buf.obs_size))
            slots = self._host_pair
        # a download of minibatch k+1 may still be in flight when the consumer stops early and drops this iterator: the
        # buffer keeps the two slots alive until its next pass (whose begin orders itself behind the copy stream) or close()
        if getattr(buf, "_host_slots", None) is not None and buf._host_slots is not slots:
            buf.sync()              # an abandoned pass may still be writing the slots about to be dropped (ppo_xp_sync also waits for the copy stream)
        buf._host_slots = slots
        a, b = slots
        while True:
            rc = check(self.batcher._L.ppo_xp_pass_next_host_pipelined(buf.handle, C.byref(a._c), C.byref(b._c)))
            if rc == 1:
                return
            yield a.view(a._c.batch_size, a._c.index)
            a, b = b, a

    def perm(self) -> np.ndarray:
        """The permutation of this pass (host copy)."""
        M = len(self.batcher)
        out = np.empty(M, np.uint32)
        check(self.batcher._L.ppo_xp_pass_perm_host(self.batcher.buffer.handle, out.ctypes.data_as(C.c_void_p), M))
        return out

    def stats(self) -> np.ndarray:
        """[K, 2] (mean, unbiased std) used for each minibatch."""
        K = self.batcher._L.ppo_xp_pass_num_batches(self.batcher.buffer.handle)
        out = np.empty((K, 2), np.float32)
        check(self.batcher._L.ppo_xp_pass_stats_host(self.batcher.buffer.handle, out.ctypes.data_as(C.c_void_p), K))
        return out
