"""ExperienceBuffer — host-side mirror of ppo/src/common/buffer.rs over the C ABI.

Same method names, argument meaning and failure behaviour as the reference:
  new / new_with_params  buffer.rs:45-69      add_experience   buffer.rs:84-121
  reset_counter          buffer.rs:131-133    training_views   buffer.rs:142-175
  returns                buffer.rs:187-225
Where the reference panics (assert_eq! on sizes) these raise PpoXpPanic (an AssertionError).
All compute happens in libppo_xp.so on the GPU; torch is only used to hand out device memory
as tensors (zero copy) — it is plumbing, not the product.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import PpoXpError, PpoXpPanic, check


def _np(a, dtype, what):
    try:
        import torch
        if isinstance(a, torch.Tensor):
            a = a.detach().cpu().numpy()
    except ImportError:  # pragma: no cover
        pass
    return np.ascontiguousarray(a, dtype=dtype)


class _DevArray:
    """Raw device memory exposed through __cuda_array_interface__ (zero copy into torch)."""

    def __init__(self, ptr, shape, typestr, owner):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}
        self._owner = owner  # keeps the handle alive


class _Pinned:
    """Owner of one ppo_xp_host_alloc block; frees it when the last numpy view is gone."""

    def __init__(self, nbytes):
        self._L = _lib.load()
        self.ptr = self._L.ppo_xp_host_alloc(nbytes)
        if not self.ptr:
            raise _lib.PpoXpError(_lib.ENOMEM, "pinned host allocation failed")

    def __del__(self):
        try:
            self._L.ppo_xp_host_free(C.c_void_p(self.ptr))
        except Exception:
            pass


def pinned_empty(shape, dtype=np.float32) -> np.ndarray:
    """numpy array in page-locked host memory (ppo_xp_host_alloc): rows handed to add_experience /
    add_many from such arrays cross PCIe by DMA instead of being staged by the runtime."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape))
    owner = _Pinned(max(n * dtype.itemsize, 1))
    buf = (C.c_char * max(n * dtype.itemsize, 1)).from_address(owner.ptr)
    buf._ppo_owner = owner                      # the ctypes object keeps the block alive for its views
    return np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)


def device_tensor(ptr, shape, dtype, device_index, owner=None):
    """torch view of `shape` elements of `dtype` at device pointer `ptr` (no copy)."""
    import torch
    typestr = {torch.float32: "<f4", torch.int64: "<i8", torch.uint32: "<u4", torch.float64: "<f8",
               torch.int32: "<i4", torch.uint8: "|u1"}[dtype]
    n = int(np.prod(shape))
    if n == 0 or not ptr:
        t = torch.empty(tuple(shape), dtype=dtype if dtype != torch.uint32 else torch.int32,
                        device=f"cuda:{device_index}")
        t._ppo_owner = owner
        return t
    if dtype == torch.uint32:  # torch has no arithmetic on uint32: view the same bits as int32
        typestr, dtype = "<i4", torch.int32
    t = torch.as_tensor(_DevArray(ptr, shape, typestr, owner), device=f"cuda:{device_index}")
    t._ppo_owner = owner  # the ExperienceBuffer whose memory this is (kept alive with the view)
    return t


class ExperienceBuffer:
    """ExperienceBuffer<NUM_ENVS, OBS_SIZE> (buffer.rs:23-39); the const generics are run-time here."""

    def __init__(self, num_envs: int, obs_size: int, capacity: int, gamma: float = 0.99, lam: float = 0.95,
                 device: int = 0):
        self._h = C.c_void_p()
        self.num_envs, self.obs_size, self.capacity = int(num_envs), int(obs_size), int(capacity)
        self.gamma, self.lam, self.device = float(gamma), float(lam), int(device)
        L = _lib.load()
        check(L.ppo_xp_create(C.byref(self._h), self.device, self.capacity, self.num_envs, self.obs_size,
                              self.gamma, self.lam))
        self._L = L
        self.streaming = False

    # -- constructors with the reference's names ------------------------------------------
    @classmethod
    def new(cls, num_envs, obs_size, capacity, device=0):
        """ExperienceBuffer::new (buffer.rs:45-47): gamma 0.99, lambda 0.95."""
        return cls(num_envs, obs_size, capacity, 0.99, 0.95, device)

    @classmethod
    def new_with_params(cls, num_envs, obs_size, capacity, gamma, lam, device=0):
        """ExperienceBuffer::new_with_params (buffer.rs:52-69)."""
        return cls(num_envs, obs_size, capacity, gamma, lam, device)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.ppo_xp_destroy(self._h)       # synchronises the handle's stream first
            self._h = C.c_void_p()
            self._pass_perm = None                # a borrowed permutation is no longer read by anything
            self._host_slots = None               # nor are host slots of a pipelined hand-off written any more
            self._ext_stream = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    # -- insert -----------------------------------------------------------------------------
    def add_experience(self, obs, rewards, actions, vals, dones, neglogps):
        """buffer.rs:84-121.  obs [NUM_ENVS, OBS_SIZE]; the rest [NUM_ENVS]; dones are bools."""
        obs = _np(obs, np.float32, "obs")
        rewards, vals, neglogps = (_np(x, np.float32, n) for x, n in ((rewards, "rewards"), (vals, "vals"), (neglogps, "neglogps")))
        actions = _np(actions, np.uint64, "actions")
        dones = _np(dones, np.uint8, "dones")
        if obs.ndim != 2 or obs.shape[1] != self.obs_size:
            raise PpoXpPanic(_lib.EINVAL, f"add_experience: obs must be [n, {self.obs_size}] (obs: &[[f32; OBS_SIZE]], buffer.rs:86)")
        n = obs.shape[0]
        for name, a in (("rewards", rewards), ("actions", actions), ("vals", vals), ("dones", dones), ("neglogps", neglogps)):
            if a.ndim != 1 or a.shape[0] != self.num_envs or n != self.num_envs:
                raise PpoXpPanic(_lib.EINVAL, f"add_experience: {name}/obs length != NUM_ENVS (reference panics, buffer.rs:97-112)")
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        check(self._L.ppo_xp_add(self._h, p(obs), p(rewards), p(actions), p(vals), p(dones), p(neglogps), n))

    def add_many(self, obs, rewards, actions, vals, dones, neglogps):
        """`steps` add_experience calls from [steps, N, ...] host arrays."""
        obs = _np(obs, np.float32, "obs")
        rewards, vals, neglogps = (_np(x, np.float32, "") for x in (rewards, vals, neglogps))
        actions = _np(actions, np.uint64, "actions")
        dones = _np(dones, np.uint8, "dones")
        steps = rewards.shape[0]
        if obs.shape != (steps, self.num_envs, self.obs_size) or any(
                a.shape != (steps, self.num_envs) for a in (rewards, actions, vals, dones, neglogps)):
            raise PpoXpPanic(_lib.EINVAL, "add_many: shape mismatch")
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        check(self._L.ppo_xp_add_many(self._h, p(obs), p(rewards), p(actions), p(vals), p(dones), p(neglogps),
                                      self.num_envs, steps))

    def add_experience_device(self, obs, rewards, actions, vals, dones_f32, neglogps):
        """add_experience from torch CUDA tensors (f32 / i64 / f32 / f32 0-1 / f32), device to device."""
        import torch
        ts = [obs, rewards, actions, vals, dones_f32, neglogps]
        want = [torch.float32, torch.float32, torch.int64, torch.float32, torch.float32, torch.float32]
        for t, w in zip(ts, want):
            if not (t.is_cuda and t.is_contiguous() and t.dtype == w):
                raise PpoXpPanic(_lib.EINVAL, "add_experience_device: contiguous CUDA tensors of f32/f32/i64/f32/f32/f32 expected")
        if obs.shape != (self.num_envs, self.obs_size) or any(t.shape != (self.num_envs,) for t in ts[1:]):
            raise PpoXpPanic(_lib.EINVAL, "add_experience_device: length != NUM_ENVS (reference panics, buffer.rs:97-112)")
        # the six copies are only ENQUEUED on the handle's stream: order them after the producers (torch's
        # current stream) and order torch's stream after them, so that neither an in-place update of the
        # source tensors by the next simulator step nor the caching allocator recycling a temporary can
        # race with a pending copy (no host synchronisation either way)
        cur, hs = self._order_streams()
        if hs is not None:
            hs.wait_stream(cur)
        check(self._L.ppo_xp_add_device(self._h, *(C.c_void_p(t.data_ptr()) for t in ts), self.num_envs))
        if hs is not None:
            cur.wait_stream(hs)

    def _order_streams(self):
        """(torch's current stream, the handle's stream as a torch ExternalStream) — the second is None
        when the handle already enqueues on torch's current stream (use_torch_stream())."""
        import torch
        cur = torch.cuda.current_stream(self.device)
        ptr = self.stream_ptr or 0
        if ptr == cur.cuda_stream:
            return cur, None
        ext = getattr(self, "_ext_stream", None)
        if ext is None or ext.cuda_stream != ptr:
            ext = self._ext_stream = torch.cuda.ExternalStream(ptr, device=torch.device("cuda", self.device))
        return cur, ext

    def reset_counter(self):
        """buffer.rs:131-133."""
        self._L.ppo_xp_reset_counter(self._h)

    def __len__(self):
        """buffer.rs:123-129 (private `len`)."""
        return self._L.ppo_xp_len(self._h)

    @property
    def counter(self):
        return self._L.ppo_xp_counter(self._h)

    @property
    def last_values_idx(self):
        return self._L.ppo_xp_last_values_idx(self._h)

    # -- views / returns --------------------------------------------------------------------
    def _views(self):
        v = _lib.ViewsT()
        check(self._L.ppo_xp_views(self._h, C.byref(v)))
        return v

    def training_views(self):
        """buffer.rs:142-175 → (observations [M, D], actions [M] i64, values [M], neglogps [M]):
        zero-copy torch views of device memory, storage order."""
        import torch
        v = self._views()
        M, D, d = v.num_steps, self.obs_size, self.device
        self.sync()
        return (device_tensor(v.observations, (M, D), torch.float32, d, self),
                device_tensor(v.actions, (M,), torch.int64, d, self),
                device_tensor(v.values, (M,), torch.float32, d, self),
                device_tensor(v.neglogps, (M,), torch.float32, d, self))

    def returns(self, last_dones, algo: int = _lib.GAE_SEQUENTIAL, to_host: bool = False):
        """buffer.rs:187-225.  Returns a torch device view [M] (or a numpy array if to_host)."""
        ld = _np(last_dones, np.uint8, "last_dones")
        if ld.ndim != 1:
            raise PpoXpPanic(_lib.EINVAL, "returns: last_dones must be 1-D")
        M = len(self) * self.num_envs
        host = np.empty(M, np.float32) if to_host else None
        check(self._L.ppo_xp_returns(self._h, ld.ctypes.data_as(C.c_void_p), ld.shape[0], int(algo),
                                     host.ctypes.data_as(C.c_void_p) if to_host else None))
        if to_host:
            return host
        import torch
        v = self._views()      # may enqueue the kernel that unpacks the flat arrays from the records
        self.sync()
        return device_tensor(v.returns, (M,), torch.float32, self.device, self)

    def explained_variance(self) -> float:
        """explained_variance(values, returns) of train() (training.rs:88-99), after returns()."""
        out = C.c_float()
        check(self._L.ppo_xp_explained_variance(self._h, C.byref(out)))
        return out.value

    def explained_variance_sums(self) -> np.ndarray:
        """This rank's (M, sum ret, sum ret^2, sum adv, sum adv^2) in f64: what an env-sharded job adds up
        over the ranks before sharded.explained_variance finishes the formula."""
        out = np.zeros(5, np.float64)
        check(self._L.ppo_xp_explained_variance_sums(self._h, out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def advantages(self):
        """`returns - values` of ExperienceBatcher::new (data.rs:62), device view [M]."""
        import torch
        v = self._views()
        self.sync()
        return device_tensor(v.advantages, (v.num_steps,), torch.float32, self.device, self)

    def stream_view(self, name):
        """Device view of one raw stream in storage order over rows 0..len ('rewards', 'dones', ...)."""
        import torch
        v = self._views()
        dt = torch.int64 if name == "actions" else torch.float32
        shape = (v.num_steps, self.obs_size) if name == "observations" else (v.num_steps,)
        self.sync()
        return device_tensor(getattr(v, name), shape, dt, self.device, self)

    # -- plumbing ---------------------------------------------------------------------------
    def sync(self):
        check(self._L.ppo_xp_sync(self._h))

    @property
    def stream_ptr(self):
        return self._L.ppo_xp_stream(self._h)

    def use_stream(self, stream_ptr):
        """Enqueue this handle's work on a caller-owned hipStream_t (e.g. torch's current stream)."""
        check(self._L.ppo_xp_set_stream(self._h, C.c_void_p(stream_ptr)))

    def use_torch_stream(self):
        import torch
        self.use_stream(torch.cuda.current_stream(self.device).cuda_stream)

    # -- env-sharded multi-GPU: the one exchange step ------------------------
    def set_exchange(self, mode: int, comm=None):
        """0 none, 1 library calls ncclAllReduce on `comm`, 2 caller all-reduces pass_sums()."""
        check(self._L.ppo_xp_set_exchange(self._h, int(mode), C.c_void_p(comm) if comm else None))

    def set_streaming(self, on: bool = True):
        """Gather minibatches one at a time into two alternating slots (half the memory of a pass, a few percent
        slower) instead of all K at once; a batch handed out by the iterator is overwritten by the batch after next."""
        check(self._L.ppo_xp_set_streaming(self._h, 1 if on else 0))
        self.streaming = bool(on)

    def pass_sums(self, K):
        """Device view [K, 3] f64 of this rank's (count, sum, sum of squares) per minibatch."""
        import torch
        return device_tensor(self._L.ppo_xp_pass_sums_device(self._h), (K, 3), torch.float64, self.device, self)

    def pass_finish(self):
        check(self._L.ppo_xp_pass_finish(self._h))

    def set_tunable(self, name, value):
        check(self._L.ppo_xp_set_tunable(self._h, name.encode(), int(value)))

    def prof_enable(self, on=True):
        """True / False, or an iterable of kernel names (see _lib.K_NAMES) to time only those."""
        if on is True:
            mask = -1
        elif not on:
            mask = 0
        else:
            mask = sum(1 << _lib.K_NAMES.index(n) for n in on)
        check(self._L.ppo_xp_prof_enable(self._h, mask))

    def prof_collect(self):
        """{kernel name: (total ms, launches)} since the last collect (synchronises)."""
        ms = (C.c_double * _lib.K_COUNT)()
        cnt = (C.c_uint64 * _lib.K_COUNT)()
        check(self._L.ppo_xp_prof_collect(self._h, ms, cnt))
        return {_lib.K_NAMES[i]: (ms[i], int(cnt[i])) for i in range(_lib.K_COUNT)}


__all__ = ["ExperienceBuffer", "PpoXpError", "PpoXpPanic", "device_tensor"]
