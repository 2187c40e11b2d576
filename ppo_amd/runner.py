"""RewardNormalizer — the reward side of VecRunner (ppo/src/common/runner.rs:20-66,114-127) with its
RunningMeanStd<Ix0> (ppo/src/common/running_mean_std.rs), resident on the GPU.  Next row of the
path: the step immediately before ExperienceBuffer.add_experience."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import PpoXpPanic, check


class RewardNormalizer:
    def __init__(self, num_envs: int, gamma: float = 0.99, epsilon: float = 1e-8, cliprew: float = 10.0, device: int = 0):
        """VecRunner::new_with_params (runner.rs:43-66): gamma 0.99, epsilon 1e-8, cliprew 10."""
        self.num_envs, self.device = int(num_envs), int(device)
        self._L = _lib.load()
        self._h = C.c_void_p()
        check(self._L.ppo_xp_rnorm_create(C.byref(self._h), self.device, self.num_envs, gamma, epsilon, cliprew))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.ppo_xp_rnorm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def normalized_rewards(self, rewards, dones):
        """runner.rs:114-127.  Host arrays in, numpy out; CUDA tensors in (f32 rewards, f32 0/1 dones), CUDA tensor out."""
        if hasattr(rewards, "is_cuda") and rewards.is_cuda:
            import torch
            if rewards.shape != (self.num_envs,) or dones.shape != (self.num_envs,) or rewards.dtype != torch.float32 \
                    or dones.dtype != torch.float32 or not (rewards.is_contiguous() and dones.is_contiguous()):
                raise PpoXpPanic(_lib.EINVAL, "normalized_rewards: contiguous f32 CUDA tensors of NUM_ENVS expected")
            out = torch.empty_like(rewards)
            stream = torch.cuda.current_stream(self.device).cuda_stream
            # NULL means "the object's own stream" in the ABI; torch's default stream is NULL too, so run
            # there explicitly after the producers have finished
            torch.cuda.current_stream(self.device).synchronize()
            check(self._L.ppo_xp_rnorm_step_device(self._h, C.c_void_p(rewards.data_ptr()), C.c_void_p(dones.data_ptr()),
                                                   C.c_void_p(out.data_ptr()), self.num_envs, C.c_void_p(stream)))
            self.state()  # synchronises the object's stream
            return out
        r = np.ascontiguousarray(rewards, np.float32)
        d = np.ascontiguousarray(dones, np.uint8)
        if r.shape != (self.num_envs,) or d.shape != (self.num_envs,):
            raise PpoXpPanic(_lib.EINVAL, "normalized_rewards: NUM_ENVS rewards and dones expected")
        out = np.empty(self.num_envs, np.float32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        check(self._L.ppo_xp_rnorm_step(self._h, p(r), p(d), p(out), self.num_envs))
        return out

    def normalized_rewards_sharded(self, rewards, dones, all_reduce):
        """runner.rs:114-127 for one rank of an env-sharded job: the batch moments (running_mean_std.rs:35-43)
        are over all ranks' envs.  rewards / dones: CUDA f32 tensors of this rank's envs; `all_reduce(t)`
        sums the 3-double CUDA tensor `t` in place over the ranks (sharded.ShardedRewardNormalizer wires it
        to torch.distributed).  Runs on torch's current stream."""
        import torch
        from .buffer import device_tensor
        if not (hasattr(rewards, "is_cuda") and rewards.is_cuda) or rewards.shape != (self.num_envs,) \
                or dones.shape != (self.num_envs,) or rewards.dtype != torch.float32 or dones.dtype != torch.float32 \
                or not (rewards.is_contiguous() and dones.is_contiguous()):
            raise PpoXpPanic(_lib.EINVAL, "normalized_rewards_sharded: contiguous f32 CUDA tensors of NUM_ENVS expected")
        ts = torch.cuda.current_stream(self.device)
        if ts.cuda_stream == 0:
            ts.synchronize()      # NULL = "the object's own stream" in the ABI: order it after the producers by hand
        stream = C.c_void_p(ts.cuda_stream)
        out = torch.empty_like(rewards)
        check(self._L.ppo_xp_rnorm_step_begin_device(self._h, C.c_void_p(rewards.data_ptr()), self.num_envs, stream))
        sums = device_tensor(self._L.ppo_xp_rnorm_sums_device(self._h), (3,), torch.float64, self.device, self)
        if ts.cuda_stream == 0:
            self.state()          # the object's stream has produced the sums
        all_reduce(sums)
        if ts.cuda_stream == 0:
            ts.synchronize()      # the collective is ordered with torch's stream, not with the object's own
        check(self._L.ppo_xp_rnorm_step_finish_device(self._h, C.c_void_p(rewards.data_ptr()), C.c_void_p(dones.data_ptr()),
                                                      C.c_void_p(out.data_ptr()), self.num_envs, stream))
        self.state()              # synchronises whichever stream ran the step
        return out

    def state(self):
        """(ret_rms.mean, ret_rms.var, count, ret[N])."""
        m, v, c = C.c_float(), C.c_float(), C.c_float()
        ret = np.empty(self.num_envs, np.float32)
        check(self._L.ppo_xp_rnorm_state(self._h, C.byref(m), C.byref(v), C.byref(c), ret.ctypes.data_as(C.c_void_p), self.num_envs))
        return m.value, v.value, c.value, ret
