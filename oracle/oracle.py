"""ctypes front-end of the CPU oracle (oracle/ppo_xp_oracle.c) + a tiny numpy twin.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under ppo_amd/ imports this module.

Every method names the reference function it restates (paths relative to /root/reference).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libppo_xp_oracle.so")


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "ppo_xp_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        vp, sz, f32, i32, u64 = C.c_void_p, C.c_size_t, C.c_float, C.c_int, C.c_uint64
        L.orc_buffer_new.restype = vp
        L.orc_buffer_new.argtypes = [sz, sz, sz, f32, f32]
        L.orc_buffer_free.argtypes = [vp]
        L.orc_buffer_add.restype = i32
        L.orc_buffer_add.argtypes = [vp, vp, vp, vp, vp, vp, vp, sz]
        L.orc_buffer_len.restype = sz
        L.orc_buffer_len.argtypes = [vp]
        L.orc_buffer_reset_counter.argtypes = [vp]
        L.orc_buffer_last_values_idx.restype = sz
        L.orc_buffer_last_values_idx.argtypes = [vp]
        L.orc_buffer_views.argtypes = [vp] + [C.POINTER(vp)] * 4 + [C.POINTER(sz)]
        L.orc_buffer_rewards.restype = vp
        L.orc_buffer_rewards.argtypes = [vp]
        L.orc_buffer_dones.restype = vp
        L.orc_buffer_dones.argtypes = [vp]
        L.orc_buffer_returns.restype = i32
        L.orc_buffer_returns.argtypes = [vp, vp, sz, vp]
        L.orc_gae_rows.restype = i32
        L.orc_gae_rows.argtypes = [vp, vp, vp, sz, sz, sz, vp, f32, f32, vp, i32]
        L.orc_gae_rows_f64.argtypes = [vp, vp, vp, sz, sz, sz, vp, f32, f32, vp]
        L.orc_advantages.argtypes = [vp, vp, sz, vp, i32]
        L.orc_num_batches.restype = sz
        L.orc_num_batches.argtypes = [sz, sz]
        L.orc_batch_window.restype = i32
        L.orc_batch_window.argtypes = [C.POINTER(sz), sz, sz, C.POINTER(sz), C.POINTER(sz)]
        L.orc_gather_f32.argtypes = [vp, sz, vp, sz, vp, i32]
        L.orc_gather_u64.argtypes = [vp, vp, sz, vp, i32]
        L.orc_standardize.argtypes = [vp, sz, vp, C.POINTER(f32), C.POINTER(f32)]
        L.orc_standardize_f64.argtypes = [vp, sz, vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_chacha_block.argtypes = [vp, u64, u64, i32, vp]
        L.orc_seed_from_u64.argtypes = [u64, vp]
        L.orc_rng_stream.argtypes = [u64, sz, vp]
        L.orc_shuffle.restype = i32
        L.orc_shuffle.argtypes = [u64, sz, vp]
        L.orc_run_path.restype = i32
        L.orc_run_path.argtypes = [vp] * 6 + [sz, sz, sz, sz, vp, f32, f32, vp, sz] + [vp] * 8 + [i32]
        L.orc_keyed_shuffle.restype = i32
        L.orc_keyed_shuffle.argtypes = [u64, sz, vp]
        L.orc_bijection_perm.restype = i32
        L.orc_bijection_perm.argtypes = [u64, sz, vp]
        L.orc_bijection_range.restype = i32
        L.orc_bijection_range.argtypes = [u64, sz, sz, sz, i32, vp]
        L.orc_var_welford_f32.restype = f32
        L.orc_var_welford_f32.argtypes = [vp, sz, f32]
        L.orc_explained_variance.restype = f32
        L.orc_explained_variance.argtypes = [vp, vp, sz]
        L.orc_explained_variance_f64.restype = C.c_double
        L.orc_explained_variance_f64.argtypes = [vp, vp, sz]
        L.orc_rnorm_new.restype = vp
        L.orc_rnorm_new.argtypes = [sz, f32, f32, f32]
        L.orc_rnorm_free.argtypes = [vp]
        L.orc_rnorm_step.restype = i32
        L.orc_rnorm_step.argtypes = [vp, vp, vp, vp, sz]
        L.orc_rnorm_state.argtypes = [vp, C.POINTER(f32), C.POINTER(f32), C.POINTER(f32), vp]
        L.orc_max_threads.restype = i32
        _lib = L
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


class OracleBuffer:
    """ExperienceBuffer<NUM_ENVS, OBS_SIZE> — ppo/src/common/buffer.rs:23-226."""

    def __init__(self, num_envs, obs_size, capacity, gamma=0.99, lam=0.95):
        self.N, self.D, self.capacity = int(num_envs), int(obs_size), int(capacity)
        self.gamma, self.lam = float(gamma), float(lam)
        self._h = lib().orc_buffer_new(self.capacity, self.N, self.D, self.gamma, self.lam)

    def __del__(self):
        try:
            if self._h:
                lib().orc_buffer_free(self._h)
                self._h = None
        except Exception:
            pass

    def add_experience(self, obs, rewards, actions, vals, dones, neglogps):
        """buffer.rs:84-121; raises where the reference panics (buffer.rs:97-112)."""
        obs = _c(obs, np.float32)
        rewards, vals, neglogps = (_c(x, np.float32) for x in (rewards, vals, neglogps))
        actions = _c(actions, np.uint64)
        dones = _c(dones, np.uint8)
        lens = {obs.shape[0] if obs.ndim == 2 else -1, rewards.size, actions.size, vals.size, dones.size, neglogps.size}
        if lens != {self.N} or (obs.ndim == 2 and obs.shape[1] != self.D):
            raise AssertionError("add_experience: length mismatch (reference panics, buffer.rs:97-112)")
        rc = lib().orc_buffer_add(self._h, _p(obs), _p(rewards), _p(actions), _p(vals), _p(dones), _p(neglogps), self.N)
        assert rc == 0

    def __len__(self):
        return lib().orc_buffer_len(self._h)

    def reset_counter(self):
        lib().orc_buffer_reset_counter(self._h)

    @property
    def last_values_idx(self):
        return lib().orc_buffer_last_values_idx(self._h)

    def _view(self, ptr, shape, dt):
        n = int(np.prod(shape))
        if n == 0:
            return np.zeros(shape, dt)
        buf = (C.c_char * (n * np.dtype(dt).itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dt).reshape(shape)

    def training_views(self):
        """buffer.rs:142-175: (obs [M,D], actions [M], values [M], neglogps [M]) copies."""
        o, a, v, n = (C.c_void_p() for _ in range(4))
        m = C.c_size_t()
        lib().orc_buffer_views(self._h, C.byref(o), C.byref(a), C.byref(v), C.byref(n), C.byref(m))
        M = m.value
        return (self._view(o.value, (M, self.D), np.float32).copy(), self._view(a.value, (M,), np.uint64).copy(),
                self._view(v.value, (M,), np.float32).copy(), self._view(n.value, (M,), np.float32).copy())

    def raw(self):
        """All six streams in storage order at full capacity (copies)."""
        cn = self.capacity * self.N
        o, a, v, n = (C.c_void_p() for _ in range(4))
        m = C.c_size_t()
        lib().orc_buffer_views(self._h, C.byref(o), C.byref(a), C.byref(v), C.byref(n), C.byref(m))
        return dict(
            obs=self._view(o.value, (cn, self.D), np.float32).copy(),
            actions=self._view(a.value, (cn,), np.uint64).copy(),
            values=self._view(v.value, (cn,), np.float32).copy(),
            neglogps=self._view(n.value, (cn,), np.float32).copy(),
            rewards=self._view(lib().orc_buffer_rewards(self._h), (cn,), np.float32).copy(),
            dones=self._view(lib().orc_buffer_dones(self._h), (cn,), np.float32).copy(),
        )

    def returns(self, last_dones):
        """buffer.rs:187-225."""
        ld = _c(last_dones, np.uint8)
        if ld.size != self.N:
            raise AssertionError("returns: last_dones length mismatch (reference panics, buffer.rs:194)")
        out = np.empty(len(self) * self.N, np.float32)
        rc = lib().orc_buffer_returns(self._h, _p(ld), ld.size, _p(out))
        assert rc == 0
        return out


def gae_rows(rewards, values, dones, last_values_idx, last_dones, gamma, lam, nthreads=1):
    """buffer.rs:187-225 on raw [len,N] arrays."""
    rewards, values, dones = (_c(x, np.float32) for x in (rewards, values, dones))
    T, N = rewards.shape
    ld = _c(last_dones, np.uint8)
    out = np.empty((T, N), np.float32)
    lib().orc_gae_rows(_p(rewards), _p(values), _p(dones), T, N, int(last_values_idx), _p(ld), gamma, lam, _p(out), nthreads)
    return out.reshape(-1)


def gae_rows_f64(rewards, values, dones, last_values_idx, last_dones, gamma, lam):
    rewards, values, dones = (_c(x, np.float32) for x in (rewards, values, dones))
    T, N = rewards.shape
    ld = _c(last_dones, np.uint8)
    out = np.empty((T, N), np.float64)
    lib().orc_gae_rows_f64(_p(rewards), _p(values), _p(dones), T, N, int(last_values_idx), _p(ld), gamma, lam, _p(out))
    return out.reshape(-1)


def advantages(returns, values):
    """data.rs:62."""
    returns, values = _c(returns, np.float32), _c(values, np.float32)
    out = np.empty_like(returns)
    lib().orc_advantages(_p(returns), _p(values), returns.size, _p(out), 1)
    return out


def num_batches(buf_size, batch_size):
    return lib().orc_num_batches(buf_size, batch_size)


def batch_windows(buf_size, batch_size):
    """All (start, end) windows of one pass — data.rs:115-126."""
    it, s, e = C.c_size_t(0), C.c_size_t(), C.c_size_t()
    out = []
    while lib().orc_batch_window(C.byref(it), batch_size, buf_size, C.byref(s), C.byref(e)):
        out.append((s.value, e.value))
        if batch_size == 0:
            break
    return out


def gather_f32(src, idx):
    src = _c(src, np.float32)
    idx = _c(idx, np.uint32)
    row = int(np.prod(src.shape[1:])) if src.ndim > 1 else 1
    out = np.empty((idx.size,) + src.shape[1:], np.float32)
    lib().orc_gather_f32(_p(src), row, _p(idx), idx.size, _p(out), 1)
    return out


def gather_u64(src, idx):
    src = _c(src, np.uint64)
    idx = _c(idx, np.uint32)
    out = np.empty(idx.size, np.uint64)
    lib().orc_gather_u64(_p(src), _p(idx), idx.size, _p(out), 1)
    return out


def standardize(a):
    """data.rs:157-164,196-200 → (out, mean, std)."""
    a = _c(a, np.float32)
    out = np.empty_like(a)
    m, s = C.c_float(), C.c_float()
    lib().orc_standardize(_p(a), a.size, _p(out), C.byref(m), C.byref(s))
    return out, m.value, s.value


def standardize_f64(a):
    a = _c(a, np.float32)
    out = np.empty(a.size, np.float64)
    m, s = C.c_double(), C.c_double()
    lib().orc_standardize_f64(_p(a), a.size, _p(out), C.byref(m), C.byref(s))
    return out, m.value, s.value


def chacha_block(key_words, counter, stream, rounds):
    key = _c(key_words, np.uint32)
    out = np.empty(16, np.uint32)
    lib().orc_chacha_block(_p(key), counter, stream, rounds, _p(out))
    return out


def seed_from_u64(seed):
    key = np.empty(8, np.uint32)
    lib().orc_seed_from_u64(seed, _p(key))
    return key


def rng_stream(seed, n):
    out = np.empty(n, np.uint32)
    lib().orc_rng_stream(seed, n, _p(out))
    return out


def shuffle(seed, M):
    """data.rs:93-95 with rand 0.8.5 semantics over seeded ChaCha12."""
    out = np.empty(M, np.uint32)
    rc = lib().orc_shuffle(seed, M, _p(out))
    assert rc == 0
    return out


def keyed_shuffle(seed, M):
    """CPU twin of the device permutation generator: stable argsort of SplitMix64(seed, i)."""
    out = np.empty(M, np.uint32)
    assert lib().orc_keyed_shuffle(seed, M, _p(out)) == 0
    return out


def np_keyed_shuffle(seed, M):
    """Independent numpy restatement of keyed_shuffle."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (np.arange(M, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return np.argsort(z, kind="stable").astype(np.uint32)


def bijection_perm(seed, M, inverse=False):
    """CPU twin of ppo_xp_pass_begin_bijection: perm(j) (or perm^-1) for every j < M — a keyed Feistel network on
    max(8, ceil(log2 M)) bits, cycle-walked into [0, M) (not in the reference: data.rs:91-96 is the host shuffle)."""
    out = np.empty(M, np.uint32)
    if M:
        assert lib().orc_bijection_range(seed, M, 0, M, int(bool(inverse)), _p(out)) == 0
    return out


def bijection_range(seed, M, first, count, inverse=False):
    out = np.empty(count, np.uint32)
    assert lib().orc_bijection_range(seed, M, first, count, int(bool(inverse)), _p(out)) == 0
    return out


def np_bijection_perm(seed, M, inverse=False):
    """Independent numpy restatement of bijection_perm (vectorised; the walk repeats on the entries still >= M)."""
    u = np.uint64
    with np.errstate(over="ignore"):
        z = u(seed) + (np.arange(6, dtype=u) + u(1)) * u(0x9E3779B97F4A7C15)
        z = (z ^ (z >> u(30))) * u(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> u(27))) * u(0x94D049BB133111EB)
        keys = (z ^ (z >> u(31))) >> u(32)
    n = max(8, int(M - 1).bit_length()) if M else 8
    lo_bits = n // 2
    hi_mask, lo_mask = u((1 << (n - lo_bits)) - 1), u((1 << lo_bits) - 1)
    m24, m32 = u(0xFFFFFF), u(0xFFFFFFFF)

    def fn(v, k):
        t = v ^ (k & u(0xFFFF))
        t = ((t & m24) * u(0xE995)) & m32
        t = t ^ (t >> u(15))
        t = ((t & m24) * u(0x2C1B3)) & m32
        t = t ^ (t >> u(13))
        return (t + (k >> u(16))) & m32

    def network(x):
        hi, lo = x >> u(lo_bits), x & lo_mask
        for r in (range(5, -1, -1) if inverse else range(6)):
            if r % 2 == 0:
                hi = hi ^ (fn(lo, keys[r]) & hi_mask)
            else:
                lo = lo ^ (fn(hi, keys[r]) & lo_mask)
        return (hi << u(lo_bits)) | lo

    x = network(np.arange(M, dtype=u))
    todo = np.nonzero(x >= u(M))[0]
    while todo.size:
        x[todo] = network(x[todo])
        todo = todo[x[todo] >= u(M)]
    return x.astype(np.uint32)


def explained_variance(values, returns):
    """training.rs:88-99 with ndarray's sequential fp32 Welford variance."""
    values, returns = _c(values, np.float32), _c(returns, np.float32)
    return lib().orc_explained_variance(_p(values), _p(returns), values.size)


def explained_variance_f64(values, returns):
    values, returns = _c(values, np.float32), _c(returns, np.float32)
    return lib().orc_explained_variance_f64(_p(values), _p(returns), values.size)


def run_pass(streams, T, N, D, last_values_idx, last_dones, gamma, lam, perm, B, nthreads=1, outs=None):
    """training.rs:186-205 for one shuffled pass on raw streams; returns dict of outputs."""
    M = T * N
    K = num_batches(M, B)
    ld = _c(last_dones, np.uint8)
    perm = _c(perm, np.uint32)
    if outs is None:
        outs = dict(
            returns=np.empty(M, np.float32), adv=np.empty(M, np.float32),
            obs=np.empty((max(K, 1) * B, D), np.float32), actions=np.empty(max(K, 1) * B, np.uint64),
            values=np.empty(max(K, 1) * B, np.float32), neglogps=np.empty(max(K, 1) * B, np.float32),
            ret=np.empty(max(K, 1) * B, np.float32), advn=np.empty(max(K, 1) * B, np.float32))
    s = streams
    k = lib().orc_run_path(_p(s["obs"]), _p(s["rewards"]), _p(s["actions"]), _p(s["values"]), _p(s["dones"]),
                           _p(s["neglogps"]), T, N, D, int(last_values_idx), _p(ld), gamma, lam, _p(perm), B,
                           _p(outs["returns"]), _p(outs["adv"]), _p(outs["obs"]), _p(outs["actions"]),
                           _p(outs["values"]), _p(outs["neglogps"]), _p(outs["ret"]), _p(outs["advn"]), nthreads)
    outs["K"] = k
    return outs


class OracleRewardNormalizer:
    """VecRunner::normalized_rewards + RunningMeanStd<Ix0> (runner.rs:114-127, running_mean_std.rs:16-67)."""

    def __init__(self, num_envs, gamma=0.99, epsilon=1e-8, cliprew=10.0):
        self.N = int(num_envs)
        self._h = lib().orc_rnorm_new(self.N, gamma, epsilon, cliprew)

    def __del__(self):
        try:
            lib().orc_rnorm_free(self._h)
        except Exception:
            pass

    def normalized_rewards(self, rewards, dones):
        rewards, dones = _c(rewards, np.float32), _c(dones, np.uint8)
        out = np.empty(self.N, np.float32)
        if lib().orc_rnorm_step(self._h, _p(rewards), _p(dones), _p(out), rewards.size) != 0:
            raise AssertionError("normalized_rewards: length mismatch")
        return out

    def state(self):
        m, v, c = C.c_float(), C.c_float(), C.c_float()
        ret = np.empty(self.N, np.float32)
        lib().orc_rnorm_state(self._h, C.byref(m), C.byref(v), C.byref(c), _p(ret))
        return m.value, v.value, c.value, ret


def max_threads():
    return lib().orc_max_threads()


# --------------------------------------------------------------------------------------
# numpy twin of the GAE scan (independent second restatement, used to cross-check the C one)
# --------------------------------------------------------------------------------------
def np_gae(rewards, values, dones, last_values_idx, last_dones, gamma, lam):
    """buffer.rs:199-222 op-for-op in numpy float32 (one ufunc call per reference op)."""
    f = np.float32
    rewards, values, dones = (np.asarray(x, f) for x in (rewards, values, dones))
    T, N = rewards.shape
    gamma, lam = f(gamma), f(lam)
    gl = f(gamma * lam)
    nonterminals = f(1.0) - dones                                       # buffer.rs:199
    last_nt = (~np.asarray(last_dones, bool)).astype(f)                 # buffer.rs:201
    last_values = values[last_values_idx]                               # buffer.rs:204
    lastgaelam = np.zeros(N, f)
    advs = np.zeros((T, N), f)
    for t in range(T - 1, -1, -1):
        nv, nnt = (last_values, last_nt) if t == T - 1 else (values[t + 1], nonterminals[t + 1])
        delta = (rewards[t] + (nv * nnt) * gamma) - values[t]           # buffer.rs:217-218
        lastgaelam = delta + (nnt * lastgaelam) * gl                    # buffer.rs:219
        advs[t] = lastgaelam
    return (advs + values).reshape(-1)                                  # buffer.rs:222
