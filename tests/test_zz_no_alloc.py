"""ppo_xp_returns_noalloc (include/ppo_xp.h): the allocation-free variant of ppo_xp_returns, whose
associative scan keeps its chunk maps and carries in a file-scope __device__ array instead of
hipMalloc'ed buffers.  Called through the C ABI and compared bit for bit with ppo_xp_returns on two
calls in a row, the second on a smaller input; sizes beyond the static array are refused.

Collected last on purpose (the round's end runs pytest -x)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _returns(L, lib, h, fn, last_dones, N, algo, M):
    out = np.full(M, np.nan, np.float32)
    lib.check(getattr(L, fn)(h, last_dones.ctypes.data_as(C.c_void_p), N, algo, out.ctypes.data_as(C.c_void_p)))
    return out


def test_returns_noalloc_matches_returns_twice_then_smaller():
    import torch
    assert torch.cuda.is_available()
    import ppo_amd
    from ppo_amd import _lib, synth
    L = _lib.load()
    N, D, T = 320, 2, 4096
    r = synth.rollout(T, N, D, 31, 0.02)
    buf = ppo_amd.ExperienceBuffer(N, D, T)
    h = buf.handle
    ld = np.ascontiguousarray(r["last_dones"], np.uint8)
    assoc = _lib.GAE_ASSOC
    buf.add_many(r["obs"], r["rewards"], r["actions"], r["values"], r["dones"], r["neglogps"])
    want = _returns(L, _lib, h, "ppo_xp_returns", ld, N, assoc, T * N)          # allocates its scan buffers
    buf.sync()
    free0 = torch.cuda.mem_get_info(0)[0]
    got = _returns(L, _lib, h, "ppo_xp_returns_noalloc", ld, N, assoc, T * N)   # first call: large input
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # second call in a row, SMALLER input (fewer steps: other chunking, stale entries of the first call behind it)
    T2 = 700
    buf.reset_counter()
    buf.add_many(r["obs"][:T2], r["rewards"][:T2], r["actions"][:T2], r["values"][:T2], r["dones"][:T2], r["neglogps"][:T2])
    got2 = _returns(L, _lib, h, "ppo_xp_returns_noalloc", ld, N, assoc, T2 * N)
    assert torch.cuda.mem_get_info(0)[0] == free0                                # the device lost no memory to either call
    want2 = _returns(L, _lib, h, "ppo_xp_returns", ld, N, assoc, T2 * N)
    assert np.array_equal(got2.view(np.uint32), want2.view(np.uint32))
    # a forced chunk length, the sequential algorithm (no scratch at all) and a pass on top of the records it built
    for algo in (assoc | _lib.gae_chunk(33), _lib.GAE_SEQUENTIAL):
        a = _returns(L, _lib, h, "ppo_xp_returns_noalloc", ld, N, algo, T2 * N)
        b = _returns(L, _lib, h, "ppo_xp_returns", ld, N, algo, T2 * N)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    _lib.check(L.ppo_xp_returns_noalloc(h, ld.ctypes.data_as(C.c_void_p), N, assoc, None))
    adv_a = [b.advantages.clone() for b in ppo_amd.ExperienceBatcher(buf, 1000).into_iter(seed=3)]
    _lib.check(L.ppo_xp_returns(h, ld.ctypes.data_as(C.c_void_p), N, assoc, None))
    adv_b = [b.advantages.clone() for b in ppo_amd.ExperienceBatcher(buf, 1000).into_iter(seed=3)]
    assert len(adv_a) == len(adv_b) > 0 and all(torch.equal(x.view(torch.int32), y.view(torch.int32)) for x, y in zip(adv_a, adv_b))
    # a second handle on the same GPU shares the array (sequentially: fine)
    buf2 = ppo_amd.ExperienceBuffer(64, 1, 512)
    r2 = synth.rollout(512, 64, 1, 5, 0.05)
    buf2.add_many(r2["obs"], r2["rewards"], r2["actions"], r2["values"], r2["dones"], r2["neglogps"])
    ld2 = np.ascontiguousarray(r2["last_dones"], np.uint8)
    a = _returns(L, _lib, buf2.handle, "ppo_xp_returns_noalloc", ld2, 64, assoc, 512 * 64)
    b = _returns(L, _lib, buf2.handle, "ppo_xp_returns", ld2, 64, assoc, 512 * 64)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    buf2.close()
    buf.close()


def test_returns_noalloc_rejects_sizes_beyond_the_static_array():
    import ppo_amd
    from ppo_amd import _lib, synth
    L = _lib.load()
    N, T = 1024, 2048                       # chunk length 1 -> 2048 chunks x 1024 envs = 2^21 maps > 2^20
    buf = ppo_amd.ExperienceBuffer(N, 1, T)
    r = synth.rollout(T, N, 1, 8, 0.02)
    buf.add_many(r["obs"], r["rewards"], r["actions"], r["values"], r["dones"], r["neglogps"])
    ld = np.ascontiguousarray(r["last_dones"], np.uint8)
    out = np.zeros(T * N, np.float32)
    rc = L.ppo_xp_returns_noalloc(buf.handle, ld.ctypes.data_as(C.c_void_p), N, _lib.GAE_ASSOC | _lib.gae_chunk(1),
                                  out.ctypes.data_as(C.c_void_p))
    assert rc == _lib.EINVAL and "static scratch" in _lib.last_error()
    assert not out.any()                    # refused before anything ran
    # chunk length 2 fits exactly (1024 chunks x 1024 envs = 2^20) and matches the allocating entry point
    a = _returns(L, _lib, buf.handle, "ppo_xp_returns_noalloc", ld, N, _lib.GAE_ASSOC | _lib.gae_chunk(2), T * N)
    b = _returns(L, _lib, buf.handle, "ppo_xp_returns", ld, N, _lib.GAE_ASSOC | _lib.gae_chunk(2), T * N)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    buf.close()
