"""ppo_xp_pass_begin_bijection on the GPU, through the C ABI: the permutation of a pass as a keyed bijection the
gather kernels evaluate in registers (ppo_amd/csrc/bijection.h).  Replaces into_iter's shuffle (data.rs:91-102).

  * indices bit-equal to the oracle's CPU twin for M in {1, 2, 1000, 2^23, a non-power-of-two >= 10^6};
  * a bijection of 0..M at TGT size;
  * minibatches bit-identical (rows, scalars AND standardised advantages: same kernels, same tiles) to
    ppo_xp_pass_begin_device fed the written-out permutation, for every way the bijection is evaluated;
  * the whole pass against the oracle at the usual bars (tests/test_gpu_parity.check_pass).
The specific permutation is parity unpinned like every shuffle of this path (data.rs:94 is OS-seeded)."""
import ctypes as C

import numpy as np
import pytest

from test_gpu_fullsize import device_fill
from test_gpu_parity import bits, both, check_pass

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def xp():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import ppo_amd
    return ppo_amd


@pytest.mark.parametrize("T,N,D", [(1, 1, 3), (1, 2, 4), (2, 1, 4), (10, 100, 8), (128, 65536, 4), (101, 10007, 4)])
def test_indices_equal_the_cpu_twin(xp, oracle, T, N, D):
    import torch
    M = T * N
    buf = xp.ExperienceBuffer(N, D, T)
    z = torch.zeros
    for t in range(T):      # contents do not matter here
        buf.add_experience_device(z(N, D, device="cuda"), z(N, device="cuda"), z(N, dtype=torch.int64, device="cuda"),
                                  z(N, device="cuda"), z(N, device="cuda"), z(N, device="cuda"))
    buf.returns(np.zeros(N, np.uint8))
    for seed in (0, 0x5EED, 2**64 - 1):
        it = xp.ExperienceBatcher(buf, max(1, M // 4)).into_iter(seed=seed, bijection=True)
        np.testing.assert_array_equal(it.perm(), oracle.bijection_perm(seed, M))
    buf.close()


@pytest.mark.parametrize("D,B,streaming", [(64, 4096, False), (64, 4000 - 32, False), (64, 4096, True), (32, 1024, False),
                                           (256, 512, False), (6, 1000, False), (64, 77, False)])
def test_pass_against_the_oracle(xp, oracle, D, B, streaming):
    """Fused route (D = 32 / 64 / 256 with whole tiles), the two-kernel route (odd D, ragged B), overlapping tail windows,
    streamed minibatches."""
    T, N = 16, 500
    g, o, r = both(xp, oracle, T, N, D, T, seed=3)
    g.set_streaming(streaming)
    check_pass(xp, oracle, g, o, r["last_dones"], B, seed=17, bijection=True)
    g.close()


def _collect(xp, buf, B, **kw):
    out = []
    it = xp.ExperienceBatcher(buf, B).into_iter(**kw)
    for b in it:
        out.append([t.clone() for t in (b.observations, b.actions, b.values, b.neglogps, b.returns, b.advantages)])
    return it, out


def _same(a, b):
    import torch
    assert len(a) == len(b)
    for ma, mb in zip(a, b):
        for x, y in zip(ma, mb):
            assert torch.equal(x.view(torch.int32) if x.dtype == torch.float32 else x,
                               y.view(torch.int32) if y.dtype == torch.float32 else y)


@pytest.mark.parametrize("T,N,D,B", [(32, 1024, 64, 4096), (32, 1000, 64, 4000 - 64), (8, 300, 128, 160), (16, 512, 10, 1024)])
def test_every_evaluation_mode_equals_the_array_pass(xp, T, N, D, B):
    import torch
    buf = xp.ExperienceBuffer(N, D, T)
    r = device_fill(buf, T, N, D, 5)
    buf.returns(r["last_dones"].cpu().numpy())
    M = T * N
    it, base = _collect(xp, buf, B, seed=21, bijection=True)
    perm = torch.from_numpy(it.perm().astype(np.int32)).cuda()
    _, want = _collect(xp, buf, B, perm=perm)
    _same(base, want)
    for mode in (0, 2, 1):
        buf.set_tunable("bij_mode", mode)
        _, got = _collect(xp, buf, B, seed=21, bijection=True)
        _same(got, want)
    buf.set_tunable("fused", 0)
    _, got = _collect(xp, buf, B, seed=21, bijection=True)
    _same(got, want)
    buf.close()


def test_tgt_size_bijection_and_rows(xp):
    """TGT in full: the written-out bijection is a permutation of 0..M, the inline pass gathers exactly source[perm],
    and standardises within 2e-6 of an fp64 evaluation."""
    import torch
    from ppo_amd import synth
    T, N, D, B = synth.CONFIGS["TGT"]
    M = T * N
    buf = xp.ExperienceBuffer(N, D, T)
    r = device_fill(buf, T, N, D, 77)
    ret = buf.returns(r["last_dones"].cpu().numpy()).clone()
    adv = buf.advantages().clone()
    it = xp.ExperienceBatcher(buf, B).into_iter(seed=123456789, bijection=True)
    assert xp._lib.load().ppo_xp_pass_gather_kind(buf.handle) == 2       # fused, four row groups per workgroup (chosen by size)
    batches = [[t.clone() for t in (b.observations, b.actions, b.values, b.neglogps, b.returns, b.advantages)] for b in it]
    perm = xp.device_tensor(xp._lib.load().ppo_xp_pass_perm_device(buf.handle), (M,), torch.uint32, 0, buf).long()
    assert torch.equal(torch.sort(perm).values, torch.arange(M, device="cuda"))
    obs = r["obs"].reshape(M, D)
    assert len(batches) == M // B
    for k, (o, a, v, n, rt, ad) in enumerate(batches):
        idx = perm[k * B:(k + 1) * B]
        assert torch.equal(o, obs[idx])
        assert torch.equal(a, r["actions"].reshape(-1)[idx])
        assert torch.equal(v, r["values"].reshape(-1)[idx])
        assert torch.equal(n, r["neglogps"].reshape(-1)[idx])
        assert torch.equal(rt, ret[idx])
        x = adv[idx].double()
        ref = (x - x.mean()) / (x.std(unbiased=True) + 1e-8)
        assert (ad.double() - ref).abs().max().item() < 2e-6
    buf.close()
    torch.cuda.empty_cache()


@pytest.mark.parametrize("T,N,D,B,bijection", [(16, 256, 64, 512, True), (9, 130, 16, 320, False), (5, 64, 8, 320, False)])
def test_pipelined_host_handoff_matches_the_oracle(xp, oracle, T, N, D, B, bijection):
    """ppo_xp_pass_next_host_pipelined (data.rs:166-182 as the caller sees it, with the download of minibatch k+1
    overlapping the consumer's work on k): every host batch against the oracle, bit for bit, under the overlapped
    schedule - the consumer scribbles over a slot after use, so a copy that lands late or in the wrong slot shows."""
    g, o, r = both(xp, oracle, T, N, D, T, seed=11)
    ret_o = o.returns(r["last_dones"])
    g.returns(r["last_dones"])
    obs_o, act_o, val_o, nlp_o = o.training_views()
    adv_o = oracle.advantages(ret_o, val_o)
    M = T * N
    batcher = xp.ExperienceBatcher(g, B)
    for rep in range(2):
        it = batcher.into_iter(seed=40 + rep, bijection=bijection)
        perm = it.perm()
        np.testing.assert_array_equal(perm, oracle.bijection_perm(40 + rep, M) if bijection else oracle.shuffle(40 + rep, M))
        windows = oracle.batch_windows(M, B)
        n = 0
        for k, b in enumerate(it.host_batches()):
            s, e = windows[k]
            idx = perm[s:e]
            assert b.index == k
            np.testing.assert_array_equal(bits(b.observations), bits(oracle.gather_f32(obs_o, idx)))
            np.testing.assert_array_equal(b.actions, oracle.gather_u64(act_o, idx))
            np.testing.assert_array_equal(bits(b.values), bits(oracle.gather_f32(val_o, idx)))
            np.testing.assert_array_equal(bits(b.neglogps), bits(oracle.gather_f32(nlp_o, idx)))
            np.testing.assert_array_equal(bits(b.returns), bits(oracle.gather_f32(ret_o, idx)))
            want64, _, s64 = oracle.standardize_f64(oracle.gather_f32(adv_o, idx))
            if s64 > 1e-3:
                np.testing.assert_allclose(b.advantages, want64, rtol=0, atol=2e-6 * max(1.0, np.abs(want64).max()) + 1e-6)
            for a in (b.observations, b.values, b.neglogps, b.returns, b.advantages):
                a.fill(np.nan)
            n += 1
        assert n == len(windows)
    g.set_streaming(True)
    it = batcher.into_iter(seed=1)
    with pytest.raises(xp.PpoXpError):
        next(it.host_batches())
    g.close()


def test_library_gives_the_committed_vectors(xp):
    """tests/golden/bijection_kats.json: whole arrays for small M, head / tail / CRC-32 of the whole array for 2^23 and 1 010 707."""
    import json
    import os
    import zlib
    import torch
    kats = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bijection_kats.json")))
    shapes = {1: (1, 1), 2: (1, 2), 5: (1, 5), 257: (1, 257), 1000: (10, 100), 65536: (16, 4096), 1 << 23: (128, 65536), 1010707: (101, 10007)}
    z = torch.zeros
    for c in kats["cases"]:
        T, N = shapes[c["M"]]
        buf = xp.ExperienceBuffer(N, 4, T)
        for t in range(T):
            buf.add_experience_device(z(N, 4, device="cuda"), z(N, device="cuda"), z(N, dtype=torch.int64, device="cuda"),
                                      z(N, device="cuda"), z(N, device="cuda"), z(N, device="cuda"))
        buf.returns(np.zeros(N, np.uint8))
        p = xp.ExperienceBatcher(buf, max(1, c["M"] // 2)).into_iter(seed=int(c["seed"]), bijection=True).perm()
        assert p[:16].tolist() == c["head"] and p[-4:].tolist() == c["tail"] and zlib.crc32(p.tobytes()) == c["crc32"]
        buf.close()


def test_abandoned_pipelined_pass_and_set_stream_to_itself(xp, oracle):
    """A consumer that stops after the first host minibatch leaves a download in flight: the next pass, new slots and a
    ppo_xp_sync must all be ordered behind it; ppo_xp_set_stream with the stream the handle already uses changes nothing."""
    g, o, r = both(xp, oracle, 16, 256, 64, 16, seed=2)
    g.returns(r["last_dones"])
    batcher = xp.ExperienceBatcher(g, 512)
    it = batcher.into_iter(seed=1, bijection=True)
    first = next(it.host_batches())
    keep = first.observations.copy()
    del it, first                                   # minibatch 1 may still be travelling into the second slot
    it = batcher.into_iter(seed=1, bijection=True)
    batches = [b.observations.copy() for b in it.host_batches()]
    assert len(batches) == 8 and np.array_equal(batches[0], keep)
    own = g.stream_ptr
    g.use_stream(own)                               # same stream: must not destroy it
    assert g.stream_ptr == own
    it = batcher.into_iter(seed=1, bijection=True)
    again = [b.observations.copy() for b in it.host_batches()]
    assert all(np.array_equal(a, b) for a, b in zip(batches, again))
    g.close()


@pytest.mark.parametrize("T,N,B,streaming", [(32, 1024, 4096, False), (32, 1000, 4000 - 32, False), (16, 1024, 2048, True), (8, 512, 128, False)])
def test_row_groups_per_workgroup(xp, T, N, B, streaming):
    """fused_gather_loop_kernel (tunable fused_iter = 2 / 4; chosen by size from 2^20 transitions up): rows, actions and scalar
    streams bit-identical to the array pass; the statistics are summed over 64- / 128-row tiles instead of 32-row ones, so the
    standardised advantages may differ in the last bit of the fp32 mean / std: within 1e-6.  B = 3968 does not tile into 128
    rows (falls back to one group) and gives an overlapping tail window with 2 groups."""
    import torch
    D = 64
    buf = xp.ExperienceBuffer(N, D, T)
    r = device_fill(buf, T, N, D, 6)
    buf.returns(r["last_dones"].cpu().numpy())
    buf.set_streaming(streaming)
    buf.set_tunable("fused_iter", 1)
    it, base = _collect(xp, buf, B, seed=33, bijection=True)
    for iters in (2, 4, 0):
        buf.set_tunable("fused_iter", iters)
        _, got = _collect(xp, buf, B, seed=33, bijection=True)
        assert len(got) == len(base)
        for ma, mb in zip(got, base):
            for x, y in zip(ma[:5], mb[:5]):
                assert torch.equal(x.view(torch.int32) if x.dtype == torch.float32 else x, y.view(torch.int32) if y.dtype == torch.float32 else y)
            assert (ma[5] - mb[5]).abs().max().item() <= 1e-6
    buf.close()
