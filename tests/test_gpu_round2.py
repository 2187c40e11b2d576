"""Round-2 additions, through the C ABI on the GPU, against the CPU oracle:
  * ppo_xp_pass_next_host: the minibatch in caller host memory (what data.rs:166-182 feeds copy_from);
  * ppo_xp_shuffle_prefetch: the worker-thread Fisher-Yates gives the permutation of the same seed;
  * a packing ppo_xp_returns no longer stores the flat returns / advantages: views, host copies and
    explained_variance must still give the same bits (buffer.rs:222, data.rs:62, training.rs:88-99);
  * stream-ordering and lifetime cases of the Python mirror (device inserts, borrowed permutations,
    the RCCL exchange on the handle's own stream, slabs that must not move between passes)."""
import ctypes as C
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32 if a.dtype.itemsize == 4 else np.uint64)


@pytest.fixture(scope="module")
def xp():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import ppo_amd
    return ppo_amd


def _both(xp, oracle, T, N, D, seed, p_done=0.1, big_actions=False):
    from ppo_amd import synth
    r = synth.rollout(T, N, D, seed, p_done)
    if big_actions:
        r["actions"][::7] = (1 << 40) + 7
    g, o = xp.ExperienceBuffer(N, D, T), oracle.OracleBuffer(N, D, T)
    g.add_many(r["obs"], r["rewards"], r["actions"], r["values"], r["dones"], r["neglogps"])
    for t in range(T):
        o.add_experience(r["obs"][t], r["rewards"][t], r["actions"][t], r["values"][t], r["dones"][t], r["neglogps"][t])
    return g, o, r


@pytest.mark.parametrize("pinned", [True, False])
@pytest.mark.parametrize("T,N,D,B", [(16, 50, 12, 128), (5, 2, 3, 6), (9, 33, 64, 297)])
def test_host_batches_match_oracle(xp, oracle, T, N, D, B, pinned):
    g, o, r = _both(xp, oracle, T, N, D, 11 + T, big_actions=True)
    ret_o = o.returns(r["last_dones"])
    g.returns(r["last_dones"])
    obs_o, act_o, val_o, nlp_o = o.training_views()
    adv_o = oracle.advantages(ret_o, val_o)
    M = T * N
    batcher = xp.ExperienceBatcher(g, B)
    batcher.prefetch_shuffle(901)                      # drawn by the library's worker thread
    it = batcher.into_iter(seed=901)
    perm = it.perm()
    np.testing.assert_array_equal(perm, oracle.shuffle(901, M))
    windows = oracle.batch_windows(M, B)
    hb = (xp.HostBatch.pinned if pinned else xp.HostBatch.pageable)(B, D)
    n = 0
    for k, (s, e) in enumerate(windows):
        b = it.next_host(hb)
        assert b is not None and b.index == k
        idx = perm[s:e]
        np.testing.assert_array_equal(bits(b.observations), bits(oracle.gather_f32(obs_o, idx)))
        np.testing.assert_array_equal(b.actions, oracle.gather_u64(act_o, idx))
        np.testing.assert_array_equal(bits(b.values), bits(oracle.gather_f32(val_o, idx)))
        np.testing.assert_array_equal(bits(b.neglogps), bits(oracle.gather_f32(nlp_o, idx)))
        np.testing.assert_array_equal(bits(b.returns), bits(oracle.gather_f32(ret_o, idx)))
        want64, _, s64 = oracle.standardize_f64(oracle.gather_f32(adv_o, idx))
        if s64 > 1e-3:
            np.testing.assert_allclose(b.advantages, want64, rtol=0, atol=2e-6 * max(1.0, np.abs(want64).max()) + 1e-6)
        n += 1
    assert it.next_host(hb) is None and n == len(windows)
    # the device minibatches of the same pass hold the same bits as the host copies
    it2 = batcher.into_iter(seed=901)
    last = None
    for b in it2:
        last = b
    it3 = batcher.into_iter(seed=901)
    for _ in windows:
        hlast = it3.next_host()
    np.testing.assert_array_equal(bits(last.advantages.cpu().numpy()), bits(hlast.advantages))
    # too small a destination: EINVAL, nothing consumed
    small = xp.HostBatch.pageable(B - 1, D)
    it4 = batcher.into_iter(seed=3)
    with pytest.raises(AssertionError):
        it4.next_host(small)
    assert len(list(it4)) == len(windows)
    g.close()


def test_prefetch_is_discarded_when_it_does_not_match(xp, oracle):
    g, o, r = _both(xp, oracle, 6, 10, 2, 4)
    g.returns(r["last_dones"])
    batcher = xp.ExperienceBatcher(g, 7)
    batcher.prefetch_shuffle(1)
    it = batcher.into_iter(seed=2)                     # another seed: drawn on the caller's thread
    np.testing.assert_array_equal(it.perm(), oracle.shuffle(2, 60))
    it = batcher.into_iter(seed=1)                     # the stale prefetch was dropped, not reused
    np.testing.assert_array_equal(it.perm(), oracle.shuffle(1, 60))
    batcher.prefetch_shuffle(5)
    batcher.prefetch_shuffle(6)                        # restarts
    g.close()                                          # joins the worker


@pytest.mark.parametrize("algo_name", ["seq", "assoc"])
def test_flat_arrays_are_materialised_on_demand(xp, oracle, algo_name):
    """ppo_xp_returns(host_out=NULL) with records: returns / advantages live only in the records until
    someone asks.  Views, explained_variance (from records and from the flat arrays) and the forced
    gae_flat=1 kernel path must all agree bit for bit."""
    from ppo_amd import _lib
    algo = {"seq": 0, "assoc": 1 | _lib.gae_chunk(5)}[algo_name]
    T, N, D = 23, 130, 4
    g, o, r = _both(xp, oracle, T, N, D, 77)
    L, h = _lib.load(), g.handle
    ld = np.ascontiguousarray(r["last_dones"], np.uint8)
    _lib.check(L.ppo_xp_returns(h, ld.ctypes.data_as(C.c_void_p), N, algo, None))
    ev_rec = g.explained_variance()                    # moments straight from the records
    ret_view = g.returns(r["last_dones"], algo=algo).cpu().numpy()      # unpacked on demand
    adv_view = g.advantages().cpu().numpy()
    ev_flat = g.explained_variance()                   # moments from the flat arrays
    ret_host = g.returns(r["last_dones"], algo=algo, to_host=True)      # the scan stores them itself
    g.set_tunable("gae_flat", 1)
    _lib.check(L.ppo_xp_returns(h, ld.ctypes.data_as(C.c_void_p), N, algo, None))
    v = g._views()
    g.sync()
    import torch
    ret_forced = xp.device_tensor(v.returns, (T * N,), torch.float32, 0, g).cpu().numpy()
    np.testing.assert_array_equal(bits(ret_view), bits(ret_host))
    np.testing.assert_array_equal(bits(ret_view), bits(ret_forced))
    _, _, val_o, _ = o.training_views()
    np.testing.assert_array_equal(bits(adv_view), bits(oracle.advantages(ret_view, val_o)))
    assert np.float32(ev_rec).view(np.uint32) == np.float32(ev_flat).view(np.uint32)
    if algo_name == "seq":
        np.testing.assert_array_equal(bits(ret_view), bits(o.returns(r["last_dones"])))
    g.close()


def test_borrowed_device_permutation_may_be_a_temporary(xp, oracle):
    """ppo_xp_pass_begin_device borrows the array; the Python mirror must keep a temporary alive and
    ordered (the caller drops its reference at once and churns torch's allocator)."""
    import torch
    g, o, r = _both(xp, oracle, 64, 512, 16, 5)
    g.returns(r["last_dones"])
    M, B = 64 * 512, 4096
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1)
    perm_host = torch.randperm(M, device="cuda", generator=gen).int().cpu().numpy()
    gen.manual_seed(1)
    it = xp.ExperienceBatcher(g, B).into_iter(perm=torch.randperm(M, device="cuda", generator=gen).int())
    junk = [torch.full((M,), 7, dtype=torch.int32, device="cuda") for _ in range(8)]   # would recycle a freed perm
    obs_o = o.training_views()[0]
    for k, b in enumerate(it):
        np.testing.assert_array_equal(b.observations.cpu().numpy(), obs_o[perm_host[k * B:(k + 1) * B].astype(np.int64)])
    np.testing.assert_array_equal(it.perm(), perm_host.astype(np.uint32))
    del junk
    g.close()


def test_device_insert_is_ordered_against_torchs_stream(xp, oracle):
    """add_experience_device only enqueues on the handle's own stream: overwriting the source tensors
    right after the call (the next simulator step) must not change what was inserted."""
    import torch
    from ppo_amd import synth
    T, N, D = 12, 4096, 64
    r = synth.rollout(T, N, D, 9, 0.1)
    g, o = xp.ExperienceBuffer(N, D, T), oracle.OracleBuffer(N, D, T)
    dev = {k: torch.empty(v.shape[1:], dtype=torch.from_numpy(v[:1]).dtype, device="cuda") for k, v in r.items() if k != "last_dones"}
    dev["actions"] = torch.empty(N, dtype=torch.int64, device="cuda")
    dev["dones"] = torch.empty(N, dtype=torch.float32, device="cuda")
    for t in range(T):
        dev["obs"].copy_(torch.from_numpy(r["obs"][t]), non_blocking=True)
        dev["rewards"].copy_(torch.from_numpy(r["rewards"][t]), non_blocking=True)
        dev["actions"].copy_(torch.from_numpy(r["actions"][t].astype(np.int64)), non_blocking=True)
        dev["values"].copy_(torch.from_numpy(r["values"][t]), non_blocking=True)
        dev["dones"].copy_(torch.from_numpy(r["dones"][t].astype(np.float32)), non_blocking=True)
        dev["neglogps"].copy_(torch.from_numpy(r["neglogps"][t]), non_blocking=True)
        g.add_experience_device(dev["obs"], dev["rewards"], dev["actions"], dev["values"], dev["dones"], dev["neglogps"])
        for x in dev.values():
            x.fill_(-1)                                   # in-place overwrite on torch's stream, no sync
        o.add_experience(r["obs"][t], r["rewards"][t], r["actions"][t], r["values"][t], r["dones"][t], r["neglogps"][t])
    obs_g, act_g, val_g, nlp_g = (t.cpu().numpy() for t in g.training_views())
    obs_o, act_o, val_o, nlp_o = o.training_views()
    np.testing.assert_array_equal(bits(obs_g), bits(obs_o))
    np.testing.assert_array_equal(act_g.astype(np.uint64), act_o)
    np.testing.assert_array_equal(bits(val_g), bits(val_o))
    np.testing.assert_array_equal(bits(g.returns(r["last_dones"], to_host=True)), bits(o.returns(r["last_dones"])))
    g.close()


def test_rccl_exchange_on_the_handles_own_stream(xp, oracle):
    """ShardedBatcher with the nccl backend and NO use_torch_stream(): the all-reduce runs on torch's
    current stream, the sums and the standardise kernel on the handle's stream; the mirror brackets the
    collective with stream waits.  One rank (all a one-GPU box allows): output == single-GPU pass."""
    import torch
    import torch.distributed as dist
    from ppo_amd import sharded
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        g, o, r = _both(xp, oracle, 32, 2048, 8, 21)
        g.returns(r["last_dones"])
        B = 8192
        ref = [b.advantages.clone() for b in xp.ExperienceBatcher(g, B).into_iter(seed=4)]
        sb = sharded.ShardedBatcher(g, B)
        assert g.stream_ptr != torch.cuda.current_stream(0).cuda_stream
        for rep in range(3):
            it = sb.into_iter(perm=oracle.shuffle(4, 32 * 2048))
            got = [b.advantages.clone() for b in it]
            assert len(got) == len(ref)
            for a, b in zip(got, ref):
                assert torch.equal(a.view(torch.int32), b.view(torch.int32))
        g.close()
    finally:
        dist.destroy_process_group()


def test_slabs_do_not_move_between_a_short_and_a_full_pass(xp, oracle):
    """A pass over a partly filled ring followed by one over the full ring (same batch size) must not
    re-allocate the minibatch slabs: tensors of the earlier pass stay valid memory."""
    from ppo_amd import synth
    T, N, D, B = 8, 64, 4, 32
    r = synth.rollout(T, N, D, 2, 0.1)
    g = xp.ExperienceBuffer(N, D, T)
    for t in range(3):
        g.add_experience(r["obs"][t], r["rewards"][t], r["actions"][t], r["values"][t], r["dones"][t], r["neglogps"][t])
    g.returns(r["last_dones"])
    first = next(iter(xp.ExperienceBatcher(g, B).into_iter(seed=1)))
    p0 = first.observations.data_ptr()
    for t in range(3, T):
        g.add_experience(r["obs"][t], r["rewards"][t], r["actions"][t], r["values"][t], r["dones"][t], r["neglogps"][t])
    g.returns(r["last_dones"])
    again = next(iter(xp.ExperienceBatcher(g, B).into_iter(seed=1)))
    assert again.observations.data_ptr() == p0
    g.close()


def test_partitioned_scalar_path_in_the_experiments_build():
    """The partitioned scalar path lost to the direct record gather (profiles/r2/partitioned_scalar_path_negative.log)
    and is no longer part of libppo_xp.so: the product rejects its tunable.  Its bit-identity evidence is kept alive
    against libppo_xp_experiments.so (make -C ppo_amd/csrc experiments), loaded by a child process."""
    import subprocess
    import sys
    import ppo_amd
    here = os.path.dirname(os.path.abspath(__file__))
    g = ppo_amd.ExperienceBuffer(4, 4, 2)
    with pytest.raises(ppo_amd.PpoXpError):
        g.set_tunable("scalar_path", 1)
    g.close()
    lib = os.path.join(os.path.dirname(here), "ppo_amd", "libppo_xp_experiments.so")
    assert os.path.exists(lib), "build it with: make -C ppo_amd/csrc experiments"
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "experiments_partition.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       env=dict(os.environ, PPO_XP_LIB=lib), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert "3 passed" in p.stdout


@pytest.mark.parametrize("T,N,D,B,fused", [(16, 128, 64, 512, 1),      # fused gather, B divides M
                                           (9, 130, 64, 320, -1),      # fused gather with an overlapping tail window
                                           (12, 100, 6, 77, 0),        # split kernels, tail, odd row width
                                           (5, 64, 8, 320, 0)])        # one minibatch
def test_streaming_mode_matches_the_whole_pass(xp, oracle, T, N, D, B, fused):
    """ppo_xp_set_streaming: minibatch k is gathered by pass_next into one of two B-row slots.  Same kernels on a
    range of rows: every minibatch must hold the bits of the whole-pass mode, the slots must alternate, and the
    host hand-off, the statistics and the switch back must work."""
    import torch
    from ppo_amd import _lib
    g, o, r = _both(xp, oracle, T, N, D, 5 + T, big_actions=True)
    g.set_tunable("fused", fused)
    g.returns(r["last_dones"])
    M = T * N
    fields = ("observations", "actions", "values", "neglogps", "returns", "advantages")
    it = xp.ExperienceBatcher(g, B).into_iter(seed=9)
    whole = [[getattr(b, f).clone() for f in fields] for b in it]
    stats_whole = it.stats()
    g.set_streaming(True)
    it = xp.ExperienceBatcher(g, B).into_iter(seed=9)
    np.testing.assert_array_equal(it.perm(), oracle.shuffle(9, M))
    ptrs, n = [], 0
    for k, b in enumerate(it):
        ptrs.append(b.observations.data_ptr())
        for f, want in zip(fields, whole[k]):
            got = getattr(b, f)
            assert torch.equal(got.view(torch.int32) if got.dtype == torch.float32 else got,
                               want.view(torch.int32) if want.dtype == torch.float32 else want), (k, f)
        n += 1
    assert n == len(whole) == oracle.num_batches(M, B)
    assert all(ptrs[i] == ptrs[i % 2] for i in range(n)) and (n < 2 or ptrs[0] != ptrs[1])   # two alternating slots
    np.testing.assert_array_equal(bits(it.stats()), bits(stats_whole))
    # the host hand-off on top of the streamed pass
    it = xp.ExperienceBatcher(g, B).into_iter(seed=9)
    hb = xp.HostBatch.pageable(B, D)
    for k in range(n):
        b = it.next_host(hb)
        np.testing.assert_array_equal(bits(b.advantages), bits(whole[k][5].cpu().numpy()))
        np.testing.assert_array_equal(bits(b.observations), bits(whole[k][0].cpu().numpy()))
    assert it.next_host(hb) is None
    # an exchange cannot be combined with streaming, and switching back restores the whole-pass slabs
    with pytest.raises(xp.PpoXpError):
        g.set_exchange(2)
    g.set_streaming(False)
    again = [[getattr(b, f).clone() for f in fields] for b in xp.ExperienceBatcher(g, B).into_iter(seed=9)]
    assert all(torch.equal(a[5].view(torch.int32), w[5].view(torch.int32)) for a, w in zip(again, whole))
    g.set_exchange(2)
    with pytest.raises(xp.PpoXpError):
        g.set_streaming(True)
    g.close()


@pytest.mark.parametrize("N,D,cap,steps,offset", [(37, 3, 5, 12, 0),      # ragged: scalar elements, ring wraps twice
                                                  (100, 6, 7, 7, 0),      # row = 600 floats: 16-byte elements
                                                  (64, 8, 4, 9, 1),       # source rows start 4 bytes off a 16-byte boundary
                                                  (300, 64, 6, 6, 0)])    # several scalar workgroups, fused-gather shape
def test_device_insert_single_launch(xp, oracle, N, D, cap, steps, offset):
    """ppo_xp_add_device is ONE kernel per row (rows, scalar streams, the records' static halves and the done bits):
    ragged shapes, unaligned sources, 64-bit actions and ring wrap against the oracle's add_experience
    (buffer.rs:84-121), then a whole pass so that the records and the done bits are read back."""
    import torch
    from ppo_amd import synth
    from test_gpu_parity import check_pass
    r = synth.rollout(steps, N, D, 1234 + N, 0.2)
    r["actions"][:, ::5] = (1 << 40) + 3                      # does not fit the record's 32 bits: escape path
    g, o = xp.ExperienceBuffer(N, D, cap), oracle.OracleBuffer(N, D, cap)
    for t in range(steps):
        def dev(a, dt):
            flat = torch.empty(a.size + offset, dtype=dt, device="cuda")
            v = flat[offset:].view(a.shape)
            v.copy_(torch.from_numpy(np.ascontiguousarray(a).astype({torch.float32: np.float32, torch.int64: np.int64}[dt])))
            return v
        g.add_experience_device(dev(r["obs"][t], torch.float32), dev(r["rewards"][t], torch.float32),
                                dev(r["actions"][t], torch.int64), dev(r["values"][t], torch.float32),
                                dev(r["dones"][t], torch.float32), dev(r["neglogps"][t], torch.float32))
        o.add_experience(r["obs"][t], r["rewards"][t], r["actions"][t], r["values"][t], r["dones"][t], r["neglogps"][t])
    assert len(g) == len(o)
    M = len(o) * N
    check_pass(xp, oracle, g, o, r["last_dones"], max(1, M // 3), seed=5)
    g.close()


def test_streamed_pass_guards(xp, oracle):
    """Round-3 advisor items: a streamed pass knows minibatch k's statistics only once pass_next has gathered it
    (ESTATE before the pass is exhausted), and a tunable changed in the middle of a streamed pass ends that pass
    (its launch shapes - tiles, partials - were fixed at pass_begin)."""
    g, o, r = _both(xp, oracle, 8, 128, 64, 5)
    g.returns(r["last_dones"])
    g.set_streaming(True)
    it = xp.ExperienceBatcher(g, 256).into_iter(seed=3)
    next(it)
    with pytest.raises(xp.PpoXpError) as e:
        it.stats()
    assert e.value.code == -1
    rest = list(it)
    assert len(rest) == 3 and it.stats().shape == (4, 2)
    it = xp.ExperienceBatcher(g, 256).into_iter(seed=3)
    next(it)
    g.set_tunable("fused_block", 256)
    with pytest.raises(xp.PpoXpError) as e:
        next(it)
    assert e.value.code == -1                      # no pass any more: begin again
    g.set_tunable("fused_block", 512)
    assert len(list(xp.ExperienceBatcher(g, 256).into_iter(seed=3))) == 4
    g.close()
