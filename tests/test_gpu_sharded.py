"""Env-sharded path on the GPU: (1) on one rank the three exchange modes (none / external
all-reduce / RCCL inside the library on a 1-rank communicator) give bit-identical outputs;
(2) two processes sharing the one GPU, each owning half the envs, exchange the sums over gloo and
reproduce the oracle's standardisation of the concatenated global minibatch."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fill(buf, r, sl=slice(None)):
    c = np.ascontiguousarray
    buf.add_many(c(r["obs"][:, sl]), c(r["rewards"][:, sl]), c(r["actions"][:, sl]), c(r["values"][:, sl]),
                 c(r["dones"][:, sl]), c(r["neglogps"][:, sl]))


@pytest.mark.parametrize("T,N,D,B,overlap", [(24, 96, 8, 500, 0),     # scalar_gather_kernel + obs_gather_kernel
                                             (24, 96, 8, 500, 1),     # the same with the rows on the side stream
                                             (16, 128, 64, 512, 0)])  # fused_gather_kernel (32 rows per workgroup)
def test_exchange_modes_agree_on_one_rank(T, N, D, B, overlap):
    import torch
    import ppo_amd
    from ppo_amd import _lib, synth
    r = synth.rollout(T, N, D, 77, 0.1)
    outs = []
    L = _lib.load()
    for mode in (0, 2, 1):
        buf = ppo_amd.ExperienceBuffer(N, D, T)
        if mode:
            buf.set_tunable("overlap", overlap)
        _fill(buf, r)
        buf.returns(r["last_dones"])
        comm = C.c_void_p()
        if mode == 1:
            uid = (C.c_char * 128)()
            _lib.check(L.ppo_xp_rccl_unique_id(uid))
            _lib.check(L.ppo_xp_rccl_comm_init(C.byref(comm), uid, 0, 1))
            assert L.ppo_xp_attach_comm(buf.handle, comm, 3, 2) == _lib.EINVAL   # rank outside [0, world)
            _lib.check(L.ppo_xp_attach_comm(buf.handle, comm, 0, 1))            # world 1: detaches (mode 0)
            buf.set_exchange(1, comm.value)
        elif mode == 2:
            buf.set_exchange(2)
        it = ppo_amd.ExperienceBatcher(buf, B).into_iter(seed=5)
        assert L.ppo_xp_pass_gather_kind(buf.handle) == (1 if D == 64 else 0)
        if mode == 2:
            with pytest.raises(ppo_amd.PpoXpError):      # statistics not exchanged yet
                next(it)
            K = L.ppo_xp_pass_num_batches(buf.handle)
            sums = buf.pass_sums(K)
            buf.sync()
            assert torch.all(sums[:, 0] == B)
            buf.pass_finish()
        batches = list(it)
        outs.append([np.concatenate([getattr(b, f).cpu().numpy().reshape(len(b.values), -1) for b in batches])
                     for f in ("observations", "actions", "values", "neglogps", "returns", "advantages")])
        buf.close()
        if mode == 1:
            _lib.check(L.ppo_xp_rccl_comm_destroy(comm))
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert np.array_equal(a.view(np.uint8), b.view(np.uint8))


def _rank(rank, world, port, T, N, D, B, seed, q, bijection=False):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        import ppo_amd
        from oracle import oracle as O
        from ppo_amd import sharded, synth
        full = synth.rollout(T, N, D, seed, 0.1)
        sl = sharded.env_slice(N, rank, world)
        n = sl.stop - sl.start
        buf = ppo_amd.ExperienceBuffer(n, D, T, device=0)
        _fill(buf, full, sl)
        ret = buf.returns(full["last_dones"][sl], to_host=True)
        ret_full = O.gae_rows(full["rewards"], full["values"], full["dones"].astype(np.float32), T - 1,
                              full["last_dones"], 0.99, 0.95).reshape(T, N)
        assert np.array_equal(ret.view(np.uint32), np.ascontiguousarray(ret_full[:, sl]).reshape(-1).view(np.uint32))
        sb = sharded.ShardedBatcher(buf, B, seed=seed)
        it = sb.into_iter(seed=seed, bijection=bijection)
        perm = it.perm()
        assert np.array_equal(perm, (O.bijection_perm if bijection else O.shuffle)(sharded.rank_seed(seed, rank), T * n))
        batches = list(it)
        Bl = sb.local_batch
        adv_loc = O.advantages(ret, np.ascontiguousarray(full["values"][:, sl]).reshape(-1))
        windows = O.batch_windows(T * n, Bl)
        assert len(batches) == len(windows)
        obs_loc = np.ascontiguousarray(full["obs"][:, sl]).reshape(T * n, D)
        raw = torch.from_numpy(np.stack([adv_loc[perm[s:e]] for s, e in windows]))
        allm = [torch.empty_like(raw) for _ in range(world)]
        dist.all_gather(allm, raw)
        worst = 0.0
        for k, ((s, e), b) in enumerate(zip(windows, batches)):
            assert np.array_equal(b.observations.cpu().numpy(), obs_loc[perm[s:e]])       # rows bit-exact
            cat = np.concatenate([a[k].numpy() for a in allm])
            want64, _, _ = O.standardize_f64(cat)
            want32, _, _ = O.standardize(cat)
            got = b.advantages.cpu().numpy()
            off = rank * Bl
            worst = max(worst, float(np.abs(got - want64[off:off + Bl]).max()))
            assert np.abs(got - want32[off:off + Bl]).max() < 2e-5
        # explained_variance over all ranks' transitions (training.rs:88-99) == the oracle's on the full arrays
        ev = sharded.explained_variance(buf)
        ev64 = O.explained_variance_f64(full["values"].reshape(-1), ret_full.reshape(-1))
        assert abs(ev - ev64) < 2e-6 * max(1.0, abs(ev64)), (ev, ev64)
        # reward normaliser (runner.rs:114-127): batch moments over ALL envs, same running statistics on every rank
        rn = sharded.ShardedRewardNormalizer(n)
        orn = O.OracleRewardNormalizer(N)
        for t in range(6):
            want = orn.normalized_rewards(full["rewards"][t], full["dones"][t])
            got = rn.normalized_rewards(torch.from_numpy(np.ascontiguousarray(full["rewards"][t, sl])).cuda(),
                                        torch.from_numpy(np.ascontiguousarray(full["dones"][t, sl]).astype(np.float32)).cuda())
            assert np.abs(got.cpu().numpy() - want[sl]).max() < 2e-5
        m, v, c, ret_state = rn.state()
        om, ov, oc, oret = orn.state()
        assert abs(m - om) < 2e-5 and abs(v - ov) < 2e-5 * max(1.0, ov) and c == oc
        assert np.array_equal(ret_state.view(np.uint32), np.ascontiguousarray(oret[sl]).view(np.uint32))
        rn.close()
        buf.close()
        q.put((rank, worst))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("shape,bijection", [((20, 64, 16, 256), False), ((16, 256, 64, 1024), True)])
def test_two_ranks_on_one_gpu_over_gloo(shape, bijection):
    """Second case: every rank's permutation is the keyed bijection of its rank seed, evaluated inside the fused gather."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, 2, port) + shape + (4242, q, bijection)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(150)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(2))
    assert [r[0] for r in res] == [0, 1] and max(r[1] for r in res) < 2e-6


def test_sharded_halves_equal_whole_on_one_rank():
    """begin / (no-op all-reduce) / finish of the reward normaliser and sums / from_sums of
    explained_variance are bit-identical to the one-call entry points."""
    import torch
    import ppo_amd
    from ppo_amd import _lib, synth
    N = 700
    r = synth.rollout(8, N, 2, 5, 0.1)
    a, b = ppo_amd.RewardNormalizer(N), ppo_amd.RewardNormalizer(N)
    for t in range(8):
        rw = torch.from_numpy(r["rewards"][t]).cuda()
        dn = torch.from_numpy(r["dones"][t].astype(np.float32)).cuda()
        x = a.normalized_rewards(rw, dn)
        y = b.normalized_rewards_sharded(rw, dn, lambda sums: None)
        assert torch.equal(x.view(torch.int32), y.view(torch.int32))
    assert a.state()[:3] == b.state()[:3]
    L = _lib.load()
    with pytest.raises(ppo_amd.PpoXpError):      # finish without begin
        _lib.check(L.ppo_xp_rnorm_step_finish_device(b._h, C.c_void_p(rw.data_ptr()), C.c_void_p(dn.data_ptr()),
                                                     C.c_void_p(x.data_ptr()), N, None))
    a.close(); b.close()
    buf = ppo_amd.ExperienceBuffer(N, 2, 8)
    _fill(buf, r)
    buf.returns(r["last_dones"])
    s5 = buf.explained_variance_sums()
    assert s5[0] == 8 * N
    out = C.c_float()
    _lib.check(L.ppo_xp_explained_variance_from_sums(s5.ctypes.data_as(C.POINTER(C.c_double)), C.byref(out)))
    assert np.float32(out.value).view(np.uint32) == np.float32(buf.explained_variance()).view(np.uint32)
    buf.close()
