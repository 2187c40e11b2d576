"""Env-sharded path on CPU: world_size 2 over gloo.  The compute of each rank is the CPU oracle
(this is a test; the product has no CPU path) and what is exercised is the host logic of
ppo_amd/sharded.py: env slicing, local batch size, per-rank permutation seeds, the (count, sum,
sum of squares) exchange through ONE all-reduce per pass and the statistics formula, against the
parity definition of SURVEY.md 8(e): standardised advantages equal the oracle's standardisation
of the CONCATENATED global minibatch; GAE of an env slice equals the slice of the full GAE."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, T, N, D, B, seed, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import oracle as O
    from ppo_amd import sharded, synth
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        full = synth.rollout(T, N, D, seed, 0.1)            # every rank draws the same global rollout
        sl = sharded.env_slice(N, rank, world)
        n = sl.stop - sl.start
        Bl = sharded.local_batch_size(B, world)
        # local shard = env slice; GAE on the slice must equal the slice of the global GAE (bit-exact)
        ret_full = O.gae_rows(full["rewards"], full["values"], full["dones"].astype(np.float32), T - 1,
                              full["last_dones"], 0.99, 0.95).reshape(T, N)
        ret_loc = O.gae_rows(full["rewards"][:, sl], full["values"][:, sl], full["dones"][:, sl].astype(np.float32),
                             T - 1, full["last_dones"][sl], 0.99, 0.95).reshape(T, n)
        assert np.array_equal(ret_loc.view(np.uint32), ret_full[:, sl].view(np.uint32))
        adv_loc = O.advantages(ret_loc.reshape(-1), np.ascontiguousarray(full["values"][:, sl]).reshape(-1))
        M = T * n
        perm = O.shuffle(sharded.rank_seed(seed, rank), M)
        windows = O.batch_windows(M, Bl)
        K = len(windows)
        sums = np.stack([sharded.local_sums(adv_loc[perm[s:e]]) for s, e in windows])   # [K, 3]
        t = torch.from_numpy(sums.copy())
        dist.all_reduce(t)                                   # the ONE exchange of the pass
        g = t.numpy()
        assert np.all(g[:, 0] == Bl * world)
        # gather every rank's local minibatches to build the concatenated global minibatch
        mine = torch.from_numpy(np.stack([adv_loc[perm[s:e]] for s, e in windows]))     # [K, Bl]
        allm = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allm, mine)
        worst = 0.0
        for k in range(K):
            mean, std = sharded.stats_from_sums(*g[k])
            got = sharded.standardize_with(mine[k].numpy(), mean, std)
            cat = np.concatenate([a[k].numpy() for a in allm])
            want, m32, s32 = O.standardize(cat)
            want64, m64, s64 = O.standardize_f64(cat)
            off = rank * Bl
            worst = max(worst, float(np.abs(got - want64[off:off + Bl]).max()))
            assert np.abs(got - want[off:off + Bl]).max() < 2e-5
            assert abs(mean - m64) < 1e-6 and abs(std - s64) < 1e-6 * max(1.0, s64)
        q.put((rank, K, worst))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("T,N,D,B", [(16, 8, 3, 32), (33, 20, 4, 100)])
def test_sharded_world2_gloo(T, N, D, B):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, T, N, D, B, 0x5EED + T, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(2))
    assert [r[0] for r in res] == [0, 1]
    assert res[0][1] == res[1][1] > 0
    assert max(r[2] for r in res) < 2e-6          # vs the fp64 standardisation of the global minibatch


def test_sharding_helpers():
    from ppo_amd import sharded
    assert sharded.env_slice(16, 1, 4) == slice(4, 8)
    assert sharded.local_batch_size(4194304, 8) == 524288
    with pytest.raises(AssertionError):
        sharded.env_slice(10, 0, 4)
    with pytest.raises(AssertionError):
        sharded.local_batch_size(10, 4)
    assert sharded.rank_seed(1, 0) != sharded.rank_seed(1, 1)
    m, s = sharded.stats_from_sums(1.0, 3.0, 9.0)
    assert m == 3.0 and np.isnan(s)                # n = 1: 0 * inf = NaN, as the reference
    m, s = sharded.stats_from_sums(5.0, 10.0, 30.0)
    assert m == 2.0 and abs(s - np.sqrt(2.5)) < 1e-6
