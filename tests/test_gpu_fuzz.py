"""Seeded random sweep of the whole path against the oracle: buffer shapes (ragged N, odd D, partial
and wrapped rings), batch sizes (one row .. larger than the buffer), every GAE algorithm and scan
shape, both gather block sizes, the split and the fused gather, library-drawn / caller-supplied permutations.  Every case goes through
tests/test_gpu_parity.check_pass, i.e. the same bit-exact / stated-tolerance bars."""
import os

import numpy as np
import pytest

from test_gpu_parity import both, check_pass, xp  # noqa: F401  (xp is a fixture)

pytestmark = pytest.mark.gpu

D_CHOICES = [1, 2, 3, 4, 5, 7, 8, 12, 16, 20, 32, 33, 64, 100, 128, 256]


def _case(g):
    cap = int(g.integers(1, 41))
    steps = int(g.integers(1, 3 * cap + 1))
    N = int(g.choice([1, 2, 3, 63, 64, 65, 130, 200])) if g.random() < 0.5 else int(g.integers(1, 201))
    D = int(g.choice(D_CHOICES))
    M = min(steps, cap) * N
    r = g.random()
    B = 1 if r < 0.05 else M if r < 0.15 else M + 1 if r < 0.2 else int(g.integers(1, M + 1))
    if B == 1 and M > 3000:
        B = 7       # K = M one-row minibatches is covered elsewhere; keep the sweep quick
    if D in (32, 64, 128, 256) and M >= 64 and g.random() < 0.6:
        B = 64 * int(g.integers(1, M // 64 + 1))    # shapes the fused rows+scalars gather accepts (B % rows-per-workgroup == 0)
    return cap, steps, N, D, B


@pytest.mark.parametrize("chunk", range(16))
def test_random_configurations(xp, oracle, chunk):
    import ppo_amd
    g = np.random.default_rng(0xF00D + chunk + 1000 * int(os.environ.get("PPO_XP_FUZZ_BASE", "0")))   # other bases: extra sweeps by hand
    for i in range(25):
        cap, steps, N, D, B = _case(g)
        algo_kind = int(g.integers(0, 3))
        algo = [0, ppo_amd.GAE_NO_PACK, ppo_amd.GAE_ASSOC | ppo_amd.gae_chunk(int(g.integers(0, 9)))][algo_kind]
        buf, orc, r = both(xp, oracle, 0, N, D, cap, seed=int(g.integers(1 << 30)), p_done=float(g.choice([0.0, 0.05, 0.5])),
                           steps=steps, bulk=bool(g.integers(0, 2)))
        buf.set_tunable("gae_waves", int(g.choice([0, 1, 2, 4, 8, 16])))
        buf.set_tunable("gather_block", int(g.choice([64, 256])))
        buf.set_tunable("fold_tiles", int(g.choice([0, 256])))
        buf.set_tunable("fused", int(g.choice([-1, 0, 1])))
        buf.set_tunable("fused_block", int(g.choice([256, 512])))
        buf.set_tunable("bij_mode", int(g.choice([0, 1, 2])))
        buf.set_tunable("fused_iter", int(g.choice([0, 1, 2, 4])))
        buf.set_tunable("overlap", int(g.choice([0, 1])))
        M = len(orc) * N
        perm = None
        seed = int(g.integers(1 << 62))
        how = g.random()
        if how < 0.3:
            perm, seed = g.permutation(M).astype(np.uint32), None
        try:
            check_pass(xp, oracle, buf, orc, r["last_dones"], B, seed=seed, perm=perm, algo=algo,
                       exact_returns=algo_kind != 2, bijection=how > 0.65)
        except AssertionError as e:
            raise AssertionError(f"case {chunk}/{i}: cap={cap} steps={steps} N={N} D={D} B={B} algo={algo:#x}: {e}") from e
        finally:
            buf.close()
