"""Run only against libppo_xp_experiments.so (PPO_XP_LIB=...; see test_partitioned_scalar_path_in_the_experiments_build):
the partitioned scalar path of tools/experiments/kernels_partition.h, tunable scalar_path = 1."""
import numpy as np
import pytest

from test_gpu_round2 import _both, bits, xp  # noqa: F401  (xp is a fixture)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("T,N,D,B,dups", [(64, 4096, 8, 8192, False), (16, 4096, 4, 4096, True), (33, 4099, 2, 33 * 4099, False)])
def test_partitioned_scalar_path_matches_the_direct_gather(xp, oracle, T, N, D, B, dups):
    """tunable scalar_path = 1 (kernels_partition.h): binning by source and destination windows instead of one
    128-byte line fetch per record.  Outputs and tile partials must be the direct gather's, bit for bit — also
    for an index array with duplicates (bins of unequal size) and for 64-bit actions."""
    import torch
    g, o, r = _both(xp, oracle, T, N, D, 3 + T, big_actions=True)
    g.returns(r["last_dones"])
    M = T * N
    rng = np.random.default_rng(1)
    perm = rng.integers(0, M, size=M).astype(np.uint32) if dups else rng.permutation(M).astype(np.uint32)
    outs = []
    for path in (0, 1):
        g.set_tunable("fused", 0)
        g.set_tunable("scalar_path", path)
        it = xp.ExperienceBatcher(g, B).into_iter(perm=perm)
        bs = list(it)
        outs.append(([torch.cat([getattr(b, f).reshape(-1) for b in bs]).cpu().numpy() for f in
                      ("observations", "actions", "values", "neglogps", "returns", "advantages")], it.stats()))
    for a, b in zip(outs[0][0], outs[1][0]):
        np.testing.assert_array_equal(bits(a), bits(b))
    np.testing.assert_array_equal(bits(outs[0][1]), bits(outs[1][1]))
    obs_o, act_o, val_o, nlp_o = o.training_views()
    np.testing.assert_array_equal(outs[1][0][1].astype(np.uint64), oracle.gather_u64(act_o, perm))
    np.testing.assert_array_equal(bits(outs[1][0][2]), bits(oracle.gather_f32(val_o, perm)))
    g.close()
