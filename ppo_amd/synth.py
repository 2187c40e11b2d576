"""Synthetic rollouts of SURVEY.md 8(d) / BASELINE.md: obs ~ U(-1,1), rewards, values ~ N(0,1),
neglogps ~ |N(0,1)|, actions ~ U{0..17}, dones, last_dones ~ Bernoulli(0.02); numpy PCG64(seed).
Used by bench.py and the tests; not part of the compute path."""
from __future__ import annotations

import numpy as np

CONFIGS = {  # name: (T, N, D, B)   seed = 0x5EED + index
    "C1": (128, 8, 4, 256),
    "C2": (128, 4096, 64, 16384),
    "C3": (2048, 16384, 256, 1048576),
    "C4": (16384, 1024, 1, 0),
    "C5": (128, 1048576, 64, 4194304),
    "TGT": (128, 65536, 64, 262144),
}
SEEDS = {name: 0x5EED + i + 1 for i, name in enumerate(CONFIGS)}


def rollout(T, N, D, seed, p_done=0.02):
    """dict of host arrays shaped [T, N, ...] plus last_dones [N]."""
    g = np.random.Generator(np.random.PCG64(seed))
    return dict(
        obs=g.uniform(-1.0, 1.0, size=(T, N, D)).astype(np.float32),
        rewards=g.standard_normal((T, N), dtype=np.float32),
        values=g.standard_normal((T, N), dtype=np.float32),
        neglogps=np.abs(g.standard_normal((T, N), dtype=np.float32)),
        actions=g.integers(0, 18, size=(T, N), dtype=np.uint64),
        dones=(g.random((T, N)) < p_done).astype(np.uint8),
        last_dones=(g.random(N) < p_done).astype(np.uint8),
    )


def rollout_device(T, N, D, seed, device, p_done=0.02):
    """Same distributions drawn on the GPU with torch (for sizes too large to stage on the host).
    Not bit-identical to rollout(); used only with size-independent property checks and bench."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    f = dict(device=device, generator=g)
    return dict(
        obs=torch.rand((T, N, D), dtype=torch.float32, **f) * 2 - 1,
        rewards=torch.randn((T, N), dtype=torch.float32, **f),
        values=torch.randn((T, N), dtype=torch.float32, **f),
        neglogps=torch.randn((T, N), dtype=torch.float32, **f).abs(),
        actions=torch.randint(0, 18, (T, N), dtype=torch.int64, **f),
        dones=(torch.rand((T, N), **f) < p_done).to(torch.float32),
        last_dones=(torch.rand((N,), **f) < p_done).to(torch.uint8),
    )
