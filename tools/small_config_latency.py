#!/usr/bin/env python3
"""Latency of one whole iteration at small sizes (the reference's default: 160 envs x 32 steps, batch 1280): T inserts
from host rows, returns, one shuffled pass handed out as host batches — the library against the CPU oracle.  Shows
where the GPU path stops paying: a context figure, off the metric."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ppo_amd
from ppo_amd import _lib, synth
from oracle import oracle as O
L = _lib.load()
for name, T, N, D, B in [("ref default", 32, 160, 16, 1280), ("C1", 128, 8, 4, 256), ("1k envs", 32, 1024, 64, 8192), ("C2", 128, 4096, 64, 16384)]:
    r = synth.rollout(T, N, D, 3, 0.05)
    rows = [[np.ascontiguousarray(r[k][t]) for k in ("obs", "rewards", "actions", "values", "dones", "neglogps")] for t in range(T)]
    for row in rows:
        row[2] = row[2].astype(np.uint64); row[4] = row[4].astype(np.uint8)
    buf = ppo_amd.ExperienceBuffer(N, D, T)
    ob = O.OracleBuffer(N, D, T)
    M = T * N
    K = (M + B - 1) // B
    def gpu():
        buf.reset_counter()
        for row in rows:
            buf.add_experience(*row)
        buf.returns(r["last_dones"], to_host=True)
        it = ppo_amd.ExperienceBatcher(buf, B).into_iter(seed=7)
        n = 0
        while it.next_host() is not None:
            n += 1
        return n
    def cpu():
        ob.reset_counter()
        for row in rows:
            ob.add_experience(*row)
        ret = ob.returns(r["last_dones"])
        obs, act, val, nlp = ob.training_views()
        adv = O.advantages(ret, val)
        perm = O.shuffle(7, M)
        n = 0
        for s, e in O.batch_windows(M, B):
            idx = perm[s:e]
            _ = (obs[idx], act[idx], val[idx], nlp[idx], ret[idx], O.standardize(adv[idx])[0])
            n += 1
        return n
    res = {}
    for f in (gpu, cpu):
        f(); ts = []
        for _ in range(7):
            t0 = time.perf_counter(); n = f(); ts.append(time.perf_counter() - t0)
        res[f.__name__] = np.median(ts) * 1e6
    print(f"{name}: T={T} N={N} D={D} B={B} (K={K}): whole iteration through the Python mirror {res['gpu']:.0f} us, "
          f"CPU oracle through its Python wrapper {res['cpu']:.0f} us")
    buf.close()
