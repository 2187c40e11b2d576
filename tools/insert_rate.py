#!/usr/bin/env python3
"""PCIe-inclusive insert rate (outside the metric): ppo_xp_add_many of a TGT-shaped
rollout from ordinary host memory vs from page-locked memory (ppo_amd.pinned_empty)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ppo_amd  # noqa: E402

T, N, D = 128, 65536, 64
if len(sys.argv) > 1:
    N = int(sys.argv[1])
buf = ppo_amd.ExperienceBuffer(N, D, T)
shapes = dict(obs=((T, N, D), np.float32), rewards=((T, N), np.float32), actions=((T, N), np.uint64),
              values=((T, N), np.float32), dones=((T, N), np.uint8), neglogps=((T, N), np.float32))
nbytes = sum(int(np.prod(s)) * np.dtype(d).itemsize for s, d in shapes.values())
for kind, alloc in (("pageable", lambda s, d: np.empty(s, d)), ("pinned", ppo_amd.pinned_empty)):
    a = {k: alloc(s, d) for k, (s, d) in shapes.items()}
    for v in a.values():
        v[...] = 1                      # touch every page
    best = 1e9
    for _ in range(3):
        buf.reset_counter()
        t0 = time.perf_counter()
        buf.add_many(a["obs"], a["rewards"], a["actions"], a["values"], a["dones"], a["neglogps"])
        buf.sync()
        best = min(best, time.perf_counter() - t0)
    print(f"{kind:9s} add_many of {nbytes / 1e9:.2f} GB: {best * 1e3:8.1f} ms = {nbytes / best / 1e9:6.1f} GB/s", flush=True)
    # step by step, the reference's call pattern (one add_experience per env step)
    buf.reset_counter()
    t0 = time.perf_counter()
    for t in range(T):
        buf.add_experience(a["obs"][t], a["rewards"][t], a["actions"][t], a["values"][t], a["dones"][t], a["neglogps"][t])
    buf.sync()
    dt = time.perf_counter() - t0
    print(f"{kind:9s} {T} x add_experience:        {dt * 1e3:8.1f} ms = {nbytes / dt / 1e9:6.1f} GB/s", flush=True)
    del a
buf.close()
