#!/usr/bin/env python3
"""Host-row insert latency (ppo_xp_add, the reference's call pattern: one add_experience per env step, buffer.rs:84-121)
for small and large rows, per call, through bare ctypes calls.  Off the metric."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ppo_amd
from ppo_amd import _lib, synth
shapes = [("ref default", 32, 160, 16), ("C1", 128, 8, 4), ("1k envs", 64, 1024, 64), ("C2", 128, 4096, 64), ("TGT", 32, 65536, 64)]
L = _lib.load()
for name, T, N, D in shapes:
    buf = ppo_amd.ExperienceBuffer(N, D, T)
    r = synth.rollout(T, N, D, 1, 0.05)
    h = buf.handle
    rows = [[np.ascontiguousarray(r[k][t]) for k in ("obs", "rewards", "actions", "values", "dones", "neglogps")] for t in range(T)]
    for row in rows:
        row[2] = row[2].astype(np.uint64); row[4] = row[4].astype(np.uint8)
    ptrs = [[x.ctypes.data_as(C.c_void_p) for x in row] for row in rows]
    best = None
    for rep in range(5):
        buf.reset_counter()
        buf.sync(); t0 = time.perf_counter()
        for p in ptrs:
            _lib.check(L.ppo_xp_add(h, *p, N))
        t1 = time.perf_counter(); buf.sync(); t2 = time.perf_counter()
        cur = ((t1 - t0) / T * 1e6, (t2 - t0) / T * 1e6)
        best = cur if best is None or cur[1] < best[1] else best
    print(f"{name}: N={N} D={D} ({N * (4 * D + 21) / 1e3:.1f} KB per row): {best[0]:.1f} us per ppo_xp_add call, {best[1]:.1f} us per row until done")
    buf.close()
