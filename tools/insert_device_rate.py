#!/usr/bin/env python3
"""Device-to-device insert rate (ppo_xp_add_device, the f4 row of SURVEY.md 8): T rows from a resident rollout, per-row
wall time through the Python mirror and through bare ctypes calls.  Off the metric."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ppo_amd
from ppo_amd import _lib, synth
for cfg in sys.argv[1:] or ["TGT", "C2"]:
    T, N, D, B = synth.CONFIGS[cfg]
    dev = torch.device("cuda", 0)
    buf = ppo_amd.ExperienceBuffer(N, D, T); buf.use_torch_stream()
    r = synth.rollout_device(T, N, D, 1, dev)
    rows = [(r["obs"][t], r["rewards"][t], r["actions"][t], r["values"][t], r["dones"][t], r["neglogps"][t]) for t in range(T)]
    L, h = _lib.load(), buf.handle
    res = {}
    for name in ("mirror", "ctypes"):
        ts = []
        for rep in range(4):
            buf.reset_counter()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            if name == "mirror":
                for row in rows:
                    buf.add_experience_device(*row)
            else:
                for row in rows:
                    _lib.check(L.ppo_xp_add_device(h, *(C.c_void_p(x.data_ptr()) for x in row), N))
            t1 = time.perf_counter()
            torch.cuda.synchronize(); t2 = time.perf_counter()
            ts.append(((t1 - t0) / T * 1e6, (t2 - t0) / T * 1e6))
        res[name] = min(ts[1:], key=lambda x: x[1])
    bytes_row = N * (D * 4 + 24)
    print(f"{cfg}: N={N} D={D}: per inserted row, us (host enqueue / until done on the GPU): " +
          " | ".join(f"{k} {v[0]:.1f} / {v[1]:.1f}" for k, v in res.items()) +
          f" ; {bytes_row / 1e6:.1f} MB per row -> {2 * bytes_row / (res['ctypes'][1] * 1e-6) / 1e9:.0f} GB/s read+write")
    buf.close(); del r, rows
