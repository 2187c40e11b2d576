#!/usr/bin/env python3
"""Per-step GPU time of the TGT step (returns + bijection pass) over the first N steps after set-up, from HIP events around
every step: how long does the box take to reach its steady state?  usage: step_curve.py [steps] [spinup_ms]"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ppo_amd
from ppo_amd import _lib, synth
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
spin = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
T, N, D, B = synth.CONFIGS["TGT"]
dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream(dev))
buf = ppo_amd.ExperienceBuffer(N, D, T); buf.use_torch_stream()
r = synth.rollout_device(T, N, D, 1, dev)
for t in range(T):
    buf.add_experience_device(r["obs"][t], r["rewards"][t], r["actions"][t], r["values"][t], r["dones"][t], r["neglogps"][t])
ld = r["last_dones"].cpu().numpy(); buf.sync(); del r
torch.cuda.empty_cache()
L, h = _lib.load(), buf.handle
ldp = ld.ctypes.data_as(C.c_void_p)
if spin > 0:
    a = torch.empty(1 << 30, dtype=torch.uint8, device=dev); b = torch.empty_like(a)
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < spin:
        b.copy_(a); torch.cuda.synchronize()
    del a, b
ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
ev[0].record()
for i in range(steps):
    _lib.check(L.ppo_xp_returns(h, ldp, N, 0, None)); _lib.check(L.ppo_xp_pass_begin_bijection(h, B, 100 + i))
    ev[i + 1].record()
torch.cuda.synchronize()
us = np.array([ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(steps)])
print(f"spin-up {spin:.0f} ms; per-step us: first 5 {np.round(us[:5], 1)}; mean of steps 5-24 {us[5:25].mean():.1f}; 25-49 {us[25:50].mean():.1f}; "
      f"50-99 {us[50:100].mean():.1f}; 100-199 {us[100:200].mean():.1f}; 200+ {us[200:].mean():.1f}", flush=True)
buf.close()
