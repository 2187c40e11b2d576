#!/usr/bin/env python3
"""Tight A/B of the sequential scan's waves per 64 envs (tunable gae_waves) in the bijection-pass context: whole-step wall
clock, variants interleaved, many rounds.  usage: ab_scan_waves.py CFG [rounds]"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ppo_amd
from ppo_amd import _lib, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "TGT"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 12
if "/" in cfg:
    base, div = cfg.split("/"); T, N, D, B = synth.CONFIGS[base]; N, B = N // int(div), B // int(div)
else:
    T, N, D, B = synth.CONFIGS[cfg]
dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream(dev))
buf = ppo_amd.ExperienceBuffer(N, D, T); buf.use_torch_stream()
tstep = max(1, (2 << 30) // (N * D * 4))
for tb in range(0, T, tstep):
    r = synth.rollout_device(min(tstep, T - tb), N, D, 1 + tb, dev)
    for t in range(r["rewards"].shape[0]):
        buf.add_experience_device(r["obs"][t], r["rewards"][t], r["actions"][t], r["values"][t], r["dones"][t], r["neglogps"][t])
    ld = r["last_dones"].cpu().numpy(); buf.sync(); del r
torch.cuda.empty_cache()
L, h = _lib.load(), buf.handle
ldp = ld.ctypes.data_as(C.c_void_p)
variants = [0, 2, 4, 8]
wall = {v: [] for v in variants}
for rnd in range(rounds):
    order = variants if rnd % 2 == 0 else variants[::-1]          # alternate the order: no variant always follows the same one
    for v in order:
        buf.set_tunable("gae_waves", v)
        for i in range(3):                                          # settle: the first steps after a switch see the other variant's caches
            _lib.check(L.ppo_xp_returns(h, ldp, N, 0, None)); _lib.check(L.ppo_xp_pass_begin_bijection(h, B, i))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(20):
            _lib.check(L.ppo_xp_returns(h, ldp, N, 0, None)); _lib.check(L.ppo_xp_pass_begin_bijection(h, B, 100 + i))
        torch.cuda.synchronize(); wall[v].append((time.perf_counter() - t0) / 20 * 1e6)
for v in variants:
    w = np.array(wall[v][1:])
    print(f"{cfg} gae_waves={v} ({'auto' if v == 0 else v}): step us median {np.median(w):.1f}  min {w.min():.1f}  max {w.max():.1f}", flush=True)
buf.close()
