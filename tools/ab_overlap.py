#!/usr/bin/env python3
"""A/B of the split path with the observation rows on a side stream (tunable overlap=1) against the split path in
one stream and the library's own choice: outputs bit-compared, whole-step wall time, interleaved in one process."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ppo_amd
from ppo_amd import _lib, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
if "/" in cfg:                      # e.g. C5/8: the per-GPU shard of an 8-rank job
    base, div = cfg.split("/")
    T, N, D, B = synth.CONFIGS[base]
    N, B = N // int(div), B // int(div)
else:
    T, N, D, B = synth.CONFIGS[cfg]
M = T * N
reps = 50 if M < (1 << 22) else 8
dev = torch.device("cuda", 0)
buf = ppo_amd.ExperienceBuffer(N, D, T); buf.use_torch_stream()
slab = max(1, min(T, (1 << 31) // max(N * D * 4, 1)))
for t0 in range(0, T, slab):
    r = synth.rollout_device(min(slab, T - t0), N, D, 1 + t0, dev)
    for t in range(r["obs"].shape[0]):
        buf.add_experience_device(r["obs"][t], r["rewards"][t], r["actions"][t], r["values"][t], r["dones"][t], r["neglogps"][t])
    ld = r["last_dones"].cpu().numpy()
    torch.cuda.synchronize(); del r
L, h = _lib.load(), buf.handle
ldp = ld.ctypes.data_as(C.c_void_p)
perm = torch.randperm(M, device=dev).to(torch.int32)
variants = {"auto": dict(fused=-1, overlap=0), "split": dict(fused=0, overlap=0), "split+side": dict(fused=0, overlap=1)}
if M <= (1 << 24):
    sig = {}
    for name, tun in variants.items():
        for k_, v_ in tun.items():
            buf.set_tunable(k_, v_)
        _lib.check(L.ppo_xp_returns(h, ldp, N, 0, None))
        bs = list(ppo_amd.ExperienceBatcher(buf, B).into_iter(perm=perm))
        sig[name] = [torch.cat([getattr(b, f).reshape(-1).view(torch.int32) if getattr(b, f).dtype != torch.int64 else getattr(b, f).reshape(-1).to(torch.int32) for b in bs]).clone()
                     for f in ("observations", "actions", "values", "neglogps", "returns", "advantages")]
    for name in ("split", "split+side"):
        print(name, "vs auto:", "bit-equal" if all(torch.equal(a, b) for a, b in zip(sig["auto"], sig[name])) else "DIFFER")
    del sig, bs
wall = {v: [] for v in variants}
for rnd in range(7):
    for name, tun in variants.items():
        for k_, v_ in tun.items():
            buf.set_tunable(k_, v_)
        for _ in range(2):
            _lib.check(L.ppo_xp_returns(h, ldp, N, 0, None))
            _lib.check(L.ppo_xp_pass_begin_device(h, B, C.c_void_p(perm.data_ptr())))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            _lib.check(L.ppo_xp_returns(h, ldp, N, 0, None))
            _lib.check(L.ppo_xp_pass_begin_device(h, B, C.c_void_p(perm.data_ptr())))
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        if rnd:
            wall[name].append(dt * 1e6)
for name in variants:
    print(cfg, name, "step us median/min: %.1f/%.1f" % (np.median(wall[name]), min(wall[name])))
buf.close()
