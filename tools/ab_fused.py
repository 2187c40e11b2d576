#!/usr/bin/env python3
"""A/B of the fused gather (tunable fused=1) against the split path: outputs bit-compared, per-kernel times."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ppo_amd
from ppo_amd import _lib, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "TGT"
if "/" in cfg:                      # e.g. C5/8: the per-GPU shard of an 8-rank job
    base, div = cfg.split("/")
    T, N, D, B = synth.CONFIGS[base]
    N, B = N // int(div), B // int(div)
else:
    T, N, D, B = synth.CONFIGS[cfg]
M = T * N
dev = torch.device("cuda", 0)
buf = ppo_amd.ExperienceBuffer(N, D, T); buf.use_torch_stream()
r = synth.rollout_device(T, N, D, 1, dev)
for t in range(T):
    buf.add_experience_device(r["obs"][t], r["rewards"][t], r["actions"][t], r["values"][t], r["dones"][t], r["neglogps"][t])
ld = r["last_dones"].cpu().numpy(); del r
L, h = _lib.load(), buf.handle
ldp = ld.ctypes.data_as(C.c_void_p)
_lib.check(L.ppo_xp_returns(h, ldp, N, 0, None))
_lib.check(L.ppo_xp_pass_begin(h, B, None, 1))
perm = ppo_amd.device_tensor(L.ppo_xp_pass_perm_device(h), (M,), torch.uint32, 0, buf).clone()
buf.sync()
outs = {}
for fused in (0, 1, 2):
    buf.set_tunable("fused", 1 if fused else 0)
    buf.set_tunable("fused_block", 256 if fused == 1 else 512)
    _lib.check(L.ppo_xp_returns(h, ldp, N, 0, None))
    it = ppo_amd.ExperienceBatcher(buf, B).into_iter(perm=perm)
    bs = list(it)
    outs[fused] = [torch.cat([getattr(b, f).reshape(-1).view(torch.int32) if getattr(b, f).dtype != torch.int64 else getattr(b, f).reshape(-1).to(torch.int32) for b in bs]).clone()
                   for f in ("observations", "actions", "values", "neglogps", "returns", "advantages")]
for which in (1, 2):
  for a, b, f in zip(outs[0], outs[which], ("observations", "actions", "values", "neglogps", "returns", "advantages")):
    print(which, f, "bit-equal" if torch.equal(a, b) else f"DIFFER (max abs diff as f32 {(a.view(torch.float32)-b.view(torch.float32)).abs().max().item():.3g})")
del outs
res = {0: [], 1: []}
wall = {0: [], 1: []}
import time
variants = {0: dict(fused=0), 1: dict(fused=1, fused_block=256), 4: dict(fused=1, fused_block=512), 5: dict(fused=-1, fused_block=512)}
res = {v: [] for v in variants}; wall = {v: [] for v in variants}
for rnd in range(8):
    for fused, tun in variants.items():
        for k_, v_ in tun.items():
            buf.set_tunable(k_, v_)
        buf.prof_enable(True)
        _lib.check(L.ppo_xp_returns(h, ldp, N, 0, None))
        _lib.check(L.ppo_xp_pass_begin_device(h, B, C.c_void_p(perm.data_ptr())))
        p = buf.prof_collect()
        buf.prof_enable(False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            _lib.check(L.ppo_xp_returns(h, ldp, N, 0, None))
            _lib.check(L.ppo_xp_pass_begin_device(h, B, C.c_void_p(perm.data_ptr())))
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        if rnd:
            res[fused].append({k: ms / max(n, 1) * 1e3 for k, (ms, n) in p.items() if n}); wall[fused].append(dt * 1e6)
for fused in variants:
    ks = sorted(res[fused][0])
    print(variants[fused], "step us median/min: %.1f/%.1f" % (np.median(wall[fused]), min(wall[fused])), "|",
          " ".join(f"{k}={np.median([x[k] for x in res[fused]]):.1f}" for k in ks))
buf.close()
