#!/usr/bin/env python3
"""A/B of the keyed-bijection pass against the array pass: per-kernel times and step wall clock.
usage: ab_bijection.py [CFG] [rounds]"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ppo_amd
from ppo_amd import _lib, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "TGT"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
if "/" in cfg:
    base, div = cfg.split("/")
    T, N, D, B = synth.CONFIGS[base]
    N, B = N // int(div), B // int(div)
else:
    T, N, D, B = synth.CONFIGS[cfg]
M = T * N
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
_lib.check(L.ppo_xp_returns(h, ldp, N, 0, None))
_lib.check(L.ppo_xp_pass_begin_bijection(h, B, 1))
perm = ppo_amd.device_tensor(L.ppo_xp_pass_perm_device(h), (M,), torch.uint32, 0, buf).clone()
buf.sync()
print(cfg, "gather kind", L.ppo_xp_pass_gather_kind(h), flush=True)

def begin_array(i): _lib.check(L.ppo_xp_pass_begin_device(h, B, C.c_void_p(perm.data_ptr())))
def begin_bij(i): _lib.check(L.ppo_xp_pass_begin_bijection(h, B, 1000 + i))
variants = {"array": (begin_array, {}), "bij lds": (begin_bij, {"bij_mode": 1, "fused_iter": 1}), "bij auto": (begin_bij, {}), "bij per-thread": (begin_bij, {"bij_mode": 2}),
            "bij written out": (begin_bij, {"bij_mode": 0})}
for a in sys.argv[3:]:
    name, rest = a.split(":", 1)
    variants[name] = (begin_bij if name.startswith("bij") else begin_array, {k: int(v) for k, v in (x.split("=") for x in rest.split(","))})
res = {v: [] for v in variants}; wall = {v: [] for v in variants}
defaults = {}
for rnd in range(rounds):
    for name, (begin, tun) in variants.items():
        for k_, v_ in tun.items():
            buf.set_tunable(k_, v_)
        buf.prof_enable(True)
        _lib.check(L.ppo_xp_returns(h, ldp, N, 0, None)); begin(0)
        p = buf.prof_collect()
        buf.prof_enable(False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(10):
            _lib.check(L.ppo_xp_returns(h, ldp, N, 0, None)); begin(i)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        if rnd:
            res[name].append({k: ms / max(n, 1) * 1e3 for k, (ms, n) in p.items() if n}); wall[name].append(dt * 1e6)
        for k_ in tun:
            buf.set_tunable(k_, {"bij_mode": 1, "fused": -1, "fused_block": 512, "fused_iter": 0, "gae_onepass": -1}.get(k_, 0))
for name in variants:
    ks = sorted(res[name][0])
    print(f"{name:18s} step us median/min: {np.median(wall[name]):.1f}/{min(wall[name]):.1f} |",
          " ".join(f"{k}={np.median([x[k] for x in res[name]]):.1f}" for k in ks), flush=True)
buf.close()
