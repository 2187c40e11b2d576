#!/usr/bin/env python3
"""A/B of the partitioned scalar path (tunable scalar_path=1) against scalar_gather_kernel, split kernels forced."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ppo_amd
from ppo_amd import _lib, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "TGT"
if "/" in cfg:
    base, div = cfg.split("/"); T, N, D, B = synth.CONFIGS[base]; N, B = N // int(div), B // int(div)
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
L, h = _lib.load(), buf.handle
ldp = ld.ctypes.data_as(C.c_void_p)
_lib.check(L.ppo_xp_returns(h, ldp, N, 0, None))
_lib.check(L.ppo_xp_pass_begin_device_shuffle(h, B, 1))
perm = ppo_amd.device_tensor(L.ppo_xp_pass_perm_device(h), (M,), torch.uint32, 0, buf).clone()
buf.sync()
buf.set_tunable("fused", 0)
res, wall = {0: [], 1: []}, {0: [], 1: []}
for rnd in range(6):
    for path in (0, 1):
        buf.set_tunable("scalar_path", path)
        buf.prof_enable(True)
        _lib.check(L.ppo_xp_returns(h, ldp, N, 0, None))
        _lib.check(L.ppo_xp_pass_begin_device(h, B, C.c_void_p(perm.data_ptr())))
        p = buf.prof_collect(); buf.prof_enable(False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            _lib.check(L.ppo_xp_returns(h, ldp, N, 0, None))
            _lib.check(L.ppo_xp_pass_begin_device(h, B, C.c_void_p(perm.data_ptr())))
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        if rnd:
            res[path].append({k: ms / max(n, 1) * 1e3 for k, (ms, n) in p.items() if n}); wall[path].append(dt * 1e6)
for path in (0, 1):
    ks = sorted(res[path][0])
    print(cfg, "scalar_path", path, "step us median/min: %.1f/%.1f" % (np.median(wall[path]), min(wall[path])), "|",
          " ".join(f"{k}={np.median([x[k] for x in res[path]]):.1f}" for k in ks), flush=True)
buf.close()
