#!/usr/bin/env python3
"""Does capturing one step (returns() + one shuffled pass = 3-4 launches) in a HIP graph shorten it?  Measures a
config with the driver-style clock (wall time over N steps between synchronisations) three ways: plain stream
launches through the C ABI, the same calls replayed from a captured graph, and the sum of the kernels' own
durations (HIP events) as the floor no launch mechanism can beat."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ppo_amd
from ppo_amd import _lib, synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 500
T, N, D, B = synth.CONFIGS[cfg]
M = T * N
dev = torch.device("cuda", 0)
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    buf = ppo_amd.ExperienceBuffer(N, D, T)
    buf.use_torch_stream()
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

    def step():
        _lib.check(L.ppo_xp_returns(h, ldp, N, 0, None))
        _lib.check(L.ppo_xp_pass_begin_device(h, B, C.c_void_p(perm.data_ptr())))

    def clock(fn, n):
        for _ in range(20):
            fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6

    plain = [clock(step, steps) for _ in range(3)]
    buf.prof_enable(True)
    for _ in range(20):
        step()
    p = buf.prof_collect(); buf.prof_enable(False)
    ksum = sum(ms / n * 1e3 for ms, n in p.values() if n)
    for _ in range(4):          # leave the last_dones slot ring at a block boundary: no event wait inside the capture
        step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side, capture_error_mode="relaxed"):   # the handle already enqueues on `side`
        for _ in range(3):      # 3 steps per graph: calls 1..3 of a slot block record no event either
            step()
    graph = [clock(g.replay, steps // 3) / 3 for _ in range(3)]
    print(f"{cfg}: plain stream launches {min(plain):.1f} us/step (runs: {', '.join('%.1f' % x for x in plain)})")
    print(f"{cfg}: graph replay (3 steps per graph) {min(graph):.1f} us/step (runs: {', '.join('%.1f' % x for x in graph)})")
    print(f"{cfg}: sum of kernel durations (HIP events) {ksum:.1f} us/step: " + " ".join(f"{k}={ms / n * 1e3:.1f}" for k, (ms, n) in p.items() if n))
    roof = M * (8 * D + 76) / 8e12 * 1e6
    print(f"{cfg}: roofline {roof:.1f} us -> plain {roof / min(plain):.3f}, graph {roof / min(graph):.3f}")
    buf.close()
