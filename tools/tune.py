#!/usr/bin/env python3
"""Interleaved A/B of kernel tunables in ONE process (cdna guide rule 24) on a bench config.
Prints per-variant median / min kernel times from HIP events.  Not part of the product."""
import argparse
import ctypes as C
import itertools
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="TGT")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--grid", default="gather_nt=0,1;gather_block=64,256")
    ap.add_argument("--kernel", default="gather")
    ap.add_argument("--algo", type=int, default=0)
    args = ap.parse_args()
    import torch
    import ppo_amd
    from ppo_amd import _lib, synth
    T, N, D, B = synth.CONFIGS[args.config]
    if B == 0:
        B = T * N
    M = T * N
    dev = torch.device("cuda", 0)
    buf = ppo_amd.ExperienceBuffer(N, D, T)
    buf.use_torch_stream()
    r = synth.rollout_device(T, N, D, 1, dev)
    for t in range(T):
        buf.add_experience_device(r["obs"][t], r["rewards"][t], r["actions"][t], r["values"][t], r["dones"][t], r["neglogps"][t])
    ld = r["last_dones"].cpu().numpy()
    del r
    L, h = _lib.load(), buf.handle
    ldp = ld.ctypes.data_as(C.c_void_p)
    _lib.check(L.ppo_xp_returns(h, ldp, N, args.algo, None))
    _lib.check(L.ppo_xp_pass_begin(h, B, None, 1))
    perm = ppo_amd.device_tensor(L.ppo_xp_pass_perm_device(h), (M,), torch.uint32, 0, buf).clone()
    buf.sync()

    # stream-copy ceiling of this box at the gather's size (read M*D*4, write M*D*4)
    src = torch.empty(M * D, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ts = []
    for _ in range(10):
        ev[0].record(); dst.copy_(src); ev[1].record(); torch.cuda.synchronize()
        ts.append(ev[0].elapsed_time(ev[1]))
    print(f"copy ceiling: {2 * M * D * 4 / (np.median(ts) * 1e-3) / 1e9:.0f} GB/s (median of 10, {M*D*4/1e9:.2f} GB each way)")
    del src, dst

    names, values = [], []
    for part in args.grid.split(";"):
        k, v = part.split("=")
        names.append(k); values.append([int(x) for x in v.split(",")])
    variants = list(itertools.product(*values))
    res = {v: [] for v in variants}
    buf.prof_enable(True)
    for rnd in range(args.rounds + 1):
        for v in variants:
            for k, x in zip(names, v):
                buf.set_tunable(k, x)
            _lib.check(L.ppo_xp_returns(h, ldp, N, args.algo, None))
            _lib.check(L.ppo_xp_pass_begin_device(h, B, C.c_void_p(perm.data_ptr())))
            p = buf.prof_collect()
            if rnd:
                res[v].append({k: ms / max(n, 1) * 1e3 for k, (ms, n) in p.items() if n})
    for v in variants:
        ks = sorted(res[v][0])
        line = " ".join(f"{k}={np.median([x[k] for x in res[v]]):.1f}/{min(x[k] for x in res[v]):.1f}" for k in ks)
        print(dict(zip(names, v)), "us median/min:", line)
    buf.close()


if __name__ == "__main__":
    main()
