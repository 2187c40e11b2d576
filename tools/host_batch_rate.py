#!/usr/bin/env python3
"""PCIe-inclusive hand-off rate (outside the metric): one TGT pass consumed through ppo_xp_pass_next_host, i.e.
every minibatch copied into caller host memory as the reference's next() needs it (data.rs:166-182), from
page-locked and from ordinary destination arrays; plus what the worker-thread shuffle takes off the caller."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ppo_amd
from ppo_amd import synth

T, N, D, B = synth.CONFIGS["TGT"]
M = T * N
buf = ppo_amd.ExperienceBuffer(N, D, T)
r = synth.rollout_device(T, N, D, 1, torch.device("cuda", 0))
for t in range(T):
    buf.add_experience_device(r["obs"][t], r["rewards"][t], r["actions"][t], r["values"][t], r["dones"][t], r["neglogps"][t])
ld = r["last_dones"].cpu().numpy(); del r
buf.returns(ld)
batcher = ppo_amd.ExperienceBatcher(buf, B)
nbytes = M * (4 * D + 8 + 16)
for kind, mk in (("pinned", ppo_amd.HostBatch.pinned), ("pageable", ppo_amd.HostBatch.pageable)):
    hb = mk(B, D)
    for a in (hb.observations, hb.actions, hb.values, hb.neglogps, hb.returns, hb.advantages):
        a[...] = 0
    best = 1e9
    for rep in range(3):
        it = batcher.into_iter(device_shuffle=True, seed=rep)
        buf.sync()
        t0 = time.perf_counter()
        n = 0
        while it.next_host(hb) is not None:
            n += 1
        best = min(best, time.perf_counter() - t0)
    print(f"{kind:9s} {n} x pass_next_host of {B} rows: {best * 1e3:7.1f} ms per pass = {nbytes / best / 1e9:5.1f} GB/s "
          f"= {M / best / 1e6:6.1f} M transitions/s", flush=True)
# a consumer that spends `work` ms on every minibatch (copy_from + learner.step, training.rs:205-207): the synchronous
# hand-off costs PCIe + consumer, the pipelined one max(PCIe, consumer)
def consume(ms):
    t_end = time.perf_counter() + ms * 1e-3
    while time.perf_counter() < t_end:
        pass

K = M // B
for work in (0.0, 1.0, 2.0, 4.0):
    res = {}
    for mode in ("synchronous", "pipelined"):
        best = 1e9
        for rep in range(3):
            it = batcher.into_iter(seed=rep, bijection=True)
            buf.sync()
            t0 = time.perf_counter()
            n = 0
            if mode == "synchronous":
                hb = it._host_batch = getattr(batcher, "_hb", None) or ppo_amd.HostBatch.pinned(B, D)
                batcher._hb = hb
                while it.next_host(hb) is not None:
                    consume(work); n += 1
            else:
                it._host_pair = getattr(batcher, "_pair", None) or (ppo_amd.HostBatch.pinned(B, D), ppo_amd.HostBatch.pinned(B, D))
                batcher._pair = it._host_pair
                for b in it.host_batches():
                    consume(work); n += 1
            best = min(best, time.perf_counter() - t0)
        assert n == K
        res[mode] = best * 1e3
    print(f"consumer {work:3.1f} ms/minibatch ({K * work:6.1f} ms per pass): synchronous {res['synchronous']:6.1f} ms per pass, "
          f"pipelined {res['pipelined']:6.1f} ms per pass", flush=True)
# host Fisher-Yates on the caller's thread vs prefetched by the library's worker while the previous pass is consumed
t0 = time.perf_counter(); it = batcher.into_iter(seed=100); t_fy = time.perf_counter() - t0
batcher.prefetch_shuffle(101)
time.sleep(0.3)                                  # "the previous pass trains"
t0 = time.perf_counter(); it = batcher.into_iter(seed=101); t_pf = time.perf_counter() - t0
t0 = time.perf_counter(); it = batcher.into_iter(seed=102, device_shuffle=True); buf.sync(); t_dev = time.perf_counter() - t0
print(f"into_iter wall time: host Fisher-Yates {t_fy * 1e3:.1f} ms, prefetched by the worker thread {t_pf * 1e3:.1f} ms, "
      f"drawn on the device {t_dev * 1e3:.1f} ms", flush=True)
buf.close()
