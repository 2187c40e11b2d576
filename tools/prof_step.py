#!/usr/bin/env python3
"""A few steps of the bench workload with nothing else around them, for rocprofv3 --pmc passes
(the program rocprofv3 runs must be the interpreter itself: rocprofv3 ... -- python3 tools/prof_step.py)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    config = sys.argv[1] if len(sys.argv) > 1 else "TGT"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    algo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    import torch
    import ppo_amd
    from ppo_amd import _lib, synth
    T, N, D, B = synth.CONFIGS[config]
    if len(sys.argv) > 4:                 # fewer time steps of the same shape (the PMC tool does not survive 34 GB buffers)
        T = int(sys.argv[4])
        B = (T * N) // 32 if B else 0
    M = T * N
    if B == 0:
        B = M
    dev = torch.device("cuda", 0)
    buf = ppo_amd.ExperienceBuffer(N, D, T)
    buf.use_torch_stream()
    for kv in filter(None, os.environ.get("PPO_XP_TUNABLES", "").split(",")):   # e.g. fused=1,gae_flat=1
        k, v = kv.split("=")
        buf.set_tunable(k, int(v))
    if T > 4096:     # thousands of row inserts under --pmc crash the profiler: bulk-insert small rollouts from the host
        r = synth.rollout(T, N, D, 1)
        buf.add_many(r["obs"], r["rewards"], r["actions"], r["values"], r["dones"], r["neglogps"])
        ld = r["last_dones"]
    else:
        r = synth.rollout_device(T, N, D, 1, dev)
        for t in range(T):
            buf.add_experience_device(r["obs"][t], r["rewards"][t], r["actions"][t], r["values"][t], r["dones"][t], r["neglogps"][t])
        ld = r["last_dones"].cpu().numpy()
    del r
    L, h = _lib.load(), buf.handle
    ldp = ld.ctypes.data_as(C.c_void_p)
    _lib.check(L.ppo_xp_returns(h, ldp, N, algo, None))
    if os.environ.get("PPO_XP_PERM", "bijection") == "bijection":   # what bench.py times: the keyed bijection inside the step
        for i in range(steps):
            _lib.check(L.ppo_xp_returns(h, ldp, N, algo, None))
            if B < M or M:
                _lib.check(L.ppo_xp_pass_begin_bijection(h, B, 100 + i))
    else:
        _lib.check(L.ppo_xp_pass_begin(h, B, None, 1))
        perm = ppo_amd.device_tensor(L.ppo_xp_pass_perm_device(h), (M,), torch.uint32, 0, buf).clone()
        buf.sync()
        for _ in range(steps):
            _lib.check(L.ppo_xp_returns(h, ldp, N, algo, None))
            _lib.check(L.ppo_xp_pass_begin_device(h, B, C.c_void_p(perm.data_ptr())))
    buf.sync()
    buf.close()


if __name__ == "__main__":
    main()
