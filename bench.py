#!/usr/bin/env python3
"""bench.py — transitions/sec through GAE + advantage standardisation + one shuffled pass of K
minibatch gathers (BASELINE.json's metric), on N GPUs of one node, env-sharded (weak scaling).

  python bench.py [--gpus N --steps K --warmup W]          (N = 1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = ExperienceBuffer.returns(last_dones) [GAE kernel] followed by one full shuffled pass
[gather + statistics + standardise kernels] over a synthetic rollout that is already resident in
HBM.  Rollout insertion and permutation generation are outside the figure; they
are timed separately and printed to stderr.  Rank 0 prints ONE JSON line on stdout.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
ACTION_BYTES, INDEX_BYTES = 8, 4


def alg_bytes_total(D):      # B_alg = 8D + 2A + I + 56
    return 8 * D + 2 * ACTION_BYTES + INDEX_BYTES + 56


# How B_alg splits over the kernels (DESIGN.md "Kernels"): GAE 20 | scalar gather: index I +
# actions 2A + values/neglogps/returns 24 + advantage reads 8 | normalize: standardised write 4 |
# observation gather: 8D.
def alg_bytes_obs_gather(D):
    return 8 * D


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(T, D, K, target_seconds=12.0):
    """Times the CPU oracle (op-for-op restatement of the reference, single thread like the
    reference) on a bounded sample of the same workload: same T, D, K, fewer envs."""
    from oracle import oracle as O
    from ppo_amd import synth
    Ns = 16384
    M = T * Ns
    B = M // K
    r = synth.rollout(T, Ns, D, 0x5EED)
    streams = dict(obs=r["obs"].reshape(M, D), rewards=r["rewards"].reshape(-1), actions=r["actions"].reshape(-1),
                   values=r["values"].reshape(-1), neglogps=r["neglogps"].reshape(-1),
                   dones=r["dones"].astype(np.float32).reshape(-1))
    perm = O.shuffle(0x5EED, M)
    outs = O.run_pass(streams, T, Ns, D, T - 1, r["last_dones"], 0.99, 0.95, perm, B, nthreads=1)
    times, t_end = [], time.time() + target_seconds
    while time.time() < t_end or len(times) < 3:
        t0 = time.perf_counter()
        O.run_pass(streams, T, Ns, D, T - 1, r["last_dones"], 0.99, 0.95, perm, B, nthreads=1, outs=outs)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    ncores = min(O.max_threads(), 16)
    tm = []
    for _ in range(3):
        t0 = time.perf_counter()
        O.run_pass(streams, T, Ns, D, T - 1, r["last_dones"], 0.99, 0.95, perm, B, nthreads=ncores, outs=outs)
        tm.append(time.perf_counter() - t0)
    return {"value": M / med, "unit": "transitions/s", "cores": 1, "kind": "port",
            "sample": f"T={T}, N={Ns} envs (of the workload's per-GPU envs), obs_dim={D}, K={K} minibatches of {B}; "
                      f"{len(times)} passes of the CPU oracle (oracle/ppo_xp_oracle.c, -O2 -ffp-contract=off), median; "
                      "the Rust reference itself cannot be built here, the oracle is an upper bound on its speed",
            "all_cores": {"value": M / min(tm), "cores": ncores}}


def quick_measure(cfg, device, steps=200, warmup=20):
    """Untimed-by-events whole-step rate of another BASELINE.json config on one GPU (same step as the main
    measurement: returns() + one shuffled pass, inputs resident), reported beside the headline."""
    import torch
    import ppo_amd
    from ppo_amd import _lib, synth
    T, N, D, B = synth.CONFIGS[cfg]
    M, K = T * N, (T * N) // B
    buf = ppo_amd.ExperienceBuffer(N, D, T, device=device)
    buf.use_torch_stream()
    L, h = _lib.load(), buf.handle
    r = synth.rollout_device(T, N, D, synth.SEEDS[cfg], torch.device("cuda", device))
    for t in range(T):
        buf.add_experience_device(r["obs"][t], r["rewards"][t], r["actions"][t], r["values"][t], r["dones"][t], r["neglogps"][t])
    last_dones = r["last_dones"].cpu().numpy()
    ld_p = last_dones.ctypes.data_as(C.c_void_p)
    _lib.check(L.ppo_xp_returns(h, ld_p, N, 0, None))
    _lib.check(L.ppo_xp_pass_begin(h, B, None, synth.SEEDS[cfg]))
    perm = ppo_amd.device_tensor(L.ppo_xp_pass_perm_device(h), (M,), torch.uint32, device, buf).clone()
    bv = _lib.BatchView()

    def step():
        _lib.check(L.ppo_xp_returns(h, ld_p, N, 0, None))
        _lib.check(L.ppo_xp_pass_begin_device(h, B, C.c_void_p(perm.data_ptr())))
        n = 0
        while L.ppo_xp_pass_next(h, C.byref(bv)) == 0:
            n += 1
        assert n == K
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    buf.close()
    value = M * steps / dt
    return {"workload": f"{cfg}: T={T}, N={N} envs, obs_dim={D}, batch={B} (K={K})", "value": value, "unit": "transitions/s",
            "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup,
            "hbm_roofline_frac_whole_step": value * alg_bytes_total(D) / (HBM_PEAK_GBS * 1e9)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="TGT",
                    help="TGT | C2 | C3: every rank owns the config's N envs; C5: its 1 048 576 envs and batch are split "
                         "over the ranks; C4: GAE only (T=16384, N=1024)")
    ap.add_argument("--gae", default=None, choices=["seq", "assoc"], help="GAE algorithm (default: seq; assoc for C4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the C2 line that a default one-GPU run reports beside TGT")
    ap.add_argument("--exchange", default=os.environ.get("PPO_XP_EXCHANGE", "torch"), choices=["torch", "rccl"])
    ap.add_argument("--backend", default=os.environ.get("PPO_XP_BACKEND", "nccl"), choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the multi-rank control flow with all ranks on one GPU")
    ap.add_argument("--tunable", action="append", default=[], help="name=value (see ppo_xp_set_tunable)")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not record per-kernel HIP events in the timed region")
    args = ap.parse_args()

    # RCCL prints a version banner on stdout at communicator creation; the contract is ONE JSON line
    # there.  Everything below writes to stderr's file until the result is printed.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}")
    if args.backend == "gloo":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # PPO_XP_FORCE_PG=1: run the multi-rank control flow (process group, exchange, barriers) with ONE
    # rank, the only RCCL rehearsal a one-GPU box allows
    multi = world > 1 or os.environ.get("PPO_XP_FORCE_PG") == "1"
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    import ppo_amd
    from ppo_amd import _lib, sharded, synth

    T, N, D, B = synth.CONFIGS[args.config]   # per-GPU shard: weak scaling, every rank owns N envs
    if args.config == "C5":                   # N/G envs and B/G rows per GPU
        assert N % world == 0 and B % world == 0
        N, B = N // world, B // world
    gae_only = B == 0
    M = T * N
    K = 0 if gae_only else M // B
    algo = {"seq": 0, "assoc": 1}[args.gae or ("assoc" if args.config == "C4" else "seq")]
    if gae_only:
        algo |= 0x100                         # PPO_XP_GAE_NO_PACK: no pass follows, no records needed
    seed = synth.SEEDS[args.config] + 1000 * rank

    buf = ppo_amd.ExperienceBuffer(N, D, T, device=local_rank)
    buf.use_torch_stream()                      # torch events / RCCL see the kernels' stream
    for t in args.tunable:
        k, v = t.split("=")
        buf.set_tunable(k, int(v))
    L, h = _lib.load(), buf.handle

    # ---- synthetic rollout, drawn on the device and inserted row by row (not timed) ----------
    t0 = time.perf_counter()
    r = synth.rollout_device(T, N, D, seed, dev)
    for t in range(T):
        buf.add_experience_device(r["obs"][t], r["rewards"][t], r["actions"][t], r["values"][t], r["dones"][t],
                                  r["neglogps"][t])
    buf.sync()
    t_insert = time.perf_counter() - t0
    last_dones = r["last_dones"].cpu().numpy()
    del r
    torch.cuda.empty_cache()

    # ---- permutations (host Fisher-Yates in the library, seeded; not timed) ------------------
    t0 = time.perf_counter()
    ld_p = last_dones.ctypes.data_as(C.c_void_p)
    _lib.check(L.ppo_xp_returns(h, ld_p, N, algo, None))
    perms = []
    for i in range(0 if gae_only else 2):
        _lib.check(L.ppo_xp_pass_begin(h, B, None, seed + i))
        p = ppo_amd.device_tensor(L.ppo_xp_pass_perm_device(h), (M,), torch.uint32, local_rank, buf)
        buf.sync()
        perms.append(p.clone())
    t_perm = (time.perf_counter() - t0) / 2
    exch = multi and not gae_only     # the exchange of minibatch sums only exists when there is a pass

    # ---- exchange of the per-minibatch sums across ranks (the only cross-GPU step) -----------
    comm = None
    if exch:
        if args.exchange == "rccl":
            uid = torch.zeros(128, dtype=torch.uint8)
            if rank == 0:
                raw = (C.c_char * 128)()
                _lib.check(L.ppo_xp_rccl_unique_id(raw))
                uid = torch.frombuffer(bytearray(raw.raw), dtype=torch.uint8).clone()
            uid = uid.to(dev) if args.backend == "nccl" else uid
            dist.broadcast(uid, 0)
            raw = (C.c_char * 128).from_buffer_copy(bytes(uid.cpu().numpy().tobytes()))
            comm = C.c_void_p()
            _lib.check(L.ppo_xp_rccl_comm_init(C.byref(comm), raw, rank, world))
            buf.set_exchange(1, comm.value)
        else:
            sb = sharded.ShardedBatcher(buf, B * world)      # global batch = world x local batch; sets exchange mode 2

    def step(i):
        _lib.check(L.ppo_xp_returns(h, ld_p, N, algo, None))
        if gae_only:
            return
        _lib.check(L.ppo_xp_pass_begin_device(h, B, C.c_void_p(perms[i & 1].data_ptr())))
        if exch and args.exchange == "torch":
            sb.exchange()                                    # ONE all-reduce of 3K doubles, then standardise
        bv = _lib.BatchView()
        n = 0
        while L.ppo_xp_pass_next(h, C.byref(bv)) == 0:
            n += 1
        assert n == K

    def barrier():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    # timed region: HIP events only around the dominant kernel (an event pair costs ~7 us of stream time)
    dom = "gae_scan" if gae_only else "gather"      # the dominant kernel of the step
    buf.prof_enable(False if args.no_kernel_events else [dom])
    buf.prof_collect()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    dt = time.perf_counter() - t0
    prof = buf.prof_collect()
    # per-kernel breakdown: a few more steps with every kernel bracketed, outside the timed region
    buf.prof_enable(True)
    for i in range(5):
        step(i)
    barrier()
    prof_all = buf.prof_collect()
    buf.prof_enable(False)

    if multi:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = world * M * args.steps / dt
        g_ms, g_n = prof[dom]
        g_avg_s = g_ms / max(g_n, 1) * 1e-3
        dom_bytes = 20 if gae_only else alg_bytes_obs_gather(D)     # per transition, SURVEY.md 8(d)
        step_bytes = 20 if gae_only else alg_bytes_total(D)
        achieved = M * dom_bytes / g_avg_s / 1e9 if g_n else 0.0
        kern = {k: {"avg_us": v[0] / v[1] * 1e3, "launches": v[1]} for k, v in prof_all.items() if v[1]}
        try:
            metric = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
        except Exception:
            metric = "transitions/sec through GAE+norm+minibatch gather; % of HBM roofline, 1\u20138 GPU"
        out = {
            "metric": metric, "value": value, "unit": "transitions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong" if args.config == "C5" else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"{args.config}: T={T}, N={N} envs per GPU, GAE + returns only, "
                                    f"{'associative (chunked)' if algo & 1 else 'sequential'} scan, gamma=0.99 lambda=0.95") if gae_only else
                                   (f"{args.config}: T={T}, N={N} envs per GPU, obs_dim={D}, batch={B} (K={K} minibatches/pass), "
                                    f"gamma=0.99 lambda=0.95, {'associative' if algo & 1 else 'sequential'} GAE scan + one shuffled pass"),
                       "global_envs": world * N, "parallelism": f"env-sharded x{world}" if multi else "single GPU",
                       "exchange": ((args.exchange + "/" + args.backend) if multi else "none")},
            "hbm_roofline_frac_whole_step": value * step_bytes / (world * HBM_PEAK_GBS * 1e9),
            "alg_bytes_per_transition": step_bytes,
            "roofline": {"bound": "hbm", "kernel": "gae_scan_kernel" if gae_only else "obs_gather_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "alg_bytes_per_launch": M * dom_bytes, "avg_launch_us": g_avg_s * 1e6,
                         "launches_timed": g_n},
            "kernels_untimed_breakdown": kern,
            "off_metric": {"insert_s": t_insert, "perm_host_fisher_yates_s": t_perm},
        }
        traffic_file = os.path.join(ROOT, "profiles", "gather_traffic.json")
        if os.path.exists(traffic_file):
            try:
                tf = json.load(open(traffic_file))
                if tf.get("config") == args.config and tf.get("kernel") == "obs_gather_kernel":
                    out["roofline"]["traffic"] = tf.get("hbm_bytes_per_launch")
                    out["roofline"]["traffic_source"] = tf.get("source")
            except Exception:
                pass
        if not args.no_cpu_baseline and world == 1 and not gae_only:
            out["cpu_baseline"] = cpu_baseline(T, D, K)
    if comm is not None:
        buf.sync()
        L.ppo_xp_rccl_comm_destroy(comm)
    buf.close()
    if rank == 0:
        # BASELINE.json's configs[1] (C2) beside the headline, measured the same way (no events at all)
        if world == 1 and not multi and args.config == "TGT" and not args.no_secondary:
            torch.cuda.empty_cache()
            out["also"] = {"C2": quick_measure("C2", local_rank)}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
