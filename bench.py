#!/usr/bin/env python3
"""bench.py — transitions/sec through GAE + advantage standardisation + one shuffled pass of K
minibatch gathers (BASELINE.json's metric), on N GPUs of one node, env-sharded (weak scaling).

  python bench.py [--gpus N --steps K --warmup W]
      N = 1: this process measures.  N > 1 without a torch.distributed environment: this process
      starts the N ranks itself (python -m torch.distributed.run ... bench.py ...) BEFORE it touches
      the GPU, passes rank 0's JSON line through and exits with the launcher's status.
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
      one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment).

A step = ExperienceBuffer.returns(last_dones) [GAE kernel] followed by one full shuffled pass
[gather + statistics + standardise kernels] over a synthetic rollout that is already resident in
HBM.  Rollout insertion and permutation generation are outside the figure; they
are timed separately and reported under "off_metric".  Rank 0 prints ONE JSON line on stdout: the
headline is TGT (per GPU: weak scaling); "also" holds the other BASELINE.json configs measured the
same way (one GPU: C2, C3, C4, C5 whole; N GPUs: C5 split over the ranks, strong scaling).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
ACTION_BYTES, INDEX_BYTES = 8, 4


def alg_bytes_total(D, index_bytes=INDEX_BYTES):      # B_alg = 8D + 2A + I + 56
    return 8 * D + 2 * ACTION_BYTES + index_bytes + 56


# How B_alg splits over the kernels (DESIGN.md "Kernels"): GAE 20 | scalar gather: index I +
# actions 2A + values/neglogps/returns 24 + advantage reads 8 | normalize: standardised write 4 |
# observation gather: 8D.
def alg_bytes_obs_gather(D):
    return 8 * D


def alg_bytes_fused_gather(D, index_bytes=INDEX_BYTES):   # rows + the scalar gather's share: index I, actions 2A, v/nlp/ret 24, advantage read + raw write 8
    return 8 * D + index_bytes + 2 * ACTION_BYTES + 24 + 8


def log(*a):
    print(*a, file=sys.stderr, flush=True)


# ---------------------------------------------------------------------------------------------
# N > 1 started as a plain `python bench.py --gpus N`: become the launcher (no torch, no HIP here)
# ---------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launcher_command(n, argv, port):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def spawn_ranks(n, argv) -> int:
    """Starts the N ranks and relays rank 0's JSON line.  Runs before `import torch` / any HIP call
    in this process: the children initialise the GPUs, the parent never does."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "8")
    env["PPO_XP_SPAWNED"] = "1"
    cmd = launcher_command(n, argv, _free_port())
    log("bench.py: starting", n, "ranks:", " ".join(cmd))
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout:                                   # the contract is ONE JSON line on stdout
        if ln.startswith('{"metric"'):
            line = ln
        else:
            sys.stderr.write(ln)
    rc = p.wait()
    if line is not None:
        sys.stdout.write(line)
        sys.stdout.flush()
    if rc == 0 and line is None:
        log("bench.py: the ranks exited cleanly but printed no result line")
        return 1
    return rc


# ---------------------------------------------------------------------------------------------
# CPU baseline
# ---------------------------------------------------------------------------------------------
def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(T, N, D, K, target_seconds=10.0):
    """Times the CPU oracle (op-for-op restatement of the reference) on the WHOLE workload (same T, N,
    D, K): one thread like the reference, then every core this process may use."""
    from oracle import oracle as O
    from ppo_amd import synth
    M = T * N
    B = M // K
    r = synth.rollout(T, N, D, 0x5EED)
    streams = dict(obs=r["obs"].reshape(M, D), rewards=r["rewards"].reshape(-1), actions=r["actions"].reshape(-1),
                   values=r["values"].reshape(-1), neglogps=r["neglogps"].reshape(-1),
                   dones=r["dones"].astype(np.float32).reshape(-1))
    perm = O.shuffle(0x5EED, M)
    outs = O.run_pass(streams, T, N, D, T - 1, r["last_dones"], 0.99, 0.95, perm, B, nthreads=1)
    times, t_end = [], time.time() + target_seconds
    while (time.time() < t_end and len(times) < 20) or len(times) < 3:
        t0 = time.perf_counter()
        O.run_pass(streams, T, N, D, T - 1, r["last_dones"], 0.99, 0.95, perm, B, nthreads=1, outs=outs)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    nproc = os.cpu_count() or 1
    env_cap = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    ncores = max(1, min(usable, O.max_threads()) if not env_cap else min(usable, env_cap, O.max_threads()))
    tm = []
    for _ in range(3):
        t0 = time.perf_counter()
        O.run_pass(streams, T, N, D, T - 1, r["last_dones"], 0.99, 0.95, perm, B, nthreads=ncores, outs=outs)
        tm.append(time.perf_counter() - t0)
    return {"value": M / med, "unit": "transitions/s", "cores": 1, "kind": "port",
            "sample": f"the whole workload: T={T}, N={N} envs, obs_dim={D}, K={K} minibatches of {B}; "
                      f"{len(times)} passes of the CPU oracle (oracle/ppo_xp_oracle.c, -O2 -ffp-contract=off), median "
                      f"{med:.3f} s per pass; the Rust reference itself cannot be built here, the oracle is an upper bound on its speed",
            "cpu_model": _cpu_model(), "nproc": nproc, "usable_cores": usable,
            "all_cores": {"value": M / min(tm), "cores": ncores,
                          "note": "OpenMP over envs / rows; threads = the cores this process may use (OMP_NUM_THREADS cap honoured)"}}


# ---------------------------------------------------------------------------------------------
# one configuration on this rank's GPU
# ---------------------------------------------------------------------------------------------
class Ctx:
    def __init__(self, args, rank, world, local_rank, dev, multi):
        self.args, self.rank, self.world, self.local_rank, self.dev, self.multi = args, rank, world, local_rank, dev, multi


def measure(ctx, cfg, steps, warmup, gae=None, events=True, breakdown=True, tunables=(), exchange=None):
    """returns() + one shuffled pass per step on config `cfg`; every rank owns the config's envs (C5: its
    share).  Returns (result dict, M, D, K, gae_only) — the dict is complete on rank 0."""
    import torch
    import torch.distributed as dist
    import ppo_amd
    from ppo_amd import _lib, sharded, synth

    args, rank, world, dev = ctx.args, ctx.rank, ctx.world, ctx.dev
    exchange = exchange or args.exchange
    T, N, D, B = synth.CONFIGS[cfg]           # per-GPU shard: weak scaling, every rank owns N envs
    if cfg == "C5":                           # N/G envs and B/G rows per GPU
        assert N % world == 0 and B % world == 0
        N, B = N // world, B // world
    gae_only = B == 0
    M = T * N
    K = 0 if gae_only else M // B
    algo = {"seq": 0, "assoc": 1}[gae or ("assoc" if cfg == "C4" else "seq")]
    if gae_only:
        algo |= 0x100                         # PPO_XP_GAE_NO_PACK: no pass follows, no records needed
    seed = synth.SEEDS[cfg] + 1000 * rank

    # an ordinary (non-NULL) stream for everything: the legacy NULL stream torch uses by default synchronises
    # implicitly with every other blocking stream of the process
    torch.cuda.set_stream(torch.cuda.Stream(dev))
    buf = ppo_amd.ExperienceBuffer(N, D, T, device=ctx.local_rank)
    buf.use_torch_stream()                    # torch events / RCCL see the kernels' stream
    for t in tunables:
        k, v = t.split("=")
        buf.set_tunable(k, int(v))
    if args.streaming and not ctx.multi:
        buf.set_streaming(True)
    L, h = _lib.load(), buf.handle

    # ---- synthetic rollout, drawn on the device and inserted row by row (not timed) ----------
    t0 = time.perf_counter()
    if T * N * D * 4 > (8 << 30):             # draw large rollouts in slabs of time steps (bounded temporaries)
        tstep = max(1, (4 << 30) // (N * D * 4))
        last_dones = None
        for tb in range(0, T, tstep):
            r = synth.rollout_device(min(tstep, T - tb), N, D, seed + tb, dev)
            for t in range(r["rewards"].shape[0]):
                buf.add_experience_device(r["obs"][t], r["rewards"][t], r["actions"][t], r["values"][t], r["dones"][t],
                                          r["neglogps"][t])
            last_dones = r["last_dones"].cpu().numpy()
            buf.sync()
            del r
    else:
        r = synth.rollout_device(T, N, D, seed, dev)
        for t in range(T):
            buf.add_experience_device(r["obs"][t], r["rewards"][t], r["actions"][t], r["values"][t], r["dones"][t],
                                      r["neglogps"][t])
        last_dones = r["last_dones"].cpu().numpy()
        buf.sync()
        del r
    t_insert = time.perf_counter() - t0
    torch.cuda.empty_cache()

    # ---- permutations: host Fisher-Yates in the library, seeded (reference-faithful; not timed), or the
    # device generator for the very large passes where the host shuffle takes seconds ------------------
    ld_p = last_dones.ctypes.data_as(C.c_void_p)
    _lib.check(L.ppo_xp_returns(h, ld_p, N, algo, None))
    perms, t_perm, perm_kind = [], 0.0, "none"
    bijection = args.perm == "bijection"
    if not gae_only and bijection:
        # the permutation of every pass is a keyed bijection of (seed, M) evaluated inside the gather: a fresh one per
        # step, generated INSIDE the timed region (nothing is drawn, uploaded or read beforehand)
        perm_kind = "inline keyed bijection"
    elif not gae_only:
        host_fy = args.perm == "host" and M <= (1 << 25)
        perm_kind = "host Fisher-Yates (rand 0.8.5 / ChaCha12)" if host_fy else "device generator (SplitMix64 keys + radix sort)"
        t0 = time.perf_counter()
        for i in range(2):
            if host_fy:
                _lib.check(L.ppo_xp_pass_begin(h, B, None, seed + i))
            else:
                _lib.check(L.ppo_xp_pass_begin_device_shuffle(h, B, seed + i))
            p = ppo_amd.device_tensor(L.ppo_xp_pass_perm_device(h), (M,), torch.uint32, ctx.local_rank, buf)
            buf.sync()
            perms.append(p.clone())
        t_perm = (time.perf_counter() - t0) / 2
    exch = ctx.multi and not gae_only     # the exchange of minibatch sums only exists when there is a pass

    # ---- exchange of the per-minibatch sums across ranks (the only cross-GPU step) -----------
    comm, sb = None, None
    rccl_ranks = (world if (ctx.multi and args.backend == "nccl") else 0)   # ranks of the RCCL communicator the exchange runs on
    if exch:
        if exchange == "rccl":
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
            nr = C.c_int(0)
            _lib.check(L.ppo_xp_rccl_comm_count(comm, C.byref(nr)))
            rccl_ranks = nr.value
        else:
            sb = sharded.ShardedBatcher(buf, B * world)      # global batch = world x local batch; sets exchange mode 2

    bv = _lib.BatchView()
    step_no = [0]
    xev = None          # torch events around the caller-side all-reduce (breakdown steps only)

    def step(i):
        _lib.check(L.ppo_xp_returns(h, ld_p, N, algo, None))
        if gae_only:
            return
        if bijection:
            _lib.check(L.ppo_xp_pass_begin_bijection(h, B, (seed << 20) + step_no[0]))
            step_no[0] += 1
        else:
            _lib.check(L.ppo_xp_pass_begin_device(h, B, C.c_void_p(perms[i & 1].data_ptr())))
        if sb is not None:
            if xev is not None:
                xev[0].record()
            sb.exchange()                                    # ONE all-reduce of 3K doubles, then standardise
            if xev is not None:
                xev[1].record()
        n = 0
        while L.ppo_xp_pass_next(h, C.byref(bv)) == 0:
            n += 1
        assert n == K

    def barrier():
        torch.cuda.synchronize()
        if ctx.multi:
            dist.barrier()
        torch.cuda.synchronize()

    # The set-up above is host work and a few small launches: the GPU is idle and takes ~0.2 s of load to reach its steady
    # clocks (the first 25 steps of a cold run take 873 us, steady state is 859: tools/step_curve.py,
    # profiles/r3/step_curve_cold_vs_warm.log).  The box's copy-rate measurement (context for the roofline)
    # therefore runs HERE, for args.spinup_ms, right before the W warm-up steps and the K timed steps of the workload itself.
    ceiling = None
    if not args.no_copy_ceiling:
        try:
            ceiling = copy_ceiling(dev, min_ms=args.spinup_ms)
        except Exception as e:          # context only: never costs the line
            log(f"bench.py: copy ceiling not measured: {type(e).__name__}: {e}")
    for i in range(warmup):
        step(i)
    barrier()
    gkind = -1 if gae_only else L.ppo_xp_pass_gather_kind(h)
    fused = gkind in (1, 2)
    # timed region: HIP events only around the dominant kernel, and only on every 6th step (an event pair costs
    # ~7 us of stream time: bracketing every launch would tax the very number it is there to explain)
    dom = "gae_scan" if gae_only else "gather"      # the dominant kernel of the step
    buf.prof_enable(False)
    buf.prof_collect()
    t0 = time.perf_counter()
    for i in range(steps):
        if events and i % 6 == min(3, steps - 1):   # not step 0 (unless it is the only one): the first launch after the barrier's host synchronisation starts on an idle queue
            buf.prof_enable([dom])
            step(i)
            buf.prof_enable(False)
        else:
            step(i)
    barrier()
    dt = time.perf_counter() - t0
    prof = buf.prof_collect()
    prof_all = {}
    exch_us = None
    if breakdown:   # per-kernel breakdown: a few more steps with every kernel bracketed, outside the timed region
        buf.prof_enable(True)
        xt = []
        for i in range(5):
            if sb is not None:
                xev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            step(i)
            if xev is not None:
                torch.cuda.synchronize()
                xt.append(xev[0].elapsed_time(xev[1]) * 1e3)
                xev = None
        barrier()
        prof_all = buf.prof_collect()
        if xt:
            exch_us = float(np.median(xt))        # all-reduce + the standardise launch behind it, on the pass's stream
        elif prof_all.get("exchange", (0, 0))[1]:
            exch_us = prof_all["exchange"][0] / prof_all["exchange"][1] * 1e3
    buf.prof_enable(False)

    extra = {}
    if cfg == "C4" and rank == 0:
        # the looser bound of the associative scan, measured here: K2 against K1 on this very rollout
        r1 = buf.returns(last_dones, algo=0).clone()
        r2 = buf.returns(last_dones, algo=1)
        err = ((r2 - r1).abs() / r1.abs().clamp(min=1.0)).max().item()
        extra["assoc_vs_sequential_max_rel_err"] = err
        extra["assoc_bound_stated"] = 1e-5
        del r1, r2

    rank_ms = [dt / steps * 1e3]
    if ctx.multi:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        every = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(every, tt)
        rank_ms = [float(t.item()) / steps * 1e3 for t in every]
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if comm is not None:
        buf.sync()
        L.ppo_xp_rccl_comm_destroy(comm)
    buf.close()
    del perms
    torch.cuda.synchronize()
    torch.cuda.set_stream(torch.cuda.default_stream(dev))
    torch.cuda.empty_cache()

    ms_per_step = dt / steps * 1e3
    value = world * M * steps / dt
    g_ms, g_n = prof[dom]
    g_avg_s = g_ms / max(g_n, 1) * 1e-3
    # I = 0 when no index is read: the fused gather evaluates the keyed bijection in registers
    ib = 0 if (bijection and fused) else INDEX_BYTES
    dom_bytes = 20 if gae_only else alg_bytes_fused_gather(D, ib) if fused else alg_bytes_obs_gather(D)   # per transition, SURVEY.md 8(d)
    step_bytes = 20 if gae_only else alg_bytes_total(D, ib)
    achieved = M * dom_bytes / g_avg_s / 1e9 if g_n else 0.0
    kern = {k: {"avg_us": v[0] / v[1] * 1e3, "launches": v[1]} for k, v in prof_all.items() if v[1]}
    scan = "associative (chunked)" if algo & 1 else "sequential"
    workload = (f"{cfg}: T={T}, N={N} envs per GPU, GAE + returns only, {scan} scan, gamma=0.99 lambda=0.95" if gae_only else
                f"{cfg}: T={T}, N={N} envs per GPU, obs_dim={D}, batch={B} (K={K} minibatches/pass), "
                f"gamma=0.99 lambda=0.95, {scan} GAE scan + one shuffled pass")
    res = {
        "value": value, "unit": "transitions/s", "ms_per_step": ms_per_step, "steps": steps, "warmup": warmup,
        "scaling": "strong" if cfg == "C5" else "weak",
        "config": {"workload": workload, "global_envs": world * N,
                   "parallelism": f"env-sharded x{world}" if ctx.multi else "single GPU",
                   "exchange": ((exchange + "/" + args.backend) if ctx.multi else "none"),
                   "minibatches": "streamed one at a time into two slots" if buf.streaming else "all K gathered by one pass launch",
                   "gpu_spinup": (f"{ceiling['busy_ms']:.0f} ms of device copies (the copy_ceiling measurement) right before the {warmup} warm-up "
                                  f"and {steps} timed steps: an idle GPU needs ~0.2 s of load to reach steady clocks" if ceiling else "none")},
        "hbm_roofline_frac_whole_step": value * step_bytes / (world * HBM_PEAK_GBS * 1e9),
        "alg_bytes_per_transition": step_bytes,
        "roofline": {"bound": "hbm", "kernel": ("gae_scan_kernel" if gae_only else "fused_gather_loop_kernel" if gkind == 2 else
                                                    "fused_gather_kernel" if fused else "obs_gather_kernel"),
                     "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "alg_bytes_per_launch": M * dom_bytes, "avg_launch_us": g_avg_s * 1e6, "launches_timed": g_n},
        "kernels_untimed_breakdown": kern,
        "off_metric": {"insert_s": t_insert, "perm_s": t_perm,
                       "perm_note": "0: the permutation is generated inside the timed step" if bijection else "drawn before the timed region"},
        "perm_kind": (perm_kind if not (bijection and not fused and not gae_only) else
                      "keyed bijection, written out as an index array by one launch inside the timed step (split gather kernels read an array)"),
        "index_bytes": ib,
        "copy_ceiling": ceiling,
        "rccl_ranks": rccl_ranks,
        "rank_ms_per_step": {"min": min(rank_ms), "max": max(rank_ms)},
        "exchange_exposed_us": exch_us,
    }
    res.update(extra)
    return res, (T, N, D, K, gae_only)


def cited_traffic(config, kernel, perm):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/gather_traffic.json): a citation of
    a separate counter run of the same kernel on the same config, not a measurement of this run (counters need their own
    rocprofv3 passes)."""
    try:
        for tf in json.load(open(os.path.join(ROOT, "profiles", "gather_traffic.json"))):
            if tf.get("config") == config and tf.get("kernel") == kernel and tf.get("perm") in (perm, "none"):
                return {"traffic": tf.get("hbm_bytes_per_launch"), "traffic_source": tf.get("source")}
    except Exception:
        pass
    return {}


def copy_ceiling(dev, nbytes=1 << 30, min_ms=600.0):
    """The box's stream-copy rate (reported for context, the fraction stays against 8 TB/s): device-to-device
    copies of 1 GiB, read + write bytes counted, torch events on the current stream.  measure() runs it right before the warm-up
    steps, for at least `min_ms`: a GPU that has been idle (the set-up is host work and a few small launches) needs ~0.2 s of load to reach its
    steady clocks - the first 25 steps of a cold run take 873 us, steady state is 859 (tools/step_curve.py,
    profiles/r3/step_curve_cold_vs_warm.log) - so this measurement doubles as the spin-up; the W warm-up steps and the K
    timed steps that follow are the workload's own, exactly as asked for."""
    import torch
    src = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    src.zero_()
    for _ in range(3):
        dst.copy_(src)
    torch.cuda.synchronize()
    best, total_ms, reps = 0.0, 0.0, 0
    t_start = time.perf_counter()
    while (time.perf_counter() - t_start) * 1e3 < min_ms or reps < 10:
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            dst.copy_(src)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 10
        best = max(best, 2 * nbytes / (ms * 1e-3) / 1e9)
        total_ms += ms * 10
        reps += 10
    del src, dst
    return {"value": best, "unit": "GB/s", "copies": reps, "busy_ms": total_ms,
            "what": "Tensor.copy_ of 1 GiB on one device (the runtime's device-to-device copy), read + write bytes, best batch of 10; "
                    "measured before the timed runs, where it also brings the idle GPU to its steady clocks; a hand-written copy "
                    "kernel reaches ~6.3 TB/s on this part (profiles/r1 membench)"}


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
    ap.add_argument("--no-secondary", action="store_true", help="skip the other configs that a default run reports under 'also'")
    ap.add_argument("--exchange", default=os.environ.get("PPO_XP_EXCHANGE", "torch"), choices=["torch", "rccl"])
    ap.add_argument("--perm", default="bijection", choices=["bijection", "host", "device_sort"],
                    help="bijection: keyed bijection evaluated inside the gather, a fresh one per step inside the timed region (default); "
                         "host: rand-0.8.5 Fisher-Yates drawn before the timed region (reference-faithful); device_sort: SplitMix64 keys + radix sort")
    ap.add_argument("--backend", default=os.environ.get("PPO_XP_BACKEND", "nccl"), choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the multi-rank control flow with all ranks on one GPU")
    ap.add_argument("--tunable", action="append", default=[], help="name=value (see ppo_xp_set_tunable)")
    ap.add_argument("--streaming", action="store_true", help="gather minibatches one at a time into two slots (ppo_xp_set_streaming): "
                                                              "half the memory of a pass; not with --gpus > 1")
    ap.add_argument("--secondary-timeout", type=float, default=240.0, help="seconds the configs under 'also' may take before the headline is printed alone")
    ap.add_argument("--no-copy-ceiling", action="store_true", help="skip the device-copy rate measurement that precedes the timed runs")
    ap.add_argument("--spinup-ms", type=float, default=600.0, help="how long that measurement keeps the GPU busy (an idle GPU needs ~0.2 s of load to reach steady clocks)")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not record per-kernel HIP events in the timed region")
    args = ap.parse_args()

    # `python bench.py --gpus N` with no torch.distributed environment: start the N ranks from here,
    # before this process imports torch or makes any HIP call.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

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
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}: the process group decides (n_gpus = {world})")
    ndev = torch.cuda.device_count()
    if args.backend == "gloo":
        local_rank %= max(ndev, 1)
    elif local_rank >= ndev:
        log(f"bench.py: rank {rank} needs GPU {local_rank} but this node shows {ndev}; use --backend gloo to rehearse on fewer GPUs")
        sys.exit(3)
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
        world = dist.get_world_size()          # n_gpus in the result line = the ranks the process group really has
        rank = dist.get_rank()

    ctx = Ctx(args, rank, world, local_rank, dev, multi)
    res, (T, N, D, K, gae_only) = measure(ctx, args.config, args.steps, args.warmup, gae=args.gae,
                                          events=not args.no_kernel_events, tunables=args.tunable)

    def headline(res):
        """The contract's line from one measure() result of the headline config."""
        res = dict(res)
        try:
            metric = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
        except Exception:
            metric = "transitions/sec through GAE+norm+minibatch gather; % of HBM roofline, 1–8 GPU"
        line = {"metric": metric, "value": res.pop("value"), "unit": res.pop("unit"), "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": res.pop("ms_per_step"), "higher_is_better": True,
                "scaling": res.pop("scaling"), "vs_baseline": None, "dtype": "f32", "data": "synthetic"}
        res.pop("steps"); res.pop("warmup")
        line.update(res)
        line["roofline"].update(cited_traffic(args.config, line["roofline"]["kernel"], args.perm))
        return line

    out = headline(res) if rank == 0 else None

    # ---- the other BASELINE.json configs beside the headline, measured the same way ---------------
    if args.config == "TGT" and not args.no_secondary:
        also = {}
        # The headline is measured; nothing after it may cost it.  A secondary leg that raises is caught below; one that
        # HANGS (a collective some rank never enters) cannot be caught, so every rank arms the same timer: when it fires,
        # rank 0 prints the line with what it has and every rank leaves.
        import threading

        def _bail(why=None):
            why = why or f"secondary configs did not finish within {args.secondary_timeout} s"
            log(f"bench.py: rank {rank}: {why}; reporting the headline alone")
            if rank == 0 and out is not None:
                out["also"] = dict(also, error=why)
                os.write(json_fd, (json.dumps(out) + "\n").encode())
            os._exit(0)
        watchdog = threading.Timer(args.secondary_timeout, _bail)
        watchdog.daemon = True
        watchdog.start()
        # one GPU: configs[1] C2, configs[2] C3, configs[3] C4 (GAE only, K2 + its error against K1) and C5 whole on
        # this GPU (= the per-GPU shard of a 1-rank job); N GPUs: C5 split over the ranks (strong scaling)
        todo = [("C5", 10, 3, None)] if world > 1 else [("C3", 12, 3, None), ("C5", 10, 3, None), ("C2", 200, 20, None), ("C4", 100, 10, None)]
        if multi:
            # the headline ran with --exchange (default: the caller all-reduces through torch.distributed); measure TGT again with
            # the other one, so that the library's own ncclAllReduce (exchange mode 1) executes whenever there are ranks
            other = "rccl" if args.exchange == "torch" else "torch"
            if not (other == "rccl" and args.backend != "nccl"):
                todo.insert(0, ("TGT", args.steps, args.warmup, other))
        for cfg, st, wu, xch in todo:
            try:
                if xch is not None:       # the headline config under the other exchange: measured exactly like the headline
                    r2, _ = measure(ctx, cfg, st, wu, gae=args.gae, events=not args.no_kernel_events, tunables=args.tunable, exchange=xch)
                    also[f"{cfg}_exchange_{xch}"] = r2
                    continue
                r2, _ = measure(ctx, cfg, st, wu, events=False, breakdown=(cfg != "C2"), exchange=xch)
                r2.pop("roofline", None)
                tr = cited_traffic(cfg, "gae_scan_kernel" if cfg == "C4" else "obs_gather_kernel", args.perm)
                if tr:
                    r2["dominant_kernel_traffic"] = tr
                also[cfg if xch is None else f"{cfg}_exchange_{xch}"] = r2
            except Exception as e:
                # a secondary line must never cost the headline.  Several ranks: the others may sit inside a collective this
                # rank never reaches - stop the secondaries here and let the timer release whoever is stuck
                log(f"bench.py: secondary config {cfg} failed on rank {rank}: {type(e).__name__}: {e}")
                also[cfg if xch is None else f"{cfg}_exchange_{xch}"] = {"error": f"{type(e).__name__}: {e}"}
                torch.cuda.empty_cache()
                if world > 1:
                    watchdog.cancel()
                    _bail(f"secondary config {cfg} failed on rank {rank}")     # leaves; the other ranks' timers release them
        watchdog.cancel()
        if out is not None:
            # Two implementations of the one exchange were timed on the same workload in this run (the caller's torch.distributed
            # all-reduce and the library's own ncclAllReduce on the pass's stream): the line reports the faster one and keeps
            # the other under "also"; config.exchange names which is which.
            for key in [k for k in also if k.startswith("TGT_exchange_")]:
                r2 = also[key]
                if "ms_per_step" in r2 and r2["ms_per_step"] < out["ms_per_step"]:
                    loser = {k: v for k, v in out.items() if k not in ("metric", "n_gpus", "higher_is_better", "vs_baseline", "dtype", "data")}
                    out = headline(r2)
                    del also[key]
                    also["TGT_exchange_" + loser["config"]["exchange"].split("/")[0]] = loser
                    out["headline_choice"] = "the faster of the two exchanges timed in this run; the other one is under also." \
                                             "TGT_exchange_" + loser["config"]["exchange"].split("/")[0]
                else:
                    r2.pop("roofline", None)
            out["also"] = also

    if multi:
        dist.destroy_process_group()
    if rank == 0:
        if out.get("copy_ceiling") is None:
            out.pop("copy_ceiling", None)
        if not args.no_cpu_baseline and world == 1 and not gae_only:
            out["cpu_baseline"] = cpu_baseline(T, N, D, K)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
