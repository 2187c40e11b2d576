"""Full-size configurations of BASELINE.json on the GPU, checked through size-independent
properties (the oracle would need minutes and tens of GB at these sizes):
  * C2 (T=128, N=4096, D=64) is still small enough for a full bit-level oracle comparison;
  * TGT (T=128, N=65536, D=64) and C4 (T=16384, N=1024): returns against an op-for-op torch fp32
    restatement of the recurrence on the device (eager torch rounds every op, like ndarray);
    gathered rows == source[perm] by device indexing; permutation is a bijection; standardised
    advantages against an fp64 computation on the device; per-minibatch mean 0 / unbiased std 1."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def torch_gae(rewards, values, dones, lvi, last_dones, gamma, lam):
    """buffer.rs:199-222 with one torch kernel per reference op (fp32, no fusion, no FMA)."""
    import torch
    T, N = rewards.shape
    g = torch.tensor(gamma, dtype=torch.float32, device=rewards.device)
    gl = torch.tensor(np.float32(gamma) * np.float32(lam), dtype=torch.float32, device=rewards.device)
    nonterm = 1.0 - dones
    last_nt = (last_dones == 0).to(torch.float32)
    A = torch.zeros(N, dtype=torch.float32, device=rewards.device)
    ret = torch.empty_like(rewards)
    for t in range(T - 1, -1, -1):
        nv, nnt = (values[lvi], last_nt) if t == T - 1 else (values[t + 1], nonterm[t + 1])
        delta = (rewards[t] + (nv * nnt) * g) - values[t]
        A = delta + (nnt * A) * gl
        ret[t] = A + values[t]
    return ret.reshape(-1)


def device_fill(buf, T, N, D, seed):
    import torch
    from ppo_amd import synth
    r = synth.rollout_device(T, N, D, seed, torch.device("cuda", 0))
    for t in range(T):
        buf.add_experience_device(r["obs"][t], r["rewards"][t], r["actions"][t], r["values"][t], r["dones"][t], r["neglogps"][t])
    buf.sync()
    return r


def test_c2_full_oracle(oracle):
    import ppo_amd
    from ppo_amd import synth
    T, N, D, B = synth.CONFIGS["C2"]
    r = synth.rollout(T, N, D, synth.SEEDS["C2"])
    buf = ppo_amd.ExperienceBuffer(N, D, T)
    buf.add_many(r["obs"], r["rewards"], r["actions"], r["values"], r["dones"], r["neglogps"])
    ret = buf.returns(r["last_dones"], to_host=True)
    want = oracle.gae_rows(r["rewards"], r["values"], r["dones"].astype(np.float32), T - 1, r["last_dones"], 0.99, 0.95)
    assert np.array_equal(ret.view(np.uint32), want.view(np.uint32))
    M = T * N
    it = ppo_amd.ExperienceBatcher(buf, B).into_iter(seed=synth.SEEDS["C2"])
    perm = it.perm()
    assert np.array_equal(perm, oracle.shuffle(synth.SEEDS["C2"], M))
    obs = r["obs"].reshape(M, D)
    adv = oracle.advantages(want, r["values"].reshape(-1))
    worst32 = worst64 = 0.0
    for k, b in enumerate(it):
        idx = perm[k * B:(k + 1) * B]
        assert np.array_equal(b.observations.cpu().numpy(), obs[idx])
        assert np.array_equal(b.actions.cpu().numpy().astype(np.uint64), r["actions"].reshape(-1)[idx])
        assert np.array_equal(b.returns.cpu().numpy(), want[idx])
        got = b.advantages.cpu().numpy()
        w32, _, _ = oracle.standardize(adv[idx])
        w64, _, _ = oracle.standardize_f64(adv[idx])
        worst32 = max(worst32, float(np.abs(got - w32).max()))
        worst64 = max(worst64, float(np.abs(got - w64).max()))
    assert k == M // B - 1
    assert worst64 < 2e-6 and worst32 < 5e-5, (worst32, worst64)
    buf.close()


# |GPU - fp32-sequential restatement| allowed on a standardised advantage, by minibatch size.  The kernel sums in
# fp64 (its distance to an fp64 evaluation stays < 2e-6 at every size); the reference's dfdx Cpu reduction is
# restated as a SEQUENTIAL fp32 accumulate (order unpinned), whose own rounding error grows with
# B: measured 2.6e-6 (B = 16 384, C2) ... see DESIGN.md section 2 for the full-size numbers.
SEQ32_BOUND = {"TGT": 1e-3, "C3": 5e-3, "C5": 2e-3}


@pytest.mark.parametrize("cfg", ["TGT", "C3", "C5/8"])
def test_fullsize_properties(cfg, oracle):
    """TGT; C3 (34 GB of observations, 1-KiB rows, the multi-wave scan); the per-GPU shard of C5 on 8 GPUs."""
    import torch
    import ppo_amd
    from ppo_amd import synth
    if cfg == "C5/8":
        T, N, D, B = synth.CONFIGS["C5"]
        N, B, cfg = N // 8, B // 8, "C5"
    else:
        T, N, D, B = synth.CONFIGS[cfg]
    M = T * N
    buf = ppo_amd.ExperienceBuffer(N, D, T)
    r = device_fill(buf, T, N, D, synth.SEEDS[cfg])
    ld = r["last_dones"].cpu().numpy()
    ret = buf.returns(ld)
    want = torch_gae(r["rewards"], r["values"], r["dones"], T - 1, r["last_dones"], 0.99, 0.95)
    assert torch.equal(ret.view(torch.int32), want.view(torch.int32))          # bit-exact at full size
    adv = buf.advantages()
    assert torch.equal(adv, ret - r["values"].reshape(-1))
    it = ppo_amd.ExperienceBatcher(buf, B).into_iter(seed=11)
    perm = ppo_amd.device_tensor(ppo_amd._lib.load().ppo_xp_pass_perm_device(buf.handle), (M,), torch.uint32, 0, buf).long()
    assert torch.equal(torch.sort(perm).values, torch.arange(M, device=perm.device))   # a bijection of 0..M
    obs = r["obs"].reshape(M, D)
    streams = {f: r[f].reshape(-1) for f in ("actions", "values", "neglogps")}
    n = 0
    for k, b in enumerate(it):
        idx = perm[k * B:(k + 1) * B]
        assert torch.equal(b.observations, obs[idx])
        assert torch.equal(b.actions, streams["actions"][idx])
        assert torch.equal(b.values, streams["values"][idx])
        assert torch.equal(b.neglogps, streams["neglogps"][idx])
        assert torch.equal(b.returns, ret[idx])
        a = adv[idx].double()
        ref = (a - a.mean()) / (a.std(unbiased=True) + 1e-8)
        assert (b.advantages.double() - ref).abs().max().item() < 2e-6
        assert abs(b.advantages.double().mean().item()) < 1e-6
        assert abs(b.advantages.double().std(unbiased=True).item() - 1.0) < 1e-5
        if k in (0, M // B - 1):
            # what the reference itself would print for this minibatch, as far as it can be restated: the oracle's
            # fp32-sequential standardisation (data.rs:157-164, 196-200) of the same gathered advantages, on the host
            w32, _, _ = oracle.standardize(adv[idx].cpu().numpy())
            d32 = float(np.abs(b.advantages.cpu().numpy() - w32).max())
            print(f"{cfg}: minibatch {k} of {B} rows: max |GPU - fp32-sequential restatement| = {d32:.3e}")
            assert d32 < SEQ32_BOUND[cfg], d32
        n += 1
    assert n == M // B
    buf.close()
    del r, obs, streams, ret, adv, want, perm, it, b
    torch.cuda.empty_cache()


def test_c4_long_horizon_assoc_vs_sequential():
    """C4: T=16384, N=1024.  Sequential kernel bit-exact vs the torch restatement; associative
    scan within the stated bound of the sequential result and of an fp64 evaluation."""
    import torch
    import ppo_amd
    from ppo_amd import synth
    T, N, D, _ = synth.CONFIGS["C4"]
    buf = ppo_amd.ExperienceBuffer(N, D, T)
    r = device_fill(buf, T, N, D, synth.SEEDS["C4"])
    ld = r["last_dones"].cpu().numpy()
    want = torch_gae(r["rewards"], r["values"], r["dones"], T - 1, r["last_dones"], 0.99, 0.95)
    seq = buf.returns(ld).clone()
    assert torch.equal(seq.view(torch.int32), want.view(torch.int32))
    assoc = buf.returns(ld, algo=ppo_amd.GAE_ASSOC).clone()
    scale = torch.clamp(seq.abs(), min=1.0)
    rel = ((assoc - seq).abs() / scale).max().item()
    # fp64 evaluation of the same recurrence
    A = torch.zeros(N, dtype=torch.float64, device="cuda")
    ret64 = torch.empty((T, N), dtype=torch.float64, device="cuda")
    rw, v, d = r["rewards"].double(), r["values"].double(), r["dones"].double()
    lnt = (r["last_dones"] == 0).double()
    for t in range(T - 1, -1, -1):
        nv, nnt = (v[T - 1], lnt) if t == T - 1 else (v[t + 1], 1.0 - d[t + 1])
        A = rw[t] + nv * nnt * 0.99 - v[t] + nnt * A * (0.99 * 0.95)
        ret64[t] = A + v[t]
    ret64 = ret64.reshape(-1)
    s64 = torch.clamp(ret64.abs(), min=1.0)
    rel_seq64 = ((seq.double() - ret64).abs() / s64).max().item()
    rel_assoc64 = ((assoc.double() - ret64).abs() / s64).max().item()
    print(f"C4 assoc vs sequential max rel {rel:.3e}; vs fp64: sequential {rel_seq64:.3e}, assoc {rel_assoc64:.3e}")
    assert rel <= 1e-5 and rel_assoc64 <= 1e-5
    buf.close()


def test_tgt_pass_is_deterministic_and_every_gather_route_agrees():
    """TGT in full: the same rollout and permutation give the same bits on a second run (no atomics, fixed-order
    statistics folds), and the fused gather, the two-kernel path, the two-kernel path with the rows on a side stream
    and the streamed pass (one minibatch per launch set) all produce identical minibatches."""
    import torch
    import ppo_amd
    from ppo_amd import synth
    T, N, D, B = synth.CONFIGS["TGT"]
    M = T * N
    buf = ppo_amd.ExperienceBuffer(N, D, T)
    r = device_fill(buf, T, N, D, 31)
    ld = r["last_dones"].cpu().numpy()
    del r
    perm = torch.randperm(M, device="cuda").to(torch.int32)
    fields = ("observations", "actions", "values", "neglogps", "returns", "advantages")

    def run():
        buf.returns(ld)
        sig = []
        for b in ppo_amd.ExperienceBatcher(buf, B).into_iter(perm=perm):
            # a 64-bit checksum per array would do, but the arrays are small next to the rows: keep the scalars
            # whole and fold the rows to one int64 sum per minibatch (order-independent integer sum: exact)
            sig.append([getattr(b, f).clone() for f in fields[1:]] + [b.observations.view(torch.int32).to(torch.int64).sum()])
        torch.cuda.synchronize()
        return sig

    def same(a, b):
        return len(a) == len(b) and all(torch.equal(x.view(torch.int32) if x.dtype == torch.float32 else x, y.view(torch.int32) if y.dtype == torch.float32 else y)
                                        for ma, mb in zip(a, b) for x, y in zip(ma, mb))

    base = run()
    assert len(base) == M // B
    assert same(base, run()), "second run of the same pass differs"
    for tun in ({"fused": 0}, {"fused": 0, "overlap": 1}):
        for k, v in tun.items():
            buf.set_tunable(k, v)
        assert same(base, run()), f"route {tun} differs from the fused gather"
    buf.set_tunable("fused", -1)
    buf.set_tunable("overlap", 0)
    buf.set_streaming(True)
    assert same(base, run()), "streamed pass differs"
    buf.close()
