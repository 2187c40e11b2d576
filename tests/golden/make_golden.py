"""Generates tests/golden/c1_path.npz and c_small_path.npz from the CPU oracle.

The reference (Rust) cannot be run here, so these are oracle outputs on seeded inputs: they cover
what no reference test pins (done-flag masking, the last_values_idx bootstrap, adv' double
rounding, gather order, standardisation) and guard against silent drift of the oracle itself.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from ppo_amd import synth  # noqa: E402


def make(path, T, N, D, B, seed, p_done):
    r = synth.rollout(T, N, D, seed, p_done)
    M = T * N
    streams = dict(obs=r["obs"].reshape(M, D), rewards=r["rewards"].reshape(-1), actions=r["actions"].reshape(-1),
                   values=r["values"].reshape(-1), neglogps=r["neglogps"].reshape(-1),
                   dones=r["dones"].astype(np.float32).reshape(-1))
    perm = O.shuffle(seed, M)
    out = O.run_pass(streams, T, N, D, T - 1, r["last_dones"], 0.99, 0.95, perm, B)
    np.savez_compressed(path, dims=np.array([T, N, D, B]), seed=np.uint64(seed), perm=perm,
                        last_dones=r["last_dones"], **{k: r[k] for k in ("obs", "rewards", "actions", "values", "neglogps", "dones")},
                        **{"out_" + k: out[k] for k in ("returns", "adv", "obs", "actions", "values", "neglogps", "ret", "advn")})
    print(path, os.path.getsize(path), "bytes, K =", out["K"])


def make_bijection(path):
    """Vectors of the keyed bijection (ppo_xp_pass_begin_bijection; no counterpart in the reference, whose shuffle is
    OS-seeded): whole permutations for small M, heads and CRC-32 checksums of the whole array for large M.  Written from the
    oracle's C twin; the numpy twin and the library are compared against the same file."""
    import json
    import zlib
    cases = []
    for seed, M in [(0, 1), (1, 2), (7, 5), (0x5EED, 257), (2**64 - 1, 1000), (42, 65536), (99, 1 << 23), (123456789, 1010707)]:
        p = O.bijection_perm(seed, M)
        q = O.bijection_perm(seed, M, inverse=True)
        cases.append({"seed": str(seed), "M": M, "head": p[:16].tolist(), "tail": p[-4:].tolist(),
                      "crc32": zlib.crc32(p.tobytes()), "inverse_head": q[:8].tolist(),
                      "all": p.tolist() if M <= 1000 else None})
    json.dump({"spec": "ppo_amd/csrc/bijection.h", "indices": "uint32 little endian", "cases": cases}, open(path, "w"))
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    make_bijection(os.path.join(here, "bijection_kats.json"))
    T, N, D, B = synth.CONFIGS["C1"]
    make(os.path.join(here, "c1_path.npz"), T, N, D, B, synth.SEEDS["C1"], 0.02)
    # ragged: N not a multiple of 64, D = 3 (scalar gather path), B does not divide M (overlapping tail)
    make(os.path.join(here, "c_small_path.npz"), 17, 70, 3, 500, 0x5EED + 100, 0.1)
