"""The keyed bijection of ppo_xp_pass_begin_bijection on the CPU: the oracle's two twins (C and numpy, written
independently from the specification in ppo_amd/csrc/bijection.h) agree, the map is a bijection with the stated
inverse, and it looks uniform over seeds.  No reference vector can exist: the reference's shuffle is OS-seeded
(ppo/src/model/data.rs:94) - parity unpinned for the specific permutation, as for the host Fisher-Yates."""
import numpy as np
import pytest

from oracle import oracle as O


@pytest.mark.parametrize("M", [1, 2, 3, 7, 255, 256, 257, 1000, 4097, 65536, 1_000_003])
def test_twins_agree_and_invert(M):
    for seed in (0, 5, 0xFFFFFFFFFFFFFFFF):
        p = O.bijection_perm(seed, M)
        np.testing.assert_array_equal(p, O.np_bijection_perm(seed, M))
        np.testing.assert_array_equal(np.sort(p), np.arange(M, dtype=np.uint32))
        q = O.bijection_perm(seed, M, inverse=True)
        np.testing.assert_array_equal(q, O.np_bijection_perm(seed, M, inverse=True))
        np.testing.assert_array_equal(q[p], np.arange(M, dtype=np.uint32))


def test_range_equals_slice():
    M = 300_001
    p = O.bijection_perm(9, M)
    np.testing.assert_array_equal(O.bijection_range(9, M, 123_456, 1000), p[123_456:124_456])
    np.testing.assert_array_equal(O.bijection_range(9, M, 0, 0), p[:0])


def test_known_values():
    """Pinned against this repository's own first evaluation (three independent implementations agreed on it:
    C, numpy, and the prototype the constants were chosen with)."""
    np.testing.assert_array_equal(O.bijection_perm(99, 1 << 23)[:8],
                                  [941447, 3208123, 4838812, 7693373, 5999904, 1791838, 7768909, 857436])


@pytest.mark.parametrize("M", [2, 3, 5, 6, 8, 13, 33, 100])
def test_first_order_uniformity_over_seeds(M):
    """P[perm(j) = v] = 1/M over seeds: chi-square over the M x M table, (M-1)^2 degrees of freedom."""
    S = 20000
    cnt = np.zeros((M, M))
    j = np.arange(M)
    for s in range(S):
        cnt[j, O.bijection_perm(s * 7919 + 1, M)] += 1
    exp = S / M
    chi2 = ((cnt - exp) ** 2 / exp).sum()
    dof = (M - 1) ** 2
    assert chi2 < dof + 6 * np.sqrt(2 * dof) + 6, (chi2, dof)


def test_small_domains_reach_every_permutation_evenly():
    for M, fact in ((3, 6), (4, 24), (5, 120)):
        seen = {}
        S = 400 * fact
        for s in range(S):
            t = tuple(O.bijection_perm(s + 77, M))
            seen[t] = seen.get(t, 0) + 1
        v = np.array(list(seen.values()))
        assert len(seen) == fact
        exp = S / fact
        assert ((v - exp) ** 2 / exp).sum() < (fact - 1) + 6 * np.sqrt(2 * (fact - 1))


def test_large_permutation_looks_shuffled():
    M, N, B = 1 << 21, 1 << 14, 1 << 16
    p = O.bijection_perm(4242, M).astype(np.int64)
    assert abs(np.corrcoef(np.arange(M), p)[0, 1]) < 5e-3
    assert abs(np.abs(np.diff(p)).mean() / M - 1 / 3) < 2e-3             # E|U1 - U2| = 1/3
    # every minibatch draws time steps, env blocks and low index bits evenly
    for name, key, bins in (("t", p // N, M // N), ("env", (p % N) >> 6, N >> 6), ("low", p & 255, 256)):
        hist = np.stack([np.bincount(row, minlength=bins) for row in key.reshape(-1, B)])
        exp = B / bins
        dof = hist.shape[0] * (bins - 1)
        chi2 = ((hist - exp) ** 2 / exp).sum()
        assert abs(chi2 - dof) < 6 * np.sqrt(2 * dof), (name, chi2, dof)
    # neighbours in the output are independent: (perm(j) mod 16, perm(j+1) mod 16) table
    tab = np.zeros((16, 16))
    np.add.at(tab, (p[:-1] & 15, p[1:] & 15), 1)
    exp = (M - 1) / 256
    assert abs(((tab - exp) ** 2 / exp).sum() - 255) < 6 * np.sqrt(2 * 255)


def test_committed_vectors():
    """tests/golden/bijection_kats.json (written by tests/golden/make_golden.py from the C twin): both twins still give them."""
    import json
    import os
    import zlib
    kats = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bijection_kats.json")))
    for c in kats["cases"]:
        seed, M = int(c["seed"]), c["M"]
        for fn in (O.bijection_perm, O.np_bijection_perm):
            p = fn(seed, M)
            assert p[:16].tolist() == c["head"] and p[-4:].tolist() == c["tail"] and zlib.crc32(p.tobytes()) == c["crc32"]
            if c["all"] is not None:
                assert p.tolist() == c["all"]
        assert O.bijection_perm(seed, M, inverse=True)[:8].tolist() == c["inverse_head"]
