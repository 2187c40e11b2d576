"""Pins the CPU oracle against every vector the reference's own tests hold for the path
(tests/golden/reference_kats.json: data from ppo/src/common/buffer.rs:233-458 and
ppo/src/model/data.rs:210-270) and against a second, independent numpy restatement."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
KATS = json.load(open(os.path.join(HERE, "golden", "reference_kats.json")))


def _fill(buf, kat):
    for name, times in kat["inserts"]:
        s = KATS["steps"][name]
        for _ in range(times):
            buf.add_experience(s["obs"], s["rewards"], s["actions"], s["vals"], s["dones"], s["neglogps"])


def _new(oracle, kat):
    return oracle.OracleBuffer(kat["num_envs"], kat["obs_size"], kat["capacity"])


@pytest.mark.parametrize("name", ["exp_buffer_below_capacity", "exp_buffer_at_capacity", "exp_buffer_over_capacity"])
def test_views_exact(oracle, name):
    kat = KATS[name]
    b = _new(oracle, kat)
    _fill(b, kat)
    obs, actions, values, neglogps = b.training_views()
    assert obs.shape == (len(kat["obs"]), kat["obs_size"])
    np.testing.assert_array_equal(obs, np.array(kat["obs"], np.float32))
    np.testing.assert_array_equal(actions, np.array(kat["actions"], np.uint64))
    np.testing.assert_array_equal(values, np.array(kat["values"], np.float32))
    np.testing.assert_array_equal(neglogps, np.array(kat["neglogps"], np.float32))
    if "returns_len" in kat:
        assert b.returns(kat["last_dones"]).shape == (kat["returns_len"],)


def test_counter_wrap(oracle):
    # buffer.rs:117-120: the counter stays in [cap, 2cap) once full
    kat = KATS["exp_buffer_over_capacity"]
    b = _new(oracle, kat)
    _fill(b, kat)
    assert len(b) == 3 and b.last_values_idx == (1 + 1366 + 3 - 1) % 3


def test_returns_sanity_intervals(oracle):
    kat = KATS["exp_buffer_returns_sanity"]
    b = _new(oracle, kat)
    _fill(b, kat)
    ret = b.returns(kat["last_dones"])
    for r, (lo, hi) in zip(ret, kat["returns_open_intervals"]):
        assert lo < r < hi
    obs, _, _, _ = b.training_views()
    assert obs.shape[0] == ret.shape[0]
    # exact fp32 bit patterns recorded in SURVEY.md section 4
    want = np.array([int(h, 16) for h in KATS["survey_fp32_restatement"]["returns_hex_last_dones_true"]], np.uint32)
    np.testing.assert_array_equal(ret.view(np.uint32), want)
    ret2 = b.returns([False, False])  # exercises the last_values_idx bootstrap (buffer.rs:204)
    np.testing.assert_allclose(ret2, KATS["survey_fp32_restatement"]["returns_last_dones_false"], rtol=0, atol=5e-7)


def test_batcher_sanity(oracle):
    kat = KATS["batcher_sanity"]
    b = oracle.OracleBuffer(kat["num_envs"], kat["obs_size"], kat["capacity"])
    for i in range(kat["steps"]):
        b.add_experience([[0.0 + i, 1.0, 2.0], [100.0 + i, 2.0, 3.0]], kat["rewards"], kat["actions"], kat["vals"],
                         kat["dones"], kat["neglogps"])
    obs, actions, values, neglogps = b.training_views()
    ret = b.returns(kat["last_dones"])
    M, B = obs.shape[0], kat["batch_size"]
    assert M == 10
    windows = oracle.batch_windows(M, B)
    assert len(windows) == kat["num_iterations"] == oracle.num_batches(M, B)
    assert windows == [(0, 6), (4, 10)]  # data.rs:121-125: the tail window overlaps
    perm = oracle.shuffle(7, M)
    seen = set()
    for s, e in windows:
        rows = oracle.gather_f32(obs, perm[s:e])
        seen |= {int(round(float(x))) for x in rows[:, 0]}
    assert seen == set(kat["seen_rounded_obs0"])
    assert ret.shape == (M,)


def test_windows_edge_cases(oracle):
    assert oracle.batch_windows(10, 11) == []          # data.rs:117: batch_size > buf_size -> None
    assert oracle.batch_windows(10, 10) == [(0, 10)]
    assert oracle.batch_windows(10, 5) == [(0, 5), (5, 10)]
    assert oracle.batch_windows(10, 3) == [(0, 3), (3, 6), (6, 9), (7, 10)]
    assert oracle.batch_windows(0, 3) == []
    assert oracle.num_batches(10, 3) == 4 and oracle.num_batches(10, 11) == 0


@pytest.mark.parametrize("T,N,seed", [(1, 1, 0), (3, 2, 1), (128, 8, 2), (33, 70, 3), (200, 5, 4)])
def test_c_oracle_equals_numpy_twin(oracle, T, N, seed):
    g = np.random.default_rng(seed)
    r, v = g.standard_normal((T, N), dtype=np.float32), g.standard_normal((T, N), dtype=np.float32)
    d = (g.random((T, N)) < 0.1).astype(np.float32)
    ld = g.random(N) < 0.3
    lvi = int(g.integers(0, T))
    a = oracle.gae_rows(r, v, d, lvi, ld, 0.99, 0.95)
    b = oracle.np_gae(r, v, d, lvi, ld, 0.99, 0.95)
    np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))
    # and the f64 companion agrees to fp32 rounding noise
    c = oracle.gae_rows_f64(r, v, d, lvi, ld, 0.99, 0.95)
    np.testing.assert_allclose(a, c, rtol=0, atol=2e-5)


def test_double_rounding_of_advantages(oracle):
    # data.rs:62: adv' = fl(fl(A + v) - v) differs from A in fp32 for some inputs
    g = np.random.default_rng(5)
    r, v = g.standard_normal((64, 32), dtype=np.float32), (100 * g.standard_normal((64, 32))).astype(np.float32)
    d = np.zeros((64, 32), np.float32)
    ret = oracle.gae_rows(r, v, d, 63, np.zeros(32, bool), 0.99, 0.95)
    adv = oracle.advantages(ret, v.reshape(-1))
    np.testing.assert_array_equal(adv, ret - v.reshape(-1))


def test_standardize_semantics(oracle):
    a = np.arange(5, dtype=np.float32)
    out, mean, std = oracle.standardize(a)
    assert mean == 2.0 and abs(std - np.sqrt(2.5)) < 1e-6  # unbiased std (data.rs:196-200)
    np.testing.assert_allclose(out, (a - 2.0) / (np.float32(np.sqrt(2.5)) + 1e-8), rtol=1e-6)
    out1, _, _ = oracle.standardize(np.array([3.0], np.float32))
    assert np.isnan(out1).all()  # n = 1: var * (1/0) = 0 * inf = NaN, as in the reference
    g = np.random.default_rng(0)
    x = g.standard_normal(4096).astype(np.float32) * 3 + 1
    o32, _, _ = oracle.standardize(x)
    o64, _, _ = oracle.standardize_f64(x)
    assert np.abs(o32 - o64).max() < 1e-4


def test_chacha_core_published_vectors(oracle):
    z = np.zeros(8, np.uint32)
    pv = KATS["chacha_published_vectors"]
    assert oracle.chacha_block(z, 0, 0, 20).tobytes().hex() == pv["rounds20"]
    assert oracle.chacha_block(z, 0, 0, 12).tobytes().hex() == pv["rounds12"]


def test_shuffle_is_permutation_and_seeded(oracle):
    for M in (0, 1, 2, 10, 1000, 65537):
        p = oracle.shuffle(42, M)
        assert np.array_equal(np.sort(p), np.arange(M, dtype=np.uint32))
    assert not np.array_equal(oracle.shuffle(1, 1000), oracle.shuffle(2, 1000))
    assert np.array_equal(oracle.shuffle(1, 1000), oracle.shuffle(1, 1000))
    # stream = consecutive ChaCha12 blocks of the seed_from_u64 key
    key = oracle.seed_from_u64(9)
    s = oracle.rng_stream(9, 40)
    np.testing.assert_array_equal(s[:16], oracle.chacha_block(key, 0, 0, 12))
    np.testing.assert_array_equal(s[16:32], oracle.chacha_block(key, 1, 0, 12))


def test_shuffle_python_restatement(oracle):
    """Independent pure-Python walk of rand 0.8.5's shuffle over the oracle's u32 stream."""
    M, seed = 257, 1234
    stream = iter(oracle.rng_stream(seed, 4 * M).tolist())
    idx = list(range(M))
    for i in range(M - 1, 0, -1):
        rng_range = i + 1
        zone = ((rng_range << (32 - rng_range.bit_length())) & 0xFFFFFFFF) - 1
        while True:
            m = next(stream) * rng_range
            if (m & 0xFFFFFFFF) <= zone:
                j = m >> 32
                break
        idx[i], idx[j] = idx[j], idx[i]
    assert idx == oracle.shuffle(seed, M).tolist()


def test_golden_fixtures_match_oracle(oracle):
    """The committed oracle-generated fixtures (tests/golden/make_golden.py) still reproduce."""
    path = os.path.join(HERE, "golden", "c1_path.npz")
    z = np.load(path)
    T, N, D, B = (int(x) for x in z["dims"])
    streams = {k: np.ascontiguousarray(z[k].reshape(T * N, -1) if k == "obs" else z[k].reshape(-1))
               for k in ("obs", "rewards", "actions", "values", "neglogps")}
    streams["dones"] = z["dones"].astype(np.float32).reshape(-1)
    perm = oracle.shuffle(int(z["seed"]), T * N)
    np.testing.assert_array_equal(perm, z["perm"])
    out = oracle.run_pass(streams, T, N, D, T - 1, z["last_dones"], 0.99, 0.95, perm, B)
    for k in ("returns", "adv", "obs", "actions", "values", "neglogps", "ret", "advn"):
        np.testing.assert_array_equal(out[k].view(np.uint8), z["out_" + k].view(np.uint8), err_msg=k)


def test_keyed_shuffle_twins_agree(oracle):
    for M in (0, 1, 2, 1000, 70001):
        a, b = oracle.keyed_shuffle(99, M), oracle.np_keyed_shuffle(99, M)
        np.testing.assert_array_equal(a, b)
        assert np.array_equal(np.sort(a), np.arange(M, dtype=np.uint32))
    assert not np.array_equal(oracle.keyed_shuffle(1, 1000), oracle.keyed_shuffle(2, 1000))


def test_explained_variance_restatement(oracle):
    g = np.random.default_rng(3)
    ret = g.standard_normal(5000).astype(np.float32) * 2 + 1
    val = ret + g.standard_normal(5000).astype(np.float32) * 0.5
    ev = oracle.explained_variance(val, ret)
    ev64 = oracle.explained_variance_f64(val, ret)
    assert abs(ev - ev64) < 1e-5
    # training.rs:97 is `0.0 - ratio` (not `1.0 - ratio`): a perfect predictor scores -0.0, not 1.0
    assert oracle.explained_variance(ret, ret) == 0.0
    want = -np.var((ret - val).astype(np.float64)) / np.var(ret.astype(np.float64))
    assert abs(ev64 - want) < 1e-9
