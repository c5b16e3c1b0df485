"""GPU parity of fast_levenshtein_ratio / construct_features (through the C ABI) against oracle and goldens."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _encode(kat, text):
    table = {ch: i for i, ch in enumerate(kat["alphabet"])}
    return np.array([table[ch] for ch in text], dtype=np.uint8)


def test_levenshtein_known_answers(golden_kat):
    import doppel_speller_amd as ds
    a = [_encode(golden_kat, case["a"]) for case in golden_kat["levenshtein"]]
    b = [_encode(golden_kat, case["b"]) for case in golden_kat["levenshtein"]]
    expected = np.array([case["ratio"] for case in golden_kat["levenshtein"]], dtype=np.uint8)
    for method in (0, 1):
        assert np.array_equal(ds.levenshtein_ratio_batch(a, b, method), expected)
        assert np.array_equal(ds.levenshtein_ratio_batch(b, a, method), expected)


def test_levenshtein_every_ratio_value(oracle):
    """All (LCS, L) combinations with L <= 255: pins the float64 expression of feature_engineering.py:63,
    including the four fastmath hazard points of SURVEY.md H4 (strict IEEE order is the specification)."""
    import doppel_speller_amd as ds
    a, b = [], []
    for la in range(0, 128):
        for lb in (la, min(255 - la, la + 1), min(255 - la, 2 * la + 3)):
            for common in sorted({0, la // 3, la // 2, la}):
                a.append(np.array([2] * common + [3] * (la - common), dtype=np.uint8))
                b.append(np.array([2] * common + [4] * (lb - common), dtype=np.uint8))
    a += [np.array([2] * 29 + [3] * 21, dtype=np.uint8), np.array([2] * 58 + [3] * 42, dtype=np.uint8)]
    b += [np.array([2] * 29 + [4] * 21, dtype=np.uint8), np.array([2] * 58 + [4] * 42, dtype=np.uint8)]
    expected = np.array([oracle.levenshtein_ratio(x, y) for x, y in zip(a, b)], dtype=np.uint8)
    assert expected[-2] == 57  # (58, 100) under strict IEEE
    for method in (0, 1):
        assert np.array_equal(ds.levenshtein_ratio_batch(a, b, method), expected)


def test_levenshtein_random_and_long(oracle):
    import doppel_speller_amd as ds
    rng = np.random.RandomState(99)
    a, b = [], []
    for _ in range(600):  # short alphabets make long common subsequences
        la, lb = rng.randint(0, 80), rng.randint(0, 80)
        a.append(rng.randint(1, 6, la).astype(np.uint8))
        b.append(rng.randint(1, 6, lb).astype(np.uint8))
    for _ in range(120):  # L > 255: uint8 wrap-around of the DP matrix
        la, lb = rng.randint(100, 256), rng.randint(156, 256)
        a.append(rng.randint(1, 38, la).astype(np.uint8))
        b.append(rng.randint(1, 38, lb).astype(np.uint8))
    for _ in range(40):  # pattern longer than 64, codes >= 64
        a.append(rng.randint(1, 200, rng.randint(65, 120)).astype(np.uint8))
        b.append(rng.randint(1, 200, rng.randint(65, 120)).astype(np.uint8))
    a.append(np.zeros(0, dtype=np.uint8)); b.append(np.zeros(0, dtype=np.uint8))
    a.append(np.full(255, 7, dtype=np.uint8)); b.append(np.full(255, 7, dtype=np.uint8))
    a.append(np.full(270, 7, dtype=np.uint8)); b.append(np.full(255, 9, dtype=np.uint8))
    expected = np.array([oracle.levenshtein_ratio(x, y) for x, y in zip(a, b)], dtype=np.uint8)
    for method in (0, 1):
        got = ds.levenshtein_ratio_batch(a, b, method)
        bad = np.nonzero(got != expected)[0]
        assert bad.shape[0] == 0, (method, bad[:5], got[bad[:5]], expected[bad[:5]])


def test_construct_features_golden(oracle, golden_features):
    import doppel_speller_amd as ds
    g = golden_features
    features = np.zeros_like(g["features"])
    dummy = np.zeros(ds.FEATURES_COUNT, dtype=np.uint8)
    assert ds.construct_features(g["title_len"], g["truth_len"], g["title_enc"], g["truth_enc"], g["counts"],
                                 g["space_code"], g["n_truth"], dummy, features) is None
    spec = oracle.construct_features(g["title_len"], g["truth_len"], g["title_enc"], g["truth_enc"], g["counts"],
                                     g["space_code"], g["n_truth"], "numba")
    assert np.array_equal(features.view(np.uint32), spec.view(np.uint32))
    # against the vectors captured from the reference: everything but the ranks is typing-independent
    assert np.array_equal(features[:, :51].view(np.uint32), g["features"][:, :51].view(np.uint32))
    ranks, captured = features[:, 51:], g["features"][:, 51:]
    assert np.array_equal(np.isnan(ranks), np.isnan(captured))
    ulps = np.abs(ranks.view(np.int32).astype(np.int64) - captured.view(np.int32))[~np.isnan(captured)]
    assert ulps.max() <= 1


def test_construct_features_edge_pairs_golden(oracle, golden_features_edge):
    """The reference's own answers on pairs the example data does not contain (section E of make_golden.py)."""
    import doppel_speller_amd as ds
    g = golden_features_edge
    for block, n_truth in enumerate(g["n_truth"]):
        captured = g["features"][block]
        features = np.zeros_like(captured)
        with np.errstate(all="ignore"):
            ds.construct_features(g["title_len"], g["truth_len"], g["title_enc"], g["truth_enc"], g["counts"],
                                  g["space_code"], n_truth, None, features)
            spec = oracle.construct_features(g["title_len"], g["truth_len"], g["title_enc"], g["truth_enc"], g["counts"],
                                             g["space_code"], n_truth, "numba")
        assert np.array_equal(features.view(np.uint32), spec.view(np.uint32))
        assert np.array_equal(features[:, :51].view(np.uint32), captured[:, :51].view(np.uint32))
        ranks, reference = features[:, 51:], captured[:, 51:]
        assert np.array_equal(np.isnan(ranks), np.isnan(reference))
        finite = ~np.isnan(reference)
        assert np.abs(ranks.view(np.int32).astype(np.int64) - reference.view(np.int32))[finite].max() <= 1


def _random_pairs(rng, n, long_titles=False):
    space = 1
    def title(max_len):
        words = []
        for _ in range(rng.randint(1, 9 if not long_titles else 22)):
            words.append(rng.randint(2, 12 if rng.rand() < 0.5 else 38, rng.randint(1, 14)).astype(np.uint8))
        out = []
        for i, word in enumerate(words):
            if i:
                out.append(np.array([space], dtype=np.uint8))
            out.append(word)
        return np.concatenate(out)[:max_len]
    q_enc = np.zeros((n, 255), dtype=np.uint8); t_enc = np.zeros((n, 255), dtype=np.uint8)
    q_len = np.zeros(n, dtype=np.uint8); t_len = np.zeros(n, dtype=np.uint8)
    for i in range(n):
        q, t = title(255), title(255)
        if rng.rand() < 0.5:  # related pair
            q = t.copy()
            for _ in range(rng.randint(0, 4)):
                at = rng.randint(len(q))
                q = np.delete(q, at) if rng.rand() < 0.5 and len(q) > 3 else np.insert(q, at, rng.randint(1, 38))
            q = q[:255]
        q_enc[i, :len(q)] = q; q_len[i] = len(q)
        t_enc[i, :len(t)] = t; t_len[i] = len(t)
    counts = rng.randint(1, 30000, (n, 15)).astype(np.uint32)
    counts[rng.rand(n, 15) < 0.05] = 1
    return q_len, t_len, q_enc, t_enc, counts


@pytest.mark.parametrize("long_titles", [False, True])
def test_construct_features_random(oracle, long_titles):
    import doppel_speller_amd as ds
    rng = np.random.RandomState(17 + long_titles)
    n = 1500 if not long_titles else 300
    q_len, t_len, q_enc, t_enc, counts = _random_pairs(rng, n, long_titles)
    features = np.zeros((n, ds.FEATURES_COUNT), dtype=np.float32)
    ds.construct_features(q_len, t_len, q_enc, t_enc, counts, np.uint8(1), np.uint32(30000),
                          np.zeros(66, dtype=np.uint8), features)
    expected = oracle.construct_features(q_len, t_len, q_enc, t_enc, counts, 1, 30000)
    bad = np.nonzero((features.view(np.uint32) != expected.view(np.uint32)).any(axis=1))[0]
    assert bad.shape[0] == 0, (bad[:5], features[bad[:1]], expected[bad[:1]])


def test_construct_features_edge_cases(oracle):
    import doppel_speller_amd as ds
    def pack(strings):
        enc = np.zeros((len(strings), 255), dtype=np.uint8)
        for i, s in enumerate(strings):
            enc[i, :len(s)] = s
        return enc, np.array([len(s) for s in strings], dtype=np.uint8)
    A = lambda *codes: np.array(codes, dtype=np.uint8)
    titles = [A(2), A(1, 1, 1), A(2, 3, 4), A(2, 1, 3), np.full(255, 5, np.uint8), A(1, 2, 1), A(2, 3), A(7, 7, 7, 7)]
    truths = [A(2), A(2, 3), A(1, 1), A(2, 1, 1, 3), np.full(255, 5, np.uint8), A(1), A(2, 3, 1), A(8, 1, 9)]
    q_enc, q_len = pack(titles)
    t_enc, t_len = pack(truths)
    counts = np.ones((len(titles), 15), dtype=np.uint32)
    counts[3] = 0  # zero count -> log(inf) = inf, nanmax = inf, inf - inf = nan
    features = np.zeros((len(titles), 66), dtype=np.float32)
    with np.errstate(all="ignore"):
        ds.construct_features(q_len, t_len, q_enc, t_enc, counts, 1, 1000, None, features)
        expected = oracle.construct_features(q_len, t_len, q_enc, t_enc, counts, 1, 1000)
    assert np.array_equal(features.view(np.uint32), expected.view(np.uint32)), (features[:, :10], expected[:, :10])


def test_construct_features_indexed_matches_direct(oracle):
    import doppel_speller_amd as ds
    from doppel_speller_amd import synth
    w = synth.make_workload(2000, 100, seed=3)
    queries = ds.TitleTable(w.q_enc, w.q_len)
    truth = ds.TitleTable(w.t_enc, w.t_len, w.t_counts)
    rng = np.random.RandomState(0)
    pair_q = rng.randint(0, 100, 700).astype(np.int32)
    pair_t = rng.randint(0, 2000, 700).astype(np.int32)
    got = ds.construct_features_indexed(queries, truth, pair_q, pair_t, ds.SPACE_CODE, w.n_truth)
    expected = oracle.construct_features(w.q_len[pair_q], w.t_len[pair_t], w.q_enc[pair_q], w.t_enc[pair_t],
                                         w.t_counts[pair_t], ds.SPACE_CODE, w.n_truth)
    assert np.array_equal(got.view(np.uint32), expected.view(np.uint32))


def test_idf_log_matches_libm_for_every_count():
    """float32(log(N / count)) from the device's float64 log against libm for every count."""
    import math
    import doppel_speller_amd as ds
    for n_truth in (30000, 500000):
        counts_all = np.arange(1, n_truth + 1, dtype=np.uint32)
        pad = (-counts_all.shape[0]) % 15
        counts = np.concatenate((counts_all, np.ones(pad, dtype=np.uint32))).reshape(-1, 15)
        n = counts.shape[0]
        word = np.arange(15) % 2 + 2  # "c d c d ..." 15 one-letter words
        title = np.zeros(255, dtype=np.uint8)
        title[0:29:2] = word.astype(np.uint8)
        title[1:29:2] = 1
        enc = np.broadcast_to(title, (n, 255)).copy()
        lengths = np.full(n, 29, dtype=np.uint8)
        features = np.zeros((n, 66), dtype=np.float32)
        ds.construct_features(lengths, lengths, enc, enc, counts, 1, n_truth, None, features)
        got = features[:, 36:51].reshape(-1)[:n_truth]
        expected = np.array([math.log(n_truth / c) for c in range(1, n_truth + 1)], dtype=np.float64).astype(np.float32)
        assert np.array_equal(got.view(np.uint32), expected.view(np.uint32))


def test_overlapping_launches_on_two_streams_share_one_truth_table(oracle):
    """Two indexed launches on different streams, enqueued back to back on the same truth table: each pulls its units from a
    work-queue head of its own (round 5: the heads are taken in turn), every pair is computed exactly once."""
    import ctypes
    import doppel_speller_amd as ds
    from doppel_speller_amd import _lib, synth
    from doppel_speller_amd.feature_engineering import TitleTable
    w = synth.make_workload(3000, 400, seed=31)
    queries, truth = TitleTable(w.q_enc, w.q_len), TitleTable(w.t_enc, w.t_len, w.t_counts)
    rng = np.random.RandomState(9)
    k = 10
    rows = [rng.randint(0, 3000, (400, k)).astype(np.int32) for _ in range(2)]
    streams, outs, d_rows = [], [], []
    for r in rows:
        stream = ctypes.c_void_p()
        _lib.check(_lib.lib().ds_stream_create(0, ctypes.byref(stream)), "ds_stream_create")
        streams.append(stream)
        d_rows.append(_lib.DeviceArray.from_host(r, 0))
        outs.append(_lib.DeviceArray((r.size, ds.FEATURES_COUNT), np.float32, 0))
    for repeat in range(3):   # several rounds: the launches of one round overlap, the heads rotate
        for stream, d_r, out in zip(streams, d_rows, outs):
            _lib.check(_lib.lib().ds_construct_features_indexed_device(
                queries.handle, truth.handle, ctypes.c_void_p(0), d_r.ptr, 0, k, ds.SPACE_CODE, w.n_truth, d_r.shape[0] * k,
                out.ptr, stream), "ds_construct_features_indexed_device")
    for stream in streams:
        _lib.check(_lib.lib().ds_stream_sync(stream, 0), "sync")
    for r, out in zip(rows, outs):
        pair_q, pair_t = np.repeat(np.arange(400), k), r.reshape(-1)
        expected = oracle.construct_features(w.q_len[pair_q], w.t_len[pair_t], w.q_enc[pair_q], w.t_enc[pair_t],
                                             w.t_counts[pair_t], ds.SPACE_CODE, w.n_truth)
        assert np.array_equal(out.to_host().view(np.uint32), expected.view(np.uint32))
    for stream in streams:
        _lib.lib().ds_stream_destroy(stream, 0)
