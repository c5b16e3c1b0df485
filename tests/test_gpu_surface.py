"""The reference's own call surface at speed (round 4): get_closest_matches per row, the 9-argument construct_features
through the chunked / pinned / packed host path, the single-pair Levenshtein entry of SURVEY.md 8b, any top_n <= N.

Reference call sites: predict.py:126-127 (one get_closest_matches per row in a dict comprehension), predict.py:215-219 (one
construct_features over a chunk's pairs), match_maker.py:183-190 (any top_n, `.loc` on the truth frame)."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_get_closest_matches_is_a_table_lookup(golden_match_maker):
    """ids == the reference's for every captured query, through the per-row method; the per-call cost is a NumPy row
    read (the reference's `.loc` per call costs ~270 us on a 500k-row frame), for a RangeIndex and for a labelled index."""
    import doppel_speller_amd as ds
    from conftest import golden_frames
    g = golden_match_maker
    data, truth, vocabulary = golden_frames(g)
    mm = ds.MatchMaker(data, truth, 10, vocabulary=vocabulary)
    nearest = {row: mm.get_closest_matches(row) for row in range(200)}          # predict.py:126-127
    assert all(nearest[q] == g["ids_k10"][q].tolist() for q in range(200))
    assert all(type(i) is int for i in nearest[0])
    t0 = time.perf_counter()
    for _ in range(20):
        for row in range(200):
            mm.get_closest_matches(row)
    per_call = (time.perf_counter() - t0) / 4000
    assert per_call < 20e-6, per_call          # ~1 us; the pandas look-up it replaces: 270 us
    # a truth frame whose index is NOT range(N): `.loc` label semantics are kept (match_maker.py:190)
    relabelled = truth.copy()
    relabelled.index = np.arange(len(truth))[::-1].copy()                        # label r sits at position N - 1 - r
    data2, _, _ = golden_frames(g)
    mm2 = ds.MatchMaker(data2, relabelled, 10, vocabulary=vocabulary)
    for q in (0, 7, 199):
        rows = mm2.get_closest_matches_batch([q])[0]
        assert mm2.get_closest_matches(q) == relabelled.loc[rows, "title_id"].tolist()
    # `truth_data` reassigned after the first look-up (the reference reads it on every call, :190): the ids follow it
    first = mm2.get_closest_matches(3)
    shifted = mm2.truth_data.copy()
    shifted["title_id"] = shifted["title_id"] + 1_000_000
    mm2.truth_data = shifted
    assert mm2.get_closest_matches(3) == [i + 1_000_000 for i in first]
    # duplicate labels: `.loc` returns every row of a label -- whatever the reference's per-call look-up gives
    doubled = truth.copy()
    doubled.index = np.arange(len(truth)) // 2                                   # every label twice
    data3, _, _ = golden_frames(g)
    mm3 = ds.MatchMaker(data3, doubled, 10, vocabulary=vocabulary)
    def outcome(call):   # the list, or the exception the reference's `.loc` raises for a row number that is no label
        try:
            return call()
        except Exception as error:  # noqa: BLE001
            return type(error).__name__
    for q in (5, 17, 101):
        rows = mm3.get_closest_matches_batch([q])[0]
        assert outcome(lambda: mm3.get_closest_matches(q)) == outcome(lambda: doubled.loc[rows, "title_id"].tolist())


def test_nine_argument_construct_features_through_the_staged_path(oracle):
    """Several chunks of 16384 pairs (an odd tail), several host threads, non-contiguous callers, n = 1: bit-exact vs the
    oracle and vs the indexed (device-table) path."""
    import doppel_speller_amd as ds
    from doppel_speller_amd import synth
    w = synth.make_workload(20000, 3000, seed=11)
    rng = np.random.RandomState(5)
    n = 3 * 16384 + 1237
    pair_q = rng.randint(0, w.n_queries, n).astype(np.int32)
    pair_t = rng.randint(0, w.n_truth, n).astype(np.int32)
    args = (w.q_len[pair_q], w.t_len[pair_t], w.q_enc[pair_q], w.t_enc[pair_t], w.t_counts[pair_t])
    features = np.zeros((n, ds.FEATURES_COUNT), dtype=np.float32)
    dummy = np.zeros(ds.FEATURES_COUNT, dtype=np.uint8)
    with np.errstate(all="ignore"):
        ds.construct_features(*args, ds.SPACE_CODE, w.n_truth, dummy, features)
    queries, truth = ds.TitleTable(w.q_enc, w.q_len), ds.TitleTable(w.t_enc, w.t_len, w.t_counts)
    indexed = ds.construct_features_indexed(queries, truth, pair_q, pair_t, ds.SPACE_CODE, w.n_truth)
    assert np.array_equal(features.view(np.uint32), indexed.view(np.uint32))
    sample = rng.choice(n, 600, replace=False)
    expected = oracle.construct_features(*(a[sample] for a in args), ds.SPACE_CODE, w.n_truth)
    assert np.array_equal(features[sample].view(np.uint32), expected.view(np.uint32))
    # a second call reuses the staging slots; one pair; a response that is a strided view
    one = np.zeros(ds.FEATURES_COUNT, dtype=np.float32)
    ds.construct_features(args[0][5], args[1][5], args[2][5], args[3][5], args[4][5], ds.SPACE_CODE, w.n_truth, dummy, one)
    assert np.array_equal(one.view(np.uint32), features[5].view(np.uint32))
    wide = np.zeros((1000, 2 * ds.FEATURES_COUNT), dtype=np.float32)
    view = wide[:, :ds.FEATURES_COUNT]
    ds.construct_features(*(a[:1000] for a in args), ds.SPACE_CODE, w.n_truth, dummy, view)
    assert np.array_equal(view.view(np.uint32), features[:1000].view(np.uint32)) and not wide[:, ds.FEATURES_COUNT:].any()


def test_staged_path_at_the_largest_titles_the_encoding_allows(oracle):
    """Every pair 255 + 255 characters (the uint8 length's limit, settings.py:67-68): a chunk's packed bytes fill its pinned
    staging buffer to the last byte (16,384 pairs x 580 B), the DP matrices wrap (L = 510 > 255: literal path), two chunks."""
    import doppel_speller_amd as ds
    rng = np.random.RandomState(255)
    n = 16384 + 37
    q_enc = rng.randint(1, 38, (n, 255)).astype(np.uint8)   # spaces (code 1) wherever they fall
    t_enc = rng.randint(2, 38, (n, 255)).astype(np.uint8)
    t_enc[:, ::9] = 1                                   # 29 spaces: 30 words per truth title (the first one empty), 15 of them count
    lengths = np.full(n, 255, dtype=np.uint8)
    counts = rng.randint(1, 5000, (n, 15)).astype(np.uint32)
    features = np.zeros((n, ds.FEATURES_COUNT), dtype=np.float32)
    ds.construct_features(lengths, lengths, q_enc, t_enc, counts, ds.SPACE_CODE, 500000, np.zeros(66, np.uint8), features)
    sample = np.concatenate((np.arange(8), [16383, 16384, n - 1]))     # both sides of the chunk boundary
    expected = oracle.construct_features(lengths[sample], lengths[sample], q_enc[sample], t_enc[sample], counts[sample],
                                         ds.SPACE_CODE, 500000)
    assert np.array_equal(features[sample].view(np.uint32), expected.view(np.uint32))
    assert np.isfinite(features[:, :6]).all() and (features[:, 0] == 255).all() and (features[:, 3] == 30).all()


def test_single_pair_levenshtein_entry(oracle, golden_kat):
    """`ds_levenshtein_ratio(a, la, b, lb)` as SURVEY.md 8b lists it: the reference's uint8 for one pair."""
    import ctypes
    from doppel_speller_amd import _lib
    table = {ch: i for i, ch in enumerate(golden_kat["alphabet"])}
    lib = _lib.lib()
    for case in golden_kat["levenshtein"]:
        a = np.array([table[ch] for ch in case["a"]], dtype=np.uint8)
        b = np.array([table[ch] for ch in case["b"]], dtype=np.uint8)
        assert lib.ds_levenshtein_ratio(_lib.pointer(a), len(a), _lib.pointer(b), len(b)) == case["ratio"]
        assert lib.ds_levenshtein_ratio(_lib.pointer(b), len(b), _lib.pointer(a), len(a)) == case["ratio"]
    long_a, long_b = np.full(200, 7, np.uint8), np.full(255, 9, np.uint8)       # L > 255: the uint8 matrix wraps
    assert lib.ds_levenshtein_ratio(_lib.pointer(long_a), 200, _lib.pointer(long_b), 255) == oracle.levenshtein_ratio(long_a, long_b)
    assert lib.ds_levenshtein_ratio(ctypes.c_void_p(0), 0, ctypes.c_void_p(0), 0) == 0
    assert lib.ds_levenshtein_ratio(ctypes.c_void_p(0), 3, _lib.pointer(long_b), 3) == -1     # DS_E_ARG


@pytest.mark.parametrize("k", [513, 700, 2000])
def test_top_n_above_the_selection_kernels_limit(oracle, k):
    """match_maker.py:183-190 accepts any top_n <= N; above 512 every query is answered by the row scan (round 3 returned
    DS_E_ARG)."""
    import doppel_speller_amd as ds
    from test_gpu_jaccard import _random_problem
    rng = np.random.RandomState(k)
    problem = _random_problem(rng, 30000, 1500, 24, mean_cols=12)
    index = ds.TruthIndex(problem["rowptr"], problem["truth_idx"], problem["idf32"], problem["sums32"])
    got = index.top_k(problem["q_rowptr"], problem["q_cols"], problem["q_maxint"], k)
    expected = oracle.jaccard_topk(problem["rowptr"], problem["truth_idx"], problem["idf32"], problem["sums32"],
                                   problem["q_rowptr"], problem["q_cols"], problem["q_maxint"], k)
    assert np.array_equal(got, expected)
    assert (index.status(24) == 1).all()
    # a bad column is still an argument error on this path
    bad_cols = problem["q_cols"].copy()
    bad_cols[3] = 10 ** 6
    with pytest.raises(ds.DoppelError, match="column index"):
        index.top_k(problem["q_rowptr"], bad_cols, problem["q_maxint"], k)
