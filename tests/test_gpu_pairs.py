"""GPU parity of the device-resident pair list between the fuzzy step and the model (SURVEY.md 8f-2,
predict.py:172-183, 195-204, 246-252) against the NumPy restatement in oracle/oracle.py."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_queries,k", [(3000, 10), (1025, 7), (64, 100)])
def test_remaining_pairs_flow(oracle, n_queries, k):
    """top-k -> close matches -> compaction of the unmatched queries' pairs -> features / predictions / selection, all
    on data that stays in HBM; every stage equals the restatement on the same inputs."""
    import doppel_speller_amd as ds
    from doppel_speller_amd import ForestModel, synth
    w = synth.make_workload(40000, n_queries, seed=29 + k)
    pipeline = ds.CandidatePipeline(w, k)
    pipeline.enqueue_top_k()
    pipeline.enqueue_close_matches()
    pipeline.enqueue_remaining_pairs()
    n_remaining, n_pairs = pipeline.remaining_counts()
    rows = pipeline.rows()
    _, best = pipeline.close_matches()
    expected_q, expected_t = oracle.remaining_pairs(best, rows)
    assert n_remaining == int((best < 0).sum()) and n_pairs == n_remaining * k == expected_q.shape[0]
    assert 0 < n_remaining < n_queries                      # both kinds of query are present
    pair_q, pair_t = pipeline.remaining_pairs(n_pairs)
    assert np.array_equal(pair_q, expected_q) and np.array_equal(pair_t, expected_t)

    # features of exactly those pairs (predict.py:195-219)
    pipeline.enqueue_features_remaining(n_pairs)
    features = pipeline.features(n_pairs)
    sample = np.unique(np.linspace(0, n_pairs - 1, 2000).astype(np.int64))
    reference = oracle.construct_features(w.q_len[pair_q[sample]], w.t_len[pair_t[sample]], w.q_enc[pair_q[sample]],
                                          w.t_enc[pair_t[sample]], w.t_counts[pair_t[sample]], 1, w.n_truth)
    assert np.array_equal(features[sample].view(np.uint32), reference.view(np.uint32))

    # model scores and the per-query selection (predict.py:229-252)
    f = synth.make_forest(n_trees=40)
    model = ForestModel(f["feature"], f["threshold"], f["yes"], f["no"], f["missing"], f["tree_offsets"],
                        f["n_features"], f["base_margin"])
    pipeline.enqueue_predict(model, n_pairs=n_pairs)
    for threshold in (0.9, 0.3):
        pipeline.enqueue_select_matches(n_remaining, threshold)
        match_query, match_row = pipeline.matches(n_remaining)
        predictions = pipeline.predictions(n_pairs)
        expected_query, expected_row = oracle.select_matches(pair_q, pair_t, predictions, k, threshold)
        assert np.array_equal(match_query, expected_query) and np.array_equal(match_row, expected_row)


def test_select_matches_ties_and_threshold(oracle):
    """Crafted predictions: a tie at the maximum (no match), a maximum exactly at the threshold (no match: strictly
    greater is required), a single maximum above it (match), everything below it (no match)."""
    from doppel_speller_amd import _lib
    k = 4
    predictions = np.array([[0.95, 0.95, 0.1, 0.2],      # tie at the top
                            [0.9, 0.1, 0.2, 0.3],         # float32(0.9) is not > float32(0.9)
                            [0.2, 0.97, 0.96, 0.1],       # single maximum
                            [0.5, 0.4, 0.3, 0.2],         # below the threshold
                            [0.91, 0.2, 0.91, 0.99]], dtype=np.float32)
    pair_q = np.repeat(np.array([7, 8, 11, 12, 40], dtype=np.int32), k)
    pair_t = np.arange(100, 100 + pair_q.shape[0], dtype=np.int32)
    d = [_lib.DeviceArray.from_host(x) for x in (pair_q, pair_t, predictions.reshape(-1))]
    out_q, out_t = _lib.DeviceArray((5,), np.int32), _lib.DeviceArray((5,), np.int32)
    _lib.check(_lib.lib().ds_select_matches_device(d[0].ptr, d[1].ptr, d[2].ptr, 5, k, 0.9, out_q.ptr, out_t.ptr,
                                                   ctypes.c_void_p(0)), "select")
    _lib.check(_lib.lib().ds_stream_sync(None, 0), "sync")
    expected_q, expected_t = oracle.select_matches(pair_q, pair_t, predictions, k, 0.9)
    assert np.array_equal(out_q.to_host(), expected_q) and np.array_equal(out_t.to_host(), expected_t)
    assert expected_t.tolist() == [-1, -1, 109, -1, 119]


def test_everything_or_nothing_remaining():
    """Edge cases of the compaction: every query matched by the fuzzy step, none matched, and an empty batch."""
    from doppel_speller_amd import _lib
    lib = _lib.lib()
    k, n = 3, 2500
    rows = np.arange(n * k, dtype=np.int32).reshape(n, k)
    d_rows = _lib.DeviceArray.from_host(rows)
    pair_q, pair_t = _lib.DeviceArray((n * k,), np.int32), _lib.DeviceArray((n * k,), np.int32)
    counts = _lib.DeviceArray((int(lib.ds_remaining_pairs_counts_size(n)),), np.int64)
    for best_value, expected in ((5, 0), (-1, n)):
        best = _lib.DeviceArray.from_host(np.full(n, best_value, dtype=np.int32))
        _lib.check(lib.ds_remaining_pairs_device(best.ptr, d_rows.ptr, n, k, 0, pair_q.ptr, pair_t.ptr, counts.ptr,
                                                 ctypes.c_void_p(0)), "pairs")
        _lib.check(lib.ds_stream_sync(None, 0), "sync")
        assert counts.to_host()[:2].tolist() == [expected, expected * k]
        if expected:
            assert np.array_equal(pair_q.to_host(), np.repeat(np.arange(n, dtype=np.int32), k))
            assert np.array_equal(pair_t.to_host(), rows.reshape(-1))
    _lib.check(lib.ds_remaining_pairs_device(None, None, 0, k, 0, None, None, counts.ptr, ctypes.c_void_p(0)), "pairs")
    _lib.check(lib.ds_stream_sync(None, 0), "sync")
    assert counts.to_host()[:2].tolist() == [0, 0]
