"""Larger-shape checks on the GPU: many tiles (several list-pointer blocks), top-50, fused pipeline, and the
size-independent properties of the outputs (BASELINE.json configs[2] shape, scaled to what the oracle can check)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big_workload():
    from doppel_speller_amd import synth
    return synth.make_workload(1_200_000, 3000, seed=77)


def test_many_tiles_top50_against_oracle(oracle, big_workload):
    import doppel_speller_amd as ds
    w = big_workload
    k = 50
    pipeline = ds.CandidatePipeline(w, k)
    info = pipeline.index.info()
    assert info["tiles"] == -(-w.n_truth // info["tile_rows"]) and info["tiles"] > 32  # several list-pointer blocks
    # the forward index in its narrow form (round 5): uint32 row starts, uint16 columns (V <= 65536, nnz < 2^32)
    assert info["forward_index_bytes"] == 4 * (w.n_truth + 1) + 2 * info["nnz"] and info["forward_index_bytes"] < info["device_bytes"]
    pipeline.step()
    stats = pipeline.sync()
    assert stats["error_queries"] == 0
    rows = pipeline.rows()
    # properties that do not need the oracle: in range, strictly descending row indexes (match_maker.py:71)
    assert rows.min() >= 0 and rows.max() < w.n_truth
    assert (np.diff(rows.astype(np.int64), axis=1) < 0).all()
    # idempotence: a second pass over the same resident inputs gives the same rows and features
    features_first = pipeline.features()
    pipeline.step()
    pipeline.sync()
    assert np.array_equal(rows, pipeline.rows())
    assert np.array_equal(features_first.view(np.uint32), pipeline.features().view(np.uint32))
    # oracle on a sample of queries (dense path of the reference: 1.2M rows per query)
    sample = np.r_[0:24, 1500:1524]
    first, last = w.q_rowptr[sample], w.q_rowptr[sample + 1]
    q_cols = np.concatenate([w.q_cols[a:b] for a, b in zip(first, last)])
    q_rowptr = np.concatenate(([0], np.cumsum(last - first)))
    expected = oracle.jaccard_topk(w.rowptr, w.truth_idx, w.idf32, w.sums32, q_rowptr, q_cols, w.q_maxint[sample], k)
    assert np.array_equal(rows[sample], expected)
    # the misspelled queries find the title they were derived from (recall sanity, not a parity statement)
    derived = np.nonzero(w.actual_row >= 0)[0]
    recall = np.mean([w.actual_row[q] in rows[q] for q in derived])
    assert recall > 0.75
    # features of the sampled queries' pairs against the oracle
    pair_q = np.repeat(sample, k)
    pair_t = rows[sample].reshape(-1)
    reference = oracle.construct_features(w.q_len[pair_q], w.t_len[pair_t], w.q_enc[pair_q], w.t_enc[pair_t],
                                          w.t_counts[pair_t], ds.SPACE_CODE, w.n_truth)
    got = features_first.reshape(w.n_queries, k, ds.FEATURES_COUNT)[sample].reshape(-1, ds.FEATURES_COUNT)
    assert np.array_equal(got.view(np.uint32), reference.view(np.uint32))
    # feature invariants over ALL pairs: lengths, ratios in [0, 100], NaN exactly beyond the truth word count
    all_features = features_first.reshape(w.n_queries * k, ds.FEATURES_COUNT)
    assert np.array_equal(all_features[:, 1], w.t_len[rows.reshape(-1)].astype(np.float32))
    assert ((all_features[:, 4] >= 0) & (all_features[:, 4] <= 100)).all()
    words = np.minimum(all_features[:, 3].astype(np.int64), 15)
    nan_expected = np.arange(15)[None, :] >= words[:, None]
    assert np.array_equal(np.isnan(all_features[:, 6:21]), nan_expected)


def test_end_to_end_example_runs():
    """examples/end_to_end.py: raw titles -> transform -> native index build -> top-k -> close matches -> features ->
    tree ensemble, all through the package."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "end_to_end.py")
    spec = importlib.util.spec_from_file_location("end_to_end_example", path)
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    rows, best_row, features, probabilities = module.main(5000, 300, 10)
    assert rows.shape == (300, 10) and features.shape == (3000, 66) and probabilities.shape == (300, 10)
    assert (rows >= 0).all() and (best_row >= -1).all() and ((probabilities > 0) & (probabilities < 1)).all()
    assert (best_row >= 0).sum() > 50          # the misspelled copies are mostly caught by the fuzzy step alone
