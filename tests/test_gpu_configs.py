"""BASELINE.json's single-GPU configurations at their FULL size through the fused device-resident path:
C2 = 100k queries x 500k truth titles, top-10; C3 = 1M queries x 5M truth titles, top-50 (= one GPU's shard of C4);
and ONE GPU's shard of C5 (8 GPUs: 1M queries x 50M truth titles, top-100 + all features): 125,000 queries against
the replicated 50M-row truth index.

Every output row is checked through size-independent properties (index range, strictly descending row indexes,
idempotence of a second launch, per-query status, feature identities that do not need the oracle), and the oracle is
run on an evenly spaced sample of the queries PLUS every query the fast kernel handed to the literal kernel."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check_config(oracle, n_truth, n_queries, k, oracle_sample, seed=20260101, feature_sample=400, literal_share=0.002):
    import time
    import doppel_speller_amd as ds
    from doppel_speller_amd import synth
    started = time.perf_counter()
    w = synth.make_workload(n_truth, n_queries, seed=seed)
    generated = time.perf_counter()
    pipeline = ds.CandidatePipeline(w, k)
    built = time.perf_counter()
    print(f"[{n_queries} x {n_truth}, top-{k}] workload {generated - started:.1f} s, index build + uploads "
          f"{built - generated:.1f} s", flush=True)
    pipeline.step()
    stats = pipeline.sync()
    rows = pipeline.rows()
    status = pipeline.index.status(n_queries)
    assert stats["error_queries"] == 0 and set(np.unique(status)) <= {0, 1}
    assert stats["dense_queries"] == int((status == 1).sum())
    assert stats["dense_reasons"]["ties"] == 0            # ties are served by the fast kernel (duplicate ranks)
    assert stats["dense_queries"] <= literal_share * n_queries   # the literal kernel is the rare path

    # ---- every row: range, strictly descending row indexes (match_maker.py:71 `[::-1][:k]`)
    assert rows.shape == (n_queries, k) and rows.dtype == np.int32
    assert rows.min() >= 0 and rows.max() < n_truth
    assert (np.diff(rows.astype(np.int64), axis=1) < 0).all()
    # a derived query finds the row it was derived from -- unless that row has k twins of larger index (the
    # generator repeats short titles; match_maker.py:71 keeps the k largest row indexes of a tie)
    derived = np.nonzero(w.actual_row >= 0)[0]
    found = (rows[derived] == w.actual_row[derived, None]).any(axis=1)
    assert found.mean() > 0.6

    # ---- every pair: feature identities that need no oracle (feature_engineering.py:164-169), in chunks of 4M pairs
    slots = np.arange(15)[None, :]
    chunk = 4_000_000 // k * k
    for first in range(0, n_queries * k, chunk):
        features = pipeline.features(min(chunk, n_queries * k - first), first)
        pair_t = rows.reshape(-1)[first:first + features.shape[0]]
        pair_q = np.arange(first, first + features.shape[0], dtype=np.int64) // k
        assert np.array_equal(features[:, 0], w.q_len[pair_q].astype(np.float32))
        assert np.array_equal(features[:, 1], w.t_len[pair_t].astype(np.float32))
        distinct, where = np.unique(pair_t, return_inverse=True)                 # feature_engineering.py:105
        words_t = ((w.t_enc[distinct] == 1).sum(axis=1) + 1).astype(np.int64)[where]
        assert np.array_equal(features[:, 3], words_t.astype(np.float32))
        assert (features[:, 4] >= 0).all() and (features[:, 4] <= 100).all()
        for block in (6, 21, 36, 51):                         # best ratios, word lengths, idf, ranks: NaN beyond the words
            assert np.array_equal(np.isnan(features[:, block:block + 15]), slots >= np.minimum(words_t, 15)[:, None])
        del features

    # ---- idempotence: a second launch over the same resident inputs gives the same bytes
    pipeline.enqueue_top_k()
    pipeline.sync()
    assert np.array_equal(pipeline.rows(), rows)

    # ---- the oracle on a sample + on every literal-kernel query
    sample = np.unique(np.concatenate((np.linspace(0, n_queries - 1, oracle_sample).astype(np.int64),
                                       np.nonzero(status == 1)[0])))
    counts = np.diff(w.q_rowptr)[sample]
    starts = w.q_rowptr[sample]
    flat = np.concatenate([w.q_cols[s:s + c] for s, c in zip(starts, counts)]) if sample.shape[0] else w.q_cols[:0]
    sub_rowptr = np.concatenate(([0], np.cumsum(counts))).astype(np.int64)
    expected = oracle.jaccard_topk(w.rowptr, w.truth_idx, w.idf32, w.sums32, sub_rowptr, flat, w.q_maxint[sample], k)
    bad = np.nonzero((rows[sample] != expected).any(axis=1))[0]
    assert bad.shape[0] == 0, (sample[bad][:10], rows[sample][bad[:2]], expected[bad[:2]])
    check = sample[:min(sample.shape[0], feature_sample)]
    got = pipeline.features_of(check)
    pq = np.repeat(check, k)
    pt = rows[check].reshape(-1)
    reference = oracle.construct_features(w.q_len[pq], w.t_len[pt], w.q_enc[pq], w.t_enc[pt], w.t_counts[pt], 1, n_truth)
    assert np.array_equal(got.view(np.uint32), reference.view(np.uint32))
    return stats


def test_c2_full_size(oracle):
    """BASELINE.json configs[1]: 100k synthetic queries x 500k truth titles, tri-gram vocabulary ~50k, top-10."""
    stats = _check_config(oracle, 500_000, 100_000, 10, oracle_sample=2000)
    assert stats["sparse_tiles"] > stats["dense_tiles"]


def test_c2_top100_full_size(oracle):
    """C2's truth set and queries at the reference's OWN prediction setting, top_n = 100 (settings.py:56, predict.py:289): the
    narrow geometry's large-k instantiation of the fast kernel (two bootstrap samples per thread; about half of the queries
    process an epoch of sparse tiles twice because 112 candidates and their near-ties crowd the 768-entry buffer).  2,000
    sampled queries + every literal-kernel query against the oracle, the features of 400 queries' 40,000 pairs."""
    # (636 of the 100,000 queries have fewer than 100 rows with a positive score: the literal kernel answers those without a sweep)
    stats = _check_config(oracle, 500_000, 100_000, 100, oracle_sample=2000, feature_sample=400, literal_share=0.01)
    assert stats["sparse_tiles"] > stats["dense_tiles"]
    assert stats["sparse_redos"] > 0   # the redo path is part of what this configuration covers


def test_c3_full_size(oracle):
    """BASELINE.json configs[2]: 1M queries x 5M truth titles, top-50 (175 score tiles, 50M pairs)."""
    _check_config(oracle, 5_000_000, 1_000_000, 50, oracle_sample=200)


def test_c5_shard(oracle):
    """BASELINE.json configs[4] as ONE of its 8 GPUs sees it: the truth index of 50M titles is replicated, the GPU owns
    1M / 8 = 125,000 queries, top-100, and the full feature vector of its 12.5M pairs.  The host side (native title
    generator, threaded ds_problem_create / ds_index_create, a7 encoders) must get there in about a minute."""
    threads = oracle.num_threads()
    oracle.set_num_threads(min(16, threads))    # one 600 MB N-vector pair per oracle thread at N = 50M
    try:
        stats = _check_config(oracle, 50_000_000, 125_000, 100, oracle_sample=30, feature_sample=50)
    finally:
        oracle.set_num_threads(threads)
    assert stats["sparse_tiles"] > stats["dense_tiles"]
