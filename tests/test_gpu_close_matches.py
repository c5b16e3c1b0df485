"""GPU parity of the fuzzy close-match step (predict.py:140-183) against the oracle restatement.
python-Levenshtein is not part of the reference tree: parity is pinned to oracle/doppel_oracle.c only (unpinned vs
the real third-party library)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _best_from_ratios(ratios, rows, threshold):
    best = np.full(ratios.shape[0], -1, dtype=np.int32)
    for q in range(ratios.shape[0]):
        above = ratios[q] > threshold
        if not above.any():
            continue
        top = ratios[q][above].max()
        hits = np.nonzero(ratios[q] == top)[0]
        if hits.shape[0] == 1:
            best[q] = rows[q, hits[0]]
    return best


def test_close_matches_synthetic(oracle):
    import doppel_speller_amd as ds
    from doppel_speller_amd import synth
    w = synth.make_workload(20000, 1500, seed=21)
    index = ds.TruthIndex(w.rowptr, w.truth_idx, w.idf32, w.sums32)
    rows = index.top_k(w.q_rowptr, w.q_cols, w.q_maxint, 10)
    queries = ds.TitleTable(w.q_enc, w.q_len)
    truth = ds.TitleTable(w.t_enc, w.t_len, w.t_counts)
    ratios, best = ds.find_close_matches(queries, truth, rows)
    pair_q = np.repeat(np.arange(rows.shape[0]), rows.shape[1])
    pair_t = rows.reshape(-1)
    expected = oracle.close_ratios(w.q_len[pair_q], w.t_len[pair_t], w.q_enc[pair_q], w.t_enc[pair_t], ds.SPACE_CODE,
                                   ds.SORT_KEY).reshape(rows.shape)
    assert np.array_equal(ratios, expected)
    assert (ratios > 94).sum() > 100 and (ratios == 0).sum() > 100  # both branches exercised
    assert np.array_equal(best, _best_from_ratios(expected, rows, 94))
    # most misspelled queries are found by the fuzzy step alone
    derived = w.actual_row >= 0
    assert (best[derived] == w.actual_row[derived]).mean() > 0.4


def test_close_matches_token_sort_and_long_titles(oracle):
    import doppel_speller_amd as ds
    rng = np.random.RandomState(3)
    alphabet = ds.ALLOWED_CHARACTERS
    words = ["alpha", "beta", "9zz", "a1", "limited", "ltd", "b", "zeta", "co", "0", "zz9", "a", "ab", "aa1"]
    xs, ys = [], []
    for _ in range(1500):
        x = " ".join(rng.choice(words, rng.randint(1, 9)))
        y = list(x)
        for _ in range(rng.randint(0, 3)):
            at = rng.randint(len(y))
            if rng.rand() < 0.5 and len(y) > 3:
                del y[at]
            else:
                y.insert(at, rng.choice(list("abz19 ")))
        y = " ".join("".join(y).split()) or "a"
        if rng.rand() < 0.5:
            parts = y.split()
            rng.shuffle(parts)
            y = " ".join(parts)
        xs.append(x)
        ys.append(y)
    for _ in range(200):  # long titles: pattern > 64 characters, many one-letter words
        x = " ".join(rng.choice(words, rng.randint(25, 45)))[:255]
        y = x[:rng.randint(len(x) - 3, len(x) + 1)] + "x" * rng.randint(0, 3)
        xs.append(x.strip())
        ys.append(y.strip()[:255])
    q_enc, q_len = ds.encode_titles(xs)
    t_enc, t_len = ds.encode_titles(ys)
    queries = ds.TitleTable(q_enc, q_len)
    truth = ds.TitleTable(t_enc, t_len, np.ones((len(ys), 15), dtype=np.uint32))
    rows = np.arange(len(xs), dtype=np.int32)[:, None]
    ratios, best = ds.find_close_matches(queries, truth, rows)
    expected = oracle.close_ratios(q_len, t_len, q_enc, t_enc, ds.SPACE_CODE, ds.SORT_KEY)
    bad = np.nonzero(ratios[:, 0] != expected)[0]
    assert bad.shape[0] == 0, (bad[:5], [xs[i] for i in bad[:3]], [ys[i] for i in bad[:3]], ratios[bad[:5], 0],
                               expected[bad[:5]])
    assert np.array_equal(best, np.where(expected > 94, rows[:, 0], -1))


def test_close_matches_inside_the_device_pipeline(oracle):
    """The same step enqueued behind the Jaccard kernels on the top-k rows resident in HBM (no host round trip)."""
    import doppel_speller_amd as ds
    from doppel_speller_amd import synth
    w = synth.make_workload(30000, 800, seed=33)
    pipeline = ds.CandidatePipeline(w, 10)
    pipeline.enqueue_top_k()
    pipeline.enqueue_close_matches()
    pipeline.enqueue_features()
    assert pipeline.sync()["error_queries"] == 0
    rows = pipeline.rows()
    ratios, best = pipeline.close_matches()
    pair_q = np.repeat(np.arange(rows.shape[0]), rows.shape[1])
    pair_t = rows.reshape(-1)
    expected = oracle.close_ratios(w.q_len[pair_q], w.t_len[pair_t], w.q_enc[pair_q], w.t_enc[pair_t], ds.SPACE_CODE,
                                   ds.SORT_KEY).reshape(rows.shape)
    assert np.array_equal(ratios, expected)
    assert np.array_equal(best, _best_from_ratios(expected, rows, 94))
