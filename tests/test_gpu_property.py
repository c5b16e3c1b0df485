"""Property-based parity of the Jaccard top-k path (hypothesis): small inverted indexes of adversarial shapes -- few rows, rows around
the narrow tile's 12,288-row boundary, heavy ties and twins, rows with one column, queries of 0..128 columns with unseen / zero-IDF /
every column, k from 1 to the number of rows -- against the oracle, bit for bit, through the C ABI (round 4: forward index in the exact
stage, split rank counting, two-level collect test, top_n above 512 through the row scan)."""
import os

import numpy as np
import pytest

hypothesis = pytest.importorskip("hypothesis")
from hypothesis import HealthCheck, given, settings, strategies as st  # noqa: E402

pytestmark = pytest.mark.gpu

# A soak run (scripts/r05/soak.sh) raises the number of examples and lets hypothesis draw fresh ones:
# DS_PROPERTY_EXAMPLES=1000 python -m pytest tests/test_gpu_property.py -m gpu
_SOAK = int(os.environ.get("DS_PROPERTY_EXAMPLES", "0"))


def _problem(seed, n_truth, n_columns, shape, n_queries):
    rng = np.random.RandomState(seed)
    if shape == "twins":                       # a handful of distinct column sets, each repeated many times
        kinds = [np.unique(rng.randint(0, n_columns, rng.randint(1, 9))) for _ in range(rng.randint(1, 6))]
        row_columns = [kinds[rng.randint(len(kinds))] for _ in range(n_truth)]
    elif shape == "single":                    # most rows hold one column: massive ties
        row_columns = [np.array([rng.randint(n_columns)]) if rng.rand() < 0.9 else np.unique(rng.randint(0, n_columns, 3))
                       for _ in range(n_truth)]
    elif shape == "dense":                     # a few columns present in most rows (signature columns) + a sparse tail
        heavy = min(n_columns, 6)
        row_columns = [np.unique(np.concatenate((np.nonzero(rng.rand(heavy) < 0.6)[0], rng.randint(0, n_columns, rng.randint(1, 5)))))
                       for _ in range(n_truth)]
    else:                                      # random
        row_columns = [np.unique(rng.randint(0, n_columns, rng.randint(1, 14))) for _ in range(n_truth)]
    rows = np.concatenate([np.full(len(c), t) for t, c in enumerate(row_columns)])
    cols = np.concatenate(row_columns)
    order = np.argsort(cols, kind="stable")
    truth_idx = rows[order].astype(np.int32)
    df = np.bincount(cols, minlength=n_columns)
    rowptr = np.concatenate(([0], np.cumsum(df))).astype(np.int64)
    idf64 = np.log(n_truth / np.maximum(df, 1))            # a column present in every row has idf 0 (match_maker.py:139)
    idf64[df == 0] = idf64.max() if (df > 0).any() else 1.0
    idf32 = idf64.astype(np.float32)
    sums32 = np.zeros(n_truth, dtype=np.float32)
    for t, c in enumerate(row_columns):                     # sequential float32 sum, ascending columns
        for column in c:
            sums32[t] = sums32[t] + idf32[column]
    q_cols, q_maxint = [], []
    for q in range(n_queries):
        kind = rng.randint(5)
        if kind == 0:
            c = row_columns[rng.randint(n_truth)]            # equal to a truth row
        elif kind == 1:
            c = np.unique(rng.randint(0, n_columns, rng.randint(0, min(129, n_columns + 1))))   # 0..128 random columns
        elif kind == 2:
            c = np.arange(min(n_columns, 128))                # every column (up to the fast path's limit)
        elif kind == 3:
            base = row_columns[rng.randint(n_truth)]
            c = np.unique(np.concatenate((base, rng.randint(0, n_columns, 2))))   # a truth row + noise
        else:
            c = np.unique(rng.randint(0, n_columns, rng.randint(1, 30)))
        c = c[idf32[c] != 0].astype(np.int64)                # lil_matrix(...).nonzero() drops explicit zeros (match_maker.py:118)
        q_cols.append(c)
        total = 0.0
        for value in idf64[c]:
            total = total + float(value)
        q_maxint.append(total)
    q_rowptr = np.concatenate(([0], np.cumsum([len(c) for c in q_cols]))).astype(np.int64)
    flat = np.concatenate(q_cols).astype(np.int32) if q_rowptr[-1] else np.zeros(0, np.int32)
    return dict(rowptr=rowptr, truth_idx=truth_idx, idf32=idf32, sums32=sums32, q_rowptr=q_rowptr, q_cols=flat,
                q_maxint=np.array(q_maxint, dtype=np.float64))


@settings(max_examples=_SOAK or 60, deadline=None, suppress_health_check=list(HealthCheck), derandomize=not _SOAK)
@given(seed=st.integers(0, 2 ** 31 - 1),
       n_truth=st.one_of(st.integers(1, 300), st.integers(12200, 12400), st.integers(24500, 24700), st.integers(300, 30000)),
       n_columns=st.integers(1, 400), shape=st.sampled_from(["random", "twins", "single", "dense"]),
       k_kind=st.sampled_from(["one", "small", "reference", "large", "all"]))
def test_top_k_equals_the_oracle_on_adversarial_indexes(oracle, seed, n_truth, n_columns, shape, k_kind):
    import doppel_speller_amd as ds
    problem = _problem(seed, n_truth, n_columns, shape, n_queries=12)
    k = {"one": 1, "small": min(n_truth, 10), "reference": min(n_truth, 100), "large": min(n_truth, 513),
         "all": n_truth if n_truth <= 2000 else min(n_truth, 50)}[k_kind]
    def answer(run):
        """The rows, or the text of the exception (match_maker.py:188-189 raises when fewer than k rows qualify: a 0/0 jaccard of a
        row whose columns all have zero IDF is NaN and qualifies for nothing)."""
        try:
            return run()
        except Exception as error:  # noqa: BLE001 - the reference's own exception type
            return str(error)
    index = ds.TruthIndex(problem["rowptr"], problem["truth_idx"], problem["idf32"], problem["sums32"])
    got = answer(lambda: index.top_k(problem["q_rowptr"], problem["q_cols"], problem["q_maxint"], k))
    expected = answer(lambda: oracle.jaccard_topk(problem["rowptr"], problem["truth_idx"], problem["idf32"], problem["sums32"],
                                                  problem["q_rowptr"], problem["q_cols"], problem["q_maxint"], k))
    index.close()
    if isinstance(got, str) or isinstance(expected, str):
        assert isinstance(got, str) and isinstance(expected, str) and "top_matches.shape[0] != self.top_n" in got and \
            "top_matches.shape[0] != self.top_n" in expected, (seed, n_truth, n_columns, shape, k, got, expected)
        return
    bad = np.nonzero((got != expected).any(axis=1))[0]
    assert bad.shape[0] == 0, (seed, n_truth, n_columns, shape, k, bad[:4], got[bad[:1]], expected[bad[:1]])


@settings(max_examples=_SOAK or 40, deadline=None, suppress_health_check=list(HealthCheck), derandomize=not _SOAK)
@given(seed=st.integers(0, 2 ** 31 - 1), longest=st.sampled_from([1, 3, 16, 33, 64, 65, 127, 128, 200, 255]),
       alphabet=st.sampled_from([2, 4, 38, 64, 200]), space_share=st.sampled_from([0.0, 0.1, 0.5, 1.0]),
       n_truth=st.sampled_from([1, 2, 30000, 50_000_000]))
def test_construct_features_equals_the_oracle_on_adversarial_pairs(oracle, seed, longest, alphabet, space_share, n_truth):
    """96 pairs per example through the 9-argument entry (staged path) AND the indexed entry: lengths 0..`longest` (both sides of
    the 32 / 64-character limits of the bit-parallel path and of the 255-character wrap), alphabets of 2..200 codes (codes >= 64 take
    the literal path), runs of spaces, word counts of 0 (log of infinity -> NaN ranks) and above n_truth."""
    import doppel_speller_amd as ds
    rng = np.random.RandomState(seed)
    n = 96
    q_len = rng.randint(0, longest + 1, n).astype(np.uint8)
    t_len = rng.randint(0, longest + 1, n).astype(np.uint8)
    def strings(lengths):
        enc = np.zeros((n, 255), dtype=np.uint8)
        for i, length in enumerate(lengths):
            codes = rng.randint(2, max(3, alphabet), length)
            codes[rng.rand(length) < space_share] = 1
            enc[i, :length] = codes
        return enc
    q_enc, t_enc = strings(q_len), strings(t_len)
    counts = rng.randint(0, 3, (n, 15)).astype(np.uint32) * rng.randint(1, 2 * max(n_truth, 2), (n, 15)).astype(np.uint32)
    with np.errstate(all="ignore"):
        expected = oracle.construct_features(q_len, t_len, q_enc, t_enc, counts, 1, n_truth)
        staged = np.zeros((n, ds.FEATURES_COUNT), dtype=np.float32)
        ds.construct_features(q_len, t_len, q_enc, t_enc, counts, 1, n_truth, None, staged)
        queries, truth = ds.TitleTable(q_enc, q_len), ds.TitleTable(t_enc, t_len, counts)
        pairs = np.arange(n, dtype=np.int32)
        indexed = ds.construct_features_indexed(queries, truth, pairs, pairs, 1, n_truth)
    for name, got in (("staged", staged), ("indexed", indexed)):
        bad = np.nonzero((got.view(np.uint32) != expected.view(np.uint32)).any(axis=1))[0]
        assert bad.shape[0] == 0, (name, seed, longest, alphabet, space_share, n_truth, bad[:4], got[bad[:1]], expected[bad[:1]])
