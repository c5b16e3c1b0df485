"""Input classes of the Jaccard C ABI that the random generator of test_gpu_jaccard.py does not produce: zero-IDF columns,
one IDF for every column, queries that ARE truth rows (jaccard exactly 1) or subsets of one, 1 / 128 / 129 columns,
max_intersection_possible above the columns' total.  Every answer is compared with the oracle, bit for bit."""
import numpy as np
import pytest

import test_gpu_jaccard as T

pytestmark = pytest.mark.gpu
N, V = 50000, 800


def _check(oracle, p, ks, literal=None):
    import doppel_speller_amd as ds
    index = ds.TruthIndex(p["rowptr"], p["truth_idx"], p["idf32"], p["sums32"])
    for k in ks:
        got = index.top_k(p["q_rowptr"], p["q_cols"], p["q_maxint"], k)
        expected = oracle.jaccard_topk(p["rowptr"], p["truth_idx"], p["idf32"], p["sums32"], p["q_rowptr"], p["q_cols"],
                                       p["q_maxint"], k)
        bad = np.nonzero((got != expected).any(axis=1))[0]
        assert bad.shape[0] == 0, (k, bad[:5], got[bad[:1]], expected[bad[:1]])
        stats = index.sync()
        assert stats["error_queries"] == 0
        if literal is not None:
            assert stats["dense_queries"] == literal


def _requery(p, n_queries, pick):
    q_cols, q_rowptr, q_maxint = [], [0], []
    for q in range(n_queries):
        c = np.asarray(pick(q), np.int32)
        q_cols.append(c)
        q_rowptr.append(q_rowptr[-1] + c.shape[0])
        total = 0.0
        for g in c:
            total = total + float(p["idf32"][g])
        q_maxint.append(total)
    p = dict(p)
    p["q_rowptr"] = np.array(q_rowptr, np.int64)
    p["q_cols"] = np.concatenate(q_cols).astype(np.int32)
    p["q_maxint"] = np.array(q_maxint)
    return p


@pytest.fixture(scope="module")
def base():
    rng = np.random.RandomState(99)
    problem = T._random_problem(rng, N, V, 10)
    order = np.argsort(problem["truth_idx"], kind="stable")
    cols_sorted = np.repeat(np.arange(V), np.diff(problem["rowptr"]))[order]
    starts = np.searchsorted(problem["truth_idx"][order], np.arange(N + 1))
    return problem, (lambda t: cols_sorted[starts[t]:starts[t + 1]]), np.diff(starts)


def test_zero_idf_columns(oracle, base):
    problem, row_cols, _ = base
    rng = np.random.RandomState(1)
    p = dict(problem)
    p["idf32"] = problem["idf32"].copy()
    p["idf32"][:3] = 0.0   # a tri-gram that every title has: log(N / N)
    _check(oracle, _requery(p, 150, lambda q: np.unique(np.concatenate((row_cols(rng.randint(N)), [0, 1, 2])))),
           (1, 10, 100), literal=0)
    # nothing but zero-IDF columns: max_intersection_possible = 0, the literal kernel answers
    _check(oracle, _requery(p, 20, lambda q: [0, 1, 2][:1 + q % 3]), (1, 10), literal=20)


def test_one_idf_for_every_column(oracle, base):
    problem, row_cols, per_row = base
    rng = np.random.RandomState(2)
    p = dict(problem)
    p["idf32"] = np.full(V, 2.5, np.float32)
    sums = np.zeros(N, np.float32)
    for count in np.unique(per_row):
        total = np.float32(0)
        for _ in range(int(count)):
            total = np.float32(total + np.float32(2.5))
        sums[per_row == count] = total
    p["sums32"] = sums
    _check(oracle, _requery(p, 150, lambda q: row_cols(rng.randint(N))), (1, 10, 100), literal=0)


def test_queries_that_are_truth_rows_or_parts_of_one(oracle, base):
    problem, row_cols, _ = base
    rng = np.random.RandomState(3)
    _check(oracle, _requery(problem, 200, lambda q: row_cols(rng.randint(N))), (1, 10, 100), literal=0)

    def part(q):
        c = row_cols(rng.randint(N))
        return c[rng.rand(c.shape[0]) < 0.5] if c.shape[0] > 1 else c
    _check(oracle, _requery(problem, 200, lambda q: part(q) if part(q).shape[0] else [5]), (1, 10, 100))


def test_column_counts_at_the_limits(oracle, base):
    problem, _, _ = base
    rng = np.random.RandomState(4)
    _check(oracle, _requery(problem, 200, lambda q: [rng.randint(V)]), (1, 10, 512))
    _check(oracle, _requery(problem, 40, lambda q: rng.choice(V, 128, replace=False)), (10, 100), literal=0)
    _check(oracle, _requery(problem, 10, lambda q: rng.choice(V, 129, replace=False)), (10,), literal=10)


def test_max_intersection_above_the_columns_total(oracle, base):
    """match_maker.py:197 sums the IDF of EVERY tri-gram of the query, seen in the truth set or not."""
    problem, row_cols, _ = base
    rng = np.random.RandomState(5)
    p = _requery(problem, 200, lambda q: row_cols(rng.randint(N)))
    p["q_maxint"] = p["q_maxint"] * rng.choice([1.0, 1.5, 3.0, 10.0], 200)
    _check(oracle, p, (1, 10, 100))
