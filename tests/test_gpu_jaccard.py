"""GPU parity of the Jaccard top-k path (through the C ABI) against the oracle and the golden vectors."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _random_problem(rng, n_truth, n_columns, n_queries, mean_cols=12, heavy=6, duplicates=0):
    """Random inverted index with a few heavy columns, and queries that reuse truth rows' columns."""
    rows, cols = [], []
    per_row = np.clip(rng.poisson(mean_cols, n_truth), 1, 60)
    for t in range(n_truth):
        c = set(rng.randint(heavy, n_columns, per_row[t]).tolist())
        for h in range(heavy):
            if rng.rand() < 0.3:
                c.add(h)
        rows.append(np.full(len(c), t))
        cols.append(np.array(sorted(c)))
    if duplicates:
        for t in range(1, duplicates):
            cols[t] = cols[0]
            rows[t] = np.full(len(cols[0]), t)
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    order = np.argsort(cols, kind="stable")
    truth_idx = rows[order].astype(np.int32)
    df = np.bincount(cols, minlength=n_columns)
    rowptr = np.concatenate(([0], np.cumsum(df))).astype(np.int64)
    idf64 = np.log(n_truth / np.maximum(df, 1))
    idf64[df == 0] = idf64.max()
    idf32 = idf64.astype(np.float32)
    sums32 = np.zeros(n_truth, dtype=np.float32)
    by_row = np.argsort(rows, kind="stable")
    for r, c in zip(rows[by_row], cols[by_row]):
        sums32[r] = sums32[r] + idf32[c]
    q_cols, q_rowptr, q_maxint = [], [0], []
    for q in range(n_queries):
        base = cols[rows == rng.randint(n_truth)] if rng.rand() < 0.7 else np.array([], dtype=np.int64)
        extra = rng.randint(0, n_columns, rng.randint(1, 10))
        c = np.unique(np.concatenate((base[rng.rand(base.shape[0]) < 0.8], extra))).astype(np.int64)
        c = c[idf32[c] != 0]
        q_cols.append(c)
        q_rowptr.append(q_rowptr[-1] + c.shape[0])
        total = 0.0
        for value in idf64[c]:
            total = total + float(value)
        q_maxint.append(total)
    return dict(rowptr=rowptr, truth_idx=truth_idx, idf32=idf32, sums32=sums32,
                q_rowptr=np.array(q_rowptr, dtype=np.int64), q_cols=np.concatenate(q_cols).astype(np.int32),
                q_maxint=np.array(q_maxint, dtype=np.float64))


def _check(oracle, problem, k):
    import doppel_speller_amd as ds
    index = ds.TruthIndex(problem["rowptr"], problem["truth_idx"], problem["idf32"], problem["sums32"])
    got = index.top_k(problem["q_rowptr"], problem["q_cols"], problem["q_maxint"], k)
    expected = oracle.jaccard_topk(problem["rowptr"], problem["truth_idx"], problem["idf32"], problem["sums32"],
                                   problem["q_rowptr"], problem["q_cols"], problem["q_maxint"], k)
    bad = np.nonzero((got != expected).any(axis=1))[0]
    assert bad.shape[0] == 0, (bad[:10], got[bad[:2]], expected[bad[:2]])
    return index


def test_golden_fixture(oracle, golden_match_maker):
    import doppel_speller_amd as ds
    g = golden_match_maker
    index = ds.TruthIndex(g["rowptr"], g["truth_idx"], g["idf32"], g["sums32"])
    for k in (10, 100):
        got = index.top_k(g["q_rowptr"], g["q_cols"], g["q_maxint"], k)
        ok = g[f"margin_ok_k{k}"]
        assert np.array_equal(got[ok], g[f"rows_k{k}"][ok])          # captured from the reference
        expected = oracle.jaccard_topk(g["rowptr"], g["truth_idx"], g["idf32"], g["sums32"], g["q_rowptr"],
                                       g["q_cols"], g["q_maxint"], k)
        assert np.array_equal(got, expected)                          # every vector, specification typing
        assert np.array_equal(g["title_id"][got[ok]], g[f"ids_k{k}"][ok])


@pytest.mark.parametrize("n_truth,k", [(1000, 10), (40000, 10), (70001, 50), (98304, 100)])
def test_random_against_oracle(oracle, n_truth, k):
    rng = np.random.RandomState(n_truth + k)
    problem = _random_problem(rng, n_truth, 3000, 96)
    index = _check(oracle, problem, k)
    assert index.sync()["error_queries"] == 0


def test_more_than_65536_columns_take_the_wide_forward_index(oracle):
    """The exact stage reads the forward index as uint16 columns while V <= 65536 (every tri-gram vocabulary); an index with
    more columns (possible through the C ABI) keeps int32 columns and the generic copy of the loop."""
    rng = np.random.RandomState(4242)
    problem = _random_problem(rng, 20000, 70000, 96, mean_cols=14)
    index = _check(oracle, problem, 10)
    info = index.info()
    assert info["n_columns"] == 70000 and info["forward_index_bytes"] == 4 * (info["n_truth"] + 1) + 4 * info["nnz"]
    assert index.sync()["error_queries"] == 0


@pytest.mark.parametrize("n_truth,k", [(60000, 200), (90000, 512)])
def test_large_k_without_sample_threshold(oracle, n_truth, k):
    """k above the sample-threshold limit (128): the first threshold comes from a buffer flood and its recovery."""
    rng = np.random.RandomState(n_truth + k)
    problem = _random_problem(rng, n_truth, 2000, 48, mean_cols=14)
    index = _check(oracle, problem, k)
    assert index.sync()["error_queries"] == 0


def test_heavy_queries_many_columns(oracle):
    """Queries with 65..128 columns exercise the upper half of the per-wave item map (lanes own columns j, j + 64)."""
    rng = np.random.RandomState(4242)
    problem = _random_problem(rng, 120000, 1500, 4, mean_cols=10)
    n_columns = problem["rowptr"].shape[0] - 1
    idf64 = problem["idf32"].astype(np.float64)
    q_cols = []
    for q in range(40):
        width = rng.randint(65, 129)
        c = np.sort(rng.choice(n_columns, width, replace=False)).astype(np.int64)
        q_cols.append(c[problem["idf32"][c] != 0])
    problem["q_rowptr"] = np.concatenate(([0], np.cumsum([len(c) for c in q_cols]))).astype(np.int64)
    problem["q_cols"] = np.concatenate(q_cols).astype(np.int32)
    maxint = []
    for c in q_cols:
        total = 0.0
        for value in idf64[c]:
            total = total + float(value)
        maxint.append(total)
    problem["q_maxint"] = np.array(maxint)
    index = _check(oracle, problem, 25)
    stats = index.sync()
    assert stats["error_queries"] == 0 and stats["dense_queries"] < 40  # handled by the fast kernel


def _tie_problem(rng, n_truth, duplicates):
    problem = _random_problem(rng, n_truth, 2000, 40, duplicates=duplicates)  # identical truth rows: massive ties
    # make several queries equal to the duplicated row so the ties sit at the top
    first = problem["truth_idx"] == 0
    cols0 = np.repeat(np.arange(problem["rowptr"].shape[0] - 1), np.diff(problem["rowptr"]))[first]
    q_cols = [problem["q_cols"][problem["q_rowptr"][q]:problem["q_rowptr"][q + 1]] for q in range(40)]
    for q in range(0, 40, 4):
        q_cols[q] = cols0.astype(np.int32)
    problem["q_rowptr"] = np.concatenate(([0], np.cumsum([len(c) for c in q_cols]))).astype(np.int64)
    problem["q_cols"] = np.concatenate(q_cols).astype(np.int32)
    idf64 = problem["idf32"].astype(np.float64)
    problem["q_maxint"] = np.array([float(np.sum(idf64[c])) for c in q_cols])
    return problem


@pytest.mark.parametrize("k", [1, 10, 100])
def test_ties_and_duplicates(oracle, k):
    """6000 identical truth rows (same columns, same sums32): the fast kernel keeps the k twins of largest row index
    (match_maker.py:71 returns the k largest indexes) and serves the tie queries itself, bit-exactly."""
    problem = _tie_problem(np.random.RandomState(5), 50000, 6000)
    index = _check(oracle, problem, k)
    stats = index.sync()
    assert stats["dense_reasons"]["ties"] == 0 and stats["dense_reasons"]["overflow_sparse"] == 0
    assert stats["dense_reasons"]["overflow_dense"] == 0 and stats["error_queries"] == 0


def test_twins_with_different_sums_are_different_classes(oracle):
    """Twin rows (same columns) whose sums32 differ are NOT interchangeable: their jaccard differs.  Half of the 6000
    twins get the next float32 up; both classes are long, rows of both can be among the answers."""
    problem = _tie_problem(np.random.RandomState(6), 50000, 6000)
    sums = problem["sums32"].copy()
    odd = np.arange(1, 6000, 2)
    sums[odd] = np.nextafter(sums[odd], np.float32(np.inf))
    problem["sums32"] = sums
    for k in (10, 100):
        index = _check(oracle, problem, k)
        assert index.sync()["dense_reasons"]["ties"] == 0


def test_near_ties_that_are_not_twins_still_exact(oracle):
    """Thousands of rows that tie without being twins (same matched columns, different other columns, equal sums32 by
    construction): the rank rule does not apply; whatever path serves them, the answer equals the oracle's."""
    rng = np.random.RandomState(8)
    problem = _random_problem(rng, 60000, 2000, 8)
    n_columns = problem["rowptr"].shape[0] - 1
    # rows 0..3999 share columns {10, 11, 12}; each also owns one of two filler columns with equal idf
    lists = [problem["truth_idx"][problem["rowptr"][g]:problem["rowptr"][g + 1]] for g in range(n_columns)]
    tied = np.arange(4000, dtype=np.int32)
    lists = [l[l >= 4000] for l in lists]
    for g in (10, 11, 12):
        lists[g] = np.concatenate((tied, lists[g]))
    lists[20], lists[21] = tied[::2], tied[1::2]
    problem["rowptr"] = np.concatenate(([0], np.cumsum([len(l) for l in lists]))).astype(np.int64)
    problem["truth_idx"] = np.concatenate(lists).astype(np.int32)
    idf32 = problem["idf32"].copy()
    idf32[21] = idf32[20]
    problem["idf32"] = idf32
    sums = problem["sums32"].copy()
    sums[:4000] = np.float32(idf32[10]) + np.float32(idf32[11]) + np.float32(idf32[12]) + np.float32(idf32[20])
    problem["sums32"] = sums
    q_cols = [np.array([10, 11, 12], dtype=np.int32)] * 4 + [np.array([10, 11, 12, 20], dtype=np.int32)] * 4
    problem["q_rowptr"] = np.concatenate(([0], np.cumsum([len(c) for c in q_cols]))).astype(np.int64)
    problem["q_cols"] = np.concatenate(q_cols).astype(np.int32)
    idf64 = idf32.astype(np.float64)
    problem["q_maxint"] = np.array([float(np.sum(idf64[c])) for c in q_cols])
    for k in (10, 50):
        _check(oracle, problem, k)


@pytest.mark.parametrize("geometry,tile_rows", [("wide", 28672), ("narrow", 12288)])
def test_both_geometries_on_reference_vectors(oracle, golden_match_maker_full, monkeypatch, geometry, tile_rows):
    """The kernels are compiled for two geometries (2 x 512 threads / 28672-row tiles, 4 x 256 / 12288); an index picks
    one by its size.  Forced here, each must reproduce the reference's answers on its whole example truth set and the
    oracle's on a random index with ties."""
    import doppel_speller_amd as ds
    monkeypatch.setenv("DS_GEOMETRY", geometry)
    g = golden_match_maker_full
    index = ds.TruthIndex(g["rowptr"], g["truth_idx"], g["idf32"], g["sums32"])
    assert index.info()["tile_rows"] == tile_rows
    for k in (10, 100):
        got = index.top_k(g["q_rowptr"], g["q_cols"], g["q_maxint"], k)
        ok = g[f"margin_ok_k{k}"]
        assert np.array_equal(got[ok], g[f"rows_k{k}"][ok])
    for k in (1, 10, 100, 512):
        stats = _check(oracle, _tie_problem(np.random.RandomState(5), 50000, 6000), k).sync()
        assert stats["error_queries"] == 0
        # the candidate buffers (768 / 1472 entries) are sized for the reference's top_n of 10 and 100 (settings.py:55-56);
        # at k = 512 a tie-heavy query may go to the literal kernel -- same answer.  (With the rows in sums32 order equal
        # rows are neighbours: they reach the buffer together instead of spread over the sweep.)
        assert stats["dense_reasons"]["ties"] == 0 or k == 512, (geometry, k, stats["dense_reasons"])
    problem = _random_problem(np.random.RandomState(99), 98304, 3000, 96)
    _check(oracle, problem, 25)


def test_edge_cases(oracle):
    rng = np.random.RandomState(11)
    problem = _random_problem(rng, 5000, 800, 8)
    n_columns = problem["rowptr"].shape[0] - 1
    unused = np.nonzero(np.diff(problem["rowptr"]) == 0)[0]
    q_cols = [
        np.array([], dtype=np.int32),                                  # empty query: fewer than k positives
        unused[:3].astype(np.int32) if unused.shape[0] >= 3 else np.array([], dtype=np.int32),  # unseen n-grams
        np.arange(0, 300, dtype=np.int32) % n_columns,                 # > 256 columns (not unique on purpose? no)
        np.array([5], dtype=np.int32),
    ]
    q_cols[2] = np.unique(q_cols[2]).astype(np.int32)
    wide = np.unique(rng.randint(0, n_columns, 400)).astype(np.int32)  # more than 256 distinct columns
    q_cols.append(wide)
    idf64 = problem["idf32"].astype(np.float64)
    problem["q_rowptr"] = np.concatenate(([0], np.cumsum([len(c) for c in q_cols]))).astype(np.int64)
    problem["q_cols"] = np.concatenate(q_cols).astype(np.int32)
    maxint = []
    for c in q_cols:
        total = 0.0
        for value in idf64[c]:
            total = total + float(value)
        maxint.append(total)
    maxint[0] = 1.0  # an empty query with a positive maxint (never produced by MatchMaker, still well defined)
    problem["q_maxint"] = np.array(maxint)
    for k in (1, 10, 100):
        _check(oracle, problem, k)


def test_k_larger_than_truth_raises(oracle):
    import doppel_speller_amd as ds
    rng = np.random.RandomState(3)
    problem = _random_problem(rng, 50, 100, 2)
    index = ds.TruthIndex(problem["rowptr"], problem["truth_idx"], problem["idf32"], problem["sums32"])
    with pytest.raises(Exception, match="top_matches.shape"):
        index.top_k(problem["q_rowptr"], problem["q_cols"], problem["q_maxint"], 51)
    assert index.top_k(problem["q_rowptr"], problem["q_cols"], problem["q_maxint"], 50).shape == (2, 50)


def test_match_maker_drop_in(golden_match_maker):
    """The reference's call surface on DataFrames.  The captured vocabulary order and every truth title's captured set
    iteration order are injected, so every array and every answer is EXACTLY the reference's."""
    import doppel_speller_amd as ds
    from conftest import golden_frames
    g = golden_match_maker
    data, truth, vocabulary = golden_frames(g)
    mm = ds.MatchMaker(data, truth, 10, vocabulary=vocabulary)
    assert mm.top_n == 10 and mm.number_of_truth_titles == 5000 and not hasattr(mm, "data")
    assert list(mm.truth_data.columns) == ["title_id"]
    assert np.array_equal(mm.index.idf32, g["idf32"]) and np.array_equal(mm.index.rowptr, g["rowptr"])
    assert np.array_equal(mm.index.truth_idx, g["truth_idx"])
    assert np.array_equal(mm._q_cols, g["q_cols"]) and np.array_equal(mm._q_maxint, g["q_maxint"])
    assert np.array_equal(mm.sums_matrix_truth.view(np.uint32), g["sums32"].view(np.uint32))
    for q in range(200):
        ids = mm.get_closest_matches(q)
        assert isinstance(ids, list) and len(ids) == 10
        assert ids == g["ids_k10"][q].tolist()   # margin_ok is all true at k = 10


def test_whole_example_truth_set(oracle, golden_match_maker_full):
    """The reference MatchMaker's answers on its whole example truth set (30,000 rows: crosses the 28,672-row tile
    boundary, so list-pointer spans, sparse tiles and MaxScore across tiles run on reference-captured vectors)."""
    import doppel_speller_amd as ds
    from conftest import golden_frames
    g = golden_match_maker_full
    index = ds.TruthIndex(g["rowptr"], g["truth_idx"], g["idf32"], g["sums32"])
    assert index.info()["tiles"] >= 2
    for k in (10, 100):
        got = index.top_k(g["q_rowptr"], g["q_cols"], g["q_maxint"], k)
        ok = g[f"margin_ok_k{k}"]
        assert np.array_equal(got[ok], g[f"rows_k{k}"][ok])                     # captured from the reference
        expected = oracle.jaccard_topk(g["rowptr"], g["truth_idx"], g["idf32"], g["sums32"], g["q_rowptr"],
                                       g["q_cols"], g["q_maxint"], k)
        assert np.array_equal(got, expected)                                     # every vector, specification typing
    stats = index.sync()
    assert stats["error_queries"] == 0 and stats["dense_queries"] <= 10        # served by the fast kernel
    # the drop-in surface on the same data: ids, not rows
    data, truth, vocabulary = golden_frames(g)
    mm = ds.MatchMaker(data, truth, 100, vocabulary=vocabulary)
    ok = g["margin_ok_k100"]
    for q in np.nonzero(ok)[0][::10]:
        assert mm.get_closest_matches(int(q)) == g["title_id"][g["rows_k100"][q]].tolist()


def test_match_maker_from_titles_equals_dataframe_build():
    """Next row f-3: MatchMaker.from_titles (native index build) answers like MatchMaker(data, truth) built from
    DataFrames with the same column order and the same per-title summation order."""
    import pandas as pd
    import doppel_speller_amd as ds
    from doppel_speller_amd import synth
    w = synth.make_workload(60000, 300, seed=9)
    truth, queries = synth._to_strings(w.t_flat, w.t_off), synth._to_strings(w.q_flat, w.q_off)

    def ordered(title):
        seen, out = set(), []
        for i in range(len(title) - 2):
            if title[i:i + 3] not in seen:
                seen.add(title[i:i + 3])
                out.append(title[i:i + 3])
        return out
    native = ds.MatchMaker.from_titles(queries, truth, 10, title_ids=w.title_id)
    data = pd.DataFrame({"n_grams": [ordered(t) for t in queries], "title_id": np.arange(len(queries))})
    frame = pd.DataFrame({"n_grams": [ordered(t) for t in truth], "title_id": w.title_id})
    vocabulary = [native.n_grams_decoding[g] for g in range(len(native.n_grams_decoding))]
    reference_style = ds.MatchMaker(data, frame, 10, vocabulary=vocabulary)
    assert np.array_equal(native.get_closest_matches_batch(), reference_style.get_closest_matches_batch())
    for row in (0, 17, 299):
        assert native.get_closest_matches(row) == reference_style.get_closest_matches(row)


def test_values_outside_the_bounds_assumptions_take_the_literal_kernel(oracle):
    """Negative IDF values (impossible from match_maker.py:135-142, possible through the C ABI) void the pruning
    bounds: the index then routes every query to the literal kernel and the results still match the oracle."""
    rng = np.random.RandomState(77)
    problem = _random_problem(rng, 20000, 600, 24)
    problem["idf32"] = problem["idf32"].copy()
    problem["idf32"][::7] *= -1.0
    index = _check(oracle, problem, 10)
    stats = index.sync()
    assert stats["dense_queries"] == 24 and stats["error_queries"] == 0


@pytest.mark.parametrize("k", [1, 10, 100, 512])
def test_literal_kernel_with_massive_ties(oracle, k):
    """The literal kernel's streaming selection (no N-vector): 6000 twin rows tied at the top and an index that forces
    every query onto the literal path (negative idf values).  Buffer overflow on the first tile, compaction by the
    running threshold and by the k-dominators rule, tiles finalised again -- the answers equal the oracle's."""
    problem = _tie_problem(np.random.RandomState(15), 90000, 6000)
    problem["idf32"] = problem["idf32"].copy()
    problem["idf32"][5::11] *= -1.0
    index = _check(oracle, problem, k)
    stats = index.sync()
    assert stats["dense_queries"] == 40 and stats["error_queries"] == 0


def test_literal_kernel_many_equal_values_that_are_not_twins(oracle):
    """4000 rows with the same jaccard for different reasons (not twins), literal path: rule (b) keeps the k largest
    indexes of the tie, the threshold semantics of match_maker.py:70-71 decide the rest."""
    rng = np.random.RandomState(18)
    problem = _random_problem(rng, 60000, 2000, 8)
    n_columns = problem["rowptr"].shape[0] - 1
    lists = [problem["truth_idx"][problem["rowptr"][g]:problem["rowptr"][g + 1]] for g in range(n_columns)]
    tied = np.arange(4000, dtype=np.int32) * 13 + 5          # spread over three tiles
    lists = [l[~np.isin(l, tied)] for l in lists]
    for g in (10, 11, 12):
        lists[g] = np.sort(np.concatenate((tied, lists[g])))
    lists[20], lists[21] = tied[::2], tied[1::2]
    problem["rowptr"] = np.concatenate(([0], np.cumsum([len(l) for l in lists]))).astype(np.int64)
    problem["truth_idx"] = np.concatenate(lists).astype(np.int32)
    idf32 = problem["idf32"].copy()
    idf32[21] = idf32[20]
    idf32[30::17] *= -1.0                                      # literal-only index
    problem["idf32"] = idf32
    sums = problem["sums32"].copy()
    sums[tied] = np.float32(idf32[10]) + np.float32(idf32[11]) + np.float32(idf32[12]) + np.float32(idf32[20])
    problem["sums32"] = sums
    q_cols = [np.array([10, 11, 12], dtype=np.int32)] * 4 + [np.array([10, 11, 12, 20], dtype=np.int32)] * 4
    problem["q_rowptr"] = np.concatenate(([0], np.cumsum([len(c) for c in q_cols]))).astype(np.int64)
    problem["q_cols"] = np.concatenate(q_cols).astype(np.int32)
    idf64 = idf32.astype(np.float64)
    problem["q_maxint"] = np.array([float(np.sum(idf64[c])) for c in q_cols])
    for k in (10, 50):
        index = _check(oracle, problem, k)
        assert index.sync()["dense_queries"] == 8


def test_inconsistent_sums_take_the_literal_kernel(oracle):
    """sums32 smaller than a row's own idf total (again only possible through the C ABI) is detected at index build."""
    rng = np.random.RandomState(78)
    problem = _random_problem(rng, 20000, 600, 24)
    problem["sums32"] = (problem["sums32"] * np.float32(0.5)).astype(np.float32)
    index = _check(oracle, problem, 10)
    assert index.sync()["dense_queries"] == 24


def test_small_max_intersection_takes_the_literal_kernel(oracle):
    """max_intersection_possible below the idf total of the query's columns (not what match_maker.py:197 computes)."""
    rng = np.random.RandomState(79)
    problem = _random_problem(rng, 30000, 600, 32)
    problem["q_maxint"] = problem["q_maxint"].copy()
    problem["q_maxint"][::2] *= 0.25
    index = _check(oracle, problem, 10)
    assert index.sync()["dense_queries"] >= 16


@pytest.mark.parametrize("log2_scale", [11, 20, -12])
def test_scaled_idf_values_stay_on_the_fast_kernel(oracle, log2_scale):
    """IDF values scaled by a power of two (possible through the C ABI; every float32 sum scales exactly, so the
    expected rows are those of the unscaled problem).  At 2^11 `sums32` reaches 10^4..10^5, beyond the range of the
    8-bit sums code stored with every posting, whose top value is reserved for padding entries -- such rows share the
    largest real code instead of becoming unreachable in the collect sweep; at 2^-12 every row has the smallest code."""
    rng = np.random.RandomState(4242)
    problem = _random_problem(rng, 40000, 900, 64)
    unscaled = oracle.jaccard_topk(problem["rowptr"], problem["truth_idx"], problem["idf32"], problem["sums32"],
                                   problem["q_rowptr"], problem["q_cols"], problem["q_maxint"], 10)
    scale = 2.0 ** log2_scale
    problem["idf32"] = problem["idf32"] * np.float32(scale)
    problem["sums32"] = problem["sums32"] * np.float32(scale)
    problem["q_maxint"] = problem["q_maxint"] * scale
    assert log2_scale != 11 or problem["sums32"].max() > 7936.0
    index = _check(oracle, problem, 10)
    stats = index.sync()
    assert stats["error_queries"] == 0 and stats["dense_queries"] <= 2
    got = index.top_k(problem["q_rowptr"], problem["q_cols"], problem["q_maxint"], 10)
    assert np.array_equal(got, unscaled)


@pytest.mark.parametrize("k", [1, 10, 50])
def test_a_column_listed_twice_in_a_query(oracle, k):
    """Impossible from match_maker.py:196, possible through the C ABI: the reference's loop adds such a column twice
    (so does the oracle).  The fast kernel's signature completion would count a skipped column once, so a pass after
    it hands every query with a repeated column to the literal kernel; clean queries of the same call stay fast."""
    import doppel_speller_amd as ds
    rng = np.random.RandomState(5)
    problem = _random_problem(rng, 60000, 700, 200, heavy=12)
    q_rowptr, q_cols, q_maxint, repeated = [0], [], [], []
    for q in range(200):
        c = problem["q_cols"][problem["q_rowptr"][q]:problem["q_rowptr"][q + 1]]
        if q % 4 == 3:   # every fourth query stays clean -- but lists its columns in random order (float32 sums follow it)
            listed = c.copy()
            rng.shuffle(listed)
        else:
            listed = np.concatenate((c, c[rng.rand(c.shape[0]) < 0.4], np.arange(12)[rng.rand(12) < 0.5],
                                     np.arange(12)[rng.rand(12) < 0.5]))
            rng.shuffle(listed)
        repeated.append(np.unique(listed).shape[0] != listed.shape[0])
        q_cols.append(listed.astype(np.int32))
        q_rowptr.append(q_rowptr[-1] + listed.shape[0])
        total = 0.0
        for g in listed:
            total = total + float(problem["idf32"][g])
        q_maxint.append(total)
    problem["q_rowptr"] = np.array(q_rowptr, np.int64)
    problem["q_cols"] = np.concatenate(q_cols)
    problem["q_maxint"] = np.array(q_maxint)
    index = _check(oracle, problem, k)
    status = index.status(200)
    assert np.array_equal(status == 1, np.array(repeated)) or (status[~np.array(repeated)] <= 1).all()
    assert (status[np.array(repeated)] == 1).all() and index.sync()["error_queries"] == 0


@pytest.mark.parametrize("order", ["0", "1"])
def test_work_queue_order_does_not_change_answers(oracle, order):
    """The fast kernel takes the queries with most columns first (a device-side counting sort, ds_index_option
    "query_order" = 0 switches it off): every query is answered exactly once either way, including queries without columns
    and with > 128 columns."""
    rng = np.random.RandomState(77)
    problem = _random_problem(rng, 30000, 1500, 700, mean_cols=12)
    n_columns = problem["rowptr"].shape[0] - 1
    q_cols = [problem["q_cols"][problem["q_rowptr"][q]:problem["q_rowptr"][q + 1]] for q in range(700)]
    q_cols[3] = np.zeros(0, dtype=np.int32)                                             # no columns at all
    wide = np.sort(rng.choice(n_columns, 200, replace=False)).astype(np.int32)          # more than the fast kernel takes
    q_cols[5] = wide[problem["idf32"][wide] != 0]
    problem["q_rowptr"] = np.concatenate(([0], np.cumsum([len(c) for c in q_cols]))).astype(np.int64)
    problem["q_cols"] = np.concatenate(q_cols).astype(np.int32)
    idf64 = problem["idf32"].astype(np.float64)
    maxint = []
    for c in q_cols:
        total = 0.0
        for value in idf64[c]:
            total = total + float(value)
        maxint.append(max(total, 1e-3))
    problem["q_maxint"] = np.array(maxint)
    import doppel_speller_amd as ds
    index = ds.TruthIndex(problem["rowptr"], problem["truth_idx"], problem["idf32"], problem["sums32"])
    index.option("query_order", int(order))
    got = index.top_k(problem["q_rowptr"], problem["q_cols"], problem["q_maxint"], 10)
    expected = oracle.jaccard_topk(problem["rowptr"], problem["truth_idx"], problem["idf32"], problem["sums32"],
                                   problem["q_rowptr"], problem["q_cols"], problem["q_maxint"], 10)
    assert np.array_equal(got, expected)
    assert index.sync()["error_queries"] == 0
