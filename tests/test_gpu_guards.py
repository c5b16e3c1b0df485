"""Guards for the two GPU-side failures of round 2 (causes: profiles/r03_failure_causes.md).

(a) r02t / r02u -- the collect sweep's one-compare row test let through (i) padding / idle-lane entries, whose local row
    lies behind the tile (an out-of-range gather in the refinement: the abort), and (ii) the SECOND posting of a row whose
    score another lane had already taken (score 0 + the skipped columns' mass >= need: the row was appended twice and came
    out twice -- rows "not strictly descending").  The guard runs the fast kernel of a -DDS_BOUNDS_CHECK build (every
    data-dependent global index checked, first violation recorded in ds_jaccard_sync stats[28..30]) on an index built so
    that most candidate rows receive postings from two or three essential columns inside one sparse tile.
(b) the epoch redo -- an overflow of the candidate buffer inside an epoch's later tiles processes the epoch again under a
    tighter threshold; round 2's first version reset its retry counter on the epoch's first (successful) tile and never
    ended.  `sparse_redos` (stats[31]) now counts the redos: one scenario must redo and succeed, one (a thousand non-twin
    rows tied within 1e-6) must give up after its retries and hand the query to the literal kernel.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NARROW_TILE = 12288


def build_index(n_rows, columns, extra_sums=None):
    """CSR inverted index from {column id: sorted row array}; idf = ln(N / df) as match_maker.py:135-142; sums32 = the
    sequential float32 sum of a row's idf values in ascending column order (+ extra_sums[row], the C ABI allows more)."""
    n_columns = max(columns) + 1
    lengths = np.zeros(n_columns, dtype=np.int64)
    for column, rows in columns.items():
        lengths[column] = len(rows)
    rowptr = np.concatenate(([0], np.cumsum(lengths))).astype(np.int64)
    truth_idx = np.concatenate([np.asarray(columns.get(c, []), dtype=np.int32) for c in range(n_columns)]).astype(np.int32)
    idf64 = np.array([np.log(n_rows / max(1, lengths[c])) for c in range(n_columns)])
    idf32 = idf64.astype(np.float32)
    sums32 = np.zeros(n_rows, dtype=np.float32)
    for column in range(n_columns):              # ascending column order, float32 adds
        rows = truth_idx[rowptr[column]:rowptr[column + 1]]
        sums32[rows] = sums32[rows] + idf32[column]
    if extra_sums is not None:
        sums32 = (sums32 + extra_sums.astype(np.float32)).astype(np.float32)
    return rowptr, truth_idx, idf32, idf64, sums32


def queries_of(column_lists, idf32, idf64):
    kept = [np.array(sorted(c for c in columns if idf32[c] != 0), dtype=np.int32) for columns in column_lists]
    q_rowptr = np.concatenate(([0], np.cumsum([len(c) for c in kept]))).astype(np.int64)
    q_cols = np.concatenate(kept).astype(np.int32)
    q_maxint = np.array([float(sum(float(idf64[c]) for c in columns)) for columns in kept])
    return q_rowptr, q_cols, q_maxint


_CHILD = r"""
import json, sys
import numpy as np
sys.path.insert(0, %(root)r)
sys.path.insert(0, %(tests)r)
import doppel_speller_amd as ds
from oracle import oracle
import test_gpu_guards
oracle.build()
rowptr, truth_idx, idf32, sums32, q_rowptr, q_cols, q_maxint, k = getattr(test_gpu_guards, %(problem)r)()
index = ds.TruthIndex(rowptr, truth_idx, idf32, sums32)
rows = index.top_k(q_rowptr, q_cols, q_maxint, k)
d_rowptr, d_cols, d_maxint = (ds._lib.DeviceArray.from_host(x) for x in (q_rowptr, q_cols, q_maxint))
d_rows = ds._lib.DeviceArray(rows.shape, np.int32)
index.top_k_device(d_rowptr.ptr, d_cols.ptr, d_maxint.ptr, rows.shape[0], k, d_rows.ptr)
stats = index.sync()
expected = oracle.jaccard_topk(rowptr, truth_idx, idf32, sums32, q_rowptr, q_cols, q_maxint, k)
print(json.dumps({"equal": bool(np.array_equal(rows, expected)), "device_equal": bool(np.array_equal(d_rows.to_host(), expected)),
                  "descending": bool((np.diff(rows.astype(np.int64), axis=1) < 0).all()),
                  "bounds_record": [int(x) for x in stats["bounds_record"]], "sparse_tiles": int(stats["sparse_tiles"]),
                  "dense_queries": int(stats["dense_queries"]), "error_queries": int(stats["error_queries"]),
                  "sparse_redos": int(stats["sparse_redos"]), "tiles": int(index.info()["tiles"]), "library": ds._lib.library_path()}))
"""


def second_posting_problem():
    """48 queries over 4 narrow tiles.  Twelve rare columns co-occur heavily (every "cluster" row holds three or four of
    them), six dense columns are what a threshold lets the kernel skip, every row carries three filler columns (no two
    rows are twins)."""
    rng = np.random.RandomState(31)
    n_rows = 4 * NARROW_TILE
    columns = {}
    for dense in range(6):                                       # signature-bearing, skipped after the first threshold
        columns[dense] = np.sort(rng.choice(n_rows, int(0.4 * n_rows), replace=False))
    rare = list(range(6, 18))
    cluster_rows = np.sort(rng.choice(n_rows, 2400, replace=False))      # spread over all four tiles
    members = {c: [] for c in rare}
    for row in cluster_rows:
        for c in rng.choice(rare, rng.randint(3, 5), replace=False):
            members[c].append(row)
    for c in rare:
        lonely = rng.choice(n_rows, 150, replace=False)                  # rows that hold only this rare column
        columns[c] = np.unique(np.concatenate((np.array(members[c], dtype=np.int64), lonely)))
    fillers = 18 + rng.randint(0, 3000, (n_rows, 3))
    for j in range(3):
        for column in np.unique(fillers[:, j]):
            rows = np.nonzero(fillers[:, j] == column)[0]
            columns[int(column)] = np.unique(np.concatenate((columns.get(int(column), np.zeros(0, np.int64)), rows)))
    rowptr, truth_idx, idf32, idf64, sums32 = build_index(n_rows, columns)
    query_columns = [list(range(6)) + list(rng.choice(rare, 4, replace=False)) for _ in range(48)]
    q_rowptr, q_cols, q_maxint = queries_of(query_columns, idf32, idf64)
    return rowptr, truth_idx, idf32, sums32, q_rowptr, q_cols, q_maxint, 10


def _under_the_bounds_checking_build(problem):
    from doppel_speller_amd import _lib
    variant = _lib.build_library(variant="boundscheck")      # built by __graft_entry__.build(); rebuilt here if stale
    env = dict(os.environ, DS_LIBRARY=variant)
    script = _CHILD % {"root": ROOT, "tests": os.path.join(ROOT, "tests"), "problem": problem}
    result = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=600)
    assert result.returncode == 0, result.stderr[-3000:]
    outcome = json.loads(result.stdout.strip().splitlines()[-1])
    assert outcome["library"] == variant
    return outcome


def test_second_posting_of_a_taken_row_under_the_bounds_checking_build():
    outcome = _under_the_bounds_checking_build("second_posting_problem")
    assert outcome["bounds_record"] == [0, 0, 0], outcome           # no data-dependent global index left its array
    assert outcome["equal"] and outcome["device_equal"] and outcome["descending"], outcome
    assert outcome["sparse_tiles"] > 48 and outcome["dense_queries"] == 0 and outcome["error_queries"] == 0, outcome


def descending_epochs_problem():
    """13 narrow tiles (three or more list-pointer blocks for queries of 40 columns and up), rows of 3..100 columns: in the
    internal sums32 order the long rows are the last tiles.  Three families of queries, top-100 (a weak cut: the sweeps run
    far): 64..128 random columns -- they start in the LAST tiles and descend towards tile 0 with a pointer block (span 1..3)
    SHORTER than an epoch of 4 tiles; 30..50 columns (span 4..7); 8..16 columns of a short row -- they start near tile 0 with a
    span (15..31) LONGER than the tiles below the start.  The faults of 17:47 in round 3 lived in exactly these two corners
    of `block_start = max(0, min(b, epoch_last - span + 1))` (profiles/r04_failure_causes.md (d))."""
    rng = np.random.RandomState(1747)
    n_rows, n_columns = 13 * NARROW_TILE, 3000
    per_row = rng.randint(3, 101, n_rows)
    rows = np.repeat(np.arange(n_rows, dtype=np.int64), per_row)
    # column popularity: a few dense columns (signature bits, skipped under a threshold), a long flat tail
    weights = 1.0 / (np.arange(n_columns) + 20.0)
    cols = rng.choice(n_columns, rows.shape[0], p=weights / weights.sum())
    pairs = np.unique(cols * n_rows + rows)                    # (column, row) ascending, duplicates within a row dropped
    cols, rows = pairs // n_rows, pairs % n_rows
    lengths = np.bincount(cols, minlength=n_columns)
    rowptr = np.concatenate(([0], np.cumsum(lengths))).astype(np.int64)
    truth_idx = rows.astype(np.int32)
    idf64 = np.log(n_rows / np.maximum(lengths, 1))
    idf32 = idf64.astype(np.float32)
    sums32 = np.zeros(n_rows, dtype=np.float32)
    for column in range(n_columns):                              # ascending column order, float32 adds
        members = truth_idx[rowptr[column]:rowptr[column + 1]]
        sums32[members] = sums32[members] + idf32[column]
    by_row = np.argsort(rows, kind="stable")
    row_start = np.concatenate(([0], np.cumsum(np.bincount(rows, minlength=n_rows))))
    short_rows = np.nonzero(np.bincount(rows, minlength=n_rows) <= 16)[0]
    query_columns = []
    for _ in range(24):
        query_columns.append(rng.choice(n_columns, rng.randint(64, 129), replace=False).tolist())
    for _ in range(24):
        query_columns.append(rng.choice(n_columns, rng.randint(30, 51), replace=False).tolist())
    for _ in range(24):
        row = short_rows[rng.randint(short_rows.shape[0])]
        own = cols[by_row[row_start[row]:row_start[row + 1]]].tolist()
        query_columns.append(sorted(set(own + rng.choice(n_columns, 6, replace=False).tolist())))
    q_rowptr, q_cols, q_maxint = queries_of(query_columns, idf32, idf64)
    return rowptr, truth_idx, idf32, sums32, q_rowptr, q_cols, q_maxint, 100


def test_descending_epochs_pointer_blocks_under_the_bounds_checking_build():
    outcome = _under_the_bounds_checking_build("descending_epochs_problem")
    assert outcome["tiles"] == 13, outcome
    assert outcome["bounds_record"] == [0, 0, 0], outcome           # list pointers, quads, rows: every index inside its array
    assert outcome["equal"] and outcome["device_equal"] and outcome["descending"], outcome
    assert outcome["sparse_tiles"] > 72 and outcome["error_queries"] == 0, outcome


def _redo_problem(tied):
    """6 narrow tiles, k = 100.  Tile 0 holds 300 rows with column C (they set the first threshold); tile 2 -- the second
    tile of the epoch of sparse tiles 1..4 -- holds 1000 rows with column A whose jaccard beats every C row: more than
    the 832-entry candidate buffer takes.  tied = False: their values are all different, a tighter threshold prunes them
    and the repeated epoch fits.  tied = True: they are equal (same sums32, a different filler column each, so no
    twins): no threshold separates them and the fast kernel must give up after its retries."""
    rng = np.random.RandomState(47)
    n_rows = 6 * NARROW_TILE
    a_rows = np.sort(2 * NARROW_TILE + rng.choice(NARROW_TILE, 1000, replace=False))
    c_rows = np.sort(rng.choice(NARROW_TILE, 300, replace=False))
    columns = {0: a_rows, 1: c_rows}
    for i, row in enumerate(a_rows):                                    # a filler column of its own: no twins
        columns[2 + i] = np.array([row])
    extra = np.zeros(n_rows, dtype=np.float64)
    extra[c_rows] = 60.0 + 0.01 * np.arange(300)                        # weak rows: jaccard ~ 0.09
    if not tied:
        extra[a_rows] = 0.005 * np.arange(1000)                         # strong rows, all different
    rowptr, truth_idx, idf32, idf64, sums32 = build_index(n_rows, columns, extra)
    q_rowptr, q_cols, q_maxint = queries_of([[0, 1]], idf32, idf64)
    return rowptr, truth_idx, idf32, sums32, q_rowptr, q_cols, q_maxint, 100


@pytest.mark.parametrize("tied", [False, True])
def test_candidate_overflow_inside_an_epoch_is_redone_and_counted(oracle, tied, monkeypatch):
    import doppel_speller_amd as ds
    # the scenario places its rows tile by tile: the index keeps the caller's row order (ds_index_create sorts the rows by
    # sums32 otherwise, and a query then starts at the tile of its own sums32)
    monkeypatch.setenv("DS_SORT_ROWS", "0")
    rowptr, truth_idx, idf32, sums32, q_rowptr, q_cols, q_maxint, k = _redo_problem(tied)
    index = ds.TruthIndex(rowptr, truth_idx, idf32, sums32)
    assert index.info()["tile_rows"] == NARROW_TILE
    d_rowptr, d_cols, d_maxint = (ds._lib.DeviceArray.from_host(x) for x in (q_rowptr, q_cols, q_maxint))
    d_rows = ds._lib.DeviceArray((1, k), np.int32)
    index.top_k_device(d_rowptr.ptr, d_cols.ptr, d_maxint.ptr, 1, k, d_rows.ptr)
    stats = index.sync()
    rows = d_rows.to_host()
    expected = oracle.jaccard_topk(rowptr, truth_idx, idf32, sums32, q_rowptr, q_cols, q_maxint, k)
    assert np.array_equal(rows, expected)
    assert stats["error_queries"] == 0
    print("tied" if tied else "distinct", "sparse_redos", stats["sparse_redos"], stats["dense_reasons"])
    if tied:     # no threshold and no admission floor separates a thousand equal rows: the literal kernel answers
        assert stats["sparse_redos"] >= 1 and stats["dense_queries"] == 1
        assert stats["dense_reasons"]["overflow_sparse"] + stats["dense_reasons"]["ties"] == 1
        if stats["dense_reasons"]["overflow_sparse"] == 1:
            # the give-up bound itself (`++sparse_retries > 5`): five redos are counted, the sixth overflow hands the query over.
            # (With the admission floor the bound is a safety net: scripts/redo_bound_probe.py -- up to 10,000 distinct rows above
            # the cut in one tile, rising or falling with the row index -- ends after 1..3 redos or as "ties"; none reaches it.)
            assert stats["sparse_redos"] == 5, stats["sparse_redos"]
        else:   # the admission floor's check ended it first ("ties": the k best themselves tie at the pivot): fewer redos
            assert stats["sparse_redos"] <= 5, stats["sparse_redos"]
    else:        # the epoch is repeated under the tightened threshold and the admission floor, and fits
        assert 1 <= stats["sparse_redos"] <= 3 and stats["dense_queries"] == 0

    # the same problems with the rows in sums32 order (the default): whatever path they take, the answers are the reference's
    monkeypatch.setenv("DS_SORT_ROWS", "1")
    index = ds.TruthIndex(rowptr, truth_idx, idf32, sums32)
    assert np.array_equal(index.top_k(q_rowptr, q_cols, q_maxint, k), expected)


def test_near_ties_beyond_the_literal_kernels_buffer_are_answered_by_the_row_scan(oracle):
    """More rows within 1e-6 of the k-th value than the literal kernel's LDS buffer holds (3072), none of them twins and
    none k-dominated: the values RISE towards lower row indexes in steps of 6e-14.  The fast kernel hands the query over
    (ties), the streaming selection cannot compact it, and ds_jaccard_sync answers it with the reference's own method --
    the whole float64 jaccard row in an HBM scratch vector (round 2 returned DS_E_INTERNAL here)."""
    import doppel_speller_amd as ds
    n_rows, k = 30000, 10
    band = np.arange(10000, 14500)
    columns = {0: band}
    filler = np.setdiff1d(np.arange(n_rows), band)
    for j, part in enumerate(np.array_split(filler, 50)):
        columns[1 + j] = part
    rowptr, truth_idx, idf32, idf64, _ = build_index(n_rows, columns)
    sums32 = np.full(n_rows, 3.0, dtype=np.float32)
    sums32[band] = (np.float32(2.0).view(np.uint32) + np.arange(band.shape[0], dtype=np.uint32)).view(np.float32)  # one ulp apart, >= the idf of column 0
    q_rowptr = np.array([0, 1], dtype=np.int64)
    q_cols = np.array([0], dtype=np.int32)
    q_maxint = np.array([1000.0])
    expected = oracle.jaccard_topk(rowptr, truth_idx, idf32, sums32, q_rowptr, q_cols, q_maxint, k)
    assert np.array_equal(expected[0], np.arange(14499, 14489, -1))          # the k largest row indexes of the band
    index = ds.TruthIndex(rowptr, truth_idx, idf32, sums32)
    rows = index.top_k(q_rowptr, q_cols, q_maxint, k)
    assert np.array_equal(rows, expected)
    assert index.status(1)[0] == 1          # answered (literal path), not an error
    # the same through the device entry points: the resolution happens inside ds_jaccard_sync
    d_rowptr, d_cols, d_maxint = (ds._lib.DeviceArray.from_host(x) for x in (q_rowptr, q_cols, q_maxint))
    d_rows = ds._lib.DeviceArray((1, k), np.int32)
    index.top_k_device(d_rowptr.ptr, d_cols.ptr, d_maxint.ptr, 1, k, d_rows.ptr)
    stats = index.sync()
    assert stats["error_queries"] == 0 and stats["dense_queries"] == 1 and np.array_equal(d_rows.to_host(), expected)
