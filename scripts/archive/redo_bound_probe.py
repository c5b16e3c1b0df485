"""Finds a scenario that drives the fast kernel to its give-up bound (six overflows of one epoch in a row): prints the redo count and
the hand-over reasons for a few shapes of "many distinct rows above the cut in the second tile of an epoch" (tests/test_gpu_guards.py
pins the one that reaches `overflow_sparse`).  GPU box only; DS_SORT_ROWS=0 is set here."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["DS_SORT_ROWS"] = "0"
import doppel_speller_amd as ds          # noqa: E402
from oracle import oracle                # noqa: E402
from test_gpu_guards import NARROW_TILE, build_index, queries_of   # noqa: E402

oracle.build()
for strong, step, direction in ((1000, 0.005, 1), (3000, 0.002, 1), (3000, 0.002, -1), (6000, 0.001, 1), (6000, 0.001, -1),
                                (10000, 0.0005, 1), (10000, 0.0005, -1), (6000, 1e-5, 1), (10000, 1e-6, 1)):
    rng = np.random.RandomState(47)
    n_rows = 6 * NARROW_TILE
    a_rows = np.sort(2 * NARROW_TILE + rng.choice(NARROW_TILE, strong, replace=False))
    c_rows = np.sort(rng.choice(NARROW_TILE, 300, replace=False))
    columns = {0: a_rows, 1: c_rows}
    for i, row in enumerate(a_rows):
        columns[2 + i] = np.array([row])
    extra = np.zeros(n_rows, dtype=np.float64)
    extra[c_rows] = 60.0 + 0.01 * np.arange(300)
    ramp = step * np.arange(strong)
    extra[a_rows] = ramp if direction > 0 else ramp[::-1]
    rowptr, truth_idx, idf32, idf64, sums32 = build_index(n_rows, columns, extra)
    q_rowptr, q_cols, q_maxint = queries_of([[0, 1]], idf32, idf64)
    index = ds.TruthIndex(rowptr, truth_idx, idf32, sums32)
    d = [ds._lib.DeviceArray.from_host(x) for x in (q_rowptr, q_cols, q_maxint)]
    out = ds._lib.DeviceArray((1, 100), np.int32)
    index.top_k_device(d[0].ptr, d[1].ptr, d[2].ptr, 1, 100, out.ptr)
    stats = index.sync()
    expected = oracle.jaccard_topk(rowptr, truth_idx, idf32, sums32, q_rowptr, q_cols, q_maxint, 100)
    print(strong, step, direction, "redos", stats["sparse_redos"], stats["dense_reasons"], "equal", bool(np.array_equal(out.to_host(), expected)), flush=True)
