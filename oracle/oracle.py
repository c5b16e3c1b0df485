"""ctypes front-end of oracle/libdoppel_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The product package
(doppel-speller_amd) never does: it fails loudly when its HIP library is missing.

Each wrapper names the reference function it restates; `typing` is 'numba' (the specification) or 'numpy' (what the
golden vectors were captured with) -- see the header of doppel_oracle.c.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libdoppel_oracle.so")
_TYPING = {"numba": 0, "numpy": 1}
FEATURES_COUNT = 66
WORDS = 15
_lib = None


def build(force=False):
    """Compile the C restatement with gcc (idempotent)."""
    source = os.path.join(_HERE, "doppel_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(source):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libdoppel_oracle.so"])
    subprocess.check_call(["make", "-C", _HERE, "-s", "libdoppel_cpu.so"])   # the oracle behind the product's C ABI (make: up to date or rebuilt)
    return _LIB_PATH


CPU_ABI_PATH = os.path.join(_HERE, "libdoppel_cpu.so")


def _ptr(array, ctype):
    return array.ctypes.data_as(ctypes.POINTER(ctype))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        handle = ctypes.CDLL(_LIB_PATH)
        handle.ds_oracle_num_threads.restype = ctypes.c_int
        handle.ds_oracle_fast_arg_top_k.restype = ctypes.c_int64
        handle.ds_oracle_jaccard_topk.restype = ctypes.c_int64
        handle.ds_oracle_levenshtein_ratio.restype = ctypes.c_uint8
        handle.ds_oracle_feature_cells.restype = ctypes.c_int64
        _lib = handle
    return _lib


def num_threads():
    return int(lib().ds_oracle_num_threads())


def set_num_threads(threads):
    """OpenMP threads of the batched entry points (bench.py's cpu_baseline sweeps this)."""
    lib().ds_oracle_set_num_threads(ctypes.c_int(int(threads)))


def fast_jaccard(max_intersection_possible, columns, rowptr, truth_idx, idf32, sums32):
    """match_maker.py:16-50 -> float64[N]."""
    n = sums32.shape[0]
    columns = np.ascontiguousarray(columns, dtype=np.int32)
    scores = np.empty(n, dtype=np.float32)
    out = np.empty(n, dtype=np.float64)
    lib().ds_oracle_fast_jaccard(
        ctypes.c_int64(n), ctypes.c_double(float(max_intersection_possible)), _ptr(columns, ctypes.c_int32),
        ctypes.c_int64(columns.shape[0]), _ptr(rowptr, ctypes.c_int64), _ptr(truth_idx, ctypes.c_int32),
        _ptr(idf32, ctypes.c_float), _ptr(sums32, ctypes.c_float), _ptr(scores, ctypes.c_float),
        _ptr(out, ctypes.c_double))
    return out


def fast_arg_top_k(array, k, typing="numba"):
    """match_maker.py:53-71 -> int64[<=k] row indexes, descending index order."""
    array = np.ascontiguousarray(array, dtype=np.float64)
    out = np.empty(k, dtype=np.int64)
    found = lib().ds_oracle_fast_arg_top_k(_ptr(array, ctypes.c_double), ctypes.c_int64(array.shape[0]),
                                           ctypes.c_int32(k), ctypes.c_int32(_TYPING[typing]),
                                           _ptr(out, ctypes.c_int64))
    return out[:found]


def jaccard_topk(rowptr, truth_idx, idf32, sums32, q_rowptr, q_cols, q_maxint, k, typing="numba"):
    """match_maker.py:192-203 batched -> int32[Q, k] truth row indexes (descending row index per query)."""
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    truth_idx = np.ascontiguousarray(truth_idx, dtype=np.int32)
    idf32 = np.ascontiguousarray(idf32, dtype=np.float32)
    sums32 = np.ascontiguousarray(sums32, dtype=np.float32)
    q_rowptr = np.ascontiguousarray(q_rowptr, dtype=np.int64)
    q_cols = np.ascontiguousarray(q_cols, dtype=np.int32)
    q_maxint = np.ascontiguousarray(q_maxint, dtype=np.float64)
    n_queries = q_rowptr.shape[0] - 1
    out = np.empty((n_queries, k), dtype=np.int32)
    status = lib().ds_oracle_jaccard_topk(
        _ptr(rowptr, ctypes.c_int64), _ptr(truth_idx, ctypes.c_int32), _ptr(idf32, ctypes.c_float),
        _ptr(sums32, ctypes.c_float), ctypes.c_int64(sums32.shape[0]), _ptr(q_rowptr, ctypes.c_int64),
        _ptr(q_cols, ctypes.c_int32), _ptr(q_maxint, ctypes.c_double), ctypes.c_int64(n_queries), ctypes.c_int32(k),
        ctypes.c_int32(_TYPING[typing]), _ptr(out, ctypes.c_int32))
    if status != 0:
        raise Exception("top_matches.shape[0] != self.top_n")  # match_maker.py:188-189
    return out


def levenshtein_ratio(a, b, typing="numba"):
    """feature_engineering.py:25-63 on two uint8 code arrays."""
    a = np.ascontiguousarray(a, dtype=np.uint8)
    b = np.ascontiguousarray(b, dtype=np.uint8)
    return int(lib().ds_oracle_levenshtein_ratio(_ptr(a, ctypes.c_uint8), ctypes.c_int32(a.shape[0]),
                                                 _ptr(b, ctypes.c_uint8), ctypes.c_int32(b.shape[0]),
                                                 ctypes.c_int32(_TYPING[typing])))


def construct_features(title_len, truth_len, title_enc, truth_enc, counts, space_code, n_truth, typing="numba"):
    """feature_engineering.py:75-169 -> float32[n, 66]."""
    title_len = np.ascontiguousarray(title_len, dtype=np.uint8)
    truth_len = np.ascontiguousarray(truth_len, dtype=np.uint8)
    title_enc = np.ascontiguousarray(title_enc, dtype=np.uint8)
    truth_enc = np.ascontiguousarray(truth_enc, dtype=np.uint8)
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    n = title_len.shape[0]
    assert title_enc.shape == truth_enc.shape and title_enc.shape[0] == n and counts.shape == (n, WORDS)
    out = np.empty((n, FEATURES_COUNT), dtype=np.float32)
    lib().ds_oracle_construct_features(
        _ptr(title_len, ctypes.c_uint8), _ptr(truth_len, ctypes.c_uint8), _ptr(title_enc, ctypes.c_uint8),
        _ptr(truth_enc, ctypes.c_uint8), _ptr(counts, ctypes.c_uint32), ctypes.c_uint8(int(space_code)),
        ctypes.c_uint32(int(n_truth)), ctypes.c_int64(n), ctypes.c_int64(title_enc.shape[1]),
        ctypes.c_int32(_TYPING[typing]), _ptr(out, ctypes.c_float))
    return out


def feature_cells(title_len, truth_len, title_enc, truth_enc, space_code):
    """DP cells visited per pair (SURVEY.md section 8d work formula)."""
    title_enc = np.ascontiguousarray(title_enc, dtype=np.uint8)
    truth_enc = np.ascontiguousarray(truth_enc, dtype=np.uint8)
    out = np.empty(title_enc.shape[0], dtype=np.int64)
    for i in range(title_enc.shape[0]):
        out[i] = lib().ds_oracle_feature_cells(
            ctypes.c_uint8(int(title_len[i])), ctypes.c_uint8(int(truth_len[i])),
            _ptr(title_enc[i], ctypes.c_uint8), _ptr(truth_enc[i], ctypes.c_uint8), ctypes.c_uint8(int(space_code)))
    return out


def levenshtein_ratio_rounded(a, b):
    """common.py:161-162 (python-Levenshtein ratio, restated -- parity unpinned) on two uint8 code arrays."""
    a = np.ascontiguousarray(a, dtype=np.uint8)
    b = np.ascontiguousarray(b, dtype=np.uint8)
    return int(lib().ds_oracle_levenshtein_ratio_rounded(_ptr(a, ctypes.c_uint8), ctypes.c_int32(a.shape[0]),
                                                         _ptr(b, ctypes.c_uint8), ctypes.c_int32(b.shape[0])))


def close_ratios(x_len, y_len, x_enc, y_enc, space_code, sort_key, threshold=94):
    """Prediction._get_levenshtein_ratio (predict.py:140-156) for n padded pairs -> uint8[n]."""
    x_len = np.ascontiguousarray(x_len, dtype=np.uint8)
    y_len = np.ascontiguousarray(y_len, dtype=np.uint8)
    x_enc = np.ascontiguousarray(x_enc, dtype=np.uint8)
    y_enc = np.ascontiguousarray(y_enc, dtype=np.uint8)
    sort_key = np.ascontiguousarray(sort_key, dtype=np.uint8)
    assert sort_key.shape == (256,) and x_enc.shape == y_enc.shape
    out = np.empty(x_len.shape[0], dtype=np.uint8)
    lib().ds_oracle_close_ratios(_ptr(x_len, ctypes.c_uint8), _ptr(y_len, ctypes.c_uint8), _ptr(x_enc, ctypes.c_uint8),
                                 _ptr(y_enc, ctypes.c_uint8), ctypes.c_int64(x_len.shape[0]),
                                 ctypes.c_int64(x_enc.shape[1]), ctypes.c_uint8(int(space_code)),
                                 _ptr(sort_key, ctypes.c_uint8), ctypes.c_int32(int(threshold)),
                                 _ptr(out, ctypes.c_uint8))
    return out


def forest_predict(forest, rows):
    """xgboost binary:logistic prediction restated (parity unpinned): (margins float32[n], probabilities float32[n]).
    forest: dict with feature/yes/no/missing int32[nodes], threshold float32[nodes], tree_offsets int64[T + 1],
    base_margin float."""
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    n = rows.shape[0]
    margins = np.empty(n, dtype=np.float32)
    probabilities = np.empty(n, dtype=np.float32)
    arrays = {name: np.ascontiguousarray(forest[name], dtype=dtype) for name, dtype in
              (("feature", np.int32), ("threshold", np.float32), ("yes", np.int32), ("no", np.int32),
               ("missing", np.int32), ("tree_offsets", np.int64))}
    lib().ds_oracle_forest_predict(
        _ptr(arrays["feature"], ctypes.c_int32), _ptr(arrays["threshold"], ctypes.c_float),
        _ptr(arrays["yes"], ctypes.c_int32), _ptr(arrays["no"], ctypes.c_int32),
        _ptr(arrays["missing"], ctypes.c_int32), _ptr(arrays["tree_offsets"], ctypes.c_int64),
        ctypes.c_int32(arrays["tree_offsets"].shape[0] - 1), ctypes.c_float(float(forest["base_margin"])),
        _ptr(rows, ctypes.c_float), ctypes.c_int64(n), ctypes.c_int64(rows.shape[1]),
        _ptr(margins, ctypes.c_float), _ptr(probabilities, ctypes.c_float))
    return margins, probabilities


def remaining_pairs(best_row, rows):
    """predict.py:172-183 restated in NumPy: the (query row, truth row) pairs of the queries the fuzzy step did NOT
    match (best_row < 0), in the order of the reference's `remaining` frame (query-major, candidates in order)."""
    best_row, rows = np.asarray(best_row), np.asarray(rows)
    kept = np.nonzero(best_row < 0)[0]
    k = rows.shape[1]
    return np.repeat(kept, k).astype(np.int32), rows[kept].reshape(-1).astype(np.int32)


def select_matches(pair_q, pair_t, predictions, k, threshold=0.9):
    """predict.py:246-252 (+ :158-161) restated: per query (k consecutive pairs) the rows holding the maximum
    prediction, of those the ones above the threshold, and a match only when exactly one row is left."""
    pair_q, pair_t = np.asarray(pair_q).reshape(-1, k), np.asarray(pair_t).reshape(-1, k)
    predictions = np.asarray(predictions, dtype=np.float32).reshape(-1, k)
    best = predictions.max(axis=1, keepdims=True)
    holds = (predictions == best) & (predictions > np.float32(threshold))
    single = holds.sum(axis=1) == 1
    where = holds.argmax(axis=1)
    match = np.where(single, pair_t[np.arange(pair_t.shape[0]), where], -1).astype(np.int32)
    return pair_q[:, 0].astype(np.int32), match


def transform_title(title, n_grams=3, max_characters=255):
    """common.py:20-47 restated line by line (warnings not reproduced); pinned by tests/golden/transform_title.json."""
    import re
    import unicodedata
    text = unicodedata.normalize('NFD', title)                                           # :25
    text = text.encode('ascii', 'ignore').decode('utf-8').lower().replace('-', ' ')      # :26
    text = ''.join(re.findall(r'[a-zA-Z0-9\s]', text))                                   # :17, :28
    text = re.sub(r' +', ' ', text).strip()                                              # :16, :30
    number_of_characters = len(text)
    text = text[:max_characters].strip()                                                 # :32
    if number_of_characters < n_grams:
        return text.rjust(n_grams, '0')                                                  # :38
    return text
