"""The fused device-resident hot path: Jaccard top-k -> construct_features on the surviving (query, truth) pairs.

Mirrors the two hot loops of Prediction.generate_test_predictions (predict.py:126-127 and :215-219) without the
host round trip between them: the top-k rows written by the Jaccard kernels are consumed in HBM by the feature kernel
(pair i = (query i // k, truth row rows[i])).  Inputs are uploaded once; `step()` only enqueues kernels.
"""
import ctypes

import numpy as np

from . import _lib
from .feature_engineering import FEATURES_COUNT, LEVENSHTEIN_RATIO_THRESHOLD, SORT_KEY, SPACE_CODE, TitleTable

PREDICTION_PROBABILITY_THRESHOLD = 0.9  # settings.py:76
from .match_maker import TruthIndex


class CandidatePipeline:
    def __init__(self, workload, k, device=0, q_begin=0, q_end=None, rows_ptr=None):
        """workload: an object with the fields of synth.make_workload.  Queries [q_begin, q_end) are this GPU's shard.
        rows_ptr: optional device pointer of an int32[q, k] buffer owned by the caller; allocated here when omitted."""
        from .distributed import slice_queries
        self.k = k
        self.device = device
        q_end = workload.n_queries if q_end is None else q_end
        self.n_queries = q_end - q_begin
        self.n_truth = workload.n_truth
        self.index = TruthIndex(workload.rowptr, workload.truth_idx, workload.idf32, workload.sums32, device)
        self.truth_titles = TitleTable(workload.t_enc, workload.t_len, workload.t_counts, device)
        self.query_titles = TitleTable(workload.q_enc[q_begin:q_end], workload.q_len[q_begin:q_end], None, device)
        rowptr, cols, maxint = slice_queries(workload.q_rowptr, workload.q_cols, workload.q_maxint, q_begin, q_end)
        self.d_rowptr = _lib.DeviceArray.from_host(rowptr, device)
        self.d_cols = _lib.DeviceArray.from_host(cols if cols.shape[0] else np.zeros(1, np.int32), device)
        self.d_maxint = _lib.DeviceArray.from_host(maxint, device)
        self._rows = None
        if rows_ptr is None:
            self._rows = _lib.DeviceArray((self.n_queries, k), np.int32, device)
            rows_ptr = self._rows.ptr
        self.rows_ptr = rows_ptr if isinstance(rows_ptr, ctypes.c_void_p) else ctypes.c_void_p(int(rows_ptr))
        self.d_features = _lib.DeviceArray((self.n_queries * k, FEATURES_COUNT), np.float32, device)
        self._close = None
        self._predictions = None
        self._pairs = None
        self._matches = None

    def enqueue_top_k(self, stream=None):
        self.index.top_k_device(self.d_rowptr.ptr, self.d_cols.ptr, self.d_maxint.ptr, self.n_queries, self.k,
                                self.rows_ptr, stream)

    def enqueue_features(self, stream=None):
        _lib.check(_lib.lib().ds_construct_features_indexed_device(
            self.query_titles.handle, self.truth_titles.handle, ctypes.c_void_p(0), self.rows_ptr, 0, self.k,
            SPACE_CODE, self.n_truth, self.n_queries * self.k, self.d_features.ptr, ctypes.c_void_p(stream or 0)),
            "ds_construct_features_indexed_device")

    def enqueue_close_matches(self, stream=None, threshold=LEVENSHTEIN_RATIO_THRESHOLD):
        """Next row f-1 in the same device-resident flow (predict.py:140-183): the fuzzy ratio of every (query,
        candidate) pair and the unique best candidate per query, read from the top-k rows in HBM."""
        if self._close is None:
            self._close = (_lib.DeviceArray((self.n_queries, self.k), np.uint8, self.device),
                           _lib.DeviceArray((self.n_queries,), np.int32, self.device),
                           _lib.DeviceArray.from_host(np.ascontiguousarray(SORT_KEY, dtype=np.uint8), self.device))
        ratios, best, sort_key = self._close
        _lib.check(_lib.lib().ds_close_matches_device(
            self.query_titles.handle, self.truth_titles.handle, self.rows_ptr, 0, self.k, self.n_queries, SPACE_CODE,
            sort_key.ptr, int(threshold), ratios.ptr, best.ptr, ctypes.c_void_p(stream or 0)),
            "ds_close_matches_device")

    def close_matches(self):
        """(ratios uint8[Q, k], best_row int32[Q]) of the last `enqueue_close_matches`."""
        ratios, best, _ = self._close
        return ratios.to_host(), best.to_host()

    def enqueue_remaining_pairs(self, stream=None):
        """Next row f-2 (predict.py:172-183): drop the queries the fuzzy step matched (`enqueue_close_matches` first)
        and compact the (query row, truth row) pairs of the others, in order, on the device."""
        if self._pairs is None:
            size = int(_lib.lib().ds_remaining_pairs_counts_size(self.n_queries))
            self._pairs = (_lib.DeviceArray((self.n_queries * self.k,), np.int32, self.device),
                           _lib.DeviceArray((self.n_queries * self.k,), np.int32, self.device),
                           _lib.DeviceArray((size,), np.int64, self.device))
        pair_q, pair_t, counts = self._pairs
        _lib.check(_lib.lib().ds_remaining_pairs_device(self._close[1].ptr, self.rows_ptr, self.n_queries, self.k, 0,
                                                        pair_q.ptr, pair_t.ptr, counts.ptr,
                                                        ctypes.c_void_p(stream or 0)), "ds_remaining_pairs_device")

    def remaining_counts(self, stream=None):
        """(remaining queries, pairs) of the last `enqueue_remaining_pairs` (synchronises the stream)."""
        _lib.check(_lib.lib().ds_stream_sync(ctypes.c_void_p(stream or 0), self.device), "sync")
        out = np.empty(2, dtype=np.int64)
        _lib.check(_lib.lib().ds_memcpy_d2h(_lib.pointer(out), self._pairs[2].ptr, 16, self.device), "d2h")
        return int(out[0]), int(out[1])

    def remaining_pairs(self, n_pairs):
        pair_q, pair_t, _ = self._pairs
        return pair_q.to_host()[:n_pairs], pair_t.to_host()[:n_pairs]

    def enqueue_features_remaining(self, n_pairs, stream=None):
        """construct_features on the compacted pair list only (predict.py:195-219), into the first n_pairs rows."""
        pair_q, pair_t, _ = self._pairs
        _lib.check(_lib.lib().ds_construct_features_indexed_device(
            self.query_titles.handle, self.truth_titles.handle, pair_q.ptr, pair_t.ptr, 0, self.k, SPACE_CODE,
            self.n_truth, n_pairs, self.d_features.ptr, ctypes.c_void_p(stream or 0)),
            "ds_construct_features_indexed_device")

    def enqueue_predict(self, model, stream=None, n_pairs=None):
        """Next row f-4: the tree ensemble (predict.py:229-234) on the feature matrix resident in HBM."""
        if self._predictions is None:
            self._predictions = _lib.DeviceArray((self.n_queries * self.k,), np.float32, self.device)
        n_pairs = self.n_queries * self.k if n_pairs is None else n_pairs
        model.predict_device(self.d_features.ptr, n_pairs, None, self._predictions.ptr, stream)

    def enqueue_select_matches(self, n_remaining, threshold=PREDICTION_PROBABILITY_THRESHOLD, stream=None):
        """predict.py:246-252 on the predictions of the compacted pairs: per remaining query the single pair with the
        maximum prediction above the threshold."""
        if self._matches is None:
            self._matches = (_lib.DeviceArray((self.n_queries,), np.int32, self.device),
                             _lib.DeviceArray((self.n_queries,), np.int32, self.device))
        pair_q, pair_t, _ = self._pairs
        _lib.check(_lib.lib().ds_select_matches_device(pair_q.ptr, pair_t.ptr, self._predictions.ptr, n_remaining, self.k,
                                                       float(threshold), self._matches[0].ptr, self._matches[1].ptr,
                                                       ctypes.c_void_p(stream or 0)), "ds_select_matches_device")

    def matches(self, n_remaining):
        """(query rows, matched truth row or -1) of the last `enqueue_select_matches`."""
        return self._matches[0].to_host()[:n_remaining], self._matches[1].to_host()[:n_remaining]

    def predictions(self, n_pairs=None):
        out = self._predictions.to_host()
        return out.reshape(self.n_queries, self.k) if n_pairs is None else out[:n_pairs]

    def step(self, stream=None):
        self.enqueue_top_k(stream)
        self.enqueue_features(stream)

    def sync(self, stream=None):
        return self.index.sync(stream)

    def rows(self):
        out = np.empty((self.n_queries, self.k), dtype=np.int32)
        _lib.check(_lib.lib().ds_memcpy_d2h(_lib.pointer(out), self.rows_ptr, out.nbytes, self.device), "d2h")
        return out

    def features_of(self, queries):
        """float32[len(queries) * k, 66]: the feature rows of the given queries' pairs (one small copy per query)."""
        queries = np.asarray(queries, dtype=np.int64)
        out = np.empty((queries.shape[0] * self.k, FEATURES_COUNT), dtype=np.float32)
        row_bytes = FEATURES_COUNT * 4 * self.k
        for i, q in enumerate(queries):
            source = ctypes.c_void_p(self.d_features.ptr.value + int(q) * row_bytes)
            _lib.check(_lib.lib().ds_memcpy_d2h(ctypes.c_void_p(out.ctypes.data + i * row_bytes), source, row_bytes,
                                                self.device), "d2h")
        return out

    def features(self, n_pairs=None, first_pair=0):
        """float32[n_pairs, 66] of the last `enqueue_features`, pairs [first_pair, first_pair + n_pairs) (default: all)."""
        if n_pairs is None:
            n_pairs = self.n_queries * self.k - first_pair
        out = np.empty((n_pairs, FEATURES_COUNT), dtype=np.float32)
        source = ctypes.c_void_p(self.d_features.ptr.value + first_pair * FEATURES_COUNT * 4)
        _lib.check(_lib.lib().ds_memcpy_d2h(_lib.pointer(out), source, out.nbytes, self.device), "d2h")
        return out
