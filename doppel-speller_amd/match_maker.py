"""MatchMaker: the reference's call surface (doppelspeller/match_maker.py:74-203) over the HIP Jaccard top-k kernels.

`MatchMaker(data, truth_data, top_n)` takes the same two DataFrames as the reference (columns `n_grams` = set of
character n-grams, `title_id`), builds the same vocabulary / IDF table / query rows / truth inverted index, uploads
the index to HBM once (`TruthIndex`), and answers `get_closest_matches(row_number) -> list[title_id]`.  The
reference computes one query per call inside a Python dict comprehension (predict.py:126-127); here the first call
evaluates every row of `data` in one launch and later calls are cache reads.  `get_closest_matches_batch(rows)` is
the explicit batched form.

Host-side numerics follow the reference exactly (citations inline): IDF values are float64 `math.log`, stored as
float32 in the matrices; `sums_matrix_truth[t]` is a sequential float32 sum in the iteration order of the title's
n-gram `set`; `max_intersection_possible` is a sequential float64 sum in ascending column order.
"""
import ctypes
import logging
import math

import numpy as np

from . import _lib

LOGGER = logging.getLogger(__name__)

COLUMN_N_GRAMS = "n_grams"    # constants.py:6
COLUMN_TITLE_ID = "title_id"  # constants.py:2
ENCODING_FLOAT_TYPE = np.float32  # settings.py:71


def sequential_sums(values, lengths, dtype):
    """Row-wise sums of a ragged array with the rounding of a left-to-right loop (what Python's `sum` does).

    values: flat array, row i = values[offsets[i]:offsets[i+1]]; returns one `dtype` per row.  Adding +0.0 to a
    partial sum is exact, so padding the rows to a common length does not change any rounding.
    """
    lengths = np.asarray(lengths, dtype=np.int64)
    offsets = np.concatenate(([0], np.cumsum(lengths)))
    out = np.zeros(lengths.shape[0], dtype=dtype)
    values = np.asarray(values, dtype=dtype)
    longest = int(lengths.max()) if lengths.shape[0] else 0
    for position in range(longest):
        active = np.nonzero(lengths > position)[0]
        out[active] = out[active] + values[offsets[active] + position]
    return out


class TruthIndex:
    """The truth inverted index of match_maker.py:122-133 resident in HBM (ds_index_create)."""

    def __init__(self, rowptr, truth_idx, idf32, sums32, device=0):
        self.rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
        self.truth_idx = np.ascontiguousarray(truth_idx, dtype=np.int32)
        self.idf32 = np.ascontiguousarray(idf32, dtype=np.float32)
        self.sums32 = np.ascontiguousarray(sums32, dtype=np.float32)
        self.device = device
        self.n_columns = self.rowptr.shape[0] - 1
        self.n_truth = self.sums32.shape[0]
        if self.idf32.shape[0] != self.n_columns:
            raise ValueError("idf32 must have one entry per column")
        self.handle = ctypes.c_void_p()
        _lib.check(_lib.lib().ds_index_create(
            _lib.pointer(self.rowptr), _lib.pointer(self.truth_idx), _lib.pointer(self.idf32),
            _lib.pointer(self.sums32), self.n_columns, self.n_truth, device, ctypes.byref(self.handle)),
            "ds_index_create")

    def info(self):
        raw = (ctypes.c_int64 * 8)()
        _lib.check(_lib.lib().ds_index_info(self.handle, raw), "ds_index_info")
        keys = ("n_truth", "n_columns", "nnz", "tile_rows", "tiles", "device_bytes", "padded_postings", "forward_index_bytes")
        return dict(zip(keys, list(raw)[:8]))

    def option(self, name, value):
        """Diagnostics switch of the index (ds_index_option), e.g. option("count_bytes", 1)."""
        _lib.check(_lib.lib().ds_index_option(self.handle, name.encode(), int(value)), "ds_index_option")

    def top_k(self, q_rowptr, q_cols, q_maxint, k):
        """fast_jaccard + fast_arg_top_k for a batch: int32[Q, k] truth rows, descending row index per query."""
        q_rowptr = np.ascontiguousarray(q_rowptr, dtype=np.int64)
        q_cols = np.ascontiguousarray(q_cols, dtype=np.int32)
        q_maxint = np.ascontiguousarray(q_maxint, dtype=np.float64)
        n_queries = q_rowptr.shape[0] - 1
        if q_maxint.shape[0] != n_queries:
            raise ValueError("q_maxint must have one entry per query")
        out = np.empty((n_queries, k), dtype=np.int32)
        _lib.check(_lib.lib().ds_jaccard_topk(self.handle, _lib.pointer(q_rowptr), _lib.pointer(q_cols),
                                              _lib.pointer(q_maxint), n_queries, k, _lib.pointer(out)),
                   "ds_jaccard_topk")
        return out

    def top_k_device(self, d_q_rowptr, d_q_cols, d_q_maxint, n_queries, k, d_out_rows, stream=None):
        """Enqueue on `stream` with every operand already in HBM (raw device pointers as ints / c_void_p)."""
        as_ptr = lambda x: x if isinstance(x, ctypes.c_void_p) else ctypes.c_void_p(int(x))
        _lib.check(_lib.lib().ds_jaccard_topk_device(
            self.handle, as_ptr(d_q_rowptr), as_ptr(d_q_cols), as_ptr(d_q_maxint), n_queries, k, as_ptr(d_out_rows),
            ctypes.c_void_p(stream or 0)), "ds_jaccard_topk_device")

    def sync(self, stream=None):
        stats = (ctypes.c_int64 * 32)()
        _lib.check(_lib.lib().ds_jaccard_sync(self.handle, ctypes.c_void_p(stream or 0), stats), "ds_jaccard_sync")
        names = ("setup", "list_pointers", "scatter_dense", "scan_dense", "select", "exact", "scatter_sparse",
                 "collect_sparse")
        return {"dense_queries": stats[0], "error_queries": stats[1], "exact_candidates": stats[2],
                "selections": stats[3], "phase_cycles": dict(zip(names, list(stats)[4:12])),
                "sparse_tiles": stats[12], "dense_tiles": stats[13], "skipped_columns": stats[14],
                "requested_bytes": stats[15],
                "dense_reasons": dict(zip(("shape", "items", "overflow_sparse", "overflow_dense", "ties", "few"),
                                          list(stats)[16:22])),
                "refines": stats[22], "raw_entries": stats[23], "refine_survivors": stats[24],
                "raw_entries_sparse": stats[25], "topk_kernel_ms": stats[26] / 1000.0,
                "dense_kernel_ms": stats[27] / 1000.0, "bounds_record": list(stats)[28:31],
                "sparse_redos": stats[31]}

    def status(self, n_queries, stream=None):
        """int32[n_queries] of the last call: 0 fast kernel, 1 literal kernel, 2 fewer than k rows, 3 bad column."""
        out = np.empty(n_queries, dtype=np.int32)
        _lib.check(_lib.lib().ds_jaccard_status(self.handle, ctypes.c_void_p(stream or 0), _lib.pointer(out),
                                                n_queries), "ds_jaccard_status")
        return out

    def close(self):
        if self.handle:
            _lib.lib().ds_index_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _ptr(array):
    return array.ctypes.data_as(ctypes.c_void_p)


class NativeProblem:
    """ctypes wrapper of `ds_problem_create` (include/doppel_amd.h, next row f-3): the host arrays of the index build
    computed natively from the transformed titles.  Host code: works without a GPU."""

    def __init__(self, truth_titles, query_titles, n_gram=3):
        def pack(titles):
            encoded = [t if isinstance(t, bytes) else str(t).encode("ascii") for t in titles]
            offsets = np.zeros(len(encoded) + 1, dtype=np.int64)
            np.cumsum([len(e) for e in encoded], out=offsets[1:])
            chars = np.frombuffer(b"".join(encoded), dtype=np.uint8) if offsets[-1] else np.zeros(1, dtype=np.uint8)
            return np.ascontiguousarray(chars), offsets
        self._create(*pack(truth_titles), *pack(query_titles), n_gram)

    @classmethod
    def from_flat(cls, truth_chars, truth_offsets, query_chars, query_offsets, n_gram=3):
        """From concatenated byte strings + offsets[n + 1] (no per-title Python objects: the form that scales)."""
        self = cls.__new__(cls)
        self._create(truth_chars, truth_offsets, query_chars, query_offsets, n_gram)
        return self

    def _create(self, t_chars, t_offsets, q_chars, q_offsets, n_gram):
        t_chars = np.ascontiguousarray(t_chars, dtype=np.uint8)
        q_chars = np.ascontiguousarray(q_chars, dtype=np.uint8)
        t_offsets = np.ascontiguousarray(t_offsets, dtype=np.int64)
        q_offsets = np.ascontiguousarray(q_offsets, dtype=np.int64)
        self.handle = ctypes.c_void_p()
        _lib.check(_lib.lib().ds_problem_create(_ptr(t_chars), _ptr(t_offsets), t_offsets.shape[0] - 1, _ptr(q_chars),
                                                _ptr(q_offsets), q_offsets.shape[0] - 1, n_gram,
                                                ctypes.byref(self.handle)), "ds_problem_create")
        info = (ctypes.c_int64 * 8)()
        _lib.check(_lib.lib().ds_problem_info(self.handle, info), "ds_problem_info")
        self.n_truth, self.n_queries, self.n_columns, self.nnz, self.q_nnz, self.n_gram = list(info)[:6]

    def arrays(self, copy=True):
        """The arrays (vocabulary keys, idf32, idf64, rowptr, truth_idx, sums32, q_rowptr, q_cols, q_maxint).
        copy=False returns read-only views of the handle's memory: keep this object alive as long as they are used."""
        pointers = [ctypes.c_void_p() for _ in range(9)]
        _lib.check(_lib.lib().ds_problem_arrays(self.handle, *[ctypes.byref(p) for p in pointers]), "ds_problem_arrays")
        spec = (("vocabulary_keys", np.uint32, self.n_columns), ("idf32", np.float32, self.n_columns),
                ("idf64", np.float64, self.n_columns), ("rowptr", np.int64, self.n_columns + 1),
                ("truth_idx", np.int32, self.nnz), ("sums32", np.float32, self.n_truth),
                ("q_rowptr", np.int64, self.n_queries + 1), ("q_cols", np.int32, self.q_nnz),
                ("q_maxint", np.float64, self.n_queries))
        out = {}
        for (name, dtype, count), pointer in zip(spec, pointers):
            if count == 0 or not pointer.value:
                out[name] = np.zeros(0, dtype=dtype)
                continue
            buffer = (ctypes.c_char * (count * np.dtype(dtype).itemsize)).from_address(pointer.value)
            view = np.frombuffer(buffer, dtype=dtype, count=count)
            out[name] = view.copy() if copy else view
        return out

    def vocabulary(self):
        """The n-gram string of every column, in column order."""
        keys = self.arrays()["vocabulary_keys"]
        n = self.n_gram
        return [bytes((int(key) >> (8 * (n - 1 - i))) & 0xff for i in range(n)).decode("latin-1") for key in keys]

    def close(self):
        if self.handle:
            _lib.lib().ds_problem_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MatchMaker:
    """
    The class responsible for getting the closest (based on Jaccard distance) titles, given a collection of titles.

    :param data: (dataframe) The collection of titles for which to find the closest titles
    :param truth_data: (dataframe) The "truth" database
    :param top_n: (int) Top n values to fetch
    :param device: HIP device ordinal
    :param vocabulary: optional explicit column order (list of n-grams); the reference uses the iteration order
        of a Python set (match_maker.py:145-147), which depends on PYTHONHASHSEED -- tests inject the captured one.

    * Main public method: get_closest_matches(...)
    """

    def __init__(self, data, truth_data, top_n, device=0, vocabulary=None):
        self.data = data
        self.truth_data = truth_data
        self.top_n = top_n

        LOGGER.info(f'[{self.__class__.__name__}] Loading pre-requisite data!')

        self._host_arrays(vocabulary)
        rowptr, truth_idx, idf32 = self._truth_rowptr, self._truth_idx, self._idf32
        self.truth_data = self.truth_data.loc[:, [COLUMN_TITLE_ID]]                         # :104
        self.index = TruthIndex(rowptr, truth_idx, idf32, self.sums_matrix_truth, device)
        self._rows = None
        self._ids = None
        self._ids_of = None

        LOGGER.info(f'[{self.__class__.__name__}] Loaded pre-requisite data!')

    def _host_arrays(self, vocabulary=None):
        """Everything `__init__` derives on the host (match_maker.py:91-107), without touching the GPU."""
        self.n_grams_counter = self._count(self.data[COLUMN_N_GRAMS])               # match_maker.py:91
        self.n_grams_counter_truth = self._count(self.truth_data[COLUMN_N_GRAMS])   # :92
        self.number_of_truth_titles = len(self.truth_data)                           # :93
        self.idf_s_mapping = {key: math.log(self.number_of_truth_titles / count)     # :135-142 (float64)
                              for key, count in self.n_grams_counter_truth.items()}
        self.max_idf_value = max(self.idf_s_mapping.values())                        # :95
        if vocabulary is None:                                                       # :144-147
            vocabulary = set(list(self.n_grams_counter.keys()) + list(self.n_grams_counter_truth.keys()))
        self.n_grams_decoding = {index: n_gram for index, n_gram in enumerate(vocabulary)}
        self.n_grams_encoding = {v: k for k, v in self.n_grams_decoding.items()}     # :97
        n_columns = len(self.n_grams_decoding)
        # idf per column, float64 (_get_idf_given_index, :180-181) and its float32 image (the matrix dtype, :152)
        self._idf64 = np.array([self.idf_s_mapping.get(self.n_grams_decoding[g], self.max_idf_value)
                                for g in range(n_columns)], dtype=np.float64)
        self._idf32 = self._idf64.astype(ENCODING_FLOAT_TYPE)

        self._q_rowptr, self._q_cols, self._q_maxint = self._construct_data_rows(self._idf32)   # :99, :106, :197
        del self.data                                                                             # :100

        self._truth_rowptr, self._truth_idx, self.sums_matrix_truth = \
            self._construct_truth_index(self._idf32, n_columns)                                   # :102-107

    @classmethod
    def host_arrays(cls, data, truth_data, vocabulary=None):
        """The host-side arrays of `MatchMaker(data, truth_data, ...)` as a dict, without a GPU (parity tests of the
        native index build compare against this)."""
        self = cls.__new__(cls)
        self.data, self.truth_data = data, truth_data
        self._host_arrays(vocabulary)
        return dict(vocabulary=[self.n_grams_decoding[g] for g in range(len(self.n_grams_decoding))],
                    idf32=self._idf32, idf64=self._idf64, rowptr=self._truth_rowptr, truth_idx=self._truth_idx,
                    sums32=self.sums_matrix_truth, q_rowptr=self._q_rowptr, q_cols=self._q_cols,
                    q_maxint=self._q_maxint)

    @classmethod
    def from_titles(cls, titles, truth_titles, top_n, title_ids=None, n_gram=3, device=0):
        """Next row f-3: the same object built by the native index build (`ds_problem_create`) straight from the
        transformed titles (`transform_title` output: ASCII byte strings) -- no per-title Python sets, no DataFrames.
        `title_ids` (default: the truth row numbers) plays the role of `truth_data[title_id]`."""
        import pandas as pd
        problem = NativeProblem(truth_titles, titles, n_gram)
        self = cls.__new__(cls)
        self.top_n = top_n
        self.number_of_truth_titles = problem.n_truth
        arrays = problem.arrays()
        self.n_grams_decoding = dict(enumerate(problem.vocabulary()))
        self.n_grams_encoding = {v: k for k, v in self.n_grams_decoding.items()}
        self._idf64, self._idf32 = arrays["idf64"], arrays["idf32"]
        self._q_rowptr, self._q_cols, self._q_maxint = arrays["q_rowptr"], arrays["q_cols"], arrays["q_maxint"]
        self.sums_matrix_truth = arrays["sums32"]
        ids = np.arange(problem.n_truth) if title_ids is None else np.asarray(title_ids)
        self.truth_data = pd.DataFrame({COLUMN_TITLE_ID: ids})
        self.index = TruthIndex(arrays["rowptr"], arrays["truth_idx"], arrays["idf32"], arrays["sums32"], device)
        self._rows = None
        self._ids = None
        self._ids_of = None
        return self

    @classmethod
    def from_index(cls, index, q_rowptr, q_cols, q_maxint, title_ids, top_n):
        """The call surface over a `TruthIndex` that is already resident (shared, not copied) and query rows that are
        already built: what `bench.py`'s `surface` record and the tests time -- `get_closest_matches(row)` per row exactly
        as predict.py:126-127 calls it."""
        import pandas as pd
        self = cls.__new__(cls)
        self.top_n = top_n
        self.number_of_truth_titles = index.n_truth
        self._q_rowptr, self._q_cols, self._q_maxint = q_rowptr, q_cols, q_maxint
        self.truth_data = pd.DataFrame({COLUMN_TITLE_ID: np.asarray(title_ids)})
        self.index = index
        self._rows = None
        self._ids = None
        self._ids_of = None
        return self

    @staticmethod
    def _count(column):  # common.py:145-147 get_n_grams_counter
        counter = {}
        for n_grams in column:
            for n_gram in set(n_grams):
                counter[n_gram] = counter.get(n_gram, 0) + 1
        return counter

    def _flatten(self, column):
        """(row lengths, flat column ids in each set's iteration order) -- the order `_get_encoding_values` sees."""
        encoding = self.n_grams_encoding
        lengths = np.fromiter((len(n_grams) for n_grams in column), dtype=np.int64, count=len(column))
        flat = np.fromiter((encoding[n_gram] for n_grams in column for n_gram in n_grams), dtype=np.int64,
                           count=int(lengths.sum()))
        return lengths, flat

    def _construct_data_rows(self, idf32):
        """matrix_non_zero_columns (:111-120) as CSR + max_intersection_possible (:197) per row."""
        lengths, flat = self._flatten(self.data[COLUMN_N_GRAMS])
        rows = np.repeat(np.arange(lengths.shape[0], dtype=np.int64), lengths)
        keep = idf32[flat] != 0                       # lil_matrix(...).nonzero() drops explicit zeros (:118)
        rows, flat = rows[keep], flat[keep]
        order = np.lexsort((flat, rows))              # .nonzero()[1] of a lil row: ascending column ids
        rows, flat = rows[order], flat[order]
        counts = np.bincount(rows, minlength=lengths.shape[0]).astype(np.int64)
        rowptr = np.concatenate(([0], np.cumsum(counts))).astype(np.int64)
        maxint = sequential_sums(self._idf64[flat], counts, np.float64)   # Python sum of float64, column order
        return rowptr, flat.astype(np.int32), maxint

    def _construct_truth_index(self, idf32, n_columns):
        """matrix_truth_non_zero_columns_and_values (:122-133) as CSR + sums_matrix_truth (:102, :174)."""
        lengths, flat = self._flatten(self.truth_data[COLUMN_N_GRAMS])
        sums = sequential_sums(idf32[flat], lengths, ENCODING_FLOAT_TYPE)  # sum(uniqueness_values), set order
        rows = np.repeat(np.arange(lengths.shape[0], dtype=np.int64), lengths)
        keep = idf32[flat] != 0
        rows, flat = rows[keep], flat[keep]
        order = np.argsort(flat, kind="stable")       # rows stay ascending within a column
        truth_idx = rows[order].astype(np.int32)
        rowptr = np.concatenate(([0], np.cumsum(np.bincount(flat, minlength=n_columns)))).astype(np.int64)
        return rowptr, truth_idx, sums

    def get_closest_matches_batch(self, row_numbers=None):
        """Truth ROW indexes int32[len(rows), top_n] (descending row index) for the given rows of `data`."""
        if self._rows is None:
            self._rows = self.index.top_k(self._q_rowptr, self._q_cols, self._q_maxint, self.top_n)
        if row_numbers is None:
            return self._rows
        return self._rows[np.asarray(row_numbers, dtype=np.int64)]

    def _title_ids(self):
        """title_id of every candidate of every row of `data`, [rows, top_n]: `self.truth_data.loc[top_matches, title_id]`
        (:190) for the whole batch in ONE look-up.  The reference pays that `.loc` once per call of get_closest_matches
        (270 us on a 500k-row frame: 27 s for 100k queries, against 16 ms for the kernels); here it is paid once."""
        if self._ids is None or self._ids_of is not self.truth_data:   # (`truth_data` reassigned since: look the ids up again)
            rows = self.get_closest_matches_batch()
            ids = self.truth_data[COLUMN_TITLE_ID]
            index = self.truth_data.index
            positional = (type(index).__name__ == "RangeIndex" and index.start == 0 and index.step == 1)   # common.py:60
            if positional:
                self._ids = ids.to_numpy()[rows]
            elif index.is_unique:   # any other index: `.loc` keeps its label semantics, still one call
                self._ids = ids.loc[rows.reshape(-1)].to_numpy().reshape(rows.shape)
            else:   # duplicate labels: `.loc` returns more than one row per label -- per call, as the reference does (:190)
                return None
            self._ids_of = self.truth_data
        return self._ids

    def _get_top_n_matches(self, top_matches):
        """For the selected truth rows, gets the title_id's from self.truth_data (:183-190)."""
        if top_matches.shape[0] != self.top_n:
            raise Exception('top_matches.shape[0] != self.top_n')
        return self.truth_data.loc[top_matches, COLUMN_TITLE_ID].tolist()

    def get_closest_matches(self, row_number):
        """
        Given the "row_number" of self.data, gets the closest (self.top_n) titles in self.truth_data
        """
        table = self._title_ids()
        if table is None:   # a truth index with duplicate labels: the reference's own per-call look-up, its own outcome
            return self._get_top_n_matches(self.get_closest_matches_batch([row_number])[0])
        ids = table[row_number]
        if ids.shape[0] != self.top_n:                                                        # :188-189
            raise Exception('top_matches.shape[0] != self.top_n')
        return ids.tolist()
