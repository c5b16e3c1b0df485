# doppelspeller/_amd.py  (new file in the reference)
import ctypes, numpy as np
import os
_lib = ctypes.CDLL(os.environ.get("DOPPEL_AMD_LIBRARY", "libdoppel_amd.so"))
_lib.ds_last_error.restype = ctypes.c_char_p
_p = lambda a: a.ctypes.data_as(ctypes.c_void_p)

def _check(status):
    if status == -3:                                   # DS_E_TOP_N  == match_maker.py:188-189
        raise Exception('top_matches.shape[0] != self.top_n')
    if status != 0:
        raise RuntimeError(_lib.ds_last_error().decode())

class AmdIndex:
    """Built once at the end of MatchMaker.__init__ (match_maker.py:106-107) from the same typed lists."""
    def __init__(self, match_maker, device=0):
        lists = match_maker.matrix_truth_non_zero_columns_and_values          # match_maker.py:122-133
        rowptr = np.zeros(len(lists) + 1, np.int64)
        rowptr[1:] = np.cumsum([len(columns) for columns, _ in lists])
        truth_idx = np.concatenate([columns for columns, _ in lists]).astype(np.int32)
        idf32 = np.array([match_maker._get_idf_given_index(g) for g in range(len(lists))], np.float32)  # :130
        sums32 = np.ascontiguousarray(match_maker.sums_matrix_truth, np.float32)                      # :102,174
        self.handle = ctypes.c_void_p()
        _check(_lib.ds_index_create(_p(rowptr), _p(truth_idx), _p(idf32), _p(sums32),
                                    ctypes.c_int64(len(lists)), ctypes.c_int64(sums32.shape[0]), device,
                                    ctypes.byref(self.handle)))

    def top_rows(self, match_maker, rows, k):
        """fast_jaccard + fast_arg_top_k (match_maker.py:199-203, :187) for many rows at once."""
        columns = [np.asarray(match_maker.matrix_non_zero_columns[r], np.int32) for r in rows]       # :196
        q_rowptr = np.zeros(len(rows) + 1, np.int64); q_rowptr[1:] = np.cumsum([len(c) for c in columns])
        q_cols = np.concatenate(columns) if columns else np.zeros(0, np.int32)
        q_maxint = np.array([sum([match_maker._get_idf_given_index(c) for c in cols]) for cols in columns])  # :197
        out = np.empty((len(rows), k), np.int32)
        _check(_lib.ds_jaccard_topk(self.handle, _p(q_rowptr), _p(q_cols), _p(q_maxint),
                                    ctypes.c_int64(len(rows)), ctypes.c_int32(k), _p(out)))
        return out      # truth row indexes, descending; map with truth_data.loc[..., 'title_id'] as :190 does

def construct_features(n_q, n_t, title, title_truth, counts, space_code, n_truth, dummy, response):
    """Same 9 arguments as feature_engineering.py:77-80; response is float32[n, 66], C-contiguous."""
    n, stride = response.shape[0], title.shape[1]
    _check(_lib.ds_construct_features(_p(np.ascontiguousarray(n_q, np.uint8)), _p(np.ascontiguousarray(n_t, np.uint8)),
                                      _p(np.ascontiguousarray(title)), _p(np.ascontiguousarray(title_truth)),
                                      _p(np.ascontiguousarray(counts, np.uint32)), ctypes.c_uint8(int(space_code)),
                                      ctypes.c_uint32(int(n_truth)), ctypes.c_int64(n), ctypes.c_int64(stride),
                                      0, _p(response)))
