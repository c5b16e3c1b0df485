"""construct_features / fast_levenshtein_ratio call surface (doppelspeller/feature_engineering.py:25-169, 298-319).

`construct_features(...)` keeps the reference's 9 positional arguments and in-place `response` output, so the call
sites predict.py:216-219 and feature_engineering.py:358-361 work unchanged.  `TitleTable` +
`construct_features_indexed` are the upload-once form: encoded titles live in HBM and a pair is two row indexes.
"""
import ctypes

import numpy as np

from . import _lib

NUMBER_OF_WORDS_FEATURES = 15                                  # settings.py:65
WORDS_COUNT_DATA_TYPE = np.uint32                              # settings.py:66
NUMBER_OF_CHARACTERS_DATA_TYPE = np.uint8                      # settings.py:67
MAX_CHARACTERS_ALLOWED_IN_THE_TITLE = 255                      # settings.py:68
ENCODING_FLOAT_TYPE = np.float32                               # settings.py:71
FEATURES_COUNT = 6 + (4 * NUMBER_OF_WORDS_FEATURES)            # feature_engineering.py:67
ALLOWED_CHARACTERS = "- abcdefghijklmnopqrstuvwxyz0123456789"  # feature_engineering.py:200 ('-' = fill, code 0)
_ENCODING = {character: index for index, character in enumerate(ALLOWED_CHARACTERS)}
SPACE_CODE = _ENCODING[" "]                                    # feature_engineering.py:203


def encode_title(title):
    """feature_engineering.py:298-307: character codes, right-padded with 0 to 255 (uint8)."""
    out = np.zeros(MAX_CHARACTERS_ALLOWED_IN_THE_TITLE, dtype=NUMBER_OF_CHARACTERS_DATA_TYPE)
    codes = [_ENCODING[character] for character in title[:MAX_CHARACTERS_ALLOWED_IN_THE_TITLE]]
    out[:len(codes)] = codes
    return out


def encode_titles(titles):
    """Rows of encode_title plus the lengths: (uint8[n, 255], uint8[n])."""
    table = np.zeros(256, dtype=np.uint8)
    for character, code in _ENCODING.items():
        table[ord(character)] = code
    enc = np.zeros((len(titles), MAX_CHARACTERS_ALLOWED_IN_THE_TITLE), dtype=np.uint8)
    lengths = np.zeros(len(titles), dtype=np.uint8)
    for row, title in enumerate(titles):
        raw = np.frombuffer(title[:MAX_CHARACTERS_ALLOWED_IN_THE_TITLE].encode("ascii"), dtype=np.uint8)
        enc[row, :raw.shape[0]] = table[raw]
        lengths[row] = raw.shape[0]
    return enc, lengths


def get_truth_words_counts(title, words_counter):
    """feature_engineering.py:309-319: truth-database document frequency of the first 15 words (uint32, 0-padded)."""
    counts = [words_counter.get(word) for word in title.split()][:NUMBER_OF_WORDS_FEATURES]
    out = np.zeros(NUMBER_OF_WORDS_FEATURES, dtype=WORDS_COUNT_DATA_TYPE)
    out[:len(counts)] = counts
    return out


def _optional_pointer(array):
    return ctypes.c_void_p(array.ctypes.data) if array is not None else ctypes.c_void_p(0)


def encode_collection(flat, offsets, code_of=None, stride=MAX_CHARACTERS_ALLOWED_IN_THE_TITLE):
    """encode_title (feature_engineering.py:298-307) for a whole collection in one native, threaded call
    (ds_encode_titles): titles = flat[offsets[i]:offsets[i + 1]] (bytes), code_of = uint8[256] character -> code table (None:
    the bytes are codes already) -> (uint8[n, stride], uint8[n])."""
    count = offsets.shape[0] - 1
    enc = np.empty((count, stride), dtype=np.uint8)
    lengths = np.empty(count, dtype=np.uint8)
    flat = np.ascontiguousarray(flat, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    _lib.check(_lib.lib().ds_encode_titles(_optional_pointer(flat), _optional_pointer(offsets), count,
                                           _optional_pointer(code_of), stride, _optional_pointer(enc),
                                           _optional_pointer(lengths)), "ds_encode_titles")
    return enc, lengths


def truth_word_counts(flat, offsets, separators=(SPACE_CODE,)):
    """get_truth_words_counts (feature_engineering.py:309-319) over the counter of common.py:140-142 for a whole truth
    collection in one native, threaded call (ds_truth_word_counts): uint32[n, 15].  separators: the byte values str.split()
    splits on (the space code for encoded titles; ASCII white space for text)."""
    count = offsets.shape[0] - 1
    out = np.empty((count, NUMBER_OF_WORDS_FEATURES), dtype=np.uint32)
    table = np.zeros(256, dtype=np.uint8)
    table[list(separators)] = 1
    flat = np.ascontiguousarray(flat, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    _lib.check(_lib.lib().ds_truth_word_counts(_optional_pointer(flat), _optional_pointer(offsets), count,
                                               _optional_pointer(table), _optional_pointer(out)), "ds_truth_word_counts")
    return out


def construct_features(title_number_of_characters, truth_number_of_characters, title, title_truth,
                       truth_words_counts, space_code, number_of_truth_titles, dummy, response):
    """
    The main (vectorized) function to generate features for pairs of title and title_truth
    (feature_engineering.py:75-169; gufunc layout '(),(),(l),(l),(m),(),(),(n)->(n)').

    Same arguments as the reference; `response` (float32[n, 66]) is updated in place, `dummy` is ignored.
    NaN entries are part of the result (words beyond the truth title's word count).
    """
    response_array = np.asarray(response)
    if response_array.dtype != ENCODING_FLOAT_TYPE or response_array.shape[-1] != FEATURES_COUNT:
        raise TypeError(f"response must be float32[..., {FEATURES_COUNT}]")
    single = response_array.ndim == 1
    n = 1 if single else int(np.prod(response_array.shape[:-1]))
    title = np.asarray(title, dtype=NUMBER_OF_CHARACTERS_DATA_TYPE)
    title_truth = np.asarray(title_truth, dtype=NUMBER_OF_CHARACTERS_DATA_TYPE)
    stride = max(title.shape[-1], title_truth.shape[-1])

    def rows(array, width, dtype):
        array = np.asarray(array, dtype=dtype)
        if array.shape[-1] < width:
            padded = np.zeros(array.shape[:-1] + (width,), dtype=dtype)
            padded[..., :array.shape[-1]] = array
            array = padded
        array = array.reshape((-1, width))
        if array.shape[0] != n:
            array = np.broadcast_to(array, (n, width))
        return np.ascontiguousarray(array)

    q_len = np.ascontiguousarray(np.broadcast_to(
        np.asarray(title_number_of_characters, dtype=NUMBER_OF_CHARACTERS_DATA_TYPE).reshape(-1), (n,)))
    t_len = np.ascontiguousarray(np.broadcast_to(
        np.asarray(truth_number_of_characters, dtype=NUMBER_OF_CHARACTERS_DATA_TYPE).reshape(-1), (n,)))
    q_enc = rows(title, stride, NUMBER_OF_CHARACTERS_DATA_TYPE)
    t_enc = rows(title_truth, stride, NUMBER_OF_CHARACTERS_DATA_TYPE)
    counts = rows(truth_words_counts, NUMBER_OF_WORDS_FEATURES, WORDS_COUNT_DATA_TYPE)
    direct = response_array.flags["C_CONTIGUOUS"]
    out = response_array.reshape((n, FEATURES_COUNT)) if direct else np.empty((n, FEATURES_COUNT), np.float32)
    device = 0
    _lib.check(_lib.lib().ds_construct_features(
        _lib.pointer(q_len), _lib.pointer(t_len), _lib.pointer(q_enc), _lib.pointer(t_enc), _lib.pointer(counts),
        int(space_code), int(number_of_truth_titles), n, stride, device, _lib.pointer(out)), "ds_construct_features")
    if not direct:
        response_array[...] = out.reshape(response_array.shape)
    return None


class TitleTable:
    """Encoded titles uploaded once to HBM (ds_titles_create): rows of encode_title (+ word counts for truth)."""

    def __init__(self, enc, lengths, word_counts=None, device=0):
        self.enc = np.ascontiguousarray(enc, dtype=np.uint8)
        self.lengths = np.ascontiguousarray(lengths, dtype=np.uint8)
        self.word_counts = None if word_counts is None else np.ascontiguousarray(word_counts, dtype=np.uint32)
        self.n, self.stride = self.enc.shape
        self.device = device
        self.handle = ctypes.c_void_p()
        counts_ptr = ctypes.c_void_p(0) if self.word_counts is None else _lib.pointer(self.word_counts)
        _lib.check(_lib.lib().ds_titles_create(_lib.pointer(self.enc), self.stride, _lib.pointer(self.lengths),
                                               counts_ptr, self.n, device, ctypes.byref(self.handle)),
                   "ds_titles_create")

    def option(self, name, value):
        """ds_titles_option, e.g. option("truth_records", 0): no per-row records of the truth-only features (160 B of HBM per row)."""
        _lib.check(_lib.lib().ds_titles_option(self.handle, name.encode(), int(value)), "ds_titles_option")

    def close(self):
        if self.handle:
            _lib.lib().ds_titles_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def construct_features_indexed(queries, truth, pair_q, pair_t, space_code, number_of_truth_titles):
    """float32[n, 66] for pairs (queries[pair_q[i]], truth[pair_t[i]]) of two TitleTables."""
    pair_q = np.ascontiguousarray(pair_q, dtype=np.int32)
    pair_t = np.ascontiguousarray(pair_t, dtype=np.int32)
    out = np.empty((pair_q.shape[0], FEATURES_COUNT), dtype=np.float32)
    _lib.check(_lib.lib().ds_construct_features_indexed(
        queries.handle, truth.handle, _lib.pointer(pair_q), _lib.pointer(pair_t), int(space_code),
        int(number_of_truth_titles), pair_q.shape[0], _lib.pointer(out)), "ds_construct_features_indexed")
    return out


def levenshtein_ratio_batch(a_sequences, b_sequences, method=0, device=0):
    """fast_levenshtein_ratio (feature_engineering.py:25-63) for lists of uint8 code arrays -> uint8[n].

    method 0: bit-parallel LCS kernel (literal uint8-wrap DP where lengths require it); 1: anti-diagonal DP kernel.
    """
    def flatten(sequences):
        lengths = np.array([len(x) for x in sequences], dtype=np.int64)
        offsets = np.concatenate(([0], np.cumsum(lengths))).astype(np.int64)
        chars = (np.concatenate([np.asarray(x, dtype=np.uint8) for x in sequences])
                 if offsets[-1] else np.zeros(1, dtype=np.uint8))
        return np.ascontiguousarray(chars), offsets

    a_chars, a_off = flatten(a_sequences)
    b_chars, b_off = flatten(b_sequences)
    out = np.empty(len(a_sequences), dtype=np.uint8)
    _lib.check(_lib.lib().ds_levenshtein_ratio_batch(
        _lib.pointer(a_chars), _lib.pointer(a_off), _lib.pointer(b_chars), _lib.pointer(b_off), len(a_sequences),
        method, device, _lib.pointer(out)), "ds_levenshtein_ratio_batch")
    return out


LEVENSHTEIN_RATIO_THRESHOLD = 94                               # settings.py:75
# order of the character codes under Python's sorted() (common.py:166): the code point of the character
SORT_KEY = np.zeros(256, dtype=np.uint8)
for _character, _code in _ENCODING.items():
    SORT_KEY[_code] = ord(_character)


def find_close_matches(queries, truth, rows, threshold=LEVENSHTEIN_RATIO_THRESHOLD, space_code=SPACE_CODE,
                       sort_key=SORT_KEY):
    """The fuzzy step of Prediction._find_close_matches (predict.py:140-183) for the first len(rows) queries.

    queries / truth: TitleTables; rows: int32[Q, k] candidate truth rows per query (the Jaccard top-k).
    Returns (ratios uint8[Q, k], best_row int32[Q]): ratios[q, j] = Prediction._get_levenshtein_ratio(query q,
    candidate j) (predict.py:147-156); best_row[q] = the single candidate with the highest ratio above `threshold`
    (predict.py:172-176), -1 when none or when several tie (_remove_duplicated_matches, predict.py:158-161).
    """
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    n_queries, k = rows.shape
    ratios = np.empty((n_queries, k), dtype=np.uint8)
    best_row = np.empty(n_queries, dtype=np.int32)
    sort_key = np.ascontiguousarray(sort_key, dtype=np.uint8)
    _lib.check(_lib.lib().ds_close_matches(queries.handle, truth.handle, _lib.pointer(rows), k, n_queries,
                                           int(space_code), _lib.pointer(sort_key), int(threshold),
                                           _lib.pointer(ratios), _lib.pointer(best_row)), "ds_close_matches")
    return ratios, best_row
