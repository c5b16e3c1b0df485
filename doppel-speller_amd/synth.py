"""Deterministic synthetic title sets shaped like the reference's example data.

Truth titles: words drawn from a Zipf(1.1) vocabulary of synthetic words over [a-z0-9] (length ~ lognormal), 1 +
Poisson(2.5) words per title, plus a company-suffix word ("limited", "ltd", "bv", ...) on ~45 % of the titles so that
a handful of tri-grams have posting lists covering ~27 % of the truth set (the "limited/ltd" family of the example
data).  A small share of the word slots holds a word seen nowhere else (a hapax: proper names, registration numbers
-- 18 % of the example data's word slots hold a word that occurs once; 5 % here), spelled from uniformly random
characters: these are what lets the tri-gram vocabulary approach its 37^3 ceiling at N >= 500k, as SURVEY.md 8d
requires.  Suffix shares follow the example data (ltd 27 %, limited 26 %, bv 16 % of the truth titles).
Queries: 60 % are a truth title with 1-2 keyboard-style edits (the recipe of
feature_engineering_prepare.py:90-173), 40 % are fresh titles.

Round 3: the titles are drawn by a small native generator (csrc/ds_synth.cpp -> libdoppel_synth.so, g++, threaded;
one random stream per title, so the output does not depend on the thread count) and everything derived from them comes
from the product's own host entry points -- `ds_problem_create` (vocabulary, IDF, inverted index, query rows: next
row f-3), `ds_encode_titles` / `ds_truth_word_counts` (row a7).  The NumPy generator of rounds 1-2 took 316 s for the
50M truth titles of C5; this one takes seconds.  Same recipe and parameters, different random draws: the workload of a
given seed is NOT the round-2 workload of that seed.

Measured at N = 500k (seed 20260101; SURVEY.md 8d targets in brackets): see
tests/test_host_cpu.py::test_synthetic_workload_acceptance_at_c2_truth_size.

`make_workload` returns everything both kernels consume, built the way MatchMaker.__init__ /
FeatureEngineering.encode_title build them (match_maker.py:84-181, feature_engineering.py:298-319), with the
column order fixed to ascending tri-gram code instead of Python-set order:
    rowptr / truth_idx / idf32 / sums32          truth inverted index (CSR over tri-gram columns)
    q_rowptr / q_cols / q_maxint                 query rows (ascending column ids) and max_intersection_possible
    q_enc, q_len, t_enc, t_len, t_counts         encoded titles (uint8[*, 255]), lengths, truth word counts
"""
import ctypes
import hashlib
import os
import subprocess
import math
from types import SimpleNamespace

import numpy as np

from .feature_engineering import (ALLOWED_CHARACTERS, MAX_CHARACTERS_ALLOWED_IN_THE_TITLE, NUMBER_OF_WORDS_FEATURES,
                                  encode_collection, truth_word_counts)  # noqa: F401 - re-exported for tests

DEFAULT_SEED = 20260101
_BASE = len(ALLOWED_CHARACTERS)  # 38 codes: '-'=0 (fill), ' '=1, a-z=2..27, 0-9=28..37
_SPACE = 1
_SUFFIXES = ("limited", "ltd", "bv", "plc", "inc", "llc", "gmbh", "company")
_SUFFIX_WEIGHTS = np.array([0.355, 0.37, 0.215, 0.02, 0.01, 0.01, 0.01, 0.01])
SUFFIX_SHARE = 0.72        # example data: ltd 27.3 %, limited 26.2 %, bv 16.4 % of the truth titles
ZIPF_OFFSET = 4.0
SYLLABLES = 600
RANDOM_WORD_FRACTION = 0.25
SYLLABLE_EXPONENT = 1.2
COMMON_SYLLABLES = 30     # the most frequent syllables are whole tri-grams ("ion", "ing", "ter" in the example data)
SYLLABLE_OFFSET = 8.0
HAPAX_FRACTION = 0.05      # word slots filled with a word that occurs nowhere else
HAPAX_DIGIT_SHARE = 0.2    # characters of a hapax that are digits
_KEYBOARD_ROWS = ("1234567890", "qwertyuiop", "asdfghjkl", "zxcvbnm")


def _codes(text):
    return np.array([ALLOWED_CHARACTERS.index(ch) for ch in text], dtype=np.uint8)


class _Vocabulary:
    def __init__(self, rng, size):
        # words are chains of syllables drawn (Zipf) from a small inventory, so that tri-grams are shared between
        # words the way they are in natural-language company names (vocabulary of tri-grams grows slowly with N)
        lengths = np.clip(np.rint(rng.lognormal(1.7, 0.45, size)), 1, 20).astype(np.int64)
        width = 20
        syllable_length = rng.choice([1, 2, 3], size=SYLLABLES, p=[0.15, 0.5, 0.35])
        syllable_length[:COMMON_SYLLABLES] = 3  # "ion", "ing", "ter": the frequent syllables are whole tri-grams
        syllable_letters = rng.randint(2, 28, (SYLLABLES, 3))
        syllable_digits = rng.randint(28, 38, (SYLLABLES, 3))
        syllables = np.where(rng.rand(SYLLABLES, 3) < 0.96, syllable_letters, syllable_digits).astype(np.uint8)
        syllable_weights = 1.0 / (np.arange(1, SYLLABLES + 1) + SYLLABLE_OFFSET) ** SYLLABLE_EXPONENT
        syllable_cumulative = np.cumsum(syllable_weights / syllable_weights.sum())
        chars = np.zeros((size, width + 3), dtype=np.uint8)
        filled = np.zeros(size, dtype=np.int64)
        rows = np.arange(size)
        while (filled < lengths).any():
            pick = np.minimum(np.searchsorted(syllable_cumulative, rng.rand(size)), SYLLABLES - 1)
            for j in range(3):
                active = (filled < lengths) & (j < syllable_length[pick])
                chars[rows[active], (filled + j)[active]] = syllables[pick[active], j]
            filled = np.where(filled < lengths, filled + syllable_length[pick], filled)
        chars = chars[:, :width]
        # a share of the words is spelled from uniformly random characters: rare tri-grams, so that the tri-gram
        # vocabulary keeps growing towards its 37^3 ceiling as more of the word vocabulary gets sampled
        random_word = rng.rand(size) < RANDOM_WORD_FRACTION
        uniform = np.where(rng.rand(size, width) < 0.9, rng.randint(2, 28, (size, width)),
                           rng.randint(28, 38, (size, width))).astype(np.uint8)
        chars[random_word] = uniform[random_word]
        chars[np.arange(width)[None, :] >= lengths[:, None]] = 0
        for i, suffix in enumerate(_SUFFIXES):  # the forced head
            chars[i] = 0
            chars[i, :len(suffix)] = _codes(suffix)
        # the reference counts words as strings: make the strings unique
        keys = chars.view(np.dtype((np.void, width))).reshape(-1)
        _, first = np.unique(keys, return_index=True)
        keep = np.sort(first)
        self.chars = chars[keep]
        self.lengths = (self.chars != 0).sum(axis=1).astype(np.int64)
        self.size = self.chars.shape[0]
        ranks = np.arange(1, self.size - len(_SUFFIXES) + 1, dtype=np.float64)
        weights = 1.0 / (ranks + ZIPF_OFFSET) ** 1.1  # Zipf-Mandelbrot: flattens the head (see make_workload doc)
        self.cumulative = np.cumsum(weights / weights.sum())

    def sample(self, rng, count):
        """Zipf(1.1) over the non-suffix words."""
        return len(_SUFFIXES) + np.minimum(np.searchsorted(self.cumulative, rng.rand(count)),
                                           self.size - len(_SUFFIXES) - 1)


_HERE = os.path.dirname(os.path.abspath(__file__))
_native = None


def _native_source():
    return os.path.join(_HERE, "csrc", "ds_synth.cpp")


def build_native(force=False):
    """g++ csrc/ds_synth.cpp -> libdoppel_synth.so next to this file; rebuilt when the source's hash changes."""
    target = os.path.join(_HERE, "libdoppel_synth.so")
    with open(_native_source(), "rb") as handle:
        wanted = hashlib.sha256(handle.read()).hexdigest()[:16]
    stamp = target + ".id"
    if not force and os.path.exists(target) and os.path.exists(stamp) and open(stamp).read().strip() == wanted:
        return target
    scratch = f"{target}.{os.getpid()}.tmp"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-pthread", _native_source(), "-o", scratch])
    os.replace(scratch, target)
    with open(stamp + f".{os.getpid()}", "w") as handle:
        handle.write(wanted)
    os.replace(stamp + f".{os.getpid()}", stamp)
    return target


def _generator():
    global _native
    if _native is None:
        handle = ctypes.CDLL(build_native())
        p, i64, u64 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64
        handle.synth_titles.argtypes = [u64, u64, i64, i64, p, p, i64, i64, p, p, p, p, p, ctypes.c_int, p, p, p]
        handle.synth_queries.argtypes = [u64, i64, p, p, p, p, p, p, i64, i64, p, p, p, p, p, ctypes.c_int, p, p, p]
        _native = handle
    return _native


def host_threads():
    """Threads of the native generator (the product library reads DS_HOST_THREADS itself)."""
    wanted = int(os.environ.get("DS_HOST_THREADS", "0") or 0)
    return wanted if wanted >= 1 else min(32, len(os.sched_getaffinity(0)))


def _ptr(array):
    return ctypes.c_void_p(array.ctypes.data) if array is not None else ctypes.c_void_p(0)


def _tables(vocabulary):
    """The distributions of the recipe as cumulative tables (the native side only searches them)."""
    # words per title: clip(1 + Poisson(2.5), 1, 20)
    pmf = np.array([math.exp(-2.5) * 2.5 ** k / math.factorial(k) for k in range(19)])
    words = np.minimum(np.concatenate((np.cumsum(pmf), [1.0])), 1.0)
    # hapax length: clip(rint(lognormal(1.7, 0.45)), 4, 20)
    def below(x):
        return 0.5 * (1.0 + math.erf((math.log(x) - 1.7) / (0.45 * math.sqrt(2.0))))
    hapax = np.array([below(length + 0.5) for length in range(4, 20)] + [1.0])
    suffix = np.cumsum(_SUFFIX_WEIGHTS / _SUFFIX_WEIGHTS.sum())
    suffix[-1] = 1.0
    shares = np.array([SUFFIX_SHARE, HAPAX_FRACTION, HAPAX_DIGIT_SHARE], dtype=np.float64)
    neighbours = np.zeros((256, 2), dtype=np.uint8)
    for row in _KEYBOARD_ROWS:
        for at, character in enumerate(row):
            code = ALLOWED_CHARACTERS.index(character)
            if at > 0:
                neighbours[code, 0] = ALLOWED_CHARACTERS.index(row[at - 1])
            if at + 1 < len(row):
                neighbours[code, 1] = ALLOWED_CHARACTERS.index(row[at + 1])
    return dict(word_chars=np.ascontiguousarray(vocabulary.chars, dtype=np.uint8),
                word_lengths=np.ascontiguousarray(vocabulary.lengths, dtype=np.uint8),
                zipf=np.ascontiguousarray(vocabulary.cumulative, dtype=np.float64),
                words=np.ascontiguousarray(words), suffix=np.ascontiguousarray(suffix),
                hapax=np.ascontiguousarray(hapax), shares=shares, neighbours=neighbours)


def _vocabulary_arguments(tables):
    return (_ptr(tables["word_chars"]), _ptr(tables["word_lengths"]), tables["word_chars"].shape[0], len(_SUFFIXES),
            _ptr(tables["zipf"]), _ptr(tables["words"]), _ptr(tables["suffix"]), _ptr(tables["hapax"]),
            _ptr(tables["shares"]), host_threads())


def _make_titles(tables, seed, count, purpose=0):
    """`count` fresh titles of stream (seed, purpose) -> (flat codes, offsets)."""
    lengths = np.zeros(count, dtype=np.int32)
    arguments = (seed, purpose, 0, count) + _vocabulary_arguments(tables)
    assert _generator().synth_titles(*arguments, _ptr(lengths), None, None) == 0
    offsets = np.zeros(count + 1, dtype=np.int64)
    np.cumsum(lengths, out=offsets[1:])
    flat = np.zeros(max(1, int(offsets[-1])), dtype=np.uint8)
    assert _generator().synth_titles(*arguments, None, _ptr(offsets), _ptr(flat)) == 0
    return flat[:int(offsets[-1])], offsets


def _make_queries(tables, seed, source, t_flat, t_off):
    count = source.shape[0]
    lengths = np.zeros(count, dtype=np.int32)
    arguments = (seed, count, _ptr(source), _ptr(t_flat), _ptr(t_off), _ptr(tables["neighbours"])) + \
        _vocabulary_arguments(tables)
    assert _generator().synth_queries(*arguments, _ptr(lengths), None, None) == 0
    offsets = np.zeros(count + 1, dtype=np.int64)
    np.cumsum(lengths, out=offsets[1:])
    flat = np.zeros(max(1, int(offsets[-1])), dtype=np.uint8)
    assert _generator().synth_queries(*arguments, None, _ptr(offsets), _ptr(flat)) == 0
    return flat[:int(offsets[-1])], offsets


def _to_strings(flat, offsets):
    text = "".join(ALLOWED_CHARACTERS[c] for c in range(_BASE))
    table = np.frombuffer(text.encode("ascii"), dtype=np.uint8)
    raw = table[flat].tobytes().decode("ascii")
    return [raw[offsets[i]:offsets[i + 1]] for i in range(offsets.shape[0] - 1)]


def _from_strings(titles):
    lengths = np.array([len(t) for t in titles], dtype=np.int64)
    offsets = np.concatenate(([0], np.cumsum(lengths)))
    table = np.zeros(256, dtype=np.uint8)
    for code, ch in enumerate(ALLOWED_CHARACTERS):
        table[ord(ch)] = code
    flat = table[np.frombuffer("".join(titles).encode("ascii"), dtype=np.uint8)]
    return flat, offsets


def make_truth(n_truth, seed=DEFAULT_SEED, vocabulary_size=None):
    """The truth side of a workload (depends on `seed` only, identical on every rank)."""
    rng = np.random.RandomState(seed)
    vocabulary = _Vocabulary(rng, vocabulary_size or max(20000, n_truth // 25))
    tables = _tables(vocabulary)
    t_flat, t_off = _make_titles(tables, seed, n_truth)
    return tables, t_flat, t_off


def make_workload(n_truth, n_queries, seed=DEFAULT_SEED, vocabulary_size=None, query_seed=None):
    """Truth titles depend on `seed` only (identical on every rank); queries on `query_seed` (default seed + 1)."""
    from . import _lib
    from .match_maker import NativeProblem
    tables, t_flat, t_off = make_truth(n_truth, seed, vocabulary_size)

    # ---- queries: 60 % misspelled truth titles, 40 % fresh titles, in random order
    query_seed = seed + 1 if query_seed is None else query_seed
    rng = np.random.RandomState(query_seed)
    n_edited = int(round(0.6 * n_queries))
    actual = np.full(n_queries, -1, dtype=np.int64)   # truth row the query was derived from (-1 = none)
    actual[rng.permutation(n_queries)[:n_edited]] = rng.randint(0, n_truth, n_edited)
    q_flat, q_off = _make_queries(tables, query_seed, actual, t_flat, t_off)

    # ---- MatchMaker.__init__ (match_maker.py:91-107) by the native index build; columns ascend with the tri-gram code
    problem = NativeProblem.from_flat(t_flat, t_off, q_flat, q_off, 3)
    arrays = problem.arrays(copy=False)   # views of the handle's memory: the workload keeps the handle alive

    # ---- encoded titles and truth word counts (feature_engineering.py:298-319)
    t_enc, t_len = encode_collection(t_flat, t_off)
    q_enc, q_len = encode_collection(q_flat, q_off)
    t_counts = truth_word_counts(t_flat, t_off)
    title_id = np.random.RandomState(seed + 2).permutation(n_truth).astype(np.int64)

    return SimpleNamespace(
        n_truth=n_truth, n_queries=n_queries, n_columns=int(arrays["idf32"].shape[0]), seed=seed,
        rowptr=arrays["rowptr"], truth_idx=arrays["truth_idx"], idf32=arrays["idf32"], idf64=arrays["idf64"],
        sums32=arrays["sums32"], q_rowptr=arrays["q_rowptr"], q_cols=arrays["q_cols"], q_maxint=arrays["q_maxint"],
        t_enc=t_enc, t_len=t_len, t_counts=t_counts, q_enc=q_enc, q_len=q_len,
        title_id=title_id, actual_row=actual,
        t_flat=t_flat, t_off=t_off, q_flat=q_flat, q_off=q_off, problem=problem)


def reorder_queries(w, order):
    """The workload with its queries in another order (query i of the result = query order[i]): experiments on the order in
    which a launch takes its queries.  The text of the queries (q_flat / q_off) is dropped."""
    order = np.asarray(order, dtype=np.int64)
    lengths = np.diff(w.q_rowptr)[order]
    rowptr = np.concatenate(([0], np.cumsum(lengths))).astype(np.int64)
    take = np.repeat(w.q_rowptr[:-1][order] - rowptr[:-1], lengths) + np.arange(rowptr[-1])
    fields = dict(vars(w))
    fields.update(q_rowptr=rowptr, q_cols=np.ascontiguousarray(w.q_cols[take]), q_maxint=np.ascontiguousarray(w.q_maxint[order]),
                  q_enc=np.ascontiguousarray(w.q_enc[order]), q_len=np.ascontiguousarray(w.q_len[order]),
                  actual_row=w.actual_row[order], q_flat=None, q_off=None)
    return SimpleNamespace(**fields)


_PUBLISHED = ("rowptr", "truth_idx", "idf32", "idf64", "sums32", "q_rowptr", "q_cols", "q_maxint", "t_enc", "t_len",
              "t_counts", "q_enc", "q_len", "title_id", "actual_row", "t_flat", "t_off", "q_flat", "q_off")


def publish_workload(w, directory):
    """Write a workload as .npy files (+ meta.json, last) into `directory` -- /dev/shm for the ranks of one node: rank 0
    generates the workload once, the other ranks map it (`load_workload`) instead of repeating minutes of host work and
    holding eight private copies of a 12.75 GB title table."""
    import json
    os.makedirs(directory, exist_ok=True)
    for name in _PUBLISHED:
        np.save(os.path.join(directory, name + ".npy"), np.asarray(getattr(w, name)))
    with open(os.path.join(directory, "meta.json.tmp"), "w") as handle:
        json.dump({"n_truth": int(w.n_truth), "n_queries": int(w.n_queries), "n_columns": int(w.n_columns),
                   "seed": int(w.seed)}, handle)
    os.replace(os.path.join(directory, "meta.json.tmp"), os.path.join(directory, "meta.json"))


def load_workload(directory):
    """The workload of `publish_workload`, arrays memory-mapped read-only (shared page cache between the ranks)."""
    import json
    with open(os.path.join(directory, "meta.json")) as handle:
        meta = json.load(handle)
    arrays = {name: np.load(os.path.join(directory, name + ".npy"), mmap_mode="r") for name in _PUBLISHED}
    return SimpleNamespace(**meta, **arrays)


def workload_statistics(w, positive_sample=0):
    """The acceptance numbers of SURVEY.md section 8d for a workload.  `positive_sample` > 0 also measures the share
    of truth titles with a positive score (at least one shared tri-gram) on that many evenly spaced queries."""
    posting_lengths = np.diff(w.rowptr)
    touched = posting_lengths[w.q_cols]
    per_query = np.add.reduceat(touched, w.q_rowptr[:-1][np.diff(w.q_rowptr) > 0]) if touched.shape[0] else touched
    stats = {
        "tri_grams_per_truth_title": float(w.rowptr[-1]) / w.n_truth,
        "tri_grams_per_query": float(w.q_rowptr[-1]) / w.n_queries,
        "columns": int(w.n_columns),
        "postings_touched_per_query_over_n": float(per_query.mean()) / w.n_truth if per_query.shape[0] else 0.0,
        "chars_per_truth_title": float(w.t_len.mean()),
        "words_per_truth_title": float((w.t_counts > 0).sum(axis=1).mean()),
    }
    if positive_sample > 0:
        shares = []
        hit = np.zeros(w.n_truth, dtype=bool)
        for q in np.linspace(0, w.n_queries - 1, min(positive_sample, w.n_queries)).astype(np.int64):
            hit[:] = False
            for column in w.q_cols[w.q_rowptr[q]:w.q_rowptr[q + 1]]:
                hit[w.truth_idx[w.rowptr[column]:w.rowptr[column + 1]]] = True
            shares.append(hit.mean())
        stats["positive_score_fraction"] = float(np.mean(shares))
    return stats


def algorithmic_bytes_jaccard(w, k, q_begin=0, q_end=None):
    """B_jac summed over the queries [q_begin, q_end): 4*sum|P_g| + 4*N + 16*|G_q| + 4*k per query (SURVEY.md section
    8d) -- what the REFERENCE's algorithm reads."""
    q_end = w.n_queries if q_end is None else q_end
    first, last = int(w.q_rowptr[q_begin]), int(w.q_rowptr[q_end])
    posting_lengths = np.diff(w.rowptr)
    touched = int(posting_lengths[w.q_cols[first:last]].sum())
    return 4 * touched + (q_end - q_begin) * (4 * w.n_truth + 4 * k) + 16 * (last - first)


def make_forest(seed=DEFAULT_SEED, n_trees=300, depth=6, n_features=66):
    """A random complete-binary-tree ensemble in the flat layout of `ForestModel` (bench.py's stand-in for the pickled
    booster of predict.py:80-82: the real model file is not part of the reference tree)."""
    rng = np.random.RandomState(seed)
    size = 2 ** (depth + 1) - 1
    inner = 2 ** depth - 1
    feature = np.full((n_trees, size), -1, dtype=np.int32)
    feature[:, :inner] = rng.randint(0, n_features, (n_trees, inner))
    threshold = rng.normal(0, 0.1, (n_trees, size)).astype(np.float32)          # leaves
    threshold[:, :inner] = rng.uniform(0, 100, (n_trees, inner)).astype(np.float32)
    node = np.arange(size, dtype=np.int32)
    yes = np.tile(np.where(node < inner, 2 * node + 1, 0), (n_trees, 1)).astype(np.int32)
    no = np.tile(np.where(node < inner, 2 * node + 2, 0), (n_trees, 1)).astype(np.int32)
    missing = np.where(rng.rand(n_trees, size) < 0.5, yes, no).astype(np.int32)
    return dict(feature=feature.reshape(-1), threshold=threshold.reshape(-1), yes=yes.reshape(-1), no=no.reshape(-1),
                missing=missing.reshape(-1), tree_offsets=np.arange(n_trees + 1, dtype=np.int64) * size,
                base_margin=0.0, n_features=n_features)
