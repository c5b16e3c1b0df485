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

Measured at N = 500k (seed 20260101; SURVEY.md 8d targets in brackets): 21.2 tri-grams per title (21 +- 2), 50.3k
tri-gram columns (~50k), postings touched per query 0.90 N (0.9 N +- 0.15 N), truth titles with a positive score per
query 0.31 N (~0.33 N), 23.5 characters and 3.5 words per title (23.4, 3.5) -- asserted by
tests/test_host_cpu.py::test_synthetic_workload_acceptance_at_c2_truth_size.

`make_workload` returns everything both kernels consume, built the way MatchMaker.__init__ /
FeatureEngineering.encode_title build them (match_maker.py:84-181, feature_engineering.py:298-319), with the
column order fixed to ascending tri-gram code instead of Python-set order:
    rowptr / truth_idx / idf32 / sums32          truth inverted index (CSR over tri-gram columns)
    q_rowptr / q_cols / q_maxint                 query rows (ascending column ids) and max_intersection_possible
    q_enc, q_len, t_enc, t_len, t_counts         encoded titles (uint8[*, 255]), lengths, truth word counts
"""
import math
from types import SimpleNamespace

import numpy as np

from .feature_engineering import ALLOWED_CHARACTERS, MAX_CHARACTERS_ALLOWED_IN_THE_TITLE, NUMBER_OF_WORDS_FEATURES
from .match_maker import sequential_sums

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


def _ragged_arange(lengths):
    """[0..l0), [0..l1), ... concatenated, plus the row id of every element."""
    lengths = np.asarray(lengths, dtype=np.int64)
    total = int(lengths.sum())
    starts = np.cumsum(lengths) - lengths
    rows = np.repeat(np.arange(lengths.shape[0], dtype=np.int64), lengths)
    return np.arange(total, dtype=np.int64) - starts[rows], rows


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


def _make_titles(rng, vocabulary, count):
    """Titles as ragged arrays of words -> (flat codes, offsets, word ids per title as (flat, offsets), word table).

    Word ids index the returned word table = the vocabulary's words followed by this call's hapaxes."""
    n_words = np.clip(1 + rng.poisson(2.5, count), 1, 20).astype(np.int64)
    has_suffix = rng.rand(count) < SUFFIX_SHARE
    body_words = np.where(has_suffix & (n_words > 1), n_words - 1, n_words)
    position, row = _ragged_arange(n_words)
    word_ids = vocabulary.sample(rng, position.shape[0])
    suffix_choice = rng.choice(len(_SUFFIXES), size=count, p=_SUFFIX_WEIGHTS)
    is_suffix_slot = has_suffix[row] & (n_words[row] > 1) & (position == body_words[row])
    word_ids = np.where(is_suffix_slot, suffix_choice[row], word_ids)
    # hapaxes: fresh words of 4..20 uniformly random characters, appended to the word table
    hapax_slot = (rng.rand(position.shape[0]) < HAPAX_FRACTION) & ~is_suffix_slot
    n_hapax = int(hapax_slot.sum())
    width = vocabulary.chars.shape[1]
    hapax_lengths = np.clip(np.rint(rng.lognormal(1.7, 0.45, n_hapax)), 4, width).astype(np.int64)
    hapax_chars = np.where(rng.rand(n_hapax, width) < HAPAX_DIGIT_SHARE, rng.randint(28, 38, (n_hapax, width)),
                           rng.randint(2, 28, (n_hapax, width))).astype(np.uint8)
    hapax_chars[np.arange(width)[None, :] >= hapax_lengths[:, None]] = 0
    word_ids = word_ids.copy()
    word_ids[hapax_slot] = vocabulary.size + np.arange(n_hapax)
    table_chars = np.concatenate([vocabulary.chars, hapax_chars])
    table_lengths = np.concatenate([vocabulary.lengths, hapax_lengths])
    # keep the leading words that fit in 255 characters
    lengths = table_lengths[word_ids]
    starts = np.cumsum(n_words) - n_words
    running = np.cumsum(lengths + 1)
    before = np.concatenate(([0], running))[starts]
    end_in_title = running - before[row] - 1  # position of the word's last char + 1 within the title
    fits = end_in_title <= MAX_CHARACTERS_ALLOWED_IN_THE_TITLE
    word_ids, row, lengths = word_ids[fits], row[fits], lengths[fits]
    n_words = np.bincount(row, minlength=count).astype(np.int64)
    word_offsets = np.concatenate(([0], np.cumsum(n_words)))
    first_of_title = np.zeros(word_ids.shape[0], dtype=bool)
    first_of_title[word_offsets[:-1]] = True
    # destination of every word inside the flat character array (one space before every non-first word)
    piece = lengths + (~first_of_title)
    piece_start = np.cumsum(piece) - piece
    title_lengths = np.bincount(row, weights=piece, minlength=count).astype(np.int64)
    offsets = np.concatenate(([0], np.cumsum(title_lengths)))
    flat = np.full(int(offsets[-1]), _SPACE, dtype=np.uint8)
    within, word_row = _ragged_arange(lengths)
    destination = (piece_start + (~first_of_title))[word_row] + within
    flat[destination] = table_chars[word_ids[word_row], within]
    return flat, offsets, word_ids, word_offsets, table_chars


def _to_strings(flat, offsets):
    text = "".join(ALLOWED_CHARACTERS[c] for c in range(_BASE))
    table = np.frombuffer(text.encode("ascii"), dtype=np.uint8)
    raw = table[flat].tobytes().decode("ascii")
    return [raw[offsets[i]:offsets[i + 1]] for i in range(offsets.shape[0] - 1)]


def _neighbour(rng, character):
    for row in _KEYBOARD_ROWS:
        at = row.find(character)
        if at >= 0:
            candidates = [row[i] for i in (at - 1, at + 1) if 0 <= i < len(row)]
            return candidates[rng.randint(len(candidates))]
    return "e"


def _misspell(rng, title):
    """1-2 edits: delete / insert neighbour / replace with neighbour / insert space / remove space / swap words."""
    for _ in range(1 + int(rng.rand() < 0.4)):
        kind = rng.randint(6)
        at = rng.randint(len(title))
        if kind == 0 and len(title) > 4:
            title = title[:at] + title[at + 1:]
        elif kind == 1:
            title = title[:at] + _neighbour(rng, title[at]) + title[at:]
        elif kind == 2 and title[at] != " ":
            title = title[:at] + _neighbour(rng, title[at]) + title[at + 1:]
        elif kind == 3 and 0 < at < len(title) - 1 and title[at] != " " and title[at - 1] != " ":
            title = title[:at] + " " + title[at:]
        elif kind == 4 and " " in title:
            spaces = [i for i, ch in enumerate(title) if ch == " "]
            cut = spaces[rng.randint(len(spaces))]
            title = title[:cut] + title[cut + 1:]
        elif kind == 5 and " " in title:
            words = title.split(" ")
            i = rng.randint(len(words) - 1)
            words[i], words[i + 1] = words[i + 1], words[i]
            title = " ".join(words)
    title = " ".join(title.split())[:MAX_CHARACTERS_ALLOWED_IN_THE_TITLE].strip()
    return title if len(title) >= 3 else title.rjust(3, "0")  # common.py:34-38


def _from_strings(titles):
    lengths = np.array([len(t) for t in titles], dtype=np.int64)
    offsets = np.concatenate(([0], np.cumsum(lengths)))
    table = np.zeros(256, dtype=np.uint8)
    for code, ch in enumerate(ALLOWED_CHARACTERS):
        table[ord(ch)] = code
    flat = table[np.frombuffer("".join(titles).encode("ascii"), dtype=np.uint8)]
    return flat, offsets


def _tri_grams(flat, offsets):
    """Unique (title, tri-gram code) pairs, sorted by title then code (common.py:150-151 get_n_grams)."""
    lengths = np.diff(offsets)
    n_grams = np.maximum(lengths - 2, 0)
    within, row = _ragged_arange(n_grams)
    at = offsets[:-1][row] + within
    codes = (flat[at].astype(np.int64) * _BASE + flat[at + 1]) * _BASE + flat[at + 2]
    keys = np.unique(row * (_BASE ** 3) + codes)
    return keys // (_BASE ** 3), keys % (_BASE ** 3)


def _encode(flat, offsets):
    count = offsets.shape[0] - 1
    lengths = np.diff(offsets)
    enc = np.zeros((count, MAX_CHARACTERS_ALLOWED_IN_THE_TITLE), dtype=np.uint8)
    within, row = _ragged_arange(lengths)
    enc[row, within] = flat
    return enc, lengths.astype(np.uint8)


def make_workload(n_truth, n_queries, seed=DEFAULT_SEED, vocabulary_size=None, query_seed=None):
    """Truth titles depend on `seed` only (identical on every rank); queries on `query_seed` (default seed + 1)."""
    rng = np.random.RandomState(seed)
    vocabulary = _Vocabulary(rng, vocabulary_size or max(20000, n_truth // 25))
    t_flat, t_off, t_words, t_word_off, t_word_table = _make_titles(rng, vocabulary, n_truth)

    # ---- queries: 60 % misspelled truth titles, 40 % fresh titles
    title_id = rng.permutation(n_truth).astype(np.int64)
    rng = np.random.RandomState(seed + 1 if query_seed is None else query_seed)
    n_edited = int(round(0.6 * n_queries))
    source = rng.randint(0, n_truth, n_edited)
    truth_strings_needed = _to_strings(
        np.concatenate([t_flat[t_off[s]:t_off[s + 1]] for s in source]) if n_edited else np.zeros(0, np.uint8),
        np.concatenate(([0], np.cumsum(t_off[source + 1] - t_off[source]))) if n_edited else np.zeros(1, np.int64))
    edited = [_misspell(rng, title) for title in truth_strings_needed]
    f_flat, f_off, _, _, _ = _make_titles(rng, vocabulary, n_queries - n_edited)
    fresh = _to_strings(f_flat, f_off)
    order = rng.permutation(n_queries)
    query_strings = [None] * n_queries
    actual = np.full(n_queries, -1, dtype=np.int64)  # truth row the query was derived from (-1 = none)
    for slot, title in zip(order[:n_edited], edited):
        query_strings[slot] = title
    actual[order[:n_edited]] = source
    for slot, title in zip(order[n_edited:], fresh):
        query_strings[slot] = title
    q_flat, q_off = _from_strings(query_strings)

    # ---- MatchMaker.__init__ (match_maker.py:91-107) with ascending tri-gram codes as the column order
    t_row, t_code = _tri_grams(t_flat, t_off)
    q_row, q_code = _tri_grams(q_flat, q_off)
    vocabulary_codes = np.union1d(t_code, q_code)
    n_columns = vocabulary_codes.shape[0]
    t_col = np.searchsorted(vocabulary_codes, t_code)
    q_col = np.searchsorted(vocabulary_codes, q_code)
    df = np.bincount(t_col, minlength=n_columns)
    idf64 = np.full(n_columns, 0.0)
    seen = df > 0
    idf64[seen] = [math.log(n_truth / int(c)) for c in df[seen]]        # match_maker.py:135-139
    idf64[~seen] = idf64[seen].max()                                     # :95, :151
    idf32 = idf64.astype(np.float32)

    t_counts_per_row = np.bincount(t_row, minlength=n_truth)
    sums32 = sequential_sums(idf32[t_col], t_counts_per_row, np.float32)  # :174
    keep = idf32[t_col] != 0
    order_by_column = np.argsort(t_col[keep], kind="stable")
    truth_idx = t_row[keep][order_by_column].astype(np.int32)
    rowptr = np.concatenate(([0], np.cumsum(np.bincount(t_col[keep], minlength=n_columns)))).astype(np.int64)

    keep_q = idf32[q_col] != 0
    q_row, q_col = q_row[keep_q], q_col[keep_q]
    q_counts = np.bincount(q_row, minlength=n_queries)
    q_rowptr = np.concatenate(([0], np.cumsum(q_counts))).astype(np.int64)
    q_maxint = sequential_sums(idf64[q_col], q_counts, np.float64)       # :197

    # ---- encoded titles and truth word counts (feature_engineering.py:298-319)
    t_enc, t_len = _encode(t_flat, t_off)
    q_enc, q_len = _encode(q_flat, q_off)
    word_row = np.repeat(np.arange(n_truth, dtype=np.int64), np.diff(t_word_off))
    # the reference counts words as strings (common.py:140-142): two hapaxes spelled alike are one word
    used, t_words = np.unique(t_words, return_inverse=True)
    spelled = np.ascontiguousarray(t_word_table[used]).view(np.dtype((np.void, t_word_table.shape[1]))).reshape(-1)
    _, canonical = np.unique(spelled, return_inverse=True)
    t_words = canonical[t_words]
    n_distinct_words = int(canonical.max()) + 1 if canonical.shape[0] else 0
    pairs = np.unique(word_row * n_distinct_words + t_words)
    word_df = np.bincount(pairs % n_distinct_words, minlength=n_distinct_words)
    slot, _ = _ragged_arange(np.diff(t_word_off))
    first = slot < NUMBER_OF_WORDS_FEATURES
    t_counts = np.zeros((n_truth, NUMBER_OF_WORDS_FEATURES), dtype=np.uint32)
    t_counts[word_row[first], slot[first]] = word_df[t_words[first]]

    return SimpleNamespace(
        n_truth=n_truth, n_queries=n_queries, n_columns=n_columns, seed=seed,
        rowptr=rowptr, truth_idx=truth_idx, idf32=idf32, idf64=idf64, sums32=sums32,
        q_rowptr=q_rowptr, q_cols=q_col.astype(np.int32), q_maxint=q_maxint,
        t_enc=t_enc, t_len=t_len, t_counts=t_counts, q_enc=q_enc, q_len=q_len,
        title_id=title_id, actual_row=actual,
        t_flat=t_flat, t_off=t_off, q_flat=q_flat, q_off=q_off)


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


def algorithmic_bytes_jaccard(w, k):
    """B_jac summed over the queries: 4*sum|P_g| + 4*N + 16*|G_q| + 4*k per query."""
    posting_lengths = np.diff(w.rowptr)
    touched = int(posting_lengths[w.q_cols].sum())
    return 4 * touched + w.n_queries * (4 * w.n_truth + 4 * k) + 16 * int(w.q_rowptr[-1])


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
