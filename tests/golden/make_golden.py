#!/usr/bin/env python3
"""Golden-vector generator for the doppel-speller hot path (run in the BUILD container only).

What this does
--------------
Imports the reference package from /root/reference (read-only) and runs ITS OWN function bodies
(`fast_jaccard`, `fast_arg_top_k`, `MatchMaker`, `fast_levenshtein_ratio`, `construct_features`,
`FeatureEngineering.encode_title` ...) on slices of the example data set that ships with the reference, then
stores inputs + outputs at the two kernel boundaries as small .npz/.json fixtures next to this file.

The reference needs `numba` and `Levenshtein`, neither of which is installed here (no network).  They are
replaced, in THIS process only, by the tiny modules built in `_install_shims()`:

* `numba.njit` / `numba.guvectorize` become pass-through decorators, so the decorated reference functions execute
  as plain Python/NumPy.  Two pieces of numba *typing* are emulated because they change results:
    - a Python `float` argument is widened to `np.float64` (numba types it float64; NumPy-2 would treat it as a
      weak scalar and compute `match_maker.py:50` in float32);
    - an explicit return type in a signature (`numba.uint8(...)`, `feature_engineering.py:25`) is applied as a cast.
* `Levenshtein.ratio` raises: it is not on the hot path (only `common.py:161-167` uses it).

Remaining differences between "reference source under NumPy" (what is captured here) and "reference source under
numba 0.45" are listed in SURVEY.md section 8c and in oracle/README.md; every captured vector carries a
`*_margin_ok` flag saying whether it is insensitive to those differences.  Nothing is written under /root/reference
(PYTHONDONTWRITEBYTECODE is forced, the example CSVs are unpacked into a temp dir).

Usage:  PYTHONHASHSEED=0 python tests/golden/make_golden.py            (every fixture)
        PYTHONHASHSEED=0 python tests/golden/make_golden.py --only edge (sections E and F only: construct_features_edge.npz, arg_top_k_cases.npz)
"""
import os
import sys

if os.environ.get("PYTHONHASHSEED") != "0":
    # vocabulary order is `enumerate(set(...))` (match_maker.py:145-147): pin it.
    os.environ["PYTHONHASHSEED"] = "0"
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    os.execv(sys.executable, [sys.executable] + sys.argv)

import gzip
import json
import shutil
import tempfile
import types
import warnings

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = "/root/reference"


def _install_shims():
    numba = types.ModuleType("numba")

    class _Type:
        def __init__(self, np_type):
            self.np_type = np_type

        def __getitem__(self, item):  # numba.uint8[:]
            return self

        def __call__(self, *args):  # numba.uint8(numba.uint8[:], numba.uint8[:]) -> signature
            return _Signature(self)

    class _Signature:
        def __init__(self, return_type):
            self.return_type = return_type

    for name in ("uint8", "uint32", "int32", "int64", "float32", "float64"):
        setattr(numba, name, _Type(getattr(np, name)))

    def njit(*args, **kwargs):
        signature = args[0] if args and isinstance(args[0], _Signature) else None

        def decorate(function):
            def wrapper(*call_args):
                call_args = [np.float64(a) if type(a) is float else a for a in call_args]
                out = function(*call_args)
                if signature is not None:
                    with np.errstate(all="ignore"):
                        out = signature.return_type.np_type(out)
                return out

            wrapper.__wrapped__ = function
            wrapper.__name__ = function.__name__
            return wrapper

        if args and callable(args[0]) and not isinstance(args[0], _Signature):
            return decorate(args[0])
        return decorate

    def guvectorize(signatures, layout, **kwargs):
        def decorate(function):
            def wrapper(*call_args):
                *inputs, response = call_args
                n = response.shape[0]
                core_dims = [len(x.strip("()").split(",")) if x.strip("()") else 0
                             for x in layout.split("->")[0].split("),(")]
                for i in range(n):
                    row = []
                    for value, core in zip(inputs, core_dims):
                        value = np.asarray(value)
                        row.append(value[i] if value.ndim > core else (value[()] if core == 0 else value))
                    function(*row, response[i])

            wrapper.__wrapped__ = function
            return wrapper

        return decorate

    typed = types.ModuleType("numba.typed")
    typed.List = list
    numba.njit = njit
    numba.guvectorize = guvectorize
    numba.typed = typed
    sys.modules["numba"] = numba
    sys.modules["numba.typed"] = typed

    levenshtein = types.ModuleType("Levenshtein")

    def ratio(*_):
        raise NotImplementedError("python-Levenshtein is not available; not on the hot path")

    levenshtein.ratio = ratio
    sys.modules["Levenshtein"] = levenshtein


def _stage_example_data(directory):
    for name in ("example_truth", "example_test", "example_train"):
        with gzip.open(f"{REFERENCE}/example_dataset/{name}.csv.gz", "rb") as source, \
                open(f"{directory}/{name}.csv", "wb") as target:
            shutil.copyfileobj(source, target)


def _set_order_ranks(n_gram_sets, encoding):
    """The iteration order of every title's n-gram set (what `sum(uniqueness_values)` of match_maker.py:174 adds in),
    stored compactly: for the i-th element of the iteration, its rank among the title's ascending column ids."""
    ranks = []
    for n_grams in n_gram_sets:
        columns = [encoding[n_gram] for n_gram in n_grams]
        position = {column: rank for rank, column in enumerate(sorted(columns))}
        ranks.extend(position[column] for column in columns)
    assert max(ranks) < 256
    return np.array(ranks, dtype=np.uint8)


def _match_maker_answers(match_maker, s, mm, q_maxint, n_query, top_n):
    rows = np.zeros((n_query, top_n), dtype=np.int32)
    margin_ok = np.zeros(n_query, dtype=bool)
    for q in range(n_query):
        jaccard = match_maker.fast_jaccard(
            mm.number_of_truth_titles, float(q_maxint[q]), mm.matrix_non_zero_columns[q],
            mm.matrix_truth_non_zero_columns_and_values, mm.sums_matrix_truth)
        rows[q] = match_maker.fast_arg_top_k(jaccard, top_n)
        positive = np.sort(jaccard[jaccard > 0])[::-1]
        kth = np.float32(positive[top_n - 1]) if positive.shape[0] >= top_n else np.float32(0)
        threshold = np.float64(kth) - np.float64(np.float32(s.ENCODING_FLOAT_BUFFER))
        margin_ok[q] = np.abs(jaccard - threshold).min() > 5e-7
    return rows, margin_ok


def _edge_section(feature_engineering, encoding, alphabet):
    """E. construct_features on pairs the example data does not contain: leading / trailing / repeated spaces, more
    than 15 words, one-character titles, 255-character titles, words longer than the other title, identical and
    disjoint titles, zero and huge word counts, n_truth = 1.  Inputs are built AT the kernel boundary (encoded arrays),
    the answers come from the reference's own construct_features."""
    rng = np.random.RandomState(777)
    letters = [ch for ch in alphabet if ch not in (alphabet[0], " ")]

    def word(n):
        return "".join(rng.choice(letters) for _ in range(n))

    def sentence(n_words, longest=9):
        return " ".join(word(rng.randint(1, longest + 1)) for _ in range(n_words))

    texts = [
        ("a", "a"), ("a", "b"), ("a", "ab cd"), ("ab cd", "a"), (" a", "a "), ("a  b", "a b"), ("  ", "a"), ("a", "   "),
        (" leading", "leading"), ("trailing ", "trailing"), ("two  spaces  here", "two spaces here"),
        ("x" * 127, "x" * 128), ("x" * 128, "y" * 127), ("ab " * 42, "ab " * 42), ("a " * 63 + "a", "a " * 63 + "b"),
        ("abc", "x" * 200), ("x" * 200, "abc"), ("abcdefghij" * 20, "abcdefghij"), ("abcdefghij", "abcdefghij" * 20),
        ("x" * 64, "x" * 65), ("x" * 65, "xy" * 32), ("ab" * 60, "ba" * 60), ("a" * 254, "a"), ("a", "b" * 254),
        (sentence(16), sentence(16)), (sentence(20, 4), sentence(20, 4)), (sentence(15), sentence(15)),
        (sentence(14), sentence(14)), (sentence(1, 60), sentence(1, 60)), (sentence(30, 3), sentence(2)), (sentence(40, 2), sentence(40, 2)),
        ("the same title ltd", "the same title ltd"), ("same words other order", "order other words same"),
        ("0123456789", "9876543210"), ("a b c d e f g h i j k l m n o p q", "a b c d e f g h i j k l m n o p q"),
        ("a b c d e f g h i j k l m n o p q", "q p o n m l k j i h g f e d c b a"),
    ]
    for _ in range(90):
        base = sentence(rng.randint(1, 8))
        other = list(base)
        for _ in range(rng.randint(0, 6)):
            at = rng.randint(len(other) + 1)
            action = rng.randint(3)
            if action == 0 and len(other) > 1:
                del other[min(at, len(other) - 1)]
            elif action == 1:
                other.insert(at, rng.choice(letters + [" "]))
            else:
                other.insert(at, " ")
        texts.append(("".join(other)[:255] or "a", base))
    # len(q) + len(t) <= 255: beyond that numba wraps the uint8 total (H5) while NumPy 2 raises OverflowError, so the
    # reference cannot be run here on such pairs (the oracle follows the numba semantics; tests/test_gpu_features.py)
    texts = [(q, t) for q, t in texts if len(q) > 0 and len(t) > 0 and len(q) + len(t) <= 255]

    def encode(text):
        out = np.zeros(255, dtype=np.uint8)
        out[:len(text)] = [encoding[ch] for ch in text]
        return out

    n = len(texts)
    title_len = np.array([len(q) for q, _ in texts], dtype=np.uint8)
    truth_len = np.array([len(t) for _, t in texts], dtype=np.uint8)
    title_enc = np.vstack([encode(q) for q, _ in texts])
    truth_enc = np.vstack([encode(t) for _, t in texts])
    counts = rng.randint(1, 30000, (n, 15)).astype(np.uint32)
    counts[rng.rand(n, 15) < 0.15] = 0
    counts[rng.rand(n, 15) < 0.10] = 1
    counts[rng.rand(n, 15) < 0.05] = 4000000000
    blocks = []
    for n_truth in (np.uint32(30000), np.uint32(1), np.uint32(4000000000)):
        features = np.zeros((n, feature_engineering.FEATURES_COUNT), dtype=np.float32)
        dummy = np.zeros((feature_engineering.FEATURES_COUNT,), dtype=np.uint8)
        with np.errstate(all="ignore"):
            feature_engineering.construct_features(title_len, truth_len, title_enc, truth_enc, counts,
                                                   np.uint8(encoding[" "]), n_truth, dummy, features)
        blocks.append(features)
    np.savez_compressed(
        f"{HERE}/construct_features_edge.npz", title_len=title_len, truth_len=truth_len, title_enc=title_enc,
        truth_enc=truth_enc, counts=counts, space_code=np.uint8(encoding[" "]),
        n_truth=np.array([30000, 1, 4000000000], dtype=np.uint32), features=np.stack(blocks),
        titles=np.array([q for q, _ in texts]), truth_titles=np.array([t for _, t in texts]))
    print("construct_features edge pairs:", n, "x 3 n_truth values; NaNs:", int(np.isnan(np.stack(blocks)).sum()))


def _arg_top_k_section(match_maker):
    """F. fast_arg_top_k of the reference on crafted float64 arrays: ties at the k-th value, fewer than k positive values,
    all zeros, negatives, values within float32 resolution of the k-th, k = 1 / k = len / k > len, few distinct values."""
    rng = np.random.RandomState(4711)
    cases = []

    def add(array, k):
        cases.append((np.asarray(array, dtype=np.float64), int(k)))

    add([0.5, 0.25, 0.75, 0.1, 0.9], 1); add([0.5, 0.25, 0.75, 0.1, 0.9], 3); add([0.5, 0.25, 0.75, 0.1, 0.9], 5)
    add([0.5, 0.25, 0.75, 0.1, 0.9], 8)                      # k > len
    add(np.zeros(50), 10); add(np.full(50, 0.3), 10)         # nothing positive / everything tied
    add([0.0, 0.2, 0.0, 0.0, 0.4, 0.0], 4)                   # fewer than k positive values
    add([-1.0, 0.2, -0.5, 0.0, 0.4, -2.0], 2); add([-1.0, -0.2, -0.5], 2)
    ties = np.zeros(400); ties[rng.choice(400, 60, replace=False)] = 0.625; ties[rng.choice(400, 5, replace=False)] = 0.8
    for k in (1, 5, 6, 10, 64, 100):
        add(ties, k)
    base = rng.rand(600)
    for k in (1, 10, 50, 600):
        add(base, k)
    kth = np.sort(base)[::-1][9]
    near = base.copy()
    near[rng.choice(600, 40, replace=False)] = kth - rng.choice([1e-9, 4e-7, 9.9e-7, 1.0e-6, 1.1e-6, 2e-6], 40)
    near[rng.choice(600, 10, replace=False)] = kth + rng.choice([1e-9, 1e-7, 1e-6], 10)
    add(near, 10); add(near, 11); add(near.astype(np.float32).astype(np.float64), 10)
    for levels in (2, 3, 7):
        few = rng.randint(0, levels, 900) / float(levels)
        for k in (1, 10, 100, 512):
            add(few, k)
    sparse = np.zeros(5000); sparse[rng.choice(5000, 300, replace=False)] = rng.rand(300) ** 3
    for k in (10, 100, 299, 300, 301, 512):
        add(sparse, k)
    arrays, ks, answers = [], [], []
    for array, k in cases:
        out = np.asarray(match_maker.fast_arg_top_k(array, k), dtype=np.int64)
        arrays.append(array); ks.append(k); answers.append(out)
    np.savez_compressed(
        f"{HERE}/arg_top_k_cases.npz", k=np.array(ks, dtype=np.int32),
        array_offsets=np.cumsum([0] + [a.shape[0] for a in arrays]).astype(np.int64), arrays=np.concatenate(arrays),
        answer_offsets=np.cumsum([0] + [a.shape[0] for a in answers]).astype(np.int64),
        answers=np.concatenate(answers) if answers else np.zeros(0, np.int64))
    print("fast_arg_top_k cases:", len(cases), "answers shorter than k:", sum(a.shape[0] < k for a, k in zip(answers, ks)))


def main():
    data_dir = tempfile.mkdtemp(prefix="ds_golden_")
    _stage_example_data(data_dir)
    os.environ["PROJECT_DATA_PATH"] = data_dir
    _install_shims()
    sys.path.insert(0, REFERENCE)
    warnings.simplefilter("ignore")

    import doppelspeller.settings as s
    import doppelspeller.constants as c
    from doppelspeller import common, match_maker, feature_engineering

    truth_all = common.get_ground_truth()
    test_all = common.get_test_data()

    # ------------------------------------------------------------------ A. Levenshtein / encoding known answers
    alphabet = f"{s.R_FILL_CHARACTER} abcdefghijklmnopqrstuvwxyz0123456789"
    encoding = {ch: i for i, ch in enumerate(alphabet)}

    def encode(text):
        return np.array([encoding[ch] for ch in text], dtype=np.uint8)

    pairs = [
        ("coolblue bv", "coolblu bv"), ("abc", "abc"), ("abc", "xyz"), ("a", "ab"), ("kitten", "sitting"),
        ("limited", "ltd"), ("systematica imnvestments services limited", "systematica investment services limited"),
        ("feld s ullivan limited", "feld sullivan limited"), ("a" * 29 + "b" * 21, "a" * 29 + "c" * 21),
        ("a", "a"), ("a", "b"), ("ab", "ba"), ("0", "0123456789"),
    ]
    rng = np.random.RandomState(12345)
    truth_titles = list(truth_all[c.COLUMN_TRANSFORMED_TITLE])
    test_titles = list(test_all[c.COLUMN_TRANSFORMED_TITLE])
    for _ in range(400):
        pairs.append((test_titles[rng.randint(len(test_titles))], truth_titles[rng.randint(len(truth_titles))]))
    for _ in range(100):  # word-sized strings, as in the window loop of construct_features
        a = rng.choice(truth_titles).split()[0]
        b = rng.choice(test_titles).replace(" ", "")
        start = rng.randint(max(1, len(b)))
        pairs.append((b[start:start + len(a)] or "a", a))
    lev = []
    for a, b in pairs:
        if len(a) + len(b) > 255:
            continue
        r1 = int(feature_engineering.fast_levenshtein_ratio(encode(a), encode(b)))
        r2 = int(feature_engineering.fast_levenshtein_ratio(encode(b), encode(a)))
        assert r1 == r2
        lev.append({"a": a, "b": b, "ratio": r1})

    fe = feature_engineering.FeatureEngineering.__new__(feature_engineering.FeatureEngineering)
    fe.allowed_characters = alphabet
    fe.encoding = encoding
    fe.words_counter = common.get_words_counter(truth_all)
    kat = {
        "alphabet": alphabet,
        "space_code": encoding[" "],
        "levenshtein": lev,
        "encode_title": {t: fe.encode_title(t)[:len(t) + 2].tolist() for t in ("coolblue bv 42", "a", test_titles[0])},
        "n_grams": {t: sorted(common.get_n_grams(t, s.N_GRAMS)) for t in ("coolblue bv", "abc", "ab", test_titles[3])},
        "transform_title": {t: common.transform_title(t) for t in
                            ("Great Expectations Ministries", "Topdrill  Ltd.", "Ünïcode-Näme B.V.", "A")},
        "truth_words_counts": {t: fe.get_truth_words_counts(t).tolist() for t in truth_titles[:5]},
        "settings": {"N_GRAMS": s.N_GRAMS, "NUMBER_OF_WORDS_FEATURES": s.NUMBER_OF_WORDS_FEATURES,
                     "MAX_CHARACTERS": int(s.MAX_CHARACTERS_ALLOWED_IN_THE_TITLE),
                     "ENCODING_FLOAT_BUFFER_f32_bits": int(np.float32(s.ENCODING_FLOAT_BUFFER).view(np.uint32)),
                     "FEATURES_COUNT": feature_engineering.FEATURES_COUNT},
    }
    only_edge = "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "edge"
    if only_edge:
        _edge_section(feature_engineering, encoding, alphabet)
        _arg_top_k_section(match_maker)
        shutil.rmtree(data_dir)
        return
    with open(f"{HERE}/kat.json", "w") as handle:
        json.dump(kat, handle, indent=1, sort_keys=True)

    # ------------------------------------------------------------------ B. MatchMaker at the kernel boundary
    n_truth, n_query = 5000, 200
    truth = truth_all.iloc[:n_truth].reset_index(drop=True)
    query = test_all.iloc[:n_query].reset_index(drop=True)
    fixture = {}
    for top_n in (10, 100):
        mm = match_maker.MatchMaker(query.copy(), truth.copy(), top_n)
        vocab = len(mm.n_grams_decoding)
        if top_n == 10:
            rowptr = np.zeros(vocab + 1, dtype=np.int64)
            for g in range(vocab):
                rowptr[g + 1] = rowptr[g] + len(mm.matrix_truth_non_zero_columns_and_values[g][0])
            truth_idx = np.concatenate([np.asarray(x[0], dtype=np.int32)
                                        for x in mm.matrix_truth_non_zero_columns_and_values])
            idf32 = np.array([mm._get_idf_given_index(g) for g in range(vocab)], dtype=np.float32)
            for g in range(vocab):  # the per-posting value array is the constant idf (match_maker.py:130)
                values = mm.matrix_truth_non_zero_columns_and_values[g][1]
                assert values.dtype == np.float32 and (values == idf32[g]).all()
            q_rowptr = np.zeros(n_query + 1, dtype=np.int64)
            for q in range(n_query):
                q_rowptr[q + 1] = q_rowptr[q] + len(mm.matrix_non_zero_columns[q])
            q_cols = np.concatenate([np.asarray(x, dtype=np.int32) for x in mm.matrix_non_zero_columns])
            q_maxint = np.array([sum([mm._get_idf_given_index(r) for r in mm.matrix_non_zero_columns[q]])
                                 for q in range(n_query)], dtype=np.float64)
            fixture.update(rowptr=rowptr, truth_idx=truth_idx, idf32=idf32,
                           sums32=mm.sums_matrix_truth.astype(np.float32), q_rowptr=q_rowptr, q_cols=q_cols,
                           q_maxint=q_maxint, title_id=np.asarray(truth[c.COLUMN_TITLE_ID], dtype=np.int64),
                           vocab=np.array(sorted(mm.n_grams_encoding, key=mm.n_grams_encoding.get)),
                           truth_titles=np.array(list(truth[c.COLUMN_TRANSFORMED_TITLE])),
                           query_titles=np.array(list(query[c.COLUMN_TRANSFORMED_TITLE])))
            # full float64 jaccard arrays for a few queries (pins fast_jaccard on its own)
            jac_rows = [0, 1, 2, 3, 50, 199]
            jac = np.stack([match_maker.fast_jaccard(
                mm.number_of_truth_titles, float(q_maxint[q]), mm.matrix_non_zero_columns[q],
                mm.matrix_truth_non_zero_columns_and_values, mm.sums_matrix_truth) for q in jac_rows])
            assert jac.dtype == np.float64
            fixture.update(jac_rows=np.array(jac_rows), jac=jac)

        if top_n == 10:
            fixture.update(truth_order_rank=_set_order_ranks(truth[c.COLUMN_N_GRAMS], mm.n_grams_encoding))
        rows = np.zeros((n_query, top_n), dtype=np.int32)
        ids = np.zeros((n_query, top_n), dtype=np.int64)
        margin_ok = np.zeros(n_query, dtype=bool)
        for q in range(n_query):
            jaccard = match_maker.fast_jaccard(
                mm.number_of_truth_titles, float(fixture["q_maxint"][q]), mm.matrix_non_zero_columns[q],
                mm.matrix_truth_non_zero_columns_and_values, mm.sums_matrix_truth)
            top = match_maker.fast_arg_top_k(jaccard, top_n)
            rows[q] = top
            ids[q] = mm.get_closest_matches(q)
            assert (ids[q] == fixture["title_id"][top]).all()
            # margin: no value within 5e-7 of the select threshold, so the float32-vs-float64 subtraction at
            # match_maker.py:70 (NumPy vs numba typing) cannot change the selected set.
            positive = np.sort(jaccard[jaccard > 0])[::-1]
            kth = np.float32(positive[top_n - 1]) if positive.shape[0] >= top_n else np.float32(0)
            threshold = np.float64(kth) - np.float64(np.float32(s.ENCODING_FLOAT_BUFFER))
            margin_ok[q] = np.abs(jaccard - threshold).min() > 5e-7
        fixture[f"rows_k{top_n}"] = rows
        fixture[f"ids_k{top_n}"] = ids
        fixture[f"margin_ok_k{top_n}"] = margin_ok
    np.savez_compressed(f"{HERE}/match_maker_5000x200.npz", **fixture)

    # ------------------------------------------------------------------ C. construct_features at the kernel boundary
    mm10_rows = fixture["rows_k10"]
    pair_q, pair_t = [], []
    for q in range(60):
        for t in mm10_rows[q][:5]:
            pair_q.append(q)
            pair_t.append(int(t))
    for _ in range(100):  # unrelated pairs too
        pair_q.append(int(rng.randint(n_query)))
        pair_t.append(int(rng.randint(n_truth)))
    q_titles = list(query[c.COLUMN_TRANSFORMED_TITLE])
    t_titles = list(truth[c.COLUMN_TRANSFORMED_TITLE])
    fe.words_counter = common.get_words_counter(truth_all)  # counts over the whole 30k truth set
    n_truth_titles = np.uint32(len(truth_all))
    title_len = np.array([len(q_titles[q]) for q in pair_q], dtype=np.uint8)
    truth_len = np.array([len(t_titles[t]) for t in pair_t], dtype=np.uint8)
    title_enc = np.vstack([fe.encode_title(q_titles[q]) for q in pair_q])
    truth_enc = np.vstack([fe.encode_title(t_titles[t]) for t in pair_t])
    counts = np.vstack([fe.get_truth_words_counts(t_titles[t]) for t in pair_t])
    features = np.zeros((len(pair_q), feature_engineering.FEATURES_COUNT), dtype=np.float32)
    dummy = np.zeros((feature_engineering.FEATURES_COUNT,), dtype=np.uint8)
    with np.errstate(all="ignore"):
        feature_engineering.construct_features(title_len, truth_len, title_enc, truth_enc, counts,
                                               np.uint8(encoding[" "]), n_truth_titles, dummy, features)
    np.savez_compressed(
        f"{HERE}/construct_features_400.npz", title_len=title_len, truth_len=truth_len, title_enc=title_enc,
        truth_enc=truth_enc, counts=counts, space_code=np.uint8(encoding[" "]), n_truth=n_truth_titles,
        features=features, titles=np.array([q_titles[q] for q in pair_q]),
        truth_titles=np.array([t_titles[t] for t in pair_t]))

    # ------------------------------------------------------------------ D. MatchMaker on the WHOLE example truth set
    # 30,000 truth rows = two score tiles of the HIP kernel (28,672 rows each): pointer-cache spans, sparse tiles and
    # cross-tile MaxScore are exercised by vectors captured from the reference.  Stored: titles, the vocabulary order,
    # each truth title's set iteration order, sums_matrix_truth, and the answers (rows) for k = 10 and k = 100.
    n_query_full = 1000
    truth_full = truth_all.reset_index(drop=True)
    query_full = test_all.iloc[:n_query_full].reset_index(drop=True)
    full = {}
    for top_n in (10, 100):
        mm = match_maker.MatchMaker(query_full.copy(), truth_full.copy(), top_n)
        if top_n == 10:
            q_maxint = np.array([sum([mm._get_idf_given_index(r) for r in mm.matrix_non_zero_columns[q]])
                                 for q in range(n_query_full)], dtype=np.float64)
            full.update(
                truth_titles=np.array(list(truth_full[c.COLUMN_TRANSFORMED_TITLE])),
                query_titles=np.array(list(query_full[c.COLUMN_TRANSFORMED_TITLE])),
                vocab=np.array(sorted(mm.n_grams_encoding, key=mm.n_grams_encoding.get)),
                title_id=np.asarray(truth_full[c.COLUMN_TITLE_ID], dtype=np.int32),
                truth_order_rank=_set_order_ranks(truth_full[c.COLUMN_N_GRAMS], mm.n_grams_encoding),
                sums32=mm.sums_matrix_truth.astype(np.float32), q_maxint=q_maxint)
        rows, margin_ok = _match_maker_answers(match_maker, s, mm, full["q_maxint"], n_query_full, top_n)
        for q in (0, 1, 500, 999):
            assert mm.get_closest_matches(q) == full["title_id"][rows[q]].tolist()
        full[f"rows_k{top_n}"] = rows
        full[f"margin_ok_k{top_n}"] = margin_ok
    np.savez_compressed(f"{HERE}/match_maker_30000x1000.npz", **full)
    print("30000 x 1000 margin_ok k10/k100:", int(full["margin_ok_k10"].sum()), int(full["margin_ok_k100"].sum()))

    _edge_section(feature_engineering, encoding, alphabet)
    _arg_top_k_section(match_maker)
    shutil.rmtree(data_dir)
    print("levenshtein KATs:", len(lev))
    print("match_maker margin_ok k10/k100:", int(fixture["margin_ok_k10"].sum()), int(fixture["margin_ok_k100"].sum()),
          "of", n_query)
    print("construct_features pairs:", features.shape, "NaNs:", int(np.isnan(features).sum()))
    print("first query top-10 ids:", fixture["ids_k10"][0].tolist())


if __name__ == "__main__":
    main()
