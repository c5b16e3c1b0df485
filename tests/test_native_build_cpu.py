"""Next row f-3: the native index build (`ds_problem_create`, host code in libdoppel_amd.so) against the Python
restatement of `MatchMaker.__init__` (which itself is pinned by the golden vectors) -- no GPU needed."""
import numpy as np
import pandas as pd
import pytest


def _n_grams_in_order(title, n=3):
    """get_n_grams (common.py:150-151) as a LIST in first-occurrence order: the order the native build sums in."""
    seen, out = set(), []
    for i in range(len(title) - n + 1):
        gram = title[i:i + n]
        if gram not in seen:
            seen.add(gram)
            out.append(gram)
    return out


def _frames(titles, truth_titles):
    data = pd.DataFrame({"n_grams": [_n_grams_in_order(t) for t in titles], "title_id": np.arange(len(titles))})
    truth = pd.DataFrame({"n_grams": [_n_grams_in_order(t) for t in truth_titles],
                          "title_id": np.arange(len(truth_titles))[::-1]})
    return data, truth


def _compare(titles, truth_titles):
    import doppel_speller_amd as ds
    native = ds.NativeProblem(truth_titles, titles)
    got = native.arrays()
    vocabulary = native.vocabulary()
    assert vocabulary == sorted(vocabulary)                       # ascending byte strings
    data, truth = _frames(titles, truth_titles)
    expected = ds.MatchMaker.host_arrays(data, truth, vocabulary=vocabulary)
    assert expected["vocabulary"] == vocabulary
    for name in ("idf64", "idf32", "sums32", "q_maxint"):
        assert np.array_equal(got[name].view(np.uint8), np.ascontiguousarray(expected[name]).view(np.uint8)), name
    for name in ("rowptr", "truth_idx", "q_rowptr", "q_cols"):
        assert np.array_equal(got[name], expected[name]), name
    return native, got


def test_small_hand_made_collection():
    truth = ["coolblue bv", "coolblue nv", "blue cool bv", "ab", "", "aaaa", "limited limited"]
    queries = ["coolblu bv", "xyz", "ab", "aaaaaa", "limited"]
    native, got = _compare(queries, truth)
    assert native.n_truth == 7 and native.n_queries == 5
    assert got["sums32"][3] == 0 and got["sums32"][4] == 0       # titles shorter than one tri-gram
    assert got["q_rowptr"][2] - got["q_rowptr"][1] == 1           # 'xyz': one n-gram, unseen in truth -> max idf
    assert got["q_rowptr"][3] - got["q_rowptr"][2] == 0           # 'ab': no tri-gram


def test_column_present_in_every_truth_title_is_dropped():
    truth = ["abc one", "abc two", "abc three"]
    native, got = _compare(["abc", "abc one"], truth)
    vocabulary = native.vocabulary()
    column = vocabulary.index("abc")
    assert got["idf32"][column] == 0 and got["rowptr"][column + 1] == got["rowptr"][column]   # explicit zero vanishes
    assert got["q_rowptr"][1] == 0                                                              # query 'abc' has no column


def test_synthetic_collection_against_python_build():
    from doppel_speller_amd import synth
    w = synth.make_workload(3000, 400, seed=5)
    _compare(synth._to_strings(w.q_flat, w.q_off), synth._to_strings(w.t_flat, w.t_off))


def test_bad_arguments():
    import doppel_speller_amd as ds
    with pytest.raises(ds.DoppelError):
        ds.NativeProblem(["abc"], ["abc"], n_gram=4)
    with pytest.raises(ds.DoppelError):
        ds.NativeProblem([], ["abc"])


def test_duplicate_ranks_against_a_dictionary():
    """ds_index_duplicate_ranks (host part of ds_index_create): rank = twins (same column set, same sums32 bits) with
    a larger row index.  Checked against a Python dictionary on a workload with many duplicated titles."""
    import ctypes
    from doppel_speller_amd import _lib, synth
    w = synth.make_workload(40000, 10, seed=13)
    sums = w.sums32.copy()
    sums[::7] = np.nextafter(sums[::7], np.float32(np.inf))   # same columns, other sums: another class
    ranks = np.full(w.n_truth, 9999, dtype=np.uint16)
    _lib.check(_lib.lib().ds_index_duplicate_ranks(_lib.pointer(w.rowptr), _lib.pointer(w.truth_idx), _lib.pointer(sums),
                                                   w.n_columns, w.n_truth, _lib.pointer(ranks)), "duplicate ranks")
    columns = np.repeat(np.arange(w.n_columns), np.diff(w.rowptr))
    order = np.argsort(w.truth_idx, kind="stable")
    per_row = np.split(columns[order], np.cumsum(np.bincount(w.truth_idx, minlength=w.n_truth))[:-1])
    seen, expected = {}, np.zeros(w.n_truth, dtype=np.uint16)
    for t in range(w.n_truth - 1, -1, -1):
        key = (per_row[t].tobytes(), sums[t].tobytes())
        expected[t] = seen.get(key, 0)
        seen[key] = expected[t] + 1
    assert np.array_equal(ranks, expected)
    assert expected.max() >= 10   # the workload does contain long runs of twins
