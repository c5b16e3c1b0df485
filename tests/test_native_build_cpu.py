"""Next row f-3: the native index build (`ds_problem_create`, host code in libdoppel_amd.so) against the Python
restatement of `MatchMaker.__init__` (which itself is pinned by the golden vectors) -- no GPU needed."""
import os

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


# ---- round 3: the host side is threaded (DS_HOST_THREADS); nothing may depend on the thread count ----------------------
def _digest(w, tile_rows):
    from doppel_speller_amd import _lib
    digest = np.zeros(8, dtype=np.uint64)
    _lib.check(_lib.lib().ds_index_image_digest(_lib.pointer(w.rowptr), _lib.pointer(w.truth_idx), _lib.pointer(w.idf32),
                                                _lib.pointer(w.sums32), w.n_columns, w.n_truth, tile_rows,
                                                _lib.pointer(digest)), "ds_index_image_digest")
    return digest


@pytest.mark.parametrize("tile_rows", [12288, 28672, 64])
def test_index_image_does_not_depend_on_the_thread_count(monkeypatch, tile_rows):
    """The arrays ds_index_create uploads (list pointers, parity-grouped postings, per-posting info, row records with
    signatures and duplicate ranks, tile minima) digested for 1, 3 and 8 build threads."""
    from doppel_speller_amd import synth
    w = synth.make_workload(40_000, 10, seed=11)
    digests = []
    for threads in (1, 3, 8):
        monkeypatch.setenv("DS_HOST_THREADS", str(threads))
        digests.append(_digest(w, tile_rows))
    assert np.array_equal(digests[0], digests[1]) and np.array_equal(digests[0], digests[2])
    assert digests[0][6] > 0 and digests[0][7] == 0


def test_index_image_rejects_bad_posting_lists(monkeypatch):
    import doppel_speller_amd as ds
    from doppel_speller_amd import synth
    w = synth.make_workload(2_000, 10, seed=11)
    for threads in (1, 4):
        monkeypatch.setenv("DS_HOST_THREADS", str(threads))
        for value in (-1, w.n_truth, None):
            broken = w.truth_idx.copy()
            column = int(np.argmax(np.diff(w.rowptr)))
            at = int(w.rowptr[column]) + 3
            broken[at] = broken[at - 1] if value is None else value     # not strictly ascending / outside [0, N)
            saved, w.truth_idx = w.truth_idx, broken
            with pytest.raises(ds.DoppelError, match="not strictly ascending"):
                _digest(w, 64)
            w.truth_idx = saved


def test_problem_create_does_not_depend_on_the_thread_count(monkeypatch):
    from doppel_speller_amd import synth
    from doppel_speller_amd.match_maker import NativeProblem
    w = synth.make_workload(20_000, 3_000, seed=12)
    results = []
    for threads in (1, 5):
        monkeypatch.setenv("DS_HOST_THREADS", str(threads))
        results.append(NativeProblem.from_flat(w.t_flat, w.t_off, w.q_flat, w.q_off, 3).arrays())
    for name, array in results[0].items():
        assert np.array_equal(array.view(np.uint8), results[1][name].view(np.uint8)), name
    # and a binary alphabet (all 256 byte values: the dense key space is 2^24 tri-grams) still builds
    rng = np.random.RandomState(1)
    chars = rng.randint(0, 256, 5000).astype(np.uint8)
    offsets = np.arange(0, 5001, 50, dtype=np.int64)
    problem = NativeProblem.from_flat(chars, offsets, chars[:500], offsets[:11], 3)
    keys = problem.arrays()["vocabulary_keys"]
    assert (np.diff(keys.astype(np.int64)) > 0).all() and problem.n_columns > 4000


def test_generator_does_not_depend_on_the_thread_count(monkeypatch):
    from doppel_speller_amd import synth
    runs = []
    for threads in (1, 6):
        monkeypatch.setenv("DS_HOST_THREADS", str(threads))
        w = synth.make_workload(5_000, 2_000, seed=13)
        runs.append((w.t_flat.copy(), w.t_off.copy(), w.q_flat.copy(), w.q_off.copy(), w.t_counts.copy(), w.t_enc.copy()))
    for a, b in zip(*runs):
        assert np.array_equal(a, b)
    lengths = np.diff(runs[0][3])
    assert lengths.min() >= 3 and lengths.max() <= 255          # common.py:28-38: at most 255, '0'-padded to a tri-gram


def test_encoders_for_whole_collections_against_the_per_title_functions(monkeypatch):
    """ds_encode_titles / ds_truth_word_counts (row a7, native + threaded) against encode_title /
    get_truth_words_counts over collections.Counter (feature_engineering.py:298-319, common.py:140-142)."""
    from collections import Counter
    from doppel_speller_amd import synth
    from doppel_speller_amd.feature_engineering import ALLOWED_CHARACTERS, encode_title, get_truth_words_counts
    w = synth.make_workload(6_000, 10, seed=14)
    titles = synth._to_strings(w.t_flat, w.t_off)
    titles[5] = "bv bv limited bv"                              # a word repeated inside a title counts once
    titles[6] = " ".join(f"w{i}" for i in range(20))            # more than 15 words
    titles[7] = "   spaced   out  "                             # runs of separators: str.split() semantics
    titles[8] = ""
    counter = Counter(word for title in titles for word in set(title.split()))   # common.py:140-142
    code_of = np.zeros(256, dtype=np.uint8)
    for code, character in enumerate(ALLOWED_CHARACTERS):
        code_of[ord(character)] = code
    raw = np.frombuffer("".join(titles).encode("ascii"), dtype=np.uint8)
    offsets = np.concatenate(([0], np.cumsum([len(t) for t in titles]))).astype(np.int64)
    for threads in (1, 7):
        monkeypatch.setenv("DS_HOST_THREADS", str(threads))
        enc, lengths = synth.encode_collection(raw, offsets, code_of)
        counts = synth.truth_word_counts(raw, offsets, separators=[ord(c) for c in " \t\n\r\x0b\x0c\x1c\x1d\x1e\x1f"])
        for row in list(range(12)) + list(range(12, len(titles), 97)):
            assert np.array_equal(enc[row], encode_title(titles[row])), row
            assert lengths[row] == len(titles[row])
            assert np.array_equal(counts[row], get_truth_words_counts(titles[row], counter)), (row, titles[row])
    assert counts[5, 0] == counter["bv"] and counts[5, 1] == counter["bv"] and counts[5, 4] == 0


def test_fast_kernel_contains_no_function_call(tmp_path):
    """Round 3's diagnostics-build fault (profiles/r03_failure_causes.md (c)) went away when the one device function the
    compiler had not inlined was: the Jaccard kernels keep ~120 spilled scalar registers in VGPR lanes and must not call
    out.  The generated ISA of the narrow geometry holds no s_swappc_b64."""
    import shutil
    import subprocess
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "narrow.s"
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", '-DDS_BUILD_ID="t"',
                    "-I", os.path.join(root, "include"), "-S", "--cuda-device-only",
                    os.path.join(root, "doppel-speller_amd", "csrc", "ds_jaccard_narrow.hip"), "-o", str(out)],
                   check=True, capture_output=True, timeout=600)
    text = out.read_text()
    assert "ds_jaccard_topk_kernel" in text
    assert "s_swappc_b64" not in text and "s_setpc_b64" not in text
