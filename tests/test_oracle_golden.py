"""The CPU oracle (oracle/doppel_oracle.c) against the golden vectors captured from the reference's own function
bodies (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np


def _encode(kat, text):
    table = {ch: i for i, ch in enumerate(kat["alphabet"])}
    return np.array([table[ch] for ch in text], dtype=np.uint8)


def test_levenshtein_known_answers(oracle, golden_kat):
    for typing in ("numba", "numpy"):
        for case in golden_kat["levenshtein"]:
            a, b = _encode(golden_kat, case["a"]), _encode(golden_kat, case["b"])
            assert oracle.levenshtein_ratio(a, b, typing) == case["ratio"], (case, typing)
            assert oracle.levenshtein_ratio(b, a, typing) == case["ratio"], (case, typing)


def test_levenshtein_survey_values(oracle, golden_kat):
    # SURVEY.md section 8c known answers (captured by import in the survey session)
    expected = {("coolblue bv", "coolblu bv"): 95, ("abc", "abc"): 100, ("abc", "xyz"): 0, ("a", "ab"): 66,
                ("kitten", "sitting"): 61, ("limited", "ltd"): 60,
                ("systematica imnvestments services limited", "systematica investment services limited"): 97,
                ("feld s ullivan limited", "feld sullivan limited"): 97,
                ("a" * 29 + "b" * 21, "a" * 29 + "c" * 21): 57}  # hazard point (58,100): strict IEEE -> 57
    for (a, b), ratio in expected.items():
        assert oracle.levenshtein_ratio(_encode(golden_kat, a), _encode(golden_kat, b)) == ratio


def test_fast_jaccard_bit_exact(oracle, golden_match_maker):
    g = golden_match_maker
    for row, expected in zip(g["jac_rows"], g["jac"]):
        columns = g["q_cols"][g["q_rowptr"][row]:g["q_rowptr"][row + 1]]
        got = oracle.fast_jaccard(g["q_maxint"][row], columns, g["rowptr"], g["truth_idx"], g["idf32"], g["sums32"])
        assert got.dtype == np.float64
        assert np.array_equal(got.view(np.uint64), expected.view(np.uint64))


def test_fast_arg_top_k_on_captured_jaccard(oracle, golden_match_maker):
    g = golden_match_maker
    for row, jaccard in zip(g["jac_rows"], g["jac"]):
        for k in (10, 100):
            got = oracle.fast_arg_top_k(jaccard, k, "numpy")
            assert np.array_equal(got, g[f"rows_k{k}"][row])
            if g[f"margin_ok_k{k}"][row]:
                assert np.array_equal(oracle.fast_arg_top_k(jaccard, k, "numba"), g[f"rows_k{k}"][row])


def test_jaccard_topk_batch(oracle, golden_match_maker):
    g = golden_match_maker
    for k in (10, 100):
        numpy_typed = oracle.jaccard_topk(g["rowptr"], g["truth_idx"], g["idf32"], g["sums32"], g["q_rowptr"],
                                          g["q_cols"], g["q_maxint"], k, "numpy")
        assert np.array_equal(numpy_typed, g[f"rows_k{k}"])  # every vector, captured typing
        numba_typed = oracle.jaccard_topk(g["rowptr"], g["truth_idx"], g["idf32"], g["sums32"], g["q_rowptr"],
                                          g["q_cols"], g["q_maxint"], k, "numba")
        ok = g[f"margin_ok_k{k}"]
        assert ok.sum() >= 0.95 * ok.shape[0]
        assert np.array_equal(numba_typed[ok], g[f"rows_k{k}"][ok])  # specification typing, margin-checked vectors
        assert np.array_equal(g["title_id"][numba_typed[ok]], g[f"ids_k{k}"][ok])


def test_jaccard_topk_raises_when_fewer_than_k(oracle, golden_match_maker):
    g = golden_match_maker
    import pytest
    with pytest.raises(Exception, match="top_matches.shape"):
        oracle.jaccard_topk(g["rowptr"], g["truth_idx"], g["idf32"], g["sums32"], g["q_rowptr"][:2], g["q_cols"],
                            g["q_maxint"][:1], g["sums32"].shape[0] + 1)


def test_construct_features_bit_exact(oracle, golden_features):
    g = golden_features
    expected = g["features"]
    got = oracle.construct_features(g["title_len"], g["truth_len"], g["title_enc"], g["truth_enc"], g["counts"],
                                    g["space_code"], g["n_truth"], "numpy")
    assert np.array_equal(got.view(np.uint32), expected.view(np.uint32))  # NaN pattern included
    spec = oracle.construct_features(g["title_len"], g["truth_len"], g["title_enc"], g["truth_enc"], g["counts"],
                                     g["space_code"], g["n_truth"], "numba")
    # numba typing differs only in the 15 rank features (float64 vs float32 arithmetic), by at most one ulp
    assert np.array_equal(spec[:, :51].view(np.uint32), expected[:, :51].view(np.uint32))
    ranks_spec, ranks_np = spec[:, 51:], expected[:, 51:]
    assert np.array_equal(np.isnan(ranks_spec), np.isnan(ranks_np))
    different = (ranks_spec.view(np.uint32) != ranks_np.view(np.uint32)) & ~np.isnan(ranks_np)
    assert different.mean() < 0.05
    assert np.all(np.abs(ranks_spec.view(np.int32).astype(np.int64) - ranks_np.view(np.int32))[~np.isnan(ranks_np)] <= 1)


def test_fast_arg_top_k_on_crafted_arrays(oracle):
    """The reference's own fast_arg_top_k (make_golden.py, section F) on arrays the example data does not produce: ties
    at the k-th value, fewer than k positive values (the answer is then shorter than k or falls back to the largest
    indexes), all zeros, negatives, values within float32 resolution of the k-th, k = 1 / len / > len."""
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "arg_top_k_cases.npz"), allow_pickle=False))
    assert g["k"].shape[0] >= 40
    for case, k in enumerate(g["k"]):
        array = g["arrays"][g["array_offsets"][case]:g["array_offsets"][case + 1]]
        expected = g["answers"][g["answer_offsets"][case]:g["answer_offsets"][case + 1]]
        for typing in ("numpy", "numba"):
            got = oracle.fast_arg_top_k(array, int(k), typing)
            assert np.array_equal(got, expected), (case, int(k), typing, got[:8], expected[:8])


def test_construct_features_edge_pairs_bit_exact(oracle, golden_features_edge):
    """Pairs the example data does not contain, answers captured from the reference's own construct_features
    (tests/golden/make_golden.py, section E): leading / trailing / repeated spaces, more than 15 words, one-character
    and 254-character titles, zero / huge word counts, n_truth = 1 and 4e9."""
    g = golden_features_edge
    assert g["features"].shape[0] == 3 and g["features"].shape[1] >= 120
    for block, n_truth in enumerate(g["n_truth"]):
        expected = g["features"][block]
        with np.errstate(all="ignore"):
            got = oracle.construct_features(g["title_len"], g["truth_len"], g["title_enc"], g["truth_enc"], g["counts"],
                                            g["space_code"], n_truth, "numpy")
            spec = oracle.construct_features(g["title_len"], g["truth_len"], g["title_enc"], g["truth_enc"], g["counts"],
                                             g["space_code"], n_truth, "numba")
        assert np.array_equal(got.view(np.uint32), expected.view(np.uint32))  # NaN / inf patterns included
        assert np.array_equal(spec[:, :51].view(np.uint32), expected[:, :51].view(np.uint32))
        ranks_spec, ranks_np = spec[:, 51:], expected[:, 51:]
        assert np.array_equal(np.isnan(ranks_spec), np.isnan(ranks_np))
        finite = ~np.isnan(ranks_np)
        assert np.all(np.abs(ranks_spec.view(np.int32).astype(np.int64) - ranks_np.view(np.int32))[finite] <= 1)


def test_construct_features_survey_example(oracle, golden_kat):
    # q='systematica imnvestments services limited', t='maxima technologies nl bv'
    q, t = "systematica imnvestments services limited", "maxima technologies nl bv"
    enc = lambda text: np.pad(_encode(golden_kat, text), (0, 255 - len(text)))[None, :]
    counts = np.zeros((1, 15), dtype=np.uint32)
    counts[0, :4] = [1, 306, 51, 4923]
    with np.errstate(all="ignore"):
        out = oracle.construct_features([len(q)], [len(t)], enc(q), enc(t), counts, 1, 30000)[0]
    assert out[:6].tolist() == [41, 25, 4, 4, 39, 56]
    assert out[6:10].tolist() == [66, 41, 50, 50] and np.isnan(out[10:21]).all()
    assert out[21:25].tolist() == [6, 12, 2, 2]
    assert np.allclose(out[36:40], [10.309, 4.5854, 6.3771, 1.8073], atol=1e-4)
    assert np.allclose(out[51:55], [1, 2.4309, 1.983, 3.1254], atol=1e-4)


def test_whole_example_truth_set(oracle, golden_match_maker_full):
    """30,000 truth rows x 1,000 queries captured from the reference MatchMaker (k = 10 and 100): the host-side index
    build reproduces sums_matrix_truth / max_intersection_possible bit for bit and the oracle reproduces the answers."""
    g = golden_match_maker_full
    assert g["sums32"].shape[0] == 30000
    fixture = np.load(os.path.join(os.path.dirname(__file__), "golden", "match_maker_30000x1000.npz"))
    assert np.array_equal(g["sums32"].view(np.uint32), fixture["sums32"].view(np.uint32))
    assert np.array_equal(g["q_maxint"].view(np.uint64), fixture["q_maxint"].view(np.uint64))
    for k in (10, 100):
        numpy_typed = oracle.jaccard_topk(g["rowptr"], g["truth_idx"], g["idf32"], g["sums32"], g["q_rowptr"],
                                          g["q_cols"], g["q_maxint"], k, "numpy")
        assert np.array_equal(numpy_typed, g[f"rows_k{k}"])
        numba_typed = oracle.jaccard_topk(g["rowptr"], g["truth_idx"], g["idf32"], g["sums32"], g["q_rowptr"],
                                          g["q_cols"], g["q_maxint"], k, "numba")
        ok = g[f"margin_ok_k{k}"]
        assert ok.sum() >= 0.99 * ok.shape[0]
        assert np.array_equal(numba_typed[ok], g[f"rows_k{k}"][ok])
