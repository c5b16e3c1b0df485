"""CPU-only checks: host-side logic of the package, the C ABI surface, the synthetic generator."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def library():
    import doppel_speller_amd as ds
    return ctypes.CDLL(ds.build_library())


def test_library_exports_every_declared_symbol(library):
    header = open(os.path.join(ROOT, "include", "doppel_amd.h")).read()
    declared = set(re.findall(r"\b(ds_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 25
    for name in sorted(declared):
        assert hasattr(library, name), name
    from doppel_speller_amd import _lib
    assert declared == set(_lib.EXPORTED_SYMBOLS)


def test_argument_errors_without_gpu(library):
    library.ds_last_error.restype = ctypes.c_char_p
    out = ctypes.c_void_p()
    assert library.ds_index_create(None, None, None, None, ctypes.c_int64(0), ctypes.c_int64(0), 0,
                                   ctypes.byref(out)) == -1
    assert b"null" in library.ds_last_error() or b"positive" in library.ds_last_error()
    rowptr = np.array([0, 2], dtype=np.int64)
    idx = np.array([3, 1], dtype=np.int32)  # not ascending
    one = np.ones(4, dtype=np.float32)
    status = library.ds_index_create(rowptr.ctypes.data_as(ctypes.c_void_p), idx.ctypes.data_as(ctypes.c_void_p),
                                     one.ctypes.data_as(ctypes.c_void_p), one.ctypes.data_as(ctypes.c_void_p),
                                     ctypes.c_int64(1), ctypes.c_int64(4), 0, ctypes.byref(out))
    assert status == -1 and b"ascending" in library.ds_last_error()


def test_product_never_imports_the_oracle():
    """No file of the product (package sources, HIP sources, the C ABI header) imports, links or calls the oracle."""
    forbidden = ("import oracle", "from oracle", "libdoppel_oracle", "ds_oracle_", "oracle/", "oracle.")
    roots = [os.path.join(ROOT, "doppel-speller_amd"), os.path.join(ROOT, "doppel_speller_amd"),
             os.path.join(ROOT, "include")]
    for root in roots:
        for directory, _, files in os.walk(root):
            for name in files:
                if name.endswith((".py", ".hip", ".h", ".cpp")):
                    text = open(os.path.join(directory, name)).read()
                    for needle in forbidden:
                        assert needle not in text, (os.path.join(directory, name), needle)


def test_encoders_known_answers(golden_kat):
    import doppel_speller_amd as ds
    assert ds.ALLOWED_CHARACTERS == golden_kat["alphabet"] and ds.SPACE_CODE == golden_kat["space_code"]
    for title, expected in golden_kat["encode_title"].items():
        got = ds.encode_title(title)
        assert got.dtype == np.uint8 and got.shape == (255,)
        assert got[:len(expected)].tolist() == expected and not got[len(title):].any()
    enc, lengths = ds.encode_titles(list(golden_kat["encode_title"]))
    for row, title in enumerate(golden_kat["encode_title"]):
        assert np.array_equal(enc[row], ds.encode_title(title)) and lengths[row] == len(title)
    counter = {"great": 4, "expectations": 2}
    assert ds.get_truth_words_counts("great expectations", counter).tolist() == [4, 2] + [0] * 13


def test_sequential_sums_are_left_to_right():
    from doppel_speller_amd.match_maker import sequential_sums
    rng = np.random.RandomState(0)
    lengths = rng.randint(0, 40, 200)
    values = (rng.rand(lengths.sum()) * 10).astype(np.float32)
    got = sequential_sums(values, lengths, np.float32)
    at = 0
    for i, n in enumerate(lengths):
        total = 0
        for v in values[at:at + n]:
            total = total + v
        at += n
        assert np.float32(total) == got[i]


def test_match_maker_host_side_matches_captured_structures(golden_match_maker, monkeypatch):
    """MatchMaker's host construction against the structures captured from the reference (no GPU needed)."""
    import pandas as pd
    from doppel_speller_amd import match_maker
    g = golden_match_maker

    class FakeIndex:
        def __init__(self, rowptr, truth_idx, idf32, sums32, device=0):
            self.rowptr, self.truth_idx, self.idf32, self.sums32 = rowptr, truth_idx, idf32, sums32

    monkeypatch.setattr(match_maker, "TruthIndex", FakeIndex)
    n_grams = lambda title: set(title[i:i + 3] for i in range(len(title)) if len(title[i:i + 3]) == 3)
    truth = pd.DataFrame({"title_id": g["title_id"], "n_grams": [n_grams(str(t)) for t in g["truth_titles"]]})
    data = pd.DataFrame({"n_grams": [n_grams(str(t)) for t in g["query_titles"]]})
    mm = match_maker.MatchMaker(data, truth, 10, vocabulary=[str(v) for v in g["vocab"]])
    assert np.array_equal(mm.index.rowptr, g["rowptr"]) and np.array_equal(mm.index.truth_idx, g["truth_idx"])
    assert np.array_equal(mm.index.idf32, g["idf32"])
    assert np.array_equal(mm._q_rowptr, g["q_rowptr"]) and np.array_equal(mm._q_cols, g["q_cols"])
    assert np.array_equal(mm._q_maxint, g["q_maxint"])
    # the float32 rounding of sums_matrix_truth follows the iteration order of each title's n-gram set, which depends
    # on PYTHONHASHSEED: close here, bit-exact in the pinned-seed test below
    assert np.allclose(mm.sums_matrix_truth, g["sums32"], rtol=4e-6)


def test_sums_matrix_truth_bit_exact_under_the_captured_hash_seed():
    """Same check in a child interpreter with PYTHONHASHSEED=0 (the seed the goldens were captured with)."""
    import subprocess
    import sys
    script = """
import numpy as np, pandas as pd, sys
sys.path.insert(0, %r)
from doppel_speller_amd import match_maker
g = np.load(%r)
class FakeIndex:
    def __init__(self, *a, **k): pass
match_maker.TruthIndex = FakeIndex
n_grams = lambda title: set([title[i:i + 3] for i in range(len(title)) if len(title[i:i + 3]) == 3])
truth = pd.DataFrame({"title_id": g["title_id"], "n_grams": [n_grams(str(t)) for t in g["truth_titles"]]})
data = pd.DataFrame({"n_grams": [n_grams(str(t)) for t in g["query_titles"]]})
mm = match_maker.MatchMaker(data, truth, 10, vocabulary=[str(v) for v in g["vocab"]])
assert np.array_equal(mm.sums_matrix_truth.view(np.uint32), g["sums32"].view(np.uint32))
print("exact")
""" % (ROOT, os.path.join(ROOT, "tests", "golden", "match_maker_5000x200.npz"))
    env = dict(os.environ, PYTHONHASHSEED="0")
    result = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=300)
    assert result.returncode == 0 and "exact" in result.stdout, result.stderr[-2000:]


def test_synthetic_workload_acceptance(oracle):
    from doppel_speller_amd import synth
    w = synth.make_workload(30000, 500, seed=synth.DEFAULT_SEED)
    again = synth.make_workload(30000, 500, seed=synth.DEFAULT_SEED)
    assert np.array_equal(w.truth_idx, again.truth_idx) and np.array_equal(w.q_cols, again.q_cols)
    stats = synth.workload_statistics(w)
    assert 19 <= stats["tri_grams_per_truth_title"] <= 23
    assert 0.75 <= stats["postings_touched_per_query_over_n"] <= 1.05
    assert stats["columns"] <= 37 ** 3
    assert (np.diff(w.q_cols.astype(np.int64))[np.diff(np.repeat(np.arange(500), np.diff(w.q_rowptr))) == 0] > 0).all()
    rows = oracle.jaccard_topk(w.rowptr, w.truth_idx, w.idf32, w.sums32, w.q_rowptr, w.q_cols, w.q_maxint, 10)
    derived = w.actual_row >= 0
    recall = np.mean([w.actual_row[q] in rows[q] for q in np.nonzero(derived)[0]])
    assert 0.55 <= derived.mean() <= 0.65 and recall > 0.8


def test_synthetic_workload_acceptance_at_c2_truth_size():
    """SURVEY.md 8d acceptance checks of the generator at the truth size of C2 (N = 500k)."""
    from doppel_speller_amd import synth
    w = synth.make_workload(500000, 2000, seed=synth.DEFAULT_SEED)
    stats = synth.workload_statistics(w, positive_sample=400)
    assert 19 <= stats["tri_grams_per_truth_title"] <= 23            # 21 +- 2
    assert 45000 <= stats["columns"] <= 37 ** 3                       # ~50k, ceiling 50,653
    assert 0.75 <= stats["postings_touched_per_query_over_n"] <= 1.05  # 0.9 N +- 0.15 N
    assert 0.28 <= stats["positive_score_fraction"] <= 0.38           # ~0.33 N
    assert 3.3 <= stats["words_per_truth_title"] <= 3.7 and 22 <= stats["chars_per_truth_title"] <= 26


def test_library_is_tied_to_its_sources(monkeypatch):
    """ds_build_id() is the hash of csrc/ + include/ the binary was compiled from; the loader refuses a mismatch."""
    from doppel_speller_amd import _lib
    assert _lib.lib().ds_build_id().decode() == _lib.source_id() == _lib.binary_id(_lib.library_path())
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "source_id", lambda: "0123456789abcdef")
    monkeypatch.setenv("DS_AUTO_REBUILD", "0")
    with pytest.raises(_lib.DoppelError, match="stale library is refused"):
        _lib.lib()


def test_pair_list_restatement_against_a_pandas_transcription(oracle):
    """oracle.remaining_pairs / select_matches (NumPy) against the reference's own pandas formulation of
    predict.py:172-183 and :246-252 (groupby / transform(max) / duplicated), on random inputs with ties."""
    import pandas as pd
    rng = np.random.RandomState(4)
    n, k = 300, 6
    rows = rng.randint(0, 1000, (n, k)).astype(np.int32)
    best = np.where(rng.rand(n) < 0.4, rng.randint(0, 1000, n), -1).astype(np.int32)
    frame = pd.DataFrame({"test_index": np.repeat(np.arange(n), k), "match": rows.reshape(-1)})
    matched_so_far = set(np.nonzero(best >= 0)[0].tolist())
    remaining = frame.loc[~(frame["test_index"].isin(matched_so_far)), :]          # predict.py:183
    pair_q, pair_t = oracle.remaining_pairs(best, rows)
    assert np.array_equal(pair_q, remaining["test_index"].to_numpy()) and np.array_equal(pair_t, remaining["match"].to_numpy())
    predictions = rng.choice(np.array([0.1, 0.5, 0.9, 0.93, 0.97], dtype=np.float32), pair_q.shape[0])
    remaining = remaining.assign(prediction=predictions)
    at_max = remaining.groupby(["test_index"])["prediction"].transform(max) == remaining["prediction"]   # :246-247
    matches = remaining.loc[at_max, :]
    matches = matches.loc[matches["prediction"] > np.float32(0.9), :]                                      # :249-250
    duplicated = matches.loc[matches.duplicated(["test_index"]), "test_index"]                             # :158-161
    matches = matches.loc[~(matches["test_index"].isin(duplicated)), :]
    query, match = oracle.select_matches(pair_q, pair_t, predictions, k, 0.9)
    expected = dict(zip(matches["test_index"], matches["match"]))
    assert {int(q): int(m) for q, m in zip(query, match) if m >= 0} == expected
    assert len(expected) > 10 and (match < 0).sum() > 10


def test_bench_configurations_follow_baseline_json():
    """bench.py --config C2..C5: BASELINE.json's shapes, per GPU, for the GPU counts the driver uses."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from doppel_speller_amd.distributed import shard_sizes
    # resolve_config returns the queries of the WHOLE job; rank r owns shard_range(total, r, world)
    assert bench.resolve_config("C2", 1)[:4] == (100_000, 500_000, 10, "weak")
    assert bench.resolve_config("C2", 8)[:4] == (800_000, 500_000, 10, "weak")       # fixed per-GPU batch
    assert shard_sizes(800_000, 8) == [100_000] * 8
    assert bench.resolve_config("C3", 1)[:4] == (1_000_000, 5_000_000, 50, "weak")
    assert bench.resolve_config("C4", 8)[:4] == (8_000_000, 5_000_000, 50, "strong")  # 8M queries in all
    assert bench.resolve_config("C4", 2)[:4] == (8_000_000, 5_000_000, 50, "strong") and shard_sizes(8_000_000, 8) == [1_000_000] * 8
    assert bench.resolve_config("C5", 8)[:4] == (1_000_000, 50_000_000, 100, "strong")  # 1M queries in all
    assert shard_sizes(1_000_000, 8) == [125_000] * 8
    assert bench.resolve_config("C5", 1)[:4] == (1_000_000, 50_000_000, 100, "strong")
    total, truth, k, scaling, _, custom = bench.resolve_config("C2", 4, queries=2000, truth=60000)
    assert (total, truth, k, scaling, custom) == (8000, 60000, 10, "weak", True)     # --queries = per-GPU batch (weak)
    total, truth, k, scaling, _, custom = bench.resolve_config("C5", 2, queries=4001, truth=60000)
    assert (total, scaling) == (4001, "strong") and shard_sizes(total, 2) == [2000, 2001]   # --queries = job total (strong)
    import json
    with open(os.path.join(ROOT, "BASELINE.json")) as handle:
        configs = json.load(handle)["configs"]
    assert "100k synthetic queries" in configs[1] and "1M queries" in configs[2] and "8M queries" in configs[3]


def test_reorder_queries_keeps_every_query():
    """synth.reorder_queries (tuning experiments on the order of a launch's queries) permutes every per-query array alike."""
    import doppel_speller_amd as ds  # noqa: F401
    from doppel_speller_amd import synth
    w = synth.make_workload(4000, 300)
    order = np.argsort(w.q_maxint, kind="stable")[::-1]
    r = synth.reorder_queries(w, order)
    assert r.q_rowptr[-1] == w.q_rowptr[-1] and np.array_equal(np.sort(r.q_maxint), np.sort(w.q_maxint))
    for i in (0, 17, 299):
        q = order[i]
        assert np.array_equal(r.q_cols[r.q_rowptr[i]:r.q_rowptr[i + 1]], w.q_cols[w.q_rowptr[q]:w.q_rowptr[q + 1]])
        assert r.q_maxint[i] == w.q_maxint[q] and np.array_equal(r.q_enc[i], w.q_enc[q]) and r.q_len[i] == w.q_len[q]
        assert r.actual_row[i] == w.actual_row[q]


def test_private_directory_refuses_a_directory_others_could_write_to(tmp_path):
    """an existing ds_<uid> with group / world access is refused, not silently tightened (what somebody else
    planted there before would stay)."""
    import os
    from doppel_speller_amd import _lib
    from doppel_speller_amd.distributed import private_directory
    base = str(tmp_path)
    path = private_directory(base)
    assert os.stat(path).st_mode & 0o777 == 0o700 and private_directory(base) == path
    os.chmod(path, 0o770)
    with pytest.raises(_lib.DoppelError):
        private_directory(base)
