"""`libdoppel_cpu.so` (oracle/doppel_cpu_abi.c): the CPU oracle behind the SAME C ABI as the product ("identical
signatures in libdoppel_cpu for parity").  It is compiled against include/doppel_amd.h, so a drifting signature is a compile error;
here the ctypes stub a maintainer of the reference would add (examples/reference_binding.py = INTEGRATION.md section 2) runs
UNCHANGED against it -- on the CPU, against the vectors captured from the reference -- and, on the GPU box, the same binding code
runs against both libraries on the same inputs."""
import ctypes
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CORE_ENTRY_POINTS = ("ds_last_error", "ds_version", "ds_build_id", "ds_device_count", "ds_index_create", "ds_index_destroy",
                     "ds_index_info", "ds_jaccard_topk", "ds_construct_features", "ds_levenshtein_ratio")


def _stub(library_path):
    """examples/reference_binding.py bound to `library_path` (a fresh module per library)."""
    os.environ["DOPPEL_AMD_LIBRARY"] = library_path
    spec = importlib.util.spec_from_file_location("reference_binding_" + os.path.basename(library_path).replace(".", "_"),
                                                  os.path.join(ROOT, "examples", "reference_binding.py"))
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    return module


def _cpu_library(oracle):
    oracle.build()
    return oracle.CPU_ABI_PATH


def test_cpu_library_exports_the_core_entry_points_of_the_header(oracle):
    handle = ctypes.CDLL(_cpu_library(oracle))
    for name in CORE_ENTRY_POINTS:
        assert hasattr(handle, name), name
    with open(os.path.join(ROOT, "include", "doppel_amd.h")) as header:
        declared = header.read()
    for name in CORE_ENTRY_POINTS:
        assert name + "(" in declared, name          # every one of them is the header's, nothing of its own
    handle.ds_build_id.restype = ctypes.c_char_p
    assert handle.ds_build_id() == b"cpu-oracle" and handle.ds_version() == 200


def test_reference_stub_runs_unchanged_against_the_cpu_library(oracle, golden_match_maker, golden_features):
    from test_gpu_integration_stub import _ReferenceMatchMaker
    stub = _stub(_cpu_library(oracle))
    g = golden_match_maker
    index = stub.AmdIndex(_ReferenceMatchMaker(g))
    rows = index.top_rows(_ReferenceMatchMaker(g), list(range(200)), 10)
    assert np.array_equal(rows, g["rows_k10"])               # what the reference itself answered
    with pytest.raises(Exception, match="top_matches.shape"):     # same exception text as match_maker.py:189
        index.top_rows(_ReferenceMatchMaker(g), [0], 5001)
    f = golden_features
    response = np.zeros((f["title_len"].shape[0], 66), dtype=np.float32)
    stub.construct_features(f["title_len"], f["truth_len"], f["title_enc"], f["truth_enc"], f["counts"], f["space_code"],
                            f["n_truth"], np.zeros(66, np.uint8), response)
    assert np.array_equal(response[:, :51].view(np.uint32), f["features"][:, :51].view(np.uint32))


def test_single_pair_levenshtein_of_the_cpu_library(oracle, golden_kat):
    handle = ctypes.CDLL(_cpu_library(oracle))
    table = {ch: i for i, ch in enumerate(golden_kat["alphabet"])}
    for case in golden_kat["levenshtein"]:
        a = np.array([table[ch] for ch in case["a"]], dtype=np.uint8)
        b = np.array([table[ch] for ch in case["b"]], dtype=np.uint8)
        pointer = lambda x: x.ctypes.data_as(ctypes.c_void_p)
        assert handle.ds_levenshtein_ratio(pointer(a), len(a), pointer(b), len(b)) == case["ratio"]
    assert handle.ds_levenshtein_ratio(None, 3, None, 3) == -1       # DS_E_ARG, as the product


@pytest.mark.gpu
def test_same_binding_same_inputs_both_libraries(oracle):
    """The stub bound to the HIP library and to the CPU library: identical rows and identical feature bits."""
    from doppel_speller_amd import _lib, synth
    from test_gpu_integration_stub import _ReferenceMatchMaker
    w = synth.make_workload(40000, 300, seed=12)
    g = {"rowptr": np.asarray(w.rowptr), "truth_idx": np.asarray(w.truth_idx), "idf32": np.asarray(w.idf32),
         "sums32": np.asarray(w.sums32), "q_rowptr": np.asarray(w.q_rowptr), "q_cols": np.asarray(w.q_cols)}
    answers = {}
    for name, path in (("hip", _lib.library_path()), ("cpu", _cpu_library(oracle))):
        stub = _stub(path)
        match_maker = _ReferenceMatchMaker(g)
        index = stub.AmdIndex(match_maker)
        rows = index.top_rows(match_maker, list(range(300)), 10)
        pair_q, pair_t = np.repeat(np.arange(300), 10), rows.reshape(-1)
        response = np.zeros((3000, 66), dtype=np.float32)
        stub.construct_features(w.q_len[pair_q], w.t_len[pair_t], w.q_enc[pair_q], w.t_enc[pair_t], w.t_counts[pair_t], 1,
                                w.n_truth, np.zeros(66, np.uint8), response)
        answers[name] = (rows, response)
    assert np.array_equal(answers["hip"][0], answers["cpu"][0])
    assert np.array_equal(answers["hip"][1].view(np.uint32), answers["cpu"][1].view(np.uint32))
