"""Hand-overs at the C5 shape (50M truth titles, top-100): who are they?"""
import sys, numpy as np, time
sys.path.insert(0, ".")
import doppel_speller_amd as ds
from doppel_speller_amd import synth
k, n_truth, n_queries = 100, 50_000_000, 20_000
w = synth.make_workload(n_truth, n_queries, seed=20260101)
index = ds.TruthIndex(w.rowptr, w.truth_idx, w.idf32, w.sums32)
d = [ds._lib.DeviceArray.from_host(x) for x in (w.q_rowptr, w.q_cols, w.q_maxint)]
rows = ds._lib.DeviceArray((n_queries, k), np.int32)
index.top_k_device(d[0].ptr, d[1].ptr, d[2].ptr, n_queries, k, rows.ptr)
stats = index.sync()
status = index.status(n_queries)
print(stats["dense_reasons"], "redos", stats["sparse_redos"], "handed over:", int((status == 1).sum()), "kernel ms", stats["topk_kernel_ms"], stats["dense_kernel_ms"])
handed = np.nonzero(status == 1)[0]
for q in handed[:8]:
    cols = w.q_cols[w.q_rowptr[q]:w.q_rowptr[q + 1]]
    M = w.q_maxint[q]
    scores = np.zeros(n_truth, dtype=np.float32)
    for c in cols:
        scores[w.truth_idx[w.rowptr[c]:w.rowptr[c + 1]]] += w.idf32[c]
    jac = scores / (w.sums32 + (np.float32(M) - scores))
    top = np.sort(np.partition(jac, -5000)[-5000:])[::-1]
    kth = top[k - 1]
    n = len(cols)
    margin = (6 * n + 64) * 5.96e-8 + n * 0.3 * (4.0 / 65000)
    cut = kth - 2 * margin - 2e-6
    title = synth._to_strings(w.q_flat[w.q_off[q]:w.q_off[q + 1]], np.array([0, w.q_off[q + 1] - w.q_off[q]]))[0]
    print(f"q={q} '{title}' n={n} M={M:.2f} kth={kth:.5f} margin={margin:.2e} rows>=cut: {(top >= cut).sum()} rows>=kth-1e-6: {(top >= kth-1e-6).sum()} rows>=0.98kth {(top >= 0.98*kth).sum()} rows>=0.95kth {(top>=0.95*kth).sum()} top1 {top[0]:.4f} distinct in top 2000: {len(np.unique(top[:2000]))}")

# ---- are the duplicate ranks of the tied rows what they should be (caller's order vs the index's sums32 order)?
from doppel_speller_amd import _lib
q = handed[0]
cols = w.q_cols[w.q_rowptr[q]:w.q_rowptr[q + 1]]
M = w.q_maxint[q]
scores = np.zeros(n_truth, dtype=np.float32)
for c in cols:
    scores[w.truth_idx[w.rowptr[c]:w.rowptr[c + 1]]] += w.idf32[c]
jac = scores / (w.sums32 + (np.float32(M) - scores))
kth = np.partition(jac, -k)[-k]
tied = np.nonzero(jac == kth)[0]
ranks = np.zeros(n_truth, dtype=np.uint16)
t0 = time.time()
_lib.check(_lib.lib().ds_index_duplicate_ranks(_lib.pointer(w.rowptr), _lib.pointer(w.truth_idx), _lib.pointer(w.sums32), w.n_columns, n_truth, _lib.pointer(ranks)), "ranks")
print("tied rows", tied.shape[0], "with rank <", k, ":", int((ranks[tied] < k).sum()), "distinct sums among tied", len(np.unique(w.sums32[tied])), "ranks took", round(time.time() - t0, 1), "s")
print("rank histogram of the tied rows (first 12 values):", np.bincount(np.minimum(ranks[tied], 300))[:12], " max rank", int(ranks[tied].max()))
lengths = np.diff(w.t_off)[tied]
print("title lengths of tied rows:", np.unique(lengths)[:10], "example titles:", synth._to_strings(w.t_flat[w.t_off[tied[0]]:w.t_off[tied[0] + 1]], np.array([0, lengths[0]])), synth._to_strings(w.t_flat[w.t_off[tied[-1]]:w.t_off[tied[-1] + 1]], np.array([0, lengths[-1]])))

# ---- the first handed-over query alone, with and without the sums32 order
import os
from doppel_speller_amd.distributed import slice_queries
for sort in ("1", "0"):
    os.environ["DS_SORT_ROWS"] = sort
    idx = ds.TruthIndex(w.rowptr, w.truth_idx, w.idf32, w.sums32)
    for kk in (100, 10):
        rp, cc, mm = slice_queries(w.q_rowptr, w.q_cols, w.q_maxint, int(q), int(q) + 1)
        dd = [ds._lib.DeviceArray.from_host(np.ascontiguousarray(x)) for x in (rp, cc, mm)]
        out = ds._lib.DeviceArray((1, kk), np.int32)
        idx.top_k_device(dd[0].ptr, dd[1].ptr, dd[2].ptr, 1, kk, out.ptr)
        st = idx.sync()
        print("sort", sort, "k", kk, {key: st[key] for key in ("dense_queries", "dense_reasons", "selections", "sparse_tiles", "dense_tiles", "sparse_redos", "exact_candidates", "topk_kernel_ms", "dense_kernel_ms", "refines", "raw_entries", "refine_survivors", "raw_entries_sparse")})
    idx.close()
