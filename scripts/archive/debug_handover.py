"""Why does the fast kernel hand queries over?  Runs C2-shaped work, lists the handed-over queries and looks at the jaccard
values around their k-th best row (globally and in the start tile of the sums32 order)."""
import sys, numpy as np
sys.path.insert(0, ".")
import doppel_speller_amd as ds
from doppel_speller_amd import synth
k = int(sys.argv[1]) if len(sys.argv) > 1 else 10
w = synth.make_workload(500_000, 100_000, seed=20260101)
pipeline = ds.CandidatePipeline(w, k)
pipeline.enqueue_top_k(); stats = pipeline.sync()
status = pipeline.index.status(w.n_queries)
print(stats["dense_reasons"], "handed over:", int((status == 1).sum()))
order = np.argsort(w.sums32, kind="stable")
sums_sorted = w.sums32[order]
tile = 12288
handed = np.nonzero(status == 1)[0]
shown = 0
for q in handed:
    cols = w.q_cols[w.q_rowptr[q]:w.q_rowptr[q + 1]]
    M = w.q_maxint[q]
    if len(cols) == 0: continue
    scores = np.zeros(w.n_truth, dtype=np.float32)
    for c in cols:
        scores[w.truth_idx[w.rowptr[c]:w.rowptr[c + 1]]] += w.idf32[c]
    if (scores > 0).sum() < k: continue          # 'few'
    jac = scores / (w.sums32 + (np.float32(M) - scores))
    js = np.sort(jac)[::-1]
    kth = js[k - 1]
    n = len(cols)
    margin = (6 * n + 64) * 5.96e-8 + n * 0.3 * (4.0 / 65000)
    cut = kth - 2 * margin - 2e-6
    start = min(int(np.searchsorted(sums_sorted[tile - 1::tile], np.float32(M))), (w.n_truth - 1) // tile)
    title = synth._to_strings(w.q_flat[w.q_off[q]:w.q_off[q + 1]], np.array([0, w.q_off[q + 1] - w.q_off[q]]))[0]
    print(f"q={q} '{title}' n={n} M={M:.2f} kth={kth:.5f} margin={margin:.2e} rows>=cut: {(jac >= cut).sum()} rows>=kth-1e-6: {(jac >= kth - 1e-6).sum()} "
          f"rows>=0.97kth: {(jac >= 0.97*kth).sum()} positive: {(scores>0).sum()} start tile {start}")
    shown += 1
    if shown >= 25: break
