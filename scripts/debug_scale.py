import sys, numpy as np
sys.path.insert(0, '.')
import doppel_speller_amd as ds
from doppel_speller_amd import synth
from oracle import oracle
w = synth.make_workload(1_200_000, 3000, seed=77)
k = 50
index = ds.TruthIndex(w.rowptr, w.truth_idx, w.idf32, w.sums32)
rows = index.top_k(w.q_rowptr, w.q_cols, w.q_maxint, k)
stats = index.sync(); print(stats['dense_reasons'], stats['dense_queries'])
status = index.status(3000)
bad = np.nonzero(~(np.diff(rows.astype(np.int64), axis=1) < 0).all(axis=1))[0]
print('non-descending queries', bad[:20], len(bad), 'status', status[bad][:20])
sample = bad[:6]
first, last = w.q_rowptr[sample], w.q_rowptr[sample + 1]
q_cols = np.concatenate([w.q_cols[a:b] for a, b in zip(first, last)])
q_rowptr = np.concatenate(([0], np.cumsum(last - first)))
expected = oracle.jaccard_topk(w.rowptr, w.truth_idx, w.idf32, w.sums32, q_rowptr, q_cols, w.q_maxint[sample], k)
for i, q in enumerate(sample):
    got = rows[q]
    print('query', q, 'cols', last[i]-first[i], 'missing', sorted(set(expected[i]) - set(got))[:10], 'extra', sorted(set(got) - set(expected[i]))[:10])
    print('   got', got[:12], '... expected', expected[i][:12])
