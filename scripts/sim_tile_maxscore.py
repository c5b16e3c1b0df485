"""Offline model of MaxScore per tile (profiles/r04_tuning.txt, batches 8-9): essential postings per query under the final thresholds
with the selection-wide skip set, with one per tile, and without the rule that a skipped column owns a signature bit.
usage: sim_tile_maxscore.py <truth titles> <k>   (numpy only, ~1 min at 500k)"""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from doppel_speller_amd import synth
N, Q, K, TILE = int(sys.argv[1]), 150, int(sys.argv[2]), 12288
w = synth.make_workload(N, 2000, seed=20260101)
rowptr, tidx, idf32, sums32 = np.asarray(w.rowptr), np.asarray(w.truth_idx), np.asarray(w.idf32), np.asarray(w.sums32)
order = np.argsort(sums32, kind='stable'); pos = np.empty(N, np.int64); pos[order] = np.arange(N)
sums_sorted = sums32[order]
ntiles = (N + TILE - 1) // TILE
tile_min = np.array([sums_sorted[t*TILE] for t in range(ntiles)]); tile_max = np.array([sums_sorted[min(N, (t+1)*TILE)-1] for t in range(ntiles)])
df = np.diff(rowptr)
dense_rank = np.argsort(-df, kind='stable')
has_sig = np.zeros(len(df), bool); top = dense_rank[:128]; has_sig[top[df[top]*256 >= N]] = True
print('signature columns', has_sig.sum(), 'min df among them', df[has_sig].min() if has_sig.any() else None, 'N', N)
rng = np.random.RandomState(1)
tot_g = tot_t = tot_all = 0; tiles_g = 0; skips_g = []; skips_t = []
tot_t_nosig = 0
for q in rng.choice(2000, Q, replace=False):
    cols = np.asarray(w.q_cols[w.q_rowptr[q]:w.q_rowptr[q+1]]); n = len(cols)
    if n == 0: continue
    maxint = float(w.q_maxint[q])
    scores = np.zeros(N, np.float32)
    for c in cols: scores[tidx[rowptr[c]:rowptr[c+1]]] += idf32[c]
    jac = scores.astype(np.float64) / (sums32.astype(np.float64) + (maxint - scores))
    kth = np.partition(jac, N-K)[N-K]
    if kth <= 0: continue
    cut = kth - 1e-5; coef = cut / (1 + cut)
    o = np.argsort(idf32[cols], kind='stable'); cs = cols[o]; mass = np.cumsum(idf32[cs].astype(np.float64))
    sig_ok = np.cumprod(has_sig[cs]).astype(bool)
    def skip_for(pre, need_sig=True):
        ok = (mass < pre) & (sig_ok if need_sig else True)
        return int(np.argmin(ok)) if not ok.all() else n
    # per column per tile counts
    counts = np.zeros((n, ntiles), np.int64)
    for i, c in enumerate(cs):
        t = pos[tidx[rowptr[c]:rowptr[c+1]]] // TILE
        counts[i] = np.bincount(t, minlength=ntiles)
    band = (tile_min * cut <= maxint) & (tile_max >= cut * maxint)
    g = skip_for(coef * (sums_sorted[0] + maxint)); skips_g.append(g)
    tot_all += counts[:, band].sum()
    tot_g += counts[g:, band].sum(); tiles_g += band.sum()
    for t in np.nonzero(band)[0]:
        st = skip_for(coef * (tile_min[t] + maxint)); skips_t.append(st)
        tot_t += counts[st:, t].sum()
        tot_t_nosig += counts[skip_for(coef * (tile_min[t] + maxint), False):, t].sum()
print('k', K, 'tiles in band per query', tiles_g / Q, 'of', ntiles)
print('postings in band, all columns', tot_all / Q, ' global skip', tot_g / Q, ' per-tile skip', tot_t / Q, ' per-tile skip without the signature limit', tot_t_nosig / Q)
print('mean skipped columns: global', np.mean(skips_g), 'per tile', np.mean(skips_t))
