"""Offline model for round 5's per-wave row ownership: essential posting quads per (tile, row quarter) under the final thresholds,
padding of the per-quarter cut, and lane utilisation of two work distributions (flattened items of 64 quads / G lanes per list).
usage: sim_wave_tiles.py <truth titles> <k> [tile rows] [waves]   (numpy only)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from doppel_speller_amd import synth
N, Q, K = int(sys.argv[1]), 150, int(sys.argv[2])
TILE = int(sys.argv[3]) if len(sys.argv) > 3 else 12288
WAVES = int(sys.argv[4]) if len(sys.argv) > 4 else 4
SUB = TILE // WAVES
w = synth.make_workload(N, 2000, seed=20260101)
rowptr, tidx, idf32, sums32 = np.asarray(w.rowptr), np.asarray(w.truth_idx), np.asarray(w.idf32), np.asarray(w.sums32)
order = np.argsort(sums32, kind='stable'); pos = np.empty(N, np.int64); pos[order] = np.arange(N)
sums_sorted = sums32[order]
ntiles = (N + TILE - 1) // TILE
tile_min = np.array([sums_sorted[t*TILE] for t in range(ntiles)]); tile_max = np.array([sums_sorted[min(N, (t+1)*TILE)-1] for t in range(ntiles)])
df = np.diff(rowptr)
dense_rank = np.argsort(-df, kind='stable')
has_sig = np.zeros(len(df), bool); top = dense_rank[:128]; has_sig[top[df[top]*256 >= N]] = True
rng = np.random.RandomState(1)
stat = dict(tiles=0, quads_tile=0, quads_sub=0, postings=0, lists_sub=0, subs=0,
            items_flat=0, slots_g8=0, slots_g16=0, slots_g4=0, maxq=[], lists=[])
nq = 0
for q in rng.choice(2000, Q, replace=False):
    cols = np.asarray(w.q_cols[w.q_rowptr[q]:w.q_rowptr[q+1]]); n = len(cols)
    if n == 0: continue
    maxint = float(w.q_maxint[q])
    scores = np.zeros(N, np.float32)
    for c in cols: scores[tidx[rowptr[c]:rowptr[c+1]]] += idf32[c]
    jac = scores.astype(np.float64) / (sums32.astype(np.float64) + (maxint - scores))
    kth = np.partition(jac, N-K)[N-K]
    if kth <= 0: continue
    nq += 1
    cut = kth - 1e-5; coef = cut / (1 + cut)
    o = np.argsort(idf32[cols], kind='stable'); cs = cols[o]; mass = np.cumsum(idf32[cs].astype(np.float64))
    sig_ok = np.cumprod(has_sig[cs]).astype(bool)
    ok = (mass < coef * (sums_sorted[0] + maxint)) & sig_ok
    g = int(np.argmin(ok)) if not ok.all() else n
    band = (tile_min * cut <= maxint) & (tile_max >= cut * maxint)
    nsub = ntiles * WAVES
    # per essential column: count per (sub-tile, parity)
    cnt = np.zeros((n - g, nsub, 2), np.int64)
    for i, c in enumerate(cs[g:]):
        p = pos[tidx[rowptr[c]:rowptr[c+1]]]
        np.add.at(cnt[i], (p // SUB, p & 1), 1)
    quads_sub = ((cnt + 3) // 4).sum(axis=2)            # [list, sub]: quads with per-(sub,parity) padding
    cnt_tile = cnt.reshape(n - g, ntiles, WAVES, 2).sum(axis=2)
    quads_tile = ((cnt_tile + 3) // 4).sum(axis=2)       # [list, tile]: padding per (tile, parity) as today
    bt = np.nonzero(band)[0]
    stat['tiles'] += len(bt)
    stat['quads_tile'] += quads_tile[:, bt].sum()
    stat['postings'] += cnt_tile[:, bt].sum()
    for t in bt:
        for s in range(t * WAVES, (t + 1) * WAVES):
            ql = quads_sub[:, s]; ql = ql[ql > 0]
            tot = ql.sum()
            stat.setdefault('totq', []).append(tot); stat['subs'] += 1; stat['quads_sub'] += tot; stat['lists_sub'] += len(ql)
            stat['items_flat'] += (tot + 63) // 64
            for G, key in ((4, 'slots_g4'), (8, 'slots_g8'), (16, 'slots_g16')):
                groups = 64 // G
                # lists dealt round-robin to groups in compact order; a group's slots = sum of ceil(len / G); the wave runs the max
                per = np.zeros(groups, np.int64)
                for i, L in enumerate(ql): per[i % groups] += (L + G - 1) // G
                stat[key] += per.max() if len(ql) else 0
            stat['maxq'].append(ql.max() if len(ql) else 0); stat['lists'].append(len(ql))
            qa0 = quads_sub[:, s]
            for G, budget in ((8, (4, 2, 1, 1)), (8, (3, 2, 2, 1)), (8, (5, 2, 1)), (8, (4, 2, 2)), (8, (3, 2, 1)), (8, (6, 3, 2, 1)), (8, (2, 2, 1, 1)), (16, (3, 2, 2, 1)), (16, (2, 2, 1, 1, 1, 1)), (4, (4, 2, 1, 1)), (8, (4, 4)), (8, (2, 1, 1)), (8, (2,2))):
                key = 'budget_G%d_%s' % (G, '_'.join(map(str, budget)))
                groups = 64 // G
                covered = 0
                for e, L in enumerate(qa0):
                    j = e // groups
                    covered += min(L, G * budget[j]) if j < len(budget) else 0
                d = stat.setdefault(key, [0, 0, 0])
                d[0] += covered; d[1] += qa0.sum(); d[2] += covered == qa0.sum()
            for name, first_lists, G1, G2 in (('h8_8_4', 8, 8, 4), ('h8_8_2', 8, 8, 2), ('h4_16_4', 4, 16, 4), ('h4_16_8', 4, 16, 8), ('h16_4_2', 16, 4, 2), ('h8_8_1', 8, 8, 1)):
                slots = 0
                head = qa0[:first_lists]
                for e0 in range(0, len(head), 64 // G1):
                    chunk = head[e0:e0 + 64 // G1]; slots += (chunk.max() + G1 - 1) // G1
                tail = qa0[first_lists:]
                for e0 in range(0, len(tail), 64 // G2):
                    chunk = tail[e0:e0 + 64 // G2]; slots += (chunk.max() + G2 - 1) // G2
                stat['static_' + name] = stat.get('static_' + name, 0) + slots
            qa = quads_sub[:, s]   # static assignment: list e (all essential lists, compact rank order) -> group (e // passes) or e % groups
            for G, key in ((4, 'static_g4'), (8, 'static_g8'), (16, 'static_g16')):
                groups = 64 // G
                tot_slots = 0
                for e0 in range(0, len(qa), groups):
                    chunk = qa[e0:e0 + groups]
                    tot_slots += ((chunk.max() + G - 1) // G) if len(chunk) else 0
                stat[key] = stat.get(key, 0) + tot_slots
                # unrolled by two: iterations of two slots
                t2 = 0
                for e0 in range(0, len(qa), groups):
                    chunk = qa[e0:e0 + groups]
                    t2 += ((chunk.max() + 2 * G - 1) // (2 * G)) if len(chunk) else 0
                stat[key + '_x2'] = stat.get(key + '_x2', 0) + t2
S = stat
print(f"N {N} k {K} tile {TILE} waves {WAVES}: queries {nq}, band tiles/query {S['tiles']/nq:.1f} of {ntiles}")
print(f"essential postings/query {S['postings']/nq:.0f}; quads/query padded per tile {S['quads_tile']/nq:.0f}, per sub-tile {S['quads_sub']/nq:.0f} (+{100*(S['quads_sub']/S['quads_tile']-1):.1f} %)")
print(f"per sub-tile: quads {S['quads_sub']/S['subs']:.1f}, non-empty lists {S['lists_sub']/S['subs']:.1f}; longest list p50/p90/p99 {np.percentile(S['maxq'],[50,90,99])}; lists p50/p90/p99 {np.percentile(S['lists'],[50,90,99])}")
print('quads per sub-tile percentiles 50/75/90/95/99/99.9', np.percentile(S['totq'], [50, 75, 90, 95, 99, 99.9]), 'share of sub-tiles above 128 / 192 / 256 quads', [float(np.mean(np.array(S['totq']) > x)) for x in (128, 192, 256)])
for k_, v_ in S.items():
    if k_.startswith('budget'): print(k_, 'quads covered %.3f' % (v_[0] / v_[1]), 'sub-tiles fully covered %.3f' % (v_[2] / S['subs']))
print('static pass assignment (lists in ascending-IDF order, 64/G per pass, slots = longest list of the pass):', {k: round(v / S['subs'], 2) for k, v in S.items() if k.startswith('static')})
print(f"wave-instruction slots per sub-tile (one quad per lane each): flattened {S['items_flat']/S['subs']:.2f}, G=4 {S['slots_g4']/S['subs']:.2f}, G=8 {S['slots_g8']/S['subs']:.2f}, G=16 {S['slots_g16']/S['subs']:.2f}")
