"""Dry run (no data, no GPU) for DESIGN 7 "pass-splitting overlap": could the LAST pass in front of a remap and the FIRST
pass behind it be launched in 2-4 slices by a tile-number bit, so that the wire starts after the first slice (VERDICT r5, item
1c)?  The two slice qubits are chosen rank-independently -- the local qubits with the farthest next use after the victims --
and placed right below the chunk bits by the remap's re-labelling; per rank and remap: are they outside the tile of the last
pass (`last_ok`) and of the next stretch's first pass (`first_ok`)?  usage: python tools/dist_slice_dry.py"""
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import dist_schedule_dry as T
from deepquantum_amd import distributed as D, executor, fusion

def tiles_of(pl):
    """per fused step: set of tile positions on the read side (stretch-start labels are lost after permuted stores, so use logical labels via plan internals)"""
    return None

def run(n, g, rank, ns=2):
    prims = T.gate_prims(n); L = n - g
    D._EVICT[0] = True
    ph = D.initial_placement(prims, n, L, 0)
    order = D._order_for_remaps(prims, ph, n, L, 0)
    i = 0; pending = []
    prevS = None
    res = []
    while True:
        hit = None
        while i < len(order):
            loc = D._localize_at(L, rank, D._translate(order[i], ph))
            if loc is None: i += 1; continue
            if loc == 'exchange': hit = i; break
            pending.append(loc); i += 1
        # first-tile check for previous remap's S (positions now L-k-1.. after bookkeeping)
        if prevS is not None and pending:
            mp = T.merged(pending)
            pl = executor.make_plan(mp, L, False, True, None, amps=1 << L)
            st0 = [s for s in pl.steps if isinstance(s, fusion.FusedStep)][0]
            d = st0.desc
            tile0 = set(range(d.L)) | {d.high_pos[j] for j in range(d.h)}
            res[-1]['first_ok'] = [p not in tile0 for p in prevS]
        if hit is None: break
        pairs = sorted(D._plan_remap(ph, order, hit, n, L, 0), key=lambda pr: ph[pr[0]])
        k = len(pairs)
        rbits = [ph[lq] - L for lq, _ in pairs]; ent = [ph[eq] for _, eq in pairs]
        # S: local qubits, not entering, farthest next use
        nxt = D._next_use(order, hit, n)
        cand = sorted((q for q in range(n) if ph[q] < L and ph[q] not in ent and ph[q] >= 4), key=lambda q: -nxt[q])
        S = cand[:ns]
        Spos = [ph[q] for q in S]
        rest = [b for b in range(L) if b not in ent and b not in Spos]
        src_of_dst = rest + Spos[::-1] + ent        # S at the top of the chunk
        out_perm = [0] * L
        for d_, sp in enumerate(src_of_dst): out_perm[sp] = d_
        mp = T.merged(pending)
        pl = executor.make_plan(mp, L, False, True, out_perm, amps=1 << L)
        steps = [s for s in pl.steps if isinstance(s, fusion.FusedStep)]
        last = steps[-1].desc
        # which WRITE positions are tile-number bits of the last pass?
        nblk = n  # unknown count; recompute
        tile_w = {last.store_low_pos[j] for j in range(last.L)} | {last.store_high_pos[j] for j in range(last.h)}
        want = [L - k - 1 - j for j in range(ns)]
        res.append({'remap': len(res) + 1, 'k': k, 'S_next_use_gap': [nxt[q] - hit for q in S], 'folded': pl.steps.applied_final_perm,
                    'last_ok': [w not in tile_w for w in want], 'npass': len(steps)})
        D._remap_bookkeeping(ph, pairs, rbits, out_perm, L)
        prevS = want
        pending = []
    return res

for rank in (0, 3, 5, 6):
    print('rank', rank)
    for r in run(34, 3, rank): print('  ', r)
