"""Dry run (no data, no GPU) for DESIGN 7 "tail deferral": when the last pass of a stretch is under-filled, move its gates
behind the exchange (legal when none of them targets a qubit that leaves for the rank bits) and count the passes of the whole
step again.  usage: python tools/dist_defer_dry.py"""
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import dist_schedule_dry as T
from deepquantum_amd import distributed as D, executor, fusion

def plan(pending, L, out_perm):
    mp = T.merged(pending)
    pl = executor.make_plan(mp, L, False, True, out_perm, amps=1 << L)
    steps = [s for s in pl.steps if isinstance(s, fusion.FusedStep)]
    return mp, pl, steps

def merged_members(pending):
    groups, order, multi, levels = executor._merge_structure(pending)
    out = []
    for kind, idx in order:
        if kind == 's': continue
        out.append([idx] if kind == 'p' else list(groups[idx][0]))
    return out

def run(n, g, rank, defer_max):
    prims = T.gate_prims(n); L = n - g
    D._EVICT[0] = True
    ph = D.initial_placement(prims, n, L, 0)
    order = D._order_for_remaps(prims, ph, n, L, 0)
    total = 0; detail = []
    i = 0; pending = []; src = []; carry = []
    def localize(p):
        return D._localize_at(L, rank, D._translate(p, ph))
    while True:
        # carried logical prims first
        for p in carry:
            loc = localize(p)
            assert loc != 'exchange'
            if loc is not None: pending.append(loc); src.append(p)
        carry = []
        hit = None
        while i < len(order):
            loc = localize(order[i])
            if loc is None: i += 1; continue
            if loc == 'exchange': hit = i; break
            pending.append(loc); src.append(order[i]); i += 1
        if hit is None:
            mp, pl, steps = plan(pending, L, None) if pending else (None, None, [])
            total += len(steps); detail.append([len(s.ops) for s in steps]); break
        pairs = sorted(D._plan_remap(ph, order, hit, n, L, 0), key=lambda pr: ph[pr[0]])
        rbits = [ph[lq] - L for lq, _ in pairs]; ent = [ph[eq] for _, eq in pairs]
        src_of_dst = [b for b in range(L) if b not in ent] + ent
        out_perm = [0] * L
        for d, sp in enumerate(src_of_dst): out_perm[sp] = d
        mp, pl, steps = plan(pending, L, out_perm)
        if defer_max and len(steps) > 1 and len(steps[-1].ops) <= defer_max:
            members = merged_members(pending)
            tail = [m for oi in steps[-1].ops for m in members[oi]]
            ok = all(not (set(pending[m].targets) & set(ent)) for m in tail)
            if ok:
                tailset = set(tail)
                carry = [src[m] for m in sorted(tail)]
                pending = [p for m, p in enumerate(pending) if m not in tailset]
                src2 = [p for m, p in enumerate(src) if m not in tailset]
                mp, pl, steps = plan(pending, L, out_perm)
        total += len(steps); detail.append([len(s.ops) for s in steps])
        D._remap_bookkeeping(ph, pairs, rbits, out_perm, L)
        pending = []; src = []
    return total, detail

for rank in (0, 5):
    for dm in (0, 8, 16, 24):
        t, d = run(34, 3, rank, dm)
        print('rank', rank, 'defer<=', dm, 'passes', t, d)
