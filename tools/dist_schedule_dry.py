"""Dry run (no data, no GPU) of one rank's schedule of a sharded circuit: the commutation-DAG order, the remaps, and the
fused passes of every local stretch with their gate counts -- what `bench.py --rehearse-rank` runs, for sizing the
exchange schedule offline.  usage: python tools/dist_schedule_dry.py [n] [g = log2 ranks] [rank] [--place] [--defer]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from deepquantum_amd import distributed as D, executor, fusion


def gate_prims(n, depth=40, seed=1234):
    prims = []
    for op in bench.random_circuit_spec(n, depth, seed):
        if op[0] == 'cnot':
            prims.append(executor.Prim('x', None, (n - 1 - op[2],), (n - 1 - op[1],), 0))
        else:
            prims.append(executor.Prim('gen', None, (n - 1 - op[1],), (), 3 if op[0] == 'h' else 2))
    return prims


def merged(prims):
    groups, order, multi, levels = executor._merge_structure(prims)
    out = []
    for kind, idx in order:
        if kind == 's':          # (a scalar product rides on another gate's matrix: executor._merge_structure)
            continue
        out.append(prims[idx] if kind == 'p' else executor.Prim('gen', None, prims[groups[idx][0][0]].targets, (), groups[idx][1]))
    return out


def stretches(n, g, rank, prims, place=False):
    """[(localized prims of the stretch, pairs of the remap behind it or None)] for rank ``rank``."""
    L = n - g
    ph = list(range(n))
    if place:
        ph = D.initial_placement(prims, n, L, 0)
    order = D._order_for_remaps(prims, ph, n, L, 0)
    out, pending, i = [], [], 0
    while i < len(order):
        p = D._translate(order[i], ph)
        loc = D._localize_at(L, rank, p)
        if loc is None:
            i += 1
            continue
        if loc != 'exchange':
            pending.append(loc)
            i += 1
            continue
        pairs = D._plan_remap(ph, order, i, n, L, 0)
        pairs = sorted(pairs, key=lambda pr: ph[pr[0]])
        rbits = [ph[lq] - L for lq, _ in pairs]
        ent = [ph[eq] for _, eq in pairs]
        src_of_dst = [b for b in range(L) if b not in ent] + ent
        out_perm = [0] * L
        for d, sp in enumerate(src_of_dst):
            out_perm[sp] = d
        out.append((pending, out_perm))
        D._remap_bookkeeping(ph, pairs, rbits, out_perm, L)
        pending = []
    out.append((pending, None))
    return out


def plan_stretch(prims, L, out_perm):
    prims = merged(prims)
    plan = executor.make_plan(prims, L, False, True, out_perm, amps=1 << L)
    steps = [s for s in plan.steps if isinstance(s, fusion.FusedStep)]
    return [len(s.ops) for s in steps], len(prims)


if __name__ == '__main__':
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    n = int(args[0]) if args else 34
    g = int(args[1]) if len(args) > 1 else 3
    rank = int(args[2]) if len(args) > 2 else 1
    prims = gate_prims(n)
    st = stretches(n, g, rank, prims, place='--place' in sys.argv)
    total = 0
    for k, (pending, out_perm) in enumerate(st):
        passes, ng = plan_stretch(pending, n - g, out_perm) if pending else ([], 0)
        total += len(passes)
        print(f'stretch {k}: {len(pending)} gates ({ng} after merging) -> {len(passes)} passes {passes}' + ('' if out_perm is not None else '  (last)'))
    print('passes', total, 'remaps', len(st) - 1)
