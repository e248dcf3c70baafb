"""Per-pass instruction counts of a schedule from the library's own records (dq_wave_descriptor: no GPU needed) and the
generator's handler bodies: VALU / DS / SALU instructions a wave executes for every record of every pass of the headline
circuit, next to the durations measured on the GPU (profiles/r04/passes_headline.txt), and a least-squares model
ms = max(floor, a + b * valu) fitted to them.  usage: python tools/pass_cost_model.py [measured.txt]"""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import bench
from deepquantum_amd import fusion, executor
import _wave_emulator as emu


def handler_costs(g):
    """{handler id: (VALU, DS, SALU) instructions a wave EXECUTES in its body}: masked bodies run whole; of the two variants
    of a deferred Rx body (f [[1, it], [it, 1]] / f [[it, 1], [1, it]]) one runs."""
    out = {}
    for i, (ctl, lines) in g.handlers().items():
        v = sum(1 for ln in lines if ln.lstrip().startswith('v_'))
        if g.ID_GEN_U + 12 <= i < g.ID_GEN_U + 18:
            v = v // 2 - 1
        d = sum(1 for ln in lines if ln.lstrip().startswith('ds_'))
        s = sum(1 for ln in lines if ln.lstrip().startswith('s_'))
        out[i] = (v + (2 if ctl else 0), d, s + (6 if ctl else 0))
    return out


def headline_steps(n=28, depth=40, seed=1234, wide=True):
    prims = []
    for op in bench.random_circuit_spec(n, depth, seed):
        if op[0] == 'cnot':
            prims.append(executor.Prim('x', None, (n - 1 - op[2],), (n - 1 - op[1],), 0))
        else:
            prims.append(executor.Prim('gen', None, (n - 1 - op[1],), (), 3 if op[0] == 'h' else 2))
    groups, order, multi, levels = executor._merge_structure(prims)
    merged = []
    for kind, idx in order:
        if kind == 's':          # (a scalar product rides on another gate's matrix: executor._merge_structure)
            continue
        merged.append(prims[idx] if kind == 'p' else executor.Prim('gen', None, prims[groups[idx][0][0]].targets, (), groups[idx][1]))
    ops = [fusion.PrimOp(p.kind, p.targets, p.controls, 4 * i, p.mode) for i, p in enumerate(merged)]
    geom = fusion.default_geometry(False)
    geom.permute_store = True
    if wide:
        geom.plan_width, geom.plan_branch, geom.plan_restarts = 8, 4, 6
    return fusion.schedule(ops, n, geom), ops


def pass_counts(steps, n):
    g = emu.gen()
    hc = handler_costs(g)
    rows = []
    for st in steps:
        kp = emu.descriptor(st.desc, n)
        ids = [kp.rec[j][0] for j in range(kp.nrec_bytes // 32)]
        v = sum(hc.get(i, (0, 0, 0))[0] for i in ids)
        d = sum(hc.get(i, (0, 0, 0))[1] for i in ids)
        unknown = [i for i in ids if i not in hc]
        rows.append({'records': len(ids), 'valu': v, 'ds': d, 'unknown': len(unknown), 'ids': ids})
    return rows


if __name__ == '__main__':
    steps, ops = headline_steps()
    rows = pass_counts(steps, 28)
    meas = None
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'profiles', 'r05', 'passes_headline.txt')
    if os.path.exists(path):
        meas = [float(m.group(1)) for m in re.finditer(r'trips \d+\s+([\d.]+) ms', open(path).read())]
    for i, r in enumerate(rows):
        print(f'pass {i:2d}: records {r["records"]:3d} valu {r["valu"]:5d} ds {r["ds"]:4d} unknown-handlers {r["unknown"]}'
              + (f'  measured {meas[i]:6.2f} ms' if meas and i < len(meas) else ''))
    if meas and len(meas) == len(rows):
        full = [(r['valu'], m) for r, m in zip(rows[4:-1], meas[4:-1])]
        light = [m for v, m in full if v <= 3300]
        if light:
            print(f'full passes of at most 3300 VALU instructions per tile: {len(light)}, {min(light):.2f} .. {max(light):.2f} ms (the plateau)')
        x = np.array([v for v, m in full if v > 3300 or not light], float); y = np.array([m for v, m in full if v > 3300 or not light])
        A = np.stack([np.ones_like(x), x], 1)
        coef, res, *_ = np.linalg.lstsq(A, y, rcond=None)
        print(f'fit over the full passes 4..17 above the plateau: ms = {coef[0]:.2f} + {coef[1] * 1e3:.3f}e-3 * VALU; residual rms '
              f'{float(np.sqrt(((A @ coef - y) ** 2).mean())):.2f} ms   (a purely VALU-bound kernel: 1024 tiles per SIMD x 2.07 ns = 2.12e-3)')
        g = emu.gen()
        hc = handler_costs(g)
        names = sorted((getattr(g, k), k) for k in dir(g) if k.startswith('ID_'))
        cls = lambda i: [k for v_, k in names if v_ <= i][-1]         # noqa: E731
        from collections import Counter
        cv, cn = Counter(), Counter()
        for r in rows[4:]:
            for i in r['ids']:
                key = cls(i) + (f' mode {i // 6}' if i < 24 else '')
                cv[key] += hc.get(i, (0, 0, 0))[0]
                cn[key] += 1
        tot = sum(cv.values())
        print(f'VALU instructions executed per tile over the full passes of a step: {tot}')
        for k, v_ in cv.most_common():
            if v_:
                print(f'  {k:18s} {cn[k]:4d} records  {v_:6d}  {v_ / tot:.3f}  ({v_ / cn[k]:.0f} per record)')
