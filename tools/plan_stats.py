import sys, time
sys.path.insert(0, '/root/repo')
import bench, torch
import deepquantum_amd as dq
from deepquantum_amd import fusion, executor
n = 28
wide = len(sys.argv) > 1 and sys.argv[1] == 'wide'
prims = []
for op in bench.random_circuit_spec(n, 40, 1234):
    if op[0] == 'cnot':
        prims.append(executor.Prim('x', None, (n - 1 - op[2],), (n - 1 - op[1],), 0))
    else:
        prims.append(executor.Prim('gen', None, (n - 1 - op[1],), (), 3 if op[0] == 'h' else 2))
groups, order, multi, levels = executor._merge_structure(prims)
merged = []
for kind, idx in order:
    if kind == 's':
        continue
    if kind == 'p':
        merged.append(prims[idx])
    else:
        g = groups[idx]
        merged.append(executor.Prim('gen', None, prims[g[0][0]].targets, (), g[1]))
print('merged gates', len(merged), {m: sum(1 for p in merged if p.kind == 'gen' and p.mode == m) for m in range(4)}, 'x', sum(p.kind == 'x' for p in merged))
ops = [fusion.PrimOp(p.kind, p.targets, p.controls, 4 * i, p.mode) for i, p in enumerate(merged)]
geom = fusion.default_geometry(False)
geom.permute_store = True
if wide:
    geom.plan_width, geom.plan_branch, geom.plan_restarts = 8, 4, 6
import deepquantum_amd.fusion as F
orig = F._place_writes
def spy(ops_, n_, pending, permute, final_perm=None):
    tiles = [set(it[1]) | set(it[2]) for it in pending if not isinstance(it, F.SingleStep)]
    print('passes', len(tiles), 'shared with next:', [len(a & b) for a, b in zip(tiles, tiles[1:])], 'gates', [sum(len(r.ops) for r in it[3]) for it in pending])
    return orig(ops_, n_, pending, permute, final_perm)
F._place_writes = spy
t0 = time.time()
steps = fusion.schedule(ops, n, geom)
print(len(steps), 'passes', time.time() - t0, 's')
runs = []
for st in steps:
    d = st.desc
    wt = [d.store_low_pos[i] for i in range(d.L)] + [d.store_high_pos[i] for i in range(d.h)]
    lanes = sorted(wt[d.store_tb[i]] for i in range(6))
    r = 1
    while r < 12 and (r in lanes or False) and all(x in lanes + [0] for x in range(1, r + 1)):
        r += 1
    k = 0
    while (k + 1) in lanes: k += 1
    runs.append(16 << k)
print('write run bytes per pass', runs)
