"""Offline (CPU) evaluation of the pass scheduler on the headline circuit family: passes, LDS trips and gates per
pass for a few seeds.  usage: python tools/sched_eval.py [--n 28] [--depth 40] [--seeds 1234,7,99] [--c128]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepquantum_amd import fusion  # noqa: E402
from bench import random_circuit_spec  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=28)
ap.add_argument('--depth', type=int, default=40)
ap.add_argument('--seeds', default='1234,7,99')
ap.add_argument('--c128', action='store_true')
args = ap.parse_args()


def prim_ops(n, spec):
    ops = []
    for op in spec:
        if op[0] == 'h':
            ops.append(fusion.PrimOp('gen', (n - 1 - op[1],), (), 0, 3))
        elif op[0] == 'rx':
            ops.append(fusion.PrimOp('gen', (n - 1 - op[1],), (), 0, 2))
        else:
            ops.append(fusion.PrimOp('x', (n - 1 - op[2],), (n - 1 - op[1],), 0, 0))
    return ops


for seed in [int(s) for s in args.seeds.split(',')]:
    ops = prim_ops(args.n, random_circuit_spec(args.n, args.depth, seed))
    steps = fusion.schedule(ops, args.n, fusion.default_geometry(args.c128))
    fused = [s for s in steps if isinstance(s, fusion.FusedStep)]
    big = sum(1 for s in fused if s.desc.m == fusion.default_geometry(args.c128).m)
    print(f'seed {seed}: {len(ops)} gates -> {len(fused)} passes ({big} on the big tile), '
          f'{sum(s.ntranspose for s in fused)} LDS trips, {len(steps) - len(fused)} single steps')
