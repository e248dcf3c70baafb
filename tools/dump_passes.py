"""Per-pass breakdown of the headline workload: gates, rounds, LDS trips and measured duration."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepquantum_amd as dq
import bench
from bench import random_circuit_spec
n, depth, batch = 28, 40, int(os.environ.get('B', 16))
dev = torch.device('cuda', 0)
spec = random_circuit_spec(n, depth, 1234)
cir, data = bench.build_circuit(dq, n, spec, batch, torch.complex64, dev)
with torch.no_grad():
    cir(data)
    dq.executor.PROFILE['enabled'] = True
    dq.executor.PROFILE['events'].clear()
    cir(data)
    torch.cuda.synchronize()
plan = list(dq.executor._PLAN_CACHE.values())[-1]
steps = [s for s in plan.steps if isinstance(s, dq.fusion.FusedStep)]
ev = dq.executor.PROFILE['events']
tot = 0
for i, (s, (a, b, ng, _nb)) in enumerate(zip(steps, ev)):
    ms = a.elapsed_time(b); tot += ms
    kinds = {}
    for oi in s.ops:
        op = plan.prim_ops[oi]
        k = 'x' if op.kind == 'x' else ('h' if op.mode == 1 else 'rx' if op.mode == 2 else 'g')
        kinds[k] = kinds.get(k, 0) + 1
    extra = ''
    if s.desc.slots == 6:        # wave tile: what the library made of the rounds (csrc/dq_wave.hip)
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
        import _wave_emulator as emu
        g = emu.gen()
        kp = emu.descriptor(s.desc, n)
        ids = [kp.rec[j][0] for j in range(kp.nrec_bytes // 32)]
        trips = [g.TRIP_MASKS[j - g.ID_TRIP] for j in ids if g.ID_TRIP <= j < g.ID_SWAP]
        extra = (f' records {len(ids)} trips k={[bin(t).count("1") for t in trips]} lane-perm {ids.count(g.ID_TRIP0)} '
                 f'swaps {sum(j >= g.ID_SWAP for j in ids)} read {sorted(s.desc.high_sorted[j] for j in range(s.desc.h))} '
                 f'write-lanes {sorted(kp.store_lane_shift[j] - 3 for j in range(6))} write-slots {sorted(int(kp.store_off[j]).bit_length() - 4 for j in range(5))}')
    zm = plan._zero_masks[plan.steps.index(s)] if plan._zero_masks else 0
    if zm:      # a circuit started from its own |0..0> (executor.CONFIG['zero_state']): index bits still known to be zero
        extra = f' known-zero bits {bin(zm).count("1")} ({_nb / 2**30:.2f} GiB moved)' + extra
    print(f'pass {i:2d}: gates {len(s.ops):3d} {kinds} rounds {s.nrounds} trips {s.ntranspose}  {ms:6.2f} ms{extra}')
print('total', tot)
