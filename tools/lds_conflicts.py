"""Offline: exact LDS bank-conflict count of every layout change of the headline circuit's passes, from the pass
descriptors the host emits (same address formula as csrc/dq_fused.hip: swizzle(thread base) * 8 XOR table entry).
Model (MI355X_MICROARCH.md, LDS): ds_write_b64 is served in 4 groups of 16 contiguous lanes over 32 banks (16 8-byte
slots), ds_read_b64 in 2 groups of 32 lanes over 64 banks (32 slots); a group costs as many LDS cycles as the most
loaded slot has distinct addresses.  Prints extra cycles / ideal cycles (what SQ_LDS_BANK_CONFLICT / (SQ_LDS_IDX_ACTIVE -
SQ_LDS_BANK_CONFLICT) would show).  usage: python tools/lds_conflicts.py [N=28]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepquantum_amd as dq
from deepquantum_amd import _lib, fusion
import bench


def swz(e):
    return fusion.lds_swizzle(e, 5) if hasattr(fusion, 'lds_swizzle') else e ^ ((e >> 5) & 31)


def group_cost(addrs, groups, slots):
    tot = 0
    for g in groups:
        b = {}
        for l in g:
            b.setdefault((addrs[l] >> 3) % slots, set()).add(addrs[l])
        tot += max(len(v) for v in b.values())
    return tot


WG = [list(range(i, i + 16)) for i in range(0, 64, 16)]
RG = [list(range(0, 32)), list(range(32, 64))]


def layout_cost(tbits, tab, nthreads, write):
    """Cycles of the 16 accesses of every wave of a workgroup in layout (thread-bit list, table)."""
    tot = ideal = 0
    for w in range(nthreads // 64):
        base = []
        for l in range(64):
            tid = w * 64 + l
            e = 0
            for i, p in enumerate(tbits):
                e |= ((tid >> i) & 1) << p
            base.append(swz(e) * 8)
        for j in range(16):
            addrs = [b ^ tab[j] for b in base]
            tot += group_cost(addrs, WG if write else RG, 16 if write else 32)
            ideal += 4 if write else 2
    return tot, ideal


def main():
    n = int(os.environ.get('N', 28))
    spec = bench.random_circuit_spec(n, 40, 1234)
    cir, data = bench.build_circuit(dq, n, spec, 2, torch.complex64, torch.device('cpu'))
    cir.encode(data)
    merged = dq.executor.merge_one_qubit_runs(cir.prims())
    plan = dq.executor.make_plan(merged, n, False, True)
    W = I = RW = RI = 0
    for st in plan.steps:
        if not isinstance(st, fusion.FusedStep):
            continue
        d = st.desc
        m, R = d.m, 4
        nthreads = 1 << (m - R)
        def io_tbits(rb):
            return [b for b in range(m) if b not in list(rb)]
        cur_t, cur_tab = io_tbits(d.load_rb), list(d.lds_tab[0])
        for r in range(d.nrounds):
            rd = d.rounds[r]
            if rd.flags & _lib.ROUND_TRANSPOSE:
                nt, ntab = list(rd.tb)[: m - R], list(d.lds_tab[1 + r])
                a, b = layout_cost(cur_t, cur_tab, nthreads, True); W += a; I += b
                a, b = layout_cost(nt, ntab, nthreads, False); RW += a; RI += b
                cur_t, cur_tab = nt, ntab
            if rd.flags & _lib.ROUND_TRANSPOSE_AFTER:
                nt, ntab = [d.store_tb[i] for i in range(m - R)], list(d.lds_tab[_lib.FUSED_MAX_ROUNDS + 1])
                a, b = layout_cost(cur_t, cur_tab, nthreads, True); W += a; I += b
                a, b = layout_cost(nt, ntab, nthreads, False); RW += a; RI += b
    print(f'writes: {W} cycles, ideal {I} (x{W / I:.2f}); reads: {RW}, ideal {RI} (x{RW / RI:.2f}); '
          f'conflict / ideal overall {(W + RW - I - RI) / (I + RI):.2f}')


if __name__ == '__main__':
    main()
