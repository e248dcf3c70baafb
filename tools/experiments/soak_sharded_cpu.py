"""Randomised soak of the sharded state's known-zero logic WITHOUT a GPU: gloo worlds of 2 and 4 on the CPU test double
(which writes NaN wherever a masked pass would leave its output untouched), random benchmark-generator circuits,
batched and un-batched shards, virtual rank bits 0 / 1 / 2 -- shards and <Z0> with executor.CONFIG['zero_state'] on
against off and against the dense circuit.  usage: python tools/experiments/soak_sharded_cpu.py [first_seed] [count]"""
import os
import socket
import sys
import traceback

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def worker(rank, world, port, first, count, ret):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
        torch.set_num_threads(1)
        import bench
        import deepquantum_amd as dq
        from _cpu_backend import CpuTestBackend
        from deepquantum_amd import distributed as D
        from deepquantum_amd import executor

        dq.backend.set_test_backend(CpuTestBackend())
        dq.setup_distributed('gloo')
        executor.CONFIG['permute_min_bits'] = 12
        g = world.bit_length() - 1
        for seed in range(first, first + count):
            n = 14 + g + seed % 2
            batch = None if seed % 3 else 2
            vb = 0 if batch is not None else seed % 3
            depth = 6 + seed % 5
            dtype = torch.complex128 if seed % 4 == 1 else torch.complex64
            spec = bench.random_circuit_spec(n, depth, seed=seed)
            dense, data = bench.build_circuit(dq, n, spec, batch, dtype, 'cpu')
            with torch.no_grad():
                ref = dense(data).reshape(-1, 1 << n)
                ref_ev = dense.expectation().reshape(-1)
            per = (1 << n) // world
            D.CONFIG['virtual_bits'] = vb
            got = {}
            for on in (True, False):
                executor.CONFIG['zero_state'] = on
                for lazy in (False, True):
                    cir, _ = bench.build_circuit(dq, n, spec, batch, dtype, 'cpu', distributed=True)
                    cir.lazy_layout = lazy
                    with torch.no_grad():
                        st = cir(data)
                        ev = cir.expectation().reshape(-1)
                        stats = dict(D.LAST_RUN)
                        amps = st.amps.reshape(-1, per).clone()
                    tol = 1e-10 if dtype == torch.complex128 else 3e-5
                    err = (amps - ref[:, rank * per:(rank + 1) * per]).abs().max().item()
                    assert err < tol and not torch.isnan(amps.real).any(), (seed, on, lazy, err)
                    assert (ev - ref_ev).abs().max().item() < 10 * tol, (seed, on, lazy, ev, ref_ev)
                    got[(on, lazy)] = stats
            zs = got[(True, True)]
            if rank == 0:
                print(f'seed {seed}: world {world} n {n} batch {batch} virtual bits {vb} depth {depth} {str(dtype)[-9:]}: ok '
                      f'(remaps {zs["remaps"]}, virtual {zs["virtual_remaps"]}, known-zero stretches {zs["known_zero_stretches"]})', flush=True)
        D.CONFIG['virtual_bits'] = 0
        dq.cleanup_distributed()
        ret[rank] = 'ok'
    except Exception:  # noqa: BLE001
        ret[rank] = traceback.format_exc()


if __name__ == '__main__':
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    for world in (2, 4):
        mgr = mp.Manager()
        ret = mgr.dict()
        mp.spawn(worker, args=(world, _free_port(), first, count, ret), nprocs=world, join=True)
        for r in range(world):
            assert ret.get(r) == 'ok', f'world {world} rank {r}: {ret.get(r)}'
    print(f'{count} seeds from {first}, worlds 2 and 4: all agree')
