"""Round 6 soak of the sharded state WITHOUT a GPU: gloo worlds of 2 and 4 on the CPU test double (NaN wherever a masked pass
leaves its output untouched AND wherever a lazy reset() did not clear), random benchmark-generator circuits deep enough for
several exchanges, un-batched and batched shards, the round's switches drawn per seed -- slice_exchange 0 .. 3,
first_exchange_local, evict_foldable, defer_tail, virtual bits -- two forwards each (the second starts from dirty buffers), shards and <Z0>
against the dense circuit.  usage: python tools/experiments/r06_soak_sharded_cpu.py [first_seed] [count]
DQ_SOAK_DEVICE=cuda: the same with the REAL kernels, the ranks sharing the GPU (gloo, host-staged exchanges), states 3 qubits bigger."""
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
        import random

        import bench
        import deepquantum_amd as dq
        from _cpu_backend import CpuTestBackend
        from deepquantum_amd import distributed as D
        from deepquantum_amd import executor

        device = os.environ.get('DQ_SOAK_DEVICE', 'cpu')
        if device == 'cpu':
            dq.backend.set_test_backend(CpuTestBackend())
        else:
            torch.cuda.set_device(0)
        dq.DistributedQubitState.POISON_LAZY = True
        dq.setup_distributed('gloo')
        executor.CONFIG['permute_min_bits'] = 11
        g = world.bit_length() - 1
        sliced = 0
        for seed in range(first, first + count):
            rng = random.Random(seed)
            dtype = torch.complex128 if rng.random() < 0.35 else torch.complex64
            batch = 2 if rng.random() < 0.25 else None
            n = (15 if dtype == torch.complex128 else 16) + g + rng.randrange(2) + (3 if device != 'cpu' else 0)
            depth = rng.choice((8, 14, 22, 30))
            nb = rng.choice((0, 1, 2, 3))
            cfg = {'slice_exchange': nb, 'first_exchange_local': rng.random() < 0.8, 'evict_foldable': rng.choice((None, True, False)),
                   'virtual_bits': 0 if (nb or batch is not None) else rng.choice((0, 0, 1, 2)), 'defer_tail': rng.choice((0, 12, 12, 40))}
            spec = bench.random_circuit_spec(n, depth, seed=1000 + seed)
            dense, data = bench.build_circuit(dq, n, spec, batch, dtype, device)
            with torch.no_grad():
                ref = dense(data).reshape(-1, 1 << n)
                ref_ev = dense.expectation().reshape(-1)
            per = (1 << n) // world
            D.CONFIG.update(cfg)
            tol = 1e-10 if dtype == torch.complex128 else 4e-5
            for lazy in (True, False):
                cir, _ = bench.build_circuit(dq, n, spec, batch, dtype, device, distributed=True)
                cir.lazy_layout = lazy
                with torch.no_grad():
                    for rep in range(2):
                        st = cir(data)
                        stats = dict(D.LAST_RUN)
                        ev = cir.expectation().reshape(-1)
                        amps = st.amps.reshape(-1, per).clone()
                        err = (amps - ref[:, rank * per:(rank + 1) * per].to(amps.device)).abs().max().item()
                        assert err < tol and not torch.isnan(amps.real).any(), (seed, cfg, lazy, rep, err, stats)
                        assert (ev - ref_ev.to(ev.device)).abs().max().item() < 10 * tol, (seed, cfg, lazy, rep, ev, ref_ev)
            sliced += stats['sliced_remaps']
            if rank == 0:
                print(f'seed {seed}: world {world} n {n} batch {batch} depth {depth} {str(dtype)[-9:]} {cfg}: ok (remaps {stats["remaps"]}, sliced '
                      f'{stats["sliced_remaps"]}: launches {stats["slice_launches_last"]} / {stats["slice_launches_first"]}, local first '
                      f'{stats["local_first_exchanges"]}, zero fills {stats["zero_fills"]}, deferred tails {stats["deferred_tails"]} ({stats["deferred_gates"]} gates))', flush=True)
        D.CONFIG.update({'slice_exchange': None, 'first_exchange_local': True, 'evict_foldable': None, 'virtual_bits': None, 'defer_tail': 12})
        dq.cleanup_distributed()
        ret[rank] = 'ok' if sliced or count < 4 else 'no remap was ever sliced'
    except Exception:  # noqa: BLE001
        ret[rank] = traceback.format_exc()


if __name__ == '__main__':
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    bad = 0
    for world in (2, 4):
        mgr = mp.Manager()
        ret = mgr.dict()
        mp.spawn(worker, args=(world, _free_port(), first, count, ret), nprocs=world, join=True)
        for r in range(world):
            if ret.get(r) != 'ok':
                bad += 1
                print(f'world {world} rank {r}: {ret.get(r)}')
    print('soak', 'FAILED' if bad else 'passed', f'(seeds {first} .. {first + count - 1}, worlds 2 and 4)')
    sys.exit(1 if bad else 0)
