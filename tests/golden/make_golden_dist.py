"""Golden fixtures for the SHARDED path: runs the real reference's
DistributedQubitCircuit under 1, 2, 4 and 8 gloo ranks in the build container and stores per-rank shards,
expectation values and data gradients (adjoint differentiation).  Only inputs and outputs are stored.

usage: python tests/golden/make_golden_dist.py     (about a minute)
"""

import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, name, ret):
    import specs
    from make_golden import import_reference, to_np

    torch.set_num_threads(1)
    dq = import_reference()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    if world > 1:
        dq.setup_distributed('gloo')
    case = specs.DIST_CASES[name]
    cir = getattr(specs, case['builder'])(dq, dq.DistributedQubitCircuit, **case['kwargs'])
    data = None
    if case['data'] is not None:
        data = torch.tensor(case['data'], dtype=torch.float, requires_grad=True)
    state = cir(data=data)
    res = {'shard': to_np(state.amps)}
    ev = cir.expectation()
    res['expectation'] = to_np(ev)
    if data is not None:
        ev.sum().backward()
        res['grad'] = to_np(data.grad)
    ret[rank] = res
    if world > 1:
        dq.cleanup_distributed()


def main():
    import specs

    out = {}
    for name, case in specs.DIST_CASES.items():
        if case['data'] is not None:
            out[f'{name}/data'] = np.array(case['data'], dtype=np.float32)
        for world in case['worlds']:
            mgr = mp.Manager()
            ret = mgr.dict()
            mp.spawn(_worker, args=(world, _free_port(), name, ret), nprocs=world, join=True)
            out[f'{name}/W{world}/shards'] = np.stack([ret[r]['shard'] for r in range(world)])
            out[f'{name}/W{world}/expectation'] = ret[0]['expectation']
            for r in range(world):   # expectation / gradients are replicated on every rank
                assert np.allclose(ret[r]['expectation'], ret[0]['expectation'])
            if 'grad' in ret[0]:
                out[f'{name}/W{world}/grad'] = ret[0]['grad']
            print(name, world, out[f'{name}/W{world}/shards'].shape, ret[0]['expectation'][:4])
    np.savez_compressed(os.path.join(HERE, 'golden_dist.npz'), **out)
    print('wrote golden_dist.npz', os.path.getsize(os.path.join(HERE, 'golden_dist.npz')), 'bytes')


if __name__ == '__main__':
    main()
