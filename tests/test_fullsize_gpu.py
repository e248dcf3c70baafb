"""BASELINE configs 4 and 5 at their stated sizes on the ONE GPU of the test box: the ranks are gloo
processes that share the GPU (host-staged exchanges: functional, not a performance figure), every rank checks the pinned
amplitudes of ITS shard against the single-GPU pins of tests/golden/pin_n*.npz (tools/make_pins_large.py: the plain
one-GPU route of the HIP path, validated there against the real reference's n = 28 pin) with the relative criterion of
``bench.pin_verdict``; the QAOA ring's gradient through the sharded adjoint sweep against its closed form.  What stays
untested after this: the RCCL transport between GPUs.  Reference: distributed.py:57-202, README.md:223-255."""

from __future__ import annotations

import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(*args, need_gib, timeout=900):
    import gc

    gc.collect()
    torch.cuda.empty_cache()        # (what this pytest process still caches from earlier tests is not free for the ranks)
    torch.cuda.synchronize()
    free, _total = torch.cuda.mem_get_info()
    if free < need_gib * 2**30:
        pytest.skip(f'needs {need_gib} GiB of free device memory')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from _helpers import pick_transport

    world = int(args[args.index('--gpus') + 1])
    transport, _devices = pick_transport(world)       # RCCL, one device per rank, when the box shows enough devices
    pr = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--backend', transport, '--steps', '1', '--warmup', '0',
                         '--functional', '--no-cpu-baseline'] + [str(a) for a in args],
                        capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)
    assert pr.returncode == 0, pr.stderr[-3000:]
    lines = [json.loads(ln) for ln in pr.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    return lines[0]


def _check(line, n, world):
    assert line['n_gpus'] == world and line['config']['nqubit'] == n
    par = line['parity']
    assert line['parity_checked'] is True, par
    assert par['amplitudes_checked_per_rank_sum'] == 4096
    assert par['max_amplitude_error_relative_to_largest_amplitude'] < 1e-4 and par['l2_error_relative'] < 1e-4
    assert par['max_expectation_z_error'] < 1e-4 and abs(par['norm2'] - par['norm2_reference']) < 1e-4
    assert abs(line['config']['norm2_sample0'] - par['norm2_reference']) < 1e-4


@pytest.mark.parametrize('vbits', [2, 0])
def test_config4_n32_on_four_ranks(vbits):
    """QubitCircuit(32): generator circuit + cx(0, 31) (global control) + cx(31, 0) (GLOBAL target), 30 local qubits per
    rank, 'remap' mode with and without virtual rank bits: 4 x (8 + 8) GiB."""
    line = _bench('--gpus', 4, '--config', 4, '--virtual-bits', vbits, need_gib=80)
    _check(line, 32, 4)
    assert line['config']['virtual_rank_bits'] == vbits
    assert line['config']['exchange_per_step']['remaps'] >= 3


def test_config5_n33_on_eight_ranks_and_the_qaoa_gradient():
    """Config 5 at the largest size eight ranks sharing one 288-GB GPU can hold: the forward at n = 33 (8 x 16 GiB), the
    QAOA ring (hlayer, cnot . rz . cnot per ring edge, rx layer, one <Z_i Z_j> per edge; examples/qaoa.py:21-64) with
    its gradient through the fused sharded reverse sweep at n = 31 (the adjoint holds eight shard-sized buffers per
    rank) against the closed form n sin(4 beta) sin(4 gamma) / 2."""
    line = _bench('--gpus', 8, '--config', 5, '--nqubit', 30, '--qaoa-nqubit', 31, need_gib=170, timeout=1500)
    _check(line, 33, 8)
    q = line['config']['qaoa_ring']
    assert q['nqubit'] == 31 and q['fused_reverse_sweep'] and q['matches_closed_form'], q
    assert q['max_relative_error_vs_closed_form'] < 1e-4
