"""bench.py as a harness, without a GPU: the self-launch of ``--gpus N`` (a plain ``python bench.py --gpus 2``
silently measured ONE rank), the parity criterion and the sharded pin check, the rehearsal of one rank of a world that is
not there.  The ranks run on the CPU test double (installed by a wrapper that bench.py's launcher re-executes for every
rank through ``sys.orig_argv``); the pin of the small workload comes from the oracle."""

from __future__ import annotations

import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

WRAP = f"""
import sys, runpy
sys.path.insert(0, {os.path.join(ROOT, 'tests')!r}); sys.path.insert(0, {ROOT!r})
from _cpu_backend import CpuTestBackend
from deepquantum_amd import backend
backend.set_test_backend(CpuTestBackend())
sys.argv = ['bench.py'] + sys.argv[1:]
runpy.run_path({os.path.join(ROOT, 'bench.py')!r}, run_name='__main__')
"""


def run_bench(*args, env=None, double=True, timeout=600):
    cmd = [sys.executable, '-c', WRAP] if double else [sys.executable, os.path.join(ROOT, 'bench.py')]
    e = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    e.update(env or {})
    pr = subprocess.run(cmd + [str(a) for a in args], capture_output=True, text=True, env=e, timeout=timeout, cwd=ROOT)
    lines = [ln for ln in pr.stdout.splitlines() if ln.startswith('{')]
    return pr, [json.loads(ln) for ln in lines]


def make_oracle_pin(path, n, extra_cx=False):
    """The layout of tools/make_pins_large.py, from the oracle (sample 0 = the generator's angles)."""
    from oracle import statevec_oracle as oracle

    spec = oracle.random_circuit_spec(n, 40, 1234)
    if extra_cx:
        spec = spec + [('cnot', 0, n - 1), ('cnot', n - 1, 0)]
    with torch.no_grad():
        state = oracle.run_spec(n, spec, dtype=torch.complex64)[0]
    idx = torch.randint(0, 2**n, (4096,), generator=torch.Generator().manual_seed(n))
    p = state.real.double() ** 2 + state.imag.double() ** 2
    ez = [float((p.reshape(2**q, 2, -1)[:, 0].sum() - p.reshape(2**q, 2, -1)[:, 1].sum())) for q in range(n)]
    np.savez_compressed(os.path.join(path, f'pin_n{n}{"_cx" if extra_cx else ""}.npz'), nqubit=np.array(n), indices=idx.numpy(),
                        amplitudes=state[idx].numpy(), norm2=np.array(float(p.sum())), expectation_z=np.array(ez))


def test_gpus_2_without_torchrun_runs_two_ranks_and_checks_their_shards(tmp_path):
    make_oracle_pin(str(tmp_path), 13)
    pr, lines = run_bench('--gpus', 2, '--backend', 'gloo', '--nqubit', 12, '--batch', 2, '--steps', 1, '--warmup', 0,
                          '--no-cpu-baseline', env={'DQ_PIN_DIR': str(tmp_path)})
    assert pr.returncode == 0, pr.stderr[-2000:]
    assert len(lines) == 1, pr.stdout                        # rank 0 prints the ONE line
    line = lines[0]
    assert line['n_gpus'] == 2 and line['config']['nqubit'] == 13
    assert line['parity_checked'] is True, line.get('parity')
    par = line['parity']
    assert par['amplitudes_checked_per_rank_sum'] == 4096 and par['l2_error_relative'] < 1e-4
    assert line['config']['exchange_per_step']['remaps'] > 0
    # `value` is the drop-in step (the reference's shard order restored by every forward); the lazy layout is beside it
    assert line['config']['lazy_layout'] is False and line['value_lazy_layout'] > 0 and line['ms_per_step_lazy_layout'] > 0


def test_sharded_config_with_the_cx_pair_and_a_wrong_pin(tmp_path):
    make_oracle_pin(str(tmp_path), 14, extra_cx=True)
    pr, lines = run_bench('--gpus', 4, '--backend', 'gloo', '--config', 4, '--nqubit', 12, '--steps', 1, '--warmup', 0,
                          '--no-cpu-baseline', env={'DQ_PIN_DIR': str(tmp_path)})
    assert pr.returncode == 0, pr.stderr[-2000:]
    assert lines[0]['n_gpus'] == 4 and lines[0]['parity_checked'] is True, lines[0].get('parity')
    # a pin of ANOTHER circuit (without the cx pair) under the same name: the relative criterion must fail
    make_oracle_pin(str(tmp_path), 14, extra_cx=False)
    os.replace(os.path.join(tmp_path, 'pin_n14.npz'), os.path.join(tmp_path, 'pin_n14_cx.npz'))
    pr, lines = run_bench('--gpus', 4, '--backend', 'gloo', '--config', 4, '--nqubit', 12, '--steps', 1, '--warmup', 0,
                          '--no-cpu-baseline', env={'DQ_PIN_DIR': str(tmp_path)})
    assert pr.returncode == 0 and lines[0]['parity_checked'] is False


def test_rccl_job_on_a_box_without_the_gpus_measures_nothing():
    if torch.cuda.device_count() >= 2:
        pytest.skip('this box could run it')
    pr, lines = run_bench('--gpus', 2, '--steps', 1, '--warmup', 0, double=False)
    assert pr.returncode == 2 and not lines and 'nothing was measured' in pr.stderr


def test_parity_criterion_is_relative():
    import bench

    rng = np.random.default_rng(0)
    ref = (rng.normal(size=4096) + 1j * rng.normal(size=4096)) * 2e-5          # amplitudes of an n = 31 state
    ez = np.zeros(31)
    ok, rep = bench.pin_verdict(ref * (1 + 1e-5), ref, 1.0, 1.0, ez, ez)
    assert ok and rep['l2_error_relative'] < 2e-5
    assert not bench.pin_verdict(np.zeros_like(ref), ref, 1.0, 1.0, ez, ez)[0]        # (passed the absolute 1e-4 of round 4)
    assert not bench.pin_verdict(np.roll(ref, 1), ref, 1.0, 1.0, ez, ez)[0]
    assert not bench.pin_verdict(ref, ref, 0.9, 1.0, ez, ez)[0]


def test_rehearsal_of_one_rank_of_a_world_that_is_not_there():
    for r in (0, 1):
        pr, lines = run_bench('--gpus', 4, '--rehearse-rank', r, '--strong', '--nqubit', 14, '--virtual-bits', 1, '--steps', 1,
                              '--warmup', 0, '--no-cpu-baseline')
        assert pr.returncode == 0, pr.stderr[-2000:]
        line = lines[0]
        assert line['rank'] == r and line['world'] == 4 and line['virtual_rank_bits'] == 1
        assert line['schedule']['remaps'] > line['schedule']['virtual_remaps'] > 0
        # rank 0 starts from |0..0>, the others from zeros: no pass before the first exchange
        assert (line['schedule']['zero_shard_stretches'] > 0) == (r != 0)


def test_transport_selector_of_the_sharded_gpu_tests():
    """``_helpers.pick_transport``: RCCL (one device per rank) as soon as the box shows enough devices, gloo with every
    rank on device 0 otherwise; a world of one never needs it; DQ_TEST_TRANSPORT=gloo pins the fallback."""
    import os

    from _helpers import pick_transport

    assert pick_transport(2, device_count=1) == ('gloo', [0, 0])
    assert pick_transport(4, device_count=2) == ('gloo', [0, 0, 0, 0])
    assert pick_transport(2, device_count=2) == ('nccl', [0, 1])
    assert pick_transport(4, device_count=8) == ('nccl', [0, 1, 2, 3])
    assert pick_transport(8, device_count=8) == ('nccl', list(range(8)))
    assert pick_transport(1, device_count=8) == ('gloo', [0])
    assert pick_transport(2, device_count=0) == ('gloo', [0, 0])
    os.environ['DQ_TEST_TRANSPORT'] = 'gloo'
    try:
        assert pick_transport(2, device_count=8) == ('gloo', [0, 0])
    finally:
        del os.environ['DQ_TEST_TRANSPORT']
    assert pick_transport(2)[0] in ('gloo', 'nccl')        # (whatever this box shows)
