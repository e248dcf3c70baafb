"""Every pass of a fused reverse sweep on the GPU against the CPU emulator of the record stream (tests/_wave_emulator.py)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import deepquantum_amd as dq
from deepquantum_amd import backend
import _wave_emulator as emu

real = backend.apply_fused


def checked(state, mats, mat_batch_stride, desc, out=None, grads=None):
    n = state.shape[-1].bit_length() - 1
    wave = desc.m - desc.slots == 6
    if wave:
        g_e = None if grads is None else grads.cpu().numpy().copy()
        g_0 = None if grads is None else grads.cpu().numpy().copy()
        e = emu.run_pass(desc, n, state.detach().cpu().numpy().copy(), mats.detach().cpu().numpy(), mat_batch_stride, grads=g_e)
    r = real(state, mats, mat_batch_stride, desc, out=out, grads=grads)
    if wave:
        torch.cuda.synchronize()
        err = np.abs(e - r.detach().cpu().numpy()).max()
        kp = emu.descriptor(desc, n)
        ids = [kp.rec[i][0] for i in range(kp.nrec_bytes // 32)]
        gerr = 0 if grads is None else np.abs((g_e - g_0) - (grads.cpu().numpy() - g_0)).max()
        g = emu.gen()
        print(f'pass n={n} records {len(ids)} state err {err:.2e} grad err {gerr:.2e} gen2 {sum(i >= g.ID_GEN2 for i in ids)} '
              f'grad {sum(g.ID_GRAD <= i < g.ID_EXPZ for i in ids)} ids {ids if err > 1e-3 else ""}', flush=True)
    return r


backend.apply_fused = checked
dq.executor.backend.apply_fused = checked
from _helpers import check_fused_sweep_with_sloppy_user_matrices
try:
    check_fused_sweep_with_sloppy_user_matrices(dq, device=torch.device('cuda', 0), n=12)
    print('agree')
except AssertionError as ex:
    print('FAILED', ex)
