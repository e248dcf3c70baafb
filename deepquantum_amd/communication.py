"""Process-group plumbing for the sharded state: one process per GPU, ``torch.distributed`` with the
``'nccl'`` backend (= RCCL over xGMI on ROCm) or ``'gloo'`` for CPU tests.  Mirrors the helper set of
the reference's communication.py:9-91."""

from __future__ import annotations

import os

import torch
import torch.distributed as dist


def setup_distributed(backend: str = 'nccl', port: str = '29500') -> tuple[int, int, int]:
    """Initialise the default process group from the torchrun environment (RANK, WORLD_SIZE, LOCAL_RANK;
    MASTER_ADDR/MASTER_PORT) and pin this process to its GPU.  Returns (rank, world_size, local_rank)."""
    try:
        rank = int(os.environ['RANK'])
        world_size = int(os.environ['WORLD_SIZE'])
        local_rank = int(os.environ['LOCAL_RANK'])
    except KeyError:
        rank, world_size, local_rank = 0, 1, 0
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', port)
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
    if backend == 'nccl':
        torch.cuda.set_device(local_rank)
    if not dist.is_initialized():
        kwargs = {}
        if backend == 'nccl':
            kwargs['device_id'] = torch.device('cuda', local_rank)
        dist.init_process_group(backend, world_size=world_size, rank=rank, **kwargs)
    return rank, world_size, local_rank


def cleanup_distributed() -> None:
    if dist.is_initialized():
        dist.destroy_process_group()


def comm_get_rank() -> int:
    return dist.get_rank() if dist.is_initialized() else 0


def comm_get_world_size() -> int:
    return dist.get_world_size() if dist.is_initialized() else 1


def all_to_all_flat(recv: torch.Tensor, send: torch.Tensor, splits: list[int], async_op: bool = False):
    """``all_to_all_single`` on flat real buffers with the same split table both ways.  RCCL moves GPU
    memory directly (``async_op``: returns the work handle; ``wait()`` orders it before the current stream's next
    kernel without blocking the host); under the ``gloo`` backend (CPU tests, or several ranks sharing one GPU in the
    GPU test-suite) device buffers are staged through host memory, synchronously."""
    if send.is_cuda and dist.get_backend() == 'gloo':
        h_send = send.cpu()
        h_recv = torch.empty_like(h_send)
        dist.all_to_all_single(h_recv, h_send, output_split_sizes=splits, input_split_sizes=splits)
        recv.copy_(h_recv)
        return None
    if async_op and send.is_cuda:
        return dist.all_to_all_single(recv, send, output_split_sizes=splits, input_split_sizes=splits, async_op=True)
    dist.all_to_all_single(recv, send, output_split_sizes=splits, input_split_sizes=splits)
    return None


def comm_exchange_arrays(send_data: torch.Tensor, recv_data: torch.Tensor, pair_rank: int | None) -> None:
    """Pairwise exchange with ``pair_rank``.  Every rank of the group must call it for every exchange
    step (ranks with nothing to move pass ``pair_rank=None``): it is expressed as one
    ``all_to_all_single`` with a single non-zero split, which RCCL turns into one send/recv pair over
    the xGMI link to the partner (reference: communication.py:58-91)."""
    world_size = comm_get_world_size()
    if not dist.is_initialized() or world_size <= 1:
        return
    active = pair_rank is not None and 0 <= pair_rank < world_size
    splits = [0] * world_size
    if active:
        assert send_data.shape == recv_data.shape and send_data.dtype == recv_data.dtype
        assert send_data.is_contiguous() and recv_data.is_contiguous(), 'exchange buffers must be contiguous'
        # complex amplitudes travel as interleaved reals (RCCL has no complex dtype)
        send_flat = torch.view_as_real(send_data).reshape(-1) if send_data.is_complex() else send_data.reshape(-1)
        recv_flat = torch.view_as_real(recv_data).reshape(-1) if recv_data.is_complex() else recv_data.reshape(-1)
        splits[pair_rank] = send_flat.numel()
    else:
        real = send_data.real.dtype if send_data.is_complex() else send_data.dtype
        send_flat = torch.empty(0, dtype=real, device=send_data.device)
        recv_flat = torch.empty(0, dtype=real, device=send_data.device)
    all_to_all_flat(recv_flat, send_flat, splits)
