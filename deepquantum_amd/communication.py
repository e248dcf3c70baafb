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


# ---- shard exchanges: one coalesced batch per group of samples, named, and watched -------------------------------
#: ``grouped_exchange``: the chunks of ALL samples of a group travel as one coalesced batch of point-to-point
#: operations (RCCL: one ncclGroupStart / End, one launch) instead of one ``all_to_all_single`` per sample;
#: ``watchdog_seconds``: an exchange that has not completed after this long is reported on stderr with what it is
#: (which remap, which peers, how many bytes) -- once per period, from a daemon thread, without blocking anybody;
#: 0 switches the watchdog off.  The process group's own timeout still aborts a dead job.
COMM_CONFIG = {'grouped_exchange': True,
               'watchdog_seconds': float(os.environ.get('DQ_COMM_WATCHDOG', '120'))}

#: collectives / coalesced batches issued since the last reset (bench, tests)
COMM_STATS = {'collectives': 0, 'p2p_ops': 0, 'staged': 0}


class Exchange:
    """A shard exchange in flight: its work handles and a description for the watchdog."""

    __slots__ = ('works', 'what', 'issued', 'reported', 'device')

    def __init__(self, works: list, what: str, device: torch.device | None = None) -> None:
        import time

        self.works = [w for w in works if w is not None]
        self.what = what
        self.issued = time.monotonic()
        self.reported = 0
        self.device = device

    def wait(self) -> None:
        """Order the exchange before what the current stream does next (RCCL: no host block; gloo: blocks)."""
        for w in self.works:
            w.wait()

    def completed(self) -> bool:
        return all(w.is_completed() for w in self.works)


_watch_lock = None
_watched: list = []
_watch_thread = None


def _watch(ex: Exchange) -> None:
    """Hand an exchange to the watchdog thread (started on first use)."""
    global _watch_lock, _watch_thread
    period = COMM_CONFIG['watchdog_seconds']
    if not period or period <= 0 or not ex.works:
        return
    import threading

    if _watch_lock is None:
        _watch_lock = threading.Lock()
    with _watch_lock:
        _watched.append(ex)
        if len(_watched) > 256:                      # (completed ones are dropped by the thread; a backstop)
            del _watched[:128]
    if _watch_thread is None or not _watch_thread.is_alive():
        _watch_thread = threading.Thread(target=_watchdog_loop, name='dq-comm-watchdog', daemon=True)
        _watch_thread.start()


def _watchdog_loop() -> None:
    import sys
    import time

    while True:
        period = COMM_CONFIG['watchdog_seconds']
        time.sleep(min(max(period / 4.0, 0.05), 5.0) if period and period > 0 else 1.0)
        if not period or period <= 0:
            continue
        now = time.monotonic()
        with _watch_lock:
            pending = list(_watched)
        done = []
        for ex in pending:
            try:
                if ex.device is not None and ex.device.type == 'cuda':
                    torch.cuda.set_device(ex.device)
                finished = ex.completed()
            except Exception:                         # noqa: BLE001  (a destroyed process group: forget the exchange)
                finished = True
            if finished:
                done.append(ex)
            elif now - ex.issued >= period * (ex.reported + 1):
                ex.reported += 1
                rank = dist.get_rank() if dist.is_initialized() else 0
                print(f'[deepquantum_amd watchdog] rank {rank}: {ex.what} has not completed after '
                      f'{now - ex.issued:.0f} s (a peer that never reached this exchange, or a link that is down)',
                      file=sys.stderr, flush=True)
        if done:
            with _watch_lock:
                for ex in done:
                    if ex in _watched:
                        _watched.remove(ex)


def exchange_chunks(recv: torch.Tensor, send: torch.Tensor, peers: list[int], chunk: int, what: str,
                    async_op: bool = False) -> Exchange | None:
    """The all-to-all of a k-qubit remap for a GROUP of samples.  ``send`` / ``recv``: (R, 2^k * chunk) real views of
    the rows of the group (interleaved complex); chunk c of every row goes to rank ``peers[c]`` and what that rank
    sends back lands in the same slot (``peers`` contains this rank once: that chunk is copied locally).

    RCCL / gloo on host memory: ONE coalesced batch of R * (2^k - 1) send / receive pairs (``batch_isend_irecv``: one
    group call, one launch on RCCL) -- every piece is contiguous where it lies, nothing is packed.  The reference
    issues one collective per swap gate and sample (communication.py:58-91); per sample it would be 80 collectives per
    step of the headline circuit here, now 20.  gloo with device memory (ranks sharing one GPU in the test-suite):
    staged through the host, one ``all_to_all_single`` per sample.  Returns the handle to wait on (``async_op``) or
    None when everything has completed."""
    world = dist.get_world_size()
    me = dist.get_rank()
    rows = send.shape[0]
    assert send.shape == recv.shape and send.shape[1] == len(peers) * chunk
    # every piece handed to a point-to-point operation is a slice [i, c * chunk : (c + 1) * chunk]: contiguous exactly
    # when the elements of a row are (rows themselves may lie apart: a group of samples is a slice of the batch)
    assert send.stride(1) == 1 and recv.stride(1) == 1, 'exchange_chunks: the rows of send / recv must be contiguous'
    staged = send.is_cuda and dist.get_backend() == 'gloo'
    if staged or not COMM_CONFIG['grouped_exchange']:
        splits = [0] * world
        for p_ in peers:
            splits[p_] = chunk
        works = []
        for i in range(rows):                           # one collective per sample: contiguous chunks
            works.append(all_to_all_flat(recv[i].reshape(-1), send[i].reshape(-1), splits, async_op=async_op and not staged))
            COMM_STATS['collectives'] += 1
            COMM_STATS['staged'] += int(staged)
        ex = Exchange(works, what, send.device)
        if not ex.works:
            return None
        _watch(ex)
        return ex
    ops = []
    for i in range(rows):
        for c, peer in enumerate(peers):
            piece_s, piece_r = send[i, c * chunk:(c + 1) * chunk], recv[i, c * chunk:(c + 1) * chunk]
            assert piece_s.is_contiguous() and piece_r.is_contiguous()
            if peer == me:
                piece_r.copy_(piece_s)
            else:
                ops.append(dist.P2POp(dist.isend, piece_s, peer))
                ops.append(dist.P2POp(dist.irecv, piece_r, peer))
    if not ops:
        return None
    works = dist.batch_isend_irecv(ops)
    COMM_STATS['collectives'] += 1
    COMM_STATS['p2p_ops'] += len(ops)
    ex = Exchange(list(works), what, send.device)
    _watch(ex)
    if not async_op:
        ex.wait()
        if not send.is_cuda:
            return None
    return ex


def exchange_pieces(recv: list[torch.Tensor], send: list[torch.Tensor], peers: list[int], what: str,
                    async_op: bool = False) -> Exchange | None:
    """One SLICE of a k-qubit remap (round 6; `distributed._remap` with the last pass in slices): piece c of ``send`` goes
    to rank ``peers[c]`` and what that rank sends back lands in piece c of ``recv`` -- real, contiguous tensors, one per
    member of the 2^k group (this rank's own piece is copied locally).  One coalesced batch of point-to-point operations
    (RCCL: one group call); gloo with device memory (ranks sharing a GPU in the test-suite): staged through the host.
    Returns the handle to wait on (``async_op``) or None when everything has completed."""
    me = dist.get_rank()
    assert len(send) == len(recv) == len(peers)
    staged = send[0].is_cuda and dist.get_backend() == 'gloo'
    ops, back = [], []
    for c, peer in enumerate(peers):
        ps, pr = send[c], recv[c]
        assert ps.is_contiguous() and pr.is_contiguous() and ps.shape == pr.shape
        if peer == me:
            pr.copy_(ps)
            continue
        if staged:
            hs, hr = ps.cpu(), torch.empty(pr.shape, dtype=pr.dtype)
            back.append((pr, hr))
            ps, pr = hs, hr
        ops.append(dist.P2POp(dist.isend, ps, peer))
        ops.append(dist.P2POp(dist.irecv, pr, peer))
    if not ops:
        return None
    works = dist.batch_isend_irecv(ops)
    COMM_STATS['collectives'] += 1
    COMM_STATS['p2p_ops'] += len(ops)
    COMM_STATS['staged'] += int(staged)
    ex = Exchange(list(works), what, send[0].device)
    _watch(ex)
    if staged or not async_op:
        ex.wait()
        for pr, hr in back:
            pr.copy_(hr)
        if staged or not send[0].is_cuda:
            return None
    return ex


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
