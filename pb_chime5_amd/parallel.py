"""Utterance-level work sharing over the GPUs of one node.

Replaces the reference's ``dlp_mpi`` usage (/root/reference/pb_chime5/core.py:363,
369,381: ``IS_MASTER``, ``barrier()``, ``split_managed()``).  Utterances are
independent, so there is no data-path collective and no RCCL: one process per GPU
(RANK / LOCAL_RANK / WORLD_SIZE from the launcher, e.g. torch.distributed.run),
each pulling utterance indices either

* dynamically from a shared counter kept in the launcher's TCP store (the analogue
  of dlp_mpi's master handing out indices on request), or
* statically (``index % world == rank`` over a longest-first ordering) when no
  store is reachable.

Single-process runs degrade to a plain loop (``allow_single_worker=True`` upstream).
"""
import os

_STATE = {'store': None, 'epoch': 0}


def rank():
    return int(os.environ.get('RANK', 0))


def world_size():
    return int(os.environ.get('WORLD_SIZE', 1))


def local_rank():
    return int(os.environ.get('LOCAL_RANK', 0))


def is_master():
    return rank() == 0


def _dist():
    if world_size() == 1:
        return None
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return None
    return dist


def init(backend=None):
    """Join the process group of the launcher (gloo on CPU-only hosts, nccl = RCCL
    otherwise).  Only host-side rendezvous, barrier and the work counter use it."""
    if world_size() == 1:
        return None
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local_rank())
        dist.init_process_group(backend=backend)
    return dist


def barrier():
    dist = _dist()
    if dist is not None:
        dist.barrier()


def _store():
    dist = _dist()
    if dist is None:
        return None
    if _STATE['store'] is None:
        try:
            _STATE['store'] = dist.distributed_c10d._get_default_store()
        except Exception:
            _STATE['store'] = False
    return _STATE['store'] or None


def shard_indices(num_items, costs=None, rank_=None, world=None):
    """Static shard: longest-first ordering dealt round-robin."""
    rank_ = rank() if rank_ is None else rank_
    world = world_size() if world is None else world
    order = list(range(num_items))
    if costs is not None:
        order.sort(key=lambda i: (-costs[i], i))
    return order[rank_::world]


def split_managed(sequence, costs=None, dynamic=True):
    """Yield the items of ``sequence`` this process should handle."""
    items = list(sequence) if not hasattr(sequence, '__getitem__') else sequence
    n = len(items)
    world = world_size()
    if world == 1:
        for i in range(n):
            yield items[i]
        return
    order = list(range(n))
    if costs is not None:
        order.sort(key=lambda i: (-costs[i], i))
    store = _store() if dynamic else None
    if store is None:
        for i in order[rank()::world]:
            yield items[i]
        return
    _STATE['epoch'] += 1
    key = f'pb_chime5_amd/next/{_STATE["epoch"]}'
    while True:
        pos = store.add(key, 1) - 1
        if pos >= n:
            break
        yield items[order[pos]]
    barrier()
