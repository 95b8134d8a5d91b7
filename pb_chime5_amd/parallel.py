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

``launch_local(n, argv)`` is the stand-in for ``mpiexec -np n`` (reference
README.md:108-111): it starts n copies of a command on this node, one per GPU, with
the rendezvous variables torch.distributed.run would set.
"""
import os
import socket
import subprocess
import sys

_STATE = {'store': None, 'epoch': 0}


def rank():
    return int(os.environ.get('RANK', 0))


def world_size():
    return int(os.environ.get('WORLD_SIZE', 1))


def local_rank():
    return int(os.environ.get('LOCAL_RANK', 0))


def device_index():
    """The GPU of this rank: LOCAL_RANK modulo the number of visible GPUs (ranks share
    devices on a node with fewer GPUs than ranks)."""
    from pb_chime5_amd import _capi
    return _capi.default_device()


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
    """Join the process group of the launcher.  Only host-side rendezvous, barriers and
    the work counter use it, so gloo always suffices and is the default; nccl (= RCCL) on
    request (`backend='nccl'` / GSS_DIST_BACKEND=nccl).  On a node with fewer GPUs than ranks
    the ranks share devices (device = LOCAL_RANK % device count)."""
    if world_size() == 1:
        return None
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        if backend is None:
            backend = os.environ.get('GSS_DIST_BACKEND')
        if backend is None:
            # gloo: rendezvous, barriers and the work counter are host-side; a session must
            # not depend on RCCL coming up on the node (GSS_DIST_BACKEND=nccl selects it)
            backend = 'gloo'
        if backend == 'nccl':
            os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
            torch.cuda.set_device(device_index())
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


def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def launch_local(nprocs, argv, extra_env=None, timeout=None):
    """Run ``argv`` as ``nprocs`` ranks on this node (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR=127.0.0.1 / MASTER_PORT in their environment) and wait for all of
    them.  Rank 0 inherits this process's stdout; the other ranks' stdout goes to
    stderr, so a command that prints one result line on rank 0 still prints exactly
    one.  Returns the first non-zero exit status (the remaining ranks are terminated
    by PID), else 0."""
    port = free_port()
    procs = []
    for r in range(nprocs):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nprocs),
                   LOCAL_WORLD_SIZE=str(nprocs), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.update(extra_env or {})
        procs.append(subprocess.Popen(list(argv), env=env,
                                      stdout=None if r == 0 else sys.stderr))
    import time
    deadline = None if timeout is None else time.monotonic() + timeout
    status = 0
    pending = list(procs)
    while pending:
        for p in list(pending):
            rc = p.poll()
            if rc is None:
                continue
            pending.remove(p)
            if rc != 0 and status == 0:
                status = rc
        if status != 0 or (deadline is not None and time.monotonic() > deadline):
            for p in pending:
                p.terminate()
            for p in pending:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
            return status or 124
        time.sleep(0.05)
    return status


def _main(argv=None):
    """``python -m pb_chime5_amd.parallel -n N <python args...>``: the node-local
    ``mpiexec -np N python <python args...>``."""
    argv = sys.argv[1:] if argv is None else list(argv)
    if len(argv) < 3 or argv[0] not in ('-n', '-np'):
        raise SystemExit('usage: python -m pb_chime5_amd.parallel -n N [-m module | script.py] args...')
    return launch_local(int(argv[1]), [sys.executable] + argv[2:])


if __name__ == '__main__':
    sys.exit(_main())
