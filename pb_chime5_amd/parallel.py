"""Utterance-level work sharing over the GPUs of one node.

Replaces the reference's ``dlp_mpi`` usage (/root/reference/pb_chime5/core.py:363,
369,381: ``IS_MASTER``, ``barrier()``, ``split_managed()``).  Utterances are
independent, so there is no data-path collective and no RCCL: one process per GPU
(RANK / LOCAL_RANK / WORLD_SIZE from the launcher, e.g. torch.distributed.run),
each pulling utterance indices either

* dynamically from a shared counter (the analogue of dlp_mpi's master handing out indices
  on request): a few int64 slots in a file under /dev/shm when the ranks were started by
  `launch_local` (all node-local: no torch import, a rank is up in well under a second), or
  a key in the launcher's TCP store under torch.distributed.run (gloo), or
* statically (``index % world == rank`` over a longest-first ordering) when neither is
  reachable.

Single-process runs degrade to a plain loop (``allow_single_worker=True`` upstream).

``launch_local(n, argv)`` is the stand-in for ``mpiexec -np n`` (reference
README.md:108-111): it starts n copies of a command on this node, one per GPU, with
the rendezvous variables torch.distributed.run would set.

Every rank binds itself (and the loader / feeder / writer threads it starts later) to the
CPUs of its GPU's NUMA node (`bind_to_gpu_numa`): page-locked staging memory is then
allocated on the socket the GPU hangs off.
"""
import os
import socket
import subprocess
import sys

_STATE = {'store': None, 'epoch': 0, 'local': None, 'affinity': None}


class LocalGroup:
    """Counters and a barrier for the ranks of ONE node, in a file the launcher created
    (GSS_LOCAL_GROUP): SLOTS int64 values, mapped shared, changed under flock().  Slot 0 is
    the barrier's arrival counter (it only grows: the g-th barrier is passed once it reached
    g * world), slot e the hand-out counter of the e-th `split_managed` of the run."""
    SLOTS = 4096

    def __init__(self, path, world):
        import mmap
        import threading
        import numpy as np
        self.world = world
        self.path = path
        self.fd = os.open(path, os.O_RDWR)
        self._mm = mmap.mmap(self.fd, 8 * self.SLOTS)
        self._v = np.frombuffer(self._mm, dtype=np.int64)
        self._thread_lock = threading.Lock()       # flock() does not exclude threads of one fd
        self._generation = 0

    @classmethod
    def create(cls, directory=None):
        import tempfile
        if directory is None:
            directory = '/dev/shm' if os.path.isdir('/dev/shm') else None
        fd, path = tempfile.mkstemp(prefix='gss_group_', dir=directory)
        os.ftruncate(fd, 8 * cls.SLOTS)
        os.close(fd)
        return path

    def add(self, slot, n=1):
        import fcntl
        with self._thread_lock:
            fcntl.flock(self.fd, fcntl.LOCK_EX)
            try:
                value = int(self._v[slot]) + n
                self._v[slot] = value
            finally:
                fcntl.flock(self.fd, fcntl.LOCK_UN)
        return value

    def barrier(self, poll=0.0005, timeout=None):
        """Wait until every rank arrived.  Ranks of a session finish minutes apart, so the
        default limit is long (GSS_BARRIER_TIMEOUT_S, 3600 s); a launcher that died (this rank
        re-parented) or a limit that passed raises instead of spinning forever -- the launcher
        then terminates the remaining ranks."""
        import time
        if timeout is None:
            timeout = float(os.environ.get('GSS_BARRIER_TIMEOUT_S', 3600))
        self._generation += 1
        target = self._generation * self.world
        self.add(0, 1)
        parent = os.getppid()
        deadline = time.monotonic() + timeout
        spins = 0
        while int(self._v[0]) < target:
            time.sleep(poll)
            spins += 1
            if spins % 2000 == 0:
                if os.getppid() != parent:
                    raise RuntimeError('LocalGroup.barrier: the launcher is gone')
                if time.monotonic() > deadline:
                    raise RuntimeError(f'LocalGroup.barrier: {int(self._v[0])} of {target} arrivals '
                                       f'after {timeout:.0f} s (GSS_BARRIER_TIMEOUT_S)')

    def broadcast(self, obj, is_source):
        """Rank 0's picklable `obj` for every rank (dlp_mpi.bcast): a side file next to the
        counter file, published before a barrier, removed after a second one."""
        import pickle
        self._broadcasts = getattr(self, '_broadcasts', 0) + 1
        side = f'{self.path}.b{self._broadcasts}'
        if is_source:
            with open(side + '.tmp', 'wb') as fd:
                pickle.dump(obj, fd)
            os.replace(side + '.tmp', side)
        self.barrier()
        if not is_source:
            with open(side, 'rb') as fd:
                obj = pickle.load(fd)
        self.barrier()
        if is_source:
            os.unlink(side)
        return obj

    def close(self):
        self._v = None
        try:
            self._mm.close()
        except BufferError:
            pass
        os.close(self.fd)


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
    if world_size() == 1 or 'torch.distributed' not in sys.modules:
        return None          # (a process group nobody imported torch for cannot be up)
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return None
    return dist


def init(backend=None):
    """Join the process group of the launcher.  Only host-side rendezvous, barriers and
    the work counter use it, so gloo always suffices and is the default; nccl (= RCCL) on
    request (`backend='nccl'` / GSS_DIST_BACKEND=nccl).  On a node with fewer GPUs than ranks
    the ranks share devices (`_capi.pick_device`: distinct physical packages first).

    Side effect: the calling thread (and every thread it starts afterwards) is bound to the
    CPUs of its GPU's NUMA node (`bind_to_gpu_numa`; GSS_NUMA_BIND=0 leaves the affinity
    alone; skipped when ranks share GPUs).  Threads the HIP runtime started while the GPU's PCI
    address was looked up keep the full mask."""
    if world_size() > 1:
        # ROCr reads it when it initialises, i.e. at the first HIP call below (the host driver
        # only supports dmabuf IPC; the launchers export it, a hand-made environment may not)
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    bind_to_gpu_numa()       # (a single rank also wants its threads on its GPU's socket)
    if world_size() == 1:
        return None
    _warn_if_ranks_share_a_package()
    if backend is None and not os.environ.get('GSS_DIST_BACKEND') and _local_group() is not None:
        return None                       # node-local ranks of launch_local: no process group
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
            torch.cuda.set_device(device_index())
        dist.init_process_group(backend=backend)
    return dist


def _warn_if_ranks_share_a_package():
    """Rank 0 of a node says so once when its ranks outnumber the physical GPUs although every
    rank has a logical device of its own (a CPX / NPS-partitioned node: 64 logical devices on 8
    packages): the throughput of such a run is not that of LOCAL_WORLD_SIZE GPUs."""
    if local_rank() != 0:
        return
    try:
        from pb_chime5_amd import _capi
        local_world = int(os.environ.get('LOCAL_WORLD_SIZE', world_size()))
        bus_ids = [_capi.device_pci_bus_id(i) for i in range(_capi.device_count())]
        packages = len({_capi.pci_package(b) for b in bus_ids})
        if packages and packages < local_world <= len(bus_ids):
            print(f'pb_chime5_amd.parallel: {local_world} ranks on {packages} physical GPUs '
                  f'({len(bus_ids)} logical devices: a partitioned node) -- ranks share packages',
                  file=sys.stderr)
    except Exception:            # noqa: BLE001 -- a diagnostic, never an error
        pass


def _local_group():
    """The launcher's counter file, if this rank was started by `launch_local`.  A rank that
    cannot open the file it was given must fail (the launcher then stops the job): falling back
    to static sharding alone, beside ranks that share the counter, would process utterances
    twice and leave the others waiting in a barrier."""
    if _STATE['local'] is None:
        path = os.environ.get('GSS_LOCAL_GROUP')
        _STATE['local'] = LocalGroup(path, world_size()) if path and world_size() > 1 else False
    return _STATE['local'] or None


def barrier():
    dist = _dist()
    if dist is not None:
        dist.barrier()
    elif _local_group() is not None:
        _local_group().barrier()


def broadcast_object(obj, src=0):
    """The `src` rank's (picklable) object on every rank -- dlp_mpi.bcast in the reference."""
    if world_size() == 1:
        return obj
    dist = _dist()
    if dist is not None:
        box = [obj if rank() == src else None]
        dist.broadcast_object_list(box, src=src)
        return box[0]
    local = _local_group()
    if local is None:
        raise RuntimeError('broadcast_object: neither a process group nor a launcher file '
                           '(call parallel.init() first)')
    return local.broadcast(obj, rank() == src)


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
    """Yield the items of ``sequence`` this process should handle.  The generator ends without
    a barrier (a consumer that prefetches -- the session driver's loader threads -- would sit
    in it while it still holds prepared and enqueued utterances): callers synchronise after
    their own pipeline has drained (`Enhancer.enhance_session` ends with `parallel.barrier()`)."""
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
    local = _local_group() if dynamic and store is None and _dist() is None else None
    if store is None and local is None:
        for i in order[rank()::world]:
            yield items[i]
        return
    _STATE['epoch'] += 1
    if store is not None:
        key = f'pb_chime5_amd/next/{_STATE["epoch"]}'

        def take():
            return store.add(key, 1) - 1
    else:
        slot = _STATE['epoch']
        assert slot < LocalGroup.SLOTS, 'more split_managed() calls than counter slots'

        def take():
            return local.add(slot, 1) - 1
    while True:
        pos = take()
        if pos >= n:
            break
        yield items[order[pos]]


# ------------------------------------------------------------------ timed regions
class RankTimer:
    """The barriers and reductions a multi-rank measurement puts around its timed region
    (bench.py: `barrier(); t0; work; barrier(); wall = max over ranks`): ranks finish at
    different times -- the slowest rank's clock is the job's.  `dist` is the joined
    torch.distributed module (None: a single rank), `group` / `device` an optional second group
    for the reductions (an RCCL sub-group on 'cuda'); `sync` waits for this rank's GPU work."""

    def __init__(self, dist=None, group=None, device='cpu', sync=None):
        self.dist, self.group, self.device = dist, group, device
        self.sync = sync or (lambda: None)

    def barrier(self):
        self.sync()
        if self.dist is not None:
            if self.group is not None:
                self.dist.barrier(group=self.group)
            self.dist.barrier()
        self.sync()

    def _reduce(self, x, op):
        if self.dist is None:
            return float(x)
        import torch
        t = torch.tensor([x], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=getattr(self.dist.ReduceOp, op), group=self.group)
        return float(t.item())

    def max(self, x):
        return self._reduce(x, 'MAX')

    def sum(self, x):
        return self._reduce(x, 'SUM')

    def gather(self, values):
        """[values of rank 0, values of rank 1, ...] (a short list of floats per rank)."""
        if self.dist is None:
            return [list(map(float, values))]
        import torch
        t = torch.zeros((world_size(), len(values)), dtype=torch.float64, device=self.device)
        t[rank()] = torch.tensor(list(values), dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t.cpu().tolist()


def scaling_rows(value, n_gpus, n1_value=None):
    """What a SCALE record needs from an N-GPU line to describe itself: the per-GPU figure
    and, given the N = 1 figure of the same workload, the scaling efficiency."""
    rows = {'value_per_gpu': value / max(n_gpus, 1)}
    if n1_value:
        rows['n1_value'] = float(n1_value)
        rows['scaling_efficiency_vs_n1'] = value / (n_gpus * float(n1_value))
    return rows


# ------------------------------------------------------------------ CPU placement
def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(','):
        if not part:
            continue
        a, _, b = part.partition('-')
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def _format_cpulist(cpus):
    cpus = sorted(cpus)
    runs, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        runs.append(str(cpus[i]) if i == j else f'{cpus[i]}-{cpus[j]}')
        i = j + 1
    return ','.join(runs)


def gpu_numa_cpus(device, sysfs='/sys/bus/pci/devices'):
    """(NUMA node, CPUs local to it) of a GPU, from the PCI device's sysfs entry
    (`numa_node`, `local_cpulist`); (None, None) where the kernel does not say (one node,
    a VM without topology, no sysfs)."""
    from pb_chime5_amd import _capi
    try:
        bus_id = _capi.device_pci_bus_id(device).lower()
        base = os.path.join(sysfs, bus_id)
        with open(os.path.join(base, 'numa_node')) as fd:
            node = int(fd.read())
        with open(os.path.join(base, 'local_cpulist')) as fd:
            cpus = _parse_cpulist(fd.read())
    except Exception:            # noqa: BLE001 -- placement is an optimisation, never an error
        return None, None
    if node < 0 or not cpus:
        return None, None
    return node, cpus


def bind_to_gpu_numa(device=None, force=False):
    """Restrict the calling thread -- call it from the main thread before any worker thread
    exists: they inherit the mask -- to the CPUs of its GPU's NUMA node (intersected with the
    CPUs the process may use at all).  Eight ranks x (3 loaders + feeder + writer) threads
    floating over two sockets put page-locked staging blocks on the far socket: H2D / D2H
    through the inter-socket link.  GSS_NUMA_BIND=0 switches it off; so do more node-local
    ranks than GPUs (ranks sharing a GPU would all be squeezed onto its socket).  Only the
    calling thread and its future children move: threads the HIP runtime created during the
    PCI lookup keep their mask, and an application embedding the package gets its calling
    thread's affinity changed -- GSS_NUMA_BIND=0 is the switch for that.  Returns (and keeps
    for `affinity_info`) what was done; never raises."""
    if _STATE['affinity'] is not None and not force:
        return _STATE['affinity']
    info = {'bound': False, 'numa_node': None, 'cpus': None}
    try:
        allowed = os.sched_getaffinity(0)
        info['cpus'] = _format_cpulist(allowed)
        from pb_chime5_amd import _capi
        sharing = int(os.environ.get('LOCAL_WORLD_SIZE', world_size())) > max(_capi.device_count(), 1)
        info['ranks_share_gpus'] = sharing
        if os.environ.get('GSS_NUMA_BIND', '1') != '0' and not sharing:
            node, cpus = gpu_numa_cpus(device_index() if device is None else device)
            info['numa_node'] = node
            if cpus is not None and (allowed & cpus) and (allowed & cpus) != allowed:
                os.sched_setaffinity(0, allowed & cpus)
                info['bound'] = True
                info['cpus'] = _format_cpulist(allowed & cpus)
    except Exception as e:       # noqa: BLE001
        info['error'] = f'{type(e).__name__}: {e}'
    _STATE['affinity'] = info
    return info


def affinity_info():
    """What `bind_to_gpu_numa` did in this rank ({'bound', 'numa_node', 'cpus'})."""
    return _STATE['affinity'] or {'bound': False, 'numa_node': None, 'cpus': None}


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
    group_file = LocalGroup.create()
    try:
        for r in range(nprocs):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nprocs),
                       LOCAL_WORLD_SIZE=str(nprocs), MASTER_ADDR='127.0.0.1',
                       MASTER_PORT=str(port), GSS_LOCAL_GROUP=group_file)
            # the driver's environment has it; a hand-made one may not (dmabuf IPC only)
            env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
            env.update(extra_env or {})
            procs.append(subprocess.Popen(list(argv), env=env,
                                          stdout=None if r == 0 else sys.stderr))
        return _wait_all(procs, timeout)
    finally:
        import glob
        for path in glob.glob(glob.escape(group_file) + '*'):
            try:
                os.unlink(path)
            except OSError:
                pass


def _wait_all(procs, timeout):
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
