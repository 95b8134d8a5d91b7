"""The multi-GPU path is utterance sharding with no data-path collective
(pb_chime5_amd/parallel.py replaces dlp_mpi's split_managed / barrier / IS_MASTER).
Covered here with world_size-2 gloo process groups on CPU."""
import os
import socket
import subprocess
import sys
import textwrap

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
