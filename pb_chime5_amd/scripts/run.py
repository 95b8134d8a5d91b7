"""Enhance whole sessions of the CHiME-5 JSON database
(/root/reference/pb_chime5/scripts/run.py).

    python -m pb_chime5_amd.scripts.run with session_id=dev wpe=True multiarray
    python -m pb_chime5_amd.scripts.run test_run with session_id=S02 database_path=cache/chime5.json
    # all GPUs of a node: utterances are shared out dynamically, no data-path collective
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        -m pb_chime5_amd.scripts.run with session_id=dev

Config keys: every keyword of ``get_enhancer`` plus ``session_id`` (session ids and/or
dataset names 'train' / 'dev' / 'eval' / 'all') and ``chime6``; named config
``multiarray`` (= ``multiarray=True bf_drop_context=True``).  Output:
``<file_storage>/<run id>/audio/<dataset>/<example_id>.wav`` (``-F``, default ./sacred).
"""
import itertools
import sys
from pathlib import Path

from pb_chime5_amd import mapping, parallel
from pb_chime5_amd.scripts import _cli

NAMED_CONFIGS = {'multiarray': {'bf_drop_context': True, 'multiarray': True}}
EXTRA_KEYS = ('activity_store', 'iterator_factory', 'device_id')


def enhancer_factory(chime6):
    if chime6:
        from pb_chime5_amd.core_chime6 import get_enhancer
    else:
        from pb_chime5_amd.core import get_enhancer
    return get_enhancer


def get_session_ids(session_id):
    """Dataset names expand to their sessions; the result is sorted and unique
    (run.py:46-72)."""
    if isinstance(session_id, str):
        session_id = [session_id]
    ordered = sorted(mapping.session_to_dataset.items(), key=lambda x: (x[1], x[0]))
    dataset_to_session = {dataset: [s for s, _ in group]
                          for dataset, group in itertools.groupby(ordered, key=lambda x: x[1])}
    dataset_to_session['all'] = [s for v in dataset_to_session.values() for s in v]
    return sorted({s for key in session_id for s in dataset_to_session.get(key, [key])})


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    chime6 = any(a.replace(' ', '') in ('chime6=True', 'chime6=1') for a in argv)
    get_enhancer = enhancer_factory(chime6)
    defaults = {'chime6': False, **_cli.enhancer_defaults(get_enhancer, drop=EXTRA_KEYS),
                'session_id': 'dev'}
    command, config, file_storage = _cli.parse(argv, defaults, named_configs=NAMED_CONFIGS)
    if command == 'print_config':
        _cli.print_config(config)
        return config
    test_run = command == 'test_run'

    parallel.init()
    run_dir = None
    if parallel.is_master():
        _cli.print_config(config)
        run_dir = _cli.new_run_dir(file_storage or 'sacred', config)
        print('Experiment dir:', run_dir)
    run_dir = _cli.broadcast_path(run_dir)

    kwargs = {k: v for k, v in config.items() if k not in ('chime6', 'session_id')}
    enhancer = get_enhancer(**kwargs, device_id=parallel.device_index())
    if test_run:
        print('Database', enhancer.db)
    session_ids = get_session_ids(config['session_id'])
    if parallel.is_master():
        print('Enhancer:', enhancer)
        print(session_ids)
    enhancer.enhance_session(session_ids, Path(run_dir) / 'audio', dataset_slice=test_run,
                             audio_dir_exist_ok=True)
    if parallel.is_master():
        print('Finished experiment dir:', run_dir)
    return run_dir


if __name__ == '__main__':
    main()
