"""Kaldi-recipe entry point with a static job split
(/root/reference/pb_chime5/scripts/kaldi_run.py).

    python -m pb_chime5_amd.scripts.kaldi_run with storage_dir=<...> session_id=dev \
        job_id=1 number_of_jobs=1

Job ``job_id`` of ``number_of_jobs`` enhances examples ``job_id - 1, job_id - 1 +
number_of_jobs, ...`` of the session(s) (``dataset_slice = slice(job_id - 1, None,
number_of_jobs)``).  Output: ``storage_dir/audio/<dataset>/<example_id>.wav``.
"""
import sys
from pathlib import Path

from pb_chime5_amd import parallel
from pb_chime5_amd.scripts import _cli
from pb_chime5_amd.scripts.run import EXTRA_KEYS, enhancer_factory

SCRIPT_KEYS = ('chime6', 'session_id', 'storage_dir', 'job_id', 'number_of_jobs')


def run(get_enhancer, config, test_run, script_keys=SCRIPT_KEYS):
    _cli.print_config(config)
    storage_dir = config['storage_dir']
    assert storage_dir is not None, (storage_dir, 'overwrite the storage_dir from the command line')
    job_id, number_of_jobs = config['job_id'], config['number_of_jobs']
    assert job_id >= 1 and job_id <= number_of_jobs, (job_id, number_of_jobs)

    parallel.init()
    if parallel.is_master():
        _cli.new_run_dir(Path(storage_dir) / 'sacred', config)
    kwargs = {k: v for k, v in config.items() if k not in script_keys}
    enhancer = get_enhancer(**kwargs, device_id=parallel.device_index())
    if test_run:
        print('Database', enhancer.db)
        dataset_slice = True
    else:
        dataset_slice = slice(job_id - 1, None, number_of_jobs)
    if parallel.is_master():
        print('Enhancer:', enhancer)
        print(config['session_id'])
    enhancer.enhance_session(config['session_id'], Path(storage_dir) / 'audio',
                             dataset_slice=dataset_slice, audio_dir_exist_ok=True)
    if parallel.is_master():
        print('Finished experiment dir:', storage_dir)
    return Path(storage_dir)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    chime6 = any(a.replace(' ', '') in ('chime6=True', 'chime6=1') for a in argv)
    get_enhancer = enhancer_factory(chime6)
    defaults = {'chime6': False, **_cli.enhancer_defaults(get_enhancer, drop=EXTRA_KEYS),
                'session_id': 'dev', 'storage_dir': None, 'job_id': 1, 'number_of_jobs': 1}
    command, config, _ = _cli.parse(argv, defaults)
    if command == 'print_config':
        _cli.print_config(config)
        return config
    return run(get_enhancer, config, test_run=command == 'test_run')


if __name__ == '__main__':
    main()
