"""CHiME-6 track 2 entry point: segmentation and activity from RTTM files
(/root/reference/pb_chime5/scripts/kaldi_run_rttm.py).

    python -m pb_chime5_amd.scripts.kaldi_run_rttm with storage_dir=<...> database_rttm=<...> \
        [activity_rttm=<...>] chime6_dir=<...> session_id=dev job_id=1 number_of_jobs=1

``activity_rttm`` defaults to ``database_rttm``.  The argparse front end
``pb_chime5_amd.scripts.enhance_rttm`` drives the same code.
"""
import sys

from pb_chime5_amd.scripts import _cli
from pb_chime5_amd.scripts.kaldi_run import run
from pb_chime5_amd.scripts.run import EXTRA_KEYS

SCRIPT_KEYS = ('session_id', 'storage_dir', 'job_id', 'number_of_jobs')


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    from pb_chime5_amd.core_chime6_rttm import get_enhancer
    defaults = {**_cli.enhancer_defaults(get_enhancer, drop=EXTRA_KEYS),
                'session_id': 'dev', 'storage_dir': None, 'database_rttm': None,
                'activity_rttm': None, 'job_id': 1, 'number_of_jobs': 1}
    command, config, _ = _cli.parse(argv, defaults)
    if config['activity_rttm'] is None:
        config['activity_rttm'] = config['database_rttm']
    if command == 'print_config':
        _cli.print_config(config)
        return config
    assert config['database_rttm'] is not None, (
        config['database_rttm'], 'overwrite the database_rttm from the command line')
    return run(get_enhancer, config, test_run=command == 'test_run', script_keys=SCRIPT_KEYS)


if __name__ == '__main__':
    main()
