"""Command line front end of the RTTM-driven path: the knobs of the reference's
``scripts/kaldi_run_rttm.py`` (:21-40, 61-89) without sacred.

    python -m pb_chime5_amd.scripts.enhance_rttm --chime6-dir CHiME6 \
        --database-rttm dev_rttm --activity-rttm dev_rttm --session-id S02 --out out

    # all GPUs of a node (utterances are sharded, no collective on the data path):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        -m pb_chime5_amd.scripts.enhance_rttm ...

``--job-id / --number-of-jobs`` give the Kaldi-style static split
(``dataset_slice = slice(job_id - 1, None, number_of_jobs)``, kaldi_run_rttm.py:73).
"""
import argparse
from pathlib import Path


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawTextHelpFormatter)
    ap.add_argument('--chime6-dir', required=True)
    ap.add_argument('--database-rttm', required=True, nargs='+')
    ap.add_argument('--activity-rttm', nargs='+', default=None)
    ap.add_argument('--session-id', nargs='+', default=['dev'])
    ap.add_argument('--out', required=True)
    ap.add_argument('--multiarray', default='outer_array_mics')
    ap.add_argument('--context-samples', type=int, default=240000)
    ap.add_argument('--no-wpe', action='store_true')
    ap.add_argument('--wpe-tabs', type=int, default=10)
    ap.add_argument('--wpe-delay', type=int, default=2)
    ap.add_argument('--wpe-iterations', type=int, default=3)
    ap.add_argument('--bss-iterations', type=int, default=20)
    ap.add_argument('--bss-iterations-post', type=int, default=1)
    ap.add_argument('--bf', default='mvdrSouden_ban')
    ap.add_argument('--postfilter', default=None)
    ap.add_argument('--no-bf-drop-context', action='store_true')
    ap.add_argument('--job-id', type=int, default=1)
    ap.add_argument('--number-of-jobs', type=int, default=1)
    ap.add_argument('--test-run', action='store_true', help='first two examples only')
    args = ap.parse_args(argv)

    from pb_chime5_amd import parallel
    from pb_chime5_amd.core_chime6_rttm import get_enhancer
    parallel.init()
    multiarray = True if args.multiarray in ('True', 'true', '1') else args.multiarray
    enhancer = get_enhancer(
        database_rttm=args.database_rttm, activity_rttm=args.activity_rttm or args.database_rttm,
        chime6_dir=args.chime6_dir, multiarray=multiarray,
        context_samples=args.context_samples, wpe=not args.no_wpe, wpe_tabs=args.wpe_tabs,
        wpe_delay=args.wpe_delay, wpe_iterations=args.wpe_iterations,
        bss_iterations=args.bss_iterations, bss_iterations_post=args.bss_iterations_post,
        bf_drop_context=not args.no_bf_drop_context, bf=args.bf, postfilter=args.postfilter,
        device_id=parallel.device_index())
    if parallel.is_master():
        Path(args.out).mkdir(parents=True, exist_ok=True)
    parallel.barrier()
    if args.test_run:
        dataset_slice = True
    elif args.number_of_jobs > 1:
        dataset_slice = slice(args.job_id - 1, None, args.number_of_jobs)
    else:
        dataset_slice = False
    enhancer.enhance_session(args.session_id, Path(args.out) / 'audio',
                             dataset_slice=dataset_slice, audio_dir_exist_ok=True)
    if parallel.is_master():
        print(f'Finished: {Path(args.out) / "audio"}')


if __name__ == '__main__':
    main()
