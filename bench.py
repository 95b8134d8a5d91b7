"""Headline benchmark: utterance-seconds enhanced per second (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the whole hot path (STFT -> WPE -> CACGMM EM -> MVDR+BAN ->
iSTFT) over one synthetic utterance of BASELINE.json configs[1]: 24 microphones
(6 arrays x 4), 15 s at 16 kHz, 4 speakers + noise class, WPE taps 10 / delay 2 /
3 iterations, 20 EM iterations (+1 predict), MVDR-Souden with BAN.  The time-domain
observation and the activity are resident in HBM before the timed region starts;
every step runs the full pipeline on them (nothing is cached between steps).

Multi-GPU: utterances are independent, so each rank (one per GPU) enhances its own
utterances; there is no data-path collective.  torch.distributed is used only for
the launch rendezvous, the barriers around the timed region and the max-over-ranks
of the elapsed time.  scaling = "weak" (work per GPU is fixed).

Rank 0 prints ONE JSON line with the contract fields plus
  roofline      for the kernel that dominates device time (HIP-event timing of every
                launch on the stream the kernels run on, collected over the timed
                region), priced with the algorithmic work of pb_chime5_amd/roofline.py
  cpu_baseline  the NumPy oracle (float64, single thread, per-frequency loops like
                the reference) timed on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

WORKLOAD = dict(num_channels=24, seconds=15.0, num_speakers=4, wpe_taps=10, wpe_delay=2,
                wpe_iterations=3, bss_iterations=20, bss_iterations_post=1)


def cpu_baseline(utt, sample_bins=24):
    """Time the oracle (kind 'port') on one host core: full STFT / masks / MVDR /
    iSTFT, WPE + EM on `sample_bins` of the 513 frequency bins (the reference loops
    over frequencies in Python, so cost is linear in the number of bins), then
    extrapolate to 513 bins."""
    for var in ('OMP_NUM_THREADS', 'MKL_NUM_THREADS', 'OPENBLAS_NUM_THREADS'):
        os.environ[var] = '1'
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=1)
    except Exception:
        limiter = None
    sys.path.insert(0, str(REPO / 'oracle'))
    import gss_oracle as oracle
    F = 513
    bins = np.linspace(0, F - 1, sample_bins).astype(int)
    t0 = time.perf_counter()
    Obs = oracle.stft(utt.obs)
    act_f = oracle.activity_time_to_frequency(utt.activity_array, 1024, 256, True)
    t_stft = time.perf_counter() - t0
    t0 = time.perf_counter()
    Xs = oracle.wpe_block(Obs[..., bins], WORKLOAD['wpe_taps'], WORKLOAD['wpe_delay'],
                          WORKLOAD['wpe_iterations'])
    t_wpe = time.perf_counter() - t0
    t0 = time.perf_counter()
    post = oracle.gss_block(Xs, act_f, WORKLOAD['bss_iterations'],
                            WORKLOAD['bss_iterations_post'])
    t_em = time.perf_counter() - t0
    # beamformer + istft at full size, on stand-in masks of the right shape
    masks = np.repeat(post[..., :1], F, axis=-1)
    t0 = time.perf_counter()
    sf, ef = oracle.start_end_context_frames(utt.ex, 1024, 256, True)
    masks[:, :sf] = 0
    masks[:, -ef:] = 0
    tm = masks[utt.target_index]
    dm = np.sum(np.delete(masks, utt.target_index, axis=0), axis=0)
    X_hat = oracle.beamform_mvdr_souden_from_masks(Obs, tm, dm, ban=True)
    oracle.istft(X_hat)
    t_bf = time.perf_counter() - t0
    if limiter is not None:
        limiter.restore_original_limits()
    scale = F / float(sample_bins)
    total = t_stft + t_bf + scale * (t_wpe + t_em)
    return {
        'value': utt.seconds / total, 'unit': 'utterance-seconds/s', 'cores': 1,
        'kind': 'port',
        'sample': (f'one config-2 utterance; WPE+EM on {sample_bins} of {F} frequency bins '
                   f'({t_wpe + t_em:.1f} s measured, x{scale:.1f}), STFT/MVDR/iSTFT in full '
                   f'({t_stft + t_bf:.1f} s); NumPy float64 restatement of the pb_chime5 CPU '
                   'path (reference numeric libraries unavailable), 1 thread'),
        'seconds_per_utterance': total,
    }


PROFILE_STEPS = 3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-bins', type=int, default=96)
    ap.add_argument('--overlap-info', action='store_true',
                    help='also measure two utterances in flight with PCIe transfers (session mode; '
                         'informational, off by default so that a rocprofv3 --stats run of the '
                         'default command sees only back-to-back launches)')
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run for --gpus > 1')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an AMD GPU (there is no CPU fallback for the hot path)')
    # GSS_BENCH_BACKEND=gloo lets the multi-rank control flow be exercised on a box with
    # fewer GPUs than ranks (ranks then share devices); the driver's runs use RCCL.
    backend = os.environ.get('GSS_BENCH_BACKEND', 'nccl')
    device_index = local_rank if backend == 'nccl' else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    coll_device = 'cuda' if backend == 'nccl' else 'cpu'
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend=backend)

    from pb_chime5_amd import ops, roofline, synthetic
    from pb_chime5_amd._capi import Context

    ctx = Context(device_index)
    params = ops.make_params(wpe=True, wpe_taps=WORKLOAD['wpe_taps'],
                             wpe_delay=WORKLOAD['wpe_delay'],
                             wpe_iterations=WORKLOAD['wpe_iterations'],
                             bss_iterations=WORKLOAD['bss_iterations'],
                             bss_iterations_post=WORKLOAD['bss_iterations_post'])
    ops._prepare_windows(ctx, params.stft_size, params.stft_shift)
    utt = synthetic.config2(seed=2 + rank, num_channels=WORKLOAD['num_channels'],
                            seconds=WORKLOAD['seconds'],
                            num_speakers=WORKLOAD['num_speakers'])
    ctx_samples = utt.ex['start_orig']['original']
    resident = ops.ResidentUtterance(ctx, utt.obs, utt.activity_array, params)

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        resident.enqueue(utt.target_index, ctx_samples, ctx_samples)

    for _ in range(args.warmup):
        step()
    barrier()
    # Untimed pass with HIP events around EVERY launch: the per-kernel table and the
    # name of the dominant kernel.  (Two events per launch cost ~4 us of stream time
    # each; around all ~260 launches of an utterance that is 6 % of the step, so the
    # timed region below only carries events around the dominant kernel.)
    ctx.profile_filter(None)
    ctx.profile_enable(True)
    ctx.profile_reset()
    for _ in range(PROFILE_STEPS):
        step()
    ctx.synchronize()
    prof_all = ctx.profile_report()
    dominant = max(prof_all, key=lambda k: prof_all[k]['ms'])
    if dist is not None:       # every rank times the same kernel
        names = sorted(prof_all)
        idx = torch.tensor([names.index(dominant)], device=coll_device)
        dist.broadcast(idx, src=0)
        dominant = names[int(idx.item())]
    ctx.profile_filter(dominant)
    ctx.profile_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    prof = ctx.profile_report()
    ctx.profile_enable(False)
    ctx.profile_filter(None)
    x_hat = resident.result()
    assert np.all(np.isfinite(x_hat)) and x_hat.shape[0] == resident.n_out

    # Informational only (not part of `value`): the session driver's mode -- two
    # utterances in flight on two contexts / HIP streams (ops.UtterancePipeline), which
    # lets one utterance's latency-bound kernels overlap the other's compute-bound
    # ones.  This figure INCLUDES the H2D copy of every utterance's samples and the D2H
    # copy of the result; like the session driver it uploads the samples as 16-bit PCM
    # (the STFT kernel converts them).
    overlap = None
    if args.overlap_info and rank == 0:
        pipe = ops.UtterancePipeline(params, depth=2, first_ctx=ctx)
        n2 = max(args.steps, 4)
        pcm = np.clip(np.rint(utt.obs / np.abs(utt.obs).max() * 30000), -32768, 32767).astype(np.int16)

        def run(count):
            for i in range(count):
                if pipe.full():
                    pipe.pop()
                pipe.enqueue(i, pcm, utt.activity_array, utt.target_index, ctx_samples,
                             ctx_samples)
            while len(pipe):
                pipe.pop()
        run(2)
        t2 = time.perf_counter()
        run(n2)
        e2 = time.perf_counter() - t2
        overlap = {'utterances_in_flight': 2, 'utterances': n2, 'includes': 'H2D (PCM16) + D2H per utterance',
                   'value': n2 * utt.seconds / e2, 'unit': 'utterance-seconds/s'}
        pipe.close()

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        size = dict(F=params.stft_size // 2 + 1, T=resident.T, D=resident.D, K=resident.K,
                    taps=WORKLOAD['wpe_taps'], N=resident.N)
        total_ms = sum(v['ms'] for v in prof_all.values())
        kernels = {}
        for name, v in sorted(prof_all.items(), key=lambda kv: -kv[1]['ms']):
            avg = v['ms'] / max(v['calls'], 1)
            entry = roofline.roofline_entry(name, avg, **size)
            kernels[name] = {
                'calls_per_step': v['calls'] / PROFILE_STEPS, 'avg_ms': round(avg, 5),
                'share': round(v['ms'] / total_ms, 4),
                'frac_of_roof': round(entry['frac'], 4) if entry else None,
                'bound': entry['bound'] if entry else None}
        roof = roofline.roofline_entry(dominant, prof[dominant]['ms'] / prof[dominant]['calls'],
                                       **size)
        traffic_file = REPO / 'profiles' / 'traffic.json'
        if roof is not None and traffic_file.exists():
            try:
                roof['traffic'] = json.loads(traffic_file.read_text()).get(dominant, {}).get('bytes')
                roof['traffic_note'] = ('HBM-side bytes per launch from profiles/traffic.json '
                                        '(rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)')
            except Exception:
                pass
        line = {
            'metric': 'utterance-seconds enhanced/sec/GPU (24ch, 20 EM iters)',
            'value': args.gpus * args.steps * utt.seconds / elapsed,
            'unit': 'utterance-seconds/s',
            'n_gpus': args.gpus, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {
                'workload': ('BASELINE.json configs[1]: synthetic 24-mic (6 arrays x 4) 15 s '
                             'utterance, 4 speakers + noise class, WPE taps=10 delay=2 iters=3, '
                             '20 EM iterations + predict, MVDR-Souden + BAN; inputs resident in HBM'),
                'utterance_seconds': utt.seconds, 'channels': resident.D,
                'frames': resident.T, 'classes': resident.K,
                'utterances_per_step_per_gpu': 1, 'parallelism': f'utterance-sharded x{args.gpus}',
            },
            'realtime_factor_per_gpu': args.steps * utt.seconds / elapsed,
            'roofline': roof,
            'kernels': kernels,
            'kernels_note': (f'per-kernel table from an untimed pass of {PROFILE_STEPS} steps with HIP '
                             'events around every launch; `roofline` is the dominant kernel timed '
                             'inside the timed region'),
            'device_ms_per_step': total_ms / PROFILE_STEPS,
            'workspace_bytes': ctx.workspace_bytes(),
            'pipelined_session_info': overlap,
        }
        if args.gpus == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(utt, args.cpu_bins)
            line['speedup_vs_cpu_core'] = line['value'] / line['cpu_baseline']['value']
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
