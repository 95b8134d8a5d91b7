"""Headline benchmark: utterance-seconds enhanced per second (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3]

``--gpus N`` with N > 1 may be started either way:

    python bench.py --gpus N ...                       (spawns its N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One rank per GPU; when the node has fewer GPUs than ranks the ranks share devices
(LOCAL_RANK % device count) and the host-side rendezvous uses gloo instead of RCCL.
Utterances are independent, so there is NO data-path collective: torch.distributed
carries the rendezvous, the barriers around the timed regions, the max-over-ranks
of the elapsed time and (through its TCP store) the shared work counter of
``pb_chime5_amd.parallel.split_managed`` -- the replacement of the reference's
``dlp_mpi.split_managed`` (/root/reference/pb_chime5/core.py:381).

--config 2 (default; BASELINE.json configs[1], the configuration the metric is quoted on)
    A "step" is one pass of the whole hot path (STFT -> WPE -> CACGMM EM -> MVDR+BAN ->
    iSTFT) over one synthetic 24-microphone 15 s utterance whose samples and activity
    are resident in HBM when the timed region starts (`value`).  Every rank times its
    own K steps ("weak" scaling).  Besides `value` the line carries
      value_incl_pcie   the session driver's mode: two utterances in flight per GPU,
                        PCM16 upload (H2D) and result download (D2H) inside the wall clock
      configs           in-process timings of the other BASELINE configs (1, one dev-shaped
                        item of 3, 5; 4 = "not run: corpus unavailable")
      config3_sharded   BASELINE configs[2]: 512 dev-shaped utterances pulled by all ranks
                        from the shared longest-first queue, 2 in flight per GPU, H2D/D2H timed
      em_loop           EM-loop time per iteration against the HBM and f64-VALU roofs for
                        24 channels and for one array (4 channels)
      roofline          the dominant kernel, timed with HIP events inside the timed region
      cpu_baseline      the NumPy oracle on W = physical cores - 1 worker processes
    ``--only-headline`` skips everything but `value` / `roofline` / `kernels` (the command
    the rocprofv3 summaries under profiles/ are taken with: back-to-back launches of one
    stream and one shape only).

--config 3
    The sharded session alone: `value` = sum of utterance seconds / wall clock over all
    ranks, "strong" scaling (the 512 items are fixed), steps = items.
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
PROFILE_STEPS = 3
SR = 16000


# ----------------------------------------------------------------------------------
# CPU baseline (the oracle; test / bench infrastructure, never on the product path)
# ----------------------------------------------------------------------------------
def _limit_threads():
    for var in ('OMP_NUM_THREADS', 'MKL_NUM_THREADS', 'OPENBLAS_NUM_THREADS'):
        os.environ[var] = '1'
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=1)
    except Exception:
        pass


def _cpu_worker(job):
    """One worker process of the CPU baseline = one rank of `mpiexec -np W+1`
    (/root/reference/README.md:108-111), numeric libraries pinned to one thread like
    /root/reference/pb_chime5/__init__.py:3-14.  Enhances whole utterances with the
    oracle: config 1 in full; config 2 with WPE + EM on a sample of the frequency bins
    (the reference loops over frequencies in Python, cost is linear in the bins) and
    everything else in full, extrapolated to 513 bins."""
    _limit_threads()
    sys.path.insert(0, str(REPO / 'oracle'))
    import gss_oracle as oracle
    path, sample_bins, n_cfg1 = job
    data = np.load(path)
    out = {}
    # ---- config 1: full utterances
    obs, act = data['obs1'], data['act1'].astype(bool)
    ex1 = dict(start={'original': 0}, start_orig={'original': 0},
               end_orig={'original': obs.shape[1]}, end={'original': obs.shape[1]})
    t0 = time.perf_counter()
    for _ in range(n_cfg1):
        oracle.enhance_observation(obs, act, 0, ex1, wpe=False, bss_iterations=5,
                                   bss_iterations_post=1)
    out['cfg1_s'] = (time.perf_counter() - t0) / n_cfg1
    # ---- config 2: bin-sampled
    obs, act = data['obs2'], data['act2'].astype(bool)
    ctx = int(data['ctx2'])
    ex2 = dict(start={'original': 0}, start_orig={'original': ctx},
               end_orig={'original': obs.shape[1] - ctx}, end={'original': obs.shape[1]})
    F = 513
    bins = np.linspace(0, F - 1, sample_bins).astype(int)
    t0 = time.perf_counter()
    Obs = oracle.stft(obs)
    act_f = oracle.activity_time_to_frequency(act, 1024, 256, True)
    t_full = time.perf_counter() - t0
    t0 = time.perf_counter()
    Xs = oracle.wpe_block(Obs[..., bins], WORKLOAD['wpe_taps'], WORKLOAD['wpe_delay'],
                          WORKLOAD['wpe_iterations'])
    post = oracle.gss_block(Xs, act_f, WORKLOAD['bss_iterations'],
                            WORKLOAD['bss_iterations_post'])
    t_bins = time.perf_counter() - t0
    masks = np.repeat(post[..., :1], F, axis=-1)       # stand-in masks of the full shape
    t0 = time.perf_counter()
    sf, ef = oracle.start_end_context_frames(ex2, 1024, 256, True)
    masks[:, :sf] = 0
    masks[:, -ef:] = 0
    X_hat = oracle.beamform_mvdr_souden_from_masks(
        Obs, masks[0], np.sum(masks[1:], axis=0), ban=True)
    oracle.istft(X_hat)
    t_full += time.perf_counter() - t0
    out['cfg2_full_part_s'] = t_full
    out['cfg2_bins_part_s'] = t_bins
    out['cfg2_s'] = t_full + t_bins * F / float(sample_bins)
    # ---- config 5 (120 s, 12 ch, 40 EM iterations, GEV + BAN): bin-sampled the same way
    if 'obs5' in data.files:
        del Obs, Xs, post, masks, X_hat
        obs, act = data['obs5'], data['act5'].astype(bool)
        ctx = int(data['ctx5'])
        ex5 = dict(start={'original': 0}, start_orig={'original': ctx},
                   end_orig={'original': obs.shape[1] - ctx}, end={'original': obs.shape[1]})
        bins5 = np.linspace(0, F - 1, max(sample_bins // 3, 2)).astype(int)
        t0 = time.perf_counter()
        Obs = oracle.stft(obs)
        act_f = oracle.activity_time_to_frequency(act, 1024, 256, True)
        t_full = time.perf_counter() - t0
        t0 = time.perf_counter()
        Xs = oracle.wpe_block(Obs[..., bins5], 10, 2, 3)
        post = oracle.gss_block(Xs, act_f, 40, 1)
        t_bins = time.perf_counter() - t0
        masks = np.repeat(post[..., :1], F, axis=-1)
        t0 = time.perf_counter()
        sf, ef = oracle.start_end_context_frames(ex5, 1024, 256, True)
        masks[:, :sf] = 0
        masks[:, -ef:] = 0
        X_hat = oracle.beamform_gev_from_masks(Obs, masks[0], np.sum(masks[1:], axis=0), ban=True)
        oracle.istft(X_hat)
        t_full += time.perf_counter() - t0
        out['cfg5_s'] = t_full + t_bins * F / float(len(bins5))
        out['cfg5_bins'] = int(len(bins5))
    return out


def _cpu_worker_config3(job):
    """One item of BASELINE configs[2] (dev-shaped, 24 ch, 2 x 15 s context) through the oracle
    on one core: the pool hands the items out on request, like dlp_mpi's master
    (/root/reference/pb_chime5/core.py:381).  WPE + EM on `sample_bins` of the 513 bins and
    extrapolated, everything else in full (see _cpu_worker)."""
    _limit_threads()
    sys.path.insert(0, str(REPO / 'oracle'))
    import gss_oracle as oracle
    path, base, core, sample_bins = job
    data = np.load(path)
    pcm, act = data[f'pcm{base}'], data[f'act{base}'].astype(bool)
    ctx, core_max = int(data['context']), int(data['core_max'])
    a, b = ctx + core, ctx + core_max
    obs = np.concatenate([pcm[:, :a], pcm[:, b:]], axis=1).astype(np.float64) / 2 ** 15
    activity = np.concatenate([act[:, :a], act[:, b:]], axis=1)
    ex = dict(start={'original': 0}, start_orig={'original': ctx},
              end_orig={'original': obs.shape[1] - ctx}, end={'original': obs.shape[1]})
    F = 513
    bins = np.linspace(0, F - 1, sample_bins).astype(int)
    t0 = time.perf_counter()
    Obs = oracle.stft(obs)
    act_f = oracle.activity_time_to_frequency(activity, 1024, 256, True)
    t_full = time.perf_counter() - t0
    t0 = time.perf_counter()
    Xs = oracle.wpe_block(Obs[..., bins], WORKLOAD['wpe_taps'], WORKLOAD['wpe_delay'],
                          WORKLOAD['wpe_iterations'])
    post = oracle.gss_block(Xs, act_f, WORKLOAD['bss_iterations'], WORKLOAD['bss_iterations_post'])
    t_bins = time.perf_counter() - t0
    masks = np.repeat(post[..., :1], F, axis=-1)
    t0 = time.perf_counter()
    sf, ef = oracle.start_end_context_frames(ex, 1024, 256, True)
    masks[:, :sf] = 0
    masks[:, -ef:] = 0
    oracle.istft(oracle.beamform_mvdr_souden_from_masks(Obs, masks[0], np.sum(masks[1:], axis=0),
                                                        ban=True))
    t_full += time.perf_counter() - t0
    return dict(seconds=obs.shape[1] / SR, measured_s=t_full + t_bins,
                estimate_s=t_full + t_bins * F / float(sample_bins))


def _host_cpu():
    model, phys = None, set()
    try:
        pid = cid = None
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name') and model is None:
                model = line.split(':', 1)[1].strip()
            elif line.startswith('physical id'):
                pid = line.split(':', 1)[1].strip()
            elif line.startswith('core id'):
                cid = line.split(':', 1)[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    logical = os.cpu_count() or 1
    try:
        logical = len(os.sched_getaffinity(0))
    except Exception:
        pass
    physical = min(len(phys), logical) if phys else logical
    # a container may be limited to fewer CPUs than it can see (cgroup v2 cpu.max / v1 quota)
    quota = None
    try:
        q, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if q != 'max':
            quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            q = float(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            period = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    mem_gb = None
    try:
        for line in open('/proc/meminfo'):
            if line.startswith('MemAvailable'):
                mem_gb = int(line.split()[1]) / 1e6
    except OSError:
        pass
    return model or 'unknown', physical, logical, mem_gb, quota


def cpu_baseline(utt2, sample_bins=24, max_workers=None, with_config5=True, with_config3=True):
    """the NumPy oracle (kind 'port') on W = physical cores - 1
    single-threaded worker processes, every worker enhancing whole utterances at the same
    time (so the figure includes what the cores cost each other in memory bandwidth)."""
    import multiprocessing as mp
    import tempfile
    from pb_chime5_amd import synthetic
    model, physical, logical, mem_gb, quota = _host_cpu()
    usable = physical if quota is None else max(1, min(physical, int(quota)))
    W = max(usable - 1, 1)
    if mem_gb is not None:                       # ~1.5 GB per worker at config 2
        W = max(1, min(W, int(mem_gb / 2.0)))
    if max_workers:
        W = max(1, min(W, max_workers))
    utt1 = synthetic.config1()
    sr5 = SR
    iv5 = [(50 * sr5, 70 * sr5), (10 * sr5, 60 * sr5), (40 * sr5, 100 * sr5), (65 * sr5, 115 * sr5)]
    utt5 = synthetic.make_utterance(5, 12, 120 * sr5, iv5, start_context=50 * sr5,
                                    end_context=50 * sr5, fast=True) if with_config5 else None
    saved = {k: os.environ.get(k) for k in ('OMP_NUM_THREADS', 'MKL_NUM_THREADS',
                                             'OPENBLAS_NUM_THREADS')}
    tmp = tempfile.NamedTemporaryFile(suffix='.npz', delete=False)
    tmp.close()
    tmp3 = tempfile.NamedTemporaryFile(suffix='.npz', delete=False)
    tmp3.close()
    try:
        extra = {}
        if utt5 is not None:
            extra = dict(obs5=utt5.obs, act5=utt5.activity_array.astype(np.uint8),
                         ctx5=utt5.ex['start_orig']['original'])
        np.savez(tmp.name, obs1=utt1.obs, act1=utt1.activity_array.astype(np.uint8),
                 obs2=utt2.obs, act2=utt2.activity_array.astype(np.uint8),
                 ctx2=utt2.ex['start_orig']['original'], **extra)
        for k in saved:
            os.environ[k] = '1'                  # inherited by the spawned workers
        t0 = time.perf_counter()
        # (an executor, not mp.Pool: a worker that dies raises BrokenProcessPool here
        # instead of being respawned for ever)
        from concurrent.futures import ProcessPoolExecutor
        with ProcessPoolExecutor(W, mp_context=mp.get_context('spawn')) as pool:
            futures = [pool.submit(_cpu_worker, (tmp.name, sample_bins, 2)) for _ in range(W)]
            res = [f.result(timeout=600) for f in futures]
            wall = time.perf_counter() - t0
            # ---- all-core aggregate on a 2 W-item subset of config 3, items
            # handed out on request (the executor's queue), longest first
            config3 = None
            if with_config3:
                pool3 = Config3Pool(2 * W, 2)
                np.savez(tmp3.name, context=pool3.context, core_max=pool3.core_max,
                         **{f'pcm{i}': b[0] for i, b in enumerate(pool3.bases)},
                         **{f'act{i}': b[1] for i, b in enumerate(pool3.bases)})
                order = np.argsort(pool3.num_samples)[::-1]
                bins3 = max(sample_bins // 3, 2)
                t3 = time.perf_counter()
                futures = [pool.submit(_cpu_worker_config3,
                                       (tmp3.name, int(i) % 2, pool3.cores[int(i)], bins3))
                           for i in order]
                res3 = [f.result(timeout=900) for f in futures]
                wall3 = time.perf_counter() - t3
                secs3 = sum(r['seconds'] for r in res3)
                est3 = sum(r['estimate_s'] for r in res3)
                config3 = {
                    'aggregate_value': W * secs3 / est3, 'per_core_value': secs3 / est3,
                    'items': len(res3), 'utterance_seconds': secs3,
                    'core_seconds_extrapolated': est3, 'wall_s_sampled': wall3,
                    'sample': f'BASELINE configs[2]: the first {len(res3)} = 2 W dev-shaped items, '
                              f'pulled longest first from one queue by {W} single-thread workers; per '
                              f'item WPE + EM on {bins3} of 513 bins (extrapolated), the rest in full; '
                              'aggregate = W x audio seconds / extrapolated core seconds'}
    finally:
        os.unlink(tmp.name)
        if os.path.exists(tmp3.name):
            os.unlink(tmp3.name)
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    cfg1 = float(np.mean([r['cfg1_s'] for r in res]))
    cfg2 = float(np.mean([r['cfg2_s'] for r in res]))
    config5 = None
    if utt5 is not None:
        cfg5 = float(np.mean([r['cfg5_s'] for r in res]))
        config5 = {'per_core_value': utt5.seconds / cfg5, 'aggregate_value': W * utt5.seconds / cfg5,
                   'seconds_per_utterance_per_core': cfg5,
                   'sample': f'BASELINE configs[4] (120 s, 12 ch, 40 EM iterations, GEV + BAN), one segment '
                             f'per worker, WPE + EM on {res[0]["cfg5_bins"]} of 513 bins, the rest in full'}
    return {
        'value': W * utt2.seconds / cfg2, 'unit': 'utterance-seconds/s', 'cores': W,
        'kind': 'port', 'cpu_model': model, 'physical_cores': physical,
        'logical_cpus': logical, 'cgroup_cpu_quota': quota,
        'per_core_value': utt2.seconds / cfg2,
        'seconds_per_utterance_per_core': cfg2,
        'config1': {'per_core_value': utt1.seconds / cfg1, 'aggregate_value': W * utt1.seconds / cfg1,
                    'seconds_per_utterance_per_core': cfg1,
                    'sample': 'BASELINE configs[0] (4 ch, 5 s, WPE off, 5 EM iterations), 2 whole '
                              'utterances per worker, nothing sampled'},
        'config5': config5,
        'config3': config3,
        'wall_s': wall,
        'sample': (f'{W} worker processes (usable physical cores - 1, like `mpiexec -np W+1`), each 1 '
                   f'thread, all running at the same time; per worker one config-2 utterance: '
                   f'STFT / masks / MVDR+BAN / iSTFT in full '
                   f'({np.mean([r["cfg2_full_part_s"] for r in res]):.1f} s), WPE + 20 EM '
                   f'iterations on {sample_bins} of 513 frequency bins '
                   f'({np.mean([r["cfg2_bins_part_s"] for r in res]):.1f} s, x{513 / sample_bins:.1f}); '
                   'NumPy float64 restatement of the pb_chime5 CPU path (reference numeric '
                   'libraries unavailable)'),
    }


# ----------------------------------------------------------------------------------
# workloads
# ----------------------------------------------------------------------------------
def to_pcm16(obs):
    return np.clip(np.rint(obs / np.abs(obs).max() * 30000), -32768, 32767).astype(np.int16)


class Config3Pool:
    """BASELINE configs[2]: 512 dev-shaped utterances (24 ch, K = 5, core length
    ~ LogNormal(ln 2.5 s, 0.7) clipped to [0.5 s, 15 s] drawn with the seeds of
    synthetic.config3_item, 240000 context samples on each side).  Generating 512
    distinct 24-channel recordings costs minutes of host time, so the samples come from
    a pool of `pool` seeded base recordings with a 15 s core; item i is base i % pool
    with its core cut to item i's length (context | first U_i of the core | context).
    Every item is uploaded and enhanced in full; nothing is cached between items."""

    def __init__(self, items, pool, num_channels=24, context=240000):
        from pb_chime5_amd import synthetic
        self.context = context
        self.core_max = 15 * SR
        self.cores = [synthetic.config3_core_samples(i) for i in range(items)]
        self.num_samples = [c + 2 * context for c in self.cores]
        n = self.core_max + 2 * context
        self.bases = []
        for p in range(pool):
            rng = np.random.default_rng(77 + p)
            iv = [(context, context + self.core_max)]
            for _ in range(3):
                length = int(rng.uniform(0.3, 0.6) * n)
                a = int(rng.integers(0, n - length))
                iv.append((a, a + length))
            u = synthetic.make_utterance(5000 + p, num_channels, n, iv, target=0,
                                         start_context=context, end_context=context,
                                         rir_taps=1024, noise=3e-2, fast=True)
            self.bases.append((to_pcm16(u.obs), u.activity_array.astype(np.uint8)))

    def item(self, i):
        pcm, act = self.bases[i % len(self.bases)]
        c, ctx = self.cores[i], self.context
        a, b = ctx + c, ctx + self.core_max
        obs = np.concatenate([pcm[:, :a], pcm[:, b:]], axis=1)
        activity = np.concatenate([act[:, :a], act[:, b:]], axis=1)
        return obs, activity, 0, ctx, ctx

    def seconds(self, i):
        return self.num_samples[i] / SR


def run_session(pipe, indices, get_item, clock=None):
    """What Enhancer._enhance_and_write does per GPU: keep the pipeline full, pop the
    oldest result when it is.  Returns the number of utterances handled.  `clock` (a dict)
    collects where the host thread of this rank spends its time: waiting for the next index
    of the shared queue, preparing the item on the host, uploading + enqueueing, and blocked
    on the GPU for the oldest result (download included)."""
    count = 0
    clock = clock if clock is not None else {}
    for k in ('queue_wait_s', 'host_prepare_s', 'enqueue_s', 'gpu_wait_s'):
        clock.setdefault(k, 0.0)
    it = iter(indices)
    while True:
        t0 = time.perf_counter()
        try:
            i = next(it)
        except StopIteration:
            break
        t1 = time.perf_counter()
        prepared = get_item(i)
        t2 = time.perf_counter()
        if pipe.full():
            pipe.pop()
        t3 = time.perf_counter()
        pipe.enqueue(i, *prepared)
        t4 = time.perf_counter()
        clock['queue_wait_s'] += t1 - t0
        clock['host_prepare_s'] += t2 - t1
        clock['gpu_wait_s'] += t3 - t2
        clock['enqueue_s'] += t4 - t3
        count += 1
    t0 = time.perf_counter()
    while len(pipe):
        pipe.pop()
    clock['gpu_wait_s'] += time.perf_counter() - t0
    return count


def other_workload(name, synthetic, ops):
    """(utterance, params, description) of the BASELINE shapes that are not the headline; the
    same scenes as the `configs` table of the default line."""
    sr = SR
    n3 = 554490                          # synthetic.config3_item(0): 4.66 s core + 2 x 15 s
    iv3 = [(240000, n3 - 240000), (100000, 400000), (50000, 250000), (300000, 520000)]
    meta = dict(WORKLOAD, name=name, steps=5)
    params = ops.make_params(wpe=True, wpe_taps=WORKLOAD['wpe_taps'], wpe_delay=WORKLOAD['wpe_delay'],
                             wpe_iterations=WORKLOAD['wpe_iterations'],
                             bss_iterations=WORKLOAD['bss_iterations'],
                             bss_iterations_post=WORKLOAD['bss_iterations_post'])
    if name == '3i':
        utt = synthetic.make_utterance(1000, 24, n3, iv3, start_context=240000, end_context=240000,
                                       rir_taps=1024, noise=3e-2, fast=True)
        meta['description'] = ('one dev-shaped item of BASELINE.json configs[2]: 24 ch, 34.7 s incl. '
                               '2 x 15 s context, otherwise as configs[1]; inputs resident in HBM')
    elif name == '1a':
        utt = synthetic.make_utterance(1001, 4, n3, iv3, start_context=240000, end_context=240000,
                                       rir_taps=1024, noise=3e-2, fast=True)
        meta.update(num_channels=4, steps=20)
        meta['description'] = ('the reference default multiarray=False (core.py:572-575): one array '
                               '(4 ch) of a dev-shaped item, 34.7 s incl. 2 x 15 s context, WPE taps=10, '
                               '20 EM iterations, MVDR-Souden + BAN; inputs resident in HBM')
    elif name == '5':
        iv5 = [(50 * sr, 70 * sr), (10 * sr, 60 * sr), (40 * sr, 100 * sr), (65 * sr, 115 * sr)]
        utt = synthetic.make_utterance(5, 12, 120 * sr, iv5, start_context=50 * sr,
                                       end_context=50 * sr, fast=True)
        params = ops.make_params(wpe=True, wpe_taps=10, wpe_delay=2, wpe_iterations=3,
                                 bss_iterations=40, bss_iterations_post=1, bf='gev_ban')
        meta.update(num_channels=12, seconds=120.0, bss_iterations=40, steps=5)
        meta['description'] = ('BASELINE.json configs[4]: 120 s RTTM-style segment, 12 ch '
                               '(outer_array_mics of 6 arrays), WPE taps=10 delay=2 iters=3, 40 EM '
                               'iterations + predict, GEV + BAN; inputs resident in HBM')
    else:
        raise ValueError(name)
    return utt, params, meta


def time_resident(ctx, ops, utt, params, steps, warmup=1):
    """ms per utterance, one stream, inputs resident in HBM."""
    ops._prepare_windows(ctx, params.stft_size, params.stft_shift)
    res = ops.ResidentUtterance(ctx, utt.obs, utt.activity_array, params)
    c0, c1 = utt.ex['start_orig']['original'], utt.ex['end']['original'] - utt.ex['end_orig']['original']
    for _ in range(warmup):
        res.enqueue(utt.target_index, c0, c1)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        res.enqueue(utt.target_index, c0, c1)
    ctx.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    x = res.result()
    assert np.all(np.isfinite(x)) and x.shape[0] == res.n_out
    return ms, res


def profile_kernels(ctx, res, utt, steps):
    c0, c1 = utt.ex['start_orig']['original'], utt.ex['end']['original'] - utt.ex['end_orig']['original']
    ctx.profile_filter(None)
    ctx.profile_enable(True)
    ctx.profile_reset()
    for _ in range(steps):
        res.enqueue(utt.target_index, c0, c1)
    ctx.synchronize()
    prof = ctx.profile_report()
    ctx.profile_enable(False)
    return prof


def em_loop_entry(prof, steps, iterations, F, T, D, K, roofline):
    """EM loop (E-step, M-step, model update) per iteration against both roofs
    (bytes = one pass over the STFT tensor, B_Y = 16 F T D in c128).  One array
    (D = 4): the whole loop is ONE launch (em_onchip: 20 iterations + predict, the observation
    streamed once per pass and nothing else -- no weight tensor, no partial sums)."""
    size = dict(F=F, T=T, D=D, K=K, taps=1, N=0)
    by = roofline.stft_bin_bytes(F, T, D)
    flops = sum(roofline.kernel_work(n, **size)['flops'] for n in ('em_estep', 'em_mstep', 'em_chol'))
    if 'em_onchip' in prof:
        passes = iterations + 1                      # + the predict pass, same traffic, E only
        ms_pass = prof['em_onchip']['ms'] / steps / passes
        sec = ms_pass * 1e-3
        return {
            'kernel': 'em_onchip (one launch per utterance: all iterations + predict)',
            'channels': D, 'frames': T, 'classes': K, 'launches_per_utterance': 1,
            'ms_per_utterance': prof['em_onchip']['ms'] / steps,
            'ms_per_iteration': ms_pass, 'passes': passes,
            'algorithmic_bytes_per_iteration': by,
            'bytes_note': 'ONE pass over the unit-normalised observation per iteration; the '
                          'model, the M-step weights and all sums stay on chip.  The observation of one '
                          'array (71 MB at T = 2169) is re-read 21 times out of the Infinity Cache, not '
                          'out of HBM: the fraction below is that stream priced against the HBM roof for '
                          'comparison with the other shapes, not HBM traffic',
            'infinity_cache_stream_frac_of_hbm_roof': by / sec / 1e9 / roofline.PEAK_HBM_GBS,
            'hbm_frac': None,
            'executed_flops_per_iteration': flops,
            'valu_f64_frac': flops / sec / 1e12 / roofline.PEAK_F64_TFLOPS,
            'binding_roof': 'VALU issue at 2 workgroups per CU (513 frequencies on 256 CUs): phase E runs '
                            'at 8 cycles per instruction with two waves per SIMD; 27 % of the kernel is the '
                            'per-iteration sums + class update (tools/em4_trace.py, DESIGN 8.13)',
        }
    names = [n for n in ('em_estep', 'em_mstep', 'em_chol', 'em_eigh') if n in prof]
    ms_iter = sum(prof[n]['ms'] for n in names) / steps / iterations
    sec = ms_iter * 1e-3
    # the E-step alone against the roof that binds it: VALU issue (a wave64 f64 instruction
    # holds its SIMD for 4 cycles; tools/micro/valu_f64_bench.hip).  Priced at the NOMINAL
    # 2.4 GHz; under f64 load the clock settles near 2.05 GHz (profiles/r03_valu_f64_bench.txt),
    # so the fraction of the issue slots the silicon really had is ~1.17 x this.  The scalar
    # cache the model rows come through is reported beside it; it does not bind
    # (experiment 11).
    estep = None
    if 'em_estep' in prof and D in (4, 10, 12, 20, 24) and 2 <= K <= 6:
        e_sec = prof['em_estep']['ms'] / prof['em_estep']['calls'] * 1e-3
        rate = roofline.estep_scalar_bytes(F, T, D, K) / e_sec / roofline.NUM_CUS / (roofline.CLOCK_GHZ * 1e9)
        instr = roofline.estep_valu_instructions(F, T, D, K)
        issue_sec = instr * roofline.VALU_F64_ISSUE_CYCLES / roofline.NUM_SIMDS / (roofline.CLOCK_GHZ * 1e9)
        estep = {'bound': 'valu_issue' if issue_sec / e_sec >= 0.3 else 'latency',
                 'walk_valu_instructions_per_launch': instr,
                 'valu_issue_frac_at_nominal_clock': issue_sec / e_sec,
                 'scalar_cache_bytes_per_launch': roofline.estep_scalar_bytes(F, T, D, K),
                 'scalar_cache_frac_of_measured_ceiling': rate / roofline.PEAK_SMEM_BYTES_PER_CYCLE_PER_CU,
                 'valu_f64_frac': roofline.kernel_work('em_estep', **size)['flops'] / e_sec / 1e12
                 / roofline.PEAK_F64_TFLOPS}
    return {
        'estep_binding_roof': estep,
        'channels': D, 'frames': T, 'classes': K, 'ms_per_iteration': ms_iter,
        'kernel_ms_per_iteration': {n: prof[n]['ms'] / steps / iterations for n in names},
        'algorithmic_bytes_per_iteration': by,
        'hbm_frac': by / sec / 1e9 / roofline.PEAK_HBM_GBS,
        'executed_flops_per_iteration': flops,
        'valu_f64_frac': flops / sec / 1e12 / roofline.PEAK_F64_TFLOPS,
        'binding_roof': 'valu_f64' if flops / roofline.PEAK_F64_TFLOPS / 1e12 >
        by / roofline.PEAK_HBM_GBS / 1e9 else 'hbm',
    }


# ----------------------------------------------------------------------------------
def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None,
                    help='config 2: timed steps per rank (default 20); config 3: items (default 512)')
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', default='2', choices=('2', '3', '4s'),
                    help="2: the headline (+ everything else unless --only-headline); 3: BASELINE "
                         "configs[2] alone; 4s: the file-backed stand-in for configs[3] alone")
    ap.add_argument('--session-seconds', type=float, default=660.0,
                    help='config 4s: length of the synthetic dev-shaped session')
    ap.add_argument('--session-utterances', type=int, default=220)
    ap.add_argument('--sessions', default='S02,S09',
                    help="config 4s: the sessions of the stand-in corpus; the default is what "
                         "`session_id=dev` resolves to (scripts/run.py:45-71): S02 (6 arrays = 24 ch) "
                         "and S09 (5 arrays = 20 ch, mapping.py:67), --session-seconds and "
                         "--session-utterances split evenly between them")
    ap.add_argument('--loaders', type=int, default=3, help='config 4s: loader threads per rank')
    ap.add_argument('--no-config4s', action='store_true')
    ap.add_argument('--multiarray', default='True', choices=('True', 'False', 'outer_array_mics'),
                    help="config 4s: get_enhancer(multiarray=...) -- True: all 6 arrays (24 ch); "
                         "False: the reference's default, the reference array only (4 ch, "
                         "core.py:572-575); outer_array_mics: the CHiME-6 default (12 ch)")
    ap.add_argument('--items', type=int, default=512,
                    help='config-3 items in the config3_sharded block of a --config 2 run')
    ap.add_argument('--pool', type=int, default=2, help='config-3 base recordings')
    ap.add_argument('--inflight', type=int, default=2)
    ap.add_argument('--static', action='store_true', help='config 3: static instead of dynamic sharding')
    ap.add_argument('--only-headline', action='store_true')
    ap.add_argument('--workload', default='2', choices=('2', '5', '3i', '1a'),
                    help="what the headline section times (anything but '2' implies "
                         "--only-headline): 2 = BASELINE configs[1]; 5 = configs[4] (12 ch, 120 s, "
                         "40 iterations, GEV+BAN); 3i = one dev-shaped item of configs[2] (24 ch, "
                         "34.7 s); 1a = the same item on one array (4 ch, the reference default "
                         "multiarray=False).  The profiles of those shapes are taken with it.")
    ap.add_argument('--n1-value', type=float, default=None,
                    help="the same workload's `value` at --gpus 1 (a BENCH record): the line then "
                         "carries scaling_efficiency_vs_n1 = value / (N x n1) beside value_per_gpu")
    ap.add_argument('--one-at-a-time', action='store_true',
                    help='with --only-headline / --workload: also time the loop with the WPE stage on two '
                         'streams (value_one_at_a_time_api); off by default so that a profile of the '
                         'command sees one-stream launches only')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-bins', type=int, default=24)
    ap.add_argument('--cpu-workers', type=int, default=None)
    return ap.parse_args()


def main():
    args = parse_args()
    # before anything initialises ROCr (the host driver only supports dmabuf IPC; the driver's
    # environment exports it, a hand-made one may not)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world == 1 and args.gpus > 1:
        # plain `python bench.py --gpus N`: be our own mpiexec
        from pb_chime5_amd import parallel
        sys.exit(parallel.launch_local(args.gpus, [sys.executable, str(Path(__file__).resolve())]
                                       + sys.argv[1:]))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    # stdout carries exactly ONE line, the result: anything a library prints there (gloo's
    # connection banner, ...) is sent to stderr, the JSON goes to the saved descriptor.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(line):
        line.setdefault('placement', placement)
        line.setdefault('physical_gpus', len(set(packages)) or None)
        line.setdefault('ranks_sharing_a_physical_gpu', gpus_shared_by_ranks)
        if isinstance(line.get('value'), (int, float)):
            for k, v in parallel.scaling_rows(line['value'], args.gpus, args.n1_value).items():
                line.setdefault(k, v)
        os.write(result_fd, (json.dumps(line) + '\n').encode())

    import torch
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an AMD GPU (there is no CPU fallback for the hot path)')
    from pb_chime5_amd import ops, parallel, roofline, synthetic
    from pb_chime5_amd import _capi
    from pb_chime5_amd._capi import Context

    rank, local_rank = parallel.rank(), parallel.local_rank()
    n_dev = torch.cuda.device_count()
    shared_devices = n_dev < int(os.environ.get('LOCAL_WORLD_SIZE', world))
    # distinct physical packages first (a CPX-partitioned node exposes 8 logical devices per
    # GPU: LOCAL_RANK % count would put 8 ranks on the partitions of ONE package)
    device_index = _capi.default_device()
    torch.cuda.set_device(device_index)
    dist = None
    coll_device = 'cpu'
    backend = None
    coll_group = None
    if world > 1:
        # The process group only carries barriers and a few scalars (no collective on the data
        # path).  It is ALWAYS joined over gloo -- nothing of the run depends on RCCL coming
        # up --, and when every rank has a GPU of its own (the contract's launch) the timing
        # barriers and reductions go through an RCCL sub-group that is probed first: one
        # all-reduce on the device; if it fails on any rank, every rank stays on gloo and the
        # line says so (`rank_backend`, `rccl_ranks`).  GSS_BENCH_BACKEND=gloo|nccl forces one.
        want = os.environ.get('GSS_BENCH_BACKEND', 'gloo' if shared_devices else 'nccl')
        dist = parallel.init(backend='gloo')
        backend = 'gloo'

        def all_agree(flag):
            t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return float(t.item()) == 1.0
        # Every step of the probe is agreed over gloo BEFORE the next one: a rank that fails
        # alone must not leave the others inside an RCCL call (communicator set-up needs
        # every rank).  What remains is a failure inside the first all-reduce itself: the
        # probe group has a short timeout and waits in blocking mode, so that shows up as an
        # exception on every rank instead of a hang.
        if all_agree(want == 'nccl'):
            g = None
            try:
                from datetime import timedelta
                os.environ.setdefault('TORCH_NCCL_BLOCKING_WAIT', '1')
                g = dist.new_group(backend='nccl', timeout=timedelta(seconds=180))
            except Exception as e:           # noqa: BLE001 -- anything RCCL throws
                print(f'bench.py: rank {rank}: no RCCL sub-group ({type(e).__name__}: '
                      f'{str(e)[:200]})', file=sys.stderr)
            ok = all_agree(g is not None)
            if ok:
                try:
                    probe = torch.ones(1, dtype=torch.float64, device='cuda')
                    dist.all_reduce(probe, group=g)
                    torch.cuda.synchronize()
                    ok = float(probe.item()) == world
                except Exception as e:       # noqa: BLE001
                    print(f'bench.py: rank {rank}: RCCL probe failed ({type(e).__name__}: '
                          f'{str(e)[:200]}); staying on gloo', file=sys.stderr)
                    ok = False
                ok = all_agree(ok)
            if ok:
                coll_group, backend, coll_device = g, 'nccl', 'cuda'

    # threads started from here on (loaders, feeder, writer, the CPU baseline's pool excepted:
    # it resets its mask) run on the socket of this rank's GPU
    full_mask = os.sched_getaffinity(0)
    affinity = parallel.bind_to_gpu_numa(device_index)
    try:
        bus_id = _capi.device_pci_bus_id(device_index)
    except Exception:            # noqa: BLE001
        bus_id = None
    placement = [dict(rank=rank, device=device_index, pci_bus_id=bus_id, **affinity)]
    if dist is not None:
        gathered = [None] * world
        dist.all_gather_object(gathered, placement[0])
        placement = gathered
    # two ranks on one physical GPU although the node has a logical device for every rank:
    # the line says so instead of reporting one GPU's throughput as N GPUs'
    packages = [_capi.pci_package(p['pci_bus_id']) for p in placement if p.get('pci_bus_id')]
    gpus_shared_by_ranks = len(packages) - len(set(packages))
    if gpus_shared_by_ranks and not shared_devices and rank == 0:
        print(f'bench.py: WARNING: {world} ranks on {len(set(packages))} physical GPUs although '
              f'{n_dev} logical devices are visible: {placement}', file=sys.stderr)

    ctx = Context(device_index)
    _capi._DEFAULT_CTX[device_index] = ctx

    def gpu_sync():
        ctx.synchronize()
        torch.cuda.synchronize()
    # barriers and reductions around every timed region (ranks finish at different times: the
    # slowest rank's clock is the job's); CPU-tested with gloo in tests/test_parallel_gloo.py
    timer = parallel.RankTimer(dist, coll_group, coll_device, gpu_sync)
    barrier, max_over_ranks, sum_over_ranks = timer.barrier, timer.max, timer.sum
    gather_over_ranks = timer.gather

    params = ops.make_params(wpe=True, wpe_taps=WORKLOAD['wpe_taps'],
                             wpe_delay=WORKLOAD['wpe_delay'],
                             wpe_iterations=WORKLOAD['wpe_iterations'],
                             bss_iterations=WORKLOAD['bss_iterations'],
                             bss_iterations_post=WORKLOAD['bss_iterations_post'])
    ops._prepare_windows(ctx, params.stft_size, params.stft_shift)
    F = params.stft_size // 2 + 1

    # ------------------------------------------------------------------ config 3 session
    def config3_session(items):
        """All ranks pull the items of BASELINE configs[2] from the shared longest-first
        queue (parallel.split_managed), `inflight` utterances in flight per GPU, PCM16
        H2D and float64 D2H inside the wall clock."""
        t_gen = time.perf_counter()
        pool = Config3Pool(items, args.pool)
        t_gen = time.perf_counter() - t_gen
        pipe = ops.UtterancePipeline(params, depth=args.inflight, first_ctx=ctx)
        longest = int(np.argmax(pool.num_samples))
        run_session(pipe, [longest] * (args.inflight + 1), pool.item)     # warm-up: arenas sized
        barrier()
        t0 = time.perf_counter()
        mine = parallel.split_managed(range(items), costs=pool.num_samples,
                                      dynamic=not args.static)
        handled = []

        def get(i):
            handled.append(i)
            return pool.item(i)
        clock = {}
        run_session(pipe, mine, get, clock)
        local = time.perf_counter() - t0
        barrier()
        wall = max_over_ranks(time.perf_counter() - t0)
        per_rank = gather_over_ranks([local, len(handled), sum(pool.seconds(i) for i in handled),
                                      clock['queue_wait_s'], clock['host_prepare_s'],
                                      clock['enqueue_s'], clock['gpu_wait_s']])
        secs = sum_over_ranks(sum(pool.seconds(i) for i in handled))
        total = sum_over_ranks(len(handled))
        busiest = max_over_ranks(len(handled))
        slowest_local = max_over_ranks(local)
        pipe.close()
        assert int(total) == items, (total, items)
        return {
            'workload': (f'BASELINE.json configs[2]: {items} dev-shaped utterances (24 ch, K = 5, '
                         'core 0.5-15 s + 2 x 15 s context, T = 1906-2815), samples from '
                         f'{args.pool} seeded base recordings cut to each item\'s length; WPE '
                         'taps=10, 20 EM iterations, MVDR+BAN'),
            'sharding': ('static longest-first round-robin' if args.static else
                         'dynamic: shared counter in the launcher\'s TCP store, longest first '
                         '(parallel.split_managed)') if world > 1 else 'single rank',
            'items': items, 'utterances_in_flight_per_gpu': args.inflight,
            'includes': 'H2D (PCM16) + D2H per utterance, host-side slicing of the item',
            'utterance_seconds': secs, 'wall_s': wall,
            'value': secs / wall, 'unit': 'utterance-seconds/s',
            'ms_per_utterance': 1e3 * wall / items * world,
            'items_on_busiest_rank': int(busiest), 'slowest_rank_busy_s': slowest_local,
            'host_generation_s': t_gen,
            # one row per rank: where its host thread spent the session (a rank whose
            # gpu_wait_s is small and host_prepare_s + enqueue_s large is host bound)
            'per_rank': [dict(rank=r, busy_s=v[0], items=int(v[1]), utterance_seconds=v[2],
                              queue_wait_s=v[3], host_prepare_s=v[4], enqueue_s=v[5],
                              gpu_wait_s=v[6]) for r, v in enumerate(per_rank)],
        }

    # ------------------------------------------------------------------ config 4 stand-in
    def config4s_session():
        """BASELINE configs[3] stand-in (the CHiME-5 audio is not available): a synthetic
        session in the CHiME-5 layout at dev shape -- 24 per-channel PCM16 WAV files,
        `--session-utterances` utterances with 2 x 15 s of context -- on tmpfs / local disk,
        through Enhancer.enhance_session exactly as `python -m pb_chime5_amd.scripts.run with
        session_id=S02 multiarray=True` runs it: JSON database, annotation activity, WAV slice
        reads, PCM16 upload, enhancement, context trim, WAV writes; wall clock around all of it,
        all ranks pulling examples from the shared longest-first queue."""
        import shutil
        import tempfile
        from pb_chime5_amd.core import get_enhancer
        from pb_chime5_amd.synthetic_corpus import write_dev_shaped_corpus
        sessions = [x for x in args.sessions.split(',') if x]
        base = os.environ.get('GSS_BENCH_SCRATCH')
        if base is None:
            shm = Path('/dev/shm')
            ok = shm.is_dir() and os.access(shm, os.W_OK) and shutil.disk_usage(shm).free > 4 << 30
            base = str(shm) if ok else tempfile.gettempdir()
        tag = os.environ.get('MASTER_PORT', '0') if world > 1 else f'p{os.getpid()}'
        root = Path(base) / f'gss_bench_c4s_{tag}'
        t_gen = time.perf_counter()
        if rank == 0:
            shutil.rmtree(root, ignore_errors=True)
            write_dev_shaped_corpus(root / 'corpus', sessions,
                                    seconds=args.session_seconds / len(sessions),
                                    num_utterances=args.session_utterances // len(sessions))
        barrier()
        t_gen = time.perf_counter() - t_gen
        json_path = root / 'corpus' / 'chime5.json'
        multiarray = {'True': True, 'False': False}.get(args.multiarray, args.multiarray)
        kw = dict(database_path=str(json_path), multiarray=multiarray, context_samples=240000,
                  device_id=device_index)

        def obs_samples(ex):
            return max(ex['num_samples']['observation'].values())
        try:
            # warm-up on a second Enhancer (workspaces sized, code objects loaded, page cache
            # of the WAV files touched); the timed Enhancer below starts cold on the host side:
            # JSON parse, activity tracks from the annotations, file opens are in the wall clock
            warm = get_enhancer(**kw)
            warm.inflight, warm.loaders = args.inflight, args.loaders
            examples = sorted(warm.get_iterator(sessions), key=obs_samples)
            (root / f'warm{rank}' / 'dev').mkdir(parents=True)
            # the longest utterances of EVERY session: arenas and staging sized for both
            # channel counts
            warm_set = [ex for sid in sessions
                        for ex in [e for e in examples if e['session_id'] == sid][-(args.inflight + 1):]]
            warm._enhance_and_write(warm_set, root / f'warm{rank}')
            del warm
            enh = get_enhancer(**kw)
            enh.inflight, enh.loaders = args.inflight, args.loaders
            barrier()
            t0 = time.perf_counter()
            enh.enhance_session(sessions, root / 'out', audio_dir_exist_ok=True)
            local = time.perf_counter() - t0
            barrier()
            wall = max_over_ranks(time.perf_counter() - t0)
            clock = enh.session_clock
            keys = ('wall_s', 'examples', 'host_wait_s', 'enqueue_s', 'gpu_wait_s', 'write_wait_s',
                    'loader_busy_s', 'writer_busy_s')
            per_rank = gather_over_ranks([local] + [clock[k] for k in keys])
            written = sorted((root / 'out' / 'dev').glob('*.wav')) if rank == 0 else []
            if rank == 0:
                assert len(written) == len(examples), (len(written), len(examples))
                out_bytes = sum(p.stat().st_size for p in written)
                in_bytes = sum(p.stat().st_size for p in (root / 'corpus' / 'audio' / 'dev').glob('*.wav'))
            secs = sum(obs_samples(ex) for ex in examples) / SR
            core = sum(ex['num_samples_orig']['observation'][ex['reference_array']]
                       for ex in examples) / SR
        finally:
            barrier()
            if rank == 0:
                shutil.rmtree(root, ignore_errors=True)
        if rank != 0:
            return None
        return {
            'workload': (f'BASELINE.json configs[3] stand-in: synthetic CHiME-5-layout session(s) '
                         f'{"+".join(sessions)} (session_id=dev is S02 with 6 arrays AND S09 with 5: the '
                         f'channel count changes mid-run), {args.session_seconds:.0f} s in all x 24 / 20 '
                         f'per-channel PCM16 WAV files '
                         f'({in_bytes / 1e6:.0f} MB) under {base}, {len(examples)} utterances '
                         '(dev-shaped lengths, 2 x 15 s context, '
                         + {'True': 'multiarray=True: all arrays = 24 ch (S02) / 20 ch (S09)',
                            'False': 'multiarray=False, the reference default: the reference array = 4 ch',
                            'outer_array_mics': 'multiarray=outer_array_mics: 2 microphones of each '
                                                'of the 6 arrays = 12 ch'}[args.multiarray]
                         + ', K = 5), get_enhancer() defaults: WPE taps=10, 20 EM iterations, MVDR+BAN'),
            'multiarray': args.multiarray,
            'sessions': sessions,
            'utterances_per_session': {sid: sum(ex['session_id'] == sid for ex in examples)
                                       for sid in sessions},
            'driver': f'Enhancer.enhance_session({sessions}, out) as scripts/run.py calls it',
            'includes': ('JSON database + annotation activity, WAV slice reads into page-locked '
                         'int16 blocks, H2D, enhancement, D2H of the trimmed utterance, peak '
                         'normalisation + PCM16 WAV writes'),
            'sharding': ('dynamic: shared counter in the launcher\'s TCP store, longest first '
                         '(parallel.split_managed)') if world > 1 else 'single rank',
            'utterances': len(examples), 'utterances_in_flight_per_gpu': args.inflight,
            'loader_threads_per_rank': args.loaders,
            'utterance_seconds': secs, 'core_seconds_written': core,
            'wav_bytes_written': out_bytes, 'wall_s': wall,
            'value': secs / wall, 'unit': 'utterance-seconds/s',
            'ms_per_utterance': 1e3 * wall / len(examples) * world,
            'corpus_generation_s': t_gen,
            # one row per rank.  gpu_wait_s: blocked in pop() for the oldest utterance (the GPU is
            # the bottleneck when this dominates); host_wait_s: blocked on the loader threads;
            # loader_busy_s / writer_busy_s: summed busy time of those threads
            'per_rank': [dict(rank=r, busy_s=v[0], **{k: (int(x) if k == 'examples' else x)
                                                      for k, x in zip(keys, v[1:])})
                         for r, v in enumerate(per_rank)],
            'gpu_wait_share_of_wall': min(v[1 + keys.index('gpu_wait_s')] / wall for v in per_rank),
        }

    if args.config == '4s':
        block = config4s_session()
        if rank == 0:
            line = {
                'metric': 'utterance-seconds enhanced/sec (%sch, 20 EM iters), file-backed session' % {'True': 24, 'False': 4, 'outer_array_mics': 12}[args.multiarray],
                'value': block['value'], 'unit': 'utterance-seconds/s', 'n_gpus': args.gpus,
                'steps': block['utterances'], 'warmup': args.inflight + 1,
                'ms_per_step': 1e3 * block['wall_s'] / block['utterances'],
                'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
                'dtype': 'f64', 'data': 'synthetic',
                'config': {'workload': block['workload'], 'sharding': block['sharding'],
                           'parallelism': f'utterance-sharded x{args.gpus}'
                                          + (' (ranks share GPUs)' if shared_devices else '')},
                'ranks': world, 'rank_backend': backend if world > 1 else None,
                'rccl_ranks': world if backend == 'nccl' else 0, 'collectives_on_data_path': 0,
                'config4_standin': block,
            }
            emit(line)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    if args.config == '3':
        items = args.steps or 512
        block = config3_session(items)
        if rank == 0:
            line = {
                'metric': 'utterance-seconds enhanced/sec (24ch, 20 EM iters), sharded session',
                'value': block['value'], 'unit': 'utterance-seconds/s', 'n_gpus': args.gpus,
                'steps': items, 'warmup': args.inflight + 1,
                'ms_per_step': 1e3 * block['wall_s'] / items,
                'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
                'dtype': 'f64', 'data': 'synthetic',
                'config': {'workload': block['workload'], 'sharding': block['sharding'],
                           'parallelism': f'utterance-sharded x{args.gpus}'
                                          + (' (ranks share GPUs)' if shared_devices else '')},
                'ranks': world, 'rank_backend': backend if world > 1 else None,
                'rccl_ranks': world if backend == 'nccl' else 0, 'collectives_on_data_path': 0,
                'config3_sharded': block,
            }
            emit(line)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ------------------------------------------------------------------ config 2 headline
    workload = dict(WORKLOAD, name='2', description=(
        'BASELINE.json configs[1]: synthetic 24-mic (6 arrays x 4) 15 s '
        'utterance, 4 speakers + noise class, WPE taps=10 delay=2 iters=3, '
        '20 EM iterations + predict, MVDR-Souden + BAN; inputs resident in HBM'))
    if args.workload == '2':
        steps = args.steps or 20
        utt = synthetic.config2(seed=2 + rank, num_channels=WORKLOAD['num_channels'],
                                seconds=WORKLOAD['seconds'],
                                num_speakers=WORKLOAD['num_speakers'])
    else:
        # the other BASELINE shapes through the same timed section (their rocprofv3 passes are
        # taken with this command: one stream, one shape, back-to-back launches)
        args.only_headline = True
        utt, params, workload = other_workload(args.workload, synthetic, ops)
        ops._prepare_windows(ctx, params.stft_size, params.stft_shift)
        steps = args.steps or workload['steps']
    ctx_samples = utt.ex['start_orig']['original']
    end_samples = utt.ex['end']['original'] - utt.ex['end_orig']['original']
    resident = ops.ResidentUtterance(ctx, utt.obs, utt.activity_array, params)

    def step():
        resident.enqueue(utt.target_index, ctx_samples, end_samples)

    for _ in range(args.warmup):
        step()
    barrier()
    # Untimed pass with HIP events around EVERY launch: the per-kernel table and the
    # name of the dominant kernel.  (Two events per launch cost ~4 us of stream time
    # each; around all ~260 launches of an utterance that is 6 % of the step, so the
    # timed region below only carries events around the dominant kernel.)
    prof_all = profile_kernels(ctx, resident, utt, PROFILE_STEPS)
    dominant = max(prof_all, key=lambda k: prof_all[k]['ms'])
    if dist is not None:       # every rank times the same kernel
        names = sorted(prof_all)
        idx = torch.tensor([names.index(dominant)], device=coll_device)
        dist.broadcast(idx, src=0)
        dominant = names[int(idx.item())]
    ctx.profile_filter(dominant)
    ctx.profile_enable(True)
    ctx.profile_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
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
    elapsed_local = elapsed
    elapsed = max_over_ranks(elapsed)
    headline_rows = gather_over_ranks([elapsed_local])

    extras = not args.only_headline
    # ---- the same timed loop the way Enhancer.enhance_example runs an utterance: the caller has
    # said "one utterance at a time" and the WPE stage's frequencies go in two sets on two
    # streams (gss_set_utterances_in_flight(ctx, 1); same bits).  NOT the headline: overlapped
    # launches have no durations of their own, the kernel table and `roofline` are one-stream.
    one_at_a_time = None
    if extras or args.one_at_a_time:
        ctx.set_utterances_in_flight(1)
        for _ in range(2):
            step()
        barrier()
        t1 = time.perf_counter()
        for _ in range(steps):
            step()
        gpu_sync()
        e1 = max_over_ranks(time.perf_counter() - t1)
        ctx.set_utterances_in_flight(0)
        barrier()
        one_at_a_time = {'value': args.gpus * steps * utt.seconds / e1, 'unit': 'utterance-seconds/s',
                         'ms_per_step': 1e3 * e1 / steps,
                         'what': 'gss_set_utterances_in_flight(ctx, 1): the WPE stage as two sets of '
                                 'frequencies side by side on two streams, as Enhancer.enhance_example '
                                 'runs an utterance (bit-identical output)'}
    # ---- session mode on the same workload: 2 in flight, H2D + D2H inside the wall clock
    incl = None
    if extras:
        pipe = ops.UtterancePipeline(params, depth=args.inflight, first_ctx=ctx)
        pcm = to_pcm16(utt.obs)
        item = (pcm, utt.activity_array, utt.target_index, ctx_samples, ctx_samples)
        n2 = max(steps, 8)
        run_session(pipe, range(args.inflight + 1), lambda i: item)
        barrier()
        t2 = time.perf_counter()
        clock2 = {}
        run_session(pipe, range(n2), lambda i: item, clock2)
        e2_local = time.perf_counter() - t2
        barrier()
        e2 = max_over_ranks(e2_local)
        rows2 = gather_over_ranks([e2_local, clock2['enqueue_s'], clock2['gpu_wait_s']])
        pipe.close()
        incl = {'value': args.gpus * n2 * utt.seconds / e2, 'unit': 'utterance-seconds/s',
                'utterances_per_rank': n2, 'utterances_in_flight_per_gpu': args.inflight,
                'ms_per_utterance': 1e3 * e2 / n2,
                'per_rank': [dict(rank=r, wall_s=v[0], value=n2 * utt.seconds / v[0],
                                  enqueue_s=v[1], gpu_wait_s=v[2]) for r, v in enumerate(rows2)],
                'includes': 'H2D of the 24 x 240000 PCM16 samples + activity and D2H of the float64 '
                            'result for every utterance (pageable host memory)'}

    # ---- the other BASELINE configs, in-process (rank 0 of a single-GPU run only)
    configs = None
    em_loop = None
    if extras and rank == 0 and args.gpus == 1:
        configs = {}
        size2 = dict(F=F, T=resident.T, D=resident.D, K=resident.K)
        em_loop = {'D24': em_loop_entry(prof_all, PROFILE_STEPS, WORKLOAD['bss_iterations'],
                                        roofline=roofline, **size2)}

        def timed(name, u, p, nsteps, note):
            ms, res = time_resident(ctx, ops, u, p, nsteps)
            configs[name] = {'ms_per_utterance': ms, 'utterance_seconds': u.seconds,
                             'value': 1e3 * u.seconds / ms, 'unit': 'utterance-seconds/s',
                             'channels': res.D, 'frames': res.T, 'classes': res.K,
                             'f64_peak_frac': roofline.step_peak_frac(
                                 ms, F=F, T=res.T, D=res.D, K=res.K, taps=p.wpe_taps, N=res.N,
                                 wpe_iterations=p.wpe_iterations, iterations=p.bss_iterations,
                                 wpe=bool(p.wpe))['frac'],
                             'workload': note, 'mode': 'one stream, inputs resident in HBM'}
            return res
        timed('1', synthetic.config1(),
              ops.make_params(wpe=False, bss_iterations=5, bss_iterations_post=1), 10,
              'configs[0]: 4 mics, 5 s, 2 speakers + Noise, WPE off, 5 EM iterations, MVDR+BAN')
        configs['2'] = {'ms_per_utterance': 1e3 * elapsed / steps, 'utterance_seconds': utt.seconds,
                        'value': steps * utt.seconds / elapsed, 'unit': 'utterance-seconds/s',
                        'channels': resident.D, 'frames': resident.T, 'classes': resident.K,
                        'workload': 'configs[1] (the headline)', 'mode': 'one stream, inputs resident in HBM'}
        # the accuracy switch of the correlation (GSS_VARIANT=corr_blocked: chunk-wise accumulation
        # of R and P, read by the library on every call): what it costs on the headline
        os.environ['GSS_VARIANT'] = 'corr_blocked'
        try:
            ms_b, res_b = time_resident(ctx, ops, utt, params, 10)
            prof_b = profile_kernels(ctx, res_b, utt, PROFILE_STEPS)
        finally:
            del os.environ['GSS_VARIANT']
        configs['2-corr-blocked'] = {
            'ms_per_utterance': ms_b, 'utterance_seconds': utt.seconds,
            'value': 1e3 * utt.seconds / ms_b, 'unit': 'utterance-seconds/s',
            'wpe_corr_ms_per_launch': prof_b['wpe_corr']['ms'] / prof_b['wpe_corr']['calls'],
            'wpe_corr_ms_per_launch_default': prof_all['wpe_corr']['ms'] / prof_all['wpe_corr']['calls'],
            'workload': 'configs[1] with GSS_VARIANT=corr_blocked (default: off)',
            'what': 'R and P summed in 64-frame blocks (64 + T / 64 roundings per sum instead of T): the '
                    'distance from an extended-precision WPE falls from 1.8 - 2.0 x the oracle\'s own to '
                    '1.1 - 1.4 x (tests/test_gpu_stages.py, tools/wpe_accuracy.py)',
            'mode': 'one stream, inputs resident in HBM'}
        sr = SR
        n3 = 554490                          # synthetic.config3_item(0): 4.66 s core + 2 x 15 s
        iv3 = [(240000, n3 - 240000), (100000, 400000), (50000, 250000), (300000, 520000)]
        u3 = synthetic.make_utterance(1000, 24, n3, iv3, start_context=240000, end_context=240000,
                                      rir_taps=1024, noise=3e-2, fast=True)
        res3 = timed('3-item', u3, params, 5,
                     'one dev-shaped item of configs[2]: 24 ch, 34.7 s incl. 2 x 15 s context, as config 2')
        prof3 = profile_kernels(ctx, res3, u3, PROFILE_STEPS)
        em_loop['D24_T2169'] = em_loop_entry(prof3, PROFILE_STEPS, WORKLOAD['bss_iterations'],
                                             F=F, T=res3.T, D=res3.D, K=res3.K, roofline=roofline)
        configs['3-item']['kernel_ms_per_utterance'] = {
            k: round(v['ms'] / PROFILE_STEPS, 4) for k, v in sorted(prof3.items(), key=lambda kv: -kv[1]['ms'])}
        u1a = synthetic.make_utterance(1001, 4, n3, iv3, start_context=240000, end_context=240000,
                                       rir_taps=1024, noise=3e-2, fast=True)
        res4 = timed('3-item-one-array', u1a, params, 5,
                     'the reference default multiarray=False: one array (4 ch) of the same item')
        prof4 = profile_kernels(ctx, res4, u1a, PROFILE_STEPS)
        em_loop['D4'] = em_loop_entry(prof4, PROFILE_STEPS, WORKLOAD['bss_iterations'],
                                      F=F, T=res4.T, D=res4.D, K=res4.K, roofline=roofline)
        n5 = 120 * sr
        iv5 = [(50 * sr, 70 * sr), (10 * sr, 60 * sr), (40 * sr, 100 * sr), (65 * sr, 115 * sr)]
        u5 = synthetic.make_utterance(5, 12, n5, iv5, start_context=50 * sr, end_context=50 * sr,
                                      fast=True)
        timed('5', u5, ops.make_params(wpe=True, wpe_taps=10, wpe_delay=2, wpe_iterations=3,
                                       bss_iterations=40, bss_iterations_post=1, bf='gev_ban'), 3,
              'configs[4]: 120 s segment, 12 ch (outer_array_mics), WPE, 40 EM iterations, GEV + BAN')
        configs['4'] = {'status': 'corpus unavailable; file-backed stand-in at dev shape: see '
                                  '`config4_standin` (python bench.py --config 4s)',
                        'workload': 'configs[3]: CHiME-5 dev session S02 (needs the CHiME-5 audio and '
                                    'cache/chime5.json; `python -m pb_chime5_amd.scripts.run with '
                                    'session_id=S02 multiarray=True` is the harness hook)'}

    # ---- BASELINE configs[2] through the shared queue (all ranks)
    sharded = config3_session(args.items) if extras else None
    standin = config4s_session() if extras and not args.no_config4s else None

    if rank == 0:
        size = dict(F=F, T=resident.T, D=resident.D, K=resident.K, taps=workload['wpe_taps'],
                    N=resident.N, iterations=workload['bss_iterations'])
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
        traffic_file = REPO / 'profiles' / ('traffic.json' if args.workload == '2'
                                            else f'traffic_{args.workload}.json')
        if roof is not None and traffic_file.exists():
            # a PMC figure from an earlier rocprofv3 pass (tools/collect_profiles.sh), valid only
            # for the sources it was taken on: stale -> traffic stays null and the line says why
            try:
                table = json.loads(traffic_file.read_text())
                meta = table.get('__meta__', {})
                now = roofline.source_hashes()
                changed = [f for f in roofline.kernel_sources(dominant)
                           if meta.get('source_hashes', {}).get(f) != now.get(f)]
                if changed:
                    roof['traffic_stale'] = {
                        'file': f'profiles/{traffic_file.name}', 'sources_changed_since': changed,
                        'bytes_then': table.get(dominant, {}).get('bytes')}
                else:
                    roof['traffic'] = table.get(dominant, {}).get('bytes')
                    roof['traffic_note'] = (
                        f'HBM-side bytes per launch from profiles/{traffic_file.name} (rocprofv3 --pmc '
                        'FETCH_SIZE x2 + WRITE_SIZE, separate passes; taken on sources with the same '
                        'hashes as the loaded library\'s: ' + ', '.join(roofline.kernel_sources(dominant)) + ')')
            except Exception as e:      # noqa: BLE001
                roof['traffic_stale'] = {'error': f'{type(e).__name__}: {e}'}
        if args.workload != '2':
            # the EM loop of this shape against both roofs, from the same untimed pass
            em_loop = {f'D{resident.D}_T{resident.T}': em_loop_entry(
                prof_all, PROFILE_STEPS, workload['bss_iterations'], F=F, T=resident.T,
                D=resident.D, K=resident.K, roofline=roofline)}
        line = {
            'metric': 'utterance-seconds enhanced/sec/GPU (24ch, 20 EM iters)' if args.workload == '2'
                      else f'utterance-seconds enhanced/sec/GPU (workload {args.workload}: '
                           f'{resident.D}ch, {workload["bss_iterations"]} EM iters)',
            'value': args.gpus * steps * utt.seconds / elapsed,
            'unit': 'utterance-seconds/s',
            'n_gpus': args.gpus, 'steps': steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {
                'workload': workload['description'],
                'utterance_seconds': utt.seconds, 'channels': resident.D,
                'frames': resident.T, 'classes': resident.K,
                'utterances_per_step_per_gpu': 1,
                'parallelism': f'utterance-sharded x{args.gpus}'
                               + (' (ranks share GPUs)' if shared_devices else ''),
            },
            'realtime_factor_per_gpu': steps * utt.seconds / elapsed,
            # per rank (= per GPU when every rank has its own): the timed region of each rank
            'per_rank': [dict(rank=r, ms_per_step=1e3 * v[0] / steps, value=steps * utt.seconds / v[0])
                         for r, v in enumerate(headline_rows)],
            'ranks': world, 'rank_backend': backend if world > 1 else None,
            'rccl_ranks': world if backend == 'nccl' else 0,
            'collectives_on_data_path': 0,
            # what Enhancer.enhance_session does per GPU (two utterances in flight, PCM16 H2D +
            # float64 D2H inside the wall clock): the figure to quote per GPU for a session
            'value_one_at_a_time_api': one_at_a_time,
            'value_session': incl['value'] if incl else None,
            'value_session_per_gpu': incl['value'] / args.gpus if incl else None,
            'value_incl_pcie': incl['value'] if incl else None,
            'session_mode': incl,
            'roofline': roof,
            'kernels': kernels,
            'kernel_ms_per_utterance': {k: round(v['ms'] / PROFILE_STEPS, 4) for k, v in
                                        sorted(prof_all.items(), key=lambda kv: -kv[1]['ms'])},
            'kernels_note': (f'per-kernel table from an untimed pass of {PROFILE_STEPS} steps with HIP '
                             'events around every launch; `roofline` is the dominant kernel timed '
                             'inside the timed region; frac_of_roof prices the minimum flops / bytes '
                             'of each kernel\'s formulation (pb_chime5_amd/roofline.py)'),
            'device_ms_per_step': total_ms / PROFILE_STEPS,
            # the whole step against the chip's f64 peak: executed flops / duration / 78.6 TFLOP/s
            'f64_peak_frac': roofline.step_peak_frac(
                1e3 * elapsed / steps, F=F, T=resident.T, D=resident.D, K=resident.K,
                taps=params.wpe_taps, N=resident.N, wpe_iterations=params.wpe_iterations,
                iterations=params.bss_iterations, wpe=bool(params.wpe)),
            'workspace_bytes': ctx.workspace_bytes(),
            'em_loop': em_loop,
            'configs': configs,
            'config3_sharded': sharded,
            'config4_standin': standin,
        }
        if args.gpus == 1 and extras and not args.no_cpu_baseline:
            # the reference's mpiexec workers are not tied to the GPU's socket: all host cores
            try:
                os.sched_setaffinity(0, full_mask)
            except OSError:
                pass
            line['cpu_baseline'] = cpu_baseline(utt, args.cpu_bins, args.cpu_workers)
            line['speedup_vs_cpu_all_cores'] = line['value'] / line['cpu_baseline']['value']
        emit(line)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
