"""Per-utterance guided source separation with the reference's Python surface
(/root/reference/pb_chime5/core.py), running on MI355X.

Legend (as in the reference):  n, N time | t, T frame | f, F frequency |
d, D channel | a, A array | k, K class (speakers + ``Noise``).

Kept from the reference: ``get_enhancer(**same kwargs)``, the ``WPE`` / ``GSS`` /
``Beamformer`` / ``Enhancer`` blocks with their call signatures, shapes, dtypes
(complex128 / float64 at the Python edge) and exception types,
``start_end_context_frames`` and the ``debug=True`` contract (intermediates kept
on the block).  Different by design: the numeric work is done by hand-written HIP
kernels behind ``include/gss_hip.h``; ``Enhancer.enhance_observation`` uses the
fused device pipeline (one H2D of the time signal, one D2H of the result) unless
a block was replaced or ``fused=False`` is passed; work is distributed over GPUs
by ``pb_chime5_amd.parallel`` instead of ``dlp_mpi``.
"""
from dataclasses import dataclass, field
from pathlib import Path

import numpy as np

from pb_chime5_amd import mapping, ops
from pb_chime5_amd.database.chime5 import activity_time_to_frequency
from pb_chime5_amd.io import dump_audio, load_audio
from pb_chime5_amd.utils.numpy_utils import morph


@dataclass
class WPE:
    """core.py:41-88 -> nara_wpe.wpe.wpe_v8."""
    taps: int
    delay: int
    iterations: int
    psd_context: int

    def __call__(self, Obs, stack=None, debug=False):
        kw = dict(taps=self.taps, delay=self.delay, iterations=self.iterations,
                  psd_context=self.psd_context)
        if Obs.ndim == 3:
            assert stack is None, stack
            Obs = ops.wpe_dtf(Obs, **kw)
        elif Obs.ndim == 4:
            if stack is True:
                _A = Obs.shape[0]
                Obs = morph('ACTF->A*CTF', Obs)
                Obs = ops.wpe_dtf(Obs, **kw)
                Obs = morph('A*CTF->ACTF', Obs, A=_A)
            elif stack is False:
                Obs = np.array([ops.wpe_dtf(o, **kw) for o in Obs])
            else:
                raise NotImplementedError(stack)
        else:
            raise NotImplementedError(Obs.shape)
        if debug:
            self.locals = locals()
        return Obs


def default_database_path():
    """The reference looks for ``<git root>/cache/chime5.json`` (core.py:20,98); here
    ``$PB_CHIME5_JSON`` if set, else ``cache/chime5.json`` under the working directory."""
    import os
    return os.environ.get('PB_CHIME5_JSON', str(Path('cache') / 'chime5.json'))


@dataclass
class Activity:
    """core.py:91-141.  ``activity[session_id][array][speaker][start:stop] -> bool``.
    ``type='annotation'`` derives the tracks from the utterance boundaries in the CHiME-5
    JSON (one session cached, like the reference's ``lru_cache(1)``); ``type='path'``
    reads the reference's per-session pickles; ``store`` plugs in a ready dict."""
    type: str = 'annotation'
    garbage_class: bool = False
    database_path: str = None
    path: str = None
    store: dict = None

    # sessions kept at a time: the reference keeps one (lru_cache(1), core.py:104); a rank of a
    # multi-session run ('dev' = S02 + S09, longest first) alternates between a few
    _CACHE_SESSIONS = 4

    def __post_init__(self):
        import threading
        self._db = None
        self._cache = {}                 # session_id -> tracks, insertion-ordered (LRU)
        self._cache_lock = threading.Lock()

    @property
    def db(self):
        if self._db is None:
            from pb_chime5_amd.database.chime5.database import Chime5
            self._db = Chime5(self.database_path or default_database_path())
        return self._db

    def _annotation_activity(self, session_id):
        from pb_chime5_amd.activity import get_activity
        return get_activity(
            iterator=self.db.get_datasets(session_id), perspective='array',
            garbage_class=self.garbage_class, dtype=bool,
            use_ArrayIntervall=True)[session_id]

    def _cached_annotation(self, session_id):
        """Called from every loader thread of a session (`Enhancer._prepare_into`): the look-up,
        the computation and the hand-out of ONE session's tracks happen under a lock, and the
        caller gets the object it asked for, never "whatever the cache holds now"."""
        with self._cache_lock:
            tracks = self._cache.pop(session_id, None)
            if tracks is None:
                tracks = self._annotation_activity(session_id)
                while len(self._cache) >= self._CACHE_SESSIONS:
                    self._cache.pop(next(iter(self._cache)))
            self._cache[session_id] = tracks      # most recently used last
            return tracks

    def __getitem__(self, session_id):
        if self.store is not None:
            return self.store[session_id]
        if self.type == 'annotation':
            return self._cached_annotation(session_id)
        if self.type == 'path':
            with open(Path(self.path) / f'{session_id}.pkl', 'rb') as fd:
                return _ReferenceUnpickler(fd).load()
        raise ValueError(self.type)


class _ReferenceUnpickler(__import__('pickle').Unpickler):
    """Activity pickles written with the reference (core.py:135-139, e.g. the alignment-based
    activity of its sacred runs) name classes of the `pb_chime5` package, above all
    pb_chime5.utils.intervall_array.ArrayIntervall; they resolve to their counterparts here
    (same attributes), so such a file loads without the reference installed."""

    def find_class(self, module, name):
        if module == 'pb_chime5' or module.startswith('pb_chime5.'):
            module = 'pb_chime5_amd' + module[len('pb_chime5'):]
        return super().find_class(module, name)


@dataclass
class GSS:
    """core.py:144-214 -> CACGMMTrainer.fit / predict for every frequency."""
    iterations: int
    iterations_post: int
    verbose: bool = True

    def __call__(self, Obs, acitivity_freq, debug=False):
        posterior = ops.cacgmm_posteriors(
            Obs, acitivity_freq, iterations=self.iterations,
            iterations_post=self.iterations_post)
        if debug:
            initialization = np.asarray(acitivity_freq, dtype=np.float64)
            initialization = np.where(initialization == 0, 1e-10, initialization)
            initialization = initialization / np.sum(initialization, keepdims=True, axis=0)
            source_active_mask = np.asarray(acitivity_freq, dtype=bool)
            self.locals = locals()
        return posterior


def _original(value):
    # CHiME-5 examples carry {'original': ..., 'observation': {...}} (core.py:218-219),
    # CHiME-6 examples plain integers (core_chime6.py:215-216)
    return value['original'] if isinstance(value, dict) else value


def start_end_context_samples(ex):
    """The sample counts core.py:218-222 derives from ``ex`` (asserted >= 0)."""
    start_context_samples = _original(ex['start_orig']) - _original(ex['start'])
    end_context_samples = _original(ex['end']) - _original(ex['end_orig'])
    assert start_context_samples >= 0, (start_context_samples, ex)
    assert end_context_samples >= 0, (end_context_samples, ex)
    return start_context_samples, end_context_samples


def start_end_context_frames(ex, stft_size, stft_shift, stft_fading):
    """core.py:217-238."""
    start, end = start_end_context_samples(ex)
    return (
        ops.samples_to_stft_frames(start, stft_size, stft_shift, fading=stft_fading),
        ops.samples_to_stft_frames(end, stft_size, stft_shift, fading=stft_fading),
    )


@dataclass
class Beamformer:
    """core.py:241-278."""
    type: str
    postfilter: str

    def __call__(self, Obs, target_mask, distortion_mask, debug=False):
        bf = self.type
        if bf == 'mvdrSouden_ban':
            from pb_chime5_amd.speech_enhancement.beamforming_wrapper import (
                beamform_mvdr_souden_from_masks)
            X_hat = beamform_mvdr_souden_from_masks(
                Y=Obs, X_mask=target_mask, N_mask=distortion_mask, ban=True)
        elif bf == 'gev_ban':
            # not selectable in the reference's Beamformer.__call__ (core.py:246-266); the
            # GEV code path exists beside it (beamforming_wrapper.py:192-208)
            from pb_chime5_amd.speech_enhancement.beamforming_wrapper import (
                beamform_gev_from_masks)
            X_hat = beamform_gev_from_masks(Y=Obs, X_mask=target_mask, N_mask=distortion_mask,
                                            ban=True)
        elif bf == 'ch2':
            X_hat = Obs[2]
        elif bf == 'sum':
            X_hat = np.sum(Obs, axis=0)
        else:
            raise NotImplementedError(bf)

        if self.postfilter is None:
            pass
        elif self.postfilter == 'mask_mul':
            X_hat = X_hat * target_mask
        else:
            raise NotImplementedError(self.postfilter)
        if debug:
            self.locals = locals()
        return X_hat


@dataclass
class Enhancer:
    """core.py:281-571."""
    wpe_block: WPE
    activity: Activity
    gss_block: GSS
    bf_block: Beamformer

    bf_drop_context: bool

    stft_size: int
    stft_shift: int
    stft_fading: bool

    context_samples: int
    multiarray: bool
    reference_array: [None, str]

    device_id: int = None
    iterator_factory: object = field(default=None, repr=False)
    inflight: int = 2        # utterances kept in flight per GPU by enhance_session
    loaders: int = 3         # host threads that read the next examples' audio ahead of the GPU

    # ------------------------------------------------------------------ STFT
    def stft(self, x):
        return ops.stft(x, size=self.stft_size, shift=self.stft_shift,
                        fading=self.stft_fading, ctx=self._ctx())

    def istft(self, X):
        return ops.istft(X, size=self.stft_size, shift=self.stft_shift,
                         fading=self.stft_fading, ctx=self._ctx())

    def _ctx(self):
        from pb_chime5_amd._capi import default_context
        return default_context(self.device_id)

    # ------------------------------------------------------------------ sessions
    @property
    def db(self):
        return self.activity.db

    def get_iterator(self, session_id):
        """core.py:323-331.  ``iterator_factory(session_ids, context_samples)`` (an
        addition) replaces the JSON database as the example source when given."""
        if self.iterator_factory is not None:
            return self.iterator_factory(session_id, self.context_samples)
        return self.db.get_iterator_for_session(
            session_id, audio_read=False, adjust_times=True,
            drop_unknown_target_speaker=True, context_samples=self.context_samples,
            equal_start_context=True)

    def enhance_session(self, session_ids, audio_dir, dataset_slice=False,
                        audio_dir_exist_ok=False):
        """core.py:333-394; examples are sharded over the visible GPUs by
        pb_chime5_amd.parallel when more than one process is running."""
        from pb_chime5_amd import parallel
        audio_dir = Path(audio_dir)
        it = self.get_iterator(session_ids)

        if parallel.is_master():
            audio_dir.mkdir(exist_ok=audio_dir_exist_ok)
            for dataset in set(mapping.session_to_dataset.values()):
                (audio_dir / dataset).mkdir(exist_ok=audio_dir_exist_ok)
        parallel.barrier()

        if dataset_slice is not False:
            if dataset_slice is True:
                it = it[:2]
            elif isinstance(dataset_slice, int):
                it = it[:dataset_slice]
            elif isinstance(dataset_slice, slice):
                it = it[dataset_slice]
            else:
                raise ValueError(dataset_slice)

        costs = None
        if parallel.world_size() > 1:
            # longest first, so that the last utterances handed out are short ones
            def samples(tree):
                if isinstance(tree, dict):
                    return max((samples(v) for v in tree.values()), default=0)
                return int(tree)
            costs = [samples(ex['num_samples']) for ex in it]
        self._enhance_and_write(parallel.split_managed(it, costs=costs), audio_dir)
        # (split_managed ends without a barrier: this rank's pipeline has drained by now)
        parallel.barrier()

    def _write(self, ex, x_hat, audio_dir):
        if not np.all(np.isfinite(x_hat)):
            # Same outcome as the reference: e.g. an utterance so close to the end of the
            # recording that the nominal end context (core.py:217-222 takes it from the
            # example, not from the audio actually read) zeroes every frame of the target
            # mask -> Phi_X = 0 -> 0 / 0 in the blind analytic normalisation.
            import warnings
            warnings.warn(f'{ex.get("example_id")}: the enhanced signal is not finite')
        dataset = mapping.session_to_dataset[ex['session_id']]
        if x_hat.ndim == 1:
            dump_audio(x_hat, Path(audio_dir) / f'{dataset}' / f'{ex["example_id"]}.wav')
        else:
            raise NotImplementedError(x_hat.shape)

    def _enhance_and_write(self, examples, audio_dir):
        """Enhance the examples and write ``audio_dir/<dataset>/<example_id>.wav``.
        With the fused device pipeline ``self.inflight`` (default 2) utterances are kept
        in flight on separate HIP streams: the host loads the next example's audio
        while the GPU works, and one utterance's latency-bound kernels overlap the
        other's compute-bound ones.  Results are identical to the one-at-a-time loop."""
        if self.inflight <= 1 or not self._fusable():
            for ex in examples:
                try:
                    self._write(ex, self.enhance_example(ex), audio_dir)
                except Exception:
                    print('ERROR: Failed example:', ex.get('example_id'))
                    raise
            return
        # Host side of the session (replaces core.py:363-392 + io/audioread.py): `loaders`
        # threads prepare the next examples -- WAV slices by preadv straight into page-locked
        # (D, N) int16 rows, activity tracks sliced into uint8 rows; no float conversion, no
        # stacking copies (the STFT kernel converts the PCM) --, this thread starts the DMAs
        # and the kernels (nothing in it waits for the GPU except pop()), fetches only the
        # samples that survive the context trim, and a writer thread normalises and writes the
        # WAV files.  File reads, memcpy and the ctypes calls release the GIL.
        import threading
        import time
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        from pb_chime5_amd.io.wav_slices import WavSliceReader

        loaders = max(int(self.loaders), 1)
        pipe = ops.UtterancePipeline(self._params(), depth=self.inflight, first_ctx=self._ctx(),
                                     staging_sets=self.inflight + loaders + 1)
        reader = WavSliceReader()
        clock = self.session_clock = dict(
            examples=0, loader_threads=loaders, wall_s=0.0, host_wait_s=0.0, enqueue_s=0.0,
            gpu_wait_s=0.0, write_wait_s=0.0, loader_busy_s=0.0, writer_busy_s=0.0)
        lock = threading.Lock()
        t_begin = time.perf_counter()

        def prepare(ex):
            staging = pipe.acquire_staging()
            if staging is None:
                raise RuntimeError('the session was aborted')
            t0 = time.perf_counter()
            try:
                meta = self._prepare_into(ex, staging, reader)
            except BaseException:
                pipe.release_staging(staging)
                raise
            with lock:
                clock['loader_busy_s'] += time.perf_counter() - t0
            return (staging,) + meta

        def write(ex, x_hat):
            t0 = time.perf_counter()
            self._write(ex, x_hat, audio_dir)
            with lock:
                clock['writer_busy_s'] += time.perf_counter() - t0

        loader = ThreadPoolExecutor(max_workers=loaders, thread_name_prefix='gss-loader')
        writer = ThreadPoolExecutor(max_workers=1, thread_name_prefix='gss-writer')
        ahead = deque()
        writes = deque()
        source = iter(examples)

        def refill():
            while len(ahead) < loaders:
                try:
                    ex = next(source)
                except StopIteration:
                    return
                ahead.append((ex, loader.submit(prepare, ex)))

        def pop_and_write():
            t0 = time.perf_counter()
            done, x_hat = pipe.pop()
            clock['gpu_wait_s'] += time.perf_counter() - t0
            writes.append((done, writer.submit(write, done, x_hat)))
            while len(writes) > 8 or (writes and writes[0][1].done()):
                ex_w, fut = writes.popleft()
                t0 = time.perf_counter()
                try:
                    fut.result()
                except Exception:
                    print('ERROR: Failed example:', ex_w.get('example_id'))
                    raise
                clock['write_wait_s'] += time.perf_counter() - t0

        try:
            refill()
            while ahead:
                ex, future = ahead.popleft()
                t0 = time.perf_counter()
                try:
                    prepared = future.result()
                except Exception:
                    print('ERROR: Failed example:', ex.get('example_id'))
                    raise
                clock['host_wait_s'] += time.perf_counter() - t0
                refill()
                if pipe.full():
                    pop_and_write()
                t0 = time.perf_counter()
                try:
                    pipe.enqueue_staged(ex, *prepared)
                except BaseException:
                    pipe.release_staging(prepared[0])
                    raise
                clock['enqueue_s'] += time.perf_counter() - t0
                clock['examples'] += 1
            while len(pipe):
                pop_and_write()
            t0 = time.perf_counter()
            while writes:
                ex_w, fut = writes.popleft()
                try:
                    fut.result()
                except Exception:
                    print('ERROR: Failed example:', ex_w.get('example_id'))
                    raise
            clock['write_wait_s'] += time.perf_counter() - t0
        finally:
            # wake loader threads that wait for a free staging set (after an error in this
            # thread nobody pops any more); their examples fail with the RuntimeError below
            for _ in range(loaders):
                pipe.release_staging(None)
            loader.shutdown(wait=True, cancel_futures=True)
            writer.shutdown(wait=True, cancel_futures=True)
            pipe.close()
            reader.close()
            clock['wall_s'] = time.perf_counter() - t_begin

    # -- the clock of an example: overridden by the CHiME-6 front doors (one clock for all)
    def _audio_span(self, ex, array):
        return ex['start']['observation'][array], ex['end']['observation'][array]

    def _activity_span(self, ex):
        """(speaker -> track, start, nominal end) of the tracks that guide this example."""
        reference_array = self._reference_array(ex)
        start, end = self._audio_span(ex, reference_array)
        return self.activity[ex['session_id']][reference_array], start, end

    def _keep_range(self, ex):
        """The samples of x_hat that `_trim_context` keeps (None: all)."""
        if self.context_samples <= 0:
            return None
        reference_array = self._reference_array(ex)
        keep_from = (ex['start_orig']['observation'][reference_array]
                     - ex['start']['observation'][reference_array])
        return keep_from, keep_from + ex['num_samples_orig']['observation'][reference_array]

    fast_loader = True      # False: always go through _prepare_example (RTTM front door)

    def _prepare_into(self, ex, staging, reader):
        """Host side of enhance_example (core.py:396-490) for the session driver: fills
        ``staging.obs`` (D, N) int16 with the PCM samples of the selected channels (arrays cut
        to the shortest) and ``staging.act`` (K, N_act) uint8 with the activity tracks; returns
        (target index, start context, end context, kept sample range).  One mono file per
        microphone (CHiME-5 / 6) is read slice by slice straight into the rows; anything else
        goes through `_prepare_example` and one copy."""
        speaker_id = ex['speaker_id']
        plan = None
        if self.fast_loader:
            observation = ex['audio_path']['observation']
            if self.multiarray is False:
                arrays, select = [self._reference_array(ex)], None
            elif self.multiarray is True:
                arrays, select = sorted(observation.keys()), None
            elif self.multiarray == 'outer_array_mics':
                arrays, select = sorted(observation.keys()), (0, -1)
            elif self.multiarray == 'first_array_mics':
                arrays, select = sorted(observation.keys()), (0,)
            else:
                raise ValueError(self.multiarray)
            plan, lengths = [], []
            for array in arrays:
                paths = observation[array]
                start, stop = self._audio_span(ex, array)
                if not isinstance(paths, (list, tuple)) or any(
                        reader.info(p).channels != 1 for p in paths):
                    plan = None
                    break
                n_array = {reader.slice_length(p, start, stop) for p in paths}
                assert len(n_array) == 1, (array, n_array)   # np.array([...]) needs equal lengths
                lengths.append(n_array.pop())
                chosen = paths if select is None else [paths[i] for i in select]
                plan.extend((p, start) for p in chosen)

        if plan is None:
            obs, ex_array_activity, _ = self._prepare_example(ex, dtype=np.int16)
            keys = tuple(ex_array_activity.keys())
            activity = np.array(list(ex_array_activity.values()))
            rows, act = staging.shape(*obs.shape, *activity.shape)
            rows[...] = obs
            act[...] = activity != 0
        else:
            tracks, a, b = self._activity_span(ex)
            keys = tuple(tracks.keys())
            spans = [(a, min(b, len(arr))) for arr in tracks.values()]
            n_act = {hi - lo for lo, hi in spans}
            assert len(n_act) == 1, n_act                    # np.array(list(...)) likewise
            rows, act = staging.shape(len(plan), min(lengths), len(keys), n_act.pop())
            for row, (path, start) in zip(rows, plan):
                reader.read_into(path, start, row)
            for row, arr, (lo, hi) in zip(act, tracks.values(), spans):
                if hasattr(arr, 'slice_into'):
                    arr.slice_into(lo, hi, row)
                else:
                    row[:] = np.asarray(arr[lo:hi]) != 0

        start_ctx = end_ctx = 0
        if self.bf_drop_context:
            start_ctx, end_ctx = start_end_context_samples(ex)
        return keys.index(speaker_id), start_ctx, end_ctx, self._keep_range(ex)

    # ------------------------------------------------------------------ examples
    def enhance_example(self, ex, debug=False):
        """core.py:396-512."""
        obs, ex_array_activity, speaker_id = self._prepare_example(ex)
        x_hat = self.enhance_observation(
            obs, ex_array_activity=ex_array_activity, speaker_id=speaker_id, ex=ex,
            debug=debug)
        x_hat = self._trim_context(x_hat, ex)
        if debug:
            self.enhance_example_locals = dict(
                ex=ex, obs=obs, ex_array_activity=ex_array_activity, speaker_id=speaker_id,
                x_hat=x_hat)
        return x_hat

    def _reference_array(self, ex):
        reference_array = self.reference_array
        if reference_array is None:
            try:
                reference_array = ex['reference_array']
            except KeyError:
                raise RuntimeError(
                    'Failed to get the "reference_array" from the example.\n'
                    'Probably you tried to enhance the "train" dataset.\n'
                    'Train has no "reference_array".\n'
                    'You can set a "reference_array" with get_enhancer('
                    'reference_array="U06").\n'
                    'In case of multiarray, the reference array is used for the '
                    'projection of the human annotations.') from None
        return reference_array

    def _prepare_example(self, ex, dtype=np.float64):
        """Host side of enhance_example (core.py:396-490): activity slices of the
        reference array, channel selection, arrays cut to the shortest.  ``dtype=np.int16``
        keeps the PCM samples as stored (the session driver converts on the device)."""
        session_id = ex['session_id']
        reference_array = self._reference_array(ex)
        speaker_id = ex['speaker_id']

        array_start = ex['start']['observation'][reference_array]
        array_end = ex['end']['observation'][reference_array]
        ex_array_activity = {
            k: arr[array_start:min(array_end, len(arr))]
            for k, arr in self.activity[session_id][reference_array].items()
        }

        def load_arrays(select):
            arrays = [
                load_audio(ex['audio_path']['observation'][array],
                           start=ex['start']['observation'][array],
                           stop=ex['end']['observation'][array], dtype=dtype)
                for array in sorted(ex['audio_path']['observation'].keys())
            ]
            # The context does not consider the end of an utterance: arrays can
            # differ in length, cut to the shortest.
            assert {v.ndim for v in arrays} == {2}, [v.shape for v in arrays]
            time_length = min(v.shape[-1] for v in arrays)
            return morph('ACN->A*CN', np.array(
                [select(v)[..., :time_length] for v in arrays]))

        if self.multiarray is True:
            obs = load_arrays(lambda v: v)
        elif self.multiarray == 'outer_array_mics':
            obs = load_arrays(lambda v: v[(0, -1), :])
        elif self.multiarray == 'first_array_mics':
            obs = load_arrays(lambda v: v[(0,), :])
        elif self.multiarray is False:
            obs = load_audio(ex['audio_path']['observation'][reference_array],
                             start=ex['start']['observation'][reference_array],
                             stop=ex['end']['observation'][reference_array], dtype=dtype)
        else:
            raise ValueError(self.multiarray)
        return obs, ex_array_activity, speaker_id

    def _trim_context(self, x_hat, ex):
        """core.py:500-505: cut the enhanced signal back to the original utterance."""
        if self.context_samples > 0:
            reference_array = self._reference_array(ex)
            start_orig = ex['start_orig']['observation'][reference_array]
            start = ex['start']['observation'][reference_array]
            start_context = start_orig - start
            num_samples_orig = ex['num_samples_orig']['observation'][reference_array]
            x_hat = x_hat[..., start_context:start_context + num_samples_orig]
        return x_hat

    # ------------------------------------------------------------------ the hot path
    def _fusable(self):
        return (
            (self.wpe_block is None or type(self.wpe_block) is WPE)
            and type(self.gss_block) is GSS and type(self.bf_block) is Beamformer
        )

    def _params(self):
        w = self.wpe_block
        return ops.make_params(
            stft_size=self.stft_size, stft_shift=self.stft_shift,
            stft_fading=self.stft_fading, wpe=w is not None,
            wpe_taps=w.taps if w else 10, wpe_delay=w.delay if w else 2,
            wpe_iterations=w.iterations if w else 3,
            wpe_psd_context=w.psd_context if w else 0,
            bss_iterations=self.gss_block.iterations,
            bss_iterations_post=self.gss_block.iterations_post,
            bf_drop_context=self.bf_drop_context, bf=self.bf_block.type,
            postfilter=self.bf_block.postfilter)

    def enhance_observation(self, obs, ex_array_activity, speaker_id, ex=None,
                            debug=False, fused=None):
        """core.py:514-571.  obs (D,N) float64, ex_array_activity dict
        speaker -> bool (N,), returns x_hat (N',) float64."""
        if fused is None:
            fused = self._fusable()
        if not fused:
            return self._enhance_observation_blocks(obs, ex_array_activity, speaker_id,
                                                    ex, debug)
        target_speaker_index = tuple(ex_array_activity.keys()).index(speaker_id)
        activity = np.array(list(ex_array_activity.values()))
        start_ctx = end_ctx = 0
        if self.bf_drop_context:
            start_ctx, end_ctx = start_end_context_samples(ex)
        params = self._params()     # raises NotImplementedError for unknown bf / postfilter
        # one utterance at a time (this method is the loop body of core.py:363-392): the fused
        # call may put half of the WPE stage's frequencies on the context's second stream
        ctx = self._ctx()
        ctx.set_utterances_in_flight(1)
        try:
            res = ops.enhance_observation(
                obs, activity, target_speaker_index, start_ctx, end_ctx, params=params,
                debug=debug, ctx=ctx)
        finally:
            ctx.set_utterances_in_flight(0)
        if not debug:
            return res
        x_hat, details = res
        Obs = details['Obs']
        acitivity_freq = details['acitivity_freq']
        target_mask = details['target_mask']
        distortion_mask = details['distortion_mask']
        X_hat = details['X_hat']
        masks = details['posterior'].copy()
        if self.bf_drop_context:
            start_context_frames, end_context_frames = start_end_context_frames(
                ex, self.stft_size, self.stft_shift, self.stft_fading)
            masks[:, :start_context_frames, :] = 0
            if end_context_frames > 0:
                masks[:, -end_context_frames:, :] = 0
        self.enhance_observation_locals = locals()
        return x_hat

    def _enhance_observation_blocks(self, obs, ex_array_activity, speaker_id, ex, debug):
        """Block-by-block path with the reference's control flow (one device
        round trip per block); used when a block was swapped out."""
        Obs = self.stft(obs)
        if self.wpe_block is not None:
            Obs = self.wpe_block(Obs, debug=debug)
        acitivity_freq = activity_time_to_frequency(
            np.array(list(ex_array_activity.values())),
            stft_window_length=self.stft_size, stft_shift=self.stft_shift,
            stft_fading=self.stft_fading, stft_pad=True)
        masks = self.gss_block(Obs, acitivity_freq, debug=debug)
        if self.bf_drop_context:
            start_context_frames, end_context_frames = start_end_context_frames(
                ex, stft_size=self.stft_size, stft_shift=self.stft_shift,
                stft_fading=self.stft_fading)
            masks[:, :start_context_frames, :] = 0
            if end_context_frames > 0:
                masks[:, -end_context_frames:, :] = 0
        target_speaker_index = tuple(ex_array_activity.keys()).index(speaker_id)
        target_mask = masks[target_speaker_index]
        distortion_mask = np.sum(np.delete(masks, target_speaker_index, axis=0), axis=0)
        X_hat = self.bf_block(Obs, target_mask=target_mask,
                              distortion_mask=distortion_mask, debug=debug)
        x_hat = self.istft(X_hat)
        if debug:
            self.enhance_observation_locals = locals()
        return x_hat


def get_enhancer(
    multiarray=False,
    reference_array=None,
    context_samples=240000,

    wpe=True,
    wpe_tabs=10,
    wpe_delay=2,
    wpe_iterations=3,
    wpe_psd_context=0,

    activity_type='annotation',
    activity_path=None,
    activity_garbage_class=True,

    stft_size=1024,
    stft_shift=256,
    stft_fading=True,

    bss_iterations=20,
    bss_iterations_post=1,

    bf_drop_context=True,

    bf='mvdrSouden_ban',
    postfilter=None,

    database_path=None,

    activity_store=None,
    iterator_factory=None,
    device_id=None,
):
    """core.py:574-637 (same keyword arguments and defaults; ``activity_store``,
    ``iterator_factory`` and ``device_id`` are additions)."""
    assert wpe is True or wpe is False, wpe
    assert activity_path is None or activity_type == 'path', (activity_path, activity_type)

    return Enhancer(
        multiarray=multiarray,
        reference_array=reference_array,
        context_samples=context_samples,
        wpe_block=WPE(taps=wpe_tabs, delay=wpe_delay, iterations=wpe_iterations,
                      psd_context=wpe_psd_context) if wpe else None,
        activity=Activity(type=activity_type, garbage_class=activity_garbage_class,
                          path=activity_path, database_path=database_path,
                          store=activity_store),
        gss_block=GSS(iterations=bss_iterations, iterations_post=bss_iterations_post,
                      verbose=False),
        bf_drop_context=bf_drop_context,
        bf_block=Beamformer(type=bf, postfilter=postfilter),
        stft_size=stft_size,
        stft_shift=stft_shift,
        stft_fading=stft_fading,
        device_id=device_id,
        iterator_factory=iterator_factory,
    )
