"""CHiME-6 JSON front door: the surface of /root/reference/pb_chime5/core_chime6.py.

CHiME-6 recordings are synchronised across arrays, so an example carries ONE clock:
``ex['start']`` / ``ex['end']`` / ``ex['num_samples']`` are plain integers, the activity
is ``activity[session_id][speaker_id]`` without an array level
(core_chime6.py:100-128, 214-216, 401-408), and neither ``adjust_start_end`` nor an
equalised start context is needed (core_chime6.py:322-331).  The numeric pipeline
(``enhance_observation``) is inherited from ``pb_chime5_amd.core`` -- it is the same in
all three reference front doors.
"""
from dataclasses import dataclass
from pathlib import Path

import numpy as np

from pb_chime5_amd import core
from pb_chime5_amd.core import (   # noqa: F401
    WPE, GSS, Beamformer, start_end_context_frames, default_database_path)
from pb_chime5_amd.io import load_audio
from pb_chime5_amd.utils.numpy_utils import morph


def default_database_path_chime6():
    import os
    return os.environ.get('PB_CHIME6_JSON', str(Path('cache') / 'chime6.json'))


@dataclass
class Activity(core.Activity):
    """core_chime6.py:91-140."""

    @property
    def db(self):
        if self._db is None:
            from pb_chime5_amd.database.chime5.database import Chime5
            self._db = Chime5(self.database_path or default_database_path_chime6())
        return self._db

    def _annotation_activity(self, session_id):
        from pb_chime5_amd.activity import get_activity_chime6
        return get_activity_chime6(
            iterator=self.db.get_datasets(session_id),
            garbage_class=self.garbage_class, dtype=bool,
            use_ArrayIntervall=True)[session_id]


@dataclass
class Enhancer(core.Enhancer):
    """core_chime6.py:280-569."""

    def get_iterator(self, session_id):
        if self.iterator_factory is not None:
            return self.iterator_factory(session_id, self.context_samples)
        return self.db.get_iterator_for_session(
            session_id, audio_read=False, adjust_times=False,
            drop_unknown_target_speaker=True, context_samples=self.context_samples,
            equal_start_context=False)

    def _prepare_example(self, ex, dtype=np.float64):
        """Host side of core_chime6.py:396-487: one clock for activity and all arrays."""
        session_id = ex['session_id']
        speaker_id = ex['speaker_id']
        array_start, array_end = ex['start'], ex['end']
        ex_array_activity = {
            k: arr[array_start:min(array_end, len(arr))]
            for k, arr in self.activity[session_id].items()
        }

        def load_arrays(select):
            arrays = [load_audio(ex['audio_path']['observation'][array], start=array_start,
                                 stop=array_end, dtype=dtype)
                      for array in sorted(ex['audio_path']['observation'].keys())]
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
            obs = load_audio(ex['audio_path']['observation'][self._reference_array(ex)],
                             start=array_start, stop=array_end, dtype=dtype)
        else:
            raise ValueError(self.multiarray)
        return obs, ex_array_activity, speaker_id

    def _trim_context(self, x_hat, ex):
        if self.context_samples > 0:
            start_context = ex['start_orig'] - ex['start']
            x_hat = x_hat[..., start_context:start_context + ex['num_samples_orig']]
        return x_hat

    # the session driver's loader (core.Enhancer._prepare_into) on CHiME-6's single clock
    def _audio_span(self, ex, array):
        return ex['start'], ex['end']

    def _activity_span(self, ex):
        return self.activity[ex['session_id']], ex['start'], ex['end']

    def _keep_range(self, ex):
        if self.context_samples <= 0:
            return None
        keep_from = ex['start_orig'] - ex['start']
        return keep_from, keep_from + ex['num_samples_orig']


def get_enhancer(
    multiarray=False,
    context_samples=240000,
    reference_array=None,

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
    """core_chime6.py:572-635 (same keyword arguments and defaults; the last three are
    additions)."""
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
