"""RTTM-driven enhancement of CHiME-6 style recordings (track 2): the surface of
/root/reference/pb_chime5/core_chime6_rttm.py on top of the same GPU hot path.

    enhancer = get_enhancer(database_rttm=..., activity_rttm=..., chime6_dir=...,
                            multiarray='outer_array_mics')
    enhancer.enhance_session('S02', 'out/audio')

One example = one RTTM segment of one speaker plus ``context_samples`` on both
sides; all channels of the session are loaded for that window and every speaker's
RTTM activity in the window guides the CACGMM.  Arrays are already synchronised in
CHiME-6, so start / end are plain integers (``core_chime6.py:214-216``).
"""
from dataclasses import dataclass, field

import numpy as np
from pathlib import Path

from pb_chime5_amd import core, mapping
from pb_chime5_amd.core import WPE, GSS, Beamformer, start_end_context_frames  # noqa: F401
from pb_chime5_amd.database.chime5 import rttm as rttm_module
from pb_chime5_amd.io import dump_audio


@dataclass
class Activity:
    """core_chime6_rttm.py:31-69."""
    garbage_class: bool = False
    rttm: object = None
    _data: dict = field(default=None, repr=False)

    def __getitem__(self, session_id):
        if self._data is None:
            self._data = rttm_module.strip_file_id(rttm_module.from_rttm(self.rttm))
        data = dict(self._data[session_id])
        if self.garbage_class is False:
            data['Noise'] = rttm_module.zeros()
        elif self.garbage_class is True:
            data['Noise'] = rttm_module.ones()
        elif self.garbage_class is None:
            pass
        else:
            raise ValueError(self.garbage_class)
        return data


@dataclass
class Enhancer(core.Enhancer):
    """core_chime6_rttm.py:72-282; the numeric pipeline (enhance_observation) is
    inherited -- it is the same in all three reference front doors."""
    db: object = None

    def get_dataset(self, session_id):
        return self.db.get_dataset_for_session(
            session_id, audio_read=True, adjust_times=False,
            context_samples=self.context_samples, equal_start_context=False)

    def enhance_session(self, session_ids, audio_dir, dataset_slice=False,
                        audio_dir_exist_ok=False):
        from pb_chime5_amd import parallel
        audio_dir = Path(audio_dir)
        it = self.get_dataset(session_ids)
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
        # hand out indices, not loaded examples: audio is read by the rank that works
        costs = [ex['num_samples'] for ex in it.examples]
        indices = parallel.split_managed(range(len(it)), costs=costs)
        # the examples travel without audio; _prepare_example reads it (on the loader thread)
        self._enhance_and_write((dict(it.examples[index]) for index in indices), audio_dir)
        parallel.barrier()      # (split_managed ends without one; the pipeline has drained)

    def _prepare_example(self, ex, dtype=np.float64):
        """core_chime6_rttm.py:228-258.  Examples of ``get_dataset`` carry their audio;
        inside ``enhance_session`` it is read here (``dtype=np.int16``: PCM as stored)."""
        array_start, array_end = ex['start'], ex['end']
        ex_array_activity = {
            k: arr[array_start:array_end] for k, arr in self.activity[ex['session_id']].items()
        }
        obs = ex.get('audio_data')
        if obs is None:
            obs = rttm_module.recursive_load_audio(
                ex['audio_path'], start=array_start, stop=array_end,
                min_num_samples=ex.get('end_orig', array_end) - array_start, dtype=dtype)
        return obs, ex_array_activity, ex['speaker_id']

    def _trim_context(self, x_hat, ex):
        if self.context_samples > 0:
            start_context = ex['start_orig'] - ex['start']
            x_hat = x_hat[..., start_context:start_context + ex['num_samples_orig']]
        return x_hat

    # RTTM segments may reach past the end of a recording (zero padded by
    # recursive_load_audio) and the garbage class has no length: the session driver's loader
    # threads go through _prepare_example above and copy once
    fast_loader = False

    def _keep_range(self, ex):
        if self.context_samples <= 0:
            return None
        keep_from = ex['start_orig'] - ex['start']
        return keep_from, keep_from + ex['num_samples_orig']


def get_database(chime6_dir, rttm, multiarray):
    """core_chime6_rttm.py:288-357."""
    chime6_dir = Path(chime6_dir)
    audio_paths = rttm_module.select_channels(chime6_dir, multiarray)
    alias = {}
    for p in sorted(chime6_dir.glob('transcriptions/*/*.json')):
        alias.setdefault(p.parts[-2], []).append(p.with_suffix('').name)
    return rttm_module.RTTMDatabase(rttm, audio_paths, alias=alias)


def get_enhancer(
    database_rttm,
    activity_rttm,
    chime6_dir='/net/fastdb/chime6/CHiME6',
    multiarray='outer_array_mics',
    context_samples=240000,

    wpe=True,
    wpe_tabs=10,
    wpe_delay=2,
    wpe_iterations=3,
    wpe_psd_context=0,

    activity_garbage_class=True,

    stft_size=1024,
    stft_shift=256,
    stft_fading=True,

    bss_iterations=20,
    bss_iterations_post=1,

    bf_drop_context=True,

    bf='mvdrSouden_ban',
    postfilter=None,

    device_id=None,
):
    """core_chime6_rttm.py:360-422 (same keyword arguments and defaults)."""
    assert wpe is True or wpe is False, wpe
    db = get_database(chime6_dir, database_rttm, multiarray)
    return Enhancer(
        db=db,
        context_samples=context_samples,
        multiarray=multiarray,
        reference_array=None,
        wpe_block=WPE(taps=wpe_tabs, delay=wpe_delay, iterations=wpe_iterations,
                      psd_context=wpe_psd_context) if wpe else None,
        activity=Activity(garbage_class=activity_garbage_class, rttm=activity_rttm),
        gss_block=GSS(iterations=bss_iterations, iterations_post=bss_iterations_post,
                      verbose=False),
        bf_drop_context=bf_drop_context,
        bf_block=Beamformer(type=bf, postfilter=postfilter),
        stft_size=stft_size,
        stft_shift=stft_shift,
        stft_fading=stft_fading,
        device_id=device_id,
    )
