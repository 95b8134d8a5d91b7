"""Static CHiME-5 session facts used by the session driver
(/root/reference/pb_chime5/mapping.py:1-100; consumers core.py:366,385 and
activity.py:126-150).  These are public corpus metadata: S02/S09 = dev, S01/S21 =
eval, everything else = train; four participants per session numbered in blocks of
four; six 4-channel arrays per session with U03 missing in S05/S22 and U05 in S09.

The per-recording sample counts (the reference's ``session_array_to_num_samples``
table) are NOT hard-coded here: ``num_samples_of`` reads them from the WAV headers of
the corpus at hand, so any corpus in the CHiME-5 JSON layout works, including the
synthetic one of the tests.
"""
import wave


class Dispatcher(dict):
    """dict with a more helpful KeyError."""

    def __getitem__(self, item):
        try:
            return super().__getitem__(item)
        except KeyError:
            raise KeyError(
                f'Invalid option {item!r}. Possible keys are {self.keys()!r}.') from None


# session number -> number of its first participant
_FIRST_SPEAKER = {1: 1, 2: 5, 3: 9, 4: 9, 5: 13, 6: 13, 7: 17, 8: 21, 9: 25, 12: 33, 13: 33,
                  16: 21, 17: 17, 18: 41, 19: 49, 20: 49, 21: 45, 22: 41, 23: 53, 24: 53}
_MISSING_ARRAYS = {5: (3,), 22: (3,), 9: (5,)}

session_to_dataset = Dispatcher({
    f'S{s:02d}': {2: 'dev', 9: 'dev', 1: 'eval', 21: 'eval'}.get(s, 'train')
    for s in _FIRST_SPEAKER
})

session_to_speakers = Dispatcher({
    f'S{s:02d}': [f'P{p:02d}' for p in range(first, first + 4)]
    for s, first in _FIRST_SPEAKER.items()
})

session_to_arrays = Dispatcher({
    f'S{s:02d}': [f'U{a:02d}' for a in range(1, 7) if a not in _MISSING_ARRAYS.get(s, ())]
    for s in _FIRST_SPEAKER
})


def num_samples_of(audio_path):
    """Length in samples of a recording (``audio_path``: one WAV file or the list
    of per-channel files of an array, whose first entry is read)."""
    if isinstance(audio_path, (list, tuple)):
        audio_path = audio_path[0]
    with wave.open(str(audio_path), 'rb') as w:
        return w.getnframes()
