"""CHiME-5 JSON database front door
(/root/reference/pb_chime5/database/chime5/database.py:28-131, 1056-1061).

``Chime5(path).get_iterator_for_session(session, ...)`` yields the example dicts the
Enhancer consumes: redacted utterances dropped, original boundaries backed up, array
segments re-centred on the worn-microphone duration, context added with one common
start context.  All of it is integer bookkeeping on the host.
"""
from pb_chime5_amd.database import JsonDatabase
from pb_chime5_amd.database.chime5 import (   # noqa: F401  (same import path as the reference)
    activity_time_to_frequency, activity_frequency_to_time, _adjust_start_end)
from pb_chime5_amd.database.chime5.context import (
    backup_orig_start_end, adjust_start_end, AddContext)


class CHiME5_Keys:
    WORN = 'worn'
    TARGET_SPEAKER = 'target_speaker'
    NOTES = 'notes'
    SESSION_ID = 'session_id'
    LOCATION = 'location'
    REFERENCE_ARRAY = 'reference_array'


class Chime5(JsonDatabase):
    K = CHiME5_Keys

    def __init__(self, path):
        super().__init__(path)

    datasets_train = ['train']
    datasets_eval = ['dev']
    datasets_test = ['test']

    @property
    def map_dataset_to_sessions(self):
        from pb_chime5_amd.mapping import session_to_dataset
        out = {'train': [], 'dev': [], 'test': []}
        for session, dataset in session_to_dataset.items():
            out['test' if dataset == 'eval' else dataset].append(session)
        return out

    @staticmethod
    def example_id_map_fn(example):
        """'P05_S02_0004060-0004382' + location 'kitchen' ->
        'P05_S02_KITCHEN.L-0004060-0004382' (the Kaldi recipe's utterance id)."""
        speaker, session, time = example['example_id'].split('_')
        location = example[CHiME5_Keys.LOCATION]
        tag = 'NOLOCATION' if location == 'unknown' else location.upper()
        return f'{speaker}_{session}_{tag}.L-{time}'

    def get_iterator_for_session(self, session, *, audio_read=False,
                                 drop_unknown_target_speaker=False, adjust_times=False,
                                 context_samples=0, equal_start_context=False):
        if isinstance(session, str):
            session = (session,)
        it = self.get_datasets(session)

        if drop_unknown_target_speaker:
            it = it.filter(lambda ex: ex['transcription'] != '[redacted]', lazy=False)

        with_context = not (isinstance(context_samples, int) and context_samples == 0)
        if with_context or adjust_times:
            it = it.map(backup_orig_start_end)

        if adjust_times:
            if adjust_times is not True:
                raise ValueError(adjust_times)
            assert drop_unknown_target_speaker, (
                'adjust_times is undefined for ex["target_speaker"] == "unknown". '
                'Set drop_unknown_target_speaker to True.')
            it = it.map(adjust_start_end)

        if with_context:
            it = it.map(AddContext(context_samples, equal_start_context=equal_start_context))

        if audio_read is True:
            from pb_chime5_amd.io import load_audio

            def read(ex):
                ex['audio_data'] = {
                    'observation': {
                        array: load_audio(paths, start=ex['start']['observation'][array],
                                          stop=ex['end']['observation'][array])
                        for array, paths in ex['audio_path']['observation'].items()}}
                return ex
            it = it.map(read)
        elif audio_read is not False:
            raise TypeError(audio_read)
        return it


class SessionFilter:
    def __init__(self, session_id):
        self.session_id = session_id

    def __call__(self, example):
        return example['session_id'] == self.session_id
