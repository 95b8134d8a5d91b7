"""Time bookkeeping of CHiME-5 examples: keep the original boundaries, re-centre the
array segments on the worn-microphone duration, add context
(/root/reference/pb_chime5/database/chime5/database.py:540-570, 706-710, 713-1053).

CHiME-5 examples carry nested integer structures
``ex['start'] = {'original': s, 'observation': {'U01': s1, ...}, 'worn_microphone': {...}}``
(CHiME-6 examples plain integers).  Everything here is integer-exact host logic.
"""
import copy

from pb_chime5_amd.database.chime5 import _adjust_start_end


def _map(func, *trees):
    first = trees[0]
    if isinstance(first, dict):
        assert all(isinstance(t, dict) and t.keys() == first.keys() for t in trees), trees
        return {k: _map(func, *[t[k] for t in trees]) for k in first}
    if isinstance(first, (list, tuple)):
        assert all(type(t) is type(first) and len(t) == len(first) for t in trees), trees
        return type(first)(_map(func, *args) for args in zip(*trees))
    return func(*trees)


def _leaves(tree):
    if isinstance(tree, dict):
        for v in tree.values():
            yield from _leaves(v)
    elif isinstance(tree, (list, tuple)):
        for v in tree:
            yield from _leaves(v)
    else:
        yield tree


def backup_orig_start_end(ex):
    ex['start_orig'] = ex['start']
    ex['end_orig'] = ex['end']
    ex['num_samples_orig'] = ex['num_samples']
    return ex


def adjust_start_end(ex):
    """Give every array / worn-microphone segment the duration of the 'original'
    (worn reference) one, splitting the difference around its centre."""
    worn_start, worn_end = ex['start']['original'], ex['end']['original']
    for group in ('observation', 'worn_microphone'):
        if group == 'observation':
            keys = ex['audio_path']['observation'].keys()
        else:
            keys = ex['audio_path'].get('worn_microphone', {}).keys()
        for key in keys:
            start, end = _adjust_start_end(worn_start, worn_end, ex['start'][group][key],
                                           ex['end'][group][key])
            ex['start'][group][key] = start
            ex['end'][group][key] = end
            ex['num_samples'][group][key] = end - start
    return ex


def _split(samples):
    if isinstance(samples, (tuple, list)):
        if len(samples) == 1:
            samples = (samples[0], samples[0])
        assert len(samples) == 2, samples
        start, end = samples
    else:
        start = end = samples
    assert isinstance(start, int) and isinstance(end, int), samples
    assert start >= 0 and end >= 0, f'Negative context value ({samples}) is not supported'
    return start, end


def AddContext(samples, equal_start_context=False):
    """Returns ``add_context(ex)``: start <- max(start - context, 0), end <- end + context
    for every leaf; with ``equal_start_context`` all leaves get the SMALLEST start
    context that any of them could afford, so that the arrays stay aligned."""
    start_context, end_context = _split(samples)

    def add_context(ex):
        for key in ('start_orig', 'end_orig', 'num_samples_orig'):
            assert key in ex, ex
        if ex['start_orig'] is ex['start']:          # backup aliases the same objects
            ex['start_orig'] = copy.deepcopy(ex['start'])
            ex['end_orig'] = copy.deepcopy(ex['end'])
            ex['num_samples_orig'] = copy.deepcopy(ex['num_samples'])
        ex['start'] = _map(lambda t: max(t - start_context, 0), ex['start'])
        if equal_start_context:
            delta = _map(lambda s, s_orig: s_orig - s, ex['start'], ex['start_orig'])
            smallest = min(_leaves(delta))
            ex['start'] = _map(lambda t: max(t - smallest, 0), ex['start_orig'])
        ex['end'] = _map(lambda t: t + end_context, ex['end'])
        ex['num_samples'] = _map(lambda s, e: e - s, ex['start'], ex['end'])
        return ex

    return add_context
