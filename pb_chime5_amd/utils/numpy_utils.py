"""Array helpers the hot path relies on: ``segment_axis_v2``, ``pad_axis`` and
``morph`` with the semantics of /root/reference/pb_chime5/utils/numpy_utils.py
(lines 10-222, 496-556, 662-707).  Written from the behaviour pinned by the
reference's doctests (captured in tests/golden/host_helpers.npz); only what
``core.py`` / ``beamforming_wrapper.py`` / ``activity_time_to_frequency`` use is
implemented.
"""
import numpy as np


def segment_axis_v2(x, length, shift, axis=-1, end='pad', pad_mode='constant',
                    pad_value=0):
    """Chop ``x`` along ``axis`` into frames of ``length`` every ``shift`` samples.

    end='pad': zero-extend so that the last partial frame is kept; 'cut': drop
    it; None: the length must fit exactly.  Returns a view where possible; the
    new frame axis is inserted at ``axis`` and the in-frame axis at ``axis + 1``.
    """
    x = np.asarray(x)
    axis = axis % x.ndim
    if shift == 0:
        raise ValueError(shift)
    do_flip = shift < 0
    shift = abs(shift)
    n = x.shape[axis]

    def _pad(amount, both=False):
        width = [(0, 0)] * x.ndim
        width[axis] = (amount, amount) if both else (0, amount)
        kwargs = {'constant_values': pad_value} if pad_mode == 'constant' else {}
        return np.pad(x, width, mode=pad_mode, **kwargs)

    if end == 'pad':
        if n < length:
            x = _pad(length - n)
        elif shift != 1 and (n + shift - length) % shift != 0:
            x = _pad(shift - ((n + shift - length) % shift))
    elif end == 'conv_pad':
        assert shift == 1, shift
        x = _pad(length - shift, both=True)
    elif end is None:
        assert (n + shift - length) % shift == 0, (n, shift, length)
    elif end == 'cut':
        pass
    else:
        raise ValueError(end)

    n = x.shape[axis]
    num = max((n + shift - length) // shift, 0)
    shape = x.shape[:axis] + (num, length) + x.shape[axis + 1:]
    strides = x.strides[:axis] + (shift * x.strides[axis], x.strides[axis]) \
        + x.strides[axis + 1:]
    out = np.lib.stride_tricks.as_strided(x, shape=shape, strides=strides)
    if do_flip:
        return np.flip(out, axis=axis)
    return out


def pad_axis(array, pad_width, *, axis, mode='constant', **pad_kwargs):
    """np.pad restricted to one axis."""
    array = np.asarray(array)
    npad = np.zeros([array.ndim, 2], dtype=int)
    npad[axis, :] = pad_width
    return np.pad(array, pad_width=npad, mode=mode, **pad_kwargs)


def _tokens(side):
    """'A*CN' -> [['A','C'],['N']]; '1DTF' -> [['1'],['D'],['T'],['F']]."""
    side = side.replace(' ', '').replace(',', '')
    groups = []
    i = 0
    while i < len(side):
        ch = side[i]
        if ch == '*':
            raise ValueError(side)
        group = [ch]
        i += 1
        while i < len(side) and side[i] == '*':
            group.append(side[i + 1])
            i += 2
        groups.append(group)
    return groups


def morph(operation, array, reduce=None, **shape_hints):
    """Generalised reshape / transpose / reduce, e.g. ``morph('DTF->FDT', x)``,
    ``morph('ACN->A*CN', x)``, ``morph('A*CTF->ACTF', x, A=2)``,
    ``morph('1DTF->FT', x, reduce=np.median)``."""
    array = np.asarray(array)
    source, target = operation.split('->')
    src, tgt = _tokens(source), _tokens(target)
    assert len(src) == array.ndim, (operation, array.shape)

    # expand grouped source axes
    shape = []
    letters = []
    for group, size in zip(src, array.shape):
        if len(group) == 1:
            shape.append(size)
            letters.append(group[0])
        else:
            known = [shape_hints.get(g) for g in group]
            missing = [i for i, k in enumerate(known) if k is None]
            if len(missing) > 1:
                raise ValueError('Not enough shape hints provided.')
            if missing:
                prod = int(np.prod([k for k in known if k is not None])) or 1
                known[missing[0]] = size // prod
            shape.extend(known)
            letters.extend(group)
    array = array.reshape(shape)

    # squeeze '1' axes
    keep = [i for i, l in enumerate(letters) if l != '1']
    for i, l in enumerate(letters):
        if l == '1':
            assert array.shape[i] == 1, (operation, array.shape)
    array = array.reshape([array.shape[i] for i in keep])
    letters = [letters[i] for i in keep]

    out_letters = [l for group in tgt for l in group if l != '1']
    drop = [i for i, l in enumerate(letters) if l not in out_letters]
    if drop:
        assert reduce is not None, ('Missing reduce function', reduce, operation)
        array = reduce(array, axis=tuple(drop))
        letters = [l for l in letters if l in out_letters]
    array = array.transpose([letters.index(l) for l in out_letters])

    size_of = dict(zip(out_letters, array.shape))
    final = []
    for group in tgt:
        if group == ['1']:
            final.append(1)
        else:
            final.append(int(np.prod([size_of[l] for l in group])))
    return array.reshape(final)
