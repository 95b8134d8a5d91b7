"""Sparse boolean vector kept as a sorted list of half-open intervals, with the
slice semantics the hot path needs (``arr[start:stop] -> dense bool`` at
/root/reference/pb_chime5/core.py:422-425).  Same public behaviour as the
reference's ``ArrayIntervall`` (pb_chime5/utils/intervall_array.py:103-455; its
Cython helpers ``cy_intersection`` / ``cy_non_intersection`` /
``cy_str_to_intervalls`` in utils/intervall_array_util.pyx are plain integer
interval arithmetic and are done inline here).  Integer-exact, including the
reference's lazy normalisation and its strict-inequality interval removal
(intervall_array_util.pyx:8-31): assigning a bool array over a range only carves
out stored intervals that start or end strictly inside the range, or strictly
contain it -- a stored interval that begins exactly at ``start`` is left as is.
"""
import numpy as np


def _normalize(intervals):
    """Sort, drop empties, merge touching / overlapping intervals."""
    out = []
    for s, e in sorted((int(s), int(e)) for s, e in intervals if s < e):
        if out and s <= out[-1][1]:
            if e > out[-1][1]:
                out[-1] = (out[-1][0], e)
        else:
            out.append((s, e))
    return tuple(out)


def ArrayIntervall_from_str(string, shape):
    """``'1:4, 5:20'`` -> intervals (utils/intervall_array.py:12-42).  A module-level name on
    purpose: the reference's ``__reduce__`` hands pickle ``staticmethod(ArrayIntervall_from_str)``
    (:104,164), so its activity pickles name ``...intervall_array.ArrayIntervall_from_str``."""
    ai = ArrayIntervall(shape)
    pairs = []
    for item in string.split(','):
        item = item.strip()
        if item:
            s, e = item.split(':')
            pairs.append((int(s), int(e)))
    ai._intervals = _normalize(pairs)
    return ai


class ArrayIntervall:
    from_str = staticmethod(ArrayIntervall_from_str)

    def __init__(self, shape):
        if isinstance(shape, (int, np.integer)):
            shape = [int(shape)]
        if shape is not None:
            assert len(shape) == 1, shape
            shape = tuple(shape)
        self.shape = shape
        self._intervals = ()

    # ---- construction --------------------------------------------------
    @staticmethod
    def from_array(array):
        array = np.asarray(array)
        assert array.ndim == 1, (array.ndim, array)
        assert array.dtype == bool, array.dtype
        edges = np.diff(np.concatenate([[0], array.astype(np.int8), [0]]))
        ai = ArrayIntervall(shape=array.shape)
        ai._intervals = tuple(zip(np.flatnonzero(edges > 0).tolist(),
                                  np.flatnonzero(edges < 0).tolist()))
        return ai

    def __reduce__(self):
        return self.from_str, (self._intervals_as_str, self.shape[-1])

    # ---- views ---------------------------------------------------------
    def __len__(self):
        return self.shape[0]

    @property
    def normalized_intervals(self):
        self._intervals = _normalize(self._intervals)
        return self._intervals

    @property
    def intervals(self):
        return self._intervals

    @property
    def _intervals_as_str(self):
        return ', '.join(f'{s}:{e}' for s, e in self.normalized_intervals)

    def __repr__(self):
        return f'{self.__class__.__name__}("{self._intervals_as_str}", shape={self.shape})'

    def _parse_item(self, item):
        assert isinstance(item, slice), (type(item), item)
        assert item.step is None, item
        start = 0 if item.start is None else item.start
        stop = self.shape[-1] if item.stop is None else item.stop
        for v in (start, stop):
            assert v >= 0, (v, item)
            if self.shape is not None:
                assert v <= self.shape[-1], (v, item)
        return int(start), int(stop)

    # ---- mutation ------------------------------------------------------
    def __setitem__(self, item, value):
        start, stop = self._parse_item(item)
        if np.isscalar(value) and value == 1:
            self._intervals = tuple(self._intervals) + ((start, stop),)
        elif isinstance(value, (tuple, list, np.ndarray)):
            assert len(value) == stop - start, (start, stop, len(value))
            inner = ArrayIntervall.from_array(np.asarray(value, dtype=bool))._intervals
            kept = []
            for s, e in self._intervals:      # strict-inequality removal, see module doc
                if start < s < stop:
                    s = stop
                elif start < e < stop:
                    e = start
                elif s < start and stop < e:
                    kept.append((s, start))
                    s = stop
                if s < e:
                    kept.append((s, e))
            self._intervals = tuple(kept) + tuple((s + start, e + start) for s, e in inner)
        else:
            raise NotImplementedError(value)

    def _bounds(self):
        """The normalised intervals as two sorted int64 arrays, kept until the next mutation
        (a session track holds thousands of intervals and is sliced once per utterance and
        speaker: the slice below touches only the intervals that reach into it)."""
        cache = self.__dict__.get('_bounds_cache')
        if cache is None or cache[0] is not self._intervals:
            iv = self.normalized_intervals
            b = np.array(iv, dtype=np.int64).reshape(-1, 2)
            cache = self._bounds_cache = (self._intervals, b[:, 0].copy(), b[:, 1].copy())
        return cache[1], cache[2]

    def slice_into(self, start, stop, out):
        """``out[:] = self[start:stop]`` for a preallocated bool / uint8 vector (0 / 1)."""
        start, stop = self._parse_item(slice(start, stop))
        assert out.shape == (stop - start,), (out.shape, start, stop)
        out[:] = 0
        starts, ends = self._bounds()
        lo = int(np.searchsorted(ends, start, side='right'))      # first interval ending after start
        hi = int(np.searchsorted(starts, stop, side='left'))      # first interval starting at / after stop
        for i in range(lo, hi):
            s, e = max(int(starts[i]), start), min(int(ends[i]), stop)
            if s < e:
                out[s - start:e - start] = 1
        return out

    def __getitem__(self, item):
        start, stop = self._parse_item(item)
        return self.slice_into(start, stop, np.empty(stop - start, dtype=bool))
