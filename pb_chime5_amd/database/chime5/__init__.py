"""Host-side time <-> frame activity conversion
(/root/reference/pb_chime5/database/chime5/database.py:328-472).  Integer / bool
logic, bit-exact with the reference (tests/test_host_golden.py).  The fused GPU
pipeline runs the same reduction on the device
(``gss_activity_time_to_frequency``)."""
import numpy as np

from pb_chime5_amd.utils.numpy_utils import segment_axis_v2, pad_axis


def activity_time_to_frequency(time_activity, stft_window_length, stft_shift,
                               stft_fading, stft_pad=True):
    """bool time activity (..., N) -> bool frame activity (..., T): a frame is
    active if any sample under its window (including the fading pad) is."""
    time_activity = np.asarray(time_activity)
    assert time_activity.dtype != object, (type(time_activity), time_activity.dtype)
    if stft_fading:
        pad_width = np.array([(0, 0)] * time_activity.ndim)
        pad_width[-1, :] = stft_window_length - stft_shift
        time_activity = np.pad(time_activity, pad_width, mode='constant')
    return segment_axis_v2(
        time_activity, length=stft_window_length, shift=stft_shift,
        end='pad' if stft_pad else 'cut').any(axis=-1)


def activity_frequency_to_time(frequency_activity, stft_window_length, stft_shift,
                               stft_fading, time_length=None):
    if stft_fading:
        raise NotImplementedError(stft_fading)
    frequency_activity = np.asarray(frequency_activity)
    frequency_activity = np.broadcast_to(
        frequency_activity[..., None],
        (*frequency_activity.shape, stft_window_length))
    time_activity = np.zeros(
        (*frequency_activity.shape[:-2],
         frequency_activity.shape[-2] * stft_shift + stft_window_length - stft_shift))
    seg = segment_axis_v2(time_activity, stft_window_length, stft_shift, end=None)
    seg[frequency_activity > 0] = 1
    time_activity = time_activity != 0
    if time_length is not None:
        if time_length < time_activity.shape[-1]:
            delta = time_activity.shape[-1] - time_length
            assert delta < stft_window_length - stft_shift, (delta,)
            time_activity = time_activity[..., :time_length]
        elif time_length > time_activity.shape[-1]:
            delta = time_length - time_activity.shape[-1]
            assert delta < stft_window_length - stft_shift, (delta,)
            time_activity = pad_axis(time_activity, pad_width=(0, delta), axis=-1)
        assert time_length == time_activity.shape[-1]
    return time_activity != 0


def _adjust_start_end(worn_start, worn_end, array_start, array_end):
    """database.py:475-537: make the array segment as long as the worn one,
    splitting the difference (larger half at the end)."""
    worn_duration = worn_end - worn_start
    array_duration = array_end - array_start
    delta = abs(worn_duration - array_duration)
    delta_start, delta_end = delta // 2, (delta + 1) // 2
    if worn_duration >= array_duration:
        new_start, new_end = array_start - delta_start, array_end + delta_end
    else:
        new_start, new_end = array_start + delta_start, array_end - delta_end
    assert new_end - new_start == worn_duration
    return new_start, new_end
