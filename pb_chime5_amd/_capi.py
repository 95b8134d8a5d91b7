"""ctypes binding of include/gss_hip.h (libgss_hip.so).

This is the only place the Python host touches native code.  There is no CPU
fallback: if the library is missing or no GPU is visible the product path raises.
"""
import ctypes
import json
import os
from pathlib import Path

import numpy as np

_LIB = None
LIB_PATH = Path(__file__).resolve().parent / 'lib' / 'libgss_hip.so'

GSS_OK = 0
GSS_ERR_INVALID = -1
GSS_ERR_HIP = -2
GSS_ERR_NOMEM = -3
GSS_ERR_UNSUPPORTED = -4
GSS_ABI_VERSION = 6       # include/gss_hip.h revision these prototypes are written against

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_int64 = ctypes.c_int64
c_size_t = ctypes.c_size_t


class GssParams(ctypes.Structure):
    _fields_ = [(n, c_int) for n in (
        'stft_size', 'stft_shift', 'stft_fading', 'wpe', 'wpe_taps', 'wpe_delay',
        'wpe_iterations', 'bss_iterations', 'bss_iterations_post',
        'bf_drop_context', 'bf', 'postfilter', 'wpe_psd_context')]


class GssDebugTaps(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in (
        'Obs_ftd', 'act_frames', 'gamma', 'target_mask', 'distortion_mask',
        'Xhat', 'ref_channel')]


# name -> (restype, argtypes); every symbol include/gss_hip.h declares
SIGNATURES = {
    'gss_device_count': (c_int, []),
    'gss_device_pci_bus_id': (c_int, [c_int, ctypes.c_char_p, c_int]),
    'gss_create': (c_int, [c_int, ctypes.POINTER(c_void_p)]),
    'gss_destroy': (c_int, [c_void_p]),
    'gss_last_error': (ctypes.c_char_p, [c_void_p]),
    'gss_version': (ctypes.c_char_p, []),
    'gss_abi_version': (c_int, []),
    'gss_set_stream': (c_int, [c_void_p, c_void_p]),
    'gss_set_utterances_in_flight': (c_int, [c_void_p, c_int]),
    'gss_synchronize': (c_int, [c_void_p]),
    'gss_dev_malloc': (c_int, [c_void_p, c_size_t, ctypes.POINTER(c_void_p)]),
    'gss_dev_free': (c_int, [c_void_p, c_void_p]),
    'gss_memcpy_h2d': (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    'gss_memcpy_d2h': (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    'gss_memset': (c_int, [c_void_p, c_void_p, c_int, c_size_t]),
    'gss_host_malloc': (c_int, [c_void_p, c_size_t, ctypes.POINTER(c_void_p)]),
    'gss_host_free': (c_int, [c_void_p, c_void_p]),
    'gss_memcpy_h2d_async': (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    'gss_memcpy_d2h_async': (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    'gss_profile_enable': (c_int, [c_void_p, c_int]),
    'gss_profile_filter': (c_int, [c_void_p, ctypes.c_char_p]),
    'gss_profile_reset': (c_int, [c_void_p]),
    'gss_profile_report': (c_int, [c_void_p, ctypes.c_char_p, c_size_t]),
    'gss_stft_num_frames': (c_int64, [c_int64, c_int, c_int, c_int]),
    'gss_istft_num_samples': (c_int64, [c_int64, c_int, c_int, c_int]),
    'gss_samples_to_stft_frames': (c_int64, [c_int64, c_int, c_int, c_int]),
    'gss_set_windows': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'gss_stft': (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p]),
    'gss_istft': (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    'gss_activity_time_to_frequency': (
        c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p]),
    'gss_wpe': (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int,
                        c_int, c_int, c_void_p]),
    'gss_wpe_inverse_power': (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_int,
                                      c_void_p]),
    'gss_cacgmm': (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p,
                           c_int, c_int, c_int, c_void_p]),
    'gss_masks_from_posteriors': (
        c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_int, c_int64,
                c_int64, c_void_p, c_void_p]),
    'gss_mvdr_souden': (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int,
                                c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    'gss_mvdr_souden_ref': (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int,
                                    c_void_p, c_void_p, c_int, c_int, c_void_p]),
    'gss_last_ref_channel': (c_int, [c_void_p, ctypes.POINTER(ctypes.c_int32)]),
    'gss_last_wpe_zero_pivots': (c_int, [c_void_p, ctypes.POINTER(ctypes.c_int64)]),
    'gss_gev': (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_int,
                        c_void_p]),
    'gss_layout_dtf_to_ftd': (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int,
                                      c_void_p]),
    'gss_layout_ftd_to_dtf': (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int,
                                      c_void_p]),
    'gss_layout_permute_f64': (c_int, [c_void_p, c_void_p, c_int64, c_int64,
                                       c_int64, c_int, c_void_p]),
    'gss_enhance_observation': (
        c_int, [c_void_p, ctypes.POINTER(GssParams), c_void_p, c_int, c_int64,
                c_void_p, c_int, c_int64, c_int, c_int64, c_int64, c_void_p,
                ctypes.POINTER(GssDebugTaps)]),
    'gss_enhance_observation_pcm16': (
        c_int, [c_void_p, ctypes.POINTER(GssParams), c_void_p, c_int, c_int64,
                c_void_p, c_int, c_int64, c_int, c_int64, c_int64, c_void_p,
                ctypes.POINTER(GssDebugTaps)]),
    'gss_enhance_observation_host': (
        c_int, [c_void_p, ctypes.POINTER(GssParams), c_void_p, c_int, c_int64,
                c_void_p, c_int, c_int64, c_int, c_int64, c_int64, c_void_p]),
    'gss_workspace_bytes': (c_size_t, [c_void_p]),
    'gss_selftest_mfma': (c_int, [c_void_p]),
}


class GssError(RuntimeError):
    pass


def load_library(path=None):
    """Load libgss_hip.so (no GPU needed for loading) and declare prototypes."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    p = Path(path or os.environ.get('GSS_HIP_LIBRARY', LIB_PATH))
    if not p.exists():
        raise GssError(
            f'{p} is missing: build it with `python -m pb_chime5_amd.build` '
            '(there is no CPU fallback)')
    lib = ctypes.CDLL(str(p))
    # the prototypes below are written against one revision of include/gss_hip.h: a library
    # of another revision would read shifted arguments or a shorter gss_params
    abi = getattr(lib, 'gss_abi_version', None)
    got = None
    if abi is not None:
        abi.restype, abi.argtypes = c_int, []
        got = int(abi())
    if got != GSS_ABI_VERSION:
        raise GssError(f'{p}: ABI revision {got}, this binding needs {GSS_ABI_VERSION} '
                       '(rebuild with `python -m pb_chime5_amd.build --force`)')
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    if path is None:
        _LIB = lib
    return lib


def _raise(lib, ctx, status, what):
    msg = lib.gss_last_error(ctx)
    msg = msg.decode() if msg else ''
    text = f'{what}: {msg}' if msg else what
    # mirror the exception types of the reference
    if status == GSS_ERR_INVALID:
        if 'assert' in msg:
            raise AssertionError(text)
        raise ValueError(text)
    if status == GSS_ERR_UNSUPPORTED:
        raise NotImplementedError(text)
    if status == GSS_ERR_NOMEM:
        raise MemoryError(text)
    raise GssError(text)


class DeviceBuffer:
    """A device allocation owned through the C ABI (gss_dev_malloc/free)."""

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        ptr = c_void_p()
        ctx._check(ctx.lib.gss_dev_malloc(ctx.handle, self.nbytes, ctypes.byref(ptr)),
                   'gss_dev_malloc')
        self.ptr = ptr.value

    def free(self):
        if self.ptr is not None and self.ctx.handle is not None:
            self.ctx.lib.gss_dev_free(self.ctx.handle, c_void_p(self.ptr))
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PinnedBuffer:
    """Page-locked host memory (gss_host_malloc/free), handed out as NumPy views."""

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        self.nbytes = max(int(nbytes), 16)
        ptr = c_void_p()
        ctx._check(ctx.lib.gss_host_malloc(ctx.handle, self.nbytes, ctypes.byref(ptr)),
                   'gss_host_malloc')
        self.ptr = ptr.value
        self._raw = (ctypes.c_char * self.nbytes).from_address(self.ptr)

    def view(self, shape, dtype, offset=0):
        """An array of `shape` / `dtype` over the block (no copy), starting `offset` bytes in."""
        count = int(np.prod(shape, dtype=np.int64))
        assert offset + count * np.dtype(dtype).itemsize <= self.nbytes
        return np.frombuffer(self._raw, dtype=dtype, count=count, offset=offset).reshape(shape)

    def free(self):
        if self.ptr is not None and self.ctx.handle is not None:
            self._raw = None
            self.ctx.lib.gss_host_free(self.ctx.handle, c_void_p(self.ptr))
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """One per GPU (and per host thread)."""

    def __init__(self, device_id=0):
        self.lib = load_library()
        self.handle = None
        h = c_void_p()
        st = self.lib.gss_create(int(device_id), ctypes.byref(h))
        if st != GSS_OK:
            msg = self.lib.gss_last_error(None)
            raise GssError(
                f'gss_create(device {device_id}) failed: '
                f'{msg.decode() if msg else st} -- the HIP path needs an AMD GPU '
                '(there is no CPU fallback)')
        self.handle = h
        self.device_id = device_id
        self._windows = None

    # -- plumbing ----------------------------------------------------------
    def _check(self, status, what):
        if status != GSS_OK:
            _raise(self.lib, self.handle, status, what)

    def close(self):
        if self.handle is not None:
            self.lib.gss_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        self._check(self.lib.gss_synchronize(self.handle), 'gss_synchronize')

    def set_stream(self, stream_handle):
        self._check(self.lib.gss_set_stream(self.handle, c_void_p(stream_handle)),
                    'gss_set_stream')

    def set_utterances_in_flight(self, n):
        """How many utterances the caller keeps in flight on this GPU (0 = not said, the default).
        Exactly 1 lets a fused call use the context's internal second stream for half of the WPE
        stage's frequencies (same bits, ~1.5 % sooner); anything else keeps it on one stream."""
        self._check(self.lib.gss_set_utterances_in_flight(self.handle, int(n)),
                    'gss_set_utterances_in_flight')

    def empty(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def to_device(self, array):
        a = np.ascontiguousarray(array)
        buf = DeviceBuffer(self, max(a.nbytes, 16))
        if a.nbytes:
            self._check(self.lib.gss_memcpy_h2d(
                self.handle, c_void_p(buf.ptr), a.ctypes.data_as(c_void_p), a.nbytes),
                'gss_memcpy_h2d')
        return buf

    def upload(self, buf, array):
        a = np.ascontiguousarray(array)
        assert a.nbytes <= buf.nbytes
        self._check(self.lib.gss_memcpy_h2d(
            self.handle, c_void_p(buf.ptr), a.ctypes.data_as(c_void_p), a.nbytes),
            'gss_memcpy_h2d')

    def pinned(self, nbytes):
        return PinnedBuffer(self, nbytes)

    def upload_async(self, buf, array, offset=0):
        """H2D without waiting; `array` (C-contiguous, ideally a PinnedBuffer view) must stay
        untouched until the context was synchronised."""
        assert array.flags.c_contiguous and offset + array.nbytes <= buf.nbytes
        self._check(self.lib.gss_memcpy_h2d_async(
            self.handle, c_void_p(buf.ptr + offset), array.ctypes.data_as(c_void_p),
            array.nbytes), 'gss_memcpy_h2d_async')

    def download_async(self, array, buf, offset=0):
        """D2H into `array` without waiting (valid after synchronize())."""
        assert array.flags.c_contiguous and offset + array.nbytes <= buf.nbytes
        self._check(self.lib.gss_memcpy_d2h_async(
            self.handle, array.ctypes.data_as(c_void_p), c_void_p(buf.ptr + offset),
            array.nbytes), 'gss_memcpy_d2h_async')

    def to_host(self, buf, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= buf.nbytes, (out.nbytes, buf.nbytes)
        if out.nbytes:
            self._check(self.lib.gss_memcpy_d2h(
                self.handle, out.ctypes.data_as(c_void_p), c_void_p(buf.ptr), out.nbytes),
                'gss_memcpy_d2h')
        return out

    # -- profiling ---------------------------------------------------------
    def profile_enable(self, on=True):
        self._check(self.lib.gss_profile_enable(self.handle, int(on)), 'profile_enable')

    def profile_filter(self, kernel=None):
        self._check(self.lib.gss_profile_filter(
            self.handle, kernel.encode() if kernel else None), 'profile_filter')

    def profile_reset(self):
        self._check(self.lib.gss_profile_reset(self.handle), 'profile_reset')

    def profile_report(self):
        buf = ctypes.create_string_buffer(1 << 16)
        self._check(self.lib.gss_profile_report(self.handle, buf, len(buf)),
                    'profile_report')
        return json.loads(buf.value.decode())

    def last_ref_channel(self):
        """Reference channel of the last MVDR run (synchronises); -1 = non-finite SNR."""
        out = ctypes.c_int32()
        self._check(self.lib.gss_last_ref_channel(self.handle, ctypes.byref(out)),
                    'gss_last_ref_channel')
        return int(out.value)

    def last_wpe_zero_pivots(self):
        """Pivots the WPE solve of the last call zeroed (synchronises); > 0 on live channels
        means rank-deficient normal equations (T <= taps * D), see include/gss_hip.h."""
        out = ctypes.c_int64()
        self._check(self.lib.gss_last_wpe_zero_pivots(self.handle, ctypes.byref(out)),
                    'gss_last_wpe_zero_pivots')
        return int(out.value)

    def workspace_bytes(self):
        return int(self.lib.gss_workspace_bytes(self.handle))

    # -- STFT tables -------------------------------------------------------
    def set_windows(self, size, shift, analysis, synthesis):
        key = (size, shift, analysis.tobytes(), synthesis.tobytes())
        if self._windows == key:
            return
        a = np.ascontiguousarray(analysis, dtype=np.float64)
        s = np.ascontiguousarray(synthesis, dtype=np.float64)
        assert a.shape == (size,) and s.shape == (size,)
        self._check(self.lib.gss_set_windows(
            self.handle, size, shift, a.ctypes.data_as(c_void_p),
            s.ctypes.data_as(c_void_p)), 'gss_set_windows')
        self._windows = key


_DEFAULT_CTX = {}


def device_count():
    return int(load_library().gss_device_count())


def device_pci_bus_id(device_id):
    """'0000:c1:00.0' -- the key of the device's sysfs entry (NUMA node, local CPUs)."""
    buf = ctypes.create_string_buffer(32)
    rc = load_library().gss_device_pci_bus_id(int(device_id), buf, len(buf))
    if rc != 0:
        raise GssError(f'gss_device_pci_bus_id({device_id}) failed with status {rc}')
    return buf.value.decode()


def pci_package(bus_id):
    """The part of a PCI address that names the physical package: 'domain:bus:device'.  The
    logical devices of a partitioned GPU (CPX / NPS modes: up to 8 per package) differ in the
    function digit at most."""
    return bus_id.lower().rsplit('.', 1)[0]


def pick_device(local_rank, bus_ids):
    """The logical device of a node-local rank: ranks go to DISTINCT physical packages first
    (round robin over the packages in device order), and only then to further logical devices
    of a package.  One logical device per package (the usual 8-GPU node): rank r -> device
    r % count, as before; a partitioned node with 64 logical devices on 8 packages: ranks 0-7
    land on 8 different GPUs instead of on the 8 partitions of the first."""
    packages = {}
    for index, bus_id in enumerate(bus_ids):
        packages.setdefault(pci_package(bus_id), []).append(index)
    groups = list(packages.values())
    if not groups:
        return 0
    mine = groups[local_rank % len(groups)]
    return mine[(local_rank // len(groups)) % len(mine)]


def default_device():
    """$GSS_DEVICE, else by LOCAL_RANK over the visible GPUs, distinct physical packages first
    (`pick_device`; ranks share devices when a node has fewer GPUs than ranks), else 0."""
    if 'GSS_DEVICE' in os.environ:
        return int(os.environ['GSS_DEVICE'])
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if not local_rank:
        return 0
    count = max(device_count(), 1)
    try:
        return pick_device(local_rank, [device_pci_bus_id(i) for i in range(count)])
    except GssError:
        return local_rank % count


def default_context(device_id=None):
    """Process-wide context for ``device_id`` (default: ``default_device()``)."""
    if device_id is None:
        device_id = default_device()
    ctx = _DEFAULT_CTX.get(device_id)
    if ctx is None or ctx.handle is None:
        ctx = _DEFAULT_CTX[device_id] = Context(device_id)
    return ctx
