"""Mask-based beamforming front door with the reference's call surface
(/root/reference/pb_chime5/speech_enhancement/beamforming_wrapper.py:11-124).

``beamform_mvdr_souden_from_masks(Y, X_mask, N_mask, ban=False)`` accepts the same
layouts as the reference -- Y (D,T,F) or (1,D,T,F); masks (T,F), (D,T,F) or
(1,D,T,F), median-reduced over the channel axis -- and returns X_hat (T,F)
complex128.  PSD accumulation, the Souden MVDR solve, the cross-frequency
reference-channel choice, BAN and the filter application run in the HIP library.
"""
import numpy as np

from pb_chime5_amd import ops
from pb_chime5_amd.utils.numpy_utils import morph


class _Beamformer:
    """Layout normalisation + lazily evaluated results, like the reference's
    helper class; the arithmetic is one ``gss_mvdr_souden`` call per variant."""

    def __init__(self, Y, X_mask, N_mask, debug=False, ctx=None):
        self.debug = debug
        self._ctx = ctx
        if np.ndim(Y) == 4:
            self.Y = morph('1DTF->FDT', Y)
        else:
            self.Y = morph('DTF->FDT', Y)

        if np.ndim(X_mask) == 4:
            self.X_mask = morph('1DTF->FT', X_mask, reduce=np.median)
            self.N_mask = morph('1DTF->FT', N_mask, reduce=np.median)
        elif np.ndim(X_mask) == 3:
            self.X_mask = morph('DTF->FT', X_mask, reduce=np.median)
            self.N_mask = morph('DTF->FT', N_mask, reduce=np.median)
        elif np.ndim(X_mask) == 2:
            self.X_mask = morph('TF->FT', X_mask)
            self.N_mask = morph('TF->FT', N_mask)
        else:
            raise NotImplementedError(np.shape(X_mask))

        assert self.Y.ndim == 3, self.Y.shape
        F, D, T = self.Y.shape
        assert D < 30, (D, self.Y.shape)
        assert self.X_mask.shape == (F, T), (self.X_mask.shape, F, T)
        assert self.N_mask.shape == (F, T), (self.N_mask.shape, F, T)
        self._cache = {}

    def _run(self, ban):
        if ban not in self._cache:
            self._cache[ban] = ops.mvdr_souden_from_masks(
                self.Y.transpose(1, 2, 0), self.X_mask.T, self.N_mask.T, ban=ban,
                return_ref_channel=True, ctx=self._ctx)
        return self._cache[ban]

    @property
    def X_hat_mvdr_souden(self):
        return self._run(False)[0]

    @property
    def X_hat_mvdr_souden_ban(self):
        return self._run(True)[0]

    @property
    def ref_channel(self):
        return self._run(True)[1]

    def _gev(self, ban):
        key = ('gev', ban)
        if key not in self._cache:
            self._cache[key] = ops.gev_from_masks(
                self.Y.transpose(1, 2, 0), self.X_mask.T, self.N_mask.T, ban=ban, ctx=self._ctx)
        return self._cache[key]

    @property
    def X_hat_gev(self):
        return self._gev(False)

    @property
    def X_hat_gev_ban(self):
        return self._gev(True)


def beamform_gev_from_masks(Y, X_mask, N_mask, ban=True, debug=False, ctx=None):
    """beamforming_wrapper.py:192-208."""
    bf = _Beamformer(Y=Y, X_mask=X_mask, N_mask=N_mask, debug=debug, ctx=ctx)
    if ban:
        return bf.X_hat_gev_ban
    return bf.X_hat_gev


def beamform_mvdr_souden_from_masks(Y, X_mask, N_mask, ban=False, debug=False, ctx=None):
    bf = _Beamformer(Y=Y, X_mask=X_mask, N_mask=N_mask, debug=debug, ctx=ctx)
    if ban:
        return bf.X_hat_mvdr_souden_ban
    return bf.X_hat_mvdr_souden
