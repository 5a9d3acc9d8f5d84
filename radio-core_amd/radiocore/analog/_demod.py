"""Shared plumbing of FM / MFM / WBFM: one librcfm demodulator handle per instance."""

import ctypes

import numpy as np

from radiocore._internal import Injector, hip


class Demodulator(Injector):
    """input_size -> output_size demodulator backed by rcfm_demod_* (include/rcfm.h).

    `batch` > 1 is this build's extension: the instance then demodulates `batch`
    independent channels per call from a [batch, input_size] array and returns
    [batch, output_size, channels]; batch = 1 keeps the reference's shapes.
    """

    _KIND = None
    _CHANNELS = 1

    def __init__(self, input_size, output_size, deemphasis=75e-6, cuda=False, batch=1, chunk=0):
        self._cuda = cuda
        self._input_size = int(input_size)
        self._output_size = int(output_size)
        self._tau = float(deemphasis)
        self._batch = int(batch)
        super().__init__(cuda)
        h = ctypes.c_void_p()
        hip.check(self._lib.rcfm_demod_create(self._KIND, self._batch, self._input_size, self._output_size,
                                              self._tau, int(chunk), ctypes.byref(h)))
        self._handle = hip.Handle(h, self._lib.rcfm_demod_destroy)

    @property
    def channels(self):
        """Return the number of audio channels of the output."""
        return self._CHANNELS

    def _shape(self, x):
        raise NotImplementedError

    def _demodulate(self, input_sig):
        """-> device tensor [batch, output_size, channels]."""
        t = self._torch
        if self._batch == 1:
            if len(input_sig) != self._input_size:
                raise ValueError("input_sig size and input_size mismatch")
        x = hip.to_device(input_sig, t.complex64)
        if self._batch > 1 and tuple(x.shape) != (self._batch, self._input_size):
            raise ValueError("input_sig size and input_size mismatch")
        audio = hip.empty((self._batch, self._output_size, self._CHANNELS), t.float32)
        hip.check(self._lib.rcfm_demod_run(self._handle.value, 0, self._batch, hip.ptr(x), hip.ptr(audio),
                                           hip.stream()))
        return audio

    def reset(self):
        """Back to the reference's freshly constructed filter state."""
        hip.check(self._lib.rcfm_demod_reset_state(self._handle.value, hip.stream()))

    def state(self):
        """De-emphasis state [batch, channels, 50] (Deemphasis._state per leg)."""
        out = np.zeros((self._batch, self._CHANNELS, 50), np.float32)
        hip.check(self._lib.rcfm_demod_get_state(self._handle.value,
                                                 out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), hip.stream()))
        return out

    def taps(self):
        """(51 de-emphasis taps, 41 pilot band-pass taps) as designed by the library."""
        de = np.zeros(51, np.float32)
        pl = np.zeros(41, np.float32)
        fp = ctypes.POINTER(ctypes.c_float)
        hip.check(self._lib.rcfm_demod_get_taps(self._handle.value, de.ctypes.data_as(fp), pl.ctypes.data_as(fp)))
        return de, pl

    def run(self, input_sig, numpy_output=True):
        audio = self._demodulate(input_sig)
        audio = self._shape(audio)
        return self._result(audio, self._cuda and not numpy_output)
