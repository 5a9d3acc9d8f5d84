"""Bandpass: zero-phase FIR band-pass (reference: radiocore/analog/bandpass.py:29-74)."""

import numpy as np

from radiocore._internal import Injector, hip
from radiocore._internal.design import firwin_bandpass

__all__ = ["Bandpass"]


class Bandpass(Injector):
    """firwin design on the host, filtfilt on the GPU (rcfm_filtfilt)."""

    def __init__(self, input_size, start_freq, stop_freq, dtype="float32", num_taps=61,
                 window="hamm", cuda=False):
        self._cuda = cuda
        self._dtype = dtype
        self._window = window
        self._num_taps = int(num_taps)
        self._input_size = int(input_size)
        self._stop_freq = float(stop_freq)
        self._start_freq = float(start_freq)
        super().__init__(cuda)
        nyq = 0.5 * self._input_size
        b = firwin_bandpass(self._num_taps, self._start_freq / nyq, self._stop_freq / nyq, self._window)
        self._taps = (np.array(b, dtype=self._dtype), np.array([1.0], dtype=self._dtype))

    def run(self, input_sig):
        if len(input_sig) != self._input_size:
            raise ValueError("input_sig size and input_size mismatch")
        x = hip.to_device(input_sig, self._torch.float32)
        y = hip.empty((self._input_size,), self._torch.float32)
        taps, taps_p = hip.float_array(self._taps[0])
        hip.check(self._lib.rcfm_filtfilt(1, self._input_size, taps_p, len(taps), hip.ptr(x), hip.ptr(y),
                                          hip.stream()))
        return self._result(y, self._cuda)
