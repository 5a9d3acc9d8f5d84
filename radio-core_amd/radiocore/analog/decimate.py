"""Decimate: FFT-domain resampler (reference: radiocore/analog/decimate.py:21-50)."""

import ctypes

import numpy as np

from radiocore._internal import Injector, hip

__all__ = ["Decimate"]


class Decimate(Injector):
    """scipy.signal.resample with the fftshifted Hamming spectral window, on the GPU.

    Real input gives float32, complex input complex64, like the reference.
    Plans are built on first use per input kind (rcfm_resampler_create).
    """

    def __init__(self, input_size, output_size, cuda=False):
        self._cuda = cuda
        self._input_size = int(input_size)
        self._output_size = int(output_size)
        super().__init__(cuda)
        self._plans = {}

    def _plan(self, is_complex):
        if is_complex not in self._plans:
            h = ctypes.c_void_p()
            hip.check(self._lib.rcfm_resampler_create(1, self._input_size, self._output_size,
                                                      int(is_complex), ctypes.byref(h)))
            self._plans[is_complex] = hip.Handle(h, self._lib.rcfm_resampler_destroy)
        return self._plans[is_complex].value

    def run(self, input_sig):
        if len(input_sig) != self._input_size:
            raise ValueError("input_sig size and input_size mismatch")
        t = self._torch
        x = hip.to_device(input_sig)
        is_complex = x.is_complex()
        x = x.to(t.complex64 if is_complex else t.float32)
        y = hip.empty((self._output_size,), x.dtype)
        hip.check(self._lib.rcfm_resampler_run(self._plan(is_complex), hip.ptr(x), hip.ptr(y), hip.stream()))
        return self._result(y, self._cuda)
