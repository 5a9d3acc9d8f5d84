"""Deemphasis: stateful 51-tap FIR (reference: radiocore/analog/deemphasis.py:26-66)."""

import numpy as np

from radiocore._internal import Injector, hip

__all__ = ["Deemphasis"]


def design(input_size, rate, dtype="float32"):
    """Taps and initial state, deemphasis.py:37-49 (host-side in the reference too).

    The one-pole IIR (1-x)/(z-x), x = exp(-1/(fs tau)), truncated to its first 51
    impulse-response samples; initial state = lfilter_zi = tail sums of the taps.
    """
    x = np.exp(-1.0 / (input_size * rate))
    b = np.zeros(51)
    s, u = 0.0, 1.0
    for i in range(51):
        b[i] = (1.0 - x) * s
        s, u = x * s + u, 0.0
    taps = b.astype(dtype)
    zi = np.cumsum(taps[:0:-1])[::-1].astype(dtype)
    return taps, zi


class Deemphasis(Injector):
    """lfilter(taps, 1, x, zi=state) on the GPU (rcfm_lfilter_fir); state stays on the device."""

    def __init__(self, input_size, rate=75e-6, dtype="float32", cuda=False):
        self._cuda = cuda
        self._dtype = dtype
        self._rate = rate
        self._input_size = int(input_size)
        super().__init__(cuda)
        taps, zi = design(self._input_size, self._rate, self._dtype)
        self._taps = (taps, np.array(1.0, dtype=self._dtype))
        self._state_dev = hip.to_device(zi, self._torch.float32)

    @property
    def _state(self):
        return self._result(self._state_dev, self._cuda)

    def run(self, input_sig):
        if len(input_sig) != self._input_size:
            raise ValueError("input_sig size and input_size mismatch")
        x = hip.to_device(input_sig, self._torch.float32)
        y = hip.empty((self._input_size,), self._torch.float32)
        taps, taps_p = hip.float_array(self._taps[0])
        hip.check(self._lib.rcfm_lfilter_fir(1, self._input_size, taps_p, len(taps), hip.ptr(self._state_dev),
                                             hip.ptr(x), hip.ptr(y), hip.stream()))
        return self._result(y, self._cuda)
