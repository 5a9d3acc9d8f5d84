"""Host-side filter design.

Band-pass taps are designed on the host in the reference as well
(bandpass.py:50-54 calls scipy.signal.firwin even when cuda=True); this is the
same windowed-sinc recipe written out, so the package needs neither scipy nor
the test oracle.
"""

import numpy as np

_COSINE_WINDOWS = {"hamm": 0.54, "hamming": 0.54, "hann": 0.5, "hanning": 0.5}


def symmetric_window(name, numtaps):
    if name in ("boxcar", "rect", "rectangular", "ones"):
        return np.ones(numtaps)
    if name not in _COSINE_WINDOWS:
        raise ValueError("Unknown window type.")
    a0 = _COSINE_WINDOWS[name]
    if numtaps == 1:
        return np.ones(1)
    return a0 - (1.0 - a0) * np.cos(2.0 * np.pi * np.arange(numtaps) / (numtaps - 1))


def firwin_bandpass(numtaps, lo, hi, window="hamm"):
    """firwin(numtaps, [lo, hi], pass_zero=False, window=window); lo/hi relative to Nyquist."""
    if not (0.0 < lo < hi < 1.0):
        raise ValueError("Invalid cutoff frequency: frequencies must be greater than 0 "
                         "and less than fs/2.")
    if numtaps % 2 == 0:
        raise ValueError("A filter with an even number of coefficients must have zero "
                         "response at the Nyquist frequency.")
    mm = np.arange(numtaps) - 0.5 * (numtaps - 1)
    h = (hi * np.sinc(hi * mm) - lo * np.sinc(lo * mm)) * symmetric_window(window, numtaps)
    return h / np.sum(h * np.cos(np.pi * mm * 0.5 * (lo + hi)))
