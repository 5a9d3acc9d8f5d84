"""WBFM: broadcast stereo FM (reference: radiocore/analog/wbfm.py:32-105)."""

from radiocore._internal import hip
from radiocore.analog._demod import Demodulator

__all__ = ["WBFM"]


class WBFM(Demodulator):
    """FM -> 19 kHz pilot band-pass -> Hilbert 'PLL' -> 38 kHz mix -> 2 x Decimate ->
    2 x de-emphasis -> joint DC removal -> clip.  Output: float32 (1, output_size, 2)."""

    _KIND = hip.RCFM_WBFM
    _CHANNELS = 2

    def _shape(self, audio):
        return audio      # [1, A, 2] for batch 1 is already the reference's dstack shape
