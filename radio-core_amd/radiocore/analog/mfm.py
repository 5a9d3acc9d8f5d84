"""MFM: broadcast mono FM (reference: radiocore/analog/mfm.py:29-71)."""

from radiocore._internal import hip
from radiocore.analog._demod import Demodulator

__all__ = ["MFM"]


class MFM(Demodulator):
    """FM -> de-emphasis -> DC removal -> clip.  Output: float32 (output_size, 1)."""

    _KIND = hip.RCFM_MFM
    _CHANNELS = 1

    def _shape(self, audio):
        return audio[0] if self._batch == 1 else audio
