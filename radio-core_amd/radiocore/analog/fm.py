"""FM: generic FM demodulator (reference: radiocore/analog/fm.py:26-72)."""

from radiocore._internal import hip
from radiocore.analog._demod import Demodulator

__all__ = ["FM"]


class FM(Demodulator):
    """Phase discriminator + Decimate.  `deemphasis` is accepted and unused, as in the
    reference.  Output: float32 (output_size, 1)."""

    _KIND = hip.RCFM_FM
    _CHANNELS = 1

    def _shape(self, audio):
        return audio[0] if self._batch == 1 else audio
