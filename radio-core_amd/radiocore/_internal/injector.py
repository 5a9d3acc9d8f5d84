"""Backend seam of the package (reference: radiocore/_internal/injector.py:16-29).

The reference picks numpy+scipy or cupy+cusignal here.  This build has exactly
one backend: librcfm.so on an MI355X.  ``cuda`` keeps its reference meaning for
the *results*: with ``cuda=False`` arrays come back as numpy arrays on the host,
with ``cuda=True`` they stay on the device (torch tensors stand in for the
reference's cupy arrays) unless a demodulator is asked for ``numpy_output``.
Either way the arithmetic runs in the HIP kernels; nothing is computed on the
CPU and construction fails loudly when the library or the device is missing.
"""

from radiocore._internal import hip

__all__ = ["Injector"]


class Injector:
    """Binds the HIP backend into self (``_hip``: binding module, ``_lib``: the ABI)."""

    def __init__(self, cuda=False):
        self._hip = hip
        self._lib = hip.lib()
        self._torch = hip.torch()

    def _result(self, tensor, device_output):
        """Device tensor -> what the caller of the reference would receive."""
        return tensor if device_output else hip.to_host(tensor)
