"""radiocore, MI355X build: the reference's class surface over librcfm.so."""

from radiocore.analog import *
from radiocore.tools import *


def HasCuda():
    """True when the HIP backend is usable (reference: radiocore/__init__.py:6-26,
    where it probes for cupy + cusignal)."""
    try:
        import ctypes
        from radiocore._internal import hip
        n = ctypes.c_int(0)
        return hip.lib().rcfm_device_count(ctypes.byref(n)) == 0 and n.value > 0
    except Exception:
        return False


__version__ = '1.0.0'
