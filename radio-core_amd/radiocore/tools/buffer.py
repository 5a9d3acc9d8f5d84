"""Buffer: a fixed host array with an optional mutex (reference: radiocore/tools/buffer.py:29-93)."""

import threading
from contextlib import contextmanager

from radiocore.tools import _hostmem

__all__ = ["Buffer"]


class Buffer:
    """`size` elements of `dtype`, zero-initialised.  cuda=True allocates page-locked memory
    (DMA-able to the GPU).  With lock=True `consume()` serialises access."""

    def __init__(self, size, dtype="complex64", lock=False, cuda=False):
        self._lock = bool(lock)
        self._cuda = bool(cuda)
        self._dtype = dtype
        self._size = int(size)
        self._mtx = threading.Lock() if self._lock else None
        self._buffer, self._owner = _hostmem.zeros(self._size, dtype, self._cuda)

    @property
    def dtype(self):
        return self._buffer.dtype

    @property
    def is_cuda(self):
        return self._cuda

    @property
    def size(self):
        return self._size

    def __len__(self):
        return self._size

    @property
    def is_locked(self):
        if not self._lock:
            raise ValueError("locking is not enabled in this instance")
        return self._mtx.locked()

    @property
    def data(self):
        return self._buffer

    @contextmanager
    def consume(self):
        """Yield the array itself (not a copy); holds the mutex for the duration when enabled."""
        if self._lock:
            with self._mtx:
                yield self._buffer
        else:
            yield self._buffer
