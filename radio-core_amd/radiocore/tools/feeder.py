"""Feeder: overlapped host -> device ingest in front of Tuner.load.

The reference's DSP thread receives each one-second buffer in mapped memory
(`cusignal.get_shared_mem`, radiocore/tools/buffer.py:43, ringbuffer.py:51) and hands it straight to
`Tuner.load` (examples/multi_fm_server.py:95-98).  On the MI355X the buffer lives in page-locked host
memory (`Buffer(cuda=True)` / `RingBuffer(cuda=True)`) and crosses PCIe exactly once; this class keeps
`depth` device slots and a copy stream (librcfm's rcfm_feeder_*), so the copy of buffer i+1 runs under the
kernels of buffer i:

    feeder = Feeder(N)
    feeder.submit(host[0])
    for i in range(k):
        if i + 1 < k:
            feeder.submit(host[i + 1])        # H2D of the next buffer, on the copy stream
        with feeder.next() as x:              # the current stream waits for buffer i's copy only
            tuner.load(x)
            audio = tuner.run_all(numpy_output=False)

Steady state costs max(copy, compute) per buffer instead of their sum.
"""

import ctypes
from contextlib import contextmanager

import numpy as np

from radiocore._internal import Injector, hip

__all__ = ["Feeder"]


class Feeder(Injector):
    """`depth` device slots of `size` elements of `dtype`, fed from page-locked host arrays."""

    def __init__(self, size, dtype="complex64", depth=2):
        super().__init__(True)
        self._size = int(size)
        self._dtype = np.dtype(dtype)
        tdt = {np.dtype("complex64"): self._torch.complex64, np.dtype("float32"): self._torch.float32}[self._dtype]
        self._slots = [hip.empty((self._size,), tdt) for _ in range(int(depth))]
        ptrs = (ctypes.c_void_p * len(self._slots))(*[s.data_ptr() for s in self._slots])
        h = ctypes.c_void_p()
        hip.check(self._lib.rcfm_feeder_create(self._size * self._dtype.itemsize, len(self._slots), ptrs,
                                               ctypes.byref(h)))
        self._handle = hip.Handle(h, self._lib.rcfm_feeder_destroy)
        self._sources = []          # host arrays whose copies have not landed yet (kept alive)
        self._dropped = 0           # submitted buffers already dropped from _sources

    @property
    def depth(self):
        return len(self._slots)

    def submit(self, host_array):
        """Queue the copy of one buffer.  `host_array`: contiguous numpy array of this feeder's size and dtype,
        ideally page-locked (`Buffer(cuda=True).data`); pageable memory works but the copy is then staged
        by the driver and does not overlap.  Raises RuntimeError when all slots are in flight."""
        a = np.ascontiguousarray(host_array)
        if a.dtype != self._dtype or a.size != self._size:
            raise ValueError("input_sig size and input_size mismatch")
        hip.check(self._lib.rcfm_feeder_submit(self._handle.value, ctypes.c_void_p(a.ctypes.data)))
        self._sources.append(a)

    @contextmanager
    def next(self):
        """The oldest submitted buffer as a device tensor; work queued on the current stream inside the
        block is ordered after its copy, and the slot is handed back to the copy stream on exit."""
        p = ctypes.c_void_p()
        hip.check(self._lib.rcfm_feeder_acquire(self._handle.value, hip.stream(), ctypes.byref(p)))
        slot = next(s for s in self._slots if s.data_ptr() == p.value)
        try:
            yield slot
        finally:
            hip.check(self._lib.rcfm_feeder_release(self._handle.value, hip.stream()))
            self._drop_landed()

    def _drop_landed(self):
        # A source array is let go only once its H2D copy is known to have completed (rcfm_feeder_copied, a
        # host-side event query): the host may run far ahead of the GPU, and a temporary handed to submit()
        # (an ascontiguousarray copy, a slice nobody else holds) must outlive its DMA.
        n = ctypes.c_uint64()
        hip.check(self._lib.rcfm_feeder_copied(self._handle.value, ctypes.byref(n)))
        k = int(n.value) - self._dropped
        if k > 0:
            del self._sources[:k]
            self._dropped += k

    def close(self):
        """Destroy the feeder: waits for the copy stream, THEN lets go of the device slots (they are torch
        tensors: returning them to the caching allocator before the last copy has run would hand memory under a
        pending DMA to the next allocation) and of the host arrays."""
        if self._handle is not None and self._handle.value:
            status = self._lib.rcfm_feeder_destroy(self._handle.value)   # synchronises the copy stream
            self._handle.value = None
            hip.check(status)
        self._slots = []
        self._sources = []

    def __del__(self):
        try:
            self.close()
        except Exception:               # interpreter shutdown
            pass
