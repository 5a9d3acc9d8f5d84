"""RingBuffer: single-producer / single-consumer sample ring
(reference: radiocore/tools/ringbuffer.py:31-160).

Same observable behaviour as the reference -- put() copies in (an input larger than the
vacancy resets the ring and prints "overflow", or raises when allow_overflow=False), get()
blocks until enough samples arrived or `timeout` seconds passed (returns None then, True on
success), the backing array is visible as `.data` -- but the occupancy counter is guarded by
a condition variable instead of the third-party `atomics` module.
"""

import threading

from radiocore.tools import _hostmem

__all__ = ["RingBuffer"]


class RingBuffer:
    def __init__(self, capacity, dtype="complex64", cuda=False, print_overflow=True, allow_overflow=True):
        self._print_overflow = bool(print_overflow)
        self._allow_overflow = bool(allow_overflow)
        self._capacity = int(capacity)
        self._cuda = bool(cuda)
        self._dtype = dtype
        self._head = 0          # next write position
        self._tail = 0          # next read position
        self._count = 0
        self._epoch = 0         # bumped by every reset: transfers that straddle one are forgotten
        self._cv = threading.Condition()
        self._buffer, self._owner = _hostmem.zeros(self._capacity, dtype, self._cuda)

    @property
    def capacity(self):
        return self._capacity

    @property
    def occupancy(self):
        with self._cv:
            return self._count

    @property
    def vacancy(self):
        return self._capacity - self.occupancy

    @property
    def data(self):
        return self._buffer

    def reset(self):
        with self._cv:
            self._head = self._tail = self._count = 0
            self._epoch += 1

    def __str__(self):
        return str(self._buffer)

    def _split(self, start, size):
        """Lengths of the two contiguous runs a transfer of `size` starting at `start` takes."""
        first = min(size, self._capacity - start)
        return first, size - first

    def put(self, buffer):
        """Copies `buffer` in.  The lock only guards the indices: the copy itself runs outside it (single
        producer / single consumer: the regions being written and read are disjoint), so a consumer copying a
        long buffer out never stalls the producer callback, like the reference's lock-free ring."""
        size = len(buffer)
        if size > self._capacity:
            raise ValueError("Input buffer is bigger than ring capacity.")
        with self._cv:
            if size > self._capacity - self._count:
                if not self._allow_overflow:
                    raise ValueError("Overflow happened.")
                if self._print_overflow:
                    print("overflow")
                self._head = self._tail = self._count = 0
                self._epoch += 1
            head, epoch = self._head, self._epoch
        a, b = self._split(head, size)
        if a:
            self._buffer[head:head + a] = buffer[:a]
        if b:
            self._buffer[:b] = buffer[a:size]
        with self._cv:
            if epoch == self._epoch:            # (a concurrent reset() forgets this transfer)
                self._head = (head + size) % self._capacity
                self._count += size             # published after the copy
                self._cv.notify_all()

    def get(self, buffer, timeout=3.0):
        size = len(buffer)
        if size > self._capacity:
            raise ValueError("Input buffer is bigger than ring capacity.")
        with self._cv:
            if not self._cv.wait_for(lambda: self._count >= size, timeout):
                return None
            tail, epoch = self._tail, self._epoch
        a, b = self._split(tail, size)
        if a:
            buffer[:a] = self._buffer[tail:tail + a]
        if b:
            buffer[a:size] = self._buffer[:b]
        with self._cv:
            if epoch == self._epoch:            # an overflow reset in between already dropped these samples
                self._tail = (tail + size) % self._capacity
                self._count -= size
        return True
