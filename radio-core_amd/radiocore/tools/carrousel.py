"""Carrousel: a ring of reusable items (reference: radiocore/tools/carrousel.py:27-118).

enqueue()/dequeue() are context managers that lend out the next free / oldest filled item
without ever discarding it; enqueueing into a full ring drops the oldest item and counts an
overflow.  Items that are `Buffer`s are lent through their own `consume()`.
"""

from contextlib import contextmanager

from radiocore.tools.buffer import Buffer

__all__ = ["Carrousel"]


class Carrousel:
    def __init__(self, items, print_overflow=True):
        self._items = items
        self._capacity = len(items)
        self._print_overflow = bool(print_overflow)
        self._head = 0          # oldest filled slot
        self._tail = 0          # next slot to fill
        self._occupancy = 0
        self._overflow = 0

    @property
    def occupancy(self):
        return self._occupancy

    @property
    def capacity(self):
        return self._capacity

    @property
    def is_empty(self):
        return self._occupancy == 0

    @property
    def is_full(self):
        return self._occupancy >= self._capacity

    @property
    def overflow(self):
        return self._overflow

    @property
    def is_healthy(self):
        return self._occupancy >= 1

    def reset(self):
        self._head = self._tail = self._occupancy = 0

    def __str__(self):
        return str(self._items)

    @contextmanager
    def _lend(self, item):
        if isinstance(item, Buffer):
            with item.consume() as arr:
                yield arr
        else:
            yield item

    @contextmanager
    def enqueue(self):
        if self.is_full:                      # overwrite the oldest item
            self._overflow += 1
            self._occupancy -= 1
            self._head = (self._head + 1) % self._capacity
            if self._print_overflow:
                print("overflow")
        try:
            with self._lend(self._items[self._tail]) as item:
                yield item
        finally:
            self._occupancy += 1
            self._tail = (self._tail + 1) % self._capacity

    @contextmanager
    def dequeue(self):
        if self.is_empty:
            raise ValueError("carrousel is empty")
        try:
            with self._lend(self._items[self._head]) as item:
                yield item
        finally:
            self._occupancy -= 1
            self._head = (self._head + 1) % self._capacity
