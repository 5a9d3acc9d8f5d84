"""Chopper: iterate over equal slices of an array (reference: radiocore/tools/chopper.py:21-50)."""

__all__ = ["Chopper"]


class Chopper:
    def __init__(self, size, chunk_size):
        self._size = int(size)
        self._chunk_size = int(chunk_size)
        if self._size % self._chunk_size != 0:
            raise ValueError("cannot evenly divide array by chunk size "
                             f"({self._size}, {self._chunk_size})")

    @property
    def size(self):
        return self._size

    @property
    def chunk_size(self):
        return self._chunk_size

    def chop(self, input_arr):
        """Yield views (no copies) of consecutive chunk_size slices."""
        for start in range(0, self._size, self._chunk_size):
            yield input_arr[start:start + self._chunk_size]

    @staticmethod
    def get_to_da_choppa():
        return 'https://www.youtube.com/watch?v=Xs_OacEq2Sk'
