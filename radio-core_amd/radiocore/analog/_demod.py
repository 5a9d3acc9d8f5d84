"""Shared plumbing of FM / MFM / WBFM: one librcfm demodulator handle per instance."""

import ctypes
import weakref

import numpy as np

from radiocore._internal import Injector, hip


class Demodulator(Injector):
    """input_size -> output_size demodulator backed by rcfm_demod_* (include/rcfm.h).

    `batch` > 1 is this build's extension: the instance then demodulates `batch`
    independent channels per call from a [batch, input_size] array and returns
    [batch, output_size, channels]; batch = 1 keeps the reference's shapes.
    """

    _KIND = None
    _CHANNELS = 1

    def __init__(self, input_size, output_size, deemphasis=75e-6, cuda=False, batch=1, chunk=0):
        self._cuda = cuda
        self._input_size = int(input_size)
        self._output_size = int(output_size)
        self._tau = float(deemphasis)
        self._batch = int(batch)
        self._chunk = int(chunk)
        super().__init__(cuda)
        # what rcfm_demod_create would refuse is refused here, at construction like the reference's constructors
        # (bandpass.py:50-52 designs the pilot filter in __init__) ...
        if self._batch < 1 or self._input_size < 2 or self._output_size < 1:
            raise ValueError("bad demodulator size")
        if self._KIND == hip.RCFM_WBFM:
            if (19e3 + 50) / (0.5 * self._input_size) >= 1.0:
                raise ValueError("Invalid cutoff frequency: frequencies must be greater than 0 and less than fs/2.")
            if self._input_size <= 3 * 41:
                raise ValueError("The length of the input vector x must be greater than padlen, which is 123.")
        # ... but the librcfm handle (plans, tables, workspaces: ~12 MB for a 240 kHz WBFM channel) is created on first
        # use: a Tuner with thousands of channels that only ever calls run_all() never needs the per-channel handles
        self._h = None
        self._arena = hip.current_arena()      # `with radiocore.tools.Arena(...)`: the handle is built inside it
        self._binding = None       # (weakref to the batched hip.Handle, channel index): Tuner._bind_states

    @property
    def _handle(self):
        if self._h is None:
            h = ctypes.c_void_p()
            with hip.bound(self._arena):
                hip.check(self._lib.rcfm_demod_create(self._KIND, self._batch, self._input_size, self._output_size,
                                                      self._tau, self._chunk, ctypes.byref(h)))
            self._h = hip.Handle(h, self._lib.rcfm_demod_destroy)
            self._apply_binding(move_history=0)      # new handle: the batched caller has carried the state so far
        return self._h

    def _bind(self, batched_handle, index):
        """Keep this demodulator's de-emphasis state in slot `index` of a Tuner's batched handle from now on
        (rcfm_demod_bind_state: one state per channel, whoever runs it -- deemphasis.py:48-49,64)."""
        self._binding = (weakref.ref(batched_handle), int(index))
        if self._h is not None:
            self._apply_binding(move_history=1)      # this object has run before: its history goes with it

    def _apply_binding(self, move_history):
        if self._binding is None:
            return
        owner = self._binding[0]()
        if owner is None or not owner.value:      # the tuner is gone: the state is this object's own again
            self._binding = None
            return
        hip.check(self._lib.rcfm_demod_bind_state(self._h.value, owner.value, self._binding[1], int(move_history),
                                                  hip.stream()))

    @property
    def channels(self):
        """Return the number of audio channels of the output."""
        return self._CHANNELS

    def _shape(self, x):
        raise NotImplementedError

    def _demodulate(self, input_sig):
        """-> device tensor [batch, output_size, channels]."""
        t = self._torch
        if self._batch == 1:
            if len(input_sig) != self._input_size:
                raise ValueError("input_sig size and input_size mismatch")
        x = hip.to_device(input_sig, t.complex64)
        if self._batch > 1 and tuple(x.shape) != (self._batch, self._input_size):
            raise ValueError("input_sig size and input_size mismatch")
        audio = hip.empty((self._batch, self._output_size, self._CHANNELS), t.float32)
        hip.check(self._lib.rcfm_demod_run(self._handle.value, 0, self._batch, hip.ptr(x), hip.ptr(audio),
                                           hip.stream()))
        return audio

    def reset(self):
        """Back to the reference's freshly constructed filter state."""
        hip.check(self._lib.rcfm_demod_reset_state(self._handle.value, hip.stream()))

    def state(self):
        """De-emphasis state [batch, channels, 50] (Deemphasis._state per leg)."""
        out = np.zeros((self._batch, self._CHANNELS, 50), np.float32)
        hip.check(self._lib.rcfm_demod_get_state(self._handle.value,
                                                 out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), hip.stream()))
        return out

    def taps(self):
        """(51 de-emphasis taps, 41 pilot band-pass taps) as designed by the library."""
        de = np.zeros(51, np.float32)
        pl = np.zeros(41, np.float32)
        fp = ctypes.POINTER(ctypes.c_float)
        hip.check(self._lib.rcfm_demod_get_taps(self._handle.value, de.ctypes.data_as(fp), pl.ctypes.data_as(fp)))
        return de, pl

    def run(self, input_sig, numpy_output=True):
        audio = self._demodulate(input_sig)
        audio = self._shape(audio)
        return self._result(audio, self._cuda and not numpy_output)
