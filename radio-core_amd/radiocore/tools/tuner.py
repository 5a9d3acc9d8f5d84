"""Tuner: FFT channelizer (reference: radiocore/tools/tuner.py:9-174).

Channel bookkeeping stays host-side Python with the reference's arithmetic; the
wideband FFT, the per-channel bin gather / weight / inverse FFT and (for
``run_all``) the demodulators run in librcfm.so.
"""

import ctypes
from dataclasses import dataclass
from typing import List

import numpy as np

from radiocore._internal import Injector, hip

__all__ = ["Tuner", "Channel"]

_KINDS = {"FM": hip.RCFM_FM, "MFM": hip.RCFM_MFM, "WBFM": hip.RCFM_WBFM}


@dataclass
class Channel:
    """Frequency boundaries of one channel plus its demodulator (tuner.py:9-36)."""

    index: int
    bandwidth: float
    demodulator: None
    lower_frequency: float
    center_frequency: float
    higher_frequency: float

    @property
    def address_bytes(self) -> bytes:
        return int(self.center_frequency).to_bytes(4, byteorder='little')


class Tuner(Injector):
    """Channelizes one-second wideband buffers.

    Reference-compatible use: add_channel / request_bandwidth / load / run(i).
    Batched use (this build): ``run_all()`` after ``load`` returns the audio of
    every channel in one call -- the loop of examples/multi_fm_server.py:100-106.
    """

    def __init__(self, cuda: bool = False):
        self._cuda = cuda
        super().__init__(self._cuda)
        self._input_frequency = 0.0
        self._input_bandwidth = 0.0
        self._bounds: List[Channel] = []
        self._handle = None        # librcfm tuner, rebuilt when the geometry changes
        self._handle_key = None
        self._shard = None         # (first, count) declared by shard()
        self._loaded_size = None
        self._win_size = None      # length of the reference's cached window (set by the first run)
        self._batched = {}         # (kind, C, B, A, tau, chunk) -> demod handle of run_all / run_each

    @property
    def input_frequency(self) -> float:
        """Return the center frequency of the input data."""
        return self._input_frequency

    @property
    def input_bandwidth(self) -> float:
        """Return the bandwidth of the input data."""
        return self._input_bandwidth

    def channels(self) -> List[Channel]:
        """Return list of registered channels."""
        return self._bounds

    def request_bandwidth(self, bandwidth: float):
        """Override the calculated bandwidth (must not be below it)."""
        if bandwidth < self._input_bandwidth:
            raise ValueError(f"requested bandwidth ({bandwidth}) is too low, "
                             f"minimum is {self._input_bandwidth}")
        self._input_bandwidth = bandwidth

    def add_channel(self, frequency: float, bandwidth: float, demodulator):
        """Register a channel; recalculates the input geometry."""
        self._bounds.append(Channel(
            index=len(self._bounds),
            bandwidth=bandwidth,
            demodulator=demodulator,
            lower_frequency=(frequency - (bandwidth / 2)),
            center_frequency=frequency,
            higher_frequency=(frequency + (bandwidth / 2)),
        ))
        self._recalculate()

    def reset(self):
        """Forget all channels (raises ValueError like the reference: min() of nothing)."""
        self._bounds = []
        self._recalculate()

    def _recalculate(self):
        # tuner.py:163-174
        lower = min([ch.lower_frequency for ch in self._bounds])
        higher = max([ch.higher_frequency for ch in self._bounds])
        self._input_frequency = (lower + higher) / 2
        self._input_bandwidth = (higher - lower)
        mean_bandwidth = sum([ch.bandwidth for ch in self._bounds])
        mean_bandwidth //= len(self._bounds)
        self._input_bandwidth += (self._input_bandwidth * -1) % mean_bandwidth

    # ---- device side ---------------------------------------------------------

    def _device_tuner(self, n):
        rolls = [int(self._input_frequency - ch.center_frequency) for ch in self._bounds]
        bws = [int(ch.bandwidth) for ch in self._bounds]
        key = (n, tuple(rolls), tuple(bws))
        if key != self._handle_key:
            roll_a = (ctypes.c_int64 * len(rolls))(*rolls)
            bw_a = (ctypes.c_int32 * len(bws))(*bws)
            h = ctypes.c_void_p()
            hip.check(self._lib.rcfm_tuner_create(n, len(rolls), roll_a, bw_a, ctypes.byref(h)))
            self._handle = hip.Handle(h, self._lib.rcfm_tuner_destroy)
            self._handle_key = key
            self._loaded_size = None
            if self._shard is not None:
                hip.check(self._lib.rcfm_tuner_shard(h, self._shard[0], self._shard[1]))
        return self._handle.value

    def shard(self, first, count):
        """Multi-GPU only (no reference counterpart): this process will run channels
        [first, first + count) and nothing else, so `load` may keep just the part of the spectrum
        they read (sharding.channel_range gives the range of a rank)."""
        if first < 0 or count < 0 or first + count > len(self._bounds):
            raise IndexError("channel range outside the tuner")
        self._shard = (int(first), int(count))
        if self._handle is not None:
            hip.check(self._lib.rcfm_tuner_shard(self._handle.value, int(first), int(count)))

    def load(self, input_signal):
        """Forward FFT of the one-second buffer; kept on the device (tuner.py:126-138)."""
        x = hip.to_device(input_signal, self._torch.complex64)
        n = int(x.shape[0])
        hip.check(self._lib.rcfm_tuner_load(self._device_tuner(n), hip.ptr(x), hip.stream()))
        self._input = x            # keeps the buffer alive until the FFT has consumed it
        self._loaded_size = n

    def _ready(self):
        if self._loaded_size is None:
            raise RuntimeError("Tuner.run called before Tuner.load")
        # tuner.py:155-157 builds the spectral window on the first run() with int(input_bandwidth) points and never
        # refreshes it; scipy.signal.resample then refuses any buffer of another length -- also after a later
        # request_bandwidth().  Same behaviour here: the first run fixes the size.
        if self._win_size is None:
            self._win_size = int(self._input_bandwidth)
        if self._loaded_size != self._win_size:
            raise ValueError('window must have the same length as data')
        return self._device_tuner(self._loaded_size)

    def run(self, channel_index: int):
        """Channelized complex64 signal of one channel (tuner.py:140-161)."""
        channel = self._bounds[int(channel_index)]
        handle = self._ready()
        out = hip.empty((int(channel.bandwidth),), self._torch.complex64)
        hip.check(self._lib.rcfm_tuner_run(handle, int(channel_index), 1, hip.ptr(out), hip.stream()))
        return self._result(out, self._cuda)

    def run_channels(self, first: int, count: int):
        """[count, B] complex64 device tensor for a range of equal-bandwidth channels."""
        handle = self._ready()
        if count <= 0 or first < 0 or first + count > len(self._bounds):
            raise IndexError("list index out of range")
        out = hip.empty((count, int(self._bounds[first].bandwidth)), self._torch.complex64)
        hip.check(self._lib.rcfm_tuner_run(handle, first, count, hip.ptr(out), hip.stream()))
        return out

    def run_all(self, numpy_output: bool = True, chunk: int = 0):
        """Audio of every channel, [C, A, ch] float32, in channel-index order.

        All channels must carry demodulators of one class and geometry.  The
        de-emphasis state of this batched path lives in the tuner (one state per
        channel, carried from buffer to buffer) and is independent of the
        per-channel demodulator instances.

        After ``shard(first, count)`` only that range is run (its spectrum rows are all ``load``
        kept) and the result is [count, A, ch] -- the block sharding.gather_audio expects.
        """
        handle = self._ready()
        demods = [ch.demodulator for ch in self._bounds]
        geo = {self._geometry(d) for d in demods}
        if len(geo) != 1 or None in geo:
            raise ValueError("run_all needs one demodulator class and geometry for all channels")
        kind, B, A, tau = next(iter(geo))
        ch = 2 if kind == hip.RCFM_WBFM else 1
        first, count = self._shard if self._shard is not None else (0, len(demods))
        audio = hip.empty((count, A, ch), self._torch.float32)
        hip.check(self._lib.rcfm_pipeline_run(handle, self._batched_demod(kind, B, A, tau, chunk), first, count,
                                              hip.ptr(audio), hip.stream()))
        return self._result(audio, self._cuda and not numpy_output)

    def run_each(self, numpy_output: bool = True):
        """What the reference's loop collects (examples/multi_fm_server.py:100-106,
        ``[ch.demodulator.run(tuner.run(ch.index)) for ch in tuner.channels()]``): a list with one audio array per
        channel, shaped as the reference's demodulators return it: [A, 1] for FM / MFM, [1, A, 2] for WBFM (wbfm.py:94 dstack) -- the shapes of FM.run / MFM.run / WBFM.run here.

        The channels may differ in demodulator class, bandwidth and audio rate (broadcast stations next to
        narrow-band ones): consecutive channels of one class and geometry run as one batched sequence, and the
        de-emphasis state is the tuner's, per channel, as in ``run_all``.  After ``shard(first, count)`` the list
        covers that range only.
        """
        handle = self._ready()
        first, count = self._shard if self._shard is not None else (0, len(self._bounds))
        out = []
        i = first
        while i < first + count:
            g = self._geometry(self._bounds[i].demodulator)
            if g is None:
                raise ValueError("run_each needs an FM, MFM or WBFM demodulator on every channel")
            j = i + 1
            while j < first + count and self._geometry(self._bounds[j].demodulator) == g and \
                    int(self._bounds[j].bandwidth) == int(self._bounds[i].bandwidth):
                j += 1
            kind, B, A, tau = g
            ch = 2 if kind == hip.RCFM_WBFM else 1
            audio = hip.empty((j - i, A, ch), self._torch.float32)
            hip.check(self._lib.rcfm_pipeline_run(handle, self._batched_demod(kind, B, A, tau, 0), i, j - i,
                                                  hip.ptr(audio), hip.stream()))
            block = self._result(audio, self._cuda and not numpy_output)
            out.extend(block[k:k + 1] if ch == 2 else block[k] for k in range(j - i))
            i = j
        return out

    @staticmethod
    def _geometry(demod):
        kind = _KINDS.get(type(demod).__name__)
        if kind is None:
            return None
        return kind, demod._input_size, demod._output_size, demod._tau

    def _batched_demod(self, kind, B, A, tau, chunk):
        # one handle per geometry, sized for all channels: rcfm_pipeline_run addresses the channels of tuner and
        # demodulator by the same index, so the per-channel state survives regrouping
        key = (kind, len(self._bounds), B, A, tau, int(chunk))
        if key not in self._batched:
            h = ctypes.c_void_p()
            hip.check(self._lib.rcfm_demod_create(kind, len(self._bounds), B, A, tau, int(chunk), ctypes.byref(h)))
            self._batched[key] = hip.Handle(h, self._lib.rcfm_demod_destroy)
        return self._batched[key].value
