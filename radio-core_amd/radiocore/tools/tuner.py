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
        self._arena = hip.current_arena()     # `with radiocore.tools.Arena(...)`: the device handles are built inside it
        self._input_frequency = 0.0
        self._input_bandwidth = 0.0
        self._bounds: List[Channel] = []
        self._handle = None        # librcfm tuner, rebuilt when the geometry changes
        self._handle_key = None
        self._shard = None         # (first, count) declared by shard()
        self._loaded_size = None
        self._win_size = None      # length of the reference's cached window (set by the first run)
        self._batched = {}         # (kind, C, B, A, tau, chunk) -> demod handle of run_all / run_each
        # Everything below is derived from the channel list and rebuilt only when add_channel / reset change it
        # (`_version`): load / run / run_all do no per-channel Python work in the steady state.
        self._version = 0
        self._lo = self._hi = self._bw_sum = None     # running min / max / sum of _recalculate
        self._abi_arrays = None    # (version, rolls, bws) as ctypes arrays
        self._plan = None          # (version, shard) -> run_all / run_each launch plan
        self._uniform = (None, None)   # (version, the one geometry all channels share or None)
        self._state_owner = {}     # (kind, C, B, A, tau) -> the batched handle whose buffer holds that geometry's state
        self._bound_version = {}   # batched key -> channel-list version its demodulators were bound at
        self._state_fence = False  # Lanes: this tuner's batched handles run on several streams (RCFM_OPT_STATE_FENCE)

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
        ch = Channel(
            index=len(self._bounds),
            bandwidth=bandwidth,
            demodulator=demodulator,
            lower_frequency=(frequency - (bandwidth / 2)),
            center_frequency=frequency,
            higher_frequency=(frequency + (bandwidth / 2)),
        )
        self._bounds.append(ch)
        self._recalculate(ch)

    def reset(self):
        """Forget all channels (raises ValueError like the reference: min() of nothing)."""
        self._bounds = []
        self._lo = self._hi = self._bw_sum = None
        self._recalculate()

    def _recalculate(self, added=None):
        # tuner.py:163-174.  The reference recomputes min / max / sum over every channel on each add_channel (O(C^2)
        # to set up C channels: 2.9 s for 8192); the running forms below give the same floats in O(1) per channel: min
        # and max are exact, and the running total equals sum() whenever the bandwidths are integer-valued floats below
        # 2^53 (every partial sum is exact) -- the only bandwidths with a meaning here, since int(bandwidth) is the
        # channel's sample count (tuner.py:153).  For non-integer bandwidths Python >= 3.12's compensated sum() may
        # differ from the running total in the last bit.  Channel objects are not expected to change after add_channel
        # (the device handle is keyed by the list's version, not by its contents).
        self._version += 1
        if added is not None and self._lo is not None:
            self._lo = min(self._lo, added.lower_frequency)
            self._hi = max(self._hi, added.higher_frequency)
            self._bw_sum = self._bw_sum + added.bandwidth
        else:
            self._lo = min([ch.lower_frequency for ch in self._bounds])
            self._hi = max([ch.higher_frequency for ch in self._bounds])
            self._bw_sum = sum([ch.bandwidth for ch in self._bounds])
        lower, higher = self._lo, self._hi
        self._input_frequency = (lower + higher) / 2
        self._input_bandwidth = (higher - lower)
        mean_bandwidth = self._bw_sum
        mean_bandwidth //= len(self._bounds)
        self._input_bandwidth += (self._input_bandwidth * -1) % mean_bandwidth

    # ---- device side ---------------------------------------------------------

    def _device_tuner(self, n):
        # rolls depend on the channel list only (input_frequency is recomputed by add_channel, never by
        # request_bandwidth): the handle is keyed by (n, _version) and the O(C) lists are built once per version
        key = (n, self._version)
        if key != self._handle_key:
            if self._abi_arrays is None or self._abi_arrays[0] != self._version:
                rolls = [int(self._input_frequency - ch.center_frequency) for ch in self._bounds]
                bws = [int(ch.bandwidth) for ch in self._bounds]
                self._abi_arrays = (self._version, (ctypes.c_int64 * len(rolls))(*rolls),
                                    (ctypes.c_int32 * len(bws))(*bws))
            _, roll_a, bw_a = self._abi_arrays
            h = ctypes.c_void_p()
            with hip.bound(self._arena):
                hip.check(self._lib.rcfm_tuner_create(n, len(self._bounds), roll_a, bw_a, ctypes.byref(h)))
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

    def load(self, input_signal, whole=False):
        """Forward FFT of the one-second buffer; kept on the device (tuner.py:126-138).

        whole=True (multi-GPU, rotating FFT owner): keep every row any channel reads although this process is
        sharded -- the spectrum is about to be handed to the other ranks (sharding.SpectrumRing)."""
        x = hip.to_device(input_signal, self._torch.complex64)
        n = int(x.shape[0])
        h = self._device_tuner(n)
        if whole and self._shard is not None:
            hip.check(self._lib.rcfm_tuner_shard(h, 0, len(self._bounds)))
            try:
                hip.check(self._lib.rcfm_tuner_load(h, hip.ptr(x), hip.stream()))
            finally:      # a failed load must not leave the handle sharded to every channel
                hip.check(self._lib.rcfm_tuner_shard(h, self._shard[0], self._shard[1]))
        else:
            hip.check(self._lib.rcfm_tuner_load(h, hip.ptr(x), hip.stream()))
        self._input = x            # keeps the buffer alive until the FFT has consumed it
        self._loaded_size = n

    # ---- the spectrum as an object that can travel (rcfm_tuner_attach_spectrum / _window / _adopt) ---------------

    def spectrum_slot(self, n):
        """Device storage for one spectrum of an n-sample buffer: complex64 [halo + n + halo] (torch owns it)."""
        halo, nn = ctypes.c_int64(), ctypes.c_int64()
        hip.check(self._lib.rcfm_tuner_spectrum_layout(self._device_tuner(int(n)), ctypes.byref(halo), ctypes.byref(nn)))
        t = hip.empty((int(nn.value) + 2 * int(halo.value),), self._torch.complex64)
        t.rcfm_halo = int(halo.value)
        return t

    def _same_handle(self, n, what):
        """The device handle for length n -- refusing to REBUILD it: a handle of another length would silently drop the
        attached spectrum and the loaded state (the handle is keyed by (n, channel-list version))."""
        if self._handle is not None and self._handle_key is not None and self._handle_key[0] != int(n) and \
                self._handle_key[1] == self._version:
            raise ValueError("%s: this tuner's device handle was built for %d-sample buffers, not %d"
                             % (what, self._handle_key[0], int(n)))
        return self._device_tuner(int(n))

    def attach(self, slot, n, loaded=None):
        """Use `slot` (from spectrum_slot) as the spectrum from now on; loaded = (first, count): it already holds the
        bins those channels read (None: nothing yet -- a load or adopt must follow)."""
        first, count = loaded if loaded is not None else (0, 0)
        hip.check(self._lib.rcfm_tuner_attach_spectrum(self._same_handle(n, "attach"), hip.ptr(slot), int(first), int(count)))
        self._slot = slot          # keeps the storage alive while the handle points at it
        self._loaded_size = int(n) if loaded is not None else None

    def detach(self, n):
        """Back to the handle's own spectrum storage (nothing loaded)."""
        hip.check(self._lib.rcfm_tuner_attach_spectrum(self._same_handle(n, "detach"), None, 0, 0))
        self._slot = None
        self._loaded_size = None

    def window_slot(self, n, first, count):
        """Device storage for just the bins channels [first, first + count) read -- complex64 [halo + nbins + halo], the
        window's first bin at element `halo` -- or None when that window wraps around the ends of the spectrum (such a
        range needs a whole spectrum_slot).  What a rank that never owns a buffer keeps per buffer in flight."""
        halo, nb = ctypes.c_int64(), ctypes.c_int64()
        rc = self._lib.rcfm_tuner_window_layout(self._device_tuner(int(n)), int(first), int(count), ctypes.byref(halo),
                                                ctypes.byref(nb))
        if rc != 0:
            return None
        t = hip.empty((int(nb.value) + 2 * int(halo.value),), self._torch.complex64)
        t.rcfm_halo = int(halo.value)
        t.rcfm_window = (int(first), int(count))
        return t

    def attach_window(self, slot, n):
        """Use `slot` (from window_slot) as the spectrum of its channel range; adopt() must follow once the bins are in."""
        first, count = slot.rcfm_window
        hip.check(self._lib.rcfm_tuner_attach_window(self._same_handle(n, "attach_window"), hip.ptr(slot), first, count))
        self._slot = slot
        self._loaded_size = None

    def window(self, n, first, count):
        """(first_bin, nbins): the bins of an n-point spectrum that channels [first, first + count) read, modulo n."""
        fb, nb = ctypes.c_int64(), ctypes.c_int64()
        hip.check(self._lib.rcfm_tuner_window(self._device_tuner(int(n)), int(first), int(count), ctypes.byref(fb),
                                              ctypes.byref(nb)))
        return int(fb.value), int(nb.value)

    def adopt(self, n, first, count):
        """The caller has written window(n, first, count) into the attached slot: accept it as loaded."""
        hip.check(self._lib.rcfm_tuner_adopt(self._same_handle(n, "adopt"), int(first), int(count), hip.stream()))
        self._loaded_size = int(n)

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

    def _launch_plan(self):
        """Groups of consecutive channels that share demodulator class, geometry and bandwidth, for the declared
        shard: [(first, count, kind, B, A, tau)], built once per channel-list version (O(C)) -- the steady
        state of run_all / run_each does no per-channel Python work.  Replacing ``channel.demodulator`` by hand
        after add_channel is not seen until the next add_channel / reset / shard (the reference has no such
        cache because it has no batched call)."""
        key = (self._version, self._shard)
        if self._plan is None or self._plan[0] != key:
            first, count = self._shard if self._shard is not None else (0, len(self._bounds))
            groups = []
            i = first
            while i < first + count:
                g = self._geometry(self._bounds[i].demodulator)
                bw = int(self._bounds[i].bandwidth)
                j = i + 1
                while g is not None and j < first + count and int(self._bounds[j].bandwidth) == bw and \
                        self._geometry(self._bounds[j].demodulator) == g:
                    j += 1
                groups.append((i, j - i) + (g if g is not None else (None, None, None, None)))
                i = j
            self._plan = (key, groups, first, count)
        return self._plan[1], self._plan[2], self._plan[3]

    def run_all(self, numpy_output: bool = True, chunk: int = 0):
        """Audio of every channel, [C, A, ch] float32, in channel-index order.

        All channels must carry demodulators of one class and geometry.  The
        de-emphasis state is ONE per channel, carried from buffer to buffer and
        shared with the channel's demodulator object (``_bind_states``), so
        ``ch.demodulator.run(tuner.run(i))`` and this call may alternate.

        After ``shard(first, count)`` only that range is run (its spectrum rows are all ``load``
        kept) and the result is [count, A, ch] -- the block sharding.gather_audio expects.
        """
        return self._result(self._run_all_device(chunk), self._cuda and not numpy_output)

    def _run_all_device(self, chunk=0):
        # run_all's launches on the current stream; the device tensor they fill (Lanes collects it later)
        handle = self._ready()
        _, first, count = self._launch_plan()
        geo = self._plan_uniform()     # ALL channels of the tuner, not only the shard's
        if geo is None:
            raise ValueError("run_all needs one demodulator class and geometry for all channels")
        kind, B, A, tau = geo
        ch = 2 if kind == hip.RCFM_WBFM else 1
        audio = hip.empty((count, A, ch), self._torch.float32)
        hip.check(self._lib.rcfm_pipeline_run(handle, self._batched_demod(kind, B, A, tau, chunk), first, count,
                                              hip.ptr(audio), hip.stream()))
        return audio

    def _plan_uniform(self):
        """(kind, B, A, tau) when every channel of the tuner carries the same demodulator class and geometry,
        else None; cached per channel-list version."""
        if self._uniform[0] != self._version:
            geo = {self._geometry(c.demodulator) for c in self._bounds}
            self._uniform = (self._version, next(iter(geo)) if len(geo) == 1 and None not in geo else None)
        return self._uniform[1]

    def run_each(self, numpy_output: bool = True):
        """What the reference's loop collects (examples/multi_fm_server.py:100-106,
        ``[ch.demodulator.run(tuner.run(ch.index)) for ch in tuner.channels()]``): a list with one audio array per
        channel, shaped as the reference's demodulators return it: [A, 1] for FM / MFM, [1, A, 2] for WBFM (wbfm.py:94 dstack) -- the shapes of FM.run / MFM.run / WBFM.run here.

        The channels may differ in demodulator class, bandwidth and audio rate (broadcast stations next to
        narrow-band ones): consecutive channels of one class and geometry run as one batched sequence, and the
        de-emphasis state is the tuner's, per channel, as in ``run_all``.  After ``shard(first, count)`` the list
        covers that range only.
        """
        return self._each_result(self._run_each_device(), self._cuda and not numpy_output)

    def _run_each_device(self):
        # run_each's launches on the current stream; [(n, ch, device tensor [n, A, ch])] per group of like channels
        handle = self._ready()
        groups, first, count = self._launch_plan()
        blocks = []
        for i, n, kind, B, A, tau in groups:
            if kind is None:
                raise ValueError("run_each needs an FM, MFM or WBFM demodulator on every channel")
            ch = 2 if kind == hip.RCFM_WBFM else 1
            audio = hip.empty((n, A, ch), self._torch.float32)
            hip.check(self._lib.rcfm_pipeline_run(handle, self._batched_demod(kind, B, A, tau, 0), i, n,
                                                  hip.ptr(audio), hip.stream()))
            blocks.append((n, ch, audio))
        return blocks

    def _each_result(self, blocks, device_output):
        out = []
        for n, ch, audio in blocks:
            block = self._result(audio, device_output)
            out.extend(block[k:k + 1] if ch == 2 else block[k] for k in range(n))
        return out

    @staticmethod
    def _geometry(demod):
        kind = _KINDS.get(type(demod).__name__)
        if kind is None:
            return None
        return kind, demod._input_size, demod._output_size, demod._tau

    def reset_states(self):
        """Every channel's de-emphasis state back to the reference's freshly constructed filters (deemphasis.py:48-49):
        the batched handles of run_all / run_each, whose slots the channels' demodulator objects share."""
        for h in self._batched.values():
            hip.check(self._lib.rcfm_demod_reset_state(h.value, hip.stream()))

    def set_kernel_options(self, lds_chain=True, fused_tiles=True, phase_link=True, narrow_tiles=1):
        """Which forms of the kernel chain run_all / run_each may use (rcfm_demod_set_option; no reference
        counterpart).  The audio does not depend on them beyond float32 rounding: switching all three off gives a
        second evaluation of the same path that shares no kernel schedule with the default one, which is what the
        full-size parity tests compare against.  narrow_tiles: 0 = tile kernels with 16 lines per tile always, 1 = 8 lines
        when a launch has fewer than two tiles per CU (the default), 2 = always 8.  Takes effect with the next run_all /
        run_each."""
        self._kernel_options = (bool(lds_chain), bool(fused_tiles), bool(phase_link), int(narrow_tiles))

    def _batched_demod(self, kind, B, A, tau, chunk):
        # one handle per geometry, sized for all channels: rcfm_pipeline_run addresses the channels of tuner and
        # demodulator by the same index, so the per-channel state survives regrouping
        opts = getattr(self, "_kernel_options", (True, True, True, 1))
        key = (kind, len(self._bounds), B, A, tau, int(chunk)) + ((opts,) if opts != (True, True, True, 1) else ())
        if key not in self._batched:
            h = ctypes.c_void_p()
            with hip.bound(self._arena):
                hip.check(self._lib.rcfm_demod_create(kind, len(self._bounds), B, A, tau, int(chunk), ctypes.byref(h)))
            self._batched[key] = hip.Handle(h, self._lib.rcfm_demod_destroy)
            for opt, on in zip((hip.RCFM_OPT_LDS_CHAIN, hip.RCFM_OPT_FUSED_TILES, hip.RCFM_OPT_PHASE_LINK), opts[:3]):
                if not on:
                    hip.check(self._lib.rcfm_demod_set_option(h, opt, 0))
            if opts[3] != 1:   # (rcfm_pipeline_run hands the same setting to the tuner's inverse FFT of each chunk)
                hip.check(self._lib.rcfm_demod_set_option(h, hip.RCFM_OPT_NARROW_TILES, opts[3]))
            self._bind_states(key, self._batched[key])
            if self._state_fence and kind != hip.RCFM_FM:    # after the binding: the fence travels with the state buffer
                hip.check(self._lib.rcfm_demod_set_option(h, hip.RCFM_OPT_STATE_FENCE, 1))
        elif self._bound_version.get(key) != self._version:
            self._bind_states(key, self._batched[key])
        return self._batched[key].value

    def _bind_states(self, key, handle):
        """ONE de-emphasis state per channel, as in the reference, where it lives in the demodulator object the
        Channel carries (deemphasis.py:48-49,64; wbfm.py:53-57): every channel's demodulator instance of this
        geometry keeps its state in its channel's slot of the batched handle from now on (rcfm_demod_bind_state moves
        what it has carried so far), and further batched handles of the same geometry (another `chunk`) share the
        first one's buffer.  Mixing ``ch.demodulator.run(tuner.run(i))`` and ``run_all()`` across buffers then gives
        what the reference's loop gives.  O(C) Python, once per change of the channel list; a demodulator whose own
        librcfm handle does not exist yet (it is created on first use) binds when it does; FM carries no state."""
        kind, C, B, A, tau = key[:5]
        self._bound_version[key] = self._version
        if kind == hip.RCFM_FM:
            return
        owner_key = (kind, C, B, A, tau)
        owner = self._state_owner.get(owner_key)
        if owner is None:
            self._state_owner[owner_key] = handle
        elif owner.value != handle.value:
            hip.check(self._lib.rcfm_demod_bind_state(handle.value, owner.value, 0, 0, hip.stream()))   # the owner has the history
        if getattr(self, "_is_lane", False):
            return                            # the channels' demodulator objects stay bound to the base tuner's handle
        geo = (kind, B, A, tau)
        seen = set()
        for c in self._bounds:
            d = c.demodulator
            if id(d) in seen or self._geometry(d) != geo or getattr(d, "_batch", 0) != 1:
                continue                      # (an object registered on two channels keeps its first slot)
            seen.add(id(d))
            d._bind(handle, c.index)

    # ---- consecutive buffers on several streams (radiocore.tools.Lanes) ---------------------------------------------

    def _arm_state_fence(self):
        """From now on this tuner's batched demodulator handles may be used from several streams: every launch sequence
        that touches the shared de-emphasis state waits for the previous one (rcfm_demod_set_option, RCFM_OPT_STATE_FENCE)."""
        self._state_fence = True
        for key, h in self._batched.items():
            if key[0] != hip.RCFM_FM:
                hip.check(self._lib.rcfm_demod_set_option(h.value, hip.RCFM_OPT_STATE_FENCE, 1))

    def _lane_clone(self):
        """A second Tuner over the SAME channel list, demodulator objects and de-emphasis state, with its own device
        handles (spectrum, workspaces): what one more stream needs to work on the next buffer."""
        t = Tuner(cuda=self._cuda)
        t._arena = self._arena
        t._is_lane = True
        t._sync_lane(self)
        return t

    def _sync_lane(self, base):
        if self._version == base._version and self._bounds is base._bounds and self._shard == base._shard and \
                self._input_bandwidth == base._input_bandwidth:
            return
        self._bounds = base._bounds
        self._input_frequency, self._input_bandwidth = base._input_frequency, base._input_bandwidth
        self._lo, self._hi, self._bw_sum = base._lo, base._hi, base._bw_sum
        self._version = base._version
        self._win_size = base._win_size
        self._state_owner = base._state_owner          # one state per channel, whoever runs it
        self._state_fence = True
        if hasattr(base, "_kernel_options"):
            self._kernel_options = base._kernel_options
        if base._shard is not None and self._shard != base._shard:
            self.shard(*base._shard)
