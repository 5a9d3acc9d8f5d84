"""CPU: the host-side bookkeeping of radiocore.tools.Tuner (reference: radiocore/tools/tuner.py:52-174, caller
examples/multi_fm_server.py:98-106) with the ABI replaced by a counting stand-in.

What is checked here needs no device: (i) the O(1)-per-channel geometry update gives exactly the floats of the
reference's recomputation over all channels (the oracle restates that recomputation); (ii) the steady state of
load + run_all / run_each does no per-channel Python work -- 8192 channels cost the same few microseconds as 8 --
and rebuilds nothing until add_channel / reset / shard change the channel list.  No compute call is made."""

import ctypes
import time

import numpy as np
import pytest

import radiocore_oracle as oracle


class _FakeTensor:
    is_cuda = True

    def __init__(self, shape):
        self.shape = tuple(shape)

    def __getitem__(self, k):
        return self

    def record_stream(self, stream):
        pass


class _FakeLib:
    """Every entry point returns 0 and counts its calls; *_create hands out a non-null handle."""

    def __init__(self):
        self.calls = {}

    def __getattr__(self, name):
        def fn(*args):
            self.calls[name] = self.calls.get(name, 0) + 1
            if name.endswith("_create"):
                out = args[-1]
                out._obj.value = 0x1000 + sum(self.calls.values())
            return 0
        return fn


class _FakeTorch:
    complex64 = "c64"
    float32 = "f32"
    Tensor = _FakeTensor


@pytest.fixture()
def fake_backend(monkeypatch):
    from radiocore._internal import hip
    lib = _FakeLib()
    monkeypatch.setattr(hip, "lib", lambda: lib)
    monkeypatch.setattr(hip, "torch", lambda: _FakeTorch)
    monkeypatch.setattr(hip, "empty", lambda shape, dtype: _FakeTensor(shape))
    monkeypatch.setattr(hip, "ptr", lambda t: ctypes.c_void_p(0))
    monkeypatch.setattr(hip, "stream", lambda: ctypes.c_void_p(0))
    monkeypatch.setattr(hip, "to_device", lambda x, dtype=None: x)
    monkeypatch.setattr(hip, "to_host", lambda x: x)
    return lib


class FM:                      # the Tuner recognises demodulators by class name and geometry attributes
    def __init__(self, B, A):
        self._input_size, self._output_size, self._tau = B, A, 75e-6


class WBFM(FM):
    pass


def _tuner(C, B=12500, A=8000, raster=12000, cls=FM):
    from radiocore.tools.tuner import Tuner
    t = Tuner(cuda=True)
    for i in range(C):
        t.add_channel(float(int(100e6 + (i - (C - 1) / 2.0) * raster)), B, cls(B, A))
    return t


def test_running_geometry_equals_the_recomputation(fake_backend):
    """tuner.py:163-174 recomputed over all channels (the oracle) against the running min / max / sum, on an
    untidy list: unequal float bandwidths, unsorted centres, a request_bandwidth in the middle."""
    from radiocore.tools.tuner import Tuner
    rng = np.random.default_rng(3)
    t, ref = Tuner(), oracle.Tuner()
    for i in range(300):
        f = float(rng.integers(88_000_000, 108_000_000)) + float(rng.choice([0.0, 0.5, 0.25]))
        bw = float(rng.choice([12500, 200000, 240000, 180000.5, 25000]))
        t.add_channel(f, bw, None)
        ref.add_channel(f, bw, None)
        assert t.input_frequency == ref.input_frequency and t.input_bandwidth == ref.input_bandwidth, i
        if i == 100:
            t.request_bandwidth(t.input_bandwidth + 1e6)
            ref.request_bandwidth(ref.input_bandwidth + 1e6)
    assert [c.address_bytes for c in t.channels()[:5]] == [int(c.center_frequency).to_bytes(4, "little")
                                                            for c in ref.channels()[:5]]
    with pytest.raises(ValueError):
        t.reset()                                             # min() of nothing, like the reference
    t.add_channel(100e6, 200e3, None)                         # and the tuner is usable again afterwards
    ref2 = oracle.Tuner()
    ref2.add_channel(100e6, 200e3, None)
    assert (t.input_frequency, t.input_bandwidth) == (ref2.input_frequency, ref2.input_bandwidth)


def test_steady_state_does_no_per_channel_work(fake_backend):
    """8192 channels: after the first buffer, load + run_all is a handful of attribute reads and two ABI calls.
    Budget 0.2 ms per buffer (cfg5's whole GPU step is 2 ms); measured ~0.02 ms here.  The handle and the launch
    plan are rebuilt exactly once per change of the channel list."""
    lib = fake_backend
    C = 8192
    t = _tuner(C)
    N = 100_000_000
    t.request_bandwidth(float(N))                             # cfg5: 98.3 MHz of channels in a 100 MSPS buffer
    x = _FakeTensor((N,))
    t.load(x)
    t.run_all(numpy_output=False)
    assert lib.calls["rcfm_tuner_create"] == 1 and lib.calls["rcfm_demod_create"] == 1
    reps = 200
    t0 = time.perf_counter()
    for _ in range(reps):
        t.load(x)
        t.run_all(numpy_output=False)
    per_buffer = (time.perf_counter() - t0) / reps
    assert per_buffer < 0.2e-3, per_buffer
    # the same budget for the reference-shaped list of run_each: one slice per channel is the only O(C) part, so
    # it is measured apart from the bookkeeping (plan reuse is what is asserted)
    t.run_each(numpy_output=False)
    plan = t._plan
    t.run_each(numpy_output=False)
    assert t._plan is plan
    assert lib.calls["rcfm_tuner_create"] == 1 and lib.calls["rcfm_demod_create"] == 1
    assert lib.calls["rcfm_tuner_load"] == reps + 1 and lib.calls["rcfm_pipeline_run"] == reps + 3
    # a new channel invalidates handle and plan once
    t.add_channel(t.channels()[-1].center_frequency + 12000, 12500, FM(12500, 8000))
    t.request_bandwidth(float(N))
    t.load(x)
    t.run_all(numpy_output=False)
    t.load(x)
    t.run_all(numpy_output=False)
    assert lib.calls["rcfm_tuner_create"] == 2 and lib.calls["rcfm_demod_create"] == 2


def test_launch_plan_groups_and_shard(fake_backend):
    from radiocore.tools.tuner import Tuner
    from radiocore._internal import hip
    t = Tuner()
    spec = [(WBFM, 240000, 48000)] * 3 + [(FM, 12500, 8000)] * 2 + [(WBFM, 240000, 48000)] + [(FM, 12500, 6000)]
    for i, (cls, B, A) in enumerate(spec):
        t.add_channel(100e6 + i * 250e3, B, cls(B, A))
    groups, first, count = t._launch_plan()
    assert (first, count) == (0, 7)
    assert [(g[0], g[1], g[2]) for g in groups] == [(0, 3, hip.RCFM_WBFM), (3, 2, hip.RCFM_FM), (5, 1, hip.RCFM_WBFM),
                                                    (6, 1, hip.RCFM_FM)]
    with pytest.raises(ValueError, match="one demodulator class and geometry"):
        t._loaded_size = int(t.input_bandwidth)
        t.run_all()
    t.shard(2, 3)
    groups, first, count = t._launch_plan()
    assert (first, count) == (2, 3) and [(g[0], g[1]) for g in groups] == [(2, 1), (3, 2)]
    with pytest.raises(IndexError):
        t.shard(5, 4)


class _FakeEvent:
    def __init__(self, enable_timing=False):
        self.where = None

    def record(self, stream):
        self.where = len(stream.log)             # position in the log: what had been queued when it was recorded
        stream.log.append(("record_event", stream.name))

    def synchronize(self):
        pass


class _FakeStream:
    def __init__(self, log, name):
        self.log, self.name = log, name

    def wait_stream(self, other):
        self.log.append(("wait_stream", self.name, other.name))

    def wait_event(self, ev):
        self.log.append(("wait_event", self.name, ev.where))

    def record_event(self):
        self.log.append(("record_event", self.name))
        return _FakeEvent()


def test_lanes_bookkeeping(fake_backend, monkeypatch):
    """radiocore.tools.Lanes without a device: lane k of `depth` gets buffer i = k (mod depth) on its own stream, every
    lane has its own tuner and demodulator handle, the extra demodulator handles are bound to the FIRST one's state
    (one de-emphasis state per channel, whoever runs it) and the fence option is set after the binding; a channel added
    later reaches every lane."""
    import contextlib
    from radiocore._internal import hip
    from radiocore.tools.lanes import Lanes

    lib = fake_backend
    log, order = [], []
    real_getattr = type(lib).__getattr__

    def recording(self, name):
        fn = real_getattr(self, name)

        def wrapped(*args):
            order.append((name,) + tuple(a if isinstance(a, int) else getattr(a, "value", None) for a in args[:4]))
            return fn(*args)
        return wrapped
    monkeypatch.setattr(type(lib), "__getattr__", recording)

    current = [_FakeStream(log, "default")]

    class _Cuda:
        @staticmethod
        def Stream():
            log.append(("new",))
            return _FakeStream(log, "lane%d" % (sum(1 for e in log if e[0] == "new") - 1))

        Event = _FakeEvent

        @staticmethod
        def current_stream():
            return current[0]

        @staticmethod
        @contextlib.contextmanager
        def stream(st):
            prev, current[0] = current[0], st
            log.append(("enter", st.name))
            try:
                yield
            finally:
                current[0] = prev

    class _Torch(_FakeTorch):
        cuda = _Cuda
    monkeypatch.setattr(hip, "torch", lambda: _Torch)

    t = _tuner(6, cls=WBFM)          # WBFM carries state (FM would need no fence)
    N = 1_000_000
    t.request_bandwidth(float(N))
    lanes = Lanes(t, depth=2)
    x = _FakeTensor((N,))
    tickets = [lanes.submit(x) for _ in range(5)]
    assert tickets == [0, 1, 2, 3, 4] and lanes.depth == 2
    assert lib.calls["rcfm_tuner_create"] == 2 and lib.calls["rcfm_demod_create"] == 2
    assert lib.calls["rcfm_tuner_load"] == 5 and lib.calls["rcfm_pipeline_run"] == 5
    assert lib.calls["rcfm_demod_bind_state"] == 1                 # the second lane's handle -> the first one's state
    names = [o[0] for o in order]
    bind_at = names.index("rcfm_demod_bind_state")
    fences = [i for i, o in enumerate(order) if o[0] == "rcfm_demod_set_option" and o[2:4] == (hip.RCFM_OPT_STATE_FENCE, 1)]
    assert len(fences) == 2 and fences[1] > bind_at                # each lane's handle, the bound one after its binding
    entered = [e[1] for e in log if e[0] == "enter"]
    assert entered[0] != entered[1] and entered == [entered[0], entered[1]] * 2 + [entered[0]]
    assert sum(1 for e in log if e[0] == "wait_stream") == 5      # every lane stream waits for the buffer's producer
    # the Feeder recipe's hold: the current stream waits for the event recorded right behind the buffer's LOAD (its
    # last reader), not for the end-of-buffer event -- that one would serialise the lanes through the next submit()
    records = [i for i, e in enumerate(log) if e[0] == "record_event"]
    assert len(records) == 2 * 5                                   # per buffer: behind the load, at the end
    lanes.hold_current_stream(tickets[-1])
    assert log[-1] == ("wait_event", "default", records[-2])
    for tk in reversed(tickets):
        assert lanes.result(tk, numpy_output=False).shape == (6, 8000, 2)
    with pytest.raises(KeyError):
        lanes.result(0)
    t.add_channel(101e6, 12500, WBFM(12500, 8000))                 # the channel list changes: both lanes rebuild
    before = lib.calls["rcfm_tuner_create"]
    for _ in range(2):
        lanes.result(lanes.submit(x), numpy_output=False)
    assert lib.calls["rcfm_tuner_create"] == before + 2
