"""Lanes (radiocore.tools.lanes): consecutive buffers on alternating streams give, bit for bit, what the
one-buffer-at-a-time loop gives -- the de-emphasis state is the only link between buffers (deemphasis.py:64) and
RCFM_OPT_STATE_FENCE orders it across streams."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radio-core_amd")]

pytestmark = pytest.mark.gpu


def _band(rc, kind, C, B, A, n, seed):
    rng = np.random.default_rng(seed)
    tuner = rc.Tuner(cuda=True)
    f0 = 100e6
    for c in range(C):
        tuner.add_channel(f0 + c * B * 0.9, B, getattr(rc, kind)(B, A, cuda=True))
    tuner.request_bandwidth(float(n))
    bufs = []
    t = np.arange(n) / n
    for b in range(6):
        x = np.zeros(n, np.complex128)
        for c, chn in enumerate(tuner.channels()):
            fc = chn.center_frequency - tuner.input_frequency
            msg = 0.4 * np.sin(2 * np.pi * (300 + 117 * c + 31 * b) * t) + 0.2 * np.sin(2 * np.pi * (1900 + 53 * c) * t)
            ph = 2 * np.pi * (fc * t + (0.15 * B / (2 * np.pi)) * np.cumsum(msg) / n)
            x += np.exp(1j * ph)
        x += 0.01 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
        bufs.append(x.astype(np.complex64))
    return tuner, bufs


@pytest.mark.parametrize("kind,C,B,A,n", [
    ("WBFM", 5, 60000, 12000, 600000),       # fused chain, run_deemph with two legs
    ("MFM", 6, 60000, 12000, 600000),        # pair chain + de-emphasis launches
    ("MFM", 7, 12500, 8000, 250000),         # LDS chain with the de-emphasis on chip
    ("FM", 5, 12500, 8000, 250000),          # no state at all
])
@pytest.mark.parametrize("depth", [2, 3])
def test_lanes_equal_the_sequential_loop(kind, C, B, A, n, depth):
    import radiocore as rc
    from radiocore.tools import Lanes

    ref_tuner, bufs = _band(rc, kind, C, B, A, n, seed=5)
    want = []
    for x in bufs:
        ref_tuner.load(x)
        want.append(ref_tuner.run_all())
    tuner, _ = _band(rc, kind, C, B, A, n, seed=5)
    lanes = Lanes(tuner, depth=depth)
    tickets = [lanes.submit(x) for x in bufs]           # all six in flight before the first result is read
    got = [lanes.result(t) for t in reversed(tickets)][::-1]
    for i, (g, w) in enumerate(zip(got, want)):
        assert g.shape == w.shape
        assert np.array_equal(g, w), "buffer %d differs: max %g" % (i, np.abs(g - w).max())
    if kind != "FM":      # the state really links the buffers: a fresh tuner on the last buffer gives other audio
        fresh, _ = _band(rc, kind, C, B, A, n, seed=5)
        fresh.load(bufs[-1])
        assert not np.array_equal(fresh.run_all(), want[-1])


def test_lanes_and_the_per_channel_loop_share_one_state():
    """Lane buffers, then the reference's own loop on the base tuner (ch.demodulator.run(tuner.run(i))), then lanes again."""
    import radiocore as rc
    from radiocore.tools import Lanes

    ref_tuner, bufs = _band(rc, "MFM", 4, 60000, 12000, 480000, seed=9)
    want = []
    for x in bufs[:5]:
        ref_tuner.load(x)
        want.append(ref_tuner.run_all())
    tuner, _ = _band(rc, "MFM", 4, 60000, 12000, 480000, seed=9)
    lanes = Lanes(tuner, depth=2)
    got = [lanes.result(t) for t in [lanes.submit(bufs[0]), lanes.submit(bufs[1])]]
    tuner.load(bufs[2])
    loop = np.stack([np.asarray(ch.demodulator.run(tuner.run(ch.index))) for ch in tuner.channels()])
    got.append(loop.reshape(want[2].shape))
    got += [lanes.result(t) for t in [lanes.submit(bufs[3]), lanes.submit(bufs[4])]]
    for i, (g, w) in enumerate(zip(got, want)):
        assert np.abs(g - w).max() <= 2e-6 * max(1.0, np.abs(w).max()), "buffer %d" % i


def test_result_as_device_tensor_and_drain():
    import radiocore as rc
    import torch
    from radiocore.tools import Lanes

    tuner, bufs = _band(rc, "WBFM", 3, 60000, 12000, 360000, seed=2)
    lanes = Lanes(tuner, depth=2)
    t0, t1 = lanes.submit(torch.from_numpy(bufs[0]).cuda()), lanes.submit(bufs[1])
    lanes.drain()
    a1 = lanes.result(t1, numpy_output=False)
    a0 = lanes.result(t0, numpy_output=False)
    assert a0.is_cuda and a1.is_cuda and a0.shape == (3, 12000, 2)
    assert torch.isfinite(a0).all() and torch.isfinite(a1).all() and not torch.equal(a0, a1)
    with pytest.raises(KeyError):
        lanes.result(t0)


def test_lanes_with_mixed_channel_sets():
    """submit(each=True): broadcast stations next to narrow-band ones (Tuner.run_each), three geometries, two lanes."""
    import radiocore as rc
    from radiocore.tools import Lanes

    def band():
        tuner = rc.Tuner(cuda=True)
        spec = [("WBFM", 60000, 12000), ("WBFM", 60000, 12000), ("MFM", 12500, 8000), ("MFM", 12500, 8000),
                ("MFM", 12500, 8000), ("FM", 25000, 5000)]
        f = 100e6
        for kind, B, A in spec:
            tuner.add_channel(f, B, getattr(rc, kind)(B, A, cuda=True))
            f += 70000
        tuner.request_bandwidth(600000.0)
        return tuner

    rng = np.random.default_rng(3)
    n = 600000
    t = np.arange(n) / n
    bufs = []
    tuner = band()
    for b in range(4):
        x = np.zeros(n, np.complex128)
        for c, chn in enumerate(tuner.channels()):
            fc = chn.center_frequency - tuner.input_frequency
            msg = 0.5 * np.sin(2 * np.pi * (250 + 90 * c + 40 * b) * t)
            x += np.exp(1j * 2 * np.pi * (fc * t + 0.1 * chn.bandwidth / (2 * np.pi) * np.cumsum(msg) / n))
        bufs.append((x + 0.01 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64))
    want = []
    for x in bufs:
        tuner.load(x)
        want.append(tuner.run_each())
    lanes = Lanes(band(), depth=2)
    tickets = [lanes.submit(x, each=True) for x in bufs]
    for i, tk in enumerate(tickets):
        got = lanes.result(tk)
        assert len(got) == len(want[i]) == 6
        for g, w in zip(got, want[i]):
            assert g.shape == w.shape and np.array_equal(g, w), i


def test_feeder_recipe_keeps_the_lanes_overlapped():
    """The documented host-fed recipe -- ``with feeder.next() as x: t = lanes.submit(x); lanes.hold_current_stream(t)`` --
    must not serialise the lanes: the current stream (and with it the next submit) waits for the wideband FFT of the
    buffer, the last reader of the feeder's slot, not for the buffer's audio.  Asserted on the device's own clock
    (timing events): lane i + 1 starts working BEFORE lane i has finished, and the audio is still bit-identical.
    The band is 256 overlapping 240 kHz channels over a 2.4 MHz buffer, so that a buffer's kernels (about 1.5 ms) outlast
    its 19 MB copy over PCIe (0.4 ms): with fewer channels per input sample the copy is what every lane waits for."""
    import radiocore as rc
    from radiocore.tools import Feeder, Lanes

    C, B, A, n = 256, 240000, 48000, 2_400_000

    def band():
        tuner = rc.Tuner(cuda=True)
        for c in range(C):
            tuner.add_channel(100e6 + c * 8000.0, B, rc.WBFM(B, A, cuda=True))
        tuner.request_bandwidth(float(n))
        return tuner

    rng = np.random.default_rng(9)
    bufs = [(rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64) for _ in range(6)]
    ref_tuner = band()
    want = []
    for x in bufs:
        ref_tuner.load(x)
        want.append(ref_tuner.run_all())
    del ref_tuner
    tuner = band()
    lanes = Lanes(tuner, depth=2, timing=True)
    feeder = Feeder(n, depth=2)
    tickets = []
    feeder.submit(bufs[0])
    for i in range(len(bufs)):
        if i + 1 < len(bufs):
            feeder.submit(bufs[i + 1])
        with feeder.next() as x:
            tickets.append(lanes.submit(x))
            lanes.hold_current_stream(tickets[-1])
    got = [lanes.result(t) for t in tickets]
    feeder.close()
    for i, (g, w) in enumerate(zip(got, want)):
        assert np.array_equal(g, w), i
    overlaps = [lanes.overlap_ms(tickets[i], tickets[i + 1]) for i in range(len(tickets) - 1)]
    # every buffer but the first (whose copy nothing hides) starts while its predecessor is still running
    assert sum(o > 0 for o in overlaps[1:]) >= len(overlaps) - 2, overlaps
