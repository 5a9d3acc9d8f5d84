"""CPU: the host staging classes either side of the DSP path.

Behaviour follows the reference's classes (radiocore/tools/{buffer,ringbuffer,carrousel,
chopper}.py) and what its own unit tests exercise; implementation and tests are this repo's.
"""

import threading
import time

import numpy as np
import pytest

from radiocore.tools.buffer import Buffer
from radiocore.tools.carrousel import Carrousel
from radiocore.tools.chopper import Chopper
from radiocore.tools.ringbuffer import RingBuffer
from radiocore.tools import wire


def test_buffer_consume_is_a_view():
    b = Buffer(8, dtype="float")
    assert len(b) == 8 and b.size == 8 and not b.is_cuda and b.dtype == np.float64
    with b.consume() as arr:
        assert np.array_equal(arr, np.zeros(8))
        arr[:2] = 1
    with b.consume() as arr:
        arr[2:4] = 2
    assert np.array_equal(b.data, [1, 1, 2, 2, 0, 0, 0, 0])
    with pytest.raises(ValueError, match="locking is not enabled"):
        b.is_locked


def test_buffer_lock():
    b = Buffer(4, dtype=np.complex64, lock=True)
    assert b.dtype == np.complex64 and not b.is_locked
    with b.consume():
        assert b.is_locked
    assert not b.is_locked


def test_ringbuffer_fill_drain_wrap_overflow(capsys):
    r = RingBuffer(8, dtype=np.float32)
    assert (r.occupancy, r.capacity, r.vacancy) == (0, 8, 8)
    r.put([1, 2, 3, 4])
    assert (r.occupancy, r.vacancy) == (4, 4)
    assert np.array_equal(r.data, [1, 2, 3, 4, 0, 0, 0, 0])
    r.put([5, 6, 7, 8])
    assert (r.occupancy, r.vacancy) == (8, 0)
    out = np.zeros(4, np.float32)
    assert r.get(out) is True
    assert np.array_equal(out, [1, 2, 3, 4]) and r.occupancy == 4
    r.put([9, 9, 9, 9])                                   # wraps to the front
    assert np.array_equal(r.data, [9, 9, 9, 9, 5, 6, 7, 8]) and r.occupancy == 8
    r.put([1, 1, 1, 1])                                   # no room: the ring restarts
    assert "overflow" in capsys.readouterr().out
    assert r.occupancy == 4 and np.array_equal(r.data[:4], [1, 1, 1, 1])
    six = np.zeros(6, np.float32)
    r.put([2, 3])
    assert r.get(six) is True and np.array_equal(six, [1, 1, 1, 1, 2, 3])
    r.put(np.arange(7))                                   # split write across the end
    seven = np.zeros(7, np.float32)
    assert r.get(seven) is True and np.array_equal(seven, np.arange(7))


def test_ringbuffer_errors_and_timeout():
    r = RingBuffer(4, dtype=np.float32, allow_overflow=False)
    with pytest.raises(ValueError, match="bigger than ring capacity"):
        r.put(np.zeros(5))
    r.put([1, 2, 3])
    with pytest.raises(ValueError, match="Overflow happened"):
        r.put([4, 5])
    t0 = time.time()
    assert r.get(np.zeros(4, np.float32), timeout=0.05) is None      # only 3 available
    assert time.time() - t0 < 1.0
    with pytest.raises(ValueError):
        r.get(np.zeros(9, np.float32))


def test_ringbuffer_producer_consumer_threads():
    r = RingBuffer(1 << 12, dtype=np.int64, print_overflow=False)
    total, chunk = 1 << 16, 256

    def produce():
        for s in range(0, total, chunk):
            while r.vacancy < chunk:
                time.sleep(0)
            r.put(np.arange(s, s + chunk))

    th = threading.Thread(target=produce)
    th.start()
    got = []
    buf = np.zeros(512, np.int64)
    while len(got) * 512 < total:
        assert r.get(buf, timeout=5.0) is True
        got.append(buf.copy())
    th.join()
    assert np.array_equal(np.concatenate(got), np.arange(total))


def test_carrousel_cycle_and_overflow(capsys):
    c = Carrousel([[0], [0], [0]])
    assert c.is_empty and not c.is_full and c.capacity == 3 and not c.is_healthy
    for v in (1, 2, 3):
        with c.enqueue() as item:
            item[0] = v
    assert c.is_full and c.occupancy == 3 and c.overflow == 0
    with c.enqueue() as item:                              # drops the oldest (1)
        item[0] = 4
    assert "overflow" in capsys.readouterr().out
    assert c.overflow == 1 and c.occupancy == 3
    seen = []
    while not c.is_empty:
        with c.dequeue() as item:
            seen.append(item[0])
    assert seen == [2, 3, 4]
    with pytest.raises(ValueError, match="carrousel is empty"):
        with c.dequeue():
            pass
    c.reset()
    assert c.occupancy == 0


def test_carrousel_lends_buffers_through_consume():
    c = Carrousel([Buffer(4, dtype="float32", lock=True) for _ in range(2)], print_overflow=False)
    with c.enqueue() as arr:
        assert isinstance(arr, np.ndarray)
        arr[:] = 7
    with c.dequeue() as arr:
        assert np.array_equal(arr, [7, 7, 7, 7])


def test_chopper():
    ch = Chopper(12, 4)
    x = np.arange(12)
    parts = list(ch.chop(x))
    assert [p.tolist() for p in parts] == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11]]
    parts[0][0] = 99
    assert x[0] == 99                                        # views, not copies
    assert (ch.size, ch.chunk_size) == (12, 4)
    with pytest.raises(ValueError, match="cannot evenly divide"):
        Chopper(10, 4)


def test_wire_frames_round_trip():
    from radiocore.tools.tuner import Channel
    chans = [Channel(i, 240e3, None, f - 120e3, f, f + 120e3) for i, f in enumerate((96.9e6, 97.5e6))]
    audio = np.random.default_rng(0).standard_normal((2, 100, 2)).astype(np.float32)
    msgs = wire.frames(chans, audio)
    assert msgs[0][0] == (96900000).to_bytes(4, "little") and len(msgs[0][1]) == 100 * 2 * 4
    for i, m in enumerate(msgs):
        f, pcm = wire.parse_frame(m, 2)
        assert f == int(chans[i].center_frequency) and np.array_equal(pcm, audio[i])
    # WBFM's (1, A, 2) array and a [A, 2] slice of the batched block share their byte layout
    assert audio[0][None].tobytes() == msgs[0][1]
    with pytest.raises(ValueError):
        wire.frames(chans, audio[:1])
