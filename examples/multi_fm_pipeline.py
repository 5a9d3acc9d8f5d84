#!/usr/bin/env python3
"""The reference's multi-channel server (examples/multi_fm_server.py) on this package, without SDR or sockets.

Same two threads and the same hand-overs: a producer fills a `RingBuffer` in chunks (the reference's SoapySDR
`readStream` callback, multi_fm_server.py:60-66), the DSP thread takes one-second buffers out of it
(`data_in.get`, :95), loads the `Tuner` and demodulates every channel (:98-102), and cuts the result into the wire
messages a subscriber expects (:103-106).  What differs is the device side: the buffer crosses PCIe through a
`Feeder` (the copy of second i+1 runs under the kernels of second i) and all channels run in one `run_all()`.

    python examples/multi_fm_pipeline.py [--seconds 3] [--channels 6] [--rate 1200000]
"""
import argparse
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radio-core_amd")]

import numpy as np  # noqa: E402

import workloads  # noqa: E402
from radiocore import WBFM, Buffer, Feeder, RingBuffer, Tuner  # noqa: E402
from radiocore.tools import Lanes  # noqa: E402
from radiocore.tools import wire  # noqa: E402


def run(seconds=3, channels=6, rate=1_200_000, bandwidth=60_000, audio_rate=12_000, publish=None, lanes=1):
    """Returns [(frequency, float32 [A, 2])] per second and channel, in publish order.  lanes > 1: that many seconds in
    flight on alternating streams (radiocore.tools.Lanes); a second's messages then leave one submission later."""
    centres = workloads.channel_grid(channels, 50_000)
    tuner = Tuner(cuda=True)
    for f in centres:
        tuner.add_channel(f, bandwidth, WBFM(bandwidth, audio_rate, cuda=True))
    tuner.request_bandwidth(float(rate))
    second = workloads.wideband(rate, tuner.input_frequency, centres, bandwidth, gain=0.3)

    ring = RingBuffer(rate * 2, dtype=np.complex64, cuda=True, print_overflow=False)
    done = threading.Event()

    def producer():                                   # the SDR callback: chunks of 1/8 s, one second per second ...
        chunk = rate // 8
        for s in range(seconds):
            for k in range(8):
                piece = np.roll(second, 1000 * s)[k * chunk:(k + 1) * chunk]
                while ring.vacancy < len(piece):      # (... as fast as the consumer lets it, for the example)
                    time.sleep(0.001)
                ring.put(piece)
        done.set()

    feeder = Feeder(rate, dtype=np.complex64, depth=2)
    staging = [Buffer(rate, dtype=np.complex64, cuda=True) for _ in range(feeder.depth + 1)]
    out = []
    t = threading.Thread(target=producer)
    t.start()
    pending = 0                                       # seconds submitted to the feeder, not yet processed
    got = 0
    pipe = Lanes(tuner, depth=lanes) if lanes > 1 else None
    tickets = []

    def publish_all(audio):
        for message in wire.frames(tuner.channels(), audio):
            if publish is not None:
                publish(message)                      # socket.send_multipart(message) in the reference
            out.append(wire.parse_frame(message, 2))

    while got < seconds:
        while pending < feeder.depth and got + pending < seconds:
            buf = staging[(got + pending) % len(staging)]
            if not ring.get(buf.data, timeout=5.0):   # one second of samples (multi_fm_server.py:95)
                raise RuntimeError("producer stalled")
            feeder.submit(buf.data)                   # its H2D copy starts now, on the copy stream
            pending += 1
        with feeder.next() as x:                      # orders the DSP stream behind this second's copy only
            if pipe is None:
                tuner.load(x)
                audio = tuner.run_all()               # [C, A, 2]: every channel's run -> demodulator.run
            else:
                tickets.append(pipe.submit(x))        # queued on the next lane's stream; returns at once
                pipe.hold_current_stream(tickets[-1])   # the feeder frees this slot behind the current stream
        pending -= 1
        got += 1
        if pipe is None:
            publish_all(audio)
        elif len(tickets) >= lanes:                   # `lanes` seconds in flight: collect the oldest
            publish_all(pipe.result(tickets.pop(0)))
    for tk in tickets:
        publish_all(pipe.result(tk))
    t.join()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=int, default=3)
    ap.add_argument("--channels", type=int, default=6)
    ap.add_argument("--rate", type=int, default=1_200_000)
    ap.add_argument("--lanes", type=int, default=1)
    a = ap.parse_args()
    t0 = time.perf_counter()
    msgs = run(a.seconds, a.channels, a.rate, lanes=a.lanes)
    dt = time.perf_counter() - t0
    print("%d messages (%d s x %d channels), %.1f MB of audio, %.2f s wall" %
          (len(msgs), a.seconds, a.channels, sum(m[1].nbytes for m in msgs) / 1e6, dt))
