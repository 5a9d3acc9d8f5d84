#!/usr/bin/env python3
"""The reference's single-station receiver (examples/receive_fm.py) on this package, without SDR or sound card.

Same threads and hand-overs: a producer fills a `RingBuffer` with SDR-sized chunks (receive_fm.py:44-57, 2**16 samples
per read), the DSP thread takes one-second buffers out of it (:95-98), brings the 10 MSPS capture down to the station
bandwidth with the complex `Decimate` (:100, 10 000 000 -> 250 000) and demodulates (:101, `WBFM(250e3, 48e3)`: the
reference's own default geometry, :18-19), and the audio callback's queue (:103) is a list here -- or a WAV file.

    python examples/receive_fm_offline.py [--seconds 3] [--demodulator WBFM|MFM|FM] [--wav out.wav]
"""
import argparse
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radio-core_amd")]

import numpy as np  # noqa: E402

import radiocore  # noqa: E402
import workloads  # noqa: E402
from radiocore import Buffer, Decimate, RingBuffer  # noqa: E402


def capture(input_rate, demod_rate, station=7, stereo=True, second=0):
    """One second of a synthetic 'SDR capture': the station at the centre of `input_rate`, a weaker neighbour 400 kHz up
    and a little noise -- what receive_fm.py's SoapySDR stream would deliver.  Built in the frequency domain."""
    n, b = int(input_rate), int(demod_rate)
    X = np.zeros(n, np.complex128)
    kk = np.fft.fftfreq(b, 1.0 / b).astype(np.int64)
    for i, (off, gain) in enumerate(((0, 0.5), (400_000, 0.2))):
        s = workloads.station_iq(station + i, b, deviation=75e3 * b / 240000.0, stereo=stereo)
        np.add.at(X, (kk + off) % n, np.fft.fft(np.roll(s, 1000 * second)) * (gain * n / b))
    rng = np.random.default_rng(100 + second)
    return (np.fft.ifft(X) + 0.002 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)


def run(seconds=3, demodulator="WBFM", input_rate=10_000_000, demod_rate=250_000, audio_rate=48_000, deemphasis=75e-6,
        cuda=True):
    """Returns the list of audio blocks the DSP thread queued, one per second, shaped as the reference's demodulators
    return them ((1, A, 2) for WBFM, (A, 1) for MFM / FM)."""
    demod = getattr(radiocore, demodulator)(demod_rate, audio_rate, deemphasis=deemphasis, cuda=cuda)
    decim = Decimate(input_rate, demod_rate, cuda=cuda)
    ring = RingBuffer(input_rate * 3, dtype=np.complex64, cuda=cuda, print_overflow=False)
    done = threading.Event()

    def sdr():                                             # the readStream loop: 2**16 samples per read
        for s in range(seconds):
            x = capture(input_rate, demod_rate, stereo=(demodulator == "WBFM"), second=s)
            for k in range(0, len(x), 1 << 16):
                piece = x[k:k + (1 << 16)]
                while ring.vacancy < len(piece):
                    time.sleep(0.001)
                ring.put(piece)
        done.set()

    audio = []
    t = threading.Thread(target=sdr)
    t.start()
    tmp = Buffer(input_rate, dtype=np.complex64, cuda=cuda)
    while len(audio) < seconds:
        if not ring.get(tmp.data):
            if done.is_set() and ring.occupancy < tmp.size:
                break
            continue
        audio.append(np.asarray(demod.run(decim.run(tmp.data))))
    t.join()
    return audio


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=int, default=3)
    ap.add_argument("--demodulator", default="WBFM", choices=["WBFM", "MFM", "FM"])
    ap.add_argument("--wav", default=None)
    args = ap.parse_args()
    t0 = time.time()
    blocks = run(args.seconds, args.demodulator)
    pcm = np.concatenate([b.reshape(-1, b.shape[-1]) if b.ndim == 3 else b for b in blocks])
    print("%d s of %s audio, %d channel(s), peak %.3f, %.2f s wall" % (len(blocks), args.demodulator, pcm.shape[1],
                                                                      float(np.abs(pcm).max()), time.time() - t0))
    if args.wav:
        import wave
        with wave.open(args.wav, "wb") as w:
            w.setnchannels(pcm.shape[1])
            w.setsampwidth(2)
            w.setframerate(48000)
            w.writeframes((np.clip(pcm, -1, 1) * 32767).astype("<i2").tobytes())
        print("wrote", args.wav)


if __name__ == "__main__":
    main()
