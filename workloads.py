"""Deterministic synthetic IQ inputs for tests, golden-vector capture and bench.py.

Host-side numpy only (no reference, no oracle, no GPU).  The recipe follows
SURVEY.md section 8(d): broadcast-FM stations whose modulation is a sum of
integer-Hz tones, so every station is periodic in the 1-second buffer and its
accumulated phase stays bounded (the reference unwraps in float32, fm.py:62,
so unbounded phase would put float32 noise into the *reference*).

Stations sit exactly on channel centres; the wideband buffer is assembled in
the frequency domain so that the Tuner's brick-wall extraction returns each
station (plus its neighbours' overlap and a little noise).
"""

import numpy as np


def station_mpx(i, B, stereo=True):
    """Stereo multiplex (or mono tone set) of station i sampled at B Hz for 1 s."""
    rng = np.random.default_rng(1000 + i)
    t = np.arange(B, dtype=np.float64) / B
    k = i % 89

    def tones(freqs, amp):
        acc = np.zeros(B)
        for f in freqs:
            acc += amp * np.sin(2 * np.pi * f * t + rng.uniform(0, 2 * np.pi))
        return acc

    if not stereo:
        return tones((300 + 7 * k, 1000 + 3 * k, 2200 + k), 0.3)
    L = tones((300 + 37 * k, 1000 + 11 * k, 5000 + 3 * k), 0.3)
    R = tones((440 + 29 * k, 2500 + 7 * k), 0.35)
    return (0.3 * (L + R) + 0.1 * np.sin(2 * np.pi * 19000 * t)
            + 0.3 * (L - R) * np.sin(2 * np.pi * 38000 * t))


def station_iq(i, B, deviation=None, stereo=True, noise=0.0, offset=0):
    """Complex baseband FM signal of station i, complex128 [B].

    deviation defaults to 75 kHz scaled by B / 240 kHz so that reduced-size test
    channels keep the per-sample phase step well inside (-pi, pi).
    offset (integer Hz): carrier offset from the channel centre -- the accumulated phase then grows to
    2 pi offset over the buffer, which is where the reference's float32 unwrap (fm.py:62) loses precision.
    """
    if deviation is None:
        deviation = 75e3 * B / 240000.0
    mpx = station_mpx(i, B, stereo)
    s = np.exp(2j * np.pi * (deviation * np.cumsum(mpx) + int(offset) * np.arange(B, dtype=np.float64)) / B)
    if noise:
        rng = np.random.default_rng(5000 + i)
        s = s + noise * (rng.standard_normal(B) + 1j * rng.standard_normal(B))
    return s


def single_channel(B, i=0, deviation=None, stereo=True, noise=0.01, offset=0):
    """One station at baseband as complex64 [B] (configs 1 and 2)."""
    return station_iq(i, B, deviation, stereo, noise, offset).astype(np.complex64)


def channel_grid(C, raster, f0=100e6):
    """Integer-Hz centre frequencies f0 + (i - (C-1)/2) * raster."""
    return [float(int(f0 + (i - (C - 1) / 2.0) * raster)) for i in range(C)]


def wideband(N, f_in, centres, B, deviation=None, stereo=True, noise=0.003,
             gain=None):
    """Wideband complex64 [N] buffer holding one station per centre frequency.

    Station i's B-point spectrum is added to the N-point spectrum at offset
    int(f_c - f_in) bins, scaled N/B so its time-domain amplitude is `gain`
    (default 1/sqrt(len(centres)) keeps the sum O(1)).
    """
    C = len(centres)
    if gain is None:
        gain = 1.0 / np.sqrt(C)
    Xw = np.zeros(N, np.complex128)
    kk = np.fft.fftfreq(B, 1.0 / B).astype(np.int64)      # signed bin numbers
    for i, fc in enumerate(centres):
        S = np.fft.fft(station_iq(i, B, deviation, stereo)) * (gain * N / B)
        off = int(fc - f_in)
        np.add.at(Xw, (kk + off) % N, S)
    x = np.fft.ifft(Xw)
    rng = np.random.default_rng(7)
    x += noise * (rng.standard_normal(N) + 1j * rng.standard_normal(N))
    return x.astype(np.complex64)
