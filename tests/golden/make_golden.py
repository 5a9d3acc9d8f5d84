#!/usr/bin/env python3
"""Capture golden vectors by running the *reference* (luigifcruz/radio-core).

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

Imports the reference package read-only (no bytecode written into its tree),
feeds it the seeded synthetic inputs of ``workloads.py`` and stores the outputs
as ``tests/golden/*.npz``.  Inputs are not stored: tests regenerate them from
the same seeds and verify the recorded checksum.  The reference imports the
PyPI module ``atomics`` for its RingBuffer (not on the DSP path, not installed
here); an in-memory placeholder module is registered so the import succeeds.

The fixtures are data (inputs' checksums + expected outputs); no reference
source text is stored.
"""

import hashlib
import os
import sys
import types
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import scipy  # noqa: E402

import workloads  # noqa: E402


def import_reference(path="/root/reference"):
    shim = types.ModuleType("atomics")
    shim.INT = "int"
    shim.atomic = lambda width, atype: None
    sys.modules.setdefault("atomics", shim)
    sys.path.insert(0, path)
    import radiocore
    assert radiocore.__file__.startswith(path), radiocore.__file__
    return radiocore


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def save(name, **arrays):
    meta = "numpy %s scipy %s reference radio-core 1.0.0" % (np.__version__, scipy.__version__)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), _meta=np.array(meta), **arrays)
    kb = os.path.getsize(os.path.join(HERE, name + ".npz")) / 1024
    print("%-28s %8.1f KiB  %s" % (name, kb, sorted(arrays)))


def main():
    rc = import_reference()
    warnings.filterwarnings("ignore")
    rng = np.random.default_rng(42)

    # ---- primitives ------------------------------------------------------
    out = {}
    for (n, m) in [(24000, 4800), (24001, 4801), (24000, 4801), (12500, 48000),
                   (12501, 48000), (24000, 24000), (24001, 24001)]:
        x = np.random.default_rng(n + m).standard_normal(n).astype(np.float32)
        out["real_%d_%d" % (n, m)] = rc.Decimate(n, m).run(x)
        out["real_%d_%d_in" % (n, m)] = np.array(digest(x))
    for (n, m) in [(100000, 2500), (100001, 2501), (2500, 10000), (20000, 20000)]:
        g = np.random.default_rng(n + m)
        x = (g.standard_normal(n) + 1j * g.standard_normal(n)).astype(np.complex64)
        out["cplx_%d_%d" % (n, m)] = rc.Decimate(n, m).run(x)
        out["cplx_%d_%d_in" % (n, m)] = np.array(digest(x))
    save("decimate", **out)

    out = {}
    for (n, lo, hi, taps) in [(60000, 18950.0, 19050.0, 41), (100000, 18950.0, 19050.0, 41),
                              (48000, 300.0, 3000.0, 61)]:
        x = np.random.default_rng(n).standard_normal(n).astype(np.float32)
        bp = rc.Bandpass(n, lo, hi, num_taps=taps)
        out["taps_%d_%d" % (n, taps)] = bp._taps[0]
        out["y_%d_%d" % (n, taps)] = bp.run(x)
        out["in_%d_%d" % (n, taps)] = np.array(digest(x))
    save("bandpass", **out)

    out = {}
    for (n, tau) in [(48000, 75e-6), (32000, 75e-6), (8000, 50e-6), (4800, 75e-6)]:
        de = rc.Deemphasis(n, tau)
        out["taps_%d" % n] = de._taps[0]
        out["zi_%d" % n] = de._state.copy()
        g = np.random.default_rng(n)
        for k in range(2 if n != 32000 else 0):
            x = (0.5 * g.standard_normal(n)).astype(np.float32)
            out["y%d_%d" % (k, n)] = de.run(x)
            out["in%d_%d" % (k, n)] = np.array(digest(x))
        out["zf_%d" % n] = de._state.copy()
    save("deemphasis", **out)

    out = {}
    for n in (6000, 6001):
        x = np.random.default_rng(n).standard_normal(n).astype(np.float32)
        pll = rc.PLL()
        pll.step(x)
        out["z_%d" % n] = pll._baseline
        out["image2_%d" % n] = pll.image(2)
        out["real2_%d" % n] = pll.real(2)
        out["image1_%d" % n] = pll.image()
        out["in_%d" % n] = np.array(digest(x))
    save("pll", **out)

    # ---- demodulators ------------------------------------------------------
    out = {}
    for (B, A, dev, stereo) in [(24000, 4800, 5e3, False), (24001, 4801, 5e3, False),
                                (12500, 8000, 2.5e3, False), (240000, 48000, 75e3, True)]:
        x = workloads.single_channel(B, i=1, deviation=dev, stereo=stereo)
        out["fm_%d_%d" % (B, A)] = rc.FM(B, A).run(x)
        out["in_%d_%d" % (B, A)] = np.array(digest(x))
    save("fm", **out)

    out = {}
    for (B, A, dev, stereo) in [(24000, 4800, 5e3, False), (240000, 48000, 75e3, True)]:
        d = rc.MFM(B, A)
        for k in range(2):
            x = workloads.single_channel(B, i=2 + k, deviation=dev, stereo=stereo)
            out["mfm%d_%d_%d" % (k, B, A)] = d.run(x)
            out["in%d_%d_%d" % (k, B, A)] = np.array(digest(x))
    save("mfm", **out)

    out = {}
    for (B, A) in [(60000, 12000), (240000, 48000), (256000, 32000)]:
        d = rc.WBFM(B, A)
        for k in range(2):
            x = workloads.single_channel(B, i=4 + 2 * k)
            out["wbfm%d_%d_%d" % (k, B, A)] = d.run(x)
            out["in%d_%d_%d" % (k, B, A)] = np.array(digest(x))
    # Ill-conditioned input: for station 5 at 60 kHz the pilot's analytic signal
    # passes within 8e-5 of zero at the last sample, where Im(z^2)/|z^2|
    # (pll.py:57-58) amplifies float32 rounding by max|z|/|z| ~ 1e4.
    x = workloads.single_channel(60000, i=5)
    out["illcond_60000_12000"] = rc.WBFM(60000, 12000).run(x)
    out["illcond_in"] = np.array(digest(x))
    save("wbfm", **out)

    # ---- tuner, through the reference's own caller loop ----------------------
    # (examples/multi_fm_server.py:98-106: load, then per channel run ->
    # demodulator.run -> address_bytes + payload)
    out = {}
    N = 600000
    B = 60000
    A = 12000
    specs = [(100.00e6, B, rc.WBFM), (100.05e6, B, rc.WBFM), (99.93e6, B, rc.MFM),
             (100.21e6, 50001, rc.FM), (99.80e6, B, rc.WBFM)]
    tuner = rc.Tuner()
    geo = []
    for (f, bw, cls) in specs:
        tuner.add_channel(f, bw, cls(bw, A if bw == B else 10001))
        geo.append((tuner.input_frequency, tuner.input_bandwidth))
    out["geometry"] = np.array(geo, np.float64)
    try:
        tuner.request_bandwidth(1000.0)
        out["request_low_raises"] = np.array(0)
    except ValueError:
        out["request_low_raises"] = np.array(1)
    tuner.request_bandwidth(float(N))
    out["input_frequency"] = np.array(tuner.input_frequency)
    centres = [s[0] for s in specs]
    for k in range(2):
        x = workloads.wideband(N, tuner.input_frequency, centres, B, gain=0.4)
        if k:
            x = np.roll(x, 12345)
        out["in%d" % k] = np.array(digest(x))
        tuner.load(x)
        for ch in tuner.channels():
            iq = tuner.run(ch.index)
            audio = ch.demodulator.run(iq)
            if k == 0 and ch.index in (0, 2, 3):
                out["iq%d_ch%d" % (k, ch.index)] = iq
            out["audio%d_ch%d" % (k, ch.index)] = audio
            out["addr_ch%d" % ch.index] = np.frombuffer(ch.address_bytes, np.uint8)
    # a few raw spectrum bins pin Tuner.load
    out["spectrum_bins"] = tuner._buffer[[0, 1, 2, 1000, N // 2, N - 1]]
    save("tuner", **out)

    # odd wideband size / up-sampling branch of the freq-domain resampler
    out = {}
    N = 90001
    tuner = rc.Tuner()
    tuner.add_channel(50e6, 30000, None)
    tuner.add_channel(50.02e6, 20001, None)
    tuner.request_bandwidth(float(N))
    g = np.random.default_rng(3)
    x = (g.standard_normal(N) + 1j * g.standard_normal(N)).astype(np.complex64)
    out["in"] = np.array(digest(x))
    tuner.load(x)
    out["iq_ch0"] = tuner.run(0)
    out["iq_ch1"] = tuner.run(1)
    out["geometry"] = np.array([tuner.input_frequency, tuner.input_bandwidth])
    save("tuner_odd", **out)
    del rng


if __name__ == "__main__":
    main()
