"""Replay of every golden fixture through an implementation of the class surface.

``impl`` is any namespace exposing Decimate, Bandpass, Deemphasis, PLL, FM, MFM,
WBFM and Tuner with the reference's signatures: the oracle (CPU tests) or the
HIP-backed ``radiocore`` package (GPU tests).  Each function returns a list of
(case name, error) pairs, error = max|delta| / max|expected|.
"""

import numpy as np

import workloads
from conftest import rel_err


def _asnp(x):
    return np.asarray(x)


def decimate_cases(impl, g):
    out = []
    for (n, m) in [(24000, 4800), (24001, 4801), (24000, 4801), (12500, 48000),
                   (12501, 48000), (24000, 24000), (24001, 24001)]:
        x = np.random.default_rng(n + m).standard_normal(n).astype(np.float32)
        g.check_input("real_%d_%d_in" % (n, m), x)
        y = _asnp(impl.Decimate(n, m).run(x))
        assert y.dtype == np.float32
        out.append(("real %d->%d" % (n, m), rel_err(y, g["real_%d_%d" % (n, m)])))
    for (n, m) in [(100000, 2500), (100001, 2501), (2500, 10000), (20000, 20000)]:
        r = np.random.default_rng(n + m)
        x = (r.standard_normal(n) + 1j * r.standard_normal(n)).astype(np.complex64)
        g.check_input("cplx_%d_%d_in" % (n, m), x)
        y = _asnp(impl.Decimate(n, m).run(x))
        assert y.dtype == np.complex64
        out.append(("cplx %d->%d" % (n, m), rel_err(y, g["cplx_%d_%d" % (n, m)])))
    return out


def bandpass_cases(impl, g):
    out = []
    for (n, lo, hi, taps) in [(60000, 18950.0, 19050.0, 41), (100000, 18950.0, 19050.0, 41),
                              (48000, 300.0, 3000.0, 61)]:
        x = np.random.default_rng(n).standard_normal(n).astype(np.float32)
        g.check_input("in_%d_%d" % (n, taps), x)
        bp = impl.Bandpass(n, lo, hi, num_taps=taps)
        out.append(("taps %d/%d" % (n, taps),
                    rel_err(_asnp(bp._taps[0]), g["taps_%d_%d" % (n, taps)])))
        out.append(("filtfilt %d/%d" % (n, taps),
                    rel_err(_asnp(bp.run(x)), g["y_%d_%d" % (n, taps)])))
    return out


def deemphasis_cases(impl, g):
    out = []
    for (n, tau) in [(48000, 75e-6), (32000, 75e-6), (8000, 50e-6), (4800, 75e-6)]:
        de = impl.Deemphasis(n, tau)
        out.append(("taps %d" % n, rel_err(_asnp(de._taps[0]), g["taps_%d" % n])))
        out.append(("zi %d" % n, rel_err(_asnp(de._state), g["zi_%d" % n])))
        r = np.random.default_rng(n)
        for k in range(2 if n != 32000 else 0):
            x = (0.5 * r.standard_normal(n)).astype(np.float32)
            g.check_input("in%d_%d" % (k, n), x)
            out.append(("y%d %d" % (k, n), rel_err(_asnp(de.run(x)), g["y%d_%d" % (k, n)])))
        out.append(("zf %d" % n, rel_err(_asnp(de._state), g["zf_%d" % n])))
    return out


def pll_cases(impl, g):
    out = []
    for n in (6000, 6001):
        x = np.random.default_rng(n).standard_normal(n).astype(np.float32)
        g.check_input("in_%d" % n, x)
        pll = impl.PLL()
        pll.step(x)
        zr = g["z_%d" % n]
        out.append(("analytic %d" % n, rel_err(_asnp(pll._baseline), zr)))
        # x / |x| is ill-conditioned where the analytic signal passes near zero
        # (white-noise input does): an error e in z moves the unit vector by
        # e / |z|.  Weight the error by |z| / max|z| so the case measures the
        # implementation and not the conditioning of the input.
        w = np.abs(zr) / np.max(np.abs(zr))
        for name, got in (("image2", pll.image(2)), ("real2", pll.real(2)), ("image1", pll.image())):
            d = np.abs(_asnp(got) - g["%s_%d" % (name, n)]) * w
            out.append(("%s %d" % (name, n), float(np.max(d))))
    return out


def fm_cases(impl, g):
    out = []
    for (B, A, dev, stereo) in [(24000, 4800, 5e3, False), (24001, 4801, 5e3, False),
                                (12500, 8000, 2.5e3, False), (240000, 48000, 75e3, True)]:
        x = workloads.single_channel(B, i=1, deviation=dev, stereo=stereo)
        g.check_input("in_%d_%d" % (B, A), x)
        d = impl.FM(B, A)
        assert d.channels == 1
        y = _asnp(d.run(x))
        assert y.shape == (A, 1) and y.dtype == np.float32
        out.append(("FM %d->%d" % (B, A), rel_err(y, g["fm_%d_%d" % (B, A)])))
    return out


def mfm_cases(impl, g):
    out = []
    for (B, A, dev, stereo) in [(24000, 4800, 5e3, False), (240000, 48000, 75e3, True)]:
        d = impl.MFM(B, A)
        assert d.channels == 1
        for k in range(2):
            x = workloads.single_channel(B, i=2 + k, deviation=dev, stereo=stereo)
            g.check_input("in%d_%d_%d" % (k, B, A), x)
            y = _asnp(d.run(x))
            assert y.shape == (A, 1) and y.dtype == np.float32
            out.append(("MFM buf%d %d->%d" % (k, B, A),
                        rel_err(y, g["mfm%d_%d_%d" % (k, B, A)])))
    return out


def wbfm_cases(impl, g, sizes=((60000, 12000), (240000, 48000), (256000, 32000))):
    out = []
    for (B, A) in sizes:
        d = impl.WBFM(B, A)
        assert d.channels == 2
        for k in range(2):
            x = workloads.single_channel(B, i=4 + 2 * k)
            g.check_input("in%d_%d_%d" % (k, B, A), x)
            y = _asnp(d.run(x))
            assert y.shape == (1, A, 2) and y.dtype == np.float32
            out.append(("WBFM buf%d %d->%d" % (k, B, A),
                        rel_err(y, g["wbfm%d_%d_%d" % (k, B, A)])))
    return out


GEO_44100_CASES = (("WBFM", 240000, 44100), ("WBFM", 250000, 48000), ("MFM", 250000, 48000), ("MFM", 240000, 44100))


def geo_44100_cases(impl, g):
    """tests/golden/wbfm_44100.npz (make_golden_44100.py): CD-rate audio (44 100 = 2^2 3^2 5^2 7^2) and the reference's
    single-station geometry 250 000 -> 48 000 (examples/receive_fm.py:18-19), two buffers each."""
    out = []
    for kind, B, A in GEO_44100_CASES:
        d = getattr(impl, kind)(B, A)
        for k in range(2):
            x = workloads.single_channel(B, i=6 + k, stereo=(kind == "WBFM"))
            g.check_input("in_%s%d_%d_%d" % (kind.lower(), k, B, A), x)
            y = _asnp(d.run(x))
            assert y.shape == ((1, A, 2) if kind == "WBFM" else (A, 1)) and y.dtype == np.float32
            out.append(("%s buf%d %d->%d" % (kind, k, B, A), rel_err(y, g["%s%d_%d_%d" % (kind.lower(), k, B, A)])))
    return out


def wbfm_illcond_case(impl, g):
    """Station 5 at 60 kHz: min|z| / max|z| = 7.7e-5 at the last sample, so the
    normalised 38 kHz carrier there moves by ~1e4 x the rounding error of z."""
    x = workloads.single_channel(60000, i=5)
    g.check_input("illcond_in", x)
    y = _asnp(impl.WBFM(60000, 12000).run(x))
    return [("WBFM ill-conditioned 60000->12000", rel_err(y, g["illcond_60000_12000"]))]


TUNER_N = 600000
TUNER_B = 60000
TUNER_A = 12000


def tuner_specs(impl):
    B = TUNER_B
    return [(100.00e6, B, impl.WBFM), (100.05e6, B, impl.WBFM), (99.93e6, B, impl.MFM),
            (100.21e6, 50001, impl.FM), (99.80e6, B, impl.WBFM)]


def tuner_cases(impl, g):
    """The reference's caller loop, examples/multi_fm_server.py:98-106."""
    out = []
    N, B, A = TUNER_N, TUNER_B, TUNER_A
    specs = tuner_specs(impl)
    tuner = impl.Tuner()
    geo = []
    for (f, bw, cls) in specs:
        tuner.add_channel(f, bw, cls(bw, A if bw == B else 10001))
        geo.append((tuner.input_frequency, tuner.input_bandwidth))
    assert np.array_equal(np.array(geo, np.float64), g["geometry"])
    raised = 0
    try:
        tuner.request_bandwidth(1000.0)
    except ValueError:
        raised = 1
    assert raised == int(g["request_low_raises"])
    tuner.request_bandwidth(float(N))
    assert tuner.input_frequency == float(g["input_frequency"])
    centres = [s[0] for s in specs]
    for k in range(2):
        x = workloads.wideband(N, tuner.input_frequency, centres, B, gain=0.4)
        if k:
            x = np.roll(x, 12345)
        g.check_input("in%d" % k, x)
        tuner.load(x)
        for ch in tuner.channels():
            iq = tuner.run(ch.index)
            audio = _asnp(ch.demodulator.run(iq))
            key = "iq%d_ch%d" % (k, ch.index)
            if key in g:
                out.append((key, rel_err(_asnp(iq), g[key])))
            out.append(("audio%d_ch%d" % (k, ch.index),
                        rel_err(audio, g["audio%d_ch%d" % (k, ch.index)])))
            assert ch.address_bytes == g["addr_ch%d" % ch.index].tobytes()
    return out


def tuner_odd_cases(impl, g):
    N = 90001
    tuner = impl.Tuner()
    tuner.add_channel(50e6, 30000, None)
    tuner.add_channel(50.02e6, 20001, None)
    tuner.request_bandwidth(float(N))
    assert np.array_equal(np.array([tuner.input_frequency, tuner.input_bandwidth]), g["geometry"])
    r = np.random.default_rng(3)
    x = (r.standard_normal(N) + 1j * r.standard_normal(N)).astype(np.complex64)
    g.check_input("in", x)
    tuner.load(x)
    return [("odd ch0", rel_err(_asnp(tuner.run(0)), g["iq_ch0"])),
            ("odd ch1", rel_err(_asnp(tuner.run(1)), g["iq_ch1"]))]


FM_OFFSETS = (5000, 50000)


def fm_offcentre_cases(impl, g):
    """Stations 5 kHz and 50 kHz off the channel centre (tests/golden/make_golden_offcentre.py): per offset the
    error of impl.FM against the reference's own output and against the float64 evaluation of the same
    mathematics, and the reference's own error against that truth.  Returns {offset: (vs_ref, vs_truth, ref_vs_truth)}."""
    out = {}
    for off in FM_OFFSETS:
        x = workloads.single_channel(240000, i=1, offset=off)
        g.check_input("in_%d" % off, x)
        y = _asnp(impl.FM(240000, 48000).run(x))[:, 0]
        ref, truth = g["ref_%d" % off], g["truth_%d" % off]
        out[off] = (rel_err(y, ref), rel_err(y, truth), rel_err(ref, truth))
    return out
