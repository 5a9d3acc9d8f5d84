"""CPU: every scipy routine the oracle restates, against the installed scipy,
plus the closed forms the HIP kernels implement against the step-by-step ones.

scipy is the un-vendored third-party dependency that holds the reference's
arithmetic (pyproject.toml:26); it is present in this image.
"""

import numpy as np
import pytest

import radiocore_oracle as oracle
from conftest import rel_err

ss = pytest.importorskip("scipy.signal")


@pytest.mark.parametrize("name", ["hann", "hamm"])
@pytest.mark.parametrize("n", [1, 2, 7, 8, 24000, 24001])
def test_windows(name, n):
    assert np.allclose(oracle.periodic_window(name, n), ss.get_window(name, n), atol=1e-15)
    import scipy.fft
    assert np.array_equal(oracle.shifted_window(name, n),
                          scipy.fft.fftshift(ss.get_window(name, n)))


@pytest.mark.parametrize("n,m", [(1000, 200), (1001, 201), (1000, 201), (1001, 200),
                                 (200, 1000), (201, 1000), (200, 1001), (64, 64), (4, 2), (3, 2)])
@pytest.mark.parametrize("cplx", [False, True])
def test_resample(n, m, cplx):
    r = np.random.default_rng(n * 7 + m)
    x = r.standard_normal(n).astype(np.float32)
    if cplx:
        x = (x + 1j * r.standard_normal(n)).astype(np.complex64)
    w = oracle.shifted_window("hamm", n)
    assert rel_err(oracle.resample(x, m, window=w), ss.resample(x, m, window=w)) < 2e-6
    if cplx:
        X = np.fft.fft(x)
        assert rel_err(oracle.resample(X.copy(), m, window=w, domain="freq"),
                       ss.resample(X.copy(), m, window=w, domain="freq")) < 2e-6


@pytest.mark.parametrize("taps,lo,hi", [(41, 18950 / 120000, 19050 / 120000), (61, 0.1, 0.3), (51, 0.02, 0.9)])
def test_firwin(taps, lo, hi):
    want = ss.firwin(taps, [lo, hi], pass_zero=False, window="hamm")
    assert np.allclose(oracle.firwin_bandpass(taps, lo, hi), want, rtol=1e-13, atol=1e-16)


def test_fir_state_and_filter():
    r = np.random.default_rng(0)
    b = r.standard_normal(51).astype(np.float32)
    assert np.allclose(oracle.fir_zi(b), ss.lfilter_zi(b, np.array(1.0, np.float32)), rtol=2e-6, atol=2e-6)
    zi = r.standard_normal(50).astype(np.float32)
    for L in (4800, 50, 49, 7):
        x = r.standard_normal(L).astype(np.float32)
        y, zf = oracle.fir_filter(b, x, zi)
        y2, zf2 = ss.lfilter(b, np.array(1.0, np.float32), x, zi=zi)
        assert y.dtype == np.float32
        assert rel_err(y, y2) < 2e-6 and rel_err(zf, zf2) < 2e-6


@pytest.mark.parametrize("n", [4800, 8000, 32000, 48000])
def test_deemphasis_design(n):
    x = np.exp(-1 / (n * 75e-6))
    _, d = ss.dimpulse(ss.dlti([1 - x], [1, -x]), n=51)
    b, zi = oracle.deemphasis_taps(n, 75e-6)
    assert np.array_equal(b, np.squeeze(d).astype(np.float32))
    assert np.allclose(zi, ss.lfilter_zi(b, np.array(1.0, np.float32)).astype(np.float32), atol=1e-7)


@pytest.mark.parametrize("n", [124, 1000, 60000])
@pytest.mark.parametrize("taps", [41, 61])
def test_filtfilt_both_forms(n, taps):
    if n <= 3 * taps:
        with pytest.raises(ValueError):
            oracle.filtfilt_fir(np.ones(taps, np.float32), np.ones(n, np.float32))
        return
    r = np.random.default_rng(n)
    b = ss.firwin(taps, [0.1, 0.2], pass_zero=False).astype(np.float32)
    x = r.standard_normal(n).astype(np.float32)
    want = ss.filtfilt(b, np.array([1.0], np.float32), x)
    assert rel_err(oracle.filtfilt_fir(b, x), want) < 2e-6
    # the (2T-1)-tap symmetric form with a (T-1)-sample odd extension
    assert rel_err(oracle.filtfilt_fir_closed_form(b, x), want) < 2e-6


@pytest.mark.parametrize("n", [6000, 6001, 2, 3])
def test_hilbert(n):
    x = np.random.default_rng(n).standard_normal(n).astype(np.float32)
    z = oracle.hilbert(x)
    assert z.dtype == np.complex64
    assert rel_err(z, ss.hilbert(x)) < 2e-6


def test_unwrap_matches_numpy():
    r = np.random.default_rng(1)
    p = np.cumsum(r.uniform(-3, 3, 50000)).astype(np.float32)
    p = (np.mod(p + np.pi, 2 * np.pi) - np.pi).astype(np.float32)
    assert np.array_equal(oracle.unwrap(p), np.unwrap(p))


def test_same_size_decimate_is_three_tap_circular_fir():
    """decimate.py with m == n (wbfm.py:42): y[i] = .54 x[i] + .23 (x[i-1] + x[i+1]),
    indices mod n, for even n.  The HIP discriminator kernel fuses this form."""
    n = 60000
    x = np.random.default_rng(5).standard_normal(n).astype(np.float32)
    want = oracle.Decimate(n, n).run(x)
    got = 0.54 * x + 0.23 * (np.roll(x, 1) + np.roll(x, -1))
    assert rel_err(got, want) < 2e-6


@pytest.mark.parametrize("n,bw,roll", [(90001, 20001, -777), (90000, 30000, 12345), (90000, 30001, 0),
                                      (90001, 30000, 5), (4096, 4096, 3)])
def test_tuner_pruned_equals_literal(n, bw, roll):
    r = np.random.default_rng(n + bw)
    X = (r.standard_normal(n) + 1j * r.standard_normal(n)).astype(np.complex64)
    w = oracle.shifted_window("hann", n)
    want = oracle.resample(np.roll(X, roll), bw, window=w, domain="freq")
    Y = oracle.tuner_channel_spectrum(X, n, roll, bw)
    got = np.fft.ifft(Y) * np.float32(bw / n)
    assert rel_err(got, want) < 2e-6
