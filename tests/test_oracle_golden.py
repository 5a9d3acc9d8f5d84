"""CPU: the oracle reproduces every golden vector captured from the reference.

This is what pins the oracle: the fixtures under
tests/golden/ were produced by tests/golden/make_golden.py importing the
reference itself.  The oracle restates the same float32 algorithm, so it must
agree far inside the 1e-4 product tolerance.
"""

import pytest

import golden_cases as gc
import radiocore_oracle as oracle

# float32 rounding-order differences only (different FFT call order, FIR sums)
ORACLE_TOL = 5e-6


def _check(cases, tol=ORACLE_TOL):
    bad = [(n, e) for (n, e) in cases if not e <= tol]
    assert not bad, bad


def test_decimate(golden):
    _check(gc.decimate_cases(oracle, golden("decimate")))


def test_bandpass(golden):
    _check(gc.bandpass_cases(oracle, golden("bandpass")))


def test_deemphasis(golden):
    _check(gc.deemphasis_cases(oracle, golden("deemphasis")))


def test_pll(golden):
    _check(gc.pll_cases(oracle, golden("pll")))


def test_fm(golden):
    _check(gc.fm_cases(oracle, golden("fm")))


def test_mfm(golden):
    _check(gc.mfm_cases(oracle, golden("mfm")))


def test_wbfm(golden):
    _check(gc.wbfm_cases(oracle, golden("wbfm")))


def test_cd_rate_audio_and_the_reference_example_geometry(golden):
    _check(gc.geo_44100_cases(oracle, golden("wbfm_44100")))


def test_wbfm_ill_conditioned(golden):
    # rounding in z (2e-7 of peak) x conditioning (1.3e4) / decimation smoothing
    _check(gc.wbfm_illcond_case(oracle, golden("wbfm")), tol=1e-3)


def test_tuner_server_loop(golden):
    _check(gc.tuner_cases(oracle, golden("tuner")))


def test_tuner_odd(golden):
    _check(gc.tuner_odd_cases(oracle, golden("tuner_odd")))


def test_fm_off_centre_carrier(golden):
    """fm.py:62 unwraps in float32: with the carrier 5 / 50 kHz off centre the reference's own output is 2.8e-4 /
    2.2e-3 away from the float64 evaluation of its formula.  The oracle restates that float32 algorithm, so it
    follows the reference (not the truth) to float32 rounding."""
    res = gc.fm_offcentre_cases(oracle, golden("fm_offcentre"))
    for off, (vs_ref, vs_truth, ref_vs_truth) in res.items():
        assert vs_ref <= ORACLE_TOL, (off, vs_ref)
        assert ref_vs_truth > 1e-4, (off, ref_vs_truth)          # the reference's float32 unwrap noise
        assert abs(vs_truth - ref_vs_truth) <= 0.05 * ref_vs_truth


def _stale_window_scenario(impl):
    """tuner.py:155-157: the spectral window is built by the first run() and never refreshed, so after a later
    request_bandwidth() buffers of the NEW size are refused and buffers of the old size still work (observed on the
    reference itself: 'window must have the same length as data' from scipy.signal.resample)."""
    import numpy as np
    t = impl.Tuner()
    t.add_channel(1e6, 100, None)
    t.add_channel(1.0005e6, 100, None)
    t.request_bandwidth(1000.0)
    x1 = np.random.default_rng(0).standard_normal(1000).astype(np.complex64)
    x2 = np.random.default_rng(1).standard_normal(2000).astype(np.complex64)
    t.load(x1)
    first = np.asarray(t.run(0))
    assert first.shape == (100,)
    t.request_bandwidth(2000.0)
    t.load(x2)
    with pytest.raises(ValueError, match="window must have the same length as data"):
        t.run(0)
    t.load(x1)
    assert np.array_equal(np.asarray(t.run(0)), first)
    fresh = impl.Tuner()
    fresh.add_channel(1e6, 100, None)
    fresh.request_bandwidth(1000.0)
    fresh.load(x2)                                   # wrong size from the start: refused at the first run
    with pytest.raises(ValueError, match="window must have the same length as data"):
        fresh.run(0)


def test_tuner_keeps_its_first_window():
    _stale_window_scenario(oracle)


def test_size_mismatch_raises():
    import numpy as np
    for cls in (oracle.FM, oracle.MFM, oracle.WBFM):
        with pytest.raises(ValueError, match="input_sig size and input_size mismatch"):
            cls(60000, 12000).run(np.zeros(59999, np.complex64))
    with pytest.raises(ValueError):
        oracle.Decimate(100, 10).run(np.zeros(99, np.float32))
    with pytest.raises(IndexError):
        t = oracle.Tuner()
        t.add_channel(1e6, 1000, None)
        t.load(np.zeros(1000, np.complex64))
        t.run(3)
