"""CPU oracle for the Tuner -> FM / MFM / WBFM hot path.  TEST INFRASTRUCTURE ONLY.

This file is a numpy-only restatement of the arithmetic the reference
(luigifcruz/radio-core v1.0.0) performs on its CPU path.  It exists so the
HIP kernels can be checked on a box where the reference itself is absent.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it; the product package (``radio-core_amd/``) never
does and fails loudly when its HIP library is missing.

Parity status: PINNED.  The reference's own tests hold no vectors for this
path (they only cover Buffer/Carrousel/RingBuffer), so the pins are golden
vectors captured by importing the reference in the build container
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``; numpy 2.2.6,
scipy 1.15.3).  ``tests/test_oracle_golden.py`` replays every fixture through
this file; ``tests/test_oracle_scipy.py`` additionally checks each restated
scipy routine against the installed scipy.

The reference delegates its arithmetic to third-party code that is not
vendored in its tree: numpy (pyproject.toml:25 ``^1.21``) and scipy
(pyproject.toml:26 ``^1.5``).  The scipy routines restated here, following the
published scipy 1.15.3 algorithms, are ``signal.resample``, ``signal.firwin``,
``signal.filtfilt`` (pad method, odd extension), ``signal.lfilter`` /
``lfilter_zi`` for FIR filters, ``signal.dimpulse`` for a one-pole system,
``signal.hilbert`` and ``signal.get_window('hann'|'hamm')``; plus
``numpy.unwrap``.  Working precision follows the reference: complex64 /
float32 data with float64 windows multiplied in place.

Every function cites the reference file:line it follows (paths relative to the
reference checkout).
"""

import numpy as np

__all__ = [
    "periodic_window", "shifted_window", "resample_spectrum", "resample",
    "firwin_bandpass", "fir_zi", "fir_filter", "filtfilt_fir",
    "filtfilt_fir_closed_form", "deemphasis_taps", "hilbert", "unwrap",
    "discriminator", "tuner_geometry", "tuner_channel_spectrum",
    "Decimate", "Bandpass", "Deemphasis", "PLL", "FM", "MFM", "WBFM",
    "Tuner", "Channel",
]

_SIZE_ERR = "input_sig size and input_size mismatch"


# --------------------------------------------------------------------------
# windows  (scipy.signal.get_window, fftbins=True; scipy.fft.fftshift)
# --------------------------------------------------------------------------

def periodic_window(name, n):
    """``get_window(name, n)`` for the two names the reference uses.

    tuner.py:156 ("hann"), decimate.py:32 ("hamm").  scipy builds the periodic
    (DFT-even) form by evaluating the symmetric window on n+1 points and
    dropping the last one: w[i] = a0 - (1-a0) cos(2 pi i / n).
    """
    a0 = {"hann": 0.5, "hamm": 0.54, "hamming": 0.54}[name]
    if n == 1:
        return np.ones(1)
    # scipy evaluates general_cosine on linspace(-pi, pi, n+1) and drops the
    # last point: a0 + (1-a0) cos(-pi + 2 pi i / n).
    fac = np.linspace(-np.pi, np.pi, n + 1)[:-1]
    return a0 + (1.0 - a0) * np.cos(fac)


def shifted_window(name, n):
    """fftshift(get_window(name, n)) -- tuner.py:156-157, decimate.py:32-33."""
    return np.roll(periodic_window(name, n), n // 2)


# --------------------------------------------------------------------------
# scipy.signal.resample, restated (scipy 1.15.3 _signaltools.py)
# --------------------------------------------------------------------------

def resample_spectrum(X, num, nx, real_input, W=None):
    """Spectrum-side half of ``scipy.signal.resample``.

    X is the length-nx complex spectrum (or the nx//2+1 half spectrum when
    real_input); it is modified in place exactly like scipy does.  Returns the
    output spectrum Y (num bins, or num//2+1 for real input) before the
    inverse transform.  Used by tuner.py:160-161 (domain="freq") and
    decimate.py:48 (domain="time").
    """
    if W is not None:
        if real_input:
            Wr = W.copy()
            Wr[1:] += Wr[-1:0:-1]
            Wr[1:] *= 0.5
            X *= Wr[:X.shape[0]]
        else:
            X *= W
    Y = np.zeros(num // 2 + 1 if real_input else num, X.dtype)
    n = min(num, nx)
    nyq = n // 2 + 1
    Y[:nyq] = X[:nyq]
    if not real_input and n > 2:
        Y[nyq - n:] = X[nyq - n:]
    if n % 2 == 0:
        if num < nx:
            if real_input:
                Y[n // 2] *= 2.0
            else:
                # scipy slices [-n//2 : -n//2 + 1]; for n == 2 that slice is
                # empty (stop index 0), so the merge silently does not happen.
                sl = slice(-n // 2, -n // 2 + 1)
                Y[sl] += X[sl]
        elif nx < num:
            Y[n // 2] *= 0.5
            if not real_input:
                Y[num - n // 2] = Y[n // 2]
    return Y


def resample(x, num, window=None, domain="time"):
    """``scipy.signal.resample(x, num, window=ndarray, domain=...)``."""
    x = np.asarray(x)
    nx = x.shape[0]
    real_input = np.isrealobj(x)
    if domain == "time":
        X = np.fft.rfft(x) if real_input else np.fft.fft(x)
    else:
        X = x
    Y = resample_spectrum(X, num, nx, real_input, window)
    y = np.fft.irfft(Y, num) if real_input else np.fft.ifft(Y)
    y *= float(num) / float(nx)
    return y


# --------------------------------------------------------------------------
# FIR design and filtering (firwin, lfilter, lfilter_zi, filtfilt, dimpulse)
# --------------------------------------------------------------------------

def firwin_bandpass(numtaps, lo, hi, window="hamm"):
    """``firwin(numtaps, [lo, hi], pass_zero=False, window=window)``.

    bandpass.py:50-52; lo/hi are fractions of Nyquist.  float64.
    """
    a0 = {"hamm": 0.54, "hamming": 0.54, "hann": 0.5}[window]
    alpha = 0.5 * (numtaps - 1)
    m = np.arange(numtaps) - alpha
    h = hi * np.sinc(hi * m) - lo * np.sinc(lo * m)
    # symmetric window (fftbins=False)
    fac = np.linspace(-np.pi, np.pi, numtaps)
    h = h * (a0 + (1.0 - a0) * np.cos(fac))
    # unit gain at the centre of the (first) pass band
    fc = 0.5 * (lo + hi)
    h /= np.sum(h * np.cos(np.pi * m * fc))
    return h


def fir_zi(b):
    """``lfilter_zi(b, [1])`` for an FIR filter: zi[k] = sum_{j>k} b[j].

    deemphasis.py:48, and inside filtfilt (bandpass.py:72).  scipy solves
    (I - A^T) zi = B with A the companion matrix of a = [1, 0, ...]; for an
    FIR that system is upper-bidiagonal and the solution is the reversed
    cumulative sum of b[1:].
    """
    b = np.asarray(b)
    return np.cumsum(b[:0:-1])[::-1].astype(b.dtype, copy=True)


def fir_filter(b, x, zi):
    """``lfilter(b, 1, x, zi=zi)`` for FIR b; returns (y, zf).

    deemphasis.py:64.  scipy runs a transposed direct form II in the common
    dtype; here the same sums are formed with a correlation, then the initial
    state is added to the first len(zi) outputs and the final state is rebuilt
    from the tail of the input.
    """
    b = np.asarray(b)
    x = np.asarray(x)
    dt = np.result_type(b, x, zi)
    nb, L = len(b), len(x)
    y = np.convolve(x.astype(dt), b.astype(dt))[:L].astype(dt)
    k = min(L, nb - 1)
    y[:k] += zi[:k].astype(dt)
    zf = np.zeros(nb - 1, dt)
    for s in range(nb - 1):
        # zf[s] = sum_{i>=0} b[s+1+i] x[L-1-i]  (+ surviving part of zi)
        m = min(nb - 1 - s, L)
        acc = np.dot(b[s + 1:s + 1 + m].astype(np.float64),
                     x[L - 1::-1][:m].astype(np.float64)) if m > 0 else 0.0
        if s + L < nb - 1:
            acc += float(zi[s + L])
        zf[s] = acc
    return y, zf


def _odd_ext(x, n):
    return np.concatenate((2 * x[0] - x[n:0:-1], x, 2 * x[-1] - x[-2:-(n + 2):-1]))


def filtfilt_fir(b, x):
    """``filtfilt(b, [1], x)`` (method="pad", padtype="odd") -- bandpass.py:72.

    Step-for-step restatement: odd-extend by 3*len(b), forward lfilter with
    zi*ext[0], backward lfilter with zi*y[-1], reverse, trim.
    """
    b = np.asarray(b)
    x = np.asarray(x)
    edge = 3 * len(b)
    if x.shape[0] <= edge:
        raise ValueError("The length of the input vector x must be greater "
                         "than padlen, which is %d." % edge)
    zi = fir_zi(b)
    ext = _odd_ext(x, edge)
    y, _ = fir_filter(b, ext, zi * ext[0])
    y, _ = fir_filter(b, y[::-1], zi * y[-1])
    return y[::-1][edge:-edge]


def filtfilt_fir_closed_form(b, x):
    """Closed form of ``filtfilt_fir`` that the HIP kernel implements.

    For an FIR b of length T the zero-phase result is the symmetric
    (2T-1)-tap correlation g = b * reverse(b) applied to x odd-extended by
    T-1 samples at both ends; the lfilter_zi initial conditions only affect
    padding samples that are trimmed away.  
    """
    b = np.asarray(b, np.float64)
    x = np.asarray(x)
    g = np.convolve(b, b[::-1])
    h = len(b) - 1
    e = _odd_ext(x.astype(np.float64), h)
    return np.convolve(e, g, mode="valid").astype(x.dtype)


def deemphasis_taps(input_size, rate, dtype="float32"):
    """Taps and initial state of the de-emphasis filter -- deemphasis.py:37-49.

    x = exp(-1/(fs*tau)); dimpulse of (1-x)/(z-x), 51 samples: b[0] = 0,
    b[i] = (1-x) x^(i-1).  State = lfilter_zi(b, 1).
    """
    x = np.exp(-1.0 / (input_size * rate))
    # scipy's dlsim iterates x_{k+1} = A x_k + B u_k, y_k = C x_k with
    # (A, B, C) = tf2ss([1-x], [1, -x]) = (x, 1, 1-x): repeated multiplication.
    b = np.zeros(51)
    s = 0.0
    u = 1.0
    for i in range(51):
        b[i] = (1.0 - x) * s
        s = x * s + u
        u = 0.0
    taps = b.astype(dtype)
    return taps, fir_zi(taps).astype(dtype)


def hilbert(x):
    """``scipy.signal.hilbert`` -- pll.py:34."""
    x = np.asarray(x)
    n = x.shape[0]
    Xf = np.fft.fft(x)
    h = np.zeros(n, Xf.dtype)
    if n % 2 == 0:
        h[0] = h[n // 2] = 1
        h[1:n // 2] = 2
    else:
        h[0] = 1
        h[1:(n + 1) // 2] = 2
    return np.fft.ifft(Xf * h)


def unwrap(p):
    """``numpy.unwrap`` in the array's own precision -- fm.py:62."""
    p = np.asarray(p)
    dt = p.dtype
    pi = dt.type(np.pi)
    period = dt.type(2 * np.pi)
    dd = np.diff(p)
    ddmod = np.mod(dd + pi, period) - pi
    ddmod[(ddmod == -pi) & (dd > 0)] = pi
    corr = ddmod - dd
    corr[np.abs(dd) < pi] = 0
    up = p.copy()
    up[1:] = p[1:] + np.cumsum(corr, dtype=dt)
    return up


def discriminator(iq):
    """fm.py:60-65: angle -> unwrap -> diff -> pad(1,0) -> / pi, float32."""
    a = np.angle(np.asarray(iq))
    a = unwrap(a)
    d = np.diff(a)
    d = np.pad(d, (1, 0))
    return d / np.pi


# --------------------------------------------------------------------------
# classes mirroring the reference surface
# --------------------------------------------------------------------------

class Decimate:
    """decimate.py:21-50."""

    def __init__(self, input_size, output_size, cuda=False):
        self._input_size = int(input_size)
        self._output_size = int(output_size)
        self._win = shifted_window("hamm", self._input_size)

    def run(self, input_sig):
        if len(input_sig) != self._input_size:
            raise ValueError(_SIZE_ERR)
        return resample(np.asarray(input_sig), self._output_size, window=self._win)


class Bandpass:
    """bandpass.py:29-74."""

    def __init__(self, input_size, start_freq, stop_freq, dtype="float32",
                 num_taps=61, window="hamm", cuda=False):
        self._input_size = int(input_size)
        nyq = 0.5 * self._input_size
        b = firwin_bandpass(int(num_taps), float(start_freq) / nyq,
                            float(stop_freq) / nyq, window)
        self._taps = (np.array(b, dtype=dtype), np.array([1.0], dtype=dtype))

    def run(self, input_sig):
        if len(input_sig) != self._input_size:
            raise ValueError(_SIZE_ERR)
        return filtfilt_fir(self._taps[0], np.asarray(input_sig))


class Deemphasis:
    """deemphasis.py:26-66.  Stateful: the 50-element FIR history carries over."""

    def __init__(self, input_size, rate=75e-6, dtype="float32", cuda=False):
        self._input_size = int(input_size)
        b, zi = deemphasis_taps(self._input_size, rate, dtype)
        self._taps = (b, np.array(1.0, dtype=dtype))
        self._state = zi

    def run(self, input_sig):
        if len(input_sig) != self._input_size:
            raise ValueError(_SIZE_ERR)
        y, self._state = fir_filter(self._taps[0], np.asarray(input_sig), self._state)
        return y


class PLL:
    """pll.py:19-58.  'PLL' = analytic signal of the pilot; no feedback loop."""

    def __init__(self, cuda=False):
        self._baseline = None

    def step(self, input_sig):
        self._baseline = hilbert(input_sig)

    def real(self, mult=1.0):
        t = self._baseline ** mult
        return np.real(t) / np.abs(t)

    def image(self, mult=1.0):
        t = self._baseline ** mult
        return np.imag(t) / np.abs(t)


class FM:
    """fm.py:26-72."""

    def __init__(self, input_size, output_size, deemphasis=75e-6, cuda=False):
        self._input_size = int(input_size)
        self._output_size = int(output_size)
        self._decimate = Decimate(self._input_size, self._output_size)

    @property
    def channels(self):
        return 1

    def run(self, input_sig, numpy_output=True):
        if len(input_sig) != self._input_size:
            raise ValueError(_SIZE_ERR)
        d = discriminator(input_sig)
        return np.expand_dims(self._decimate.run(d), axis=1)


class MFM:
    """mfm.py:29-71."""

    def __init__(self, input_size, output_size, deemphasis=75e-6, cuda=False):
        self._input_size = int(input_size)
        self._output_size = int(output_size)
        self._fm_demod = FM(self._input_size, self._output_size)
        self._deemphasis = Deemphasis(self._output_size, deemphasis)

    @property
    def channels(self):
        return 1

    def run(self, input_sig, numpy_output=True):
        a = self._fm_demod.run(input_sig)[:, 0]
        a = self._deemphasis.run(a)
        a -= np.mean(a)
        a = np.clip(a, -0.999, 0.999)
        return np.expand_dims(a, axis=1)


class WBFM:
    """wbfm.py:32-105."""

    def __init__(self, input_size, output_size, deemphasis=75e-6, cuda=False):
        self._input_size = int(input_size)
        self._output_size = int(output_size)
        self._fm_demod = FM(self._input_size, self._input_size)
        self._plt_filter = Bandpass(self._input_size, 19e3 - 50, 19e3 + 50, num_taps=41)
        self._pll = PLL()
        self._decimate = Decimate(self._input_size, self._output_size)
        self._left_deemphasis = Deemphasis(self._output_size, deemphasis)
        self._right_deemphasis = Deemphasis(self._output_size, deemphasis)

    @property
    def channels(self):
        return 2

    def run(self, input_sig, numpy_output=True):
        m = self._fm_demod.run(input_sig)[:, 0]
        self._pll.step(self._plt_filter.run(m))
        lmr = (self._pll.image(2) * m) * 1.0175
        l = self._decimate.run(m + lmr)
        r = self._decimate.run(m - lmr)
        l = self._left_deemphasis.run(l)
        r = self._right_deemphasis.run(r)
        lr = np.dstack((l, r))
        lr -= np.mean(lr)
        return np.clip(lr, -0.999, 0.999)


class Channel:
    """tuner.py:9-36."""

    def __init__(self, index, bandwidth, demodulator, frequency):
        self.index = index
        self.bandwidth = bandwidth
        self.demodulator = demodulator
        self.lower_frequency = frequency - (bandwidth / 2)
        self.center_frequency = frequency
        self.higher_frequency = frequency + (bandwidth / 2)

    @property
    def address_bytes(self):
        return int(self.center_frequency).to_bytes(4, byteorder="little")


def tuner_geometry(channels):
    """tuner.py:163-174: (input_frequency, padded input_bandwidth)."""
    lo = min([c.lower_frequency for c in channels])
    hi = max([c.higher_frequency for c in channels])
    f_in = (lo + hi) / 2
    bw = hi - lo
    mean_bw = sum([c.bandwidth for c in channels])
    mean_bw //= len(channels)
    bw += (bw * -1) % mean_bw
    return f_in, bw


def tuner_channel_spectrum(X, n, roll, m):
    """Bins that ``roll`` + ``resample(domain='freq')`` keep, without the O(n) work.

    tuner.py:159-161.  Returns the length-m spectrum Y such that the channel is
    ifft(Y) * (m / n).  Only valid for m <= n (the Tuner only down-samples).
    W is the fftshifted periodic Hann window of length n evaluated where
    needed: W[k] = w[(k - n//2) mod n].
    """
    def W(k):
        i = (k - n // 2) % n
        return 0.5 - 0.5 * np.cos(2.0 * np.pi * i / n)

    Y = np.zeros(m, X.dtype)
    kp = np.arange(0, m // 2 + 1)
    Y[kp] = (X[(kp - roll) % n] * W(kp)).astype(X.dtype)
    j = np.arange(1, m - m // 2)          # negative side, bins n-j -> m-j
    kn = n - j
    Y[m - j] = (X[(kn - roll) % n] * W(kn)).astype(X.dtype)
    if m % 2 == 0 and m < n and m > 2:      # (m == 2: scipy's slice is empty)
        k = n - m // 2
        Y[m - m // 2] += (X[(k - roll) % n] * W(k)).astype(X.dtype)
    return Y


class Tuner:
    """tuner.py:52-174.

    ``run`` follows the reference literally (np.roll of the whole spectrum and
    a full-length window multiply: O(n) per channel).  ``run_pruned`` gives the
    same numbers touching only the bins that survive, for larger test sizes.
    """

    def __init__(self, cuda=False):
        self._win = None
        self._win_len = None          # run_pruned never builds the window: only its length is remembered
        self._buffer = None
        self._input_frequency = 0.0
        self._input_bandwidth = 0.0
        self._bounds = []

    @property
    def input_frequency(self):
        return self._input_frequency

    @property
    def input_bandwidth(self):
        return self._input_bandwidth

    def channels(self):
        return self._bounds

    def request_bandwidth(self, bandwidth):
        if bandwidth < self._input_bandwidth:
            raise ValueError(f"requested bandwidth ({bandwidth}) is too low, "
                             f"minimum is {self._input_bandwidth}")
        self._input_bandwidth = bandwidth

    def add_channel(self, frequency, bandwidth, demodulator):
        self._bounds.append(Channel(len(self._bounds), bandwidth, demodulator, frequency))
        self._input_frequency, self._input_bandwidth = tuner_geometry(self._bounds)

    def reset(self):
        self._bounds = []
        self._input_frequency, self._input_bandwidth = tuner_geometry(self._bounds)

    def load(self, input_signal):
        self._buffer = np.fft.fft(np.asarray(input_signal))

    def _roll(self, channel_index):
        ch = self._bounds[int(channel_index)]
        return int(self._input_frequency - ch.center_frequency), int(ch.bandwidth)

    def run(self, channel_index):
        roll, m = self._roll(channel_index)
        if self._win is None:
            self._win = shifted_window("hann", int(self._input_bandwidth))
        if self._win.shape[0] != self._buffer.shape[0]:
            # scipy.signal.resample's check: the reference caches its window at the first run (tuner.py:155-157)
            raise ValueError("window must have the same length as data")
        tmp = np.roll(self._buffer, roll)
        return resample(tmp, m, window=self._win, domain="freq")

    def run_pruned(self, channel_index):
        roll, m = self._roll(channel_index)
        n = self._buffer.shape[0]
        if self._win_len is None:
            self._win_len = int(self._input_bandwidth)
        if (self._win.shape[0] if self._win is not None else self._win_len) != n:
            raise ValueError("window must have the same length as data")
        if m > n:
            return self.run(channel_index)
        Y = tuner_channel_spectrum(self._buffer, n, roll, m)
        y = np.fft.ifft(Y)
        y *= float(m) / float(n)
        return y
