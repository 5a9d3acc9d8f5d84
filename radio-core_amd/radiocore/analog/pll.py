"""PLL: Hilbert-transform carrier regeneration (reference: radiocore/analog/pll.py:19-58)."""

from radiocore._internal import Injector, hip

__all__ = ["PLL"]


class PLL(Injector):
    """step() keeps the analytic signal of its input; real()/image() return the
    cosine / sine of `mult` times its phase (rcfm_hilbert, rcfm_pll_phase)."""

    def __init__(self, cuda=False):
        self._cuda = cuda
        self._z = None
        super().__init__(self._cuda)

    @property
    def _baseline(self):
        return None if self._z is None else self._result(self._z, self._cuda)

    def step(self, input_sig):
        x = hip.to_device(input_sig, self._torch.float32)
        n = x.shape[0]
        z = hip.empty((n,), self._torch.complex64)
        hip.check(self._lib.rcfm_hilbert(1, n, hip.ptr(x), hip.ptr(z), hip.stream()))
        self._z = z

    def _phase(self, mult, want_imag):
        out = hip.empty(self._z.shape, self._torch.float32)
        hip.check(self._lib.rcfm_pll_phase(hip.ptr(self._z), self._z.numel(), float(mult), want_imag,
                                           hip.ptr(out), hip.stream()))
        return self._result(out, self._cuda)

    def real(self, mult=1.0):
        return self._phase(mult, 0)

    def image(self, mult=1.0):
        return self._phase(mult, 1)
