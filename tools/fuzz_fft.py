#!/usr/bin/env python3
"""One-off regression sweep of the FFT engine (GPU box): random {2,3,5,7}-smooth lengths and batches against numpy.fft.
    python tools/fuzz_fft.py [count] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "radio-core_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from radiocore._internal import hip  # noqa: E402


def smooth(rng, lo=256, hi=4_000_000):
    while True:
        n = 1
        target = int(np.exp(rng.uniform(np.log(lo), np.log(hi))))
        while n < target:
            n *= int(rng.choice([2, 2, 2, 3, 3, 5, 5, 5, 7]))
        if lo <= n <= hi:
            return n


def main(count=150, seed=1):
    lib = hip.lib()
    hip.torch()
    rng = np.random.default_rng(seed)
    worst, skipped = 0.0, 0
    plan = hip.FftPlan()
    for i in range(count):
        n = smooth(rng)
        import ctypes
        if lib.rcfm_fft_describe(n, 0, ctypes.byref(plan)) != 0:
            skipped += 1
            continue
        batch = int(rng.integers(1, 4))
        inverse = bool(rng.integers(0, 2))
        x = (rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))).astype(np.complex64)
        xd = torch.from_numpy(x).cuda()
        yd = torch.empty_like(xd)
        hip.check(lib.rcfm_fft_c2c(n, batch, int(inverse), hip.ptr(xd), hip.ptr(yd), hip.stream()))
        torch.cuda.synchronize()
        want = np.fft.ifft(x.astype(np.complex128), axis=1) * n if inverse else np.fft.fft(x.astype(np.complex128), axis=1)
        err = float(np.max(np.abs(yd.cpu().numpy() - want)) / np.max(np.abs(want)))
        worst = max(worst, err)
        tag = "" if err <= 3e-6 else "   <-- FAIL"
        print("n=%8d batch=%d inv=%d plan=%s err=%.2e%s" % (n, batch, inverse, [plan.passes[t].L for t in range(plan.npass)], err, tag),
              flush=True)
        if err > 3e-6:
            sys.exit(1)
    print("ok: %d lengths, %d outside the engine, worst %.2e" % (count - skipped, skipped, worst))


if __name__ == "__main__":
    main(*(int(a) for a in sys.argv[1:]))
