#!/usr/bin/env python3
"""The reference's own timing harness shapes (tests/benchmark.py:81-110) on the HIP backend.

Same objects, sizes and call pattern as the reference script -- FM/MFM/WBFM 256 000 -> 32 000,
complex Decimate 10 000 000 / 2 500 000 -> 250 000 (host arrays like the reference, and device-resident;
RCFM_FFT=rocfft gives the rocFFT-backed path for comparison), Tuner 10 000 000 with 3 x 250 000-Hz channels
(load + run(0)), 50 timeit iterations, host numpy arrays in and out (so each call pays its PCIe
copies, exactly like the reference's cuda=True numbers would) -- but on non-zero synthetic input:
the reference feeds zeros, which makes WBFM compute 0/0 (pll.py:58).  Prints seconds per call.
"""
import os
import sys
from timeit import timeit

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radio-core_amd")]
import numpy as np  # noqa: E402

import workloads  # noqa: E402
from radiocore import FM, MFM, WBFM, Decimate, Tuner  # noqa: E402

N_ITER = 50


def score(name, fn):
    fn()
    print("     %-24s %.6f s" % (name, timeit(fn, number=N_ITER) / N_ITER), flush=True)


def main():
    B, A = 256000, 32000
    x = workloads.single_channel(B, i=0)
    print("#### FM benchmark (input %d, output %d, HIP)" % (B, A))
    for cls in (WBFM, MFM, FM):
        d = cls(B, A)
        score(cls.__name__, lambda: d.run(x))
    import torch
    xd = torch.from_numpy(x).cuda()
    for cls in (WBFM, MFM, FM):
        d = cls(B, A, cuda=True)

        def resident():
            d.run(xd, numpy_output=False)
            torch.cuda.synchronize()
        score(cls.__name__ + " (device in/out)", resident)
    print("=" * 80)
    for n in (10_000_000, 2_500_000):
        g = np.random.default_rng(n)
        z = (g.standard_normal(n) + 1j * g.standard_normal(n)).astype(np.complex64)
        dec = Decimate(n, 250000)
        print("#### Decimate benchmark (input %d, output 250000, HIP)" % n)
        score("Decimate", lambda: dec.run(z))
        zd = torch.from_numpy(z).cuda()
        decd = Decimate(n, 250000, cuda=True)

        def dec_resident():
            decd.run(zd)
            torch.cuda.synchronize()
        score("Decimate (device in/out)", dec_resident)
    print("=" * 80)
    N = 10_000_000
    tuner = Tuner()
    for f in (94.5e6, 97.5e6, 96.9e6):
        tuner.add_channel(f, 250000, FM)
    tuner.request_bandwidth(N)
    g = np.random.default_rng(1)
    w = (g.standard_normal(N) + 1j * g.standard_normal(N)).astype(np.complex64)
    print("#### Tuner benchmark (input %d, channel 250000, HIP)" % N)

    def tune():
        tuner.load(w)
        tuner.run(0)
    score("Tuner load+run(0)", tune)
    wd = torch.from_numpy(w).cuda()
    tuner2 = Tuner(cuda=True)
    for f in (94.5e6, 97.5e6, 96.9e6):
        tuner2.add_channel(f, 250000, FM)
    tuner2.request_bandwidth(N)

    def tune_dev():
        tuner2.load(wd)
        tuner2.run(0)
        torch.cuda.synchronize()
    score("Tuner (device in/out)", tune_dev)


if __name__ == "__main__":
    main()
