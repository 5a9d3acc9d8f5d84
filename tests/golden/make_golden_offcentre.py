#!/usr/bin/env python3
"""Golden vectors for stations OFF their channel centre (fm.py:60-65's float32 unwrap).

Run in the build container only (needs /root/reference):  python tests/golden/make_golden_offcentre.py

With a carrier offset the unwrapped phase grows to 2 pi * offset over the one-second buffer; the reference unwraps
in float32, whose spacing at 3e4 .. 3e5 rad is 2e-3 .. 3e-2 rad, so its discriminator output carries rounding
noise far above the 1e-4 parity tolerance.  Three vectors per offset pin that behaviour:
  ref_<off>    FM(240000, 48000).run(x) of the reference itself (float32 pipeline)
  truth_<off>  the same mathematics evaluated in float64 (phase steps from arg(x[t] conj x[t-1]), scipy resample
               with the reference's fftshifted Hamming window)
plus the input digests.  tests/ use them to show reference-vs-truth >> 1e-4 and HIP-vs-truth << 1e-4.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True

import numpy as np  # noqa: E402
import scipy.fft  # noqa: E402
import scipy.signal  # noqa: E402

from make_golden import digest, import_reference, save  # noqa: E402
import workloads  # noqa: E402

B, A = 240000, 48000
OFFSETS = (5000, 50000)


def truth(x):
    z = x.astype(np.complex128)
    d = np.zeros(B)
    d[1:] = np.angle(z[1:] * np.conj(z[:-1])) / np.pi
    win = scipy.fft.fftshift(scipy.signal.get_window("hamm", B))
    return scipy.signal.resample(d, A, window=win)


def main():
    rc = import_reference()
    out = {}
    for off in OFFSETS:
        x = workloads.single_channel(B, i=1, offset=off)
        out["in_%d" % off] = np.array(digest(x))
        out["ref_%d" % off] = rc.FM(B, A).run(x)[:, 0]
        out["truth_%d" % off] = truth(x).astype(np.float32)
        t = truth(x)
        r = out["ref_%d" % off].astype(np.float64)
        print("offset %6d Hz: reference vs float64 truth  max|delta| / max|truth| = %.3e" %
              (off, np.max(np.abs(r - t)) / np.max(np.abs(t))))
    save("fm_offcentre", **out)


if __name__ == "__main__":
    main()
