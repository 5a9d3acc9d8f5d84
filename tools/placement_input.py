#!/usr/bin/env python3
"""The other half of tools/placement_fft.py: ONE tuner handle, K copies of the same input at different addresses (all alive):
how much of the wideband FFT's spread is the placement of the caller's buffer?"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radio-core_amd")]
import torch  # noqa: E402

from radiocore._internal import hip  # noqa: E402


def main(K=12, N=240_000_000, C=4, B=240000):
    lib = hip.lib()
    hip.torch()
    x0 = torch.randn(N, 2, device="cuda").view(torch.complex64).reshape(N)
    xs = [x0] + [x0.clone() for _ in range(K - 1)]
    rolls = (ctypes.c_int64 * C)(*[i * 200000 for i in range(C)])
    bws = (ctypes.c_int32 * C)(*([B] * C))
    t = ctypes.c_void_p()
    hip.check(lib.rcfm_tuner_create(N, C, rolls, bws, ctypes.byref(t)))
    s = hip.stream()
    for rep in range(2):
        row = []
        for x in xs:
            hip.check(lib.rcfm_tuner_load(t, hip.ptr(x), s))
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                hip.check(lib.rcfm_tuner_load(t, hip.ptr(x), s))
            b.record()
            torch.cuda.synchronize()
            row.append(a.elapsed_time(b) / 5)
        print("FFT_N ms per input copy:", " ".join("%.3f" % v for v in row), flush=True)


if __name__ == "__main__":
    main(*(int(a) for a in sys.argv[1:]))
