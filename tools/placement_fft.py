#!/usr/bin/env python3
"""Does the placement of the tuner's own buffers (spectrum X, forward scratch) change the wideband FFT's time?
K tuner handles of cfg4's N in ONE process (all kept alive, so every handle's buffers lie somewhere else), the same input;
rcfm_tuner_load timed per handle.  tools/microbench/pitch_sweep.hip `place` is the tile-copy version of this question."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radio-core_amd")]
import torch  # noqa: E402

from radiocore._internal import hip  # noqa: E402


def main(K=8, N=240_000_000, C=4, B=240000):
    lib = hip.lib()
    hip.torch()
    x = torch.randn(N, 2, device="cuda").view(torch.complex64).reshape(N)
    rolls = (ctypes.c_int64 * C)(*[i * 200000 for i in range(C)])
    bws = (ctypes.c_int32 * C)(*([B] * C))
    tuners = []
    for k in range(K):
        t = ctypes.c_void_p()
        hip.check(lib.rcfm_tuner_create(N, C, rolls, bws, ctypes.byref(t)))
        tuners.append(t)
    s = hip.stream()
    for rep in range(2):
        row = []
        for t in tuners:
            hip.check(lib.rcfm_tuner_load(t, hip.ptr(x), s))
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                hip.check(lib.rcfm_tuner_load(t, hip.ptr(x), s))
            b.record()
            torch.cuda.synchronize()
            row.append(a.elapsed_time(b) / 5)
        print("FFT_N ms per handle:", " ".join("%.3f" % v for v in row), flush=True)


if __name__ == "__main__":
    main(*(int(a) for a in sys.argv[1:]))
