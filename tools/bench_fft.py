#!/usr/bin/env python3
"""A/B timing of librcfm's FFT engine vs rocFFT on the hot-path lengths (GPU box).

    python tools/bench_fft.py [n[:batch] ...] [--plan f1,f2[,f3] [--layout -1|0|1|2]]
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "radio-core_amd"))
import torch  # noqa: E402

from radiocore._internal import hip  # noqa: E402


def time_fn(fn, reps):
    fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    lib = hip.lib()
    hip.torch()
    cases = [(240000, 16, 20), (240000, 64, 10), (240000, 256, 5), (48000, 64, 20), (12500, 1024, 10),
             (10_000_000, 1, 10), (100_000_000, 1, 5), (240_000_000, 1, 5)]
    plan = None
    argv = sys.argv[1:]
    if "--plan" in argv:     # pass lengths for every length on the command line (rcfm_fft_c2c_plan): plan sweeps
        k = argv.index("--plan")
        plan = [int(v) for v in argv[k + 1].split(",")]
        del argv[k:k + 2]
    layout = -1
    if "--layout" in argv:   # with --plan: 0 plain, 1 tile-blocked hand-over, 2 padded rows (rcfm_tools.h)
        k = argv.index("--layout")
        layout = int(argv[k + 1])
        del argv[k:k + 2]
    if argv:   # lengths from the command line (batch 1 unless n:batch)
        cases = []
        for a in argv:
            n, _, b = a.partition(":")
            cases.append((int(n), int(b) if b else 1, 5 if int(n) > 20_000_000 else 50))
    for n, batch, reps in cases:
        x = torch.view_as_complex(torch.randn(batch * n, 2, device="cuda"))
        y = torch.empty_like(x)
        if plan is not None:
            lens = (ctypes.c_int64 * len(plan))(*plan)
            runs = (("engine", lambda: lib.rcfm_fft_c2c_plan(n, lens, len(plan), layout, batch, 0, hip.ptr(x), hip.ptr(y), hip.stream())),)
        else:
            runs = (("engine", lambda: lib.rcfm_fft_c2c(n, batch, 0, hip.ptr(x), hip.ptr(y), hip.stream())),
                    ("rocfft", lambda: lib.rcfm_fft_c2c_rocfft(n, batch, 0, hip.ptr(x), hip.ptr(y), hip.stream())))
        for name, fn in runs:
            ms = time_fn(lambda: hip.check(fn()), reps)
            gb = 16.0 * n * batch / 1e9
            print("n=%-10d batch=%-5d %-7s %9.3f ms   %7.2f TB/s algorithmic (16 B/point)" %
                  (n, batch, name, ms, gb / ms), flush=True)
        del x, y


if __name__ == "__main__":
    main()
