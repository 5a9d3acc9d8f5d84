#!/usr/bin/env python3
"""Experiment (GPU box): the two halves of ONE buffer's channels on two HIP streams.

    python tools/chunk_streams.py [config] [steps]

One wideband FFT on stream 0, then channels [0, C/2) on stream 0 and [C/2, C) on stream 1 -- a second tuner handle
reads the first one's spectrum (rcfm_tuner_attach_spectrum), each half has its own demodulator handle created for
chunks of C/2 channels, so the workspaces together are what one handle set for C channels holds.  Compared with the
plain one-stream step on the same box, same process.  Does filling one kernel's ramps with the other half's kernels
pay inside a single buffer the way two lanes do across buffers?"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radio-core_amd")]
import torch  # noqa: E402

import bench  # noqa: E402
from radiocore._internal import hip  # noqa: E402


def main(config="cfg4", steps=20):
    lib = hip.lib()
    hip.torch()
    N, C, B, A, raster, kind = bench.CONFIGS[config]
    ch = 2 if kind == "WBFM" else 1
    k = {"FM": 0, "MFM": 1, "WBFM": 2}[kind]
    x, centres, f_in = bench.synth_wideband_on_device(N, C, B, raster, kind, lib, hip)
    rolls = (ctypes.c_int64 * C)(*[int(f_in - f) for f in centres])
    bws = (ctypes.c_int32 * C)(*([B] * C))
    half = C // 2

    def handles(chunk):
        t, d = ctypes.c_void_p(), ctypes.c_void_p()
        hip.check(lib.rcfm_tuner_create(N, C, rolls, bws, ctypes.byref(t)))
        hip.check(lib.rcfm_demod_create(k, C, B, A, 75e-6, chunk, ctypes.byref(d)))
        return t, d

    audio = torch.empty((C, A, ch), dtype=torch.float32, device="cuda")
    audio2 = torch.empty_like(audio)
    s0 = torch.cuda.current_stream()
    s1 = torch.cuda.Stream()
    p0, p1 = ctypes.c_void_p(s0.cuda_stream), ctypes.c_void_p(s1.cuda_stream)

    def timed(step):
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s0)
        for _ in range(steps):
            step()
        b.record(s0)
        torch.cuda.synchronize()
        return a.elapsed_time(b) / steps

    # one stream, one handle set
    t, d = handles(0)

    def plain():
        hip.check(lib.rcfm_tuner_load(t, hip.ptr(x), p0))
        hip.check(lib.rcfm_pipeline_run(t, d, 0, C, hip.ptr(audio), p0))

    ms_plain = timed(plain)
    hip.check(lib.rcfm_demod_destroy(d))

    # two streams: the second tuner handle reads the first one's spectrum
    tb, db = handles(half)
    da = ctypes.c_void_p()
    hip.check(lib.rcfm_demod_create(k, C, B, A, 75e-6, half, ctypes.byref(da)))
    X = ctypes.c_void_p()
    halo, n = ctypes.c_int64(), ctypes.c_int64()
    hip.check(lib.rcfm_tuner_spectrum(t, ctypes.byref(X)))
    hip.check(lib.rcfm_tuner_spectrum_layout(t, ctypes.byref(halo), ctypes.byref(n)))
    hip.check(lib.rcfm_tuner_attach_spectrum(tb, ctypes.c_void_p(X.value - 8 * halo.value), 0, C))
    per = A * ch * 4

    def split():
        hip.check(lib.rcfm_tuner_load(t, hip.ptr(x), p0))
        e = torch.cuda.Event()
        e.record(s0)
        s1.wait_event(e)
        hip.check(lib.rcfm_pipeline_run(t, da, 0, half, hip.ptr(audio2), p0))
        hip.check(lib.rcfm_pipeline_run(tb, db, half, C - half, ctypes.c_void_p(hip.ptr(audio2).value + half * per), p1))
        e2 = torch.cuda.Event()
        e2.record(s1)
        s0.wait_event(e2)

    ms_split = timed(split)
    same = bool(torch.equal(audio, audio2))
    print("%s: one stream %.4f ms, two half-channel streams %.4f ms (%+.2f %%); audio bit-identical: %s" %
          (config, ms_plain, ms_split, 100 * (ms_split / ms_plain - 1), same), flush=True)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "cfg4", int(sys.argv[2]) if len(sys.argv) > 2 else 20)
