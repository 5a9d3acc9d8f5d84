#!/usr/bin/env python3
"""Experiment: ONE buffer, its channels on two streams.  The wideband FFT runs on stream A; tuner B reads the same spectrum
(rcfm_tuner_attach_spectrum on A's storage) and runs the upper half of the channels on stream B while A runs the lower half.
Does intra-buffer overlap shorten a single buffer (cfg3: 64 channels, launches of a few tiles per CU)?"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radio-core_amd")]
import torch  # noqa: E402

import bench  # noqa: E402
from radiocore._internal import hip  # noqa: E402


def main(config="cfg3", parts=2, steps=40):
    lib = hip.lib()
    hip.torch()
    N, C, B, A, raster, kind = bench.CONFIGS[config]
    ch = 2 if kind == "WBFM" else 1
    kid = {"FM": 0, "MFM": 1, "WBFM": 2}[kind]
    x, centres, f_in = bench.synth_wideband_on_device(N, C, B, raster, kind, lib, hip)
    rolls = (ctypes.c_int64 * C)(*[int(f_in - f) for f in centres])
    bws = (ctypes.c_int32 * C)(*([B] * C))
    tuners, demods, streams = [], [], []
    for k in range(parts):
        t, d = ctypes.c_void_p(), ctypes.c_void_p()
        hip.check(lib.rcfm_tuner_create(N, C, rolls, bws, ctypes.byref(t)))
        hip.check(lib.rcfm_demod_create(kid, C, B, A, 75e-6, 0, ctypes.byref(d)))
        tuners.append(t); demods.append(d); streams.append(torch.cuda.Stream())
    halo, nn = ctypes.c_int64(), ctypes.c_int64()
    hip.check(lib.rcfm_tuner_spectrum_layout(tuners[0], ctypes.byref(halo), ctypes.byref(nn)))
    X = ctypes.c_void_p()
    hip.check(lib.rcfm_tuner_spectrum(tuners[0], ctypes.byref(X)))
    storage = ctypes.c_void_p(X.value - 8 * halo.value)
    for k in range(1, parts):
        hip.check(lib.rcfm_tuner_attach_spectrum(tuners[k], storage, 0, C))
    audio = torch.empty((C, A, ch), dtype=torch.float32, device="cuda")
    bounds = [(k * C // parts, (k + 1) * C // parts) for k in range(parts)]

    def step(split):
        s0 = ctypes.c_void_p(streams[0].cuda_stream)
        hip.check(lib.rcfm_tuner_load(tuners[0], hip.ptr(x), s0))
        if not split:
            hip.check(lib.rcfm_pipeline_run(tuners[0], demods[0], 0, C, hip.ptr(audio), s0))
            return
        ev = streams[0].record_event()
        for k, (lo, hi) in enumerate(bounds):
            if k:
                streams[k].wait_event(ev)
            out = ctypes.c_void_p(audio.data_ptr() + 4 * lo * A * ch)
            hip.check(lib.rcfm_pipeline_run(tuners[k], demods[k], lo, hi - lo, out, ctypes.c_void_p(streams[k].cuda_stream)))
        for k in range(1, parts):
            streams[0].wait_stream(streams[k])

    for split in (False, True, False, True):
        for _ in range(5):
            step(split)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(split)
        torch.cuda.synchronize()
        print("%s %s: %.4f ms per buffer" % (config, "channels on %d streams" % parts if split else "one stream", 1e3 * (time.perf_counter() - t0) / steps), flush=True)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "cfg3", int(sys.argv[2]) if len(sys.argv) > 2 else 2)
