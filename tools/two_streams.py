#!/usr/bin/env python3
"""Experiment: consecutive buffers on alternating HIP streams (two tuner + demod handle sets).
Does overlapping the kernels of buffer i+1 with those of buffer i raise the throughput of cfg4?"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "radio-core_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
from radiocore._internal import hip  # noqa: E402


def main(nstreams, steps=20, config="cfg4"):
    lib = hip.lib()
    hip.torch()
    N, C, B, A, raster, kind = bench.CONFIGS[config]
    ch = 2 if kind == "WBFM" else 1
    x, centres, f_in = bench.synth_wideband_on_device(N, C, B, raster, kind, lib, hip)
    rolls = [int(f_in - f) for f in centres]
    roll_a = (ctypes.c_int64 * C)(*rolls)
    bw_a = (ctypes.c_int32 * C)(*([B] * C))
    sets = []
    for _ in range(nstreams):
        tuner = ctypes.c_void_p()
        hip.check(lib.rcfm_tuner_create(N, C, roll_a, bw_a, ctypes.byref(tuner)))
        demod = ctypes.c_void_p()
        hip.check(lib.rcfm_demod_create({"FM": 0, "MFM": 1, "WBFM": 2}[kind], C, B, A, 75e-6, 0, ctypes.byref(demod)))
        audio = torch.empty((C, A, ch), dtype=torch.float32, device="cuda")
        sets.append((tuner, demod, audio, torch.cuda.Stream()))

    def step(i):
        tuner, demod, audio, st = sets[i % nstreams]
        s = ctypes.c_void_p(st.cuda_stream)
        hip.check(lib.rcfm_tuner_load(tuner, hip.ptr(x), s))
        hip.check(lib.rcfm_pipeline_run(tuner, demod, 0, C, hip.ptr(audio), s))

    for i in range(4):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%s streams=%d: %.3f ms per buffer" % (config, nstreams, 1e3 * dt / steps), flush=True)


if __name__ == "__main__":
    cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg4"
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 2, config=cfg)
