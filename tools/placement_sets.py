#!/usr/bin/env python3
"""K complete handle sets (tuner + demodulator) of one configuration in ONE process, all kept alive: every set's buffers lie
somewhere else (a third argument: each set also gets its own copy of the input).  The same input through each set, timed per set (whole step and per stage): how much of the "box to box"
spread of the bench is where hipMalloc happened to put the workspaces?"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radio-core_amd")]
import torch  # noqa: E402

import bench  # noqa: E402
from radiocore._internal import hip  # noqa: E402


def main(K=6, config="cfg4", clone_input=False):
    lib = hip.lib()
    hip.torch()
    N, C, B, A, raster, kind = bench.CONFIGS[config]
    ch = 2 if kind == "WBFM" else 1
    x, centres, f_in = bench.synth_wideband_on_device(N, C, B, raster, kind, lib, hip)
    rolls = (ctypes.c_int64 * C)(*[int(f_in - f) for f in centres])
    bws = (ctypes.c_int32 * C)(*([B] * C))
    sets = []
    for _ in range(K):
        t, d = ctypes.c_void_p(), ctypes.c_void_p()
        hip.check(lib.rcfm_tuner_create(N, C, rolls, bws, ctypes.byref(t)))
        hip.check(lib.rcfm_tuner_shard(t, 0, C))
        hip.check(lib.rcfm_demod_create({"FM": 0, "MFM": 1, "WBFM": 2}[kind], C, B, A, 75e-6, 0, ctypes.byref(d)))
        sets.append((t, d, torch.empty((C, A, ch), dtype=torch.float32, device="cuda"), x.clone() if clone_input else x))
    s = hip.stream()

    def step(k):
        t, d, audio, xk = sets[k]
        hip.check(lib.rcfm_tuner_load(t, hip.ptr(xk), s))
        hip.check(lib.rcfm_pipeline_run(t, d, 0, C, hip.ptr(audio), s))

    for k in range(K):
        step(k)
    torch.cuda.synchronize()
    for rep in range(2):
        row = []
        for k in range(K):
            step(k)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(8):
                step(k)
            b.record()
            torch.cuda.synchronize()
            row.append(a.elapsed_time(b) / 8)
        print("%s ms per step, per handle set:" % config, " ".join("%.3f" % v for v in row), flush=True)
    # per stage for the slowest and the fastest set
    order = sorted(range(K), key=lambda k: row[k])
    for k in (order[0], order[-1]):
        lib.rcfm_profile_reset()
        lib.rcfm_profile_enable(ctypes.c_uint64((1 << lib.rcfm_profile_stage_count()) - 1))
        for _ in range(4):
            step(k)
        torch.cuda.synchronize()
        prof = bench.read_profile(lib)
        lib.rcfm_profile_enable(ctypes.c_uint64(0))
        print("set %d (%.3f ms):" % (k, row[k]), " ".join("%s %.3f" % (n, v[1] / 4) for n, v in sorted(prof.items(), key=lambda kv: -kv[1][1]) if v[1] > 0))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 6, sys.argv[2] if len(sys.argv) > 2 else "cfg4", len(sys.argv) > 3)
