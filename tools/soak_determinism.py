#!/usr/bin/env python3
"""Soak (GPU box): the cfg4 pipeline on one wideband buffer, `reps` times from a reset state; every repetition must
be bit-identical to the first (kernels racing on workspace, uninitialised reads or order-dependent atomics would show
up here).  Also a run of `reps` consecutive buffers with state carry whose final state must equal a second such run.
    python tools/soak_determinism.py [reps] [config]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radio-core_amd")]
import torch  # noqa: E402

import bench  # noqa: E402
from radiocore._internal import hip  # noqa: E402
from workloads_device import synth_wideband_on_device  # noqa: E402


def main(reps=100, config="cfg4"):
    lib = hip.lib()
    hip.torch()
    N, C, B, A, raster, kind = bench.CONFIGS[config]
    ch = 2 if kind == "WBFM" else 1
    x, centres, f_in = synth_wideband_on_device(N, C, B, raster, kind, lib, hip)
    rolls = (ctypes.c_int64 * C)(*[int(f_in - f) for f in centres])
    bws = (ctypes.c_int32 * C)(*([B] * C))
    tuner, demod = ctypes.c_void_p(), ctypes.c_void_p()
    hip.check(lib.rcfm_tuner_create(N, C, rolls, bws, ctypes.byref(tuner)))
    hip.check(lib.rcfm_demod_create({"FM": 0, "MFM": 1, "WBFM": 2}[kind], C, B, A, 75e-6, 0, ctypes.byref(demod)))
    out = torch.empty((C, A, ch), dtype=torch.float32, device="cuda")

    def step():
        hip.check(lib.rcfm_tuner_load(tuner, hip.ptr(x), hip.stream()))
        hip.check(lib.rcfm_pipeline_run(tuner, demod, 0, C, hip.ptr(out), hip.stream()))

    first = None
    for i in range(reps):
        hip.check(lib.rcfm_demod_reset_state(demod, hip.stream()))
        step()
        if first is None:
            first = out.clone()
            assert torch.isfinite(first).all()
        elif not torch.equal(out, first):
            bad = int((out != first).sum())
            print("repetition %d differs from the first in %d values" % (i, bad))
            sys.exit(1)
    finals = []
    for _ in range(2):
        hip.check(lib.rcfm_demod_reset_state(demod, hip.stream()))
        for i in range(reps):
            step()
        finals.append(out.clone())
    assert torch.equal(finals[0], finals[1]), "state-carrying runs differ"
    assert kind == "FM" or not torch.equal(finals[0], first), "the de-emphasis state had no effect"
    print("ok: %s, %d repetitions bit-identical; %d-buffer state-carrying runs bit-identical" % (config, reps, reps))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 100, sys.argv[2] if len(sys.argv) > 2 else "cfg4")
