#!/usr/bin/env python3
"""GPU box: HIP-event time of every stage of ONE demodulator call (rcfm_profile_*), for a few batch sizes:
    python tools/stage_times.py [WBFM|MFM|FM] [B] [A] [batch ...]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "radio-core_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
from radiocore._internal import hip  # noqa: E402

lib = hip.lib()
hip.torch()
kind = sys.argv[1] if len(sys.argv) > 1 else "WBFM"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 240000
A = int(sys.argv[3]) if len(sys.argv) > 3 else 48000
batches = [int(a) for a in sys.argv[4:]] or [1, 2, 8, 32, 128]
for T in batches:
    ph = torch.cumsum(torch.randn(T, B, device="cuda") * 0.3, dim=1)
    iq = torch.polar(torch.ones_like(ph), ph).to(torch.complex64).contiguous()
    demod = ctypes.c_void_p()
    hip.check(lib.rcfm_demod_create({"FM": 0, "MFM": 1, "WBFM": 2}[kind], T, B, A, 75e-6, 0, ctypes.byref(demod)))
    ch = 2 if kind == "WBFM" else 1
    audio = torch.empty((T, A, ch), dtype=torch.float32, device="cuda")
    for _ in range(5):
        hip.check(lib.rcfm_demod_run(demod, 0, T, hip.ptr(iq), hip.ptr(audio), hip.stream()))
    torch.cuda.synchronize()
    lib.rcfm_profile_reset()
    lib.rcfm_profile_enable(ctypes.c_uint64((1 << lib.rcfm_profile_stage_count()) - 1))
    reps = 20
    for _ in range(reps):
        hip.check(lib.rcfm_demod_run(demod, 0, T, hip.ptr(iq), hip.ptr(audio), hip.stream()))
    torch.cuda.synchronize()
    prof = bench.read_profile(lib)
    lib.rcfm_profile_enable(ctypes.c_uint64(0))
    row = "  ".join("%s %.1f" % (k, 1e3 * v[1] / reps) for k, v in prof.items() if v[2])
    print("%s %d->%d batch %4d  (us per call)  %s" % (kind, B, A, T, row), flush=True)
    hip.check(lib.rcfm_demod_destroy(demod))
    del iq, ph, audio
