#!/usr/bin/env python3
"""GPU box: time of the N-point forward transform (default 2.4e8) for the library in RCFM_LIB, 3 x 10 runs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "radio-core_amd"))
import torch
from radiocore._internal import hip
lib = hip.lib(); hip.torch()
for n in [int(a) for a in sys.argv[1:]] or [240_000_000]:
    x = torch.view_as_complex(torch.randn(n, 2, device="cuda"))
    y = torch.empty_like(x)
    hip.check(lib.rcfm_fft_c2c(n, 1, 0, hip.ptr(x), hip.ptr(y), hip.stream()))
    out = []
    for rep in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            hip.check(lib.rcfm_fft_c2c(n, 1, 0, hip.ptr(x), hip.ptr(y), hip.stream()))
        e.record()
        torch.cuda.synchronize()
        out.append("%.3f" % (s.elapsed_time(e) / 10))
    print("%-28s DMA=%s n=%d  ms: %s" % (os.environ.get("RCFM_LIB", "default"), os.environ.get("RCFM_FFT_DMA", "-"), n, " ".join(out)), flush=True)
    del x, y
