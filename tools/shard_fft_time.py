#!/usr/bin/env python3
"""Time of the wideband FFT (cfg4) when the tuner is sharded to 1/G of the channels (rcfm_tuner_shard)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "radio-core_amd"))
import torch, bench
from radiocore._internal import hip
lib = hip.lib(); hip.torch()
N, C, B, A, raster, kind = bench.CONFIGS["cfg4"]
x = torch.view_as_complex(torch.randn(N, 2, device="cuda"))
centres = [float(int(100e6 + (i - (C - 1) / 2.0) * raster)) for i in range(C)]
f_in = (min(centres) - B / 2 + max(centres) + B / 2) / 2
rolls = (ctypes.c_int64 * C)(*[int(f_in - f) for f in centres]); bws = (ctypes.c_int32 * C)(*([B] * C))
t = ctypes.c_void_p(); hip.check(lib.rcfm_tuner_create(N, C, rolls, bws, ctypes.byref(t)))
for G in (1, 2, 4, 8):
    hip.check(lib.rcfm_tuner_shard(t, 0, C // G))
    for _ in range(2): hip.check(lib.rcfm_tuner_load(t, hip.ptr(x), hip.stream()))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): hip.check(lib.rcfm_tuner_load(t, hip.ptr(x), hip.stream()))
    torch.cuda.synchronize(); print("shard 1/%d: %.3f ms" % (G, (time.perf_counter() - t0) * 100))
