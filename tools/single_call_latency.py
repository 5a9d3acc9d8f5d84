#!/usr/bin/env python3
"""One demodulator call per buffer (BASELINE configs[1]; the reference times exactly these single calls,
tests/benchmark.py:29-31): synchronous latency and queued cost per call, with the captured launch chain
(RCFM_OPT_GRAPH = 1, the default) and with plain launches (0).  GPU box.

    python tools/single_call_latency.py [reps]
"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radio-core_amd")]
import torch  # noqa: E402

import workloads  # noqa: E402
from radiocore._internal import hip  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    lib = hip.lib()
    hip.torch()
    s = hip.stream()
    print("%-5s %-7s %6s %12s %12s %8s" % ("kind", "B->A", "graph", "sync us", "queued us", "chains"))
    for kind, k in (("WBFM", 2), ("MFM", 1), ("FM", 0)):
        for B, A in ((240000, 48000), (250000, 48000)):
            x = hip.to_device(workloads.single_channel(B, i=0, stereo=(kind == "WBFM")), torch.complex64)
            y = torch.empty(A, 2 if k == 2 else 1, device="cuda")
            for graph in (0, 1):
                h = ctypes.c_void_p()
                hip.check(lib.rcfm_demod_create(k, 1, B, A, ctypes.c_double(75e-6), 0, ctypes.byref(h)))
                hip.check(lib.rcfm_demod_set_option(h, hip.RCFM_OPT_GRAPH, graph))
                for _ in range(10):
                    hip.check(lib.rcfm_demod_run(h, 0, 1, hip.ptr(x), hip.ptr(y), s))
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    lib.rcfm_demod_run(h, 0, 1, hip.ptr(x), hip.ptr(y), s)
                    torch.cuda.synchronize()
                sync = (time.perf_counter() - t0) / reps
                t0 = time.perf_counter()
                for _ in range(reps):
                    lib.rcfm_demod_run(h, 0, 1, hip.ptr(x), hip.ptr(y), s)
                torch.cuda.synchronize()
                queued = (time.perf_counter() - t0) / reps
                v = ctypes.c_int()
                hip.check(lib.rcfm_demod_get_option(h, hip.RCFM_OPT_GRAPH, ctypes.byref(v)))
                print("%-5s %6d->%-5d %3d %12.1f %12.1f %8d" % (kind, B, A, graph, sync * 1e6, queued * 1e6, max(v.value - 1, 0)),
                      flush=True)
                hip.check(lib.rcfm_demod_destroy(h))


if __name__ == "__main__":
    main()
