#!/usr/bin/env python3
"""PCIe-inclusive side note for DESIGN.md: host -> device rate of one 1.92 GB wideband buffer
from pageable vs page-locked (Buffer(cuda=True)) memory."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "radio-core_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from radiocore import Buffer  # noqa: E402

N = 240_000_000
dev = torch.empty(N, dtype=torch.complex64, device="cuda")
for name, host in (("pageable numpy", np.zeros(N, np.complex64)), ("pinned Buffer(cuda=True)", Buffer(N, cuda=True).data)):
    src = torch.from_numpy(host)
    if "pinned" in name:
        src = torch.from_numpy(Buffer(N, cuda=True)._owner.numpy().view(np.complex64)) if False else src
    for _ in range(2):
        dev.copy_(src)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        dev.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("%-26s %7.1f ms per 1.92 GB buffer = %6.1f GB/s -> at most %7.0f Msamples/s end to end" %
          (name, dt * 1e3, N * 8 / dt / 1e9, N / dt / 1e6))
