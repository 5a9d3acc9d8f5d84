#!/usr/bin/env python3
"""GPU box, a -DRCFM_DMA_TRACE build: one N = 2.4e8 transform (the launcher prints workgroup 0's phase times)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "radio-core_amd"))
import torch
from radiocore._internal import hip
lib = hip.lib(); hip.torch()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 240_000_000
x = torch.view_as_complex(torch.randn(n, 2, device="cuda"))
y = torch.empty_like(x)
for _ in range(2):
    hip.check(lib.rcfm_fft_c2c(n, 1, 0, hip.ptr(x), hip.ptr(y), hip.stream()))
torch.cuda.synchronize()
