#!/usr/bin/env python3
"""GPU box: the LDS-DMA streaming passes (k_fft_tile_dma) against torch.fft on forced plans, then timings.
Run once per setting of RCFM_FFT_DMA (the switch is read once per process):
    RCFM_FFT_DMA=1 python tools/dma_check.py ; RCFM_FFT_DMA=0 python tools/dma_check.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "radio-core_amd"))
import torch  # noqa: E402

from radiocore._internal import hip  # noqa: E402


def run(lib, n, batch, inverse, x):
    y = torch.empty_like(x)
    hip.check(lib.rcfm_fft_c2c(n, batch, int(inverse), hip.ptr(x), hip.ptr(y), hip.stream()))
    torch.cuda.synchronize()
    return y


def main():
    lib = hip.lib()
    hip.torch()
    print("RCFM_FFT_DMA =", os.environ.get("RCFM_FFT_DMA"), flush=True)
    ok = True
    for n, plan, batch in ((384000, "600,640", 16), (400000, "625,640", 16), (375000, "600,625", 24)):
        os.environ["RCFM_FFT_FORCE"] = plan
        g = torch.Generator(device="cuda").manual_seed(n)
        x = torch.view_as_complex(torch.randn(batch, n, 2, generator=g, device="cuda"))
        for inverse in (False, True):
            want = torch.fft.ifft(x, dim=1) * n if inverse else torch.fft.fft(x, dim=1)
            got = run(lib, n, batch, inverse, x.reshape(-1)).reshape(batch, n)
            err = float(torch.max(torch.abs(got - want))) / float(torch.max(torch.abs(want)))
            print("n=%d plan=%s batch=%d inverse=%d  rel err %.2e" % (n, plan, batch, inverse, err), flush=True)
            ok = ok and err < 1e-5
    os.environ.pop("RCFM_FFT_FORCE")
    for n in (240_000_000, 100_000_000):
        g = torch.Generator(device="cuda").manual_seed(3)
        x = torch.view_as_complex(torch.randn(n, 2, generator=g, device="cuda"))
        a = run(lib, n, 1, False, x)
        b = torch.empty_like(x)
        hip.check(lib.rcfm_fft_c2c_rocfft(n, 1, 0, hip.ptr(x), hip.ptr(b), hip.stream()))
        torch.cuda.synchronize()
        err = float(torch.max(torch.abs(a - b))) / float(torch.max(torch.abs(b)))
        print("n=%d vs rocFFT rel err %.2e" % (n, err), flush=True)
        ok = ok and err < 2e-5
        del b
        for rep in range(3):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                hip.check(lib.rcfm_fft_c2c(n, 1, 0, hip.ptr(x), hip.ptr(a), hip.stream()))
            e.record()
            torch.cuda.synchronize()
            print("n=%d  %.3f ms" % (n, s.elapsed_time(e) / 10), flush=True)
        del x, a
    print("OK" if ok else "FAILED")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
