"""Seeded synthetic wideband buffers built ON the GPU (bench.py and the full-size -m gpu tests).

Input generation is not the product: torch operators are used freely here, and the inverse
wideband FFT borrows the library's own forward engine.  Same recipe as workloads.py (stations on
the channel centres, integer-Hz tones, periodic in the 1-second buffer).
"""

import ctypes

import numpy as np
import torch


def synth_wideband_on_device(N, C, B, raster, kind, lib, hip):
    """Seeded synthetic wideband buffer built on the GPU (input generation is not the
    product: torch ops are used freely here).  Stations sit on the channel centres;
    each is an FM signal whose modulation is integer-Hz tones (+ 19 kHz pilot and a
    38 kHz DSB L-R component for WBFM), so it is periodic in the 1-second buffer."""
    dev = "cuda"
    g = torch.Generator(device="cpu").manual_seed(1234)
    centres = [float(int(100e6 + (i - (C - 1) / 2.0) * raster)) for i in range(C)]
    lower = min(centres) - B / 2
    higher = max(centres) + B / 2
    f_in = (lower + higher) / 2
    Xw = torch.zeros(N, dtype=torch.complex64, device=dev)
    t = torch.arange(B, device=dev, dtype=torch.float64) / B
    kk = torch.fft.fftfreq(B, 1.0 / B, device=dev).round().to(torch.int64)
    dev_hz = 75e3 * B / 240000.0 if kind != "FM" else 0.2 * B
    gain = 0.5 / np.sqrt(max(C * B / N, 1.0))
    step = 32
    for c0 in range(0, C, step):
        idx = torch.arange(c0, min(c0 + step, C))
        k = (idx % 89).to(torch.float64).to(dev)[:, None]
        ph = (torch.rand((len(idx), 6), generator=g, dtype=torch.float64) * 2 * np.pi).to(dev)
        two_pi_t = 2 * np.pi * t[None, :]
        L = 0.3 * (torch.sin((300 + 37 * k) * two_pi_t + ph[:, 0:1]) + torch.sin((1000 + 11 * k) * two_pi_t + ph[:, 1:2])
                   + torch.sin((5000 + 3 * k) * two_pi_t + ph[:, 2:3]))
        R = 0.35 * (torch.sin((440 + 29 * k) * two_pi_t + ph[:, 3:4]) + torch.sin((2500 + 7 * k) * two_pi_t + ph[:, 4:5]))
        if kind == "WBFM":
            mpx = 0.3 * (L + R) + 0.1 * torch.sin(19000 * two_pi_t) + 0.3 * (L - R) * torch.sin(38000 * two_pi_t)
        else:
            mpx = 0.5 * L
        phase = 2 * np.pi * dev_hz * torch.cumsum(mpx, dim=1) / B
        s = torch.polar(torch.ones_like(phase), phase).to(torch.complex64)
        S = torch.fft.fft(s, dim=1) * (gain * N / B)
        for j, i in enumerate(idx.tolist()):
            off = int(centres[i] - f_in)
            Xw.index_add_(0, (kk + off) % N, S[j])
        del L, R, mpx, phase, s, S
    # x = IFFT_N(Xw) = conj(FFT_N(conj(Xw))) / N, using the library's own wideband FFT
    roll = (ctypes.c_int64 * 1)(0)
    bw = (ctypes.c_int32 * 1)(min(B, N))
    h = ctypes.c_void_p()
    hip.check(lib.rcfm_tuner_create(N, 1, roll, bw, ctypes.byref(h)))
    Xw = torch.conj_physical(Xw)
    hip.check(lib.rcfm_tuner_load(h, hip.ptr(Xw), hip.stream()))
    spec = ctypes.c_void_p()
    hip.check(lib.rcfm_tuner_spectrum(h, ctypes.byref(spec)))
    x = torch.empty(N, dtype=torch.complex64, device=dev)
    hip.check(lib.rcfm_memcpy_d2d(hip.ptr(x), spec, N * 8, hip.stream()))
    torch.cuda.synchronize()
    hip.check(lib.rcfm_tuner_destroy(h))
    x = torch.conj_physical(x) / N
    noise = torch.randn(N, 2, generator=torch.Generator(device=dev).manual_seed(7), device=dev) * 0.003
    x += torch.view_as_complex(noise)
    del Xw, noise
    torch.cuda.synchronize()
    return x.contiguous(), centres, f_in
