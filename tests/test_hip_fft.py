"""GPU: the hand-written multi-pass FFT against numpy.fft (float32 tolerance) and
against rocFFT on the lengths of the hot path."""

import ctypes

import numpy as np
import pytest

from conftest import have_gpu, rel_err

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not have_gpu(), reason="needs an MI355X")]


def _run(n, batch, inverse, x, rocfft=False, in_place=False):
    import torch
    from radiocore._internal import hip
    lib = hip.lib()
    xd = hip.to_device(x, torch.complex64)
    yd = xd if in_place else hip.empty(xd.shape, torch.complex64)
    fn = lib.rcfm_fft_c2c_rocfft if rocfft else lib.rcfm_fft_c2c
    hip.check(fn(n, batch, int(inverse), hip.ptr(xd), hip.ptr(yd), hip.stream()))
    torch.cuda.synchronize()
    return yd.cpu().numpy()


@pytest.mark.parametrize("n,batch", [(256, 3), (1000, 2), (4096, 5), (6000, 1), (12500, 4), (8000, 3),
                                     (48000, 3), (60000, 2), (240000, 3), (256000, 1), (600000, 1),
                                     (10_000_000, 1)])
@pytest.mark.parametrize("inverse", [False, True])
def test_engine_matches_numpy(n, batch, inverse):
    r = np.random.default_rng(n + batch)
    x = (r.standard_normal((batch, n)) + 1j * r.standard_normal((batch, n))).astype(np.complex64)
    want = np.fft.ifft(x.astype(np.complex128), axis=1) * n if inverse else np.fft.fft(x.astype(np.complex128), axis=1)
    got = _run(n, batch, inverse, x)
    assert rel_err(got, want.astype(np.complex64)) <= 2e-6
    if n <= 60000:
        assert rel_err(_run(n, batch, inverse, x, in_place=True), want.astype(np.complex64)) <= 2e-6


@pytest.mark.parametrize("n,batch", [(44100, 3), (88200, 2), (22050, 1), (7 * 4096, 2), (33600, 2), (7 ** 4 * 16, 1), (441000, 1)])
@pytest.mark.parametrize("inverse", [False, True])
def test_lengths_with_a_factor_seven(n, batch, inverse):
    """The radix-7 butterfly of the generic tile kernel (k_fft_pass): 44 100 = 210 x 210 and its neighbours, 7^4 x 16
    (radix 7 in consecutive stages), a three-pass length -- against numpy, forward and inverse, out of place and in place."""
    r = np.random.default_rng(n + batch)
    x = (r.standard_normal((batch, n)) + 1j * r.standard_normal((batch, n))).astype(np.complex64)
    want = np.fft.ifft(x.astype(np.complex128), axis=1) * n if inverse else np.fft.fft(x.astype(np.complex128), axis=1)
    assert rel_err(_run(n, batch, inverse, x), want.astype(np.complex64)) <= 2e-6
    if n <= 88200:
        assert rel_err(_run(n, batch, inverse, x, in_place=True), want.astype(np.complex64)) <= 2e-6


@pytest.mark.parametrize("n", [24000, 49152, 96000, 144000, 204800])
def test_every_specialised_tile_length(n):
    """The remaining tile lengths with a compile-time kernel (150/160, 192/256, 300/320, 375/384, 400/512),
    each as the strided pass of one plan and the rows pass of its transpose."""
    r = np.random.default_rng(n)
    x = (r.standard_normal((2, n)) + 1j * r.standard_normal((2, n))).astype(np.complex64)
    want = np.fft.fft(x.astype(np.complex128), axis=1).astype(np.complex64)
    assert rel_err(_run(n, 2, False, x), want) <= 2e-6
    back = _run(n, 2, True, want)
    assert rel_err(back, x * n) <= 4e-6


def test_engine_impulse_and_tone_are_exact_enough():
    """Transpose-detecting inputs: a shifted impulse and a single off-centre tone."""
    n = 240000
    x = np.zeros((1, n), np.complex64)
    x[0, 12345] = 1.0
    got = _run(n, 1, False, x)[0]
    k = np.arange(n)
    want = np.exp(-2j * np.pi * ((12345 * k) % n) / n)
    assert np.max(np.abs(got - want)) <= 3e-6
    t = np.exp(2j * np.pi * 54321 * np.arange(n) / n).astype(np.complex64)[None]
    got = _run(n, 1, False, t)[0]
    assert abs(got[54321] - n) <= 2e-6 * n
    got[54321] = 0
    assert np.max(np.abs(got)) <= 3e-6 * n


@pytest.mark.parametrize("n", [100_000_000, 240_000_000])
def test_engine_vs_rocfft_full_wideband(n):
    """The wideband lengths of cfg5 (N = 1e8) and cfg4 (N = 2.4e8 = 600 x 625 x 640, three big-tile passes):
    engine and rocFFT agree bin for bin, and the engine's output satisfies Parseval."""
    import torch
    from radiocore._internal import hip
    lib = hip.lib()
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.view_as_complex(torch.randn(n, 2, generator=g, device="cuda"))
    a = torch.empty_like(x)
    b = torch.empty_like(x)
    hip.check(lib.rcfm_fft_c2c(n, 1, 0, hip.ptr(x), hip.ptr(a), hip.stream()))
    hip.check(lib.rcfm_fft_c2c_rocfft(n, 1, 0, hip.ptr(x), hip.ptr(b), hip.stream()))
    torch.cuda.synchronize()
    peak = float(torch.max(torch.abs(b)))
    err = float(torch.max(torch.abs(a - b)))
    assert err <= 2e-5 * peak, (err, peak)
    # Parseval on the engine's output
    e_t = float(torch.sum(x.real.double() ** 2 + x.imag.double() ** 2))
    e_f = float(torch.sum(a.real.double() ** 2 + a.imag.double() ** 2)) / n
    assert abs(e_f - e_t) <= 1e-5 * e_t


@pytest.mark.parametrize("n", [1 << 26, 240_000_000])
def test_in_place_at_a_length_whose_plan_hands_over_tile_blocked(n):
    """Three-pass plans beyond the Infinity Cache hand the first pass's output to the second one tile-blocked: the second
    pass then reads another address set than it writes and may not run in place.  With in == out (rcfm.h allows it) the
    engine runs the PLAIN layout of the same plan: same arithmetic, so the result equals the out-of-place one bit for
    bit (round 5 ran the blocked pass in place here and returned wrong spectra)."""
    import torch
    from radiocore._internal import hip
    lib = hip.lib()
    plan = hip.FftPlan()
    hip.check(lib.rcfm_fft_describe(n, 0, ctypes.byref(plan)))
    assert plan.npass == 3 and plan.passes[0].out_t != 0 and plan.passes[1].in_t != 0, "not a blocked plan any more"
    g = torch.Generator(device="cuda").manual_seed(n & 0xffff)
    x = torch.view_as_complex(torch.randn(n, 2, generator=g, device="cuda"))
    a = torch.empty_like(x)
    hip.check(lib.rcfm_fft_c2c(n, 1, 0, hip.ptr(x), hip.ptr(a), hip.stream()))
    b = x.clone()
    hip.check(lib.rcfm_fft_c2c(n, 1, 0, hip.ptr(b), hip.ptr(b), hip.stream()))
    torch.cuda.synchronize()
    assert torch.equal(torch.view_as_real(a), torch.view_as_real(b))
    # and both are the transform: Parseval + a few bins against a float64 sum over a slice-wise DFT
    e_t = float(torch.sum(x.real.double() ** 2 + x.imag.double() ** 2))
    e_f = float(torch.sum(a.real.double() ** 2 + a.imag.double() ** 2)) / n
    assert abs(e_f - e_t) <= 1e-5 * e_t
    t = torch.arange(n, device="cuda", dtype=torch.float64)
    for k in (0, 1, 12345, n // 2 + 7, n - 1):
        ph = -2.0 * np.pi * torch.remainder(t * k, n) / n
        want = complex(float(torch.sum(x.real.double() * torch.cos(ph) - x.imag.double() * torch.sin(ph))),
                       float(torch.sum(x.real.double() * torch.sin(ph) + x.imag.double() * torch.cos(ph))))
        got = complex(a[k].item())
        assert abs(got - want) <= 2e-5 * np.sqrt(e_t), (k, got, want)


def test_engine_vs_rocfft_four_passes_at_a_billion_points():
    """N = 10^9 (a 1 GSPS one-second buffer, 8 GB): no three-pass plan exists, the engine runs four passes with
    32-bit point offsets close to their range.  Compared with rocFFT in slices (24 GB of device memory in all)."""
    import torch
    from radiocore._internal import hip
    lib = hip.lib()
    n = 1_000_000_000
    g = torch.Generator(device="cuda").manual_seed(4)
    x = torch.view_as_complex(torch.randn(n, 2, generator=g, device="cuda"))
    a = torch.empty_like(x)
    b = torch.empty_like(x)
    hip.check(lib.rcfm_fft_c2c(n, 1, 0, hip.ptr(x), hip.ptr(a), hip.stream()))
    hip.check(lib.rcfm_fft_c2c_rocfft(n, 1, 0, hip.ptr(x), hip.ptr(b), hip.stream()))
    torch.cuda.synchronize()
    del x
    err = peak = 0.0
    for lo in range(0, n, 1 << 26):
        err = max(err, float(torch.max(torch.abs(a[lo:lo + (1 << 26)] - b[lo:lo + (1 << 26)]))))
        peak = max(peak, float(torch.max(torch.abs(b[lo:lo + (1 << 26)]))))
    del a, b
    torch.cuda.empty_cache()
    assert err <= 2e-5 * peak, (err, peak)


def _run_plan(n, plan, batch, inverse, x, layout=-1):
    import ctypes
    import torch
    from radiocore._internal import hip
    lib = hip.lib()
    xd = hip.to_device(x, torch.complex64)
    yd = hip.empty(xd.shape, torch.complex64)
    lens = (ctypes.c_int64 * len(plan))(*plan)
    hip.check(lib.rcfm_fft_c2c_plan(n, lens, len(plan), layout, batch, int(inverse), hip.ptr(xd), hip.ptr(yd), hip.stream()))
    torch.cuda.synchronize()
    return yd.cpu().numpy()


@pytest.mark.parametrize("n,plan", [(384000, (640, 600)), (375000, (600, 625)), (400000, (625, 640))])
def test_big_tiles_in_both_roles(n, plan):
    """The 600 / 625 / 640-point tiles (two 1024-thread workgroups per CU, twiddles as powers of one global table
    entry, XOR-swizzled rows image) in the roles the hot-path plans do not use them in: the planner's wideband plans
    run 600 and 625 strided and 640 as rows; two-pass plans given through rcfm_fft_c2c_plan run 640 and 625 strided and
    600, 625, 640 as rows."""
    r = np.random.default_rng(n)
    x = (r.standard_normal((2, n)) + 1j * r.standard_normal((2, n))).astype(np.complex64)
    want = np.fft.fft(x.astype(np.complex128), axis=1).astype(np.complex64)
    assert rel_err(_run_plan(n, plan, 2, False, x), want) <= 2e-6
    assert rel_err(_run_plan(n, plan, 2, True, want), x * n) <= 4e-6


@pytest.mark.parametrize("n,plan,batch", [(960000, (80, 100, 120), 2), (1228800, (128, 80, 120), 1), (9600 * 25, (16, 25, 600), 3)])
def test_padded_rows_layout_runs(n, plan, batch):
    """rcfm_fft_c2c_plan layout 2 at sizes numpy checks in a moment: n_3 = 8 mod 16 (120, 600), the first pass over partial
    last tiles and the flat XCD-aware launch, forward and inverse, batched."""
    r = np.random.default_rng(n)
    x = (r.standard_normal((batch, n)) + 1j * r.standard_normal((batch, n))).astype(np.complex64)
    want = np.fft.fft(x.astype(np.complex128), axis=1).astype(np.complex64)
    assert rel_err(_run_plan(n, plan, batch, False, x, layout=2), want) <= 2e-6
    assert rel_err(_run_plan(n, plan, batch, True, want, layout=2), x * n) <= 4e-6


def test_aligned_plan_of_the_wideband_transform():
    """N = 2.4e8: the tuner runs 640 x 625 x 600 in the padded-rows layout (every store aligned) where the default plan
    600 x 625 x 640 stores half of its last pass's segments across two lines.  Same transform: both against each other
    bin for bin, through the FFT entry points and through rcfm_tuner_load with the option on and off (halos included)."""
    import torch
    from radiocore._internal import hip
    lib = hip.lib()
    n = 240_000_000
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.view_as_complex(torch.randn(n, 2, generator=g, device="cuda"))
    a, b = torch.empty_like(x), torch.empty_like(x)
    hip.check(lib.rcfm_fft_c2c(n, 1, 0, hip.ptr(x), hip.ptr(a), hip.stream()))
    lens = (ctypes.c_int64 * 3)(640, 625, 600)
    hip.check(lib.rcfm_fft_c2c_plan(n, lens, 3, 2, 1, 0, hip.ptr(x), hip.ptr(b), hip.stream()))
    torch.cuda.synchronize()
    peak = float(torch.max(torch.abs(a)))
    assert float(torch.max(torch.abs(a - b))) <= 2e-5 * peak
    del b
    rolls = (ctypes.c_int64 * 3)(0, 1_000_000, -119_000_000)       # a channel across bin 0, one inside, one at the far end
    bws = (ctypes.c_int32 * 3)(240000, 240000, 240000)
    spectra = []
    for aligned in (1, 0):
        t = ctypes.c_void_p()
        hip.check(lib.rcfm_tuner_create(n, 3, rolls, bws, ctypes.byref(t)))
        hip.check(lib.rcfm_tuner_set_option(t, hip.RCFM_TUNER_OPT_ALIGNED_PLAN, aligned))
        hip.check(lib.rcfm_tuner_load(t, hip.ptr(x), hip.stream()))
        X = ctypes.c_void_p()
        halo, nn = ctypes.c_int64(), ctypes.c_int64()
        hip.check(lib.rcfm_tuner_spectrum(t, ctypes.byref(X)))
        hip.check(lib.rcfm_tuner_spectrum_layout(t, ctypes.byref(halo), ctypes.byref(nn)))
        out = torch.empty(3, 240000, dtype=torch.complex64, device="cuda")
        hip.check(lib.rcfm_tuner_run(t, 0, 3, hip.ptr(out), hip.stream()))
        torch.cuda.synchronize()
        spectra.append(out)
        hip.check(lib.rcfm_tuner_destroy(t))
    pk = float(torch.max(torch.abs(spectra[1])))
    assert pk > 0 and float(torch.max(torch.abs(spectra[0] - spectra[1]))) <= 2e-5 * pk


def test_plan_entry_refuses_lengths_that_do_not_multiply_to_n():
    import ctypes
    import torch
    from radiocore._internal import hip
    lib = hip.lib()
    x = hip.empty((240000,), torch.complex64)
    lens = (ctypes.c_int64 * 2)(480, 480)
    assert lib.rcfm_fft_c2c_plan(240000, lens, 2, -1, 1, 0, hip.ptr(x), hip.ptr(x), hip.stream()) != 0
