"""GPU: placement control (rcfm_arena_*, radiocore.tools.Arena).  Results never depend on where the workspaces live;
the arena's bookkeeping (bytes handed out, live pieces, refusal to die under live handles) does what rcfm.h says."""
import ctypes
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, have_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not have_gpu(), reason="needs an MI355X")]

N, B, A, C = 1_200_000, 60000, 12000, 6


def _band(rc, arena_ctx=None):
    import workloads
    centres = workloads.channel_grid(C, 150000)
    t = rc.Tuner(cuda=False)
    for f in centres:
        t.add_channel(f, B, rc.WBFM(B, A))
    t.request_bandwidth(float(N))
    x = workloads.wideband(N, t.input_frequency, centres, B, gain=0.4)
    return t, x


def test_arena_handles_give_identical_audio_and_account_for_their_memory():
    import radiocore as rc
    from radiocore.tools import Arena

    plain, x = _band(rc)
    plain.load(x)
    want = plain.run_all()
    want1 = plain.channels()[1].demodulator.run(plain.run(1))

    arena = Arena(256 << 20)                       # one 256 MiB block
    before = arena.stats()
    assert before["reserved_bytes"] == 256 << 20 and before["used_bytes"] == 0 and before["live_pieces"] == 0
    with arena:
        inside, _ = _band(rc)                      # the objects remember the arena; handles are built lazily, inside it
    inside.load(x)
    got = inside.run_all()
    got1 = inside.channels()[1].demodulator.run(inside.run(1))
    assert np.array_equal(got, want) and np.array_equal(got1, want1)
    st = arena.stats()
    # spectrum + scratch of the tuner alone are 2 x 8 N bytes; every piece starts on a 2 MiB boundary
    assert st["used_bytes"] >= 16 * N and st["used_bytes"] % (2 << 20) == 0 and st["live_pieces"] >= 4
    with pytest.raises(RuntimeError, match="still alive"):
        arena.close()
    del inside
    import gc
    gc.collect()
    assert arena.stats()["live_pieces"] == 0
    arena.close()


def test_arena_over_host_owned_memory_puts_two_handle_sets_on_the_same_addresses():
    """rcfm_arena_adopt: the arena lives in a tensor of the host's allocator; a second handle set built over the same
    tensor after the first one is gone gets the same spectrum address (what tools/ab_libs.py relies on)."""
    import torch
    from radiocore._internal import hip
    lib = hip.lib()
    block = torch.empty(128 << 20, dtype=torch.uint8, device="cuda")
    rolls = (ctypes.c_int64 * 2)(1000, -2000)
    bws = (ctypes.c_int32 * 2)(60000, 60000)
    where = []
    for _ in range(2):
        arena, t = ctypes.c_void_p(), ctypes.c_void_p()
        hip.check(lib.rcfm_arena_adopt(hip.ptr(block), ctypes.c_size_t(block.numel()), ctypes.byref(arena)))
        hip.check(lib.rcfm_arena_bind(arena))
        hip.check(lib.rcfm_tuner_create(N, 2, rolls, bws, ctypes.byref(t)))
        hip.check(lib.rcfm_arena_bind(None))
        X = ctypes.c_void_p()
        hip.check(lib.rcfm_tuner_spectrum(t, ctypes.byref(X)))
        where.append(X.value)
        assert block.data_ptr() <= X.value < block.data_ptr() + block.numel()
        assert lib.rcfm_arena_destroy(arena) != 0            # the tuner is alive
        hip.check(lib.rcfm_tuner_destroy(t))
        hip.check(lib.rcfm_arena_destroy(arena))
    assert where[0] == where[1]


def test_window_storage_is_refused_where_the_window_wraps():
    """rcfm_tuner_window_layout: channels next to the band centre read bins on both sides of bin 0 -- no window
    storage; a range clear of the ends gets [halo | nbins | halo], and a handle with a window attached refuses
    rcfm_tuner_load, rcfm_tuner_spectrum and channels outside the range."""
    import torch
    from radiocore._internal import hip
    lib = hip.lib()
    # four channels: two just around the centre (bins wrap), two well above it
    rolls = (ctypes.c_int64 * 4)(20000, -20000, -300000, -380000)
    bws = (ctypes.c_int32 * 4)(60000, 60000, 60000, 60000)
    t = ctypes.c_void_p()
    hip.check(lib.rcfm_tuner_create(N, 4, rolls, bws, ctypes.byref(t)))
    halo, nb = ctypes.c_int64(), ctypes.c_int64()
    assert lib.rcfm_tuner_window_layout(t, 0, 2, ctypes.byref(halo), ctypes.byref(nb)) != 0
    hip.check(lib.rcfm_tuner_window_layout(t, 2, 2, ctypes.byref(halo), ctypes.byref(nb)))
    fb, nb2 = ctypes.c_int64(), ctypes.c_int64()
    hip.check(lib.rcfm_tuner_window(t, 2, 2, ctypes.byref(fb), ctypes.byref(nb2)))
    assert nb.value == nb2.value and 140000 <= nb.value < N // 2 and halo.value >= 30002
    slot = torch.zeros(nb.value + 2 * halo.value, dtype=torch.complex64, device="cuda")
    hip.check(lib.rcfm_tuner_attach_window(t, hip.ptr(slot), 2, 2))
    x = torch.zeros(N, dtype=torch.complex64, device="cuda")
    X = ctypes.c_void_p()
    assert lib.rcfm_tuner_load(t, hip.ptr(x), hip.stream()) != 0
    assert lib.rcfm_tuner_spectrum(t, ctypes.byref(X)) != 0
    assert lib.rcfm_tuner_adopt(t, 0, 2, hip.stream()) != 0          # another range than the window's
    hip.check(lib.rcfm_tuner_adopt(t, 2, 2, hip.stream()))
    out = torch.empty((2, 60000), dtype=torch.complex64, device="cuda")
    hip.check(lib.rcfm_tuner_run(t, 2, 2, hip.ptr(out), hip.stream()))
    assert lib.rcfm_tuner_run(t, 0, 1, hip.ptr(out), hip.stream()) != 0
    hip.check(lib.rcfm_tuner_attach_spectrum(t, None, 0, 0))          # back to the handle's own storage
    hip.check(lib.rcfm_tuner_load(t, hip.ptr(x), hip.stream()))
    hip.check(lib.rcfm_tuner_run(t, 0, 2, hip.ptr(out), hip.stream()))
    torch.cuda.synchronize()
    hip.check(lib.rcfm_tuner_destroy(t))


def test_arena_counts_handles_not_pieces():
    """A handle keeps its arena for life even when it holds no piece of it at the moment: a tuner of small buffers
    only (everything below the 1 MiB arena threshold), and a tuner whose spectrum storage was attached (its own spectrum
    piece dropped).  rcfm_arena_destroy refuses both (RCFM_ERR_STATE), as rcfm.h says."""
    import torch
    from radiocore._internal import hip
    lib = hip.lib()
    arena, small, big = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    hip.check(lib.rcfm_arena_create(ctypes.c_size_t(64 << 20), ctypes.byref(arena)))
    hip.check(lib.rcfm_arena_bind(arena))
    rolls = (ctypes.c_int64 * 1)(100)
    bws = (ctypes.c_int32 * 1)(1000)
    hip.check(lib.rcfm_tuner_create(20000, 1, rolls, bws, ctypes.byref(small)))        # 160 KB spectrum: hipMalloc
    hip.check(lib.rcfm_arena_bind(None))
    r, u, n = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
    hip.check(lib.rcfm_arena_stats(arena, ctypes.byref(r), ctypes.byref(u), ctypes.byref(n)))
    assert n.value == 0                                       # no piece ...
    assert lib.rcfm_arena_destroy(arena) == -5, "a live handle without pieces must keep its arena alive"   # RCFM_ERR_STATE
    assert b"still alive" in lib.rcfm_last_error()
    hip.check(lib.rcfm_tuner_destroy(small))

    hip.check(lib.rcfm_arena_bind(arena))
    rolls = (ctypes.c_int64 * 2)(1000, -2000)
    bws = (ctypes.c_int32 * 2)(60000, 60000)
    hip.check(lib.rcfm_tuner_create(N, 2, rolls, bws, ctypes.byref(big)))
    hip.check(lib.rcfm_arena_bind(None))
    halo, nn = ctypes.c_int64(), ctypes.c_int64()
    hip.check(lib.rcfm_tuner_spectrum_layout(big, ctypes.byref(halo), ctypes.byref(nn)))
    ext = torch.zeros(nn.value + 2 * halo.value, dtype=torch.complex64, device="cuda")
    hip.check(lib.rcfm_arena_stats(arena, None, None, ctypes.byref(n)))
    pieces_before = n.value
    hip.check(lib.rcfm_tuner_attach_spectrum(big, hip.ptr(ext), 0, 0))
    hip.check(lib.rcfm_arena_stats(arena, None, None, ctypes.byref(n)))
    assert n.value == pieces_before - 1                       # the spectrum piece went
    assert lib.rcfm_arena_destroy(arena) == -5
    x = torch.zeros(N, dtype=torch.complex64, device="cuda")
    hip.check(lib.rcfm_tuner_load(big, hip.ptr(x), hip.stream()))     # still runs inside its (living) arena
    torch.cuda.synchronize()
    hip.check(lib.rcfm_tuner_destroy(big))
    hip.check(lib.rcfm_arena_destroy(arena))
    assert lib.rcfm_arena_bind(arena) != 0                    # a destroyed arena cannot be bound again


def test_only_tuner_and_demodulator_handles_draw_from_a_bound_arena():
    """Arena use is opt-in per handle: the function-static scratch of rcfm_fft_c2c, the plan cache of rcfm_hilbert and a
    resampler handle created while an arena is bound all come from hipMalloc -- the arena stays empty and can go."""
    import torch
    from radiocore._internal import hip
    lib = hip.lib()
    arena = ctypes.c_void_p()
    hip.check(lib.rcfm_arena_create(ctypes.c_size_t(32 << 20), ctypes.byref(arena)))
    hip.check(lib.rcfm_arena_bind(arena))
    n = 1 << 20                                                # 8 MiB scratch per transform
    x = torch.randn(n, 2, device="cuda")
    y = torch.empty_like(x)
    hip.check(lib.rcfm_fft_c2c(n, 1, 0, hip.ptr(x), hip.ptr(y), hip.stream()))
    lens = (ctypes.c_int64 * 3)(64, 128, 128)
    hip.check(lib.rcfm_fft_c2c_plan(n, lens, 3, -1, 1, 0, hip.ptr(x), hip.ptr(y), hip.stream()))
    p = torch.randn(4, 240000, device="cuda")
    z = torch.empty(4, 240000, 2, device="cuda")
    hip.check(lib.rcfm_hilbert(4, 240000, hip.ptr(p), hip.ptr(z), hip.stream()))
    rs = ctypes.c_void_p()
    hip.check(lib.rcfm_resampler_create(8, 240000, 48000, 0, ctypes.byref(rs)))
    hip.check(lib.rcfm_arena_bind(None))
    torch.cuda.synchronize()
    u, live = ctypes.c_size_t(), ctypes.c_size_t()
    hip.check(lib.rcfm_arena_stats(arena, None, ctypes.byref(u), ctypes.byref(live)))
    assert u.value == 0 and live.value == 0
    hip.check(lib.rcfm_arena_destroy(arena))                  # ... while the resampler and the static buffers live on
    q = torch.empty(8, 48000, device="cuda")
    hip.check(lib.rcfm_resampler_run(rs, hip.ptr(torch.randn(8, 240000, device="cuda")), hip.ptr(q), hip.stream()))
    hip.check(lib.rcfm_fft_c2c(n, 1, 0, hip.ptr(x), hip.ptr(y), hip.stream()))
    torch.cuda.synchronize()
    hip.check(lib.rcfm_resampler_destroy(rs))


def test_nested_bindings_restore_the_outer_arena():
    from radiocore._internal import hip
    from radiocore.tools import Arena
    lib = hip.lib()
    outer, inner = Arena(8 << 20), Arena(8 << 20)
    rolls = (ctypes.c_int64 * 1)(100)
    bws = (ctypes.c_int32 * 1)(60000)
    handles = []
    with hip.bound(outer):
        with hip.bound(inner):
            t = ctypes.c_void_p()
            hip.check(lib.rcfm_tuner_create(N, 1, rolls, bws, ctypes.byref(t)))
            handles.append(t)
        t = ctypes.c_void_p()
        hip.check(lib.rcfm_tuner_create(N, 1, rolls, bws, ctypes.byref(t)))      # after the inner block: the OUTER arena again
        handles.append(t)
    assert inner.stats()["live_pieces"] >= 2 and outer.stats()["live_pieces"] >= 2
    for t in handles:
        hip.check(lib.rcfm_tuner_destroy(t))
    inner.close()
    outer.close()
