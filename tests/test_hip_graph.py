"""GPU: one channel per call replays a captured hipGraph (RCFM_OPT_GRAPH, include/rcfm.h) -- the reference's
per-channel `demodulator.run(x)` (fm.py:46, mfm.py:51, wbfm.py:66; timed as single calls by tests/benchmark.py:29-31).
Same kernels with the same arguments: the audio and the de-emphasis state are bit-identical to plain launches."""
import ctypes

import numpy as np
import pytest

from conftest import have_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not have_gpu(), reason="needs an MI355X")]

KINDS = {"FM": 0, "MFM": 1, "WBFM": 2}


def _handle(lib, hip, kind, B, A, graph):
    h = ctypes.c_void_p()
    hip.check(lib.rcfm_demod_create(KINDS[kind], 1, B, A, ctypes.c_double(75e-6), 0, ctypes.byref(h)))
    hip.check(lib.rcfm_demod_set_option(h, hip.RCFM_OPT_GRAPH, graph))
    return h


def _graphs(lib, hip, h):
    v = ctypes.c_int()
    hip.check(lib.rcfm_demod_get_option(h, hip.RCFM_OPT_GRAPH, ctypes.byref(v)))
    return v.value


@pytest.mark.parametrize("kind,B,A", [("WBFM", 240000, 48000), ("MFM", 240000, 48000), ("FM", 240000, 48000),
                                      ("WBFM", 250000, 48000), ("MFM", 12500, 8000)])
def test_replayed_chain_is_bit_identical_to_plain_launches(kind, B, A):
    import torch
    import workloads
    from radiocore._internal import hip
    lib = hip.lib()
    ch = 2 if kind == "WBFM" else 1
    plain, graph = _handle(lib, hip, kind, B, A, 0), _handle(lib, hip, kind, B, A, 1)
    assert _graphs(lib, hip, plain) == 0 and _graphs(lib, hip, graph) == 1
    bufs = [torch.view_as_real(hip.to_device(workloads.single_channel(B, i=s, stereo=(kind == "WBFM")), torch.complex64))
            for s in (1, 2, 3)]
    x = torch.empty_like(bufs[0])                 # ONE input buffer and ONE output buffer per handle: the pointers repeat
    want, got = torch.empty(A, ch, device="cuda"), torch.empty(A, ch, device="cuda")
    s = hip.stream()
    for i in range(7):                            # buffers 0, 1: plain launches + capture; from buffer 2 on: replay
        x.copy_(bufs[i % 3])
        hip.check(lib.rcfm_demod_run(plain, 0, 1, hip.ptr(x), hip.ptr(want), s))
        hip.check(lib.rcfm_demod_run(graph, 0, 1, hip.ptr(x), hip.ptr(got), s))
        torch.cuda.synchronize()
        assert torch.equal(want, got), (kind, i)
        assert float(want.abs().max()) > 1e-3
    assert _graphs(lib, hip, graph) == 2, "the chain was not captured: RCFM_OPT_GRAPH fell back to plain launches"
    if kind != "FM":                              # the state the replayed kernels carried is the plain handle's
        a, b = np.zeros((1, ch, 50), np.float32), np.zeros((1, ch, 50), np.float32)
        fp = ctypes.POINTER(ctypes.c_float)
        hip.check(lib.rcfm_demod_get_state(plain, a.ctypes.data_as(fp), s))
        hip.check(lib.rcfm_demod_get_state(graph, b.ctypes.data_as(fp), s))
        assert np.array_equal(a, b)
    # other pointers: a second chain is captured next to the first; an option change drops both
    y = torch.empty(A, ch, device="cuda")
    for _ in range(3):
        hip.check(lib.rcfm_demod_run(graph, 0, 1, hip.ptr(x), hip.ptr(y), s))
    assert _graphs(lib, hip, graph) == 3
    hip.check(lib.rcfm_demod_set_option(graph, hip.RCFM_OPT_NARROW_TILES, 0))
    assert _graphs(lib, hip, graph) == 1
    torch.cuda.synchronize()
    hip.check(lib.rcfm_demod_destroy(plain))
    hip.check(lib.rcfm_demod_destroy(graph))


def test_class_surface_reaches_the_replay():
    """`WBFM(...).run(x)` with a device tensor the caller reuses: the object's handle replays after two calls, and
    the audio equals a fresh object's plain launches buffer for buffer."""
    import torch
    import radiocore as rc
    import workloads
    from radiocore._internal import hip
    lib = hip.lib()
    B, A = 240000, 48000
    a, b = rc.WBFM(B, A, cuda=True), rc.WBFM(B, A, cuda=True)
    hip.check(lib.rcfm_demod_set_option(a._handle.value, hip.RCFM_OPT_GRAPH, 1))      # (off by default: no faster on ROCm 7)
    hip.check(lib.rcfm_demod_set_option(b._handle.value, hip.RCFM_OPT_GRAPH, 0))
    x = hip.to_device(workloads.single_channel(B, i=5), torch.complex64)
    for i in range(5):
        ya = a.run(x)
        yb = b.run(x)
        assert np.array_equal(ya, yb), i
    assert _graphs(lib, hip, a._handle.value) >= 1
