"""The tile-blocked hand-over of WBFM's mono signal and pilot band (RCFM_OPT_PILOT_BLOCKED, kernels.h PilotBlocked).

The pilot stage writes m and p in the order the pilot chain's 16-line tiles read them; the arithmetic is untouched, so
the audio must be BIT-identical with the option on and off, for every geometry the layout is planned for (row lengths
500 / 512 / 480 / 320 / 300 / 256, row counts that are not a multiple of the rows a workgroup owns, batches small enough
for the 8-line tile kernels) and for those it refuses (row lengths that are no multiple of 4 or would idle the stage).
Parity with the reference's algorithm is checked against the oracle on one of them.
"""
import ctypes

import numpy as np
import pytest

import workloads
from conftest import TOL, have_gpu, rel_err

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not have_gpu(), reason="needs an MI355X")]

RCFM_WBFM = 2


@pytest.fixture(scope="module")
def hip_lib():
    import radiocore
    from radiocore._internal import hip
    assert radiocore.HasCuda(), "librcfm.so did not load or sees no device"
    hip.torch()
    return hip, hip.lib()


def _run(hip, lib, iq, B, A, blocked, passes=1, effective=None):
    import torch
    C = iq.shape[0]
    h = ctypes.c_void_p()
    hip.check(lib.rcfm_demod_create(RCFM_WBFM, C, B, A, 75e-6, 0, ctypes.byref(h)))
    try:
        hip.check(lib.rcfm_demod_set_option(h, hip.RCFM_OPT_PILOT_BLOCKED, int(blocked)))
        if effective is not None:   # did the library plan the layout for this geometry?  (rcfm_demod_get_option)
            v = ctypes.c_int(-1)
            hip.check(lib.rcfm_demod_get_option(h, hip.RCFM_OPT_PILOT_BLOCKED, ctypes.byref(v)))
            assert v.value == int(effective), (B, A, v.value)
        x = torch.from_numpy(iq).cuda()
        out = []
        for _ in range(passes):   # (the second pass starts from the first one's de-emphasis state)
            audio = torch.empty((C, A, 2), dtype=torch.float32, device="cuda")
            hip.check(lib.rcfm_demod_run(h, 0, C, hip.ptr(x), hip.ptr(audio), hip.stream()))
            torch.cuda.synchronize()
            out.append(audio.cpu().numpy())
        return out
    finally:
        hip.check(lib.rcfm_demod_destroy(h))


def _signals(C, B, seed):
    return np.stack([workloads.single_channel(B, i=seed + i) for i in range(C)]).astype(np.complex64)


@pytest.mark.parametrize("C,B,A,planned", [
    (6, 240000, 48000, True),    # 480 x 500: the BASELINE geometry; 6 channels = 8-line tiles
    (40, 240000, 48000, True),   # the same on 16-line tiles
    (3, 256000, 32000, True),    # 500 x 512 / 512 x 500, odd channel count (the last pair is a single)
    (5, 96000, 48000, True),     # 300 x 320
    (4, 250000, 50000, True),    # 500 x 500
    (4, 125000, 25000, None),    # 250 x 500 / 500 x 250: planned one way round only (row length 250 is no multiple of 16 bytes)
    (4, 62500, 12500, False),    # 250 x 250: refused, natural order runs
    (2, 60000, 12000, None),     # 240 x 250 / 250 x 240: either
    (3, 360000, 48000, False),   # 600 x 600: two rows per workgroup would idle 40 % of the stage -- refused
    (2, 192000, 48000, None),    # 400 x 480 / 480 x 400: four rows of 400 idle a fifth of the stage -- refused one way round
    (2, 65536, 16384, True),     # 256 x 256: eight rows per workgroup
    (3, 75000, 15000, None),     # 250 x 300 / 300 x 250: six rows per workgroup, 250 rows: the last workgroup owns four
    (3, 128000, 32000, None),    # 320 x 400 / 400 x 320
    (2, 81920, 20480, None),     # 256 x 320 / 320 x 256
])
def test_blocked_layout_is_bit_identical(hip_lib, C, B, A, planned):
    hip, lib = hip_lib
    iq = _signals(C, B, 100 + C)
    a = _run(hip, lib, iq, B, A, blocked=1, passes=2, effective=planned)
    b = _run(hip, lib, iq, B, A, blocked=0, passes=2, effective=False)
    for x, y in zip(a, b):
        assert np.isfinite(x).all()
        np.testing.assert_array_equal(x, y)


def test_blocked_layout_against_the_oracle(hip_lib):
    hip, lib = hip_lib
    C, B, A = 4, 240000, 48000
    iq = _signals(C, B, 7)
    import radiocore_oracle
    got = _run(hip, lib, iq, B, A, blocked=1, effective=True)[0]
    for c in range(C):
        want = radiocore_oracle.WBFM(B, A).run(iq[c]).reshape(A, 2)
        assert rel_err(got[c], want) <= TOL, c
