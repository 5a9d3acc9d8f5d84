"""CPU: the FFT engine's planner and index maps (no GPU): the plan librcfm describes,
executed by the numpy model of its kernel, equals numpy.fft."""

import numpy as np
import pytest

import fft_model


def test_lengths_of_the_hot_path_are_planned():
    for n, npass in [(240000, 2), (48000, 2), (12500, 2), (8000, 2), (60000, 2), (12000, 2),
                     (10_000_000, 3), (100_000_000, 3), (240_000_000, 3), (600000, 3), (256000, 2),
                     (2 ** 28, 4)]:
        plan = fft_model.describe(n)
        assert plan is not None, n
        assert plan.npass == npass, (n, plan.npass)
        prod = 1
        for t in range(plan.npass):
            p = plan.passes[t]
            prod *= p.L
            r = 1
            for s in range(p.nstages):
                assert p.radix[s] in (2, 3, 4, 5, 6, 7, 8, 10, 12, 15, 16, 20, 24, 25, 32)   # 20..32: in-register composites
                r *= p.radix[s]
            # up to 512 points in ordinary tiles; the big tiles (600 / 625 / 640 points, two 1024-thread workgroups per
            # CU) only in three-pass plans: where they save a fourth pass (2.4e8) or beat two 500-point passes (1e8)
            assert r == p.L and 16 <= p.L <= (640 if plan.npass == 3 else 512)
        assert prod == n
    assert [fft_model.describe(100_000_000).passes[t].L for t in range(3)] == [400, 625, 400]
    assert [fft_model.describe(240_000_000).passes[t].L for t in range(3)] == [600, 625, 640]
    assert [fft_model.describe(240_000).passes[t].L for t in range(2)] == [480, 500]


@pytest.mark.parametrize("n", [100, 24001, 11 * 4096, 255])
def test_unsupported_lengths_are_refused(n):
    assert fft_model.describe(n) is None


@pytest.mark.parametrize("n", [256, 1000, 4096, 6000, 12500, 24000, 32768, 60000])
def test_model_of_the_plan_is_an_fft(n):
    plan = fft_model.describe(n)
    assert plan is not None
    r = np.random.default_rng(n)
    x = r.standard_normal(n) + 1j * r.standard_normal(n)
    got = fft_model.model_fft(x, plan)
    want = np.fft.fft(x)
    assert np.max(np.abs(got - want)) <= 1e-9 * np.max(np.abs(want))


@pytest.mark.parametrize("n", [44100, 88200, 22050, 7 * 4096, 33600, 7 ** 4 * 16])
def test_lengths_with_a_factor_seven(n):
    """Round 6: a radix-7 butterfly in the generic tile kernel keeps audio rates like 44 100 = 210 x 210 inside the
    engine (before, such a length sent every demodulator transform to rocFFT: 2.6 x the channel stages of 48 000,
    bench.py other_configs.path_map)."""
    plan = fft_model.describe(n)
    assert plan is not None and plan.npass == 2
    assert any(plan.passes[t].radix[s] == 7 for t in range(plan.npass) for s in range(plan.passes[t].nstages))
    r = np.random.default_rng(n)
    x = r.standard_normal(n) + 1j * r.standard_normal(n)
    got = fft_model.model_fft(x, plan)
    want = np.fft.fft(x)
    assert np.max(np.abs(got - want)) <= 1e-9 * np.max(np.abs(want))


@pytest.mark.parametrize("n,max_l,npass", [(512000, 0, 3), (16 * 16 * 20, 20, 3), (16 * 16 * 16 * 20, 20, 4),
                                           (18 * 20 * 16 * 16, 20, 4)])
def test_deep_plans(n, max_l, npass):
    plan = fft_model.describe(n, max_l)
    assert plan is not None and plan.npass == npass
    r = np.random.default_rng(n)
    x = r.standard_normal(n) + 1j * r.standard_normal(n)
    got = fft_model.model_fft(x, plan)
    want = np.fft.fft(x)
    assert np.max(np.abs(got - want)) <= 1e-9 * np.max(np.abs(want))


@pytest.mark.parametrize("lengths", [(20, 25, 32), (16, 20, 48), (64, 16, 16)])
def test_tile_blocked_hand_over_between_the_first_two_passes(lengths):
    """Three-pass plans of transforms beyond the Infinity Cache let the first pass write every tile as ONE contiguous run
    ([tile][k_1][16]) and the second pass read that layout (FftPass::out_t / in_t).  Forced here at model sizes: the
    blocked plan is an FFT, its intermediate really is tile-contiguous, and the plain plan of the same lengths is not."""
    n = lengths[0] * lengths[1] * lengths[2]
    plain = fft_model.describe_plan(n, lengths, blocked=0)
    blocked = fft_model.describe_plan(n, lengths, blocked=1)
    assert plain is not None and blocked is not None
    assert plain.passes[0].out_t == 0 and plain.passes[1].in_t == 0
    p0, p1 = blocked.passes[0], blocked.passes[1]
    assert p0.out_t == 16 * lengths[0] and p0.out_k == 16 and p1.in_t == p0.out_t and p1.in_o1 == 16
    assert p1.in_l == lengths[2] * lengths[0]
    r = np.random.default_rng(n)
    x = r.standard_normal(n) + 1j * r.standard_normal(n)
    want = np.fft.fft(x)
    for plan in (plain, blocked):
        got = fft_model.model_fft(x, plan)
        assert np.max(np.abs(got - want)) <= 1e-9 * np.max(np.abs(want))
    # the default (by size) leaves small transforms plain, and lengths whose third factor is not a multiple of 16 too
    assert fft_model.describe_plan(n, lengths, blocked=-1).passes[0].out_t == 0
    assert fft_model.describe_plan(20 * 32 * 25, (20, 32, 25), blocked=1).passes[0].out_t == 0
    big = fft_model.describe(240_000_000)
    assert big.passes[0].out_t == 16 * 600 and big.passes[1].in_l == 640 * 600


@pytest.mark.parametrize("lengths", [(20, 25, 32), (16, 16, 32), (64, 16, 16)])
def test_a_blocked_middle_pass_may_not_run_in_place(lengths):
    """Why FftEngine::c2c keeps the tile-blocked hand-over for calls with three distinct arrays: the
    blocked second pass reads [tile][k_1][16] and writes the plain layout -- in place it overwrites inputs no tile has
    read yet.  The plain plan of the same lengths reads and writes the same addresses per tile and is an FFT in place."""
    n = lengths[0] * lengths[1] * lengths[2]
    plain = fft_model.describe_plan(n, lengths, blocked=0)
    blocked = fft_model.describe_plan(n, lengths, blocked=1)
    r = np.random.default_rng(n)
    x = r.standard_normal(n) + 1j * r.standard_normal(n)
    want = np.fft.fft(x)
    got, same_set = fft_model.model_fft_middle_in_place(x, plain)
    assert same_set and np.max(np.abs(got - want)) <= 1e-9 * np.max(np.abs(want))
    got, same_set = fft_model.model_fft_middle_in_place(x, blocked)
    assert not same_set
    assert np.max(np.abs(got - want)) > 1e-3 * np.max(np.abs(want))     # garbage, as the advisor's simulation found


@pytest.mark.parametrize("lengths", [(16, 20, 24), (32, 25, 40), (16, 16, 600)])
def test_padded_rows_layout(lengths):
    """Layout 2 (fft_engine.h): the scratch rows of n_3 points at a pitch of whole 128-byte lines, [k_1][k_2][pitch] -- the
    layout the tuner's aligned plan (N = 2.4e8 as 640 x 625 x 600, pitch 608) runs in.  The plan is an FFT; every write of
    the first two passes and every read of the last two starts on a 16-point boundary; only the first pass's reads do
    not; its launch is flat (XCD-aware over all tiles)."""
    n = lengths[0] * lengths[1] * lengths[2]
    plan = fft_model.describe_plan(n, lengths, blocked=2)
    assert plan is not None
    pitch = (lengths[2] + 15) // 16 * 16
    assert plan.tmp_stride == lengths[0] * lengths[1] * pitch
    p0, p1, p2 = plan.passes[0], plan.passes[1], plan.passes[2]
    assert p0.flat_outer == 1 and p1.flat_outer == 0 and p2.flat_outer == 0
    assert (p0.n_o1, p0.n_inner, p0.in_o1, p0.in_l) == (lengths[1], lengths[2], lengths[2], lengths[1] * lengths[2])
    assert p0.out_o1 % 16 == 0 and p0.out_k % 16 == 0
    assert p1.in_o1 % 16 == 0 and p1.in_l % 16 == 0 and p1.out_o1 % 16 == 0 and p1.out_k % 16 == 0
    assert p2.in_i % 16 == 0 and p2.in_o1 % 16 == 0
    assert p2.out_k % 16 == 0 or (lengths[0] * lengths[1]) % 16 != 0
    r = np.random.default_rng(n)
    x = r.standard_normal(n) + 1j * r.standard_normal(n)
    got = fft_model.model_fft(x, plan)
    want = np.fft.fft(x)
    assert np.max(np.abs(got - want)) <= 1e-9 * np.max(np.abs(want))
    # the middle pass keeps its addresses: the plan also runs with ONE scratch array (middle pass in place)
    got, same_set = fft_model.model_fft_middle_in_place(x, plan)
    assert same_set and np.max(np.abs(got - want)) <= 1e-9 * np.max(np.abs(want))


def test_padded_rows_layout_is_for_three_passes():
    assert fft_model.describe_plan(240000, (480, 500), blocked=2) is None
