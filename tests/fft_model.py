"""numpy model of librcfm's multi-pass FFT, driven by the plan the library describes.

Mirrors radio-core_amd/csrc/fft_kernel.h step by step (tile load, in-place
decimation-in-frequency stages with their stage twiddles, digit-reversed read-out,
inter-pass twiddle, strided store) so the planner and every index formula are
checked on the CPU against numpy.fft; the GPU tests then only have to show that the
kernel executes this model.
"""

import ctypes

import numpy as np


def describe(n, max_l=0):
    from radiocore._internal import hip
    lib = hip.load_library()
    plan = hip.FftPlan()
    rc = lib.rcfm_fft_describe(ctypes.c_int64(n), max_l, ctypes.byref(plan))
    return plan if rc == 0 else None


def describe_plan(n, lengths, blocked=-1):
    """The plan for given pass lengths (rcfm_fft_describe_plan); `blocked` is its layout argument: -1 automatic, 0 plain,
    1 forces the tile-blocked hand-over, 2 the padded-rows layout."""
    from radiocore._internal import hip
    lib = hip.load_library()
    plan = hip.FftPlan()
    arr = (ctypes.c_int64 * len(lengths))(*lengths)
    rc = lib.rcfm_fft_describe_plan(ctypes.c_int64(n), arr, len(lengths), int(blocked), ctypes.byref(plan))
    return plan if rc == 0 else None


def lds_fft(tile, radices):
    """tile: [L, w] complex; in-place DIF stages, then read-out through pos[]."""
    L = tile.shape[0]
    x = tile.astype(np.complex128).copy()
    mt = L
    for r in radices:
        m = mt // r
        step = L // mt
        for g in range(L // mt):
            for kp in range(m):
                rows = g * mt + kp + m * np.arange(r)
                v = x[rows]                                           # [r, w]
                q = np.arange(r)
                dft = np.exp(-2j * np.pi * np.outer(q, q) / r)         # y[q'] = sum_q x[q] W_r^(q q')
                y = dft @ v
                tw = np.exp(-2j * np.pi * (q * kp * step) / L)         # stage_tw[q' * kp * step]
                x[rows] = y * tw[:, None]
        mt = m
    pos = np.zeros(L, np.int64)
    for k in range(L):
        rem, weight, slot = k, L, 0
        for r in radices:
            weight //= r
            slot += (rem % r) * weight
            rem //= r
        pos[k] = slot
    return x[pos]


def run_pass(p, n, src, dst):
    L = p.L
    radices = [p.radix[s] for s in range(p.nstages)]
    W = 16
    for o1 in range(p.n_o1):
        for o2 in range(p.n_o2):
            for i0 in range(0, p.n_inner, W):
                wv = min(W, p.n_inner - i0)
                i = i0 + np.arange(wv)
                l = np.arange(L)
                # tile-blocked hand-over (in_t / out_t != 0): tile t of 16 lines starts at t * in_t, lanes follow at in_i
                lane = np.arange(wv)
                tile_in = (i0 // W) * p.in_t + lane * p.in_i if p.in_t else i * p.in_i
                tile_out = (i0 // W) * p.out_t + lane * p.out_i if p.out_t else i * p.out_i
                in_addr = o1 * p.in_o1 + o2 * p.in_o2 + tile_in[None, :] + l[:, None] * p.in_l
                tile = src[in_addr]
                out = lds_fft(tile, radices)
                k = np.arange(L)
                if p.has_twiddle:
                    line = o1 * p.tw_o1 + o2 * p.tw_o2 + i * p.tw_i
                    e = line[None, :] * k[:, None]
                    assert e.max() < n                              # the kernel relies on this: no modulo
                    out = out * np.exp(-2j * np.pi * e / n)
                out_addr = o1 * p.out_o1 + o2 * p.out_o2 + tile_out[None, :] + k[:, None] * p.out_k
                dst[out_addr] = out


def run_pass_in_place(p, n, buf):
    """One strided pass reading and writing ONE array, tile after tile (what a kernel launched with in == out does, in the
    most forgiving order: a tile is read completely before it is written)."""
    run_pass_tiles = []
    L = p.L
    W = 16
    radices = [p.radix[s] for s in range(p.nstages)]
    for o1 in range(p.n_o1):
        for o2 in range(p.n_o2):
            for i0 in range(0, p.n_inner, W):
                wv = min(W, p.n_inner - i0)
                i = i0 + np.arange(wv)
                lane = np.arange(wv)
                tile_in = (i0 // W) * p.in_t + lane * p.in_i if p.in_t else i * p.in_i
                tile_out = (i0 // W) * p.out_t + lane * p.out_i if p.out_t else i * p.out_i
                l = np.arange(L)
                in_addr = o1 * p.in_o1 + o2 * p.in_o2 + tile_in[None, :] + l[:, None] * p.in_l
                out = lds_fft(buf[in_addr], radices)
                if p.has_twiddle:
                    line = o1 * p.tw_o1 + o2 * p.tw_o2 + i * p.tw_i
                    out = out * np.exp(-2j * np.pi * (line[None, :] * l[:, None]) / n)
                out_addr = o1 * p.out_o1 + o2 * p.out_o2 + tile_out[None, :] + l[:, None] * p.out_k
                run_pass_tiles.append((np.sort(in_addr.ravel()), np.sort(out_addr.ravel())))
                buf[out_addr] = out
    return run_pass_tiles


def model_fft_middle_in_place(x, plan):
    """The routing FftEngine::c2c takes when in / out / tmp are not three distinct arrays: x -> tmp, the middle pass
    tmp -> tmp IN PLACE, tmp -> out.  Returns (result, every tile of the middle pass wrote exactly what it read)."""
    n = plan.n
    assert plan.npass == 3
    tmp = np.zeros(plan.tmp_stride, np.complex128)
    out = np.zeros(n, np.complex128)
    run_pass(plan.passes[0], n, np.asarray(x, np.complex128), tmp)
    tiles = run_pass_in_place(plan.passes[1], n, tmp)
    run_pass(plan.passes[2], n, tmp, out)
    return out, all(np.array_equal(a, b) for a, b in tiles)


def model_fft(x, plan):
    n = plan.n
    tmp = np.zeros(plan.tmp_stride, np.complex128)     # two-pass plans pad the scratch rows
    out = np.zeros(n, np.complex128)
    for t in range(plan.npass):
        p = plan.passes[t]
        src = x if t == 0 else tmp
        dst = out if t == plan.npass - 1 else tmp
        if src is dst:
            src = src.copy()
        run_pass(p, n, np.asarray(src, np.complex128), dst)
    return out
