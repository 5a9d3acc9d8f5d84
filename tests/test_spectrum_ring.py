"""CPU, world_size 2 and 3, gloo: the rotating-FFT-owner protocol (radiocore.tools.sharding.SpectrumRing).

The library's tuner is replaced by a stand-in with the same six methods (window / spectrum_slot / attach / load /
adopt / shard) on host tensors, so the schedule, the ownership rotation, the window bookkeeping (wrapping windows,
halos) and the point-to-point message order run here without a GPU; tests/test_hip_sharded.py runs the real tuner
through the same class on one GPU.  What is asserted: on every rank and for every buffer, the bins its channels read
are exactly those of the FFT of THAT buffer, whoever computed it, while `lookahead` later buffers are in flight.

Reference: one process does everything (examples/multi_fm_server.py:95-106); nothing there to cite for the protocol."""

import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

N, C, HALO, ROW = 4096, 7, 64, 32
# channel c reads bins [centre - 40, centre + 40] modulo N; the first one wraps around bin 0
CENTRES = [10, 600, 1200, 1800, 2400, 3000, 3900]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class FakeTuner:
    """Host stand-in for radiocore.tools.Tuner's spectrum interface."""

    def __init__(self):
        self.slot = None
        self.base = 0                                # first bin the attached storage holds at element HALO
        self.loaded = None
        self.shard_range = None
        self.loads = 0

    def shard(self, first, count):
        self.shard_range = (first, count)

    def window(self, n, first, count):
        if count == 0:
            return 0, 0                              # a rank without channels reads nothing (rcfm_tuner_window)
        rows = n // ROW
        used = np.zeros(rows, bool)
        for c in range(first, first + count):
            for b in range(CENTRES[c] - 40, CENTRES[c] + 41):
                used[(b % n) // ROW] = True
        # everything but the longest circular run of unused rows (api.hip: rcfm_tuner_s::row_window)
        best, start, run = 0, 0, 0
        for i in range(2 * rows):
            if not used[i % rows]:
                run += 1
                if run > best and run <= rows:
                    best, start = run, i - run + 1
            else:
                run = 0
        if best == 0 or best >= rows:
            return 0, n
        lo, hi = (start + best) % rows, (start - 1) % rows
        return lo * ROW, ((hi - lo) % rows + 1) * ROW

    def spectrum_slot(self, n):
        t = torch.full((n + 2 * HALO,), complex(np.nan, np.nan), dtype=torch.complex64)
        t.rcfm_halo = HALO
        return t

    def window_slot(self, n, first, count):
        """[halo | the window's bins | halo], or None when the window wraps or touches the ends (rcfm_tuner_window_layout)."""
        fb, nb = self.window(n, first, count)
        if count == 0 or nb >= n or fb < HALO or fb + nb > n - HALO:
            return None
        t = torch.full((nb + 2 * HALO,), complex(np.nan, np.nan), dtype=torch.complex64)
        t.rcfm_halo = HALO
        t.rcfm_window = (first, count)
        t.first_bin = fb
        return t

    def attach(self, slot, n, loaded=None):
        self.slot, self.loaded, self.base = slot, loaded, 0

    def attach_window(self, slot, n):
        self.slot, self.loaded, self.base = slot, None, slot.first_bin

    def detach(self, n):
        self.slot, self.loaded, self.base = None, None, 0

    def load(self, x, whole=False):
        assert whole
        self.loads += 1
        X = torch.fft.fft(x)
        self.slot[HALO:HALO + N] = X
        self.slot[:HALO] = X[-HALO:]
        self.slot[HALO + N:] = X[:HALO]
        self.loaded = (0, C)

    def adopt(self, n, first, count):
        s = self.slot
        if getattr(s, "rcfm_window", None) is not None:
            assert s.rcfm_window == (first, count)
            self.loaded = (first, count)
            return
        s[:HALO] = s[N:N + HALO]                 # (the real adopt copies only the window's part: NaN stays NaN elsewhere)
        s[HALO + N:] = s[HALO:2 * HALO]
        self.loaded = (first, count)

    def read(self, c):
        assert self.loaded[0] <= c < self.loaded[0] + self.loaded[1]
        idx = HALO + CENTRES[c] - self.base + torch.arange(-40, 41)   # through the halo, no modulo: like the gather kernel
        return self.slot[idx]


def _buffer(i):
    g = torch.Generator().manual_seed(100 + i)
    return torch.complex(torch.randn(N, generator=g), torch.randn(N, generator=g)).to(torch.complex64)


def _worker(rank, world, port, lookahead, buffers, out_dir, channels=C):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, os.path.join(ROOT, "radio-core_amd"))
    from radiocore.tools import sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tuner = FakeTuner()
    ring = sharding.SpectrumRing(tuner, N, channels, lookahead=lookahead)
    ring.enable_timing()
    lo, hi = sharding.channel_range(rank, world, channels)
    assert tuner.shard_range == (lo, hi - lo)
    ok = True
    with pytest.raises(RuntimeError, match="after their submit"):
        ring.acquire(0)
    for j in range(min(ring.lookahead, buffers)):                 # prime
        ring.submit(j, _buffer(j) if ring.owner(j) == rank else None)
    for i in range(buffers):
        if i + ring.lookahead < buffers:
            j = i + ring.lookahead
            ring.submit(j, _buffer(j) if ring.owner(j) == rank else None)
        ring.acquire(i)
        want = torch.fft.fft(_buffer(i))
        for c in range(lo, hi):
            got = tuner.read(c)
            ref = want[(CENTRES[c] + torch.arange(-40, 41)) % N]
            ok = ok and bool(torch.equal(got, ref))
    with pytest.raises(RuntimeError, match="in order"):
        ring.submit(buffers + 5)
    owned = len([i for i in range(buffers) if i % world == rank])
    ok = ok and tuner.loads == owned                               # one FFT per owned buffer, none for the others
    t = ring.timing_summary()                                      # what bench.py prints per rank for N > 1
    ok = ok and t["fft_count"] == owned and t["send_count"] == owned and t["wait_count"] == buffers
    ok = ok and t["fft_ms"] >= 0 and t["send_ms"] >= 0 and t["wait_ms"] >= 0
    dist.barrier()
    # storage: whole slots for the buffers this rank owns, window-sized ones for the others when its window allows it
    in_flight = ring.lookahead + 1
    if ring.window_slots:
        ok = ok and ring.full_slots == -(-in_flight // world) and ring.window_slots == in_flight
        nb = tuner.window(N, lo, hi - lo)[1]
        ok = ok and ring.slot_bytes() == 8 * (ring.full_slots * (N + 2 * HALO) + in_flight * (nb + 2 * HALO))
    else:
        ok = ok and ring.full_slots == in_flight
    np.save(os.path.join(out_dir, "ok%d.npy" % rank), np.array([int(ok), ring.bytes_sent_per_buffer(), ring.window_slots]))
    ring.drain()
    ring.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,lookahead", [(2, None), (2, 1), (3, None), (3, 5)])
def test_rotating_owner_delivers_every_window(tmp_path, world, lookahead):
    mp.spawn(_worker, args=(world, _free_port(), lookahead, 9, str(tmp_path)), nprocs=world, join=True)
    windowed = []
    for r in range(world):
        ok, sent, win = np.load(os.path.join(str(tmp_path), "ok%d.npy" % r))
        assert ok == 1, r
        assert 0 < sent < 8 * N * (world - 1)                      # windows, not whole spectra
        windowed.append(int(win) > 0)
    # rank 0's first channel wraps around bin 0: it keeps whole slots; the last rank's window lies inside the spectrum
    assert windowed[0] is False and windowed[-1] is True


def test_more_ranks_than_channels(tmp_path):
    """C = 2 channels on G = 3 ranks: rank 0 owns none (sharding.channel_range), reads an empty window, takes part in no
    transfer and still walks the schedule (it owns every third buffer's FFT)."""
    mp.spawn(_worker, args=(3, _free_port(), None, 7, str(tmp_path), 2), nprocs=3, join=True)
    for r in range(3):
        ok, sent, win = np.load(os.path.join(str(tmp_path), "ok%d.npy" % r))
        assert ok == 1, r
    sys.path.insert(0, os.path.join(ROOT, "radio-core_amd"))
    from radiocore.tools.sharding import channel_range
    assert channel_range(0, 3, 2) == (0, 0)


def test_window_segments():
    sys.path.insert(0, os.path.join(ROOT, "radio-core_amd"))
    from radiocore.tools.sharding import window_segments
    assert window_segments(10, 5, 20) == [(10, 15)]
    assert window_segments(18, 5, 20) == [(18, 20), (0, 3)]
    assert window_segments(0, 20, 20) == [(0, 20)] and window_segments(7, 25, 20) == [(0, 20)]
