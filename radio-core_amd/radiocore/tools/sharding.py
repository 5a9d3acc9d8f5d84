"""Channel sharding across the GPUs of one node.

Everything after the wideband FFT is independent per channel, so rank r of G owns the
contiguous channel range [r*C//G, (r+1)*C//G).  The wideband FFT cannot be sharded by
channel and is replicated.  The only collective on the path is the gather of the per-rank
audio blocks to the publishing rank -- RCCL over xGMI when the tensors live on GPUs
(torch.distributed backend "nccl"), gloo for the CPU tests.

The reference has no multi-device code at all; its per-channel loop is
examples/multi_fm_server.py:100-106.
"""

import time

__all__ = ["channel_range", "channel_counts", "gather_audio", "GatherHandle", "SpectrumRing", "window_segments"]


def channel_range(rank, world, channels):
    """[first, last) channel indices of `rank`."""
    if not (0 <= rank < world):
        raise ValueError("rank outside the world")
    return rank * channels // world, (rank + 1) * channels // world


def channel_counts(world, channels):
    return [channel_range(r, world, channels)[1] - channel_range(r, world, channels)[0] for r in range(world)]


class GatherHandle:
    """A gather in flight (gather_audio(..., async_op=True)).  wait() orders the caller's stream (RCCL)
    or thread (gloo) after the collective and returns what gather_audio would have returned."""

    def __init__(self, work, finish):
        self._work = work
        self._finish = finish

    def wait(self):
        if self._work is not None:
            self._work.wait()
            self._work = None
        return self._finish()


def gather_audio(local, channels, dst=0, group=None, out=None, async_op=False):
    """Gather per-rank audio [C_r, A, ch] to `dst`; returns [C, A, ch] there, None elsewhere.

    `out` (on `dst`, shape [C, A, ch]) receives the blocks in place when every rank owns the
    same number of channels: no staging copy on the hot path.

    torch.distributed.gather needs equally sized blocks; when C is not divisible by G the
    shorter blocks are padded to the longest one and trimmed on arrival.

    async_op=True returns a GatherHandle at once: the collective runs on RCCL's own stream behind
    the work already queued on the caller's stream, so buffer i can travel over xGMI while the
    kernels of buffer i+1 run (`local` and `out` must stay untouched until wait()).
    """
    import os
    import torch
    import torch.distributed as dist
    # One rank: nothing travels -- `local` IS the result, whatever it is (a numpy block from run_all() included).
    # RCFM_GATHER_FORCE_COLLECTIVE=1 (bench.py's RCFM_BENCH_FORCE_DIST runs, tests/test_nccl_world1.py) runs the
    # collective on a one-rank group anyway: that is how a one-GPU box exercises the RCCL path.
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and
                                     os.environ.get("RCFM_GATHER_FORCE_COLLECTIVE") != "1"):
        if out is not None and out is not local:
            out.copy_(local if isinstance(local, torch.Tensor) else torch.as_tensor(local))
            local = out
        return GatherHandle(None, lambda: local) if async_op else local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    counts = channel_counts(world, channels)
    if local.shape[0] != counts[rank]:
        raise ValueError("local block has %d channels, rank %d owns %d" % (local.shape[0], rank, counts[rank]))
    most = max(counts)
    send = local.contiguous()
    if counts[rank] != most:
        pad = torch.zeros((most - counts[rank],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        send = torch.cat([send, pad], dim=0)
    even = min(counts) == most
    recv = None
    if rank == dst:
        if even and out is not None:
            recv = list(out.split(most, dim=0))          # views: the collective writes into `out`
        else:
            recv = [torch.empty((most,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
                    for _ in range(world)]
    work = dist.gather(send, recv, dst=dst, group=group, async_op=async_op)

    def finish():
        if rank != dst:
            return None
        if even and out is not None:
            return out
        full = torch.cat([blk[:c] for blk, c in zip(recv, counts)], dim=0)
        if out is not None:
            out.copy_(full)
            return out
        return full

    return GatherHandle(work, finish) if async_op else finish()


def window_segments(first_bin, nbins, n):
    """A circular bin window as at most two [start, stop) pieces of [0, n)."""
    if nbins >= n:
        return [(0, n)]
    stop = first_bin + nbins
    if stop <= n:
        return [(first_bin, stop)]
    return [(first_bin, n), (0, stop - n)]


class SpectrumRing:
    """Rotating FFT owner: the wideband FFT is neither replicated nor distributed -- it takes turns.

    With the FFT replicated (channel sharding as above) every rank ingests the whole buffer and runs the whole
    transform: end to end the path scales 2.8x at 8 GPUs, and not at all once the host link is the limit.  Here rank
    i mod G OWNS buffer i: it alone ingests it (its own PCIe link) and runs FFT_N, then sends every peer just the bins
    that peer's channels read (Tuner.window: about N/G bins, 240 MB at cfg4 / G = 8) over xGMI -- G - 1 point-to-point
    transfers on G - 1 different links.  Owners run `lookahead` buffers ahead of the channel stages, so the transfers
    and the other ranks' FFTs hide behind channel work; per buffer every rank then spends (T_fft + T_chan) / G, and
    PCIe ingest per rank drops to 1/G.  The price is latency: `lookahead` buffers are in flight.

    Protocol, identical on every rank, buffers in order:
        submit(i, x)    the owner of buffer i passes the samples (device tensor) -- FFT + sends; the others pass
                        None -- their receives are posted;
        acquire(i)      the tuner now holds buffer i's spectrum for this rank's channels: run_all() may follow.
    Callers prime the ring with submit(0 .. lookahead - 1) and then alternate submit(i + lookahead) / acquire(i).

    Storage: a rank needs a WHOLE spectrum only for the buffers it owns -- ceil((lookahead + 1) / G) of the lookahead + 1
    in flight, two at the default lookahead = G; every other buffer in flight needs room for this rank's window only
    (Tuner.window_slot: [halo | nbins | halo]).  At cfg4 / G = 8 that is 2 x 1.92 GB + 9 x 0.26 GB per rank instead of
    9 x 1.92 GB.  A rank whose window wraps around the ends of the spectrum (its halos repeat the other end) keeps
    whole slots for everything.

    `tuner` needs: window(n, first, count), spectrum_slot(n), attach(slot, n, loaded), load(x, whole=True),
    adopt(n, first, count) and shard(first, count) -- radiocore.tools.Tuner, or a stand-in (tests); optionally
    window_slot(n, first, count) (None: no window storage for that range) and attach_window(slot, n).
    The reference has nothing of the kind: its one process does everything (examples/multi_fm_server.py:95-106).
    """

    def __init__(self, tuner, n, channels, lookahead=None, group=None):
        import torch.distributed as dist
        self._dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.tuner = tuner
        self.n = int(n)
        self.channels = int(channels)
        self.lookahead = int(lookahead) if lookahead else self.world
        self.ranges = [channel_range(r, self.world, self.channels) for r in range(self.world)]
        lo, hi = self.ranges[self.rank]
        tuner.shard(lo, hi - lo)
        self.segments = [window_segments(*tuner.window(self.n, a, b - a), self.n) if b > a else []
                         for a, b in self.ranges]
        # whole slots for owned buffers, window-sized ones for the rest (when this rank's window allows it)
        in_flight = self.lookahead + 1
        probe = None
        if self.world > 1 and hi > lo and hasattr(tuner, "window_slot") and len(self.segments[self.rank]) == 1:
            probe = tuner.window_slot(self.n, lo, hi - lo)
        if probe is not None:
            self._full = [tuner.spectrum_slot(self.n) for _ in range(-(-in_flight // self.world))]
            self._win = [probe] + [tuner.window_slot(self.n, lo, hi - lo) for _ in range(in_flight - 1)]
        else:
            self._full = [tuner.spectrum_slot(self.n) for _ in range(in_flight)]
            self._win = []
        self.slots = self._full + self._win
        self.full_slots, self.window_slots = len(self._full), len(self._win)
        self.halo = int(getattr(self._full[0], "rcfm_halo", (self._full[0].shape[0] - self.n) // 2))
        self._pending = {}      # buffer index -> (own, outstanding works, event of the owner's FFT)
        self._next_submit = 0
        self._next_acquire = 0
        # On a GPU the owner's ingest + FFT + sends run on their own stream, so that they overlap this rank's channel
        # kernels (and a host-fed buffer's PCIe copy does not stall them); `staging` receives host-fed buffers.
        self._side = None
        self._staging = None
        self._timing = None     # enable_timing(): event pairs / wall-clock samples of the FFT, the sends and the waits
        if self.slots[0].is_cuda:
            import torch
            self._torch = torch
            self._side = torch.cuda.Stream()

    def owner(self, i):
        return i % self.world

    def _slot_of(self, i):
        """Storage of buffer i on this rank: the k-th owned buffer takes whole slot k mod F; with window slots the j-th
        buffer of another owner takes window slot j mod (lookahead + 1).  Any lookahead + 1 consecutive buffers hold at
        most F owned ones, so no two buffers in flight share storage."""
        if not self._win:
            return self._full[i % len(self._full)]
        if self.owner(i) == self.rank:
            return self._full[(i // self.world) % len(self._full)]
        owned_before = max(0, (i - self.rank + self.world - 1) // self.world)
        return self._win[(i - owned_before) % len(self._win)]

    def slot_bytes(self):
        return sum(int(t.numel()) * t.element_size() for t in self.slots)

    def _views(self, slot, rank):
        # (a rank without channels -- C < G -- reads nothing: no pieces, no transfer)
        if getattr(slot, "rcfm_window", None) is not None:      # a window slot holds this rank's one segment from `halo` on
            (a, b), = self.segments[rank]
            return [slot[self.halo:self.halo + (b - a)]]
        return [slot[self.halo + a:self.halo + b] for a, b in self.segments[rank] if b > a]

    def bytes_sent_per_buffer(self):
        """What the owner of a buffer puts on the links (all peers together)."""
        return sum(8 * (b - a) for r in range(self.world) if r != self.rank for a, b in self.segments[r])

    def submit(self, i, x=None):
        if i != self._next_submit:
            raise RuntimeError("SpectrumRing.submit: buffers must be submitted in order (expected %d)" % self._next_submit)
        if i - self._next_acquire > self.lookahead:
            raise RuntimeError("SpectrumRing.submit: every slot is in flight; acquire buffer %d first" % self._next_acquire)
        self._next_submit += 1
        dist = self._dist
        slot = self._slot_of(i)
        own = self.owner(i) == self.rank
        works = []
        event = None
        if own:
            if x is None:
                raise ValueError("SpectrumRing.submit: the owner of buffer %d must pass its samples" % i)
            if self._side is not None:
                torch = self._torch
                # the slot's previous readers (channel kernels of buffer i - slots, queued on the current stream) first
                self._side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self._side):
                    if not x.is_cuda:      # host-fed: this rank's PCIe link carries buffer i, and only this rank's
                        if self._staging is None:
                            self._staging = [torch.empty(self.n, dtype=x.dtype, device="cuda") for _ in range(2)]
                        dev = self._staging[(i // self.world) % 2]
                        dev.copy_(x, non_blocking=True)
                        x = dev
                    e0 = self._mark(self._side)
                    self.tuner.attach(slot, self.n, None)
                    self.tuner.load(x, whole=True)
                    e1 = self._mark(self._side)
                    works = self._send(slot)
                    event = torch.cuda.Event()
                    event.record(self._side)
                    if self._timing is not None:
                        for w in works:
                            w.wait()              # (the side stream waits; the host does not) -> e2 = the sends have left
                        self._timing["fft"].append((e0, e1))
                        self._timing["send"].append((e1, self._mark(self._side)))
            else:
                t0 = time.perf_counter()
                self.tuner.attach(slot, self.n, None)
                self.tuner.load(x, whole=True)
                t1 = time.perf_counter()
                works = self._send(slot)
                if self._timing is not None:
                    self._timing["fft"].append(1e3 * (t1 - t0))
                    self._timing["send"].append(1e3 * (time.perf_counter() - t1))
        elif self.world > 1:
            src = self.owner(i)
            views = self._views(slot, self.rank)
            if self._device_comm(slot):
                works = dist.batch_isend_irecv([dist.P2POp(dist.irecv, v, src, self.group) for v in views])
            else:
                for v in views:
                    h = v.cpu()
                    dist.recv(h, src, group=self.group)
                    v.copy_(h)
        self._pending[i] = (own, works, event)

    def _device_comm(self, slot):
        return slot.is_cuda and self._dist.get_backend(self.group) == "nccl"

    def _send(self, slot):
        """The owner's half of the hand-over: every peer gets the bins its channels read."""
        dist = self._dist
        if self.world == 1:
            return []
        ops = []
        for r in range(self.world):
            if r == self.rank:
                continue
            for v in self._views(slot, r):
                if self._device_comm(slot):
                    ops.append(dist.P2POp(dist.isend, v, r, self.group))
                else:                       # dry run through host memory (gloo): same messages, same order
                    dist.send(v.cpu(), r, group=self.group)
        return dist.batch_isend_irecv(ops) if ops else []

    def acquire(self, i):
        if i != self._next_acquire or i not in self._pending:
            raise RuntimeError("SpectrumRing.acquire: buffers are acquired in order, after their submit")
        self._next_acquire += 1
        own, works, event = self._pending.pop(i)
        t0 = time.perf_counter()
        ea = self._mark(None)
        if event is not None:
            self._torch.cuda.current_stream().wait_event(event)      # this buffer's FFT (side stream) only
        for w in works:
            w.wait()                  # RCCL: orders the current stream behind the transfer
        if self._timing is not None:
            self._timing["wait"].append((ea, self._mark(None)) if ea is not None else 1e3 * (time.perf_counter() - t0))
        slot = self._slot_of(i)
        lo, hi = self.ranges[self.rank]
        if own:
            self.tuner.attach(slot, self.n, (0, self.channels))
        elif getattr(slot, "rcfm_window", None) is not None:
            self.tuner.attach_window(slot, self.n)
            self.tuner.adopt(self.n, lo, hi - lo)
        else:
            self.tuner.attach(slot, self.n, None)
            self.tuner.adopt(self.n, lo, hi - lo)

    # ---- where a rank's time goes (bench.py prints it per rank for N > 1) ----------------------------------------

    def enable_timing(self):
        """From now on: time of every owned buffer's FFT and sends (on the owner's stream) and how long the channel
        stream had to wait for a buffer's bins in acquire().  timing_summary() returns the means in milliseconds.
        The owner's stream then waits for its sends before it takes the next buffer -- NOT the shipping schedule: time
        throughput with timing off (bench.py runs a pass of its own for these numbers, after the timed region)."""
        self._timing = {"fft": [], "send": [], "wait": []}

    def disable_timing(self):
        self._timing = None

    def _mark(self, stream):
        if self._timing is None or self._side is None:
            return None
        e = self._torch.cuda.Event(enable_timing=True)
        e.record(stream if stream is not None else self._torch.cuda.current_stream())
        return e

    def timing_summary(self):
        if self._timing is None:
            return None
        if self._side is not None:
            self._torch.cuda.synchronize()
        out = {}
        for key, samples in self._timing.items():
            ms = [s[0].elapsed_time(s[1]) if isinstance(s, tuple) else s for s in samples]
            out[key + "_ms"] = round(sum(ms) / len(ms), 4) if ms else 0.0
            out[key + "_count"] = len(ms)
        return out

    def drain(self):
        """Complete every transfer that has been posted (acquire whatever is still in flight, without running it)."""
        while self._next_acquire < self._next_submit:
            self.acquire(self._next_acquire)

    def close(self):
        """After drain(): hand the tuner its own spectrum storage back and let the slots go."""
        if hasattr(self.tuner, "detach"):
            self.tuner.detach(self.n)
        self.slots = self._full = self._win = []
        self._staging = None
