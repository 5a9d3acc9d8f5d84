"""Channel sharding across the GPUs of one node.

Everything after the wideband FFT is independent per channel, so rank r of G owns the
contiguous channel range [r*C//G, (r+1)*C//G).  The wideband FFT cannot be sharded by
channel and is replicated.  The only collective on the path is the gather of the per-rank
audio blocks to the publishing rank -- RCCL over xGMI when the tensors live on GPUs
(torch.distributed backend "nccl"), gloo for the CPU tests.

The reference has no multi-device code at all; its per-channel loop is
examples/multi_fm_server.py:100-106.
"""

__all__ = ["channel_range", "channel_counts", "gather_audio", "GatherHandle"]


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
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return GatherHandle(None, lambda: local) if async_op else local
    # (a one-rank group runs the collective like any other: that is how a one-GPU box exercises the RCCL path)
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
