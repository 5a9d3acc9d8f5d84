"""CPU, world_size 2, gloo: the channel-sharded path of bench.py --gpus N.

Each rank demodulates its own channel range (here with the oracle standing in for the GPU,
this container has none) and the blocks are gathered to rank 0 exactly as bench.py does with
RCCL; the result must equal the single-process loop.
"""

import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

N, B, A = 600000, 60000, 12000
CENTRES = [100.00e6, 100.05e6, 99.93e6, 100.12e6, 99.85e6]     # 5 channels over 2 ranks: 2 + 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _audio_for(channel_ids):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import radiocore_oracle as oracle
    import workloads
    tuner = oracle.Tuner()
    for f in CENTRES:
        tuner.add_channel(f, B, None)
    tuner.request_bandwidth(float(N))
    x = workloads.wideband(N, tuner.input_frequency, CENTRES, B, gain=0.4)
    tuner.load(x)                                    # the wideband FFT is replicated on every rank
    out = [oracle.MFM(B, A).run(tuner.run_pruned(i)) for i in channel_ids]
    return np.stack(out) if out else np.zeros((0, A, 1), np.float32)


def _worker(rank, world, port, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, os.path.join(ROOT, "radio-core_amd", "radiocore", "tools"))
    import sharding                                   # plain module: no GPU needed to import it
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sharding.channel_range(rank, world, len(CENTRES))
    local = torch.from_numpy(_audio_for(range(lo, hi)))
    full = sharding.gather_audio(local, len(CENTRES), dst=0)
    # the overlapped form bench.py uses (buffer i travels while buffer i+1 is computed): same result
    handle = sharding.gather_audio(local, len(CENTRES), dst=0, async_op=True)
    again = handle.wait()
    assert (again is None) == (full is None)
    if full is not None:
        assert torch.equal(again, full)
    dist.barrier()
    if rank == 0:
        np.save(result_path, full.numpy())
    else:
        assert full is None
    dist.destroy_process_group()


def test_channel_ranges_cover_everything_once():
    sys.path.insert(0, os.path.join(ROOT, "radio-core_amd", "radiocore", "tools"))
    import sharding
    for world in (1, 2, 3, 4, 8):
        for C in (1, 5, 64, 1024, 8192):
            edges = [sharding.channel_range(r, world, C) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == C
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            assert sum(sharding.channel_counts(world, C)) == C
    with pytest.raises(ValueError):
        sharding.channel_range(2, 2, 10)


@pytest.mark.timeout(300)
def test_two_rank_gather_equals_single_process(tmp_path):
    result = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(2, _free_port(), result), nprocs=2, join=True)
    got = np.load(result)
    want = _audio_for(range(len(CENTRES)))
    assert got.shape == want.shape == (len(CENTRES), A, 1)
    assert np.array_equal(got, want)


def _one_rank_worker(rank, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.pop("RCFM_GATHER_FORCE_COLLECTIVE", None)
    sys.path.insert(0, os.path.join(ROOT, "radio-core_amd"))
    from radiocore.tools import sharding
    dist.init_process_group("gloo", rank=0, world_size=1)
    block = np.arange(24, dtype=np.float32).reshape(3, 4, 2)          # what run_all() returns by default: numpy
    same = sharding.gather_audio(block, 3) is block                   # one rank: `local` IS the result
    h = sharding.gather_audio(block, 3, async_op=True)
    same = same and h.wait() is block
    t = torch.arange(24, dtype=torch.float32).reshape(3, 4, 2)
    out = torch.zeros_like(t)
    got = sharding.gather_audio(t, 3, out=out)                        # an `out` buffer still receives the block
    same = same and got is out and bool(torch.equal(out, t))
    os.environ["RCFM_GATHER_FORCE_COLLECTIVE"] = "1"                  # the collective path on the same one-rank group
    forced = sharding.gather_audio(t, 3)
    same = same and forced is not t and bool(torch.equal(forced, t))
    np.save(out_path, np.array([int(same)]))
    dist.destroy_process_group()


def test_one_rank_group_returns_the_local_block_untouched(tmp_path):
    """gather_audio on a one-rank group (torch.distributed initialised, world size 1): no collective, no copy -- a numpy
    block passes through like before round 3; RCFM_GATHER_FORCE_COLLECTIVE=1 is what the RCCL dry runs set."""
    out = str(tmp_path / "one.npy")
    mp.spawn(_one_rank_worker, args=(_free_port(), out), nprocs=1, join=True)
    assert np.load(out)[0] == 1
