"""GPU: the channel-sharded HIP path as bench.py --gpus N runs it, on ONE device.

Two processes (world_size 2, gloo rendezvous on 127.0.0.1), both on cuda:0.  Each rank declares its
channel range (Tuner.shard -> rcfm_tuner_shard: the wideband FFT stores only the spectrum rows the range
reads), loads the same wideband buffer, runs rcfm_pipeline_run(first = lo, count = n) through
Tuner.run_all() for two consecutive buffers (per-channel de-emphasis state at offset first * ch * 50),
moves its block to the host and gathers with sharding.gather_audio.  Rank 0's gathered result must equal a
single-process run_all(): bit for bit when the shard boundary keeps the channel pairing (even split),
within float32 rounding otherwise (a channel shares one complex FFT with its pair partner).

Reference loop: examples/multi_fm_server.py:100-106.
"""

import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, TOL, have_gpu, rel_err

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not have_gpu(), reason="needs an MI355X")]

N, B, A = 2_400_000, 60000, 12000
BUFFERS = 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _paths():
    for p in (ROOT, os.path.join(ROOT, "radio-core_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _inputs(kind, C):
    import workloads
    centres = workloads.channel_grid(C, 250000)          # spread: each half of the channels reads part of the band
    f_in = (min(centres) + max(centres)) / 2
    x = workloads.wideband(N, f_in, centres, B, gain=0.35, stereo=(kind == "WBFM"))
    return centres, [np.roll(x, 3111 * b) for b in range(BUFFERS)]


def _tuner(rc, kind, centres):
    t = rc.Tuner()
    for f in centres:
        t.add_channel(f, B, getattr(rc, kind)(B, A))
    t.request_bandwidth(float(N))
    return t


def _worker(rank, world, port, kind, C, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    _paths()
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    import radiocore as rc
    from radiocore.tools import sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    centres, bufs = _inputs(kind, C)
    lo, hi = sharding.channel_range(rank, world, C)
    tuner = _tuner(rc, kind, centres)
    tuner.shard(lo, hi - lo)
    gathered = []
    for x in bufs:
        tuner.load(x)
        block = tuner.run_all()                            # [hi - lo, A, ch]: only this rank's range runs
        assert block.shape[0] == hi - lo
        gathered.append(sharding.gather_audio(torch.from_numpy(block), C, dst=0))
    # a channel outside the loaded shard is refused, not silently computed from stale spectrum rows
    other = (hi % C) if hi < C else 0
    if not (lo <= other < hi):
        with pytest.raises(RuntimeError, match="outside the shard"):
            tuner.run(other)
    dist.barrier()
    if rank == 0:
        np.save(out_path, np.stack([g.numpy() for g in gathered]))
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("kind,C,exact", [("WBFM", 8, True), ("MFM", 8, True), ("FM", 7, False)])
def test_two_ranks_on_one_gpu_equal_single_process(tmp_path, kind, C, exact):
    import torch.multiprocessing as mp
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(2, _free_port(), kind, C, out), nprocs=2, join=True)
    got = np.load(out)
    _paths()
    import radiocore as rc
    centres, bufs = _inputs(kind, C)
    tuner = _tuner(rc, kind, centres)
    for b, x in enumerate(bufs):
        tuner.load(x)
        want = tuner.run_all()
        assert got[b].shape == want.shape
        if exact:
            assert np.array_equal(got[b], want), (kind, b, float(np.max(np.abs(got[b] - want))))
        else:
            for c in range(C):
                assert rel_err(got[b][c], want[c]) <= 0.05 * TOL, (kind, b, c)


def _ring_worker(rank, world, port, kind, C, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    _paths()
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    import radiocore as rc
    from radiocore.tools import sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    centres, bufs = _inputs(kind, C)
    bufs = bufs + [np.roll(bufs[0], 17), np.roll(bufs[1], 29), bufs[0] * np.float32(0.5)]      # 5 buffers
    lo, hi = sharding.channel_range(rank, world, C)
    tuner = _tuner(rc, kind, centres)
    ring = sharding.SpectrumRing(tuner, N, C)
    dev = [torch.from_numpy(b).cuda() for b in bufs]
    gathered = []
    for j in range(min(ring.lookahead, len(bufs))):
        ring.submit(j, dev[j] if ring.owner(j) == rank else None)
    for i in range(len(bufs)):
        j = i + ring.lookahead
        if j < len(bufs):
            # buffer j of an owner may also arrive from the HOST (page-locked or not): the ring stages it
            src = torch.from_numpy(bufs[j]) if j == 3 else dev[j]
            ring.submit(j, src if ring.owner(j) == rank else None)
        ring.acquire(i)
        block = tuner.run_all()
        assert block.shape[0] == hi - lo
        gathered.append(sharding.gather_audio(torch.from_numpy(block), C, dst=0))
    dist.barrier()
    np.save(out_path + ".slots%d.npy" % rank, np.array([ring.full_slots, ring.window_slots, ring.slot_bytes()]))
    ring.drain()
    ring.close()
    # a window is attached no longer: the tuner loads and runs on its own storage again
    tuner.load(dev[0])
    assert tuner.run_all().shape[0] == hi - lo
    if rank == 0:
        np.save(out_path, np.stack([g.numpy() for g in gathered]))
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("kind,C", [("WBFM", 8), ("MFM", 8)])
def test_rotating_fft_owner_equals_single_process(tmp_path, kind, C):
    """sharding.SpectrumRing with the real tuner: two ranks on one GPU take turns running the wideband FFT; the rank
    that did not run it receives only the bins its channels read (rcfm_tuner_window / _attach_spectrum / _adopt) and
    must produce bit-identical audio -- five buffers, two in flight, state carried per channel."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "ring.npy")
    mp.spawn(_ring_worker, args=(2, _free_port(), kind, C, out), nprocs=2, join=True)
    got = np.load(out)
    # each half of the band lies clear of the spectrum's ends: both ranks keep whole slots only for the buffers they own
    # (two of the three in flight) and window-sized slots (rcfm_tuner_attach_window) for the others -- each window
    # slot holds about half a spectrum here (two ranks); at eight ranks it is an eighth
    for r in range(2):
        full, win, nbytes = np.load(out + ".slots%d.npy" % r)
        assert (full, win) == (2, 3) and 2 * 8 * N < nbytes < 4 * 8 * N, (r, full, win, nbytes)
    _paths()
    import radiocore as rc
    centres, bufs = _inputs(kind, C)
    bufs = bufs + [np.roll(bufs[0], 17), np.roll(bufs[1], 29), bufs[0] * np.float32(0.5)]
    tuner = _tuner(rc, kind, centres)
    for b, x in enumerate(bufs):
        tuner.load(x)
        want = tuner.run_all()
        assert np.array_equal(got[b], want), (kind, b, float(np.max(np.abs(got[b] - want))))


def test_c_abi_gather_world_of_one():
    """rcfm_comm_* / rcfm_gather_audio (RCCL bound at run time): the degenerate one-rank communicator on this
    GPU -- unique id, init, a gather that must reproduce the block, destroy.  (RCCL refuses two ranks on one
    device, so N > 1 of this entry point runs only on a multi-GPU node; the N > 1 protocol of the path is
    covered above with torch.distributed.)"""
    import ctypes
    _paths()
    import torch
    from radiocore._internal import hip
    lib = hip.lib()
    hip.torch()
    token = (ctypes.c_char * 128)()
    hip.check(lib.rcfm_comm_unique_id(ctypes.cast(token, ctypes.c_void_p)))
    comm = ctypes.c_void_p()
    hip.check(lib.rcfm_comm_init_rank(1, 0, ctypes.cast(token, ctypes.c_void_p), ctypes.byref(comm)))
    block = torch.randn(3, 4800, 2, device="cuda")
    out = torch.zeros_like(block)
    hip.check(lib.rcfm_gather_audio(comm, 0, hip.ptr(block), block.numel(), hip.ptr(out), hip.stream()))
    torch.cuda.synchronize()
    assert torch.equal(out, block)
    with pytest.raises(IndexError):
        hip.check(lib.rcfm_gather_audio(comm, 1, hip.ptr(block), block.numel(), hip.ptr(out), hip.stream()))
    hip.check(lib.rcfm_comm_destroy(comm))
