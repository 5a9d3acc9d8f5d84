"""GPU: the RCCL ("nccl") branches of the multi-GPU path at world size ONE.

No pool box has more than one GPU, and RCCL refuses two ranks on one device, so N > 1 over xGMI only ever runs in the
driver's multi-GPU bench.  A one-rank communicator is legal, though: these tests execute -- on the real backend --
every call the N > 1 path makes that the gloo dry runs (tests/test_hip_sharded.py, tests/test_bench_multirank.py)
replace: init_process_group("nccl", device_id=...), sharding.gather_audio(..., async_op=True) writing into views of
the [C, A, ch] result on RCCL's own stream, the barrier, the max-over-ranks all_reduce on a device tensor, and
bench.py's double-buffered step loop launched by torch.distributed.run.

Publish step in the reference: examples/multi_fm_server.py:103-106."""

import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, have_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not have_gpu(), reason="needs an MI355X")]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    os.environ["RCFM_GATHER_FORCE_COLLECTIVE"] = "1"      # a one-rank group would otherwise return `local` untouched
    for p in (ROOT, os.path.join(ROOT, "radio-core_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    from radiocore.tools import sharding
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    C, A, ch = 6, 4800, 2
    blocks = [torch.randn(C, A, ch, device="cuda") for _ in range(3)]
    outs = [torch.zeros(C, A, ch, device="cuda") for _ in range(2)]
    side = torch.cuda.Stream()
    results = []
    pending = [None, None]
    for i, b in enumerate(blocks):                      # double-buffered like bench.py's step()
        slot = i % 2
        if pending[slot] is not None:
            pending[slot].wait()
        pending[slot] = sharding.gather_audio(b, C, dst=0, out=outs[slot], async_op=True)
        with torch.cuda.stream(side):                   # unrelated work while the collective is in flight
            _ = (b * 2).sum()
        if i >= 1:
            results.append(outs[(i - 1) % 2].clone() if pending[(i - 1) % 2].wait() is not None else None)
    got_last = pending[(len(blocks) - 1) % 2].wait()
    assert got_last is outs[(len(blocks) - 1) % 2]        # in place: the collective wrote into `out`
    torch.cuda.synchronize()
    ok = torch.equal(got_last, blocks[-1]) and all(torch.equal(r, blocks[i]) for i, r in enumerate(results))
    # the synchronous form, the uneven-form code path (out=None) and the timing reduction
    full = sharding.gather_audio(blocks[0], C, dst=0)
    ok = ok and torch.equal(full, blocks[0])
    t = torch.tensor([1.25], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    torch.cuda.synchronize()
    np.save(out_path, np.array([int(ok), float(t.item())]))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_gather_audio_on_a_one_rank_rccl_group(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "ok.npy")
    mp.spawn(_worker, args=(_free_port(), out), nprocs=1, join=True)
    ok, t = np.load(out)
    assert ok == 1 and t == 1.25


@pytest.mark.timeout(900)
@pytest.mark.parametrize("parallelism", ["replicated", "rotating"])
def test_bench_one_rank_through_the_rccl_code_path(parallelism):
    """bench.py --gpus 1 launched by torch.distributed.run exactly as the driver launches N > 1, forced down the
    N > 1 branches (RCFM_BENCH_FORCE_DIST=1) on the nccl backend; `rotating` adds the spectrum ring (slots, the owner's
    FFT on its own stream, attach / adopt) -- with one rank every buffer is its own, so no transfer is posted."""
    env = dict(os.environ, RCFM_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RCFM_BENCH_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4",
           "--warmup", "1", "--config", "small", "--cpu-channels", "0", "--no-extras", "--parallelism", parallelism]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=800)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 1 and r["steps"] == 4 and r["value"] > 0
    g = r["gather_check"]
    assert g["finite"] and g["own_block_equal"] and g["blocks_with_audio"] == g["blocks"] == 1
    assert r["channel_stage_value"]["value"] > 0


def _bins_worker(rank, out_path):
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    for p in (ROOT, os.path.join(ROOT, "radio-core_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import ctypes
    import torch
    torch.cuda.set_device(0)
    from radiocore._internal import hip
    lib = hip.lib()
    token = (ctypes.c_ubyte * 128)()
    hip.check(lib.rcfm_comm_unique_id(token))
    comm = ctypes.c_void_p()
    hip.check(lib.rcfm_comm_init_rank(1, 0, token, ctypes.byref(comm)))
    n = 1 << 20
    a = torch.view_as_complex(torch.randn(n, 2, device="cuda"))
    b = torch.zeros_like(a)
    segs = [(4096, 300000), (700001, n)]                   # two pieces, as a wrapping window has
    hip.check(lib.rcfm_comm_group_start(comm))
    for lo, hi in segs:
        hip.check(lib.rcfm_send_bins(comm, 0, hip.ptr(a[lo:hi]), hi - lo, hip.stream()))
        hip.check(lib.rcfm_recv_bins(comm, 0, hip.ptr(b[lo:hi]), hi - lo, hip.stream()))
    hip.check(lib.rcfm_send_bins(comm, 0, None, 0, hip.stream()))    # a rank without channels: nothing to move
    hip.check(lib.rcfm_comm_group_end(comm))
    torch.cuda.synchronize()
    ok = all(torch.equal(a[lo:hi], b[lo:hi]) for lo, hi in segs)
    ok = ok and float(b[:4096].abs().max()) == 0.0 and float(b[300000:700001].abs().max()) == 0.0
    codes = [lib.rcfm_send_bins(comm, 0, hip.ptr(a), 16, hip.stream()),       # to itself outside a group: refused
             lib.rcfm_recv_bins(comm, 1, hip.ptr(b), 16, hip.stream()),       # no such rank
             lib.rcfm_comm_group_end(comm)]                                   # no group open
    hip.check(lib.rcfm_comm_destroy(comm))
    np.save(out_path, np.array([int(ok)] + codes))


@pytest.mark.timeout(600)
def test_send_recv_bins_on_a_one_rank_communicator(tmp_path):
    """rcfm_comm_group_start / rcfm_send_bins / rcfm_recv_bins / rcfm_comm_group_end (the rotating owner's hand-over for
    hosts without torch.distributed): a rank sending to itself inside one group is a device copy -- the whole protocol
    runs on a one-GPU box (examples/c_host.c does the same from C)."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "bins.npy")
    mp.spawn(_bins_worker, args=(out,), nprocs=1, join=True)
    ok, send_self, recv_bad_peer, end_without_start = np.load(out)
    assert ok == 1
    assert send_self == -5 and recv_bad_peer == -2 and end_without_start == -5     # RCFM_ERR_STATE, _INDEX, _STATE
