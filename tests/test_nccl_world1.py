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
