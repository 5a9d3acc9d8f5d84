"""GPU: bench.py's N > 1 code (channel sharding, windowed wideband FFT per rank, gather, max-over-ranks timing, the
extra fields of the N > 1 line) as a dry run on ONE device: two ranks launched exactly as the driver launches them
(`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`), with RCFM_BENCH_DEVICE=0 and
RCFM_BENCH_BACKEND=gloo because RCCL refuses two ranks on one GPU.  The timing of such a run means nothing; the
protocol, the JSON contract and the gathered audio do."""

import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT, have_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not have_gpu(), reason="needs an MI355X")]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.timeout(900)
@pytest.mark.parametrize("parallelism", ["replicated", "rotating"])
def test_two_rank_dry_run_prints_one_contract_line(parallelism):
    env = dict(os.environ, RCFM_BENCH_DEVICE="0", RCFM_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "1", "--config", "small", "--parallelism", parallelism]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=800)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                      # rank 0 only, one line
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["warmup"] == 1
    assert r["unit"] == "Msamples/s" and r["value"] > 0 and r["scaling"] == "strong" and r["vs_baseline"] is None
    assert r["config"]["channels_per_gpu"] == r["config"]["channels"] // 2
    assert abs(r["value"] - r["config"]["wideband_samples"] / (r["ms_per_step"] * 1e-3) / 1e6) <= 0.01 * r["value"]
    assert r["roofline"]["traffic"] is None                          # the committed PMC passes describe one GPU
    assert r["cpu_baseline"] is None                                 # rank 0 at N = 1 only
    assert r["channel_stage_value"]["value"] > 0
    assert r["pcie_inclusive"]["value"] > 0                          # the N > 1 line always carries the host-fed rate
    if parallelism == "rotating":
        assert r["amdahl_bound_speedup"] == 2.0 and "rotating FFT owner" in r["config"]["parallelism"]
        ro = r["rotating_owner"]
        assert ro["lookahead"] == 2 and ro["spectrum_slots"] == 3 and ro["ffts_per_rank_per_buffer"] == 0.5
        assert 0 < ro["bytes_sent_per_owned_buffer"] <= 8 * r["config"]["wideband_samples"]
    else:
        assert 1.0 < r["amdahl_bound_speedup"] < 2.0
    g = r["gather_check"]
    assert g["finite"] and g["own_block_equal"] and g["blocks_with_audio"] == g["blocks"] == 2
