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
    assert r["rccl_ranks"] == 0                                      # a gloo dry run: no RCCL communicator
    assert "rotating" not in r                                       # one partitioning was asked for
    assert r["stages"]["tuner_fft_N"]["ms"] > 0 and r["roofline"]["stage"] in r["stages"]
    assert r["channel_stage_value"]["value"] > 0
    assert r["pcie_inclusive"]["value"] > 0                          # the N > 1 line always carries the host-fed rate
    if parallelism == "rotating":
        assert r["amdahl_bound_speedup"] == 2.0 and "rotating FFT owner" in r["config"]["parallelism"]
        ro = r["rotating_owner"]
        assert ro["lookahead"] == 2 and ro["ffts_per_rank_per_buffer"] == 0.5
        # three buffers in flight: whole slots for all of them, or two whole ones (the owned buffers) + three windows
        assert (ro["full_slots"], ro["window_slots"]) in ((3, 0), (2, 3))
        assert ro["spectrum_slots"] == ro["full_slots"] + ro["window_slots"] and ro["slot_bytes"] > 0
        assert 0 < ro["bytes_sent_per_owned_buffer"] <= 8 * r["config"]["wideband_samples"]
    else:
        assert 1.0 < r["amdahl_bound_speedup"] < 2.0
    g = r["gather_check"]
    assert g["finite"] and g["own_block_equal"] and g["blocks_with_audio"] == g["blocks"] == 2
    # the same launch timed ONE rank first (rank 0 alone) and kept its audio: this partitioning reproduced it bit for bit
    assert r["n1"]["value"] > 0 and r["n1"]["steps"] == 3
    assert abs(r["speedup_vs_n1"] - r["value"] / r["n1"]["value"]) <= 1e-3 * r["speedup_vs_n1"]
    sc = r["self_check"]
    assert sc["bit_identical"] is True and sc["max_abs_diff"] == 0.0 and sc["peak"] > 1e-3 and sc["channels"] == r["config"]["channels"]
    assert "partitionings" not in r                                  # one leg: nothing to choose between
    # where each rank's time went: one row per rank, the keys a first multi-GPU run is debugged with
    rows = r["per_rank"]
    assert [row["rank"] for row in rows] == [0, 1]
    for row in rows:
        assert {"step_ms", "fft_ms", "send_ms", "wait_ms", "chan_ms", "other_ms", "channels"} <= set(row)
        assert row["step_ms"] > 0 and row["fft_ms"] > 0 and row["chan_ms"] > 0 and row["channels"] == r["config"]["channels_per_gpu"]
        if parallelism == "rotating":
            assert row["owned_buffers"] >= 1 and row["send_ms"] >= 0 and row["wait_ms"] >= 0


@pytest.mark.timeout(300)
def test_a_stalled_rank_fails_loudly_instead_of_hanging():
    """RCFM_BENCH_TIMEOUT: rank 1 is started without its peer ever joining the rendezvous of the process group's first
    collective... simulated here by a one-rank launch that claims a world of 2 it cannot reach: the process must end
    within the limit with a JSON error line on stderr naming rank and phase, not hang."""
    env = dict(os.environ, RCFM_BENCH_DEVICE="0", RCFM_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0",
               RCFM_BENCH_TIMEOUT="8", RANK="0", WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--config", "small"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=240)
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]        # no contract line from a failed run
    err = [ln for ln in out.stderr.splitlines() if ln.startswith("{")]
    assert err, out.stderr[-2000:]
    msg = json.loads(err[-1])
    assert msg["rank"] == 0 and msg["world"] == 2 and "no progress" in msg["error"] and msg["phase"]


@pytest.mark.timeout(900)
def test_plain_python_launch_starts_its_own_ranks_and_reports_both_partitionings():
    """`python bench.py --gpus 2` with NO launcher around it (how the driver starts N = 1): bench.py re-executes itself
    under torch.distributed.run with two ranks, rank 0 prints ONE line with n_gpus == 2 whose
    `value` is the faster verified partitioning, the other one in its own block of the same line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(RCFM_BENCH_DEVICE="0", RCFM_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--config", "small"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=800)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and len(r["per_rank"]) == 2
    # the chosen-`value` rule: both legs verified -> the FASTER one is the line, the other keeps its own block
    pt = r["partitionings"]
    assert pt["verified"] == {"replicated": True, "rotating": True}
    fastest = max(pt["value"], key=pt["value"].get)
    assert pt["published"] == fastest and r["value"] == pt["value"][fastest]
    other = "rotating" if fastest == "replicated" else "replicated"
    assert other in r and fastest not in r
    assert ("rotating FFT owner" if fastest == "rotating" else "wideband FFT replicated") in r["config"]["parallelism"]
    assert abs(r["value"] - r["config"]["wideband_samples"] / (r["ms_per_step"] * 1e-3) / 1e6) <= 0.01 * r["value"]
    blk = r[other]
    assert "error" not in blk, blk
    assert blk["value"] == pt["value"][other] and blk["ms_per_step"] > 0 and len(blk["per_rank"]) == 2
    assert abs(blk["vs_published"] - blk["value"] / r["value"]) <= 1e-3
    assert ("rotating FFT owner" if other == "rotating" else "wideband FFT replicated") in blk["parallelism"]
    rot, rep = (r, blk) if fastest == "rotating" else (blk, r)
    assert rot["amdahl_bound_speedup"] == 2.0 and 1.0 < rep["amdahl_bound_speedup"] < 2.0
    assert rot["rotating_owner"]["lookahead"] == 2
    for row in rot["per_rank"]:
        assert row["owned_buffers"] >= 1
    for leg in (rot, rep):
        assert leg["gather_check"]["finite"] and leg["gather_check"]["blocks_with_audio"] == 2
        assert leg["self_check"]["bit_identical"] is True            # one buffer through each: the single-GPU audio
    # one GPU in the same launch, and the published speed-up against it
    assert r["n1"]["value"] > 0 and abs(r["speedup_vs_n1"] - r["value"] / r["n1"]["value"]) <= 1e-3 * r["speedup_vs_n1"]
    assert r["rccl_ranks"] == 0 and r["cpu_baseline"] is None


@pytest.mark.timeout(300)
def test_a_failing_rotating_leg_still_publishes_the_replicated_line():
    """The second leg of a 'both' launch dies (RCFM_BENCH_FAIL_ROTATING=1 raises on rank 1 at its start; rank 0 then
    waits for a peer that never comes and its watchdog fires): the run ends within the leg's own time limit with the
    replicated line on stdout and the diagnosis in its `rotating` block."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(RCFM_BENCH_DEVICE="0", RCFM_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0",
               RCFM_BENCH_FAIL_ROTATING="1", RCFM_BENCH_TIMEOUT_ROTATING="10")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--config", "small",
           "--no-pcie"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=280)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:] + out.stderr[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["value"] > 0
    assert "error" in r["rotating"] and r["rotating"]["rank"] == 0 and r["rotating"]["phase"]
    assert "wideband FFT replicated" in r["config"]["parallelism"] and "partitionings" not in r
    assert r["self_check"]["bit_identical"] is True and r["speedup_vs_n1"] > 0
